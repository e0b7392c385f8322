/*
 * dm_sampler.h — C ABI of libdm_sampler.so: MI355X (gfx950) reverse-diffusion sampling.
 *
 * The reference (dj-shark31/diffusion_model) is pure Python/PyTorch and has no FFI layer
 * (SURVEY 8(b)); its hot path sits behind the Python classes UNetMini / DDPMScheduler /
 * AncestralSampler / DDIMSampler.  This header is the boundary a binding (ctypes here,
 * see diffusion_model_amd/_lib.py and INTEGRATION.md) uses to replace the tensor work
 * inside those classes.  Every entry point cites the reference code it replaces.
 *
 * Conventions
 *  - All pointers named *_dev are device (HBM) pointers, fp32 unless said otherwise;
 *    *_host are host pointers.  `stream` is a hipStream_t passed as void* (NULL = default).
 *  - No entry point allocates device memory or synchronises, except dm_unet_create /
 *    dm_unet_destroy (host-only state) and graph instantiation inside dm_sample.
 *  - Return value: 0 on success, non-zero on error; dm_last_error() returns the message
 *    (thread-local).  Shape/argument errors map to the RuntimeError/ValueError the
 *    reference's torch ops would raise (host side re-raises).
 *  - Activations inside the library are NHWC fp32 ("pixel-major": one pixel's channels
 *    are contiguous).  Images at the boundary (x, eps) are NCHW exactly like the reference.
 */
#ifndef DM_SAMPLER_H
#define DM_SAMPLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dm_unet dm_unet; /* opaque: geometry + packed-weight offsets + graph cache */

/* Number of parameter tensors of UNetMini in state_dict order (diffusion/model.py:82-115;
 * key list in SURVEY 8(a) row M0): 4 time_mlp + 2 init_conv + 6 blocks x 12 + 6 bottleneck
 * + 4 final_conv = 88. */
#define DM_UNET_NUM_PARAMS 88

const char* dm_last_error(void);
/* ABI version of this header; dm_version() returns the value the library was built with.  A binding checks
 * them for equality before it calls anything else (2: dm_unet_profile takes the length of its result arrays,
 * DM_K_COUNT grew, dm_sample accepts host pointers and sizes its workspace with dm_sample_workspace_bytes;
 * 3: dm_sample_chunked, the fused low-resolution kernel class, the any-geometry kernel classes;
 * 4: dm_sample's workspace holds one carve per parallel sub-batch chain (dm_sample_workspace_bytes grew for large batches),
 *    dm_sample_branches, dm_stream_create / dm_stream_destroy). */
#define DM_ABI_VERSION 4
int dm_version(void);
/* sha256 over the compiler flags and sources this library was built from (__graft_entry__.build_key); "unkeyed" for a
 * hand build.  build() compares it with the tree it runs in, so a library built with other flags is not reused. */
const char* dm_build_key(void);

/* ---- model container: replaces UNetMini.__init__ (diffusion/model.py:72-118) ---------- */
int dm_unet_create(dm_unet** out, int in_channels, int out_channels, int model_channels,
                   int time_emb_dim);
/* attention != 0 adds the BUILD-DEFINED self-attention blocks of BASELINE configs[4] (no reference
 * counterpart: model.py:70 "No attention layers"): GroupNorm8 -> 1x1 qkv -> single-head
 * softmax(QK^T/sqrt(C))V over the (H/4)x(W/4) tokens -> 1x1 proj -> residual, applied to down2's
 * output and to up1's output.  12 more parameter tensors follow the 88, for attn_down then
 * attn_up: norm.weight, norm.bias, qkv.weight [3C,C,1,1], qkv.bias, proj.weight [C,C,1,1], proj.bias. */
int dm_unet_create_ex(dm_unet** out, int in_channels, int out_channels, int model_channels,
                      int time_emb_dim, int attention);
int dm_unet_num_params(const dm_unet* m);
/* bf16 != 0: activations between kernels are 16-bit (bfloat16; IEEE half where the consumer
 * normalises the tensor), the packed conv weights are bfloat16 (hi, lo) pairs, MFMA is
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation - except the layers behind a GroupNorm in
 * down1, down2.conv2 and up3 and the fused low-resolution kernel (raw tensors scaled by a power
 * of two there), whose operands (activation and ONE weight per product) are IEEE half on
 * v_mfma_f32_32x32x16_f16; GroupNorm statistics, the SiLU transform, the time embedding, eps and
 * the sampler update stay fp32 (BASELINE configs[3]; north_star tolerance
 * |eps - eps_fp32| <= 2e-2, measured <= 0.66e-2).  Call BEFORE dm_unet_pack_weights; drops cached graphs. */
int dm_unet_set_precision(dm_unet* m, int bf16);
void dm_unet_destroy(dm_unet* m);

/* Size (in floats) of the packed weight blob for this geometry. */
size_t dm_unet_packed_floats(const dm_unet* m);

/* Repack the 88 parameter tensors (device pointers, torch layouts: Conv2d OIHW,
 * ConvTranspose2d IOHW, Linear [out,in]) into the kernels' layout (MFMA B-fragment order
 * for the conv weights).  `freqs_dev` = the time_emb_dim/2 sinusoid frequencies
 * exp(-i*ln(1e4)/(half-1)) (diffusion/model.py:19-21) or NULL to have the library compute
 * them.  Re-run whenever the parameters change (EMA swap, load_state_dict). */
int dm_unet_pack_weights(dm_unet* m, const float* const* params_dev, int nparams,
                         const float* freqs_dev, float* packed_dev, void* stream);

/* Workspace bytes needed by dm_unet_forward / dm_sample for batch B of HxW images and a
 * time-step table of `nsteps` entries (1 for a single forward). */
size_t dm_unet_workspace_bytes(const dm_unet* m, int B, int H, int W, int nsteps);

/* ---- denoiser forward: replaces UNetMini.forward (diffusion/model.py:120-151) ---------
 * x_dev [B,Cin,H,W] NCHW, t_dev int64[nt] with nt==1 (0-dim timestep, sampler.py:71,167)
 * or nt==B (training-style, train.py:116-125).  cond_dev: optional [nt,E] embedding added
 * to the time embedding (build-defined class conditioning hook) or NULL.  eps_dev [B,Cout,H,W]. */
int dm_unet_forward(dm_unet* m, const float* packed_dev, const float* x_dev,
                    const int64_t* t_dev, int nt, const float* cond_dev, float* eps_dev,
                    int B, int H, int W, void* workspace_dev, size_t workspace_bytes,
                    void* stream);

/* ---- dm_unet_forward for callers that cannot allocate (NO counterpart in the reference).
 * For code under hipGraph capture, tight serving loops and harnesses that forbid hipMalloc: the
 * same forward as dm_unet_forward with the workspace taken from a file-scope device array of
 * DM_NOALLOC_SCRATCH_BYTES that the library holds on every GPU it runs on until the process ends.
 * Calls no hipMalloc/hipFree or other allocating routine.  Requests whose workspace exceeds the
 * array (it holds about 68 images of 32x32 at C=64) are rejected with an error.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two threads) race.
 * dm_init() performs the one-time per-device set-up (hipGetSymbolAddress) for the CURRENT
 * device; call it before capture starts - otherwise the first call on a device does it. */
#define DM_NOALLOC_SCRATCH_BYTES (64u << 20)
int dm_init(void);
size_t dm_noalloc_scratch_bytes(void);
int dm_unet_forward_noalloc(dm_unet* m, const float* packed_dev, const float* x_dev,
                            const int64_t* t_dev, int nt, const float* cond_dev, float* eps_dev,
                            int B, int H, int W, void* stream);

/* ---- time embedding: replaces SinusoidalPositionEmbeddings.forward + UNetMini.time_mlp +
 * the six Block.time_mlp heads (diffusion/model.py:18-30, 82-87, 56).
 * temb_dev [nt,E] (may be NULL), heads_dev [nt,6,C] = SiLU(Linear_k(temb)). */
int dm_time_embed(dm_unet* m, const float* packed_dev, const int64_t* t_dev, int nt,
                  const float* cond_dev, float* temb_dev, float* heads_dev, void* stream);

/* ---- sampler step: replaces DDPMScheduler.step (diffusion/schedulers.py:94-131) and
 * DDIMSampler.step (diffusion/sampler.py:102-134).
 * coef[5] = { sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev), sqrt(pv_t)*eta }.
 * out = c2*((x - c0*eps)/c1) + c3*eps (+ c4*noise when noise_dev != NULL), evaluated in the
 * reference's operation order without FMA contraction.  out_dev may alias x_dev. */
int dm_sampler_step(const float* x_dev, const float* eps_dev, const float* noise_dev,
                    const float* coef_host, float* out_dev, size_t count, void* stream);

/* Per-step coefficient table from the schedule tables (host, fp32 arithmetic as torch):
 * coefs_host[n][8] (5 used, 3 pad).  Tables are DDPMScheduler.alphas_cumprod,
 * .alphas_cumprod_prev, .posterior_variance (diffusion/schedulers.py:41-54), length T. */
int dm_sampler_coefs(const float* acp_host, const float* acp_prev_host, const float* pv_host,
                     int T, const int64_t* timesteps_host, int n, float eta, float* coefs_host);

/* ---- whole reverse-diffusion loop: replaces AncestralSampler.__call__ /
 * DDIMSampler.__call__ (diffusion/sampler.py:45-76, 136-172).
 * x_dev [B,Cin,H,W]: x_T on entry, final sample on exit (requires Cin == Cout).
 * timesteps int64[n] (ascending, as DDPMScheduler.set_timesteps produces),
 * coefs float[n][8] (dm_sampler_coefs),
 * noise: NULL (eta == 0) or [n,B,Cin,H,W] host-drawn noise in loop order.
 * x, timesteps, coefs, cond and noise are COPIED into the workspace on `stream` (hipMemcpyDefault:
 * each may be a device pointer, or a host pointer - noise is staged 50 steps at a time, so a host
 * tensor never needs a device copy of all n steps), and the result is copied back to x_dev.  A
 * captured graph therefore holds workspace addresses only and is reused by every later call with
 * the same model, workspace and (B, H, W, n, guided, noise or not, guidance).
 * Workspace size: dm_sample_workspace_bytes.
 * cond_dev / guidance: build-defined classifier-free guidance (cond_dev NULL = off):
 *   cond_dev [2B,E] = embeddings for [labels ; null label]; the denoiser runs on the doubled
 *   batch and eps = eps_u + guidance*(eps_c - eps_u).
 * use_graph != 0: the step loop is captured once into a hipGraph (chunks of <= 50 steps
 * that read a device-side step counter) and replayed; 0: eager launches on `stream`. */
int dm_sample(dm_unet* m, const float* packed_dev, float* x_dev, int B, int H, int W,
              const int64_t* timesteps, const float* coefs, int n,
              const float* noise, const float* cond, float guidance,
              void* workspace_dev, size_t workspace_bytes, int use_graph, void* stream);

/* dm_sample with a per-chunk callback (sampler.py:65-66,156-157: the reference ticks a tqdm bar per step and
 * draws `randn_like` per step INSIDE the loop, schedulers.py:125-129).  The loop is enqueued in chunks of <= 50
 * steps (one graph launch each); `chunk_fn(user, first, count)` is called on the calling thread before steps
 * [first, first + count) are enqueued - everything before `first` already is - so a binding can (a) advance its
 * progress bar by the chunk length and (b) hand over the noise of that chunk only: with_noise != 0 and
 * noise_all == NULL takes the pointer the callback returns ([count][B,Cin,H,W], host or device; it must stay
 * valid until the copy enqueued right after the callback has completed - a binding that reuses a ring of k
 * staging buffers waits, inside the callback, for an event it recorded k - 1 callbacks earlier).  The draw of
 * chunk k+1 then overlaps the GPU's work on chunk k.  noise_all != NULL: the whole [n,...] tensor as in
 * dm_sample (the callback's return value is ignored).  chunk_fn may be NULL. */
typedef const float* (*dm_chunk_fn)(void* user, int first_step, int step_count);
int dm_sample_chunked(dm_unet* m, const float* packed_dev, float* x_dev, int B, int H, int W,
                      const int64_t* timesteps, const float* coefs, int n,
                      const float* noise_all, int with_noise, dm_chunk_fn chunk_fn, void* chunk_user,
                      const float* cond, float guidance,
                      void* workspace_dev, size_t workspace_bytes, int use_graph, void* stream);

/* Workspace bytes dm_sample needs: dm_unet_workspace_bytes for the (doubled when `guided`) batch
 * and an n-step table, plus the staging area for one graph chunk of noise when `with_noise`;
 * where dm_sample runs the batch as parallel sub-batch chains (B*H*W >= 2^18 pixels with
 * use_graph, or DM_SPLIT in the environment), one such carve per chain if that is more. */
size_t dm_sample_workspace_bytes(const dm_unet* m, int B, int H, int W, int n, int guided,
                                 int with_noise);

/* A non-blocking HIP stream of the caller's own on the current device (hipStreamCreateWithFlags), and its release.
 * No reference counterpart.  For hosts whose framework hands out POOLED streams: torch.cuda.Stream() cycles through 32
 * shared streams per device, so a pooled stream used for graph capture can be the very stream another component (another
 * thread) launches into, which invalidates the capture; the Python samplers capture on a stream from here. */
int dm_stream_create(void** stream_out);
int dm_stream_destroy(void* stream);

/* Number of parallel sub-batch chains dm_sample / dm_sample_chunked run a batch of B images as when
 * `use_graph` is set (1 = one chain; the samples are the same bits either way).  No reference
 * counterpart: the reference's loop (diffusion/sampler.py:45-76) is one chain of B images. */
int dm_sample_branches(const dm_unet* m, int B, int H, int W, int guided);

/* DDPMScheduler.step with a [B] timestep tensor (diffusion/schedulers.py:113-115: the table
 * lookups become `.view(-1,1,1,1)` per-image coefficients) in ONE launch: coefs_dev float[B][8]
 * (dm_sampler_coefs rows, uploaded), count_per_image = C*H*W elements; noise_dev NULL or
 * [B*count_per_image]. */
int dm_sampler_step_batched(const float* x_dev, const float* eps_dev, const float* noise_dev,
                            const float* coefs_dev, float* out_dev, int B, size_t count_per_image,
                            void* stream);

/* ---- post-processing: replaces `(samples + 1) / 2; clamp(0, 1)` (diffusion/sample.py:117-119).
 * out_f32_dev and/or out_u8_dev (uint8 = trunc(v*255 + 0.5), torchvision save_image's rule);
 * either may be NULL.  out_f32_dev may alias x_dev. */
int dm_postprocess(const float* x_dev, float* out_f32_dev, unsigned char* out_u8_dev,
                   size_t count, void* stream);

/* ---- class embedding: replaces ClassConditioner.forward (diffusion/conditioning.py:39-55):
 * out[n] = Linear2(SiLU(Linear1(table[labels[n]]))).  table [num_rows,E]; w1,w2 [E,E]; b1,b2 [E];
 * labels int64[n] in [0,num_rows); out [n,E].  Feeds dm_unet_forward / dm_sample's cond_dev. */
int dm_class_embed(const float* table_dev, int num_rows, int E, const float* w1_dev,
                   const float* b1_dev, const float* w2_dev, const float* b2_dev,
                   const int64_t* labels_dev, int n, float* out_dev, void* stream);

/* ---- measurement aid (no reference counterpart): runs `reps` denoiser evaluations + sampler
 * updates with a hipEvent pair (recorded on `stream`) around every kernel launch and returns,
 * per kernel class, the launch count, the summed device time (ms) and the summed ALGORITHMIC
 * FLOPs and bytes of those launches.  Synchronises the stream.  bench.py's roofline leg. */
enum {
  DM_K_INIT = 0,      /* init_conv_kernel                                   (HBM-bound) */
  DM_K_CONV3_128 = 1, /* conv_mfma_kernel<2,0,64>: 3x3 convs, 128-pixel tiles (MFMA)    */
  DM_K_CONV3_64 = 2,  /* conv_mfma_kernel<1,0,64>: 3x3 convs, 64-pixel tiles  (MFMA)    */
  DM_K_DOWN = 3,      /* conv_mfma_kernel<1,1,32>: 4x4 stride-2 convs         (MFMA)    */
  DM_K_UP_128 = 4,    /* conv_mfma_kernel<2,2,64>: transposed 4x4 s2          (MFMA)    */
  DM_K_UP_64 = 5,     /* conv_mfma_kernel<1,2,64>                              (MFMA)    */
  DM_K_FINAL = 6,     /* final_conv_kernel                                  (HBM-bound) */
  DM_K_TIME = 7,      /* time_embed_kernel                                              */
  DM_K_STEP = 8,      /* sampler_step_kernel                                (HBM-bound) */
  DM_K_ATTN = 9,      /* attention block: 1x1 qkv/proj convs, QK^T / PV GEMMs, softmax  */
  DM_K_WINO = 10,     /* conv_mfma_kernel<2,5,64>: 3x3 convs as Winograd F(2x2,3x3)  (MFMA) */
  DM_K_X6_CONV3 = 11, /* conv_x6_kernel<.,0>: 3x3 convs on the bf16 pipe, 3-way split operands (6 MFMAs per fp32 product) */
  DM_K_X6_DOWN = 12,  /* conv_x6_kernel<.,1>: 4x4 stride-2 convs, same arithmetic                   */
  DM_K_X6_UP = 13,    /* conv_x6_kernel<.,4>: transposed 4x4 s2 convs, same arithmetic              */
  DM_K_GENERIC_CONV = 14, /* gconv_kernel<.>: every convolution of geometries outside the tuned set (any H, W % 8, any C % 8), fp32 MFMA */
  DM_K_GENERIC_NORM = 15, /* gstats_kernel + gnorm_kernel: GroupNorm + SiLU + temb as stand-alone passes, same geometries (HBM-bound) */
  DM_K_LOWRES = 16,       /* lowres_fused_kernel: the levels at and below 8x8 pixels of the 32x32 model in ONE launch, activations in LDS */
  DM_K_COUNT = 17
};
int dm_unet_profile(dm_unet* m, const float* packed_dev, float* x_dev, const int64_t* t_dev,
                    const float* coef_host, int B, int H, int W, void* workspace_dev,
                    size_t workspace_bytes, int reps, void* stream, int nk, int* counts, float* ms,
                    double* flops, double* bytes);  /* nk = length of the four result arrays, must equal DM_K_COUNT */

/* ---- single operators (NHWC), exported for the per-kernel parity tests ------------------
 * kind: 0 = Conv2d 3x3 pad 1 (model.py:42,45,47,99,102), 1 = Conv2d 4x4 stride 2 pad 1
 * (model.py:46), 2 = ConvTranspose2d 4x4 stride 2 pad 1 (model.py:43), 3 = Conv2d 1x1 (build-defined
 * attention block), 5 = the same Conv2d 3x3 pad 1 as kind 0 evaluated as Winograd F(2x2,3x3) in fp32
 * (H, W >= 8; what dm_unet_forward uses for its 3x3 layers from 8x8 up), 6 = the Conv2d 4x4 stride 2
 * of kind 1 on 128-pixel tiles with 16-channel K chunks (fp32, outputs of >= 128 pixels; same packed
 * weights as kind 1), 7 / 8 / 9 = kinds 0 / 1 / 2 computed on the bf16 matrix pipe with fp32-equivalent accuracy:
 * every fp32 operand as three bf16 terms, six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation
 * (conv_x6_kernel; own packed-weight format, 1.5 floats per weight; what dm_unet_forward uses for its
 * stride-2 and transposed convolutions and for the 3x3 convolutions at 4x4 pixels: DM_X6),
 * 10 / 11 / 12 = kinds 0 / 1 / 2 on the any-geometry kernel (gconv_kernel: any H, W; cin % 8 == 0; any cout; no fused
 * GroupNorm transform or statistics - see dm_groupnorm_silu): what dm_unet_forward uses whenever the image sides are
 * not powers of two or model_channels is outside {64,128,256,512} (model.py:72-73,120-151 accept any H, W % 8 == 0 and
 * any C % 8 == 0),
 * 13 / 14 / 15 = kinds 1 / 2 / 0 on the bf16-rate matrix pipe with every operand as TWO fp16 terms (three
 * v_mfma_f32_32x32x16_f16 per product, fp32 accumulation; packed weights: two fp16 planes scaled by a power of two and the
 * float that undoes it): as exact as kinds 7-9 for operands of order 1, which is what a GroupNorm in front guarantees - so
 * these kinds need the fused GroupNorm transform (stats_in_dev), or, kind 15 only, raw inputs together with their producers'
 * maxima (dm_debug_conv_amax) from which the image's power-of-two scale is taken; tiles of one image (>= 64 output pixels).
 * What dm_unet_forward uses by default for its stride-2 / transposed convolutions, every conv2, and every conv1 whose
 * producers wrote their maxima (DM_X6 bits 5, 6, 8). */
size_t dm_conv_packed_floats(int kind, int cin, int cout);
int dm_pack_conv_weight(int kind, const float* w_dev, int cin, int cout, float* packed_dev,
                        void* stream);
/* y = conv(concat(f(x1), x2)) + bias.  f = identity when stats_in_dev == NULL, else
 * f(x) = SiLU(GroupNorm8(x; gamma, beta)) + temb (Block.forward, model.py:54-62) with the
 * group statistics given as `pin` partial (mean, M2) pairs per (image, group):
 * stats_in_dev [B][pin][8][2], each over cnt_in elements.  temb_dev: NULL or [*,C1] with
 * per-image stride temb_stride (0 = shared by the batch).  stats_out_dev: NULL or
 * [B][pout][8][2] partial statistics of y (pout as dm_conv_stats_partials reports).
 * H, W are the INPUT spatial dims. */
int dm_conv_stats_partials(int kind, int H, int W);
/* x <- SiLU(GroupNorm8(x; gamma, beta)) + temb in place (Block.forward model.py:54-62; bottleneck model.py:100-101), NHWC
 * [B,H,W,C], any C % 8 == 0: the stand-alone form the any-geometry path uses (statistics kernel + elementwise kernel).
 * temb_dev: NULL or [*,C] with per-image stride temb_stride (0 = shared).  stats_dev: scratch, B*16 floats. */
int dm_groupnorm_silu(float* x_dev, const float* gamma_dev, const float* beta_dev, const float* temb_dev,
                      int temb_stride, float* stats_dev, int B, int H, int W, int C, void* stream);
/* Build-defined attention core (no reference counterpart: model.py:70 "No attention layers"; BASELINE configs[4]):
 * o[b,t,:] = softmax_s(q[b,t,:] . k[b,s,:] / sqrt(C)) v[b,s,:], single head, qkv_dev [B][T][3C] (q | k | v per token),
 * o_dev [B][T][C].  One fused kernel per (image, 64-query tile): QK^T on fp32 MFMA into a 64 x T score tile in LDS,
 * row softmax there, PV on fp32 MFMA; T in {64,128,256}, C in {64,128,256,512}. */
int dm_attention(const float* qkv_dev, float* o_dev, int B, int T, int C, void* stream);
/* final_conv (model.py:111-115,151): out[B,Cout,H,W] (NCHW) = Conv1x1(SiLU(GroupNorm8(u))), u NHWC
 * [B,H,W,C] with statistics partials as above; w [Cout,C], bias [Cout]. */
int dm_final_conv(const float* u_dev, const float* stats_dev, int pin, float cnt_in,
                  const float* gamma_dev, const float* beta_dev, const float* w_dev,
                  const float* bias_dev, float* out_dev, int B, int H, int W, int C, int Cout,
                  void* stream);
/* Test hook of the fused low-resolution kernel (dm_kernels_lowres.hip keeps the outputs of down3, the bottleneck and up1
 * in LDS: model.py:140,143,146): while taps_dev != NULL every later forward that takes that kernel also writes them, raw
 * fp32 NHWC, to taps_dev[B][6144] = down3 [4,4,64] | bottleneck [4,4,64] | up1 [8,8,64].  NULL switches it off. */
void dm_debug_lowres_taps(float* taps_dev);
/* Test hook: maxima buffers of the NEXT dm_conv2d call (kinds 7-9, 13-15, tiles of one image), cleared by that call.
 * amax_out_dev: NULL or [B][dm_conv_stats_partials(kind, H, W) * (cout / 64) * 4] - the call writes the per-wave maxima of
 * |y| there.  amax1_dev / amax2_dev ([B][namax1], [B][namax2]): kind 15 without stats_in_dev reads the maxima of its raw
 * inputs x1 / x2 from them (any partition of the image into partial maxima). */
void dm_debug_conv_amax(const float* amax1_dev, int namax1, const float* amax2_dev, int namax2, float* amax_out_dev);
int dm_conv2d(int kind, const float* x1_dev, int c1, const float* x2_dev, int c2,
              const float* packed_w_dev, const float* bias_dev, int cout,
              const float* stats_in_dev, int pin, float cnt_in, const float* gamma_dev,
              const float* beta_dev, const float* temb_dev, int temb_stride, float* y_dev,
              float* stats_out_dev, int B, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DM_SAMPLER_H */
