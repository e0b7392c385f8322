// dm_kernels.hip — hand-written gfx950 (CDNA4) kernels for the reverse-diffusion sampling
// hot path.  Device code only; the C ABI and launch logic are in dm_api.hip.
//
// Activations are NHWC fp32.  All convolutions with C-channel inputs run as implicit GEMM on
// the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (64-lane waves, 32x32 fp32
// accumulator tiles in AGPRs); GroupNorm+SiLU+time-embedding is applied to the INPUT tile
// while it is staged into LDS, and GroupNorm statistics of the OUTPUT are produced by the
// epilogue as deterministic per-tile (mean, M2) partials — so no stand-alone normalisation
// pass ever touches HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// wave priorities of the staging/epilogue phases and of the K loop (s_setprio; overridable for diagnostic builds)
#ifndef DM_PRIO_STAGE
#define DM_PRIO_STAGE 2
#endif
#ifndef DM_PRIO_K
#define DM_PRIO_K 0
#endif
#define DM_GROUPS 8
#define DM_GN_EPS 1e-5f

__device__ __forceinline__ float dm_silu(float v) {
  // v * sigmoid(v); __expf is v_exp_f32 (rel. err ~1e-7 here), well inside the 1e-4 budget
  // v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of an IEEE division: the transform runs on
  // every staged element, and the full-precision divide alone cost ~13% of the conv kernel
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}
// SiLU(v) + t in 5 instructions: mul, v_exp_f32 (= exp2), add, v_rcp_f32, fma
__device__ __forceinline__ float dm_silu_add(float v, float t) {
  const float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  return fmaf(v, __builtin_amdgcn_rcpf(1.0f + e), t);
}

// four of them with the multiply and the add packed (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32); same
// operations and roundings as dm_silu_add
typedef float dm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 dm_silu_add4(float4 g, float4 t) {
  const dm_f32x2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
  const dm_f32x2 a = {g.x, g.y}, b = {g.z, g.w};
  const dm_f32x2 ea = a * nl2e, eb = b * nl2e;
  const dm_f32x2 da = dm_f32x2{__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)} + one;
  const dm_f32x2 db = dm_f32x2{__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)} + one;
  const dm_f32x2 ra = {__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
  const dm_f32x2 rb = {__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
  const dm_f32x2 oa = __builtin_elementwise_fma(a, ra, dm_f32x2{t.x, t.y});
  const dm_f32x2 ob = __builtin_elementwise_fma(b, rb, dm_f32x2{t.z, t.w});
  return make_float4(oa.x, oa.y, ob.x, ob.y);
}

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on fp32 MFMA.
//   KIND 0: Conv2d 3x3 s1 p1      KIND 1: Conv2d 4x4 s2 p1      KIND 2: ConvTranspose2d 4x4 s2 p1
//   KIND 3: Conv2d 1x1 (attention qkv / proj; optional residual add)
//   KIND 4: ConvTranspose2d 4x4 s2 p1 with all four output parities computed from ONE staged
//           tile (Cin == KC only): the input is read and transformed once instead of four times
//   KIND 5: Conv2d 3x3 s1 p1 as Winograd F(2x2,3x3): 16 transform-domain products instead of 36
//           multiplies per 2x2 outputs, input transform on the fly from the raw LDS tile, output
//           transform in registers + one LDS exchange (see wino_epilogue); H, W >= 8
//   (<2,1,16>: KIND 1 on 128-pixel tiles with 16-channel K chunks, host-side "kind 6")
// GEMM view: M = output pixels (tile of MT = 64*MB pixels per workgroup), N = 64 output
// channels per workgroup (blockIdx.y), K = taps x input channels, walked as
// [channel chunk of KC][tap][8 channels].
// Workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); each wave owns MB 32x32 accumulator
// blocks (pixels wm*32*MB + mb*32 .. , channels wn*32 ..).
//
// A operand (pixels x k): the input tile with halo lives in LDS as [halo pixel][KC + 4 pad];
// lane l of a wave reads the 16 bytes at channel 8j + 4*(l>>5) of pixel (l&31)+tap offset, and
// feeds the four floats to four MFMAs.  (MFMA k-slot h = l>>5 therefore carries channels
// 8j+4h+e in MFMA e: a permutation of k that A and B share.)
// B operand (k x couts): weights are pre-packed by pack_conv_weight_kernel into exactly that
// fragment order — [parity][tap][Cin/8][Cout/32][64 lanes][4] — so each wave fetches its
// fragment with one fully coalesced 1 KiB load from L2, one tap ahead of use, and the K loop
// has no barrier and no LDS traffic for weights.
struct ConvArgs {
  const float* x1;        // [B,H,W,C1]
  const float* x2;        // [B,H,W,C2] or null (channel concat [x1,x2], model.py:146-148)
  const float* wp;        // packed weights
  const float* bias;      // [Cout]
  float* y;               // [B,OH,OW,Cout]
  // input transform SiLU(GN(x1)) + temb; enabled when stats_in != null (requires C2 == 0)
  const float* stats_in;  // [B][pin][8][2] (mean, M2) partials
  const float* gamma;     // [C1]
  const float* beta;      // [C1]
  const float* temb;      // null or per-channel add (after SiLU)
  const int* step_ptr;    // null or device step counter (sampling loop)
  float* stats_out;       // null or [B][pout][8][2]
  const float* res;       // null or residual [B,OH,OW,Cout] added to the output (may alias y)
  int tf_silu;            // input transform applies SiLU after GroupNorm (0: GroupNorm only)
  float cnt_in;           // elements per input partial
  int pin;
  int temb_stride;        // per-image stride of temb (0 = shared)
  int temb_step_stride;   // per-step stride of temb
  int step_off;
  int B, H, W;            // input dims
  int C1, C2, Cout;
  int Hb, Wb;             // base grid of the tile decomposition (output grid; input grid for KIND 2)
  int th_log, tw_log, ni; // tile = ni images x (1<<th_log) x (1<<tw_log) base pixels
  int tiles_x, tiles_per_img;
  int hth, htw;           // halo tile dims (input pixels)
  unsigned mg_htw, mg_hth; // ceil(2^32/d) magic dividers
  int pout;               // output partials per image
  int h_log, w_log, c1_log, c2_log, cout_log;  // log2 of H, W, C1, C2, Cout (all powers of two)
  int ppp_q, ppp_r;       // (pixels staged per workgroup load) = ppp_q * htw + ppp_r
  // Everything the kernels divide by is a power of two; the logs are passed because an integer division
  // by a run-time value costs ~30 VALU instructions (there is no hardware divide), paid in MFMA time.
  int tpi_log, tx_log;    // log2 of tiles_per_img, tiles_x
  int g_log, g1_log;      // log2 of channels per GroupNorm group of the output (Cout/8) and of input 1 (C1/8)
  int ngroups;            // image groups (ceil(B / ni)); xcd_map: tiles are dealt to workgroups per XCD (see the tile loop)
  int xcd_map;
  int ntiles;             // tiles along grid.x (persistent kernels loop: t = blockIdx.x; t < ntiles; t += gridDim.x)
  int in_wrap;            // > 0: input images n >= in_wrap read the activation and statistics of image n - in_wrap (guided sampling:
                          // both halves of the doubled batch share down1.conv1's output; one image per tile only)
  // init_conv folded into this conv's staging (KIND 0, down1.conv1): when init_x != null the input
  // tile is COMPUTED from the NCHW image (3x3 conv, init_cin -> C1 channels) instead of loaded
  // two-fp16-term form on RAW inputs (conv_x6_kernel<.., F16>, stats_in == null): the producers' per-wave maxima of |x| per image,
  // [B][namax]; the tile's image is scaled by a power of two taken from them.  amax_out: this launch writes such maxima of ITS
  // output, [B][pout * (Cout / 64) * 4] (x6_epilogue_sw)
  const float* amax1; const float* amax2;
  int namax1, namax2;
  float* amax_out;
  const float* init_x;    // [B, init_cin, H, W]
  const float* init_w;    // [C1, init_cin, 3, 3]
  const float* init_b;    // [C1]
  int init_cin;
#ifdef DM_STAMPS
  unsigned long long* stamps;  // diagnostic build only: [grid][4] s_memtime stamps
#endif
};

#if defined(DM_STAMPS) && !defined(DM_TIMELINE)
#define DM_STAMP(k) do { if (p.stamps && threadIdx.x == 0) { size_t o_ = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8; p.stamps[o_ + (k)] = __builtin_amdgcn_s_memtime(); if ((k) == 0 || (k) == 3) p.stamps[o_ + 4 + ((k) == 3)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define DM_STAMP(k) do {} while (0)
#endif
#if defined(DM_STAMPS) && !defined(DM_TIMELINE)
#define DM_STAMP_RAW(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DM_STAMP_RAW(slot) do {} while (0)
#endif

// -DDM_STAMPS -DDM_TIMELINE (diagnostic build): every workgroup appends (code << 56 | s_memtime) events to its own
// 128-entry row of the stamps buffer: 1 tile start, 2 K loop start (of a chunk), 5 next chunk's staging starts, 3 K loop end, 4 tile end; entry 0 = XCC/HW id
#ifdef DM_TIMELINE
#define DM_TL(code) do { if (p.stamps && threadIdx.x == 0 && tl_n < 126) { p.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 128 + 2 + tl_n] = ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); ++tl_n; } } while (0)
#else
#define DM_TL(code) do {} while (0)
#endif

// Epilogue of conv_mfma_kernel for one set of accumulators: + bias; GroupNorm partial statistics;
// NHWC store.  otile != null: the accumulators go through LDS ([MT][68], may overlay the input
// tile — the caller has synchronised) and leave as coalesced float4 rows; otile == null (KIND 4,
// the input tile must survive): straight from the registers, 128 B per half-wave.
// scratch: >= 128 floats of LDS for the reductions.
typedef unsigned short bf16_t;  // raw bfloat16 bits
__device__ __forceinline__ bf16_t f2bf(float f);

// Stage the input tile of down1.conv1 by evaluating init_conv (model.py:90,135) on the fly:
// tile[pixel][c] = bias[c] + sum_{ci,ky,kx} x[n][ci][iy+ky-1][ix+kx-1] * w[c][ci][ky][kx] for halo
// pixels inside the image, exactly 0 outside (the 3x3 conv's own zero padding).  Same summation
// order as init_conv_kernel, so the result is bit-identical to the two-kernel form while the
// 256 KiB/image write + re-read of the init activation disappears.  aux: LDS scratch for the
// image patch (halo + 1) and the 64 output channels' weights.  Caller barriers afterwards.
template <bool BF, int RSX, bool WINO = false>
__device__ __forceinline__ void stage_init_tile(const ConvArgs& p, void* tile_v, float* aux, int chan0, int img0,
                                                int hy0, int hx0, int hth, int htw, int tid) {
  const int Ci = p.init_cin;
  const int ph = hth + 2, pw = htw + 2;
  float* xs = aux;                                   // [ni][Ci][ph][pw]
  float* wl = aux + ((p.ni * Ci * ph * pw + 3) & ~3);  // [Ci*9][64]
  float* bl = wl + Ci * 9 * 64;                      // [64]
  for (int idx = tid; idx < p.ni * Ci * ph * pw; idx += 256) {
    const int xx = idx % pw;
    int r = idx / pw;
    const int yy = r % ph;
    r /= ph;
    const int ci = r % Ci, i = r / Ci;
    const int iy = hy0 - 1 + yy, ix = hx0 - 1 + xx, n = img0 + i;
    float v = 0.f;
    if (n < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
      v = p.init_x[(((size_t)n * Ci + ci) * p.H + iy) * p.W + ix];
    xs[idx] = v;
  }
  for (int idx = tid; idx < Ci * 9 * 64; idx += 256) {
    const int c = idx & 63, k = idx >> 6;
    wl[idx] = p.init_w[(size_t)(chan0 + c) * Ci * 9 + k];
  }
  if (tid < 64) bl[tid] = p.init_b[chan0 + tid];
  __syncthreads();
  const int NP = p.ni * hth * htw;
  // idx = tid + 256 k: the channel quad (idx & 15) never changes for a thread, so with one input channel its nine
  // weight quads and the bias are read from LDS once per tile, not once per pixel
  const int c4 = tid & 15;
  float4 wreg[9];
  const bool wr = Ci == 1;
  if (wr) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wreg[k] = *reinterpret_cast<const float4*>(wl + k * 64 + c4 * 4);
  }
  const float4 breg = *reinterpret_cast<const float4*>(bl + c4 * 4);
  for (int idx = tid; idx < NP * 16; idx += 256) {
    const int pix = idx >> 4;
    const int r = __umulhi((unsigned)pix, p.mg_htw);
    const int hx = pix - r * htw;
    const int i = __umulhi((unsigned)r, p.mg_hth);
    const int hy = r - i * hth;
    const int iy = hy0 + hy, ix = hx0 + hx, n = img0 + i;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
      a = breg;
      if (wr) {  // same summation order as the general form below
        const float* xp = xs + (i * ph + hy) * pw + hx;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float xv = xp[ky * pw + kx];
            const float4 wv = wreg[ky * 3 + kx];
            a.x += xv * wv.x; a.y += xv * wv.y; a.z += xv * wv.z; a.w += xv * wv.w;
          }
      } else {
        for (int ci = 0; ci < Ci; ++ci) {
          const float* xp = xs + ((i * Ci + ci) * ph + hy) * pw + hx;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float xv = xp[ky * pw + kx];
              const float4 wv = *reinterpret_cast<const float4*>(wl + (ci * 9 + ky * 3 + kx) * 64 + c4 * 4);
              a.x += xv * wv.x; a.y += xv * wv.y; a.z += xv * wv.z; a.w += xv * wv.w;
            }
        }
      }
    }
    // Winograd kernel: the tile is stored as four (row parity, column parity) planes per image
    const int lp = WINO ? ((i * 4 + (hy & 1) * 2 + (hx & 1)) * (hth >> 1) + (hy >> 1)) * (htw >> 1) + (hx >> 1) : pix;
    if (BF) {
      uint2 o;
      o.x = (unsigned)f2bf(a.x) | ((unsigned)f2bf(a.y) << 16);
      o.y = (unsigned)f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(tile_v) + lp * RSX + c4 * 4) = o;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(tile_v) + lp * RSX + c4 * 4) = a;
    }
  }
}

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }  // RNE (v_cvt_pk_bf16_f32)
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float((unsigned)b << 16); }
// IEEE half, used by the bf16 path for tensors whose consumer normalises them (GroupNorm subtracts the group mean:
// bf16's 8-bit mantissa on the RAW conv output is amplified by |mean|/sigma; half keeps 11 bits at the same 2 bytes).
// Clamped to the finite half range so an outlier saturates instead of turning into inf -> NaN after normalisation.
__device__ __forceinline__ bf16_t f2h(float f) {
  return __builtin_bit_cast(bf16_t, (_Float16)__builtin_amdgcn_fmed3f(f, -65504.f, 65504.f));
}
__device__ __forceinline__ float h2f(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

template <int MB, int KIND, bool BF = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[MB], float* scratch, float* otile,
                                              int par, int py, int px, int img0, int tin, int y0, int x0,
                                              int wm, int wn, int lane, int h, int tid, int wave, float bv_pre = 0.f, bool bv_have = false) {
  constexpr int MT = 64 * MB;
  constexpr bool UP = (KIND == 2 || KIND == 4);
  const int th = 1 << p.th_log, tw = 1 << p.tw_log;
  const int lc = wn * 32 + (lane & 31);  // column inside this workgroup's 64 output channels
  // (fp32 kernels pass the thread's bias value, loaded once per kernel: a global load here sits in front of the
  //  first dependent add of every epilogue - four per tile in the fused transposed conv)
  const float bv = bv_have ? bv_pre : p.bias[blockIdx.y * 64 + lc];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] += bv;
  const int OW = (KIND == 1) ? (p.W >> 1) : UP ? (p.W << 1) : p.W;
  const int OH = (KIND == 1) ? (p.H >> 1) : UP ? (p.H << 1) : p.H;
  constexpr int ORS = 68;                 // output staging row stride (floats)
  float* sred = scratch;                  // [4 waves][NSW][4]
  float* s_mu = sred + 64;                // [<= 64]

  if (otile) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * (32 * MB) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        // [MT][64+8] 16-bit: half when the consumer normalises this tensor (it asked for statistics), else bf16
        if (BF) reinterpret_cast<bf16_t*>(otile)[m * 72 + lc] = p.stats_out ? f2h(acc[mb][r]) : f2bf(acc[mb][r]);
        else otile[m * ORS + lc] = acc[mb][r];
      }
  } else {
    // (opaque copy: keeps LICM from hoisting the 32 store addresses out of the caller's K loop,
    // where they would stay live in 64 registers and spill)
    int hq = h;
    asm volatile("" : "+v"(hq));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * (32 * MB) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hq;
        const int x = x0 + (m & (tw - 1));
        const int y = y0 + ((m >> p.tw_log) & (th - 1));
        const int n = img0 + (m >> (p.tw_log + p.th_log));
        if (p.ni == 1 || n < p.B) {  // (one image per tile: the grid covers exactly B images)
          const int oy = UP ? 2 * y + py : y, ox = UP ? 2 * x + px : x;
          p.y[(((size_t)n * OH + oy) * OW + ox) * p.Cout + blockIdx.y * 64 + lc] = acc[mb][r];
        }
        __builtin_amdgcn_sched_barrier(0);  // one address at a time: this runs with the K loop's registers live
      }
  }

  if (p.stats_out) {
    // Two passes over the accumulator registers (mean, then M2 about that mean), reduced with
    // cross-lane shuffles and a few LDS words in a fixed order: the result depends neither on
    // the batch size nor on dispatch order.  A 16-row segment never straddles two images.
    constexpr int NSW = MB * 2;                          // 16-row segments per wave
    const int G = 1 << p.g_log;                          // channels per group: 8, 16, 32 or 64
    const int gl_log = p.g_log < 5 ? p.g_log : 5;
    const int GL = 1 << gl_log;                          // lanes of one group inside a wave
    const int gpt_log = 6 - p.g_log;
    const int gpt = 1 << gpt_log;                        // groups in this 64-channel tile
    const int rows_log = (p.ni > 1) ? (p.th_log + p.tw_log) : (MB == 2 ? 7 : 6);
    const int rows_img = 1 << rows_log;                  // tile rows that belong to one image
    if (rows_img < 16) {
      // Images of fewer than 4x4 base pixels (the deepest levels of 8x8 / 16x16 models): an accumulator
      // segment would straddle images, so the (mean, M2) pair of every (image, group) is taken from the
      // staged output tile by one thread each.  Negligible work: these layers hold <= 4 pixels per image.
      __syncthreads();  // the staged tile is complete
      const int nig = (MT >> rows_log) << gpt_log;
      for (int idx = tid; idx < nig; idx += 256) {
        const int i = idx >> gpt_log, g = idx & (gpt - 1);
        const int n = img0 + i;
        if (n < p.B) {
          auto at = [&](int r, int c) {
            const int m = (i << rows_log) + r;
            return BF ? h2f(reinterpret_cast<const bf16_t*>(otile)[m * 72 + g * G + c]) : otile[m * ORS + g * G + c];
          };
          float sm = 0.f;
          for (int r = 0; r < rows_img; ++r)
            for (int c = 0; c < G; ++c) sm += at(r, c);
          const float mu = sm / (float)(rows_img * G);
          float m2 = 0.f;
          for (int r = 0; r < rows_img; ++r)
            for (int c = 0; c < G; ++c) { const float dd = at(r, c) - mu; m2 += dd * dd; }
          const int pidx = (UP ? par * p.tiles_per_img : 0) + tin;
          float* so = p.stats_out + (((size_t)n * p.pout + pidx) * DM_GROUPS + blockIdx.y * gpt + g) * 2;
          so[0] = mu;
          so[1] = m2;
        }
      }
    } else {
    const bool perseg = rows_img < 32 * MB;              // (then rows_img == 16: one image per segment)
    const int nimg_t = MT >> rows_log;
    const int gg = (lane & 31) >> gl_log;
    float val[NSW];
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
          float mu = 0.f;
          if (pass == 1) {
            const int i = perseg ? (wm * MB + mb) * 2 + sg : (wm * 32 * MB) >> rows_log;
            mu = s_mu[i * gpt + (lc >> p.g_log)];
          }
          float sm = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float d = acc[mb][sg * 8 + r] - mu;
            sm += (pass == 0) ? d : d * d;
          }
          val[mb * 2 + sg] = sm;
        }
      if (!perseg) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NSW; ++k) t += val[k];
        val[0] = t;
      }
#pragma unroll
      for (int k = 0; k < NSW; ++k) {
        if (k == 0 || perseg) {
          float v = val[k];
          v += __shfl_xor(v, 32, 64);
          for (int o = 1; o < GL; o <<= 1) v += __shfl_xor(v, o, 64);
          if (h == 0 && ((lane & 31) & (GL - 1)) == 0) sred[(wave * NSW + k) * 4 + gg] = v;
        }
      }
      __syncthreads();
      if (tid < nimg_t * gpt) {
        const int i = tid >> gpt_log, g = tid - i * gpt;
        float sm = 0.f;
        // waves (wm) and slots (k) that cover image i; wn / gg that cover group g
        const int wm0 = perseg ? i / NSW : (i << rows_log) / (32 * MB);
        const int nwm = perseg ? 1 : (rows_img / (32 * MB));
        const int k = perseg ? i % NSW : 0;
        const int wn0 = (g * G) >> 5, nwn = (G == 64) ? 2 : 1, g2 = ((g * G) & 31) >> gl_log;
        for (int a = 0; a < nwm; ++a)
          for (int b = 0; b < nwn; ++b) sm += sred[(((wm0 + a) * 2 + wn0 + b) * NSW + k) * 4 + g2];
        if (pass == 0) {
          s_mu[tid] = sm / (float)(rows_img * G);
        } else {
          const int n = img0 + i;
          if (p.ni == 1 || n < p.B) {  // (one image per tile: the grid covers exactly B images)
            const int pidx = (UP ? par * p.tiles_per_img : 0) + tin;
            float* so = p.stats_out + (((size_t)n * p.pout + pidx) * DM_GROUPS + blockIdx.y * gpt + g) * 2;
            so[0] = s_mu[tid];
            so[1] = sm;
          }
        }
      }
      __syncthreads();
    }
    }  // rows_img >= 16
  } else if (otile) {
    __syncthreads();
  }


  // coalesced NHWC store: one pixel's 64 channels are 256 contiguous bytes
  if (BF) {  // bf16 activations: one pixel's 64 channels are 128 contiguous bytes
    for (int idx = tid; idx < MT * 8; idx += 256) {
      const int m = idx >> 3, c8 = idx & 7;
      const int x = x0 + (m & (tw - 1));
      const int y = y0 + ((m >> p.tw_log) & (th - 1));
      const int n = img0 + (m >> (p.tw_log + p.th_log));
      if (p.ni == 1 || n < p.B) {  // (one image per tile: the grid covers exactly B images)
        const int oy = UP ? 2 * y + py : y;
        const int ox = UP ? 2 * x + px : x;
        const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(otile) + m * 72 + c8 * 8);
        bf16_t* yb = reinterpret_cast<bf16_t*>(p.y);
        *reinterpret_cast<uint4*>(yb + (((size_t)n * OH + oy) * OW + ox) * p.Cout + blockIdx.y * 64 + c8 * 8) = v;
      }
    }
    return;
  }
  if (otile)
  for (int idx = tid; idx < MT * 16; idx += 256) {
    const int m = idx >> 4, c4 = idx & 15;
    const int x = x0 + (m & (tw - 1));
    const int y = y0 + ((m >> p.tw_log) & (th - 1));
    const int n = img0 + (m >> (p.tw_log + p.th_log));
    if (p.ni == 1 || n < p.B) {  // (one image per tile: the grid covers exactly B images)
      const int oy = UP ? 2 * y + py : y;
      const int ox = UP ? 2 * x + px : x;
      float4 v = *reinterpret_cast<const float4*>(otile + m * ORS + c4 * 4);
      const size_t o = (((size_t)n * OH + oy) * OW + ox) * p.Cout + blockIdx.y * 64 + c4 * 4;
      if (p.res) {
        const float4 r = *reinterpret_cast<const float4*>(p.res + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(p.y + o) = v;
    }
  }
}

// GroupNorm tables of one tile: scale = rstd*gamma, shift = beta - mean*scale, temb, per (image, channel),
// from the producer's (mean, M2) partials merged in a fixed order (Chan et al.).
__device__ __forceinline__ void gn_tables(const ConvArgs& p, int step, int img0, int tq, float* s_scale, float* s_shift, float* s_temb) {
  const float* te = p.temb ? p.temb + (size_t)step * p.temb_step_stride : nullptr;
  for (int idx = tq; idx < p.ni * p.C1; idx += 256) {
    const int i = idx >> p.c1_log, c = idx & (p.C1 - 1);
    const int n = img0 + i;
    float scale = 0.f, shift = 0.f, tv = 0.f;
    if (p.ni == 1 || n < p.B) {  // (one image per tile: the grid covers exactly B images)
      const int nsrc = (p.in_wrap && n >= p.in_wrap) ? n - p.in_wrap : n;  // (the time embedding below is the image's own)
      const float* st = p.stats_in + ((size_t)nsrc * p.pin * DM_GROUPS + (c >> p.g1_log)) * 2;
      float msum = 0.f, mean, m2 = 0.f, dev = 0.f;
      if (p.pin <= 8) {  // the usual case: all partials in ONE round of independent 8-byte loads, same arithmetic
        float2 pr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pr[q] = *reinterpret_cast<const float2*>(st + (q < p.pin ? q : 0) * DM_GROUPS * 2);
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < p.pin) msum += pr[q].x;
        mean = msum / (float)p.pin;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < p.pin) {
          const float d = pr[q].x - mean;
          m2 += pr[q].y;
          dev += d * d;
        }
      } else {
        for (int q = 0; q < p.pin; ++q) msum += st[q * DM_GROUPS * 2];
        mean = msum / (float)p.pin;
        for (int q = 0; q < p.pin; ++q) {
          const float d = st[q * DM_GROUPS * 2] - mean;
          m2 += st[q * DM_GROUPS * 2 + 1];
          dev += d * d;
        }
      }
      const float var = (m2 + p.cnt_in * dev) / (p.cnt_in * (float)p.pin);
      scale = rsqrtf(var + DM_GN_EPS) * p.gamma[c];
      shift = p.beta[c] - mean * scale;
      if (te) tv = te[(size_t)n * p.temb_stride + c];
    }
    s_scale[idx] = scale;
    s_shift[idx] = shift;
    s_temb[idx] = tv;
  }
}

// Epilogue of the Winograd kernel (KIND 5).  Wave i holds M[i][j] (j = 0..3) of 32 tiles x 64 channels:
// the row half of the output transform (T[i][b] = sum_j M[i][j] A^T[b][j]) happens in registers, the
// column half (Y[a][b] = sum_i A^T[a][i] T[i][b]) needs all four waves and goes through LDS.  Then, with
// each thread holding 8 pixels x 4 channels of the finished tile: + bias, GroupNorm (mean, M2) partials
// in a fixed summation order (independent of batch size and dispatch order), coalesced NHWC store.
// LDS: T [4][2][32][68] floats, then 256 + 64 reduction words.
#define DM_WINO_EPI_FLOATS (4 * 2 * 32 * 68 + 256 + 64)
__device__ __forceinline__ void wino_epilogue(const ConvArgs& p, f32x16 (&acc)[8], float* smem, int img0, int tin,
                                              int y0, int x0, int tq) {
  constexpr int TRS = 68;
  constexpr int PS = 2 * 32 * TRS;  // plane stride between waves
  const int lane = tq & 63, wave = tq >> 6, h = lane >> 5, n32 = lane & 31;
  float* Tb = smem;
  float* sred = smem + 4 * PS;  // [4 waves][8 image slots][8 groups]
  float* s_mu = sred + 256;     // [8 image slots][8 groups]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0 + nb][r], m1 = acc[2 + nb][r], m2 = acc[4 + nb][r], m3 = acc[6 + nb][r];
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      float* q = Tb + wave * PS + row * TRS + nb * 32 + n32;
      q[0] = (m0 + m1) + m2;
      q[32 * TRS] = (m1 - m2) - m3;
    }
  __syncthreads();
  const int th = 1 << p.th_log, tw = 1 << p.tw_log;
  const int tsh = p.tw_log + p.th_log;  // log2(pixels of one image inside the tile)
  const int c4 = tq & 15, p0 = tq >> 4;
  const float4 bv = *reinterpret_cast<const float4*>(p.bias + blockIdx.y * 64 + c4 * 4);
  float4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int m = p0 + 16 * k;
    const int x = m & (tw - 1), y = (m >> p.tw_log) & (th - 1), i = m >> tsh;
    const int tl = (i << (tsh - 2)) | ((y >> 1) << (p.tw_log - 1)) | (x >> 1);
    const int a = y & 1, b = x & 1;
    const float* q = Tb + ((a * 2 + b) * 32 + tl) * TRS + c4 * 4;  // plane a: a = 0 sums T0+T1+T2, a = 1 T1-T2-T3
    const float4 t0 = *reinterpret_cast<const float4*>(q);
    const float4 t1 = *reinterpret_cast<const float4*>(q + PS);
    const float4 t2 = *reinterpret_cast<const float4*>(q + 2 * PS);
    const float sg = a ? -1.f : 1.f;
    v[k].x = fmaf(sg, t2.x, fmaf(sg, t1.x, t0.x)) + bv.x;
    v[k].y = fmaf(sg, t2.y, fmaf(sg, t1.y, t0.y)) + bv.y;
    v[k].z = fmaf(sg, t2.z, fmaf(sg, t1.z, t0.z)) + bv.z;
    v[k].w = fmaf(sg, t2.w, fmaf(sg, t1.w, t0.w)) + bv.w;
  }
  if (p.stats_out) {
    const int G = 1 << p.g_log;             // channels per group: 8, 16, 32 or 64
    const int qg = G >> 2;                  // channel quads per group
    const int gpt_log = 6 - p.g_log;
    const int gpt = 1 << gpt_log;           // groups in this workgroup's 64 channels
    const int gq = c4 >> (p.g_log - 2);
    const int kk_log = tsh - 4;             // log2(k slots per image): rows of one image / 16
    const float inv_cnt = 1.0f / (float)((1 << tsh) * G);
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = 0; i < p.ni; ++i) {
        const float mu = pass ? s_mu[i * 8 + gq] : 0.f;
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dx = v[k].x - mu, dy = v[k].y - mu, dz = v[k].z - mu, dw = v[k].w - mu;
          const float part = pass ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : (dx + dy) + (dz + dw);
          sm += ((k >> kk_log) == i) ? part : 0.f;
        }
        for (int o = 1; o < qg && o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        if ((lane >> 4) == 0 && (c4 & (qg - 1)) == 0) sred[(wave * 8 + i) * 8 + gq] = sm;
      }
      __syncthreads();
      if (tq < p.ni * gpt) {
        const int i = tq >> gpt_log, g = tq - i * gpt;
        const float sm = ((sred[(0 * 8 + i) * 8 + g] + sred[(1 * 8 + i) * 8 + g]) + sred[(2 * 8 + i) * 8 + g]) + sred[(3 * 8 + i) * 8 + g];
        if (pass == 0) {
          s_mu[i * 8 + g] = sm * inv_cnt;
        } else {
          const int n = img0 + i;
          if (p.ni == 1 || n < p.B) {  // (one image per tile: the grid covers exactly B images)
            float* so = p.stats_out + (((size_t)n * p.pout + tin) * DM_GROUPS + blockIdx.y * gpt + g) * 2;
            so[0] = s_mu[i * 8 + g];
            so[1] = sm;
          }
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int m = p0 + 16 * k;
    const int x = x0 + (m & (tw - 1)), y = y0 + ((m >> p.tw_log) & (th - 1)), n = img0 + (m >> tsh);
#ifdef DM_DIAG_NOSTORE
    if (v[k].x == 1.2345f)
#else
    if (p.ni == 1 || n < p.B)
#endif
      *reinterpret_cast<float4*>(p.y + ((((size_t)n * p.H + y) * p.W + x) * p.Cout + blockIdx.y * 64 + c4 * 4)) = v[k];
  }
}

// GEO = 1 (KIND 5 only): the geometry of the model's large 3x3 layers is a compile-time fact — tile 8x16 pixels of
// one image, 64 input channels per pointer, 64 output channels — so the index math folds to constants.
template <int MB, int KIND, int KC, int GEO = 0>
__global__ __launch_bounds__(256, KIND == 5 ? 2 : 3) void conv_mfma_kernel(const ConvArgs p_in) {
  ConvArgs p = p_in;
  if constexpr (GEO == 2) {  // KIND 5 on 8x8 images: two whole images per tile (halo 10x10 each), 64 -> 64 channels
    static_assert(MB == 2 && KIND == 5, "GEO=2 instantiation");
    p.th_log = 3; p.tw_log = 3; p.ni = 2; p.hth = 10; p.htw = 10;
    p.mg_htw = (unsigned)((0x100000000ull + 9) / 10); p.mg_hth = p.mg_htw;
    p.H = 8; p.W = 8; p.h_log = 3; p.w_log = 3; p.tiles_per_img = 1; p.tiles_x = 1; p.tpi_log = 0; p.tx_log = 0;
    p.C1 = 64; p.c1_log = 6; p.c2_log = 6; p.Cout = 64; p.cout_log = 6; p.g_log = 3; p.g1_log = 3;
  }
  if constexpr (GEO == 3) {  // KIND 5, the widened model (BASELINE configs[4]): GEO = 1's tile, 256 channels per pointer and 256 outputs
    static_assert(MB == 2 && KIND == 5, "GEO=3 instantiation");
    p.th_log = 3; p.tw_log = 4; p.ni = 1; p.hth = 10; p.htw = 18;
    p.mg_htw = (unsigned)((0x100000000ull + 17) / 18); p.mg_hth = (unsigned)((0x100000000ull + 9) / 10);
    p.ppp_q = 0; p.ppp_r = 16;
    p.C1 = 256; p.c1_log = 8; p.c2_log = 8; p.Cout = 256; p.cout_log = 8; p.g_log = 5; p.g1_log = 5;
  }
  if constexpr (GEO == 1) {
    // 128-pixel tiles of one image: 8x16 (3x3 and transposed convs: halo 10x18, the fused transposed conv
    // included) or 16x8 outputs of the stride-2 conv (halo 34x18); 64 channels per input pointer, 64 outputs
    static_assert(MB == 2 && (KIND == 5 || KIND == 4 || KIND == 1), "GEO=1 instantiations");
    constexpr int HTH = (KIND == 1) ? 34 : 10, HTW = 18;
    p.th_log = (KIND == 1) ? 4 : 3; p.tw_log = (KIND == 1) ? 3 : 4; p.ni = 1; p.hth = HTH; p.htw = HTW;
    p.mg_htw = (unsigned)((0x100000000ull + HTW - 1) / HTW); p.mg_hth = (unsigned)((0x100000000ull + HTH - 1) / HTH);
    constexpr int PPPX = (KIND == 5) ? 16 : 256 / (KC / 4);
    p.ppp_q = PPPX / HTW; p.ppp_r = PPPX % HTW;
    p.C1 = 64; p.c1_log = 6; p.c2_log = 6; p.Cout = 64; p.cout_log = 6; p.g_log = 3; p.g1_log = 3;
  }
  constexpr int MT = 64 * MB;
  constexpr bool WINO = (KIND == 5);
  constexpr int RS = KC + 4;  // LDS row stride (floats): 16-B skew between consecutive pixels
  constexpr int S = (KIND == 1) ? 2 : 1;
  constexpr int NTY = (KIND == 0) ? 3 : (KIND == 1) ? 4 : (KIND == 2 || KIND == 4) ? 2 : 1;
  constexpr int NTOT = (KIND == 4) ? 16 : NTY * NTY;  // taps walked by the K loop (4 parities x 4)
  constexpr int NT = NTY * NTY;
  constexpr int NJ = KC / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the sampling loop's device-side step counter, read ONCE (it selects the time-embedding row): per tile it would
  // put a second dependent global load in front of the table computation
  const int step_now = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;
  // this thread's output-channel bias in the direct kernels' accumulator layout (channel wn*32 + lane&31)
  const float bias_lane = WINO ? 0.f : p.bias[blockIdx.y * 64 + ((threadIdx.x >> 6) & 1) * 32 + (threadIdx.x & 31)];

  DM_STAMP(0);
#ifdef DM_TIMELINE
  int tl_n = 0;
  if (p.stamps && threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 128] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
  // Staging and epilogue ask for raised wave priority, the K loop for 0.  (Measured effect: small.
  // The fp32 MFMA executes on the SIMD's fp32 lanes, so other waves' VALU work does not overlap
  // it whatever the priority: tools/diag/valu_starve.hip.)
  __builtin_amdgcn_s_setprio(DM_PRIO_STAGE);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5;
  const int Cin = p.C1 + p.C2;
  int par = (KIND == 2) ? blockIdx.z : 0;  // KIND 4: advances inside the K loop
  int py = par >> 1, px = par & 1;
  const int th = 1 << p.th_log, tw = 1 << p.tw_log;

  const int hth = p.hth, htw = p.htw;
  const int NP = p.ni * hth * htw;

  // ---- LDS carve: [scale ni*C1][shift ni*C1][temb ni*C1][tile NP*RS]
  const bool tf = p.stats_in != nullptr;
  float* smem_b = smem + (WINO ? 208 : 0);  // (KIND 5: the descriptor table sits in front)
  float* s_scale = smem_b;
  float* s_shift = s_scale + (tf ? p.ni * p.C1 : 0);
  float* s_temb = s_shift + (tf ? p.ni * p.C1 : 0);
  float* tile = s_temb + (tf ? p.ni * p.C1 : 0);

  // ---- per-lane A-fragment base offsets (floats) into the halo tile at tap (0,0)
  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB && !WINO; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    const int x = m & (tw - 1);
    const int y = (m >> p.tw_log) & (th - 1);
    const int i = m >> (p.tw_log + p.th_log);
    abase[mb] = ((i * hth + y * S) * htw + x * S) * RS + 4 * h;
  }

  // KIND 5 (Winograd): wave i owns the transform-domain points (i, 0..3) of the 32 2x2-pixel
  // tiles, for both 32-channel halves: acc[j*2 + nb].  aoff: the lane's eight LDS reads per channel
  // group — rows rA and rB of its tile's 4x4 input patch (B^T row i has two non-zeros), 4 columns.
  int albase = 0;         // lane part of the A-read offsets (floats)
  int auo[WINO ? 8 : 1];  // wave-uniform part (scalar registers)
  float wsg = 0.f;
  if constexpr (WINO) {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int rA = (wv == 0) ? 0 : (wv == 2) ? 2 : 1;
    const int rB = (wv == 0) ? 2 : (wv == 1) ? 2 : (wv == 2) ? 1 : 3;
    wsg = (wv == 1) ? 1.f : -1.f;
    const int PH = p.hth >> 1, PW = p.htw >> 1;
    const int tl = lane & 31;
    const int txw = tl & ((tw >> 1) - 1);
    const int tyw = (tl >> (p.tw_log - 1)) & ((th >> 1) - 1);
    const int iw = tl >> (p.tw_log + p.th_log - 2);
    albase = ((iw * 4 * PH + tyw) * PW + txw) * RS + 4 * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = (q < 4) ? rA : rB, sc = q & 3;
      auo[q] = __builtin_amdgcn_readfirstlane(((((r & 1) * 2 + (sc & 1)) * PH + (r >> 1)) * PW + (sc >> 1)) * RS);
    }
  }
  f32x16 acc[WINO ? 8 : MB];  // zeroed after the first chunk is staged (keeps staging's register peak low)

  const int nb32 = p.Cout >> 5;
  const size_t jstride = (size_t)nb32 * 256;
  const size_t tapstride = (size_t)(Cin >> 3) * jstride;
  const int nchunks = Cin / KC;
  constexpr int V = KC / 4;
  constexpr int U = 12;  // global loads in flight per thread while staging (one batch covers a tile)
  // KIND 5: staging descriptors, computed ONCE per workgroup (not per tile) into a 208-entry LDS table:
  // halo pixel q of the tile -> LDS float offset | hx << 16 | hy << 22 | image << 28, bit 31 = past the
  // tile.  Address math beside a streaming fp32-MFMA wave costs about an MFMA slot per instruction, so
  // what does not depend on the tile is looked up (LDS reads are not VALU) and the per-tile part is ~10
  // instructions per load.  (Kept in LDS, not registers: 13 more live registers spill beside 128 accumulators.)
  constexpr int UW = 13;
  int* ptab = reinterpret_cast<int*>(smem);
  if constexpr (WINO) {
    if (tid < 16 * UW) {
      const int pix = tid;
      const int PH = hth >> 1, PW = htw >> 1;
      const int r = __umulhi((unsigned)pix, p.mg_htw);
      const int hx = pix - r * htw;
      const int ii = __umulhi((unsigned)r, p.mg_hth);
      const int hy = r - ii * hth;
      const int lofs = (((ii * 4 + (hy & 1) * 2 + (hx & 1)) * PH + (hy >> 1)) * PW + (hx >> 1)) * RS;
      // entries past the tile point at a dump row behind it: they are written like the rest (as zeros), so the
      // staging loop has no divergent branch and the scheduler can interleave the elements' dependency chains
      ptab[tid] = (pix < NP) ? (lofs | (hx << 16) | (hy << 22) | (ii << 28)) : (int)(0x80000000u | (unsigned)(NP * RS));
    }
    __syncthreads();
  }

  // Persistent workgroups: the grid is at most 3 per CU and each workgroup walks tiles
  // t, t + gridDim.x, ...  (a finished workgroup's slot otherwise stays empty for 2-16 k cycles
  // until the dispatcher refills it: ~4 % of a CU's time at 160 k cycles per tile)
  // XCD-aware order (xcd_map; needs gridDim.x % 8 == 0): hardware deals workgroups to the 8 XCDs round-robin, so
  // workgroup b runs on XCD b & 7.  Image group g is given to XCD g & 7 and that XCD's workgroups (b >> 3) walk ITS
  // tiles in order: the tiles of one image - which share halo rows - are staged by neighbouring workgroups of one
  // XCD at about the same time and meet in that XCD's L2 instead of being fetched once per XCD.
  const int t_first = p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = p.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_xcd = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int t_end = p.xcd_map ? (((p.ngroups - t_xcd + 7) >> 3) << p.tpi_log) : p.ntiles;
  for (int ts = t_first; ts < t_end; ts += t_step) {
  const int t = p.xcd_map ? (((((ts >> p.tpi_log) << 3) + t_xcd) << p.tpi_log) | (ts & (p.tiles_per_img - 1))) : ts;
  // opaque copy of the thread id: index math derived from it is recomputed per tile instead of
  // being hoisted out of this loop and held in (spilled) registers across the K loop
  int tq = threadIdx.x;
  asm volatile("" : "+v"(tq));
  if (KIND == 4) { par = 0; py = 0; px = 0; }  // (advanced by the previous tile's K loop)
  DM_STAMP_RAW(7);  // (diagnostic build) start of this tile: slots 7,1,2,3 bracket the LAST tile's phases
  DM_TL(1);
  const int img0 = (t >> p.tpi_log) * p.ni;
  const int tin = t & (p.tiles_per_img - 1);
  const int y0 = (tin >> p.tx_log) << p.th_log;
  const int x0 = (tin & (p.tiles_x - 1)) << p.tw_log;
  const int hy0 = (KIND == 0 || KIND == 4 || KIND == 5) ? y0 - 1 : (KIND == 1) ? 2 * y0 - 1 : (KIND == 2) ? y0 - 1 + py : y0;
  const int hx0 = (KIND == 0 || KIND == 4 || KIND == 5) ? x0 - 1 : (KIND == 1) ? 2 * x0 - 1 : (KIND == 2) ? x0 - 1 + px : x0;

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cg0 = chunk * KC;
    const bool from2 = cg0 >= p.C1;
    const float* src = from2 ? p.x2 : p.x1;
    const int Cs = from2 ? p.C2 : p.C1;
    const int coff = from2 ? cg0 - p.C1 : cg0;
    // wave-uniform (scalar) part of the weight address + a per-lane 16-byte offset: lets the
    // compiler use the saddr form of global_load instead of 64-bit vector adds per load
    const int wn_s = __builtin_amdgcn_readfirstlane(wn);
    const float* wuni = p.wp + ((size_t)par * NT) * tapstride + (size_t)chunk * NJ * jstride +
                        (size_t)(blockIdx.y * 2 + wn_s) * 256;
    const float* wbase = wuni + lane * 4;
    if (chunk > 0) {
      DM_TL(5);
      __builtin_amdgcn_s_setprio(DM_PRIO_STAGE);
      __syncthreads();
    }  // previous chunk's LDS reads are done
    // ---- stage: global -> (transform) -> LDS, U loads in flight per thread (HBM latency
    // is ~2 us under load; one load at a time would make this phase latency-bound)
    if ((KIND == 0 || KIND == 5) && p.init_x) {
      stage_init_tile<false, RS, WINO>(p, tile, tile + NP * RS, cg0, img0, hy0, hx0, hth, htw, tq);
    } else if constexpr (WINO) {
      const int c4 = tq & 15;
      const int cs_log = from2 ? p.c2_log : p.c1_log;
      const int simg = (p.in_wrap && img0 >= p.in_wrap) ? img0 - p.in_wrap : img0;
      const float* srcb = src + ((size_t)simg << (p.h_log + p.w_log + cs_log)) + coff + c4 * 4;
      const int imax = p.B - 1 - img0;  // images past the batch: clamped address, zero fill
      const int hw1 = (p.H << p.w_log) - 1;
      float4 v[UW];
      int pk[UW];
      const int p0 = tq >> 4;
#pragma unroll
      for (int u = 0; u < UW; ++u) pk[u] = ptab[p0 + 16 * u] + c4 * 4;
      unsigned inbm = 0;  // bit u: pixel u is inside the image (else zero padding / past the batch)
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int k = pk[u];
        const int hxu = (k >> 16) & 63, hyu = (k >> 22) & 63;
        const int iu = (p.ni == 1) ? 0 : (k >> 28) & 7;  // (one image per tile: the grid covers exactly B images)
        const int iy = hy0 + hyu, ix = hx0 + hxu;
        const bool inb = k >= 0 && (p.ni == 1 || iu <= imax) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        // unconditional load from a clamped (always valid) address; padding is a select afterwards
        const int lin = min(max((iy << p.w_log) + ix, 0), hw1);
        const unsigned off = (unsigned)((((p.ni == 1) ? 0 : min(iu, imax)) << (p.h_log + p.w_log)) + lin) << cs_log;
#ifdef DM_DIAG_NOLOAD
        v[u] = make_float4((float)off, 1.f, 2.f, 3.f);
#else
        v[u] = *reinterpret_cast<const float4*>(srcb + off);
#endif
        inbm |= inb ? (1u << u) : 0u;
      }
      // (opaque: otherwise hipcc keeps the 26 halo coordinates alive - spilled - and redoes the four
      // comparisons per element below instead of testing one bit)
      asm volatile("" : "+v"(inbm));
      if (tf && chunk == 0) {
        // (a fresh opaque copy of the thread id: the table addresses are formed HERE, after the loads were issued;
        //  formed earlier they sit in registers across the 13 loads and are spilled - 6 dwords per thread and tile)
        int tg = threadIdx.x;
        asm volatile("" : "+v"(tg));
        gn_tables(p, step_now, img0, tg, s_scale, s_shift, s_temb);  // while the tile loads are in flight
        __syncthreads();
      }
      // three straight-line variants (the uniform conditions are hoisted out of the element loop)
      auto put = [&](int u, float4 w) {
        if (!((inbm >> u) & 1)) w = make_float4(0.f, 0.f, 0.f, 0.f);  // padding (and the dump row) stays exactly 0
        *reinterpret_cast<float4*>(tile + (pk[u] & 0xffff)) = w;
      };
      auto gn = [&](int u, float4& te) {
        // table entries come from LDS every time (three 16-byte reads, no VALU): holding them in
        // 12 registers across the loop spills beside the 128 accumulators
        const int tofs = ((p.ni == 1) ? 0 : (((pk[u] >> 28) & 7) << p.c1_log)) + coff + c4 * 4;
        const float4 sc = *reinterpret_cast<const float4*>(s_scale + tofs);
        const float4 sh = *reinterpret_cast<const float4*>(s_shift + tofs);
        te = *reinterpret_cast<const float4*>(s_temb + tofs);
        float4 w = v[u];
        w.x = fmaf(w.x, sc.x, sh.x); w.y = fmaf(w.y, sc.y, sh.y);
        w.z = fmaf(w.z, sc.z, sh.z); w.w = fmaf(w.w, sc.w, sh.w);
        return w;
      };
      if (!tf) {
#pragma unroll
        for (int u = 0; u < UW; ++u) put(u, v[u]);
      } else if (p.tf_silu) {
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          float4 te;
          const float4 w = gn(u, te);
          put(u, dm_silu_add4(w, te));
        }
      } else {
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          float4 te;
          float4 w = gn(u, te);
          w.x += te.x; w.y += te.y; w.z += te.z; w.w += te.w;
          put(u, w);
        }
      }
    } else
    {
      // Every VALU instruction here is paid for in MFMA time: the fp32 MFMA runs on the SIMD's
      // fp32 lanes, so a co-resident wave's vector work does not overlap it (measured: a VALU wave
      // beside two back-to-back MFMA waves makes no progress until they finish).  Hence: the
      // thread's channel quad is fixed, the halo pixel advances by 256/V per load with
      // add/compare/select instead of divisions, offsets are 32-bit shifts, tables are hoisted.
      constexpr int PPP = 256 / V;                 // pixels covered by one load of the workgroup
      const int c4 = tq & (V - 1), p0 = tq / V;
      const int r0 = __umulhi((unsigned)p0, p.mg_htw);
      int hx = p0 - r0 * htw;
      int ii = __umulhi((unsigned)r0, p.mg_hth);
      int hy = r0 - ii * hth;
      const int cs_log = from2 ? p.c2_log : p.c1_log;
      const float* srcb = src + ((size_t)img0 << (p.h_log + p.w_log + cs_log)) + coff + c4 * 4;
      const int imax = p.B - 1 - img0;             // clamp for images past the batch
      int pix = p0, lofs = p0 * RS + c4 * 4;
      for (int base = 0; base < NP; base += PPP * U) {  // uniform trip count (barrier inside)
        float4 v[U];
        int meta[U];  // (LDS float offset) | (table offset + 1) << 16; table part 0 = zero fill
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (GEO == 1) {  // compile-time halo width, one image: direct (hy, hx), no carried state
            hy = pix / htw; hx = pix - hy * htw; ii = 0;
          } else {  // run-time geometry: two multiply-high divisions, still branch-free and without carried state
            const int r = __umulhi((unsigned)pix, p.mg_htw);
            hx = pix - r * htw;
            ii = __umulhi((unsigned)r, p.mg_hth);
            hy = r - ii * hth;
          }
          const int iy = hy0 + hy, ix = hx0 + hx;
          const bool valid = pix < NP;
          // The load is UNCONDITIONAL from a clamped (always valid) address and the padding is a
          // select afterwards: a load under a runtime condition makes hipcc branch around it and
          // wait vmcnt(0) each time, which serialises the loads (~2 us apiece).
          const bool inb = valid && ii <= imax && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const int cn = min(ii, imax), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
          const unsigned off = (unsigned)((((cn << p.h_log) + cy) << p.w_log) + cx) << cs_log;
          v[u] = *reinterpret_cast<const float4*>(srcb + off);
          // zero padding / images past the batch stay exactly 0 and get no transform
          if (WINO)  // four (row parity, column parity) planes per image
            lofs = (((ii * 4 + (hy & 1) * 2 + (hx & 1)) * (hth >> 1) + (hy >> 1)) * (htw >> 1) + (hx >> 1)) * RS + c4 * 4;
          meta[u] = valid ? (lofs | (inb ? (((ii << p.c1_log) + coff + c4 * 4 + 1) << 16) : 0)) : -1;
          pix += PPP;
          if (!WINO) lofs += PPP * RS;

        }
        if (tf && chunk == 0 && base == 0) {
          // GroupNorm tables (scale, shift = beta - mean*scale, temb per (image, channel)),
          // computed while the first batch of tile loads is in flight
          gn_tables(p, step_now, img0, tq, s_scale, s_shift, s_temb);
          __syncthreads();  // uniform condition
        }
        float4 hsc, hsh;  // with one image per tile the thread's table entries never change
        if (tf && p.ni == 1) {
          hsc = *reinterpret_cast<const float4*>(s_scale + coff + c4 * 4);
          hsh = *reinterpret_cast<const float4*>(s_shift + coff + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (meta[u] >= 0) {
            float4 w = v[u];
            const int lo = meta[u] & 0xffff, tofs = (meta[u] >> 16) - 1;
            if (tofs < 0) w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tf && tofs >= 0) {
              float4 sc = hsc, sh = hsh;
              const float4 te = *reinterpret_cast<const float4*>(s_temb + tofs);
              if (p.ni != 1) {
                sc = *reinterpret_cast<const float4*>(s_scale + tofs);
                sh = *reinterpret_cast<const float4*>(s_shift + tofs);
              }
              w.x = fmaf(w.x, sc.x, sh.x); w.y = fmaf(w.y, sc.y, sh.y);
              w.z = fmaf(w.z, sc.z, sh.z); w.w = fmaf(w.w, sc.w, sh.w);
              if (p.tf_silu) {
                w.x = dm_silu_add(w.x, te.x); w.y = dm_silu_add(w.y, te.y);
                w.z = dm_silu_add(w.z, te.z); w.w = dm_silu_add(w.w, te.w);
              } else {
                w.x += te.x; w.y += te.y; w.z += te.z; w.w += te.w;
              }
            }
            *reinterpret_cast<float4*>(tile + lo) = w;
          }
        }
      }
    }
    if constexpr (WINO) {
      // ---- Winograd F(2x2,3x3): sixteen [32 tiles x KC] x [KC x 64] products, one per point of the
      // 4x4 transform domain; this wave owns points (i, 0..3).  Packed weights (U = G g G^T):
      // [Cin/8][point][Cout/32][64 lanes][4].  Per channel group of 8: eight 16-byte LDS reads
      // (rows rA, rB x 4 columns of the tile's input patch), 32 VALU (B^T d B on the fly), 32 MFMAs.
      const int wv = __builtin_amdgcn_readfirstlane(tq >> 6);
      const size_t gstride = (size_t)16 * nb32 * 256;
      // wave-uniform fragment pointers (scalar registers) + one 32-bit lane offset: saddr-form loads
      const float* wu5 = p.wp + ((size_t)chunk * NJ) * gstride + ((size_t)(wv * 4) * nb32 + blockIdx.y * 2) * 256;
      const unsigned lane16 = (unsigned)(tq & 63) * 16u;  // byte offset: (uniform base) + zext(32-bit) selects the saddr form
      float4 bq[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bq[j * 2 + nb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wu5 + (size_t)(j * nb32 + nb) * 256) + lane16);
      __syncthreads();
      __builtin_amdgcn_s_setprio(DM_PRIO_K);
      if (chunk == 0) {
        DM_STAMP(1);
        DM_TL(2);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
      }
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 sg2 = {wsg, wsg};
      f32x2 dl[8], dh[8];  // .xy / .zw halves of the eight 16-byte reads
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 t4 = *reinterpret_cast<const float4*>(tile + albase + auo[q]);
        dl[q] = f32x2{t4.x, t4.y}; dh[q] = f32x2{t4.z, t4.w};
      }
#pragma unroll
      for (int g = 0; g < NJ; ++g) {  // fully unrolled: LDS/global offsets become immediates, no register rotation
        f32x2 el[4], eh[4];  // e[c] = d[rA][c] +- d[rB][c]  (B^T d, this wave's row)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          el[c] = __builtin_elementwise_fma(sg2, dl[4 + c], dl[c]);
          eh[c] = __builtin_elementwise_fma(sg2, dh[4 + c], dh[c]);
        }
        const int gn = (g + 1 < NJ) ? g + 1 : g;  // (clamped: the last prefetch re-reads valid data, unused)
#ifndef DM_DIAG_NOLDS
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(tile + albase + auo[q] + gn * 8);
          dl[q] = f32x2{t4.x, t4.y}; dh[q] = f32x2{t4.z, t4.w};
        }
#endif
        const float* wnext = wu5 + (size_t)gn * gstride;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // (B^T d B)[i][j]: columns combine as e0-e2, e1+e2, e2-e1, e1-e3
          const f32x2 vl = (j == 0) ? el[0] - el[2] : (j == 1) ? el[1] + el[2] : (j == 2) ? el[2] - el[1] : el[1] - el[3];
          const f32x2 vh = (j == 0) ? eh[0] - eh[2] : (j == 1) ? eh[1] + eh[2] : (j == 2) ? eh[2] - eh[1] : eh[1] - eh[3];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const float4 b = bq[j * 2 + nb];
            acc[j * 2 + nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl.x, b.x, acc[j * 2 + nb], 0, 0, 0);
            acc[j * 2 + nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vl.y, b.y, acc[j * 2 + nb], 0, 0, 0);
            acc[j * 2 + nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh.x, b.z, acc[j * 2 + nb], 0, 0, 0);
            acc[j * 2 + nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vh.y, b.w, acc[j * 2 + nb], 0, 0, 0);
#ifndef DM_DIAG_NOB
            bq[j * 2 + nb] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(wnext + (size_t)(j * nb32 + nb) * 256) + lane16);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
    float4 bq[NJ];  // B fragments of tap 0: their L2 latency hides behind the barrier
#pragma unroll
    for (int j = 0; j < NJ; ++j) bq[j] = *reinterpret_cast<const float4*>(wbase + j * jstride);
    __syncthreads();
    __builtin_amdgcn_s_setprio(DM_PRIO_K);
    if (chunk > 0) DM_TL(2);
    if (chunk == 0) {
      DM_STAMP(1);
      DM_TL(2);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    }

    // ---- K loop over taps x KC channels, software-pipelined by hand:
    //   group j of a tap = { LDS read of the NEXT group's A fragments ; 4*MB MFMAs on the
    //   current ones ; global load of the next tap's B fragment j (into the slot just used) }.
    // sched_barrier(0) pins that order: left alone, hipcc sinks the B loads to the end of the
    // tap and waits vmcnt(0) on them, and issues each A read right before its first use.
    float4 a0[MB], a1[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a0[mb] = *reinterpret_cast<const float4*>(tile + abase[mb]);
    int ty = 0, tx = 0;
    // compile-time geometry: the taps are unrolled, so every LDS offset is an immediate and no index math runs
    // between the MFMAs (as in the Winograd K loop); run-time geometry keeps the rolled loop
    constexpr int TAP_UNROLL = (GEO == 1 && KIND == 1) ? NTOT : 1;  // (KIND 4: any unrolling spills - 4 taps: 202 registers)
#pragma unroll TAP_UNROLL
    for (int tap = 0; tap < NTOT; ++tap) {
      // KIND 4: tap = parity*4 + (ty*2+tx); parity (py,px) reads halo rows py+ty, columns px+tx
      const int tapoff = (KIND == 4) ? (((tap >> 3) + ((tap >> 1) & 1)) * htw + ((tap >> 2) & 1) + (tap & 1)) * RS
                                     : (ty * htw + tx) * RS;
      // next tap (clamped at the end: the extra prefetch re-reads valid data and is unused)
      int ntx = tx + 1, nty = ty;
      if (ntx == NTY) { ntx = 0; ++nty; }
      const bool last = tap + 1 == NTOT;
      const int nt = tap + 1;
      const int ntapoff = last ? tapoff
                        : (KIND == 4) ? (((nt >> 3) + ((nt >> 1) & 1)) * htw + ((nt >> 2) & 1) + (nt & 1)) * RS
                                      : (nty * htw + ntx) * RS;
      const float* wnext = wbase + (size_t)(last ? tap : tap + 1) * tapstride;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float4* acur = (j & 1) ? a1 : a0;
        float4* anxt = (j & 1) ? a0 : a1;
        const int noff = (j + 1 < NJ) ? tapoff + (j + 1) * 8 : ntapoff;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) anxt[mb] = *reinterpret_cast<const float4*>(tile + abase[mb] + noff);
        __builtin_amdgcn_sched_barrier(0);
        const float4 b = bq[j];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[mb].x, b.x, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[mb].y, b.y, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[mb].z, b.z, acc[mb], 0, 0, 0);
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[mb].w, b.w, acc[mb], 0, 0, 0);
        }
        bq[j] = *reinterpret_cast<const float4*>(wnext + j * jstride);
        __builtin_amdgcn_sched_barrier(0);
      }
      tx = ntx; ty = nty;
      if constexpr (KIND == 4) if ((tap & 3) == 3) {
        // one output parity is complete: store it (registers -> HBM; the input tile stays in
        // LDS for the next parity), emit its statistics partial, restart the accumulators
        par = tap >> 2; py = par >> 1; px = par & 1;
        conv_epilogue<MB, KIND>(p, acc, tile + NP * RS, nullptr, par, py, px, img0, tin, y0, x0, (tq >> 7) & 1, (tq >> 6) & 1, tq & 63, (tq >> 5) & 1, tq, tq >> 6, bias_lane, true);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
      }
    }
    }  // !WINO
  }

  DM_STAMP(2);
  DM_TL(3);
  __builtin_amdgcn_s_setprio(DM_PRIO_STAGE);
  if constexpr (WINO) {
    __syncthreads();  // every wave is done reading the input tile: LDS is reused from here
    wino_epilogue(p, acc, smem_b, img0, tin, y0, x0, tq);
  } else if constexpr (KIND != 4) {
    __syncthreads();  // every wave is done reading the input tile: LDS is reused from here
    conv_epilogue<MB, KIND>(p, acc, smem + MT * 68, smem, par, py, px, img0, tin, y0, x0, (tq >> 7) & 1, (tq >> 6) & 1, tq & 63, (tq >> 5) & 1, tq, tq >> 6, bias_lane, true);
  }
  __syncthreads();  // LDS (output staging / reduction words) is free again for the next tile
  DM_TL(4);
  }  // tile loop
  DM_STAMP(3);
#if defined(DM_STAMPS) && !defined(DM_TIMELINE)
  if (p.stamps && threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + 6] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
}

// Pack a conv weight into MFMA B-fragment order (see conv_mfma_kernel).
//   KIND 0/1: w[cout][cin][kh][kw] (Conv2d OIHW);  KIND 2: w[cin][cout][kh][kw] (ConvTranspose2d)
// out[par][tap][cin/8][cout/32][lane][e] = w(cout = nb*32 + (lane&31), cin = 8*c8 + 4*(lane>>5) + e, ky, kx)
__global__ void pack_conv_weight_kernel(int kind, const float* __restrict__ w, int cin, int cout,
                                        float* __restrict__ out, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nty = kind == 0 ? 3 : kind == 1 ? 4 : kind == 2 ? 2 : 1;
  const int nt = nty * nty;
  const int ks = kind == 0 ? 3 : kind == 3 ? 1 : 4;
  size_t r = idx;
  const int e = r & 3; r >>= 2;
  const int lane = r & 63; r >>= 6;
  const int nb = r % (cout >> 5); r /= (cout >> 5);
  const int c8 = r % (cin >> 3); r /= (cin >> 3);
  const int tap = r % nt; r /= nt;
  const int par = (int)r;
  const int ty = tap / nty, tx = tap % nty;
  int ky = ty, kx = tx;
  if (kind == 2) { ky = 3 - (par >> 1) - 2 * ty; kx = 3 - (par & 1) - 2 * tx; }
  const int co = nb * 32 + (lane & 31);
  const int ci = c8 * 8 + 4 * (lane >> 5) + e;
  const size_t src = (kind == 2) ? (((size_t)ci * cout + co) * ks + ky) * ks + kx
                                 : (((size_t)co * cin + ci) * ks + ky) * ks + kx;
  out[idx] = w[src];
}

// Winograd F(2x2,3x3) weight transform U = G g G^T (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]) of a
// Conv2d 3x3 weight w[cout][cin][3][3], packed in B-fragment order for conv_mfma_kernel<.,5,.>:
//   out[cin/8][point = i*4+j][cout/32][lane][e] = U[i][j](cout = nb*32 + (lane&31), cin = 8*c8 + 4*(lane>>5) + e)
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, int cin, int cout, float* __restrict__ out, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  size_t r = idx;
  const int e = r & 3; r >>= 2;
  const int lane = r & 63; r >>= 6;
  const int nb = r % (cout >> 5); r /= (cout >> 5);
  const int pt = r & 15; r >>= 4;
  const int c8 = (int)r;
  const int i = pt >> 2, j = pt & 3;
  const int co = nb * 32 + (lane & 31);
  const int ci = c8 * 8 + 4 * (lane >> 5) + e;
  const float* g = w + ((size_t)co * cin + ci) * 9;
  float t[3];  // (G g)[i][kx]
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const float a = g[kx], b = g[3 + kx], c = g[6 + kx];
    t[kx] = (i == 0) ? a : (i == 1) ? 0.5f * ((a + b) + c) : (i == 2) ? 0.5f * ((a - b) + c) : c;
  }
  out[idx] = (j == 0) ? t[0] : (j == 1) ? 0.5f * ((t[0] + t[1]) + t[2]) : (j == 2) ? 0.5f * ((t[0] - t[1]) + t[2]) : t[2];
}

// ------------------------------------------------------------------------------------------
// init_conv (model.py:90,135): Conv2d 3x3 p1 from the NCHW image (Cin = 1..4) to NHWC C channels.
// K = 9*Cin is far too small for MFMA; this kernel is bound by the HBM write of its output.
__global__ __launch_bounds__(256) void init_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int B, int Cin, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;  // [Cin*9][C]
  for (int i = threadIdx.x; i < Cin * 9 * C; i += 256) {
    const int co = i % C, k = i / C;
    wl[i] = w[(size_t)co * Cin * 9 + k];
  }
  __syncthreads();
  const int V = C >> 2;
  const size_t total = (size_t)B * H * W * V;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c4 = idx % V;
    const size_t pix = idx / V;
    const int xx = pix % W;
    const int yy = (pix / W) % H;
    const size_t n = pix / ((size_t)W * H);
    float4 a = *reinterpret_cast<const float4*>(bias + c4 * 4);
    for (int ci = 0; ci < Cin; ++ci) {
      const float* xp = x + (n * Cin + ci) * (size_t)H * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = yy + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = xx + kx - 1;
          float xv = 0.f;
          if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) xv = xp[iy * W + ix];
          const float4 wv = *reinterpret_cast<const float4*>(wl + (ci * 9 + ky * 3 + kx) * C + c4 * 4);
          a.x += xv * wv.x; a.y += xv * wv.y; a.z += xv * wv.z; a.w += xv * wv.w;
        }
      }
    }
    *reinterpret_cast<float4*>(y + pix * C + c4 * 4) = a;
  }
}

// Sum over the 16 lanes of a DPP row, in every lane, without the LDS crossbar (ds_bpermute): quad_perm [1,0,3,2], quad_perm
// [2,3,0,1], row_half_mirror, row_mirror.  (__shfl_xor compiles to ds_bpermute_b32: one LDS instruction per step and value.)
__device__ __forceinline__ float dm_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// The sampler update (schedulers.py:118-129 == sampler.py:121-132) folded into final_conv's store (C = 64, no guidance): the
// thread that holds eps of an (image, channel, pixel) reads x (and the step's noise) there and writes the next x instead of eps -
// one launch and an eps round trip less per step.  Same operations in the same order as sampler_step_kernel, no FMA
// contraction: the result is bit-identical to the two-kernel form.
struct FinalStep {
  const float* x;        // null: plain final_conv (eps is written)
  const float* noise;    // null or [*][count]; row noise_idx
  const float* coefs;    // device table [n][8]
  const int* step_ptr;
  float* xout;           // may alias x
  size_t count;          // elements of x
  int step_off, noise_idx;
};
__device__ __forceinline__ float dm_step_math(float x, float e, float z, bool with_noise, float c0, float c1, float c2, float c3, float c4) {
#pragma clang fp contract(off)
  float o = c2 * ((x - c0 * e) / c1) + c3 * e;
  if (with_noise) o = o + c4 * z;
  return o;
}

// ------------------------------------------------------------------------------------------
// final_conv (model.py:111-115,151): GroupNorm(8) -> SiLU -> Conv2d 1x1 (C -> Cout <= 4), NHWC in,
// NCHW out.  One pass over the activation: bound by the HBM read of u3 (4*C bytes per pixel).
// Block = 256 threads, 64 pixels of one image; LP = min(C/4,64) lanes cooperate on a pixel.
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ u, const float* __restrict__ stats,
                                                         int pin, float cnt_in, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int B, int HW, int C, int Cout, int slabs, const FinalStep fs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_mean = smem;           // [C]
  float* s_scale = s_mean + C;    // [C]
  float* s_beta = s_scale + C;    // [C]
  float* s_w = s_beta + C;        // [Cout][C]
  // a block handles `slabs` 64-pixel slabs of one image: the table prologue (dependent global loads)
  // is paid once per 16*slabs KiB streamed instead of once per 16 KiB
  const int blocks_img = HW / (64 * slabs);
  const int n = blockIdx.x / blocks_img;
  const int pixb = (blockIdx.x % blocks_img) * 64 * slabs;
  const int G = C / DM_GROUPS;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* st = stats + ((size_t)n * pin * DM_GROUPS + c / G) * 2;
    float msum = 0.f, mean, m2 = 0.f, dev = 0.f;
    if (pin <= 8) {  // all partials in one round of independent loads (same arithmetic)
      float2 pr[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) pr[q] = *reinterpret_cast<const float2*>(st + (q < pin ? q : 0) * DM_GROUPS * 2);
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q < pin) msum += pr[q].x;
      mean = msum / (float)pin;
#pragma unroll
      for (int q = 0; q < 8; ++q) if (q < pin) {
        const float d = pr[q].x - mean;
        m2 += pr[q].y;
        dev += d * d;
      }
    } else {
      for (int q = 0; q < pin; ++q) msum += st[q * DM_GROUPS * 2];
      mean = msum / (float)pin;
      for (int q = 0; q < pin; ++q) {
        const float d = st[q * DM_GROUPS * 2] - mean;
        m2 += st[q * DM_GROUPS * 2 + 1];
        dev += d * d;
      }
    }
    const float var = (m2 + cnt_in * dev) / (cnt_in * (float)pin);
    s_mean[c] = mean;
    s_scale[c] = rsqrtf(var + DM_GN_EPS) * gamma[c];
    s_beta[c] = beta[c];
  }
  for (int i = threadIdx.x; i < Cout * C; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int V = C >> 2;
  const int LP = V < 64 ? V : 64;      // lanes per pixel (power of two)
  const int ppp = 256 / LP;            // pixels per pass
  const int q = threadIdx.x % LP;
  if (C == 64) {
    // C = 64: 16 lanes per pixel, 4 pixels per thread and slab.  One load per thread in flight keeps only half of
    // the bytes in the air that 8 TB/s x 2 us needs (Little), so a slab's four loads are issued together and the
    // next slab's before the current one is reduced.  Table entries and weights live in registers.
    const int c = q * 4, p0 = threadIdx.x >> 4;
    const float4 mu = *reinterpret_cast<const float4*>(s_mean + c);
    const float4 sc = *reinterpret_cast<const float4*>(s_scale + c);
    const float4 be = *reinterpret_cast<const float4*>(s_beta + c);
    float4 wv[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) wv[co] = (co < Cout) ? *reinterpret_cast<const float4*>(s_w + co * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* ub = u + ((size_t)n * HW + pixb + p0) * C + c;
    // Output side: the 16 sums of a thread's (pixel k, channel co) pairs end up in all 16 lanes of the pixel's row; lane q keeps
    // the one of (k, co) = (q >> 2, q & 3) and stores it - one store per lane and slab.  The fused sampler update loads its x
    // (and noise) element a slab ahead, behind the activation prefetch, so that nothing waits on it.
    const int mk = q >> 2, mco = q & 3;
    const bool mine = mco < Cout;
    const size_t obase = ((size_t)n * Cout + (mine ? mco : 0)) * HW + pixb + p0 + 16 * mk;
    const float mbias = bias[mine ? mco : 0];
    float sc0 = 0.f, sc1 = 1.f, sc2 = 0.f, sc3 = 0.f, sc4 = 0.f;  // the step's coefficients (fused sampler update)
    const float* nzr = nullptr;
    const bool stepping = fs.x != nullptr;
    if (stepping) {
      const float* cc = fs.coefs + (size_t)((fs.step_ptr ? *fs.step_ptr : 0) + fs.step_off) * 8;
      sc0 = cc[0]; sc1 = cc[1]; sc2 = cc[2]; sc3 = cc[3]; sc4 = cc[4];
      if (fs.noise) nzr = fs.noise + (size_t)fs.noise_idx * fs.count;
    }
    float4 cur[4], nxt[4];
    float xcur = 0.f, zcur = 0.f, xnxt = 0.f, znxt = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = *reinterpret_cast<const float4*>(ub + (size_t)(16 * k) * C);
    if (stepping && mine) {
      xcur = fs.x[obase];
      if (nzr) zcur = nzr[obase];
    }
    for (int sl = 0; sl < slabs; ++sl) {
      const int sn = (sl + 1 < slabs) ? sl + 1 : sl;  // (clamped: the last prefetch re-reads valid data, unused)
#pragma unroll
      for (int k = 0; k < 4; ++k) nxt[k] = *reinterpret_cast<const float4*>(ub + (size_t)(sn * 64 + 16 * k) * C);
      if (stepping && mine) {
        xnxt = fs.x[obase + sn * 64];
        if (nzr) znxt = nzr[obase + sn * 64];
      }
      float keep = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 v = cur[k];
        v.x = dm_silu((v.x - mu.x) * sc.x + be.x);
        v.y = dm_silu((v.y - mu.y) * sc.y + be.y);
        v.z = dm_silu((v.z - mu.z) * sc.z + be.z);
        v.w = dm_silu((v.w - mu.w) * sc.w + be.w);
#pragma unroll
        for (int co = 0; co < 4; ++co) {
          if (co < Cout) {
            float sm = v.x * wv[co].x + v.y * wv[co].y + v.z * wv[co].z + v.w * wv[co].w;
            sm = dm_row16_sum(sm);  // the pixel's 16 lanes are one DPP row; every lane holds the same sum
            keep = (q == k * 4 + co) ? sm : keep;
          }
        }
      }
      if (mine) {
        const float e = keep + mbias;
        if (stepping) fs.xout[obase + sl * 64] = dm_step_math(xcur, e, zcur, nzr != nullptr, sc0, sc1, sc2, sc3, sc4);
        else out[obase + sl * 64] = e;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
      xcur = xnxt;
      zcur = znxt;
    }
    return;
  }
  for (int sl = 0; sl < slabs; ++sl) {
  const int pix0 = pixb + sl * 64;
  for (int pp = threadIdx.x / LP; pp < 64; pp += ppp) {
    const float* up = u + ((size_t)n * HW + pix0 + pp) * C;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = q; c4 < V; c4 += LP) {
      const int c = c4 * 4;
      float4 v = *reinterpret_cast<const float4*>(up + c);
      const float4 mu = *reinterpret_cast<const float4*>(s_mean + c);
      const float4 sc = *reinterpret_cast<const float4*>(s_scale + c);
      const float4 be = *reinterpret_cast<const float4*>(s_beta + c);
      v.x = dm_silu((v.x - mu.x) * sc.x + be.x);
      v.y = dm_silu((v.y - mu.y) * sc.y + be.y);
      v.z = dm_silu((v.z - mu.z) * sc.z + be.z);
      v.w = dm_silu((v.w - mu.w) * sc.w + be.w);
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        if (co < Cout) {
          const float4 wv = *reinterpret_cast<const float4*>(s_w + co * C + c);
          part[co] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        }
      }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      if (co < Cout) {
        float s = part[co];
        for (int o = LP >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (q == 0) out[((size_t)n * Cout + co) * HW + pix0 + pp] = s + bias[co];
      }
    }
  }
  }  // slabs
}

// ------------------------------------------------------------------------------------------
// Time embedding (model.py:18-30 sinusoid; :82-87 MLP; :56 per-block heads).  One block per
// timestep entry; fp32 GEMVs, accurate sinf/cosf/expf (arguments reach 999 rad).
struct TimeArgs {
  const float* freqs;            // [E/2]
  const float *w1, *b1, *w2, *b2;  // [E,E],[E]
  const float* hw[6];            // [C,E]
  const float* hb[6];            // [C]
  const int64_t* t;              // timestep of entry n: t[tbase + n * t_stride]
  const int* step_ptr;           // null or device step counter; tbase = *step_ptr + step_off
  int step_off, t_stride;
  const float* cond;             // null or [nt,E]
  const float* temb_in;          // null, or the time MLP's output per STEP [*, E] (row tbase), computed once up front: entries then are
                                 // temb_in[tbase] + cond[n] and only the six heads are evaluated (guided sampling: all 2B entries of a step share t)
  float* temb;                   // null or [nt,E]
  float* heads;                  // [nt,6,C]
  int E, C;
};

// One block evaluates DM_TIME_ROWS consecutive entries: every weight row is read once per block and used for all of
// them (with guidance there are 2B entries per step: one entry per block re-read the ~330 KB of weights 2B times, 81 us
// per step at B = 1024).  Per entry the arithmetic and its order are unchanged.
#define DM_TIME_ROWS 8
__global__ __launch_bounds__(256) void time_embed_kernel(const TimeArgs a, int nt) {
  constexpr int R = DM_TIME_ROWS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e = smem;            // [R][E]
  float* hbuf = e + R * a.E;  // [R][E]
  float* te = hbuf + R * a.E; // [R][E]
  const int n0 = blockIdx.x * R;
  const int nr = min(R, nt - n0);
  const int half = a.E >> 1;
  const int tbase = (a.step_ptr ? *a.step_ptr : 0) + a.step_off;
  if (a.temb_in) {  // (uniform) same value and order as below: (W2 h + b2), then + cond
    for (int idx = threadIdx.x; idx < R * a.E; idx += 256) {
      const int r = idx / a.E, o = idx - r * a.E;
      float v = a.temb_in[(size_t)tbase * a.E + o];
      if (r < nr) {
        if (a.cond) v += a.cond[(size_t)(n0 + r) * a.E + o];
        if (a.temb) a.temb[(size_t)(n0 + r) * a.E + o] = v;
      }
      te[r * a.E + o] = v;
    }
  } else {
  for (int idx = threadIdx.x; idx < nr * half; idx += 256) {
    const int r = idx / half, i = idx - r * half;
    const float tv = (float)a.t[tbase + (size_t)(n0 + r) * a.t_stride];
    const float arg = tv * a.freqs[i];
    e[r * a.E + i] = sinf(arg);
    e[r * a.E + i + half] = cosf(arg);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < a.E; o += 256) {
    const float* wr = a.w1 + (size_t)o * a.E;
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.f;
    for (int i = 0; i < a.E; ++i) {
      const float w = wr[i];
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] += w * e[r * a.E + i];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = s[r] + a.b1[o];
      hbuf[r * a.E + o] = v / (1.0f + expf(-v));
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < a.E; o += 256) {
    const float* wr = a.w2 + (size_t)o * a.E;
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.f;
    for (int i = 0; i < a.E; ++i) {
      const float w = wr[i];
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] += w * hbuf[r * a.E + i];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v = s[r] + a.b2[o];
      if (r < nr) {
        if (a.cond) v += a.cond[(size_t)(n0 + r) * a.E + o];
        if (a.temb) a.temb[(size_t)(n0 + r) * a.E + o] = v;
      }
      te[r * a.E + o] = v;
    }
  }
  }  // !temb_in
  __syncthreads();
  for (int o = threadIdx.x; o < 6 * a.C; o += 256) {
    const int k = o / a.C, c = o - k * a.C;
    const float* wr = a.hw[k] + (size_t)c * a.E;
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.f;
    for (int i = 0; i < a.E; ++i) {
      const float w = wr[i];
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] += w * te[r * a.E + i];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nr) {
        const float v = s[r] + a.hb[k][c];
        a.heads[(size_t)(n0 + r) * 6 * a.C + o] = v / (1.0f + expf(-v));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fused sampler update (schedulers.py:113-131 == sampler.py:117-134), one pass:
//   x0 = (x - c0*eps)/c1 ; out = c2*x0 + c3*eps (+ c4*noise)
// in the reference's operation order with no FMA contraction, so it is bit-exact against
// torch-CPU.  HBM-bound: 12 B/element (eta = 0) or 16 B/element.
// Optional classifier-free guidance (build-defined): eps = eps_u + g*(eps_c - eps_u) with
// eps_c = eps[0:count], eps_u = eps[count:2*count].
struct StepArgs {
  const float* x;
  const float* eps;
  const float* noise;      // null or [*][count]; this launch reads row noise_idx
  const float* coefs;      // null or device table [n][8]
  const int* step_ptr;
  float* out;
  size_t count;            // elements (multiple of 4)
  float c[5];              // by-value coefficients when coefs == null
  float guidance;
  int cfg;
  int step_off;
  int noise_idx;
  size_t img_count;        // > 0: coefs is a per-image table [B][8], img_count elements per image ([B] timesteps)
};

__global__ __launch_bounds__(256) void sampler_step_kernel(const StepArgs a) {
#pragma clang fp contract(off)
  float c0 = a.c[0], c1 = a.c[1], c2 = a.c[2], c3 = a.c[3], c4 = a.c[4];
  const float* nz = a.noise;
  if (a.coefs && !a.img_count) {
    const int step = (a.step_ptr ? *a.step_ptr : 0) + a.step_off;
    const float* cc = a.coefs + (size_t)step * 8;
    c0 = cc[0]; c1 = cc[1]; c2 = cc[2]; c3 = cc[3]; c4 = cc[4];
  }
  if (nz) nz += (size_t)a.noise_idx * a.count;
  const size_t n4 = a.count >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    if (a.img_count) {  // schedulers.py:113-115 with a [B] timestep tensor: `.view(-1,1,1,1)` coefficients
      const float* cc = a.coefs + ((i * 4) / a.img_count) * 8;
      c0 = cc[0]; c1 = cc[1]; c2 = cc[2]; c3 = cc[3]; c4 = cc[4];
    }
    const float4 x = reinterpret_cast<const float4*>(a.x)[i];
    float4 e = reinterpret_cast<const float4*>(a.eps)[i];
    if (a.cfg) {
      const float4 u = reinterpret_cast<const float4*>(a.eps + a.count)[i];
      e.x = u.x + a.guidance * (e.x - u.x);
      e.y = u.y + a.guidance * (e.y - u.y);
      e.z = u.z + a.guidance * (e.z - u.z);
      e.w = u.w + a.guidance * (e.w - u.w);
    }
    float4 o;
    o.x = c2 * ((x.x - c0 * e.x) / c1) + c3 * e.x;
    o.y = c2 * ((x.y - c0 * e.y) / c1) + c3 * e.y;
    o.z = c2 * ((x.z - c0 * e.z) / c1) + c3 * e.z;
    o.w = c2 * ((x.w - c0 * e.w) / c1) + c3 * e.w;
    if (nz) {
      const float4 z = reinterpret_cast<const float4*>(nz)[i];
      o.x = o.x + c4 * z.x; o.y = o.y + c4 * z.y; o.z = o.z + c4 * z.z; o.w = o.w + c4 * z.w;
    }
    reinterpret_cast<float4*>(a.out)[i] = o;
  }
}

// ClassConditioner.forward (diffusion/conditioning.py:39-55): Embedding gather -> Linear ->
// SiLU -> Linear.  One block per label.
__global__ __launch_bounds__(256) void class_embed_kernel(const float* __restrict__ table, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, const int64_t* __restrict__ labels,
                                                          int num_rows, int E, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e = smem;
  float* hbuf = smem + E;
  const int n = blockIdx.x;
  long long lab = labels[n];
  if (lab < 0) lab = 0;
  if (lab >= num_rows) lab = num_rows - 1;  // host validates; clamp keeps the access in bounds
  for (int i = threadIdx.x; i < E; i += 256) e[i] = table[(size_t)lab * E + i];
  __syncthreads();
  for (int o = threadIdx.x; o < E; o += 256) {
    const float* wr = w1 + (size_t)o * E;
    float s = 0.f;
    for (int i = 0; i < E; ++i) s += wr[i] * e[i];
    s += b1[o];
    hbuf[o] = s / (1.0f + expf(-s));
  }
  __syncthreads();
  for (int o = threadIdx.x; o < E; o += 256) {
    const float* wr = w2 + (size_t)o * E;
    float s = 0.f;
    for (int i = 0; i < E; ++i) s += wr[i] * hbuf[i];
    out[(size_t)n * E + o] = s + b2[o];
  }
}

// ------------------------------------------------------------------------------------------
// Attention block (build-defined, BASELINE configs[4]): batched fp32-MFMA GEMM and row softmax.
// C[b] = alpha * A[b] (MxK, row-major, lda) * B[b], with B given as [N][K] (BKN = false: Q K^T)
// or [K][N] (BKN = true: P V).  64x64 tile per workgroup, 4 waves x one 32x32 accumulator,
// K chunks of 32 staged in LDS k-contiguous for both operands (same b128 fragment scheme as the
// conv kernel).  M, N multiples of 64, K multiple of 32.  <1 % of the widened model's FLOPs.
struct GemmArgs {
  const float* A; const float* Bm; float* C;
  int M, N, K, lda, ldb, ldc;
  size_t sA, sB, sC;
  float alpha;
};

template <bool BKN>
__global__ __launch_bounds__(256) void bgemm_mfma_kernel(const GemmArgs g) {
  constexpr int RS = 36;
  __shared__ __attribute__((aligned(16))) float As[64 * RS];
  __shared__ __attribute__((aligned(16))) float Bs[64 * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* A = g.A + blockIdx.z * g.sA;
  const float* Bm = g.Bm + blockIdx.z * g.sB;
  float* C = g.C + blockIdx.z * g.sC;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * 256;
      const int row = idx >> 3, c4 = idx & 7;
      *reinterpret_cast<float4*>(As + row * RS + c4 * 4) =
          *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * g.lda + k0 + c4 * 4);
      if (!BKN) {
        *reinterpret_cast<float4*>(Bs + row * RS + c4 * 4) =
            *reinterpret_cast<const float4*>(Bm + (size_t)(n0 + row) * g.ldb + k0 + c4 * 4);
      } else {
        const int k = idx >> 4, n4 = idx & 15;
        const float4 v = *reinterpret_cast<const float4*>(Bm + (size_t)(k0 + k) * g.ldb + n0 + n4 * 4);
        Bs[(n4 * 4 + 0) * RS + k] = v.x; Bs[(n4 * 4 + 1) * RS + k] = v.y;
        Bs[(n4 * 4 + 2) * RS + k] = v.z; Bs[(n4 * 4 + 3) * RS + k] = v.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + (lane & 31)) * RS + j * 8 + 4 * h);
      const float4 b = *reinterpret_cast<const float4*>(Bs + (wn * 32 + (lane & 31)) * RS + j * 8 + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    C[(size_t)row * g.ldc + n0 + wn * 32 + (lane & 31)] = g.alpha * acc[r];
  }
}

// In-place softmax over rows of length n (one wave per row).
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, size_t rows, int n) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* r = s + row * n;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, r[i]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float e = expf(r[i] - mx);
    r[i] = e;
    sum += e;
  }
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.0f / sum;
  for (int i = lane; i < n; i += 64) r[i] *= inv;
}

// Fused single-head attention for one (image, 64-query tile): S = scale * Q K^T -> row softmax -> O = P V,
// with the 64 x T score tile held in LDS (T <= 256 tokens: 65 KiB) - the scores never travel to HBM.
//   qkv [B][T][3C] (q | k | v of a token, from the 1x1 qkv conv), o [B][T][C].
// 4 waves as 2 (query rows) x 2 (columns); fp32 MFMA 32x32x2 with the conv kernels' k-permuted b128 fragment
// reads.  NB = T/64 score blocks and CB = C/64 output blocks per wave.
//   phase 1: K loop over C in chunks of 32 channels: Q chunk [64][36] and K chunk [T][36] staged in LDS
//   phase 2: softmax of the 64 rows in LDS (16 rows per wave, shuffles), accurate expf
//   phase 3: K loop over T in chunks of 32 tokens: A = P from LDS, B = V chunk staged TRANSPOSED as [C][36]
template <int NB, int CB>
__global__ __launch_bounds__(256) void attention_fused_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                              int C, float scale) {
  constexpr int T = 64 * NB;
  constexpr int PRS = T + 4;  // score row stride (floats)
  constexpr int RS = 36;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* P = smem;                 // [64][PRS]
  float* As = P + 64 * PRS;        // [64][RS]
  float* Bs = As + 64 * RS;        // [max(T, C)][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int q0 = blockIdx.x * 64;
  const size_t ld = (size_t)3 * C;
  const float* base = qkv + (size_t)blockIdx.y * T * ld;
  // ---- phase 1: S = Q K^T.  Software-pipelined: the NEXT chunk's global loads are issued into registers before
  // the MFMAs of the current chunk and written to LDS after them.
  {
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    float4 qn[2], kn[2 * NB];
    auto fetch = [&](int c0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // Q chunk: 64 rows x 8 float4
        const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
        qn[u] = *reinterpret_cast<const float4*>(base + (size_t)(q0 + row) * ld + c0 + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 2 * NB; ++u) {  // K chunk: T rows x 8 float4
        const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
        kn[u] = *reinterpret_cast<const float4*>(base + (size_t)row * ld + C + c0 + c4 * 4);
      }
    };
    fetch(0);
    for (int c0 = 0; c0 < C; c0 += 32) {
      __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(As + row * RS + c4 * 4) = qn[u];
      }
#pragma unroll
      for (int u = 0; u < 2 * NB; ++u) {
        const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(Bs + row * RS + c4 * 4) = kn[u];
      }
      __syncthreads();
      if (c0 + 32 < C) fetch(c0 + 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(As + (wm * 32 + l31) * RS + j * 8 + 4 * h);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float4 b = *reinterpret_cast<const float4*>(Bs + (wn * (T / 2) + nb * 32 + l31) * RS + j * 8 + 4 * h);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[nb], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        P[row * PRS + wn * (T / 2) + nb * 32 + l31] = scale * acc[nb][r];
      }
  }
  __syncthreads();
  // ---- phase 2: softmax over each of the 64 rows (16 rows per wave; a lane holds NB elements of a row)
  for (int rr = 0; rr < 16; ++rr) {
    float* pr = P + (wave * 16 + rr) * PRS;
    float v[NB];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NB; ++k) { v[k] = pr[lane + 64 * k]; mx = fmaxf(mx, v[k]); }
    for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
    for (int o2 = 32; o2 > 0; o2 >>= 1) sum += __shfl_xor(sum, o2, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < NB; ++k) pr[lane + 64 * k] = v[k] * inv;
  }
  // ---- phase 3: O = P V (same pipelining)
  {
    f32x16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    float4 vn[2 * CB];
    // V chunk [32 tokens][C]: lane -> token, 16 B per lane from global
    auto fetch = [&](int t0) {
#pragma unroll
      for (int u = 0; u < 2 * CB; ++u) {
        const int idx = tid + u * 256, k = idx & 31, quad = idx >> 5;
        vn[u] = *reinterpret_cast<const float4*>(base + (size_t)(t0 + k) * ld + 2 * C + quad * 4);
      }
    };
    fetch(0);
    for (int t0 = 0; t0 < T; t0 += 32) {
      __syncthreads();  // (first round: the softmax writes; later: the previous round's B reads)
      // -> Bs[c][token] (transposed; consecutive lanes = consecutive tokens: conflict-free 4-byte LDS writes)
#pragma unroll
      for (int u = 0; u < 2 * CB; ++u) {
        const int idx = tid + u * 256, k = idx & 31, quad = idx >> 5;
        Bs[(quad * 4 + 0) * RS + k] = vn[u].x; Bs[(quad * 4 + 1) * RS + k] = vn[u].y;
        Bs[(quad * 4 + 2) * RS + k] = vn[u].z; Bs[(quad * 4 + 3) * RS + k] = vn[u].w;
      }
      __syncthreads();
      if (t0 + 32 < T) fetch(t0 + 32);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(P + (wm * 32 + l31) * PRS + t0 + j * 8 + 4 * h);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const float4 b = *reinterpret_cast<const float4*>(Bs + (wn * (32 * CB) + cb * 32 + l31) * RS + j * 8 + 4 * h);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[cb], 0, 0, 0);
        }
      }
    }
    float* ob = o + ((size_t)blockIdx.y * T + q0) * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        ob[(size_t)row * C + wn * (32 * CB) + cb * 32 + l31] = acc[cb][r];
      }
  }
}

// Post-processing of finished samples (diffusion/sample.py:117-119): (x + 1) / 2, clamp to [0,1];
// optionally quantised to uint8 as torchvision.utils.save_image does (v*255 + 0.5, clamp, truncate).
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ x, float* __restrict__ out_f,
                                                          unsigned char* __restrict__ out_u8, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    float v = (x[i] + 1.0f) / 2.0f;
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    if (out_f) out_f[i] = v;
    if (out_u8) out_u8[i] = (unsigned char)fminf(fmaxf(v * 255.0f + 0.5f, 0.0f), 255.0f);
  }
}

__global__ void advance_step_kernel(int* step_ptr, int by) { if (threadIdx.x == 0) *step_ptr += by; }
__global__ void set_step_kernel(int* step_ptr, int v) { if (threadIdx.x == 0) *step_ptr = v; }

// duplicate x [count] into xx [2*count] (CFG doubled batch)
__global__ void dup_kernel(const float* __restrict__ x, float* __restrict__ xx, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    xx[i] = v;
    xx[i + count] = v;
  }
}
