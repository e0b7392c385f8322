// dm_kernels_lowres.hip — the levels at and below 8x8 pixels of the 32x32 model in ONE launch.
//
// model.py:139-147 between down2's second GroupNorm and up2's output: down2.transform (16x16 -> 8x8), down3 (conv1, conv2 at
// 8x8, transform 8x8 -> 4x4), the bottleneck pair, up1 (conv1 over cat[bott, down3], conv2 at 4x4, ConvTranspose 4x4 -> 8x8),
// up2 (conv1 over cat[up1, down2], conv2 at 8x8, ConvTranspose 8x8 -> 16x16): twelve convolutions, seven GroupNorm+SiLU
// (+ time embedding) transforms.  As separate launches each of them is one latency-bound tile per workgroup (64 images: 32
// or 16 workgroups on a 256-CU chip; 219 of the 432 us of an ancestral-1000 step at B = 64).  Here ONE workgroup takes ONE
// image through all twelve layers: every activation between the 16x16 input and the 16x16 output stays on the CU - as the
// three-bf16-term planes the next layer's matrix instructions read, in LDS - the GroupNorm statistics are whole-image sums
// taken from the accumulator registers, and the transform of a layer's output runs in the registers that hold it.
// Only down2's 8x8 output makes a round trip through global memory (16 KiB per image: it is needed again, as the skip input
// of up2.conv1, when its LDS planes have long been reused).
//
// Arithmetic: conv_x6_kernel's - every fp32 operand as three bf16 terms, six v_mfma_f32_32x32x16_bf16 per product, small
// terms first, fp32 accumulation (fp32-equivalent accuracy) -, weights from the model's split-bf16
// packs ([split][parity][tap][Cin/16][Cout/32][64 lanes][8]), swapped operands (weights as A, pixels as B) so a lane holds
// one pixel and four runs of four consecutive channels.  Workgroup = 4 waves: at 8x8 as 2 (pixel halves) x 2 (channel
// halves); at 4x4 (16 pixels: half a 32-row block) the pixel-half index splits K instead and the two partial sums meet
// through LDS, or - transposed conv 4x4 -> 8x8 - it splits the four output parities.
// NA = 2 (the fp32 path's default): every operand as TWO fp16 terms, three v_mfma_f32_32x32x16_f16 per product (dm_split_h2,
// dm_kernels_x6.hip: as exact as the three-way bf16 split for operands of order 1).  Layer outputs that went through a
// GroupNorm are of order 1 by construction; the five tensors a consumer reads RAW (down2's, down3's, the bottleneck's, up1's
// outputs and down2's copy from global memory) are scaled by a power of two taken from the tensor's largest magnitude
// (whole-image maximum over the workgroup, exact) before they are split, and the consumer's accumulators are scaled back
// together with the weight pack's own power of two - so the form holds for activations of any magnitude.
// NA = 1 is the bf16 path's form (BASELINE configs[3]): ONE bf16 activation plane, the weights as (hi, lo) bf16 pairs - the
// first two terms of the same packing -, two MFMAs per product; input as IEEE half, output as bf16 like the layer kernels.
// Chosen by the layer geometry only (C = 64, 32x32 images, fp32, no attention), never by the batch size: per-image results
// do not depend on the batch an image rides in, so multi-GPU shards stay bit-identical to the unsharded run.
// NA = 4 is the bf16 path's default form since round 3 (DM_BF16_F16 = 2): ONE IEEE-half activation plane and ONE half weight
// per product (the first plane of the two-fp16-term packs), one v_mfma_f32_32x32x16_f16 per product; raw tensors scaled by a
// power of two as for NA = 2, 16-bit input / output as for NA = 1.
#pragma once
#include "dm_kernels_bf16.hip"
#include "dm_kernels_x6.hip"

struct LrLayer {
  const bf16_t* w;     // split-bf16 packed weights
  const float* bias;   // [64]
  const float* gamma;  // GroupNorm applied to this layer's OUTPUT (null: the consumer reads it raw)
  const float* beta;
  int temb_k;          // time-embedding head added after that GroupNorm + SiLU (block index 0..5), -1: none
  const float* wsc;    // NA = 2: the float 2^-k behind the two fp16 planes of the pack (pack_conv kinds 13-15); else null
};

struct LowresArgs {
  const float* t2;       // [B,16,16,64] raw output of down2.conv2 (bf16 path: IEEE half, 2 bytes per element)
  const float* st2;      // its GroupNorm statistics partials [B][pin2][8][2]
  int pin2; float cnt2;
  const float* g2; const float* b2;  // down2.bnorm2
  // layers in execution order: down2.transform, down3.conv1, down3.conv2, down3.transform, bottleneck.0, bottleneck.3,
  // up1.conv1, up1.conv2, up1.transform, up2.conv1, up2.conv2, up2.transform
  LrLayer d2t, d3c1, d3c2, d3t, b0, b3, u1c1, u1c2, u1t, u2c1, u2c2, u2t;
  const float* heads; int heads_img_stride, heads_step_stride;  // [.., 6, 64] time-embedding heads (see unet_body)
  const int* step_ptr; int step_off;
  float* d2;             // [B,8,8,64] scratch: down2's raw output (skip input of up2.conv1)
  float* u2;             // [B,16,16,64] up2's output (bf16 path: bf16)
#ifdef DM_STAMPS
  unsigned long long* stamps;  // diagnostic build only: s_memtime stamps [grid][32] at entry 2^20 (the conv kernels stamp the front of the buffer) of the phases of the workgroup's LAST image
#endif
  float* amax_u2;        // null, or [B][4]: per-wave maxima of |u2| (the consumer's power-of-two scale, see ConvArgs.amax1)
  float* taps;           // null, or (tests) [B][2048 + 4096] raw fp32 NHWC: down3's output (4x4x64), the bottleneck's (4x4x64), up1's (8x8x64)
  int B;
};

#ifdef DM_STAMPS
#define LR_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(1 << 20) + (size_t)blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LR_STAMP(k) do {} while (0)
#endif

namespace lr {
constexpr int RSB = 72;            // bf16 elements per pixel row: 64 channels + 8 of skew (144 B)
constexpr int PL = 100 * RSB;      // plane stride of the main buffer P: 10x10 halo pixels / 4 phases x 5x5 / 9x9 phase
constexpr int PL4 = 36 * RSB;      // plane stride of the 4x4-level buffers: 6x6 halo pixels
constexpr int P_BYTES = 3 * PL * 2;      // 43,200
constexpr int Q_BYTES = 3 * PL4 * 2;     // 15,552
constexpr int X_OFF = 2 * Q_BYTES;       // 4x4 level: K-split exchange area inside P (behind E and F), 8 KiB
constexpr int X8_BYTES = 16384;          // 8x8 level: K-split exchange area = the D3p buffer (dead outside the 4x4 level) + 832 B
// small tables (floats): input-stage GroupNorm tables, reduction words, and every layer's bias / GroupNorm affine
// parameters / time-embedding rows, loaded once per workgroup (time embedding: per image) so that no epilogue waits on L2
constexpr int SM_SCALE = 0, SM_SHIFT = 64, SM_RED = 128, SM_BIAS = 192;   // bias [12][64]
constexpr int SM_GAMMA = SM_BIAS + 12 * 64, SM_BETA = SM_GAMMA + 7 * 64, SM_TEMB = SM_BETA + 7 * 64;  // [7][64], [7][64], [3][64]
constexpr int SM_WSC = SM_TEMB + 3 * 64;   // [12] weight-pack scales (NA = 2), padded to 16
constexpr int SM_FLOATS = SM_WSC + 16;
constexpr int LDS_BYTES = P_BYTES + X8_BYTES + SM_FLOATS * 4;
constexpr size_t JS = 1024;        // weight elements per (16 input channels, 64 output channels)

// One split-term product into a 32x32 accumulator.  -DDM_LR_DIAG16 (timing experiment only, results are garbage): two
// v_mfma_f32_16x16x32_bf16 on two quarters of the accumulator instead - same operand traffic, same MFMA cycles, the shape the
// chip clocks higher under its power cap (MI355X_MICROARCH.md, DVFS item 7).
typedef float lr_f32x4 __attribute__((ext_vector_type(4)));
#ifdef DM_LR_DIAG16
#define LR_MFMA(b, a, acc, sl)                                                                                     \
  do {                                                                                                             \
    lr_f32x4 c0_ = (sl) ? __builtin_shufflevector(acc, acc, 8, 9, 10, 11) : __builtin_shufflevector(acc, acc, 0, 1, 2, 3);   \
    lr_f32x4 c1_ = (sl) ? __builtin_shufflevector(acc, acc, 12, 13, 14, 15) : __builtin_shufflevector(acc, acc, 4, 5, 6, 7); \
    c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c0_, 0, 0, 0);                                             \
    c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1_, 0, 0, 0);                                             \
    if (sl) { acc[8] = c0_[0]; acc[9] = c0_[1]; acc[10] = c0_[2]; acc[11] = c0_[3]; acc[12] = c1_[0]; acc[13] = c1_[1]; acc[14] = c1_[2]; acc[15] = c1_[3]; } \
    else { acc[0] = c0_[0]; acc[1] = c0_[1]; acc[2] = c0_[2]; acc[3] = c0_[3]; acc[4] = c1_[0]; acc[5] = c1_[1]; acc[6] = c1_[2]; acc[7] = c1_[3]; } \
  } while (0)
#else
#define LR_MFMA(b, a, acc, sl) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0)
#endif

// A wave's weight stream of one layer: fragment base (its 32 output channels), tap stride, split-term stride.
struct WS { const bf16_t* wp; size_t ts, so; };
struct WFrag { uint4 q[4][3]; };  // the three split terms of the four 16-channel slabs of one tap

// what a form NA means: activation planes in LDS, weight terms fetched per fragment, fp16 operands with power-of-two scales
// (weight packs and raw tensors), 16-bit tensors at the kernel's boundary
constexpr int lr_planes(int NA) { return NA == 4 ? 1 : NA; }
constexpr int lr_wterms(int NA) { return NA == 3 ? 3 : NA == 4 ? 1 : 2; }
constexpr bool lr_scaled(int NA) { return NA == 2 || NA == 4; }
constexpr bool lr_io16(int NA) { return NA == 1 || NA == 4; }

// fetch the fragments of packed tap wt (L2): issued a whole layer epilogue ahead of their first use
template <int NA, int NJ = 4>
__device__ __forceinline__ void kload(WFrag& f, const WS& w, int wt, unsigned lane16) {
  constexpr int NB = lr_wterms(NA);  // weight terms: three-way split / two fp16 terms (fp32 path), (hi, lo) pair / one half (bf16 path)
  int opq = 0;
  asm volatile("" : "+v"(opq));  // (the address arithmetic stays here: see krun)
  lane16 += (unsigned)opq;
  const bf16_t* w0 = w.wp + (size_t)wt * w.ts;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int s = 0; s < NB; ++s)
      f.q[j][s] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w0 + s * w.so + (size_t)j * JS) + lane16);
}

// acc += sum_{t < ntaps} sum_{j < 4} W[wt(t)][j] (A operand) x plane pixels at aoff(t), channels 16 j .. (B operand), for the
// three split terms: six MFMAs per (tap, 16 channels), one memory instruction behind each of them - the LDS reads of the NEXT
// slab's pixel fragments (double buffer) and the L2 reads of the NEXT TAP's weight fragments into the slots just consumed (24
// MFMAs ahead: one slab ahead, ~200 cycles, does not cover an L2 hit).  f: the fragments of tap 0 (kload).  ap: the lane's
// fragment address in plane 0 at tap offset 0; tapf(t, aoff, wt): LDS offset and packed-weight tap of step t.
template <int NA, class F>
__device__ __forceinline__ void krun(f32x16& acc, WFrag& f, const bf16_t* ap, int pl, const WS& w, unsigned lane16, int ntaps, F tapf) {
  constexpr int NJ = 4;
  // opaque copies: the fragment addresses are formed HERE, per call - left alone hipcc hoists the address arithmetic of all
  // call sites out of the image loop and spills it (1.5 KB of scratch per lane)
  int opq = 0;
  asm volatile("" : "+v"(opq));
  lane16 += (unsigned)opq;
  ap += opq;
  uint4 a[2][3];
  int aoff, wt;
  tapf(0, aoff, wt);
#pragma unroll
  for (int s = 0; s < lr_planes(NA); ++s) a[0][s] = *reinterpret_cast<const uint4*>(ap + s * pl + aoff);
#pragma unroll 1
  for (int t = 0; t < ntaps; ++t) {
    int aoffn, wtn;
    tapf(t + 1 < ntaps ? t + 1 : t, aoffn, wtn);  // (clamped: the last prefetch re-reads valid data, unused)
    const bf16_t* wnext = w.wp + (size_t)wtn * w.ts;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = j & 1, nx = c ^ 1;
      const bf16_t* an = ap + ((j + 1 == NJ) ? aoffn : aoff + (j + 1) * 16);
      const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, a[c][0]);
      const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, f.q[j][0]), b2 = __builtin_bit_cast(bf16x8_t, f.q[j][1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NA == 3) {
        const bf16x8_t a2 = __builtin_bit_cast(bf16x8_t, a[c][1]), a3 = __builtin_bit_cast(bf16x8_t, a[c][2]);
        const bf16x8_t b3 = __builtin_bit_cast(bf16x8_t, f.q[j][2]);
        // pixel terms a (B operand), weight terms b (A operand); products a1b3, a2b2, a1b2, a3b1, a2b1, a1b1 (small terms first)
        LR_MFMA(b3, a1, acc, j & 1);
        a[nx][0] = *reinterpret_cast<const uint4*>(an);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, a2, acc, j & 1);
        a[nx][1] = *reinterpret_cast<const uint4*>(an + pl);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, a1, acc, j & 1);
        a[nx][2] = *reinterpret_cast<const uint4*>(an + 2 * pl);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, a3, acc, j & 1);
        f.q[j][2] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + 2 * w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, a2, acc, j & 1);
        f.q[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, a1, acc, j & 1);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      } else if constexpr (NA == 2) {  // two fp16 terms each: h1 g2, h2 g1, h1 g1
        acc = dm_mfma16<true>(f.q[j][1], a[c][0], acc);
        a[nx][0] = *reinterpret_cast<const uint4*>(an);
        __builtin_amdgcn_sched_barrier(0);
        acc = dm_mfma16<true>(f.q[j][0], a[c][1], acc);
        a[nx][1] = *reinterpret_cast<const uint4*>(an + pl);
        f.q[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        acc = dm_mfma16<true>(f.q[j][0], a[c][0], acc);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      } else if constexpr (NA == 4) {  // one half term each
        acc = dm_mfma16<true>(f.q[j][0], a[c][0], acc);
        a[nx][0] = *reinterpret_cast<const uint4*>(an);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      } else {  // bf16 activations x (hi, lo) weights: a1 lo, then a1 hi
        LR_MFMA(b2, a1, acc, j & 1);
        a[nx][0] = *reinterpret_cast<const uint4*>(an);
        f.q[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, a1, acc, j & 1);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    aoff = aoffn; wt = wtn;
  }
}

// krun for TWO 32-pixel blocks per wave (all 64 pixels of an 8x8 layer): each weight fragment is fetched once per workgroup
// instead of once per pixel half - the 8x8 layers' weight traffic (442 KB per layer and image with 2 x 2 waves) is what an
// L2 -> CU path of ~64 B/clk cannot deliver beside the MFMAs - and used for 12 MFMAs.  NJ = 2: the wave covers HALF of K
// (channel slabs 2kh, 2kh+1 of every tap; caller offsets ap and w.wp) and the partial sums meet through LDS (kx_exchange);
// NJ = 4: all of K.  ap0/ap1: the lane's fragment addresses for the two blocks.
template <int NA, int NJ, class F>
__device__ __forceinline__ void krun2(f32x16& acc0, f32x16& acc1, WFrag& f, const bf16_t* ap0, const bf16_t* ap1, int pl, const WS& w,
                                      unsigned lane16, int ntaps, F tapf) {
  static_assert(NJ == 2 || NJ == 4, "even slab count: the double buffer alternates per slab");
  int opq = 0;
  asm volatile("" : "+v"(opq));
  lane16 += (unsigned)opq;
  ap0 += opq;
  ap1 += opq;
  uint4 a[2][2][3];  // [buffer][block][split plane]
  int aoff, wt;
  tapf(0, aoff, wt);
#pragma unroll
  for (int s = 0; s < lr_planes(NA); ++s) {
    a[0][0][s] = *reinterpret_cast<const uint4*>(ap0 + s * pl + aoff);
    a[0][1][s] = *reinterpret_cast<const uint4*>(ap1 + s * pl + aoff);
  }
#pragma unroll 1
  for (int t = 0; t < ntaps; ++t) {
    int aoffn, wtn;
    tapf(t + 1 < ntaps ? t + 1 : t, aoffn, wtn);
    const bf16_t* wnext = w.wp + (size_t)wtn * w.ts;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = j & 1, nx = c ^ 1;
      const int no = (j + 1 == NJ) ? aoffn : aoff + (j + 1) * 16;
      const bf16_t* an0 = ap0 + no;
      const bf16_t* an1 = ap1 + no;
      const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, f.q[j][0]), b2 = __builtin_bit_cast(bf16x8_t, f.q[j][1]);
      const bf16x8_t p1 = __builtin_bit_cast(bf16x8_t, a[c][0][0]), q1 = __builtin_bit_cast(bf16x8_t, a[c][1][0]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NA == 3) {
        const bf16x8_t b3 = __builtin_bit_cast(bf16x8_t, f.q[j][2]);
        const bf16x8_t p2 = __builtin_bit_cast(bf16x8_t, a[c][0][1]), p3 = __builtin_bit_cast(bf16x8_t, a[c][0][2]);
        const bf16x8_t q2 = __builtin_bit_cast(bf16x8_t, a[c][1][1]), q3 = __builtin_bit_cast(bf16x8_t, a[c][1][2]);
        // products a1b3, a2b2, a1b2, a3b1, a2b1, a1b1 (small terms first), block 0 then block 1; one memory instruction behind
        // each of the first nine MFMAs: the next slab's six pixel fragments (LDS), then the next tap's weight fragments (L2)
        LR_MFMA(b3, p1, acc0, j & 1);
        a[nx][0][0] = *reinterpret_cast<const uint4*>(an0);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b3, q1, acc1, j & 1);
        a[nx][1][0] = *reinterpret_cast<const uint4*>(an1);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, p2, acc0, j & 1);
        a[nx][0][1] = *reinterpret_cast<const uint4*>(an0 + pl);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, q2, acc1, j & 1);
        a[nx][1][1] = *reinterpret_cast<const uint4*>(an1 + pl);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, p1, acc0, j & 1);
        a[nx][0][2] = *reinterpret_cast<const uint4*>(an0 + 2 * pl);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, q1, acc1, j & 1);
        a[nx][1][2] = *reinterpret_cast<const uint4*>(an1 + 2 * pl);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, p3, acc0, j & 1);
        f.q[j][2] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + 2 * w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, q3, acc1, j & 1);
        f.q[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, p2, acc0, j & 1);
        LR_MFMA(b1, q2, acc1, j & 1);
        LR_MFMA(b1, p1, acc0, j & 1);
        LR_MFMA(b1, q1, acc1, j & 1);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      } else if constexpr (NA == 2) {  // two fp16 terms each: h1 g2, h2 g1, h1 g1; block 0 then block 1
        acc0 = dm_mfma16<true>(f.q[j][1], a[c][0][0], acc0);
        a[nx][0][0] = *reinterpret_cast<const uint4*>(an0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = dm_mfma16<true>(f.q[j][1], a[c][1][0], acc1);
        a[nx][1][0] = *reinterpret_cast<const uint4*>(an1);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = dm_mfma16<true>(f.q[j][0], a[c][0][1], acc0);
        a[nx][0][1] = *reinterpret_cast<const uint4*>(an0 + pl);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = dm_mfma16<true>(f.q[j][0], a[c][1][1], acc1);
        a[nx][1][1] = *reinterpret_cast<const uint4*>(an1 + pl);
        f.q[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = dm_mfma16<true>(f.q[j][0], a[c][0][0], acc0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = dm_mfma16<true>(f.q[j][0], a[c][1][0], acc1);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      } else if constexpr (NA == 4) {  // one half term each; block 0 then block 1
        acc0 = dm_mfma16<true>(f.q[j][0], a[c][0][0], acc0);
        a[nx][0][0] = *reinterpret_cast<const uint4*>(an0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = dm_mfma16<true>(f.q[j][0], a[c][1][0], acc1);
        a[nx][1][0] = *reinterpret_cast<const uint4*>(an1);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      } else {  // bf16 activations x (hi, lo) weights
        LR_MFMA(b2, p1, acc0, j & 1);
        a[nx][0][0] = *reinterpret_cast<const uint4*>(an0);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b2, q1, acc1, j & 1);
        a[nx][1][0] = *reinterpret_cast<const uint4*>(an1);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, p1, acc0, j & 1);
        f.q[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + w.so + (size_t)j * JS) + lane16);
        __builtin_amdgcn_sched_barrier(0);
        LR_MFMA(b1, q1, acc1, j & 1);
        f.q[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + (size_t)j * JS) + lane16);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    aoff = aoffn; wt = wtn;
  }
}

// The two K halves of an 8x8 layer meet: wave (kh, nb) keeps the pixel half kh of its output block and adds the partner's
// (kh ^ 1, nb) partial sum of it, sent through X ([4 waves][4][64 lanes] float4 = 16 KiB).  One barrier - which also tells
// every wave that the K loops are over (the input planes may be rewritten).  Result in acc0.
__device__ __forceinline__ void kx_exchange(f32x16& acc0, const f32x16& acc1, int kh, int wave, int lane, float* X) {
  float4* X4 = reinterpret_cast<float4*>(X);
  if (kh) {  // (wave-uniform)
#pragma unroll
    for (int q = 0; q < 4; ++q) X4[(wave * 4 + q) * 64 + lane] = make_float4(acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r];
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) X4[(wave * 4 + q) * 64 + lane] = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
  }
  __syncthreads();
  const int pw = wave ^ 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = X4[(pw * 4 + q) * 64 + lane];
    acc0[4 * q] += v.x; acc0[4 * q + 1] += v.y; acc0[4 * q + 2] += v.z; acc0[4 * q + 3] += v.w;
  }
}

__device__ __forceinline__ void zero_acc(f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
}

// lane l of wave (., wn) holds channels wn*32 + 8k + 4(l>>5) + e in acc[4k + e]; bias: this layer's row of the LDS table
__device__ __forceinline__ void add_bias(f32x16& acc, const float* bias, int cb) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + cb + 8 * k);
    acc[4 * k] += bv.x; acc[4 * k + 1] += bv.y; acc[4 * k + 2] += bv.z; acc[4 * k + 3] += bv.w;
  }
}

// acc = acc * osc + bias (osc = 1: the plain sum, bit for bit)
__device__ __forceinline__ void scale_bias(f32x16& acc, float osc, const float* bias, int cb) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + cb + 8 * k);
    acc[4 * k] = fmaf(acc[4 * k], osc, bv.x); acc[4 * k + 1] = fmaf(acc[4 * k + 1], osc, bv.y);
    acc[4 * k + 2] = fmaf(acc[4 * k + 2], osc, bv.z); acc[4 * k + 3] = fmaf(acc[4 * k + 3], osc, bv.w);
  }
}

// largest magnitude of the lane's 16 values
__device__ __forceinline__ float lane_amax(const f32x16& acc) {
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(acc[r]));
  return m;
}
// Power-of-two scale of a tensor a consumer reads raw (NA = 2): the workgroup-wide maximum of `m` (0 from lanes / waves that
// hold nothing) through 4 LDS words and ONE barrier - every wave must call it -, then s = 2^k with max * s in [2^10, 2^11)
// (exponent clamped: zeros and denormal-sized tensors keep s <= 2^117).  Returns s; inv = 1 / s.
__device__ __forceinline__ float wg_scale(float m, float* red4, int wave, float& inv) {
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 8, 64));
  m = fmaxf(m, __shfl_xor(m, 4, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  if ((threadIdx.x & 63) == 0) red4[wave] = m;
  __syncthreads();
  const float mx = fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
  int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  eb = min(max(eb, 20), 240);
  inv = __uint_as_float((unsigned)(eb - 10) << 23);
  return __uint_as_float((unsigned)(264 - eb) << 23);
}

// sums of four values over the 64 lanes of a wave, every lane receives all four: one halving butterfly (v_permlane32_swap,
// v_permlane16_swap, DPP row sum: the lanes of row r then hold the total of value r) and four v_readlane, instead of four times
// two ds_bpermute + a row sum
__device__ __forceinline__ void wave_sum4(float (&v)[4]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[q]), __float_as_uint(v[q + 2]), false, false);
    v[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
  const float t = dm_row16_sum(__uint_as_float(r16[0]) + __uint_as_float(r16[1]));
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t), 16 * k));
}

// GroupNorm(8) + SiLU (+ time embedding) of a layer's output IN the accumulator registers.  Statistics: whole-image (mean, M2)
// per group, two passes over the registers; `valid`: this lane holds a pixel; NW: waves that share a group's pixels (2 at
// 8x8: the pixel halves meet through `red` and two barriers; 1 at 4x4: lanes only, no barrier).  Every wave of the
// workgroup must call the NW = 2 form.  inv_cnt = 1 / (pixels x 8 channels).  gamma, beta, temb: rows of the LDS tables
// (temb null: no time embedding).
template <int NW>
__device__ __forceinline__ void gn_silu_regs(f32x16& acc, bool valid, float inv_cnt, const float* gamma, const float* beta,
                                             const float* temb, int cb, float* red, int wave, int wn) {
  float mu[4], m2[4];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float val[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sm = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dlt = acc[4 * k + e] - (pass ? mu[k] : 0.f);
        sm += pass ? dlt * dlt : dlt;
      }
      val[k] = valid ? sm : 0.f;
    }
    wave_sum4(val);
    if (NW == 2) {
      float* rr = red + pass * 16;
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rr[wave * 4 + k] = val[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) val[k] = rr[wn * 4 + k] + rr[(2 + wn) * 4 + k];  // pixel halves wm = 0, 1 (wave = wm*2 + wn)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (pass == 0) mu[k] = val[k] * inv_cnt;
      else m2[k] = val[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 gm = *reinterpret_cast<const float4*>(gamma + cb + 8 * k);
    const float4 bt = *reinterpret_cast<const float4*>(beta + cb + 8 * k);
    const float4 te = temb ? *reinterpret_cast<const float4*>(temb + cb + 8 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float rstd = rsqrtf(m2[k] * inv_cnt + DM_GN_EPS);
    const float4 sc = make_float4(rstd * gm.x, rstd * gm.y, rstd * gm.z, rstd * gm.w);
    const float4 sh = make_float4(bt.x - mu[k] * sc.x, bt.y - mu[k] * sc.y, bt.z - mu[k] * sc.z, bt.w - mu[k] * sc.w);
    const float4 w = make_float4(fmaf(acc[4 * k], sc.x, sh.x), fmaf(acc[4 * k + 1], sc.y, sh.y), fmaf(acc[4 * k + 2], sc.z, sh.z),
                                 fmaf(acc[4 * k + 3], sc.w, sh.w));
    const float4 o = dm_silu_add4(w, te);
    acc[4 * k] = o.x; acc[4 * k + 1] = o.y; acc[4 * k + 2] = o.z; acc[4 * k + 3] = o.w;
  }
}

// the lane's 16 values -> three bf16 planes at pixel index pix of a plane buffer (plane stride pl)
template <int NA>
__device__ __forceinline__ void put_planes(const f32x16& acc, bf16_t* buf, int pl, int pix, int cb, float sc = 1.f) {
  bf16_t* q = buf + pix * RSB + cb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if constexpr (NA == 2) {  // two fp16 planes of the values times sc (a power of two)
      uint2 s1, s2;
      dm_split_h2(acc[4 * k] * sc, acc[4 * k + 1] * sc, s1.x, s2.x);
      dm_split_h2(acc[4 * k + 2] * sc, acc[4 * k + 3] * sc, s1.y, s2.y);
      *reinterpret_cast<uint2*>(q + 8 * k) = s1;
      *reinterpret_cast<uint2*>(q + 8 * k + pl) = s2;
      continue;
    }
    if constexpr (NA == 4) {  // one half plane of the values times sc, to nearest even
      uint2 s1;
      s1.x = bf_pack_h2(acc[4 * k] * sc, acc[4 * k + 1] * sc);
      s1.y = bf_pack_h2(acc[4 * k + 2] * sc, acc[4 * k + 3] * sc);
      *reinterpret_cast<uint2*>(q + 8 * k) = s1;
      continue;
    }
    if constexpr (NA == 1) {  // one plane: round to nearest even
      uint2 s1;
      s1.x = (unsigned)f2bf(acc[4 * k]) | ((unsigned)f2bf(acc[4 * k + 1]) << 16);
      s1.y = (unsigned)f2bf(acc[4 * k + 2]) | ((unsigned)f2bf(acc[4 * k + 3]) << 16);
      *reinterpret_cast<uint2*>(q + 8 * k) = s1;
      continue;
    }
    uint2 s1, s2, s3;
    dm_split3(acc[4 * k], acc[4 * k + 1], s1.x, s2.x, s3.x);
    dm_split3(acc[4 * k + 2], acc[4 * k + 3], s1.y, s2.y, s3.y);
    *reinterpret_cast<uint2*>(q + 8 * k) = s1;
    *reinterpret_cast<uint2*>(q + 8 * k + pl) = s2;
    *reinterpret_cast<uint2*>(q + 8 * k + 2 * pl) = s3;
  }
}

// eight fp32 values (two float4) -> three bf16 planes, 16 bytes each, at q
template <int NA>
__device__ __forceinline__ void put_planes8(float4 w0, float4 w1, bf16_t* q, int pl, float sc = 1.f) {
  if constexpr (NA == 2) {
    uint4 s1, s2;
    dm_split_h2(w0.x * sc, w0.y * sc, s1.x, s2.x);
    dm_split_h2(w0.z * sc, w0.w * sc, s1.y, s2.y);
    dm_split_h2(w1.x * sc, w1.y * sc, s1.z, s2.z);
    dm_split_h2(w1.z * sc, w1.w * sc, s1.w, s2.w);
    *reinterpret_cast<uint4*>(q) = s1;
    *reinterpret_cast<uint4*>(q + pl) = s2;
    return;
  }
  if constexpr (NA == 4) {
    uint4 s1;
    s1.x = bf_pack_h2(w0.x * sc, w0.y * sc); s1.y = bf_pack_h2(w0.z * sc, w0.w * sc);
    s1.z = bf_pack_h2(w1.x * sc, w1.y * sc); s1.w = bf_pack_h2(w1.z * sc, w1.w * sc);
    *reinterpret_cast<uint4*>(q) = s1;
    return;
  }
  if constexpr (NA == 1) {
    uint4 s1;
    s1.x = (unsigned)f2bf(w0.x) | ((unsigned)f2bf(w0.y) << 16); s1.y = (unsigned)f2bf(w0.z) | ((unsigned)f2bf(w0.w) << 16);
    s1.z = (unsigned)f2bf(w1.x) | ((unsigned)f2bf(w1.y) << 16); s1.w = (unsigned)f2bf(w1.z) | ((unsigned)f2bf(w1.w) << 16);
    *reinterpret_cast<uint4*>(q) = s1;
    return;
  }
  uint4 s1, s2, s3;
  dm_split3(w0.x, w0.y, s1.x, s2.x, s3.x);
  dm_split3(w0.z, w0.w, s1.y, s2.y, s3.y);
  dm_split3(w1.x, w1.y, s1.z, s2.z, s3.z);
  dm_split3(w1.z, w1.w, s1.w, s2.w, s3.w);
  *reinterpret_cast<uint4*>(q) = s1;
  *reinterpret_cast<uint4*>(q + pl) = s2;
  *reinterpret_cast<uint4*>(q + 2 * pl) = s3;
}

// the lane's 16 values as bf16 at q (+ 8k per run): the bf16 path's 16-bit storage of a tensor its consumer reads raw
__device__ __forceinline__ void put_bf16(const f32x16& acc, bf16_t* q) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint2 v;
    v.x = (unsigned)f2bf(acc[4 * k]) | ((unsigned)f2bf(acc[4 * k + 1]) << 16);
    v.y = (unsigned)f2bf(acc[4 * k + 2]) | ((unsigned)f2bf(acc[4 * k + 3]) << 16);
    *reinterpret_cast<uint2*>(q + 8 * k) = v;
  }
}

// the lane's 16 values as fp32 at q (+ 8k per run)
__device__ __forceinline__ void put_f32(const f32x16& acc, float* q) {
#pragma unroll
  for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(q + 8 * k) = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
}

// zero the given pixel rows (all three planes) of a plane buffer: idx -> pixel index through f
template <int NA, class F>
__device__ __forceinline__ void zero_pixels(bf16_t* buf, int pl, int npix, F pixf) {
  constexpr int NPL = lr_planes(NA);
  for (int i = threadIdx.x; i < npix * 9 * NPL; i += 256) {  // planes x 9 uint4 per pixel row
    const int pp = i / (9 * NPL), r = i - pp * (9 * NPL), s = r / 9, u = r - s * 9;
    *reinterpret_cast<uint4*>(buf + s * pl + pixf(pp) * RSB + u * 8) = make_uint4(0u, 0u, 0u, 0u);
  }
}
// halo ring of an N x N layout with an (N-2) x (N-2) interior
template <int NA, int N>
__device__ __forceinline__ void zero_ring(bf16_t* buf, int pl) {
  zero_pixels<NA>(buf, pl, 4 * (N - 1), [](int i) {
    const int side = i / (N - 1), k = i - side * (N - 1);
    return side == 0 ? k : side == 1 ? (N - 1) * N + 1 + k : side == 2 ? (1 + k) * N : k * N + N - 1;
  });
}
}  // namespace lr

template <int NA>
__global__ __launch_bounds__(256, 2) void lowres_fused_kernel(const LowresArgs p) {
  using namespace lr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bf16_t* P = reinterpret_cast<bf16_t*>(smem);
  bf16_t* D3p = P + 3 * PL;
  float* X8 = reinterpret_cast<float*>(D3p);  // 8x8 level: K-split exchange area (D3p is live during the 4x4 level only)
  float* sm = reinterpret_cast<float*>(reinterpret_cast<char*>(D3p) + X8_BYTES);
  bf16_t* E = P;                    // 4x4-level buffers live in P while the 8x8 level does not need it
  bf16_t* Fb = P + 3 * PL4;
  float* X = reinterpret_cast<float*>(reinterpret_cast<char*>(P) + X_OFF);
  float* s_scale = sm + SM_SCALE;   // [64] input stage only
  float* s_shift = sm + SM_SHIFT;   // [64]
  float* red = sm + SM_RED;         // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int wn_s = __builtin_amdgcn_readfirstlane(wn), wm_s = __builtin_amdgcn_readfirstlane(wm);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int cb = wn * 32 + 4 * h;   // first channel of the lane's run 0
  const int step = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;
  // 8x8 layers: this lane's pixel; 4x4 layers: pixel l31 & 15 (lanes 16-31 of each half duplicate it, results unused)
  const int m8 = wm * 32 + l31, y8 = m8 >> 3, x8 = m8 & 7;
  const int m4 = l31 & 15, y4 = m4 >> 2, x4 = m4 & 3;
  const bool v4 = l31 < 16;
  const size_t TS64 = 4 * JS, TS128 = 8 * JS;  // tap strides for 64 / 128 input channels
  const size_t wofs = (size_t)wn_s * 512;      // this wave's 32 output channels

  // ---- once per workgroup: every layer's bias and GroupNorm affine parameters -> LDS (wave w takes layers w, w+4, w+8)
  {
    const LrLayer* Ls[12] = {&p.d2t, &p.d3c1, &p.d3c2, &p.d3t, &p.b0, &p.b3, &p.u1c1, &p.u1c2, &p.u1t, &p.u2c1, &p.u2c2, &p.u2t};
    int ni = 0;
#pragma unroll
    for (int l = 0; l < 12; ++l) {
      if ((l & 3) == wave_s) sm[SM_BIAS + l * 64 + lane] = Ls[l]->bias[lane];
      if (Ls[l]->gamma) {  // (uniform; the normed layers are the same for every launch: d3c1, d3c2, b0, u1c1, u1c2, u2c1, u2c2)
        if ((l & 3) == wave_s) {
          sm[SM_GAMMA + ni * 64 + lane] = Ls[l]->gamma[lane];
          sm[SM_BETA + ni * 64 + lane] = Ls[l]->beta[lane];
        }
        ++ni;
      }
    }
  }
  if (tid < 12) {
    const LrLayer* Ls[12] = {&p.d2t, &p.d3c1, &p.d3c2, &p.d3t, &p.b0, &p.b3, &p.u1c1, &p.u1c2, &p.u1t, &p.u2c1, &p.u2c2, &p.u2t};
    float v = 1.f;
#pragma unroll
    for (int l = 0; l < 12; ++l) if (l == tid && lr_scaled(NA)) v = *Ls[l]->wsc;
    sm[SM_WSC + tid] = v;   // (1 where the pack carries no scale)
  }
  const float* bias_t = sm + SM_BIAS;
  const float* wsc = sm + SM_WSC;
  float* red4 = sm + SM_RED + 32;   // wg_scale's exchange words

  // weight streams (wave-uniform): {fragment base, tap stride, split stride}
  // (8x8 layers: the wave pair index wm is the K half kh - channel slabs 2kh, 2kh+1 -, see krun2)
  const size_t kofs = (size_t)wm_s * 2 * JS;
  const WS w_d2t = {p.d2t.w + wofs + kofs, TS64, 16 * TS64}, w_d3c1 = {p.d3c1.w + wofs + kofs, TS64, 9 * TS64}, w_d3c2 = {p.d3c2.w + wofs + kofs, TS64, 9 * TS64};
  const WS w_d3t = {p.d3t.w + wofs, TS64, 16 * TS64}, w_b0 = {p.b0.w + wofs, TS64, 9 * TS64}, w_b3 = {p.b3.w + wofs, TS64, 9 * TS64};
  const WS w_u1c1 = {p.u1c1.w + wofs + (size_t)wm_s * 4 * JS, TS128, 9 * TS128}, w_u1c2 = {p.u1c2.w + wofs, TS64, 9 * TS64};
  const WS w_u1t = {p.u1t.w + wofs, TS64, 16 * TS64}, w_u2c1a = {p.u2c1.w + wofs + kofs, TS128, 9 * TS128};
  const WS w_u2c1b = {p.u2c1.w + wofs + 4 * JS + kofs, TS128, 9 * TS128}, w_u2c2 = {p.u2c2.w + wofs + kofs, TS64, 9 * TS64}, w_u2t = {p.u2t.w + wofs, TS64, 16 * TS64};

  WFrag wf;
  // taps of the input phase (a, b) of a stride-2 conv: ky = 2ty + 1 - a, kx = 2tx + 1 - b
  auto wt_phase = [](int pa, int pb, int ty, int tx) { return (2 * ty + 1 - pa) * 4 + 2 * tx + 1 - pb; };
  if (blockIdx.x < p.B) kload<NA, 2>(wf, w_d2t, wt_phase(0, 0, 0, 0), lane16);
  // the lane's fragment offsets for the two 32-pixel blocks of an 8x8 layer (pixel b*32 + l31), K half kh = wm
  const int y8a = l31 >> 3, y8b = 4 + (l31 >> 3), x8l = l31 & 7;
  const int kh_el = wm * 32;  // first channel of the wave's K half

  for (int n = blockIdx.x; n < p.B; n += gridDim.x) {
    f32x16 acc, acc1;
    LR_STAMP(0);
    // time-embedding rows of this image's three normed conv1 layers (blocks down3, up1, up2) -> LDS (first read in phase B)
    if (wave_s < 3) {
      const float* hd = p.heads + (size_t)step * p.heads_step_stride + (size_t)n * p.heads_img_stride;
      sm[SM_TEMB + wave_s * 64 + lane] = hd[(2 + wave_s) * 64 + lane];
    }

    // ================= A: down2.transform, Conv2d 4x4 s2 p1, 16x16 -> 8x8 (model.py:46,64) =================
    // input = SiLU(GroupNorm(t2)) (down2.bnorm2; no time embedding after the second norm), staged one input phase (row parity a,
    // column parity b) at a time: plane pixel (Y, X), 0..8, = input pixel (2Y - a, 2X - b); taps ky = 2ty + 1 - a, kx = 2tx + 1 - b
    if (tid < 64) {
      const int c = tid;
      const float* st = p.st2 + ((size_t)n * p.pin2 * DM_GROUPS + (c >> 3)) * 2;
      float msum = 0.f, m2 = 0.f, dev = 0.f;
      for (int q = 0; q < p.pin2; ++q) msum += st[q * DM_GROUPS * 2];
      const float mean = msum / (float)p.pin2;
      for (int q = 0; q < p.pin2; ++q) {
        const float dq = st[q * DM_GROUPS * 2] - mean;
        m2 += st[q * DM_GROUPS * 2 + 1];
        dev += dq * dq;
      }
      const float var = (m2 + p.cnt2 * dev) / (p.cnt2 * (float)p.pin2);
      const float sc = rsqrtf(var + DM_GN_EPS) * p.g2[c];
      s_scale[c] = sc;
      s_shift[c] = p.b2[c] - mean * sc;
    }
    zero_acc(acc);
    zero_acc(acc1);
    const float* src = p.t2 + (size_t)n * 16 * 16 * 64;
    const bf16_t* srch = reinterpret_cast<const bf16_t*>(p.t2) + (size_t)n * 16 * 16 * 64;  // (NA = 1: the tensor is IEEE half)
    float4 v0[3], v1[3];
    unsigned inbm = 0;
    // the three (pixel, channel octet) items of this thread for phase ph: loads from clamped addresses, padding by a mask
    auto fetch = [&](int ph) {
      const int pa = ph >> 1, pb = ph & 1;
      inbm = 0;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int idx = tid + 256 * u;
        const int pix = min(idx >> 3, 80), c8 = idx & 7;
        const int Y = pix / 9, Xc = pix - Y * 9;
        const int iy = 2 * Y - pa, ix = 2 * Xc - pb;
        inbm |= ((unsigned)iy < 16u && (unsigned)ix < 16u) ? (1u << u) : 0u;
        const int eo = (min(max(iy, 0), 15) * 16 + min(max(ix, 0), 15)) * 64 + c8 * 8;
        if constexpr (lr_io16(NA)) {
          v0[u] = __builtin_bit_cast(float4, *reinterpret_cast<const uint4*>(srch + eo));  // eight halves, widened when staged
        } else {
          v0[u] = *reinterpret_cast<const float4*>(src + eo);
          v1[u] = *reinterpret_cast<const float4*>(src + eo + 4);
        }
      }
    };
    fetch(0);
    for (int ph = 0; ph < 4; ++ph) {
      const int pa = ph >> 1, pb = ph & 1;
      __syncthreads();  // tables ready / the previous phase's (previous image's) reads of P are done
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int idx = tid + 256 * u;
        if (idx < 81 * 8) {
          const int pix = idx >> 3, c8 = idx & 7;
          float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
          if constexpr (lr_io16(NA)) {
            const uint4 hv = __builtin_bit_cast(uint4, v0[u]);
            v0[u] = make_float4(h2f((bf16_t)(hv.x & 0xffff)), h2f((bf16_t)(hv.x >> 16)), h2f((bf16_t)(hv.y & 0xffff)), h2f((bf16_t)(hv.y >> 16)));
            v1[u] = make_float4(h2f((bf16_t)(hv.z & 0xffff)), h2f((bf16_t)(hv.z >> 16)), h2f((bf16_t)(hv.w & 0xffff)), h2f((bf16_t)(hv.w >> 16)));
          }
          if ((inbm >> u) & 1) {
            const float4 sc0 = *reinterpret_cast<const float4*>(s_scale + c8 * 8), sc1 = *reinterpret_cast<const float4*>(s_scale + c8 * 8 + 4);
            const float4 sh0 = *reinterpret_cast<const float4*>(s_shift + c8 * 8), sh1 = *reinterpret_cast<const float4*>(s_shift + c8 * 8 + 4);
            w0 = dm_silu_add4(make_float4(fmaf(v0[u].x, sc0.x, sh0.x), fmaf(v0[u].y, sc0.y, sh0.y), fmaf(v0[u].z, sc0.z, sh0.z), fmaf(v0[u].w, sc0.w, sh0.w)), make_float4(0.f, 0.f, 0.f, 0.f));
            w1 = dm_silu_add4(make_float4(fmaf(v1[u].x, sc1.x, sh1.x), fmaf(v1[u].y, sc1.y, sh1.y), fmaf(v1[u].z, sc1.z, sh1.z), fmaf(v1[u].w, sc1.w, sh1.w)), make_float4(0.f, 0.f, 0.f, 0.f));
          }
          put_planes8<NA>(w0, w1, P + pix * RSB + c8 * 8, PL);
        }
      }
      if (ph < 3) fetch(ph + 1);  // the next phase's pixels travel while this phase's MFMAs run
      __syncthreads();
      krun2<NA, 2>(acc, acc1, wf, P + (y8a * 9 + x8l) * RSB + 8 * h + kh_el, P + (y8b * 9 + x8l) * RSB + 8 * h + kh_el, PL, w_d2t, lane16, 4,
               [&](int t, int& ao, int& wt) { ao = ((t >> 1) * 9 + (t & 1)) * RSB; wt = wt_phase(pa, pb, t >> 1, t & 1); });
      if (ph < 3) kload<NA, 2>(wf, w_d2t, wt_phase((ph + 1) >> 1, (ph + 1) & 1, 0, 0), lane16);
    }
    kload<NA, 2>(wf, w_d3c1, 0, lane16);
    kx_exchange(acc, acc1, wm, wave, lane, X8);  // (its barrier: every wave is done reading the phase planes)
    scale_bias(acc, wsc[0], bias_t + 0 * 64, cb);
    put_f32(acc, p.d2 + ((size_t)n * 64 + m8) * 64 + cb);  // down2's output, raw: skip input of up2.conv1, read back by this workgroup
    zero_ring<NA, 10>(P, PL);
    float s_d2 = 1.f, i_d2 = 1.f;   // power-of-two scale of down2's output as planes (NA = 2), and its inverse
    if constexpr (lr_scaled(NA)) s_d2 = wg_scale(lane_amax(acc), red4, wave, i_d2);
    put_planes<NA>(acc, P, PL, (y8 + 1) * 10 + x8 + 1, cb, s_d2);  // ... and, as planes, the input of down3.conv1

    LR_STAMP(1);
    // ================= B, C: down3.conv1, down3.conv2, Conv2d 3x3 p1 at 8x8 (model.py:54-62) =================
    const bf16_t* a8a = P + (y8a * 10 + x8l) * RSB + 8 * h + kh_el;  // blocks 0 / 1 of the 10x10 layout, K half kh
    const bf16_t* a8b = P + (y8b * 10 + x8l) * RSB + 8 * h + kh_el;
    auto tap3_10 = [&](int t, int& ao, int& wt) { const int ty = t / 3; ao = (ty * 10 + t - 3 * ty) * RSB; wt = t; };
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc1);
    LR_STAMP(13);
    krun2<NA, 2>(acc, acc1, wf, a8a, a8b, PL, w_d3c1, lane16, 9, tap3_10);
    LR_STAMP(14);
    kload<NA, 2>(wf, w_d3c2, 0, lane16);
    kx_exchange(acc, acc1, wm, wave, lane, X8);
    scale_bias(acc, wsc[1] * i_d2, bias_t + 1 * 64, cb);
    gn_silu_regs<2>(acc, true, 1.0f / 512.0f, sm + SM_GAMMA + 0 * 64, sm + SM_BETA + 0 * 64, sm + SM_TEMB + 0 * 64, cb, red, wave, wn);
    LR_STAMP(15);
    // (the two barriers inside: every wave is past its K loop, P may be rewritten; the halo ring stays zero)
    put_planes<NA>(acc, P, PL, (y8 + 1) * 10 + x8 + 1, cb);
    LR_STAMP(2);
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc1);
    krun2<NA, 2>(acc, acc1, wf, a8a, a8b, PL, w_d3c2, lane16, 9, tap3_10);
    kload<NA>(wf, w_d3t, wt_phase(wm_s, 0, 0, 0), lane16);
    kx_exchange(acc, acc1, wm, wave, lane, X8);
    scale_bias(acc, wsc[2], bias_t + 2 * 64, cb);
    gn_silu_regs<2>(acc, true, 1.0f / 512.0f, sm + SM_GAMMA + 1 * 64, sm + SM_BETA + 1 * 64, nullptr, cb, red, wave, wn);
    // output in the four-phase layout of the stride-2 conv: input pixel (y, x) = (2Y - a, 2X - b) -> plane (a, b) pixel (Y, X) of 5x5
    {
      const int pa = y8 & 1, pb = x8 & 1;
      put_planes<NA>(acc, P, PL, (pa * 2 + pb) * 25 + ((y8 + pa) >> 1) * 5 + ((x8 + pb) >> 1), cb);
      // padding: row Y = 4 of the planes with a = 0 (input row 8), row Y = 0 with a = 1 (input row -1); columns alike
      zero_pixels<NA>(P, PL, 36, [](int i) {
        const int ph = i / 9, k = i - ph * 9, a = ph >> 1, b = ph & 1;
        const int Yp = a ? 0 : 4, Xp = b ? 0 : 4;
        return ph * 25 + (k < 5 ? Yp * 5 + k : ((k - 5) + (Yp == 0 ? 1 : 0)) * 5 + Xp);
      });
    }

    LR_STAMP(3);
    // ================= D: down3.transform, Conv2d 4x4 s2 p1, 8x8 -> 4x4: K split by input phase over the wave pairs =================
    __syncthreads();
    zero_acc(acc);
    krun<NA>(acc, wf, P + (y4 * 5 + x4) * RSB + 8 * h, PL, w_d3t, lane16, 8, [&](int t, int& ao, int& wt) {
      const int ph = 2 * wm_s + (t >> 2), ty = (t >> 1) & 1, tx = t & 1;
      ao = (ph * 25 + ty * 5 + tx) * RSB;
      wt = wt_phase(ph >> 1, ph & 1, ty, tx);
    });
    kload<NA>(wf, w_b0, 5 * wm_s, lane16);
    __syncthreads();  // P is dead: the 4x4-level buffers E, F and the exchange area take its place
    if (wm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[(wn * 16 + r) * 64 + lane] = acc[r];
    }
    zero_ring<NA, 6>(E, PL4);
    zero_ring<NA, 6>(Fb, PL4);
    zero_ring<NA, 6>(D3p, PL4);  // (the buffer served as the 8x8 level's exchange area)
    __syncthreads();
    const int pix6 = (y4 + 1) * 6 + x4 + 1;
    if (!wm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += X[(wn * 16 + r) * 64 + lane];
      scale_bias(acc, wsc[3], bias_t + 3 * 64, cb);
    }
    float s_d3 = 1.f, i_d3 = 1.f;   // scale of down3's output as planes
    if constexpr (lr_scaled(NA)) s_d3 = wg_scale((!wm && v4) ? lane_amax(acc) : 0.f, red4, wave, i_d3);
    if (!wm) {
      if (v4) put_planes<NA>(acc, D3p, PL4, pix6, cb, s_d3);  // down3's output, raw: input of bottleneck.0 and skip input of up1.conv1
      if (p.taps && v4) put_f32(acc, p.taps + (size_t)n * 6144 + m4 * 64 + cb);
    }

    LR_STAMP(4);
    // ================= E..H: bottleneck.0, bottleneck.3, up1.conv1, up1.conv2, Conv2d 3x3 p1 at 4x4 =================
    // 16 pixels: the pixel-half index wm splits K (taps 0..4 | 5..8; up1.conv1: one input per half)
    auto tap3_6 = [&](int t, int& ao, int& wt) { const int ty = t / 3; ao = (ty * 6 + t - 3 * ty) * RSB; wt = t; };
    // in1 != null: K = [in0, in1] channels, one input per wave pair.  After the K loop the next layer's first weight
    // fragments are requested (wsn, wtn0), then the partial sums meet and the wm = 0 pair finishes the layer.
    // isc0 / isc1: inverse input scales of in0 / in1 (NA = 2; 1 otherwise).  The output is GroupNorm'ed (ni >= 0: order 1, no
    // scale) or raw (then it gets its own power-of-two scale, returned through s_out / i_out).
    auto conv4 = [&](const bf16_t* in0, const bf16_t* in1, const WS& ws, int li, int ni, const float* temb, bf16_t* out, int tap_off,
                     const WS& wsn, int wtn0, float isc0, float isc1, float& s_out, float& i_out) {
      __syncthreads();  // the input planes are complete; the exchange area is free
      zero_acc(acc);
      const int a4 = (y4 * 6 + x4) * RSB + 8 * h;
      if (in1) krun<NA>(acc, wf, (wm_s ? in1 : in0) + a4, PL4, ws, lane16, 9, tap3_6);
      else krun<NA>(acc, wf, in0 + a4, PL4, ws, lane16, wm_s ? 4 : 5, [&](int t, int& ao, int& wt) { tap3_6(t + 5 * wm_s, ao, wt); });
      kload<NA>(wf, wsn, wtn0, lane16);
      // each K half back to the true scale (weights' and its input's powers of two; 1 x 1 outside NA = 2), then the halves meet
      const float osc = wsc[li] * ((in1 && wm_s) ? isc1 : isc0);
      if (wm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(wn * 16 + r) * 64 + lane] = acc[r] * osc;
      }
      __syncthreads();
      if (!wm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmaf(acc[r], osc, X[(wn * 16 + r) * 64 + lane]);
        scale_bias(acc, 1.f, bias_t + li * 64, cb);
        if (ni >= 0) gn_silu_regs<1>(acc, v4, 1.0f / 128.0f, sm + SM_GAMMA + ni * 64, sm + SM_BETA + ni * 64, temb, cb, red, wave, wn);
      }
      s_out = 1.f; i_out = 1.f;
      if constexpr (lr_scaled(NA)) if (ni < 0) s_out = wg_scale((!wm && v4) ? lane_amax(acc) : 0.f, red4, wave, i_out);
      if (!wm) {
        if (v4) put_planes<NA>(acc, out, PL4, pix6, cb, s_out);
        if (tap_off >= 0 && p.taps && v4) put_f32(acc, p.taps + (size_t)n * 6144 + tap_off + m4 * 64 + cb);
      }
    };
    float s_tmp, i_tmp, s_b3, i_b3;
    conv4(D3p, nullptr, w_b0, 4, 2, nullptr, E, -1, w_b3, 5 * wm_s, i_d3, 1.f, s_tmp, i_tmp);
    LR_STAMP(5);
    conv4(E, nullptr, w_b3, 5, -1, nullptr, Fb, 1024, w_u1c1, 0, 1.f, 1.f, s_b3, i_b3);
    LR_STAMP(6);
    conv4(Fb, D3p, w_u1c1, 6, 3, sm + SM_TEMB + 1 * 64, E, -1, w_u1c2, 5 * wm_s, i_b3, i_d3, s_tmp, i_tmp);
    LR_STAMP(7);
    conv4(E, nullptr, w_u1c2, 7, 4, nullptr, Fb, -1, w_u1t, 2 * wm_s * 4, 1.f, 1.f, s_tmp, i_tmp);
    LR_STAMP(8);

    // ================= I: up1.transform, ConvTranspose2d 4x4 s2 p1, 4x4 -> 8x8: output parities 2wm, 2wm+1 per wave pair =================
    // out[2y+py][2x+px] = sum over ty, tx in {0,1} of in[y-1+py+ty][x-1+px+tx] w[3-py-2ty][3-px-2tx] (packed per parity)
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc1);
    {
      const bf16_t* au = Fb + (y4 * 6 + x4) * RSB + 8 * h;
      const int par0 = 2 * wm_s;
      krun<NA>(acc, wf, au, PL4, w_u1t, lane16, 4, [&](int t, int& ao, int& wt) { ao = ((wm_s + (t >> 1)) * 6 + (t & 1)) * RSB; wt = par0 * 4 + t; });
      kload<NA>(wf, w_u1t, (par0 + 1) * 4, lane16);
      krun<NA>(acc1, wf, au, PL4, w_u1t, lane16, 4, [&](int t, int& ao, int& wt) { ao = ((wm_s + (t >> 1)) * 6 + 1 + (t & 1)) * RSB; wt = (par0 + 1) * 4 + t; });
    }
    kload<NA, 2>(wf, w_u2c1a, 0, lane16);
    scale_bias(acc, wsc[8], bias_t + 8 * 64, cb);
    scale_bias(acc1, wsc[8], bias_t + 8 * 64, cb);
    if (p.taps && v4) {
      put_f32(acc, p.taps + (size_t)n * 6144 + 2048 + ((2 * y4 + wm) * 8 + 2 * x4) * 64 + cb);
      put_f32(acc1, p.taps + (size_t)n * 6144 + 2048 + ((2 * y4 + wm) * 8 + 2 * x4 + 1) * 64 + cb);
    }
    float s_u1 = 1.f, i_u1 = 1.f;   // scale of up1's output as planes
    if constexpr (lr_scaled(NA)) s_u1 = wg_scale(v4 ? fmaxf(lane_amax(acc), lane_amax(acc1)) : 0.f, red4, wave, i_u1);
    else __syncthreads();  // every wave is done with F: P becomes the 10x10 layout again (NA = 2: wg_scale's barrier)
    zero_ring<NA, 10>(P, PL);
    if (v4) {  // up1's output, raw: first input of up2.conv1 (parity py = wm, px = 0 / 1)
      put_planes<NA>(acc, P, PL, (2 * y4 + wm + 1) * 10 + 2 * x4 + 1, cb, s_u1);
      put_planes<NA>(acc1, P, PL, (2 * y4 + wm + 1) * 10 + 2 * x4 + 2, cb, s_u1);
    }

    LR_STAMP(9);
    // ================= J, K: up2.conv1 over cat[up1, down2] (model.py:147), up2.conv2, Conv2d 3x3 p1 at 8x8 =================
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc1);
    float4 d2v[2][2];  // down2's output comes back from global memory while the first half of K runs
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u;  // 64 pixels x 8 channel octets
      const float* q = p.d2 + ((size_t)n * 64 + (idx >> 3)) * 64 + (idx & 7) * 8;
      d2v[u][0] = *reinterpret_cast<const float4*>(q);
      d2v[u][1] = *reinterpret_cast<const float4*>(q + 4);
    }
    krun2<NA, 2>(acc, acc1, wf, a8a, a8b, PL, w_u2c1a, lane16, 9, tap3_10);
    kload<NA, 2>(wf, w_u2c1b, 0, lane16);
    if constexpr (lr_scaled(NA)) {  // the sums over up1's channels move to the scale of down2's planes (powers of two: exact)
      const float rs = i_u1 * s_d2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] *= rs; acc1[r] *= rs; }
    }
    __syncthreads();  // the up1 planes are consumed
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u, pix = idx >> 3, c8 = idx & 7;
      put_planes8<NA>(d2v[u][0], d2v[u][1], P + (((pix >> 3) + 1) * 10 + (pix & 7) + 1) * RSB + c8 * 8, PL, s_d2);
    }
    __syncthreads();
    krun2<NA, 2>(acc, acc1, wf, a8a, a8b, PL, w_u2c1b, lane16, 9, tap3_10);
    kload<NA, 2>(wf, w_u2c2, 0, lane16);
    kx_exchange(acc, acc1, wm, wave, lane, X8);
    scale_bias(acc, wsc[9] * i_d2, bias_t + 9 * 64, cb);
    gn_silu_regs<2>(acc, true, 1.0f / 512.0f, sm + SM_GAMMA + 5 * 64, sm + SM_BETA + 5 * 64, sm + SM_TEMB + 2 * 64, cb, red, wave, wn);
    put_planes<NA>(acc, P, PL, (y8 + 1) * 10 + x8 + 1, cb);
    LR_STAMP(10);
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc1);
    krun2<NA, 2>(acc, acc1, wf, a8a, a8b, PL, w_u2c2, lane16, 9, tap3_10);
    kload<NA>(wf, w_u2t, 2 * wm_s * 4, lane16);
    kx_exchange(acc, acc1, wm, wave, lane, X8);
    scale_bias(acc, wsc[10], bias_t + 10 * 64, cb);
    gn_silu_regs<2>(acc, true, 1.0f / 512.0f, sm + SM_GAMMA + 6 * 64, sm + SM_BETA + 6 * 64, nullptr, cb, red, wave, wn);
    put_planes<NA>(acc, P, PL, (y8 + 1) * 10 + x8 + 1, cb);

    LR_STAMP(11);
    // ================= L: up2.transform, ConvTranspose2d 4x4 s2 p1, 8x8 -> 16x16: output parities 2wm, 2wm+1 per wave pair =================
    // (all 64 pixels per wave, every weight fragment fetched by ONE wave of the workgroup)
    __syncthreads();
    {
      const bf16_t* u8a = P + (y8a * 10 + x8l) * RSB + 8 * h;
      const bf16_t* u8b = P + (y8b * 10 + x8l) * RSB + 8 * h;
      float amx = 0.f;
      for (int pp = 0; pp < 2; ++pp) {
        const int par = 2 * wm_s + pp, py = wm_s, px = pp;
        zero_acc(acc);
        zero_acc(acc1);
        krun2<NA, 4>(acc, acc1, wf, u8a, u8b, PL, w_u2t, lane16, 4,
                 [&](int t, int& ao, int& wt) { ao = ((py + (t >> 1)) * 10 + px + (t & 1)) * RSB; wt = par * 4 + t; });
        if (pp == 0) kload<NA>(wf, w_u2t, (par + 1) * 4, lane16);
        else if (n + (int)gridDim.x < p.B) kload<NA, 2>(wf, w_d2t, wt_phase(0, 0, 0, 0), lane16);  // the next image's first fragments
        scale_bias(acc, wsc[11], bias_t + 11 * 64, cb);
        scale_bias(acc1, wsc[11], bias_t + 11 * 64, cb);
        if (p.amax_u2) amx = fmaxf(amx, fmaxf(lane_amax(acc), lane_amax(acc1)));
        const size_t oa = (((size_t)n * 16 + 2 * y8a + py) * 16 + 2 * x8l + px) * 64 + cb;
        const size_t ob = (((size_t)n * 16 + 2 * y8b + py) * 16 + 2 * x8l + px) * 64 + cb;
        if constexpr (lr_io16(NA)) {
          put_bf16(acc, reinterpret_cast<bf16_t*>(p.u2) + oa);
          put_bf16(acc1, reinterpret_cast<bf16_t*>(p.u2) + ob);
        } else {
          put_f32(acc, p.u2 + oa);
          put_f32(acc1, p.u2 + ob);
        }
      }
      if (p.amax_u2) {
        amx = fmaxf(amx, __shfl_xor(amx, 32, 64));
        amx = fmaxf(amx, __shfl_xor(amx, 16, 64));
        amx = fmaxf(amx, __shfl_xor(amx, 8, 64));
        amx = fmaxf(amx, __shfl_xor(amx, 4, 64));
        amx = fmaxf(amx, __shfl_xor(amx, 2, 64));
        amx = fmaxf(amx, __shfl_xor(amx, 1, 64));
        if (lane == 0) p.amax_u2[(size_t)n * 4 + wave] = amx;
      }
    }
    LR_STAMP(12);
  }
}
