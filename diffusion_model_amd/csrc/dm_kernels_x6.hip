// dm_kernels_x6.hip — the fp32 convolutions computed on the bf16 matrix pipe with fp32-equivalent accuracy.
//
// gfx950's exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate, and a wave that mixes MFMAs with
// anything else time-shares its SIMD with the other waves (tools/diag/mfma_coissue.hip), so the fp32 kernels are bound
// by MFMA time + vector time.  Here every fp32 operand is split into THREE bf16 terms, x = x1 + x2 + x3 with
// x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (24 mantissa bits: the split is exact to 2^-27 |x|), and a
// product is taken as  a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1  — six v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; the dropped terms are <= 2^-26 |a b|, below the rounding of an fp32 product.  Six bf16 MFMAs cost
// 6/16 of the fp32 MFMA time for the same K: less than the Winograd form (4/9) and with none of its transforms.
// tools/diag/bf16x6_sim.py: the result is as close to the fp64 dot product as fp32-MFMA accumulation is.
//
// Structure as conv_bf16_kernel: persistent workgroups, XCD-aware tile order, GroupNorm+SiLU+temb applied while the
// fp32 NHWC input tile is staged — then split and written to LDS as three bf16 planes [plane][halo pixel][KC + 8] —,
// weights pre-split and packed in B-fragment order [split][parity][tap][Cin/16][Cout/32][64 lanes][8 bf16], fp32
// accumulators, the fp32 epilogue of conv_mfma_kernel (bias, GroupNorm partials, coalesced NHWC stores).
//   KIND 0: Conv2d 3x3 s1 p1.
//   KIND 1: Conv2d 4x4 s2 p1 as four stride-1 2x2 convolutions over the four (row, column) phases of the input
//           (space-to-depth): one K chunk per phase, so a tile holds (th+1) x (tw+1) pixels of ONE phase at a time
//           and every global read is a whole 256-byte pixel (the 16-channel chunks of conv_mfma_kernel<2,1,16>
//           fetched each 128-byte line twice).
//   KIND 4: ConvTranspose2d 4x4 s2 p1, all four output parities from one staged tile (64 input channels: one K chunk).
//   KIND 2: ConvTranspose2d 4x4 s2 p1, one output parity per workgroup (blockIdx.z; any number of K chunks).
#pragma once
#include "dm_kernels_bf16.hip"

typedef float dm_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 dm_bf2 __attribute__((ext_vector_type(2)));

// (a, b) -> three packed bf16 pairs; a = lo half.  v_cvt_pk_bf16_f32 rounds to nearest even; the residuals are exact.
__device__ __forceinline__ void dm_split3(float a, float b, unsigned& s1, unsigned& s2, unsigned& s3) {
  dm_f2 v = {a, b};
  s1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, dm_bf2));
  v -= dm_f2{__uint_as_float(s1 << 16), __uint_as_float(s1 & 0xffff0000u)};
  s2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, dm_bf2));
  v -= dm_f2{__uint_as_float(s2 << 16), __uint_as_float(s2 & 0xffff0000u)};
  s3 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, dm_bf2));
}

// Two-way fp16 split (the "f16x3" form, for inputs that went through GroupNorm): x = h1 + h2 with h1 = fp16(x) rounded TOWARD ZERO
// (v_cvt_pkrtz_f16_f32: saturates instead of overflowing), h2 = fp16(x - h1) to nearest; the residual is exact, fp16 subnormals
// are honoured by the conversions and by the MFMA (tools/diag/mfma_f16_denorm.hip), so |x - h1 - h2| <= max(2^-21 |x|, 3e-8).
// Products h1 g1 + h1 g2 + h2 g1 (the dropped h2 g2 is <= 2^-20 of the product): three v_mfma_f32_32x32x16_f16 per fp32
// product.  tools/diag/f16x3_sim.py: on layer-like data (activations of order 1) the result is as close to the fp64 dot
// product as fp32-MFMA accumulation and the three-way bf16 split are; it loses relative accuracy only when ALL activations are
// tiny (no normalisation in front) - which is why it serves the layers behind a GroupNorm only.
typedef _Float16 dm_h2v __attribute__((ext_vector_type(2)));
typedef _Float16 dm_h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void dm_split_h2(float a, float b, unsigned& s1, unsigned& s2) {
  dm_f2 v = {a, b};
  const dm_h2v h = __builtin_bit_cast(dm_h2v, __builtin_amdgcn_cvt_pkrtz(a, b));
  s1 = __builtin_bit_cast(unsigned, h);
  v -= __builtin_convertvector(h, dm_f2);
  s2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, dm_h2v));
}
// Power-of-two scale of an image from its producers' maxima (every lane of the wave gets the same value): s = 2^k with
// max|x| s in [2^10, 2^11), exponent clamped (an all-zero image keeps a finite scale); inv = 1 / s.
__device__ __forceinline__ float dm_amax_scale(const float* a1, int n1, const float* a2, int n2, int lane, float& inv) {
  float mx = 0.f;
  for (int i = lane; i < n1; i += 64) mx = fmaxf(mx, a1[i]);
  for (int i = lane; i < n2; i += 64) mx = fmaxf(mx, a2[i]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  eb = min(max(eb, 20), 240);
  inv = __uint_as_float((unsigned)(eb - 10) << 23);
  return __uint_as_float((unsigned)(264 - eb) << 23);
}

// one 32x32x16 MFMA on eight 16-bit elements per lane: bf16 or fp16 operands
template <bool F16>
__device__ __forceinline__ f32x16 dm_mfma16(const uint4& a, const uint4& b, const f32x16& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dm_h8v, a), __builtin_bit_cast(dm_h8v, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Epilogue for the SWAPPED accumulator layout (weights as the MFMA's A operand, pixels as B): lane l holds pixel
// (l & 31) of its 32-pixel block and the 16 channels  wn*32 + 8k + 4(l >> 5) + e  (k = r >> 2, e = r & 3), i.e. four
// runs of four consecutive channels -> + bias, GroupNorm (mean, M2) partials, and 16-byte NHWC stores straight from the
// registers: no LDS staging, no per-element address math.  One image per tile (ni == 1).  scr: 64 floats of LDS.
template <int MB, int KIND>
__device__ __forceinline__ void x6_epilogue_sw(const ConvArgs& p, f32x16 (&acc)[MB], float* scr, int par, int py, int px,
                                               int img0, int tin, int y0, int x0, int wm, int wn, int lane, int h, int tid, int wave,
                                               float oscale = 1.f) {  // (fp16 form: the weights were packed times a power of two)
  constexpr bool UP = (KIND == 4 || KIND == 2);
  constexpr int MT = 64 * MB;
  const int OW = (KIND == 1) ? (p.W >> 1) : UP ? (p.W << 1) : p.W;
  const int OH = (KIND == 1) ? (p.H >> 1) : UP ? (p.H << 1) : p.H;
  const int cb = blockIdx.y * 64 + wn * 32 + 4 * h;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + cb + 8 * k);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      acc[mb][4 * k] = fmaf(acc[mb][4 * k], oscale, bv.x); acc[mb][4 * k + 1] = fmaf(acc[mb][4 * k + 1], oscale, bv.y);
      acc[mb][4 * k + 2] = fmaf(acc[mb][4 * k + 2], oscale, bv.z); acc[mb][4 * k + 3] = fmaf(acc[mb][4 * k + 3], oscale, bv.w);  // (oscale = 1: the plain sum)
    }
  }
  if (p.amax_out) {  // largest |y| of this wave's part of the tile: the consumer's power-of-two scale (two-fp16-term form on raw inputs)
    float mx = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[mb][r]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    const int pidx = (UP ? par * p.tiles_per_img : 0) + tin;
    if (lane == 0) p.amax_out[(((size_t)img0 * p.pout + pidx) * gridDim.y + blockIdx.y) * 4 + wave] = mx;
  }
  if (p.stats_out) {
    // (mean, M2) of the tile's pixels per group, two passes.  A wave holds four runs of eight channels (run k: channels
    // wn*32 + 8k .. + 7 over the lane halves); its four per-lane sums are reduced over the 64 lanes in one halving butterfly
    // (v_permlane32_swap, v_permlane16_swap, DPP row sum: no LDS crossbar), after which the lanes of row r hold the total of
    // run r; the waves' totals meet in LDS and thread g adds the runs of group g (G = 8 .. 64 channels: 1 .. 8 runs).
    const int gpt = 64 >> p.g_log;            // groups in the workgroup's 64 channels
    const int rpg_log = p.g_log - 3;          // log2(runs per group)
    float* sredA = scr;        // [4 waves][4 runs]
    float* sredB = scr + 16;
    float* s_mu = scr + 32;    // [<= 8]
    const float inv_cnt = 1.0f / (float)(MT << p.g_log);
    for (int pass = 0; pass < 2; ++pass) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mu = pass ? s_mu[(wn * 4 + k) >> rpg_log] : 0.f;
        float sm = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = acc[mb][4 * k + e] - mu;
            sm += pass ? d * d : d;
          }
        v[k] = sm;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {   // lanes l, l ^ 32: the low half keeps runs 0, 1, the high half runs 2, 3
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[q]), __float_as_uint(v[q + 2]), false, false);
        v[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      }
      const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);   // rows: even keep the first, odd the second
      const float t = dm_row16_sum(__uint_as_float(r16[0]) + __uint_as_float(r16[1]));
      float* sred = pass ? sredB : sredA;
      if ((lane & 15) == 0) sred[wave * 4 + (lane >> 4)] = t;
      __syncthreads();
      if (tid < gpt) {
        float sm = 0.f;
        for (int r = tid << rpg_log; r < (tid + 1) << rpg_log; ++r) sm += sred[(r >> 2) * 4 + (r & 3)] + sred[(2 + (r >> 2)) * 4 + (r & 3)];   // waves wm = 0, 1
        if (pass == 0) s_mu[tid] = sm * inv_cnt;
        else {
          const int pidx = (UP ? par * p.tiles_per_img : 0) + tin;
          float* so = p.stats_out + (((size_t)img0 * p.pout + pidx) * DM_GROUPS + blockIdx.y * gpt + tid) * 2;
          so[0] = s_mu[tid];
          so[1] = sm;
        }
      }
      if (pass == 0) __syncthreads();
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    const int x = x0 + (m & ((1 << p.tw_log) - 1)), y = y0 + (m >> p.tw_log);
    const int oy = UP ? 2 * y + py : y, ox = UP ? 2 * x + px : x;
    float* q = p.y + ((((size_t)img0 * OH + oy) * OW + ox) << p.cout_log) + cb;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<float4*>(q + 8 * k) = make_float4(acc[mb][4 * k], acc[mb][4 * k + 1], acc[mb][4 * k + 2], acc[mb][4 * k + 3]);
  }
}

// NS = 3: the fp32-equivalent form above.  NS = 2 (opt-in, DM_X6 bit 4): two bf16 terms per operand and THREE MFMAs per product
// (a1 b1 + a1 b2 + a2 b1): operands carry 16-17 mantissa bits, about 7x the rounding error of an fp32 dot product.
// F16 (with NS = 2, SW): the two-way fp16 split above; the pack holds two fp16 planes and, behind them, the float 2^-k that undoes
// the power-of-two scale of the packed weights.
template <int MB, int KIND, bool SW = false, int NS = 3, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_x6_kernel(const ConvArgs p) {
  static_assert(!F16 || (NS == 2 && SW), "the fp16 form has two planes and the swapped epilogue");
  constexpr int KC = 64;
  constexpr int MT = 64 * MB;
  constexpr int RSB = KC + 8;   // LDS row stride in bf16 (144 B: 16-B skew between pixels)
  constexpr int SS = (KIND == 1) ? 2 : 1;      // input pixels per halo step
  constexpr int NTY = (KIND == 0) ? 3 : 2;     // taps per axis walked per chunk (KIND 1: 2x2 per phase)
  constexpr int NT = NTY * NTY;
  constexpr int NTW = (KIND == 0) ? 9 : (KIND == 1) ? 16 : 4;  // taps per parity in the packed weights
  constexpr bool UPK = (KIND == 4 || KIND == 2);
  constexpr int NTOT = (KIND == 4) ? 16 : NT;
  constexpr int NJ = KC / 16;
  // staging items (8 channels = two 16-byte loads) per thread and batch: one batch covers the model's tiles without an
  // idle slot (KIND 1: 17x9 / 9x9 phase planes = 1224 / 648 items; 3x3 and transposed: 10x18 = 1440, 10x10 = 800, 4 x 6x6 = 1152)
  constexpr int U = (KIND == 1) ? (MB == 2 ? 5 : 3) : (MB == 2 ? 6 : 5);
  extern __shared__ __attribute__((aligned(16))) float smem[];

  __builtin_amdgcn_s_setprio(DM_PRIO_STAGE);
#ifdef DM_TIMELINE
  int tl_n = 0;
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 128] = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int Cin = p.C1 + p.C2;
  int par = (KIND == 2) ? (int)blockIdx.z : 0, py = par >> 1, px = par & 1;
  const int th = 1 << p.th_log, tw = 1 << p.tw_log;
  const int hth = p.hth, htw = p.htw;
  const int NP = p.ni * hth * htw;
  const int items = NP * 8;
  const int step_now = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;
  const float bias_lane = SW ? 0.f : p.bias[blockIdx.y * 64 + wn * 32 + (lane & 31)];

  const bool tf = p.stats_in != nullptr;
  float* s_scale = smem;
  float* s_shift = s_scale + (tf ? p.ni * p.C1 : 0);
  float* s_temb = s_shift + (tf ? p.ni * p.C1 : 0);
  bf16_t* tile = reinterpret_cast<bf16_t*>(s_temb + (tf ? p.ni * p.C1 : 0));
  const int PL = NP * RSB;      // plane stride (bf16 elements; a multiple of 8)

  int abase[MB];  // bf16 offsets of this lane's A fragments at tap (0,0)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    const int x = m & (tw - 1);
    const int y = (m >> p.tw_log) & (th - 1);
    const int i = m >> (p.tw_log + p.th_log);
    abase[mb] = ((i * hth + y) * htw + x) * RSB + 8 * h;
  }

  f32x16 acc[MB];
  const int nb32 = p.Cout >> 5;
  const size_t jstride = (size_t)nb32 * 512;               // bf16 elements per (16 channels, all output channels)
  const size_t tapstride = (size_t)(Cin >> 4) * jstride;
  const size_t split_off = (size_t)(UPK ? 4 * NTW : NTW) * tapstride;  // one split term's whole block
  const int cchunks = Cin / KC;
  const int nchunks = (KIND == 1) ? 4 * cchunks : cchunks;  // KIND 1: (phase, channel chunk)
  const bf16_t* wpk = reinterpret_cast<const bf16_t*>(p.wp);
  const float oscale = F16 ? p.wp[NS * split_off / 2] : 1.f;  // the float behind the two fp16 planes

  // first-batch halo coordinates do not depend on the tile: (image << 16 | hy << 8 | hx), once per workgroup
  int hyx0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int pix = (u * 256 + tid) >> 3;
    const int r = __umulhi((unsigned)pix, p.mg_htw);
    const int i = __umulhi((unsigned)r, p.mg_hth);
    hyx0[u] = (i << 16) | ((r - i * hth) << 8) | (pix - r * htw);
  }

  const int t_first = p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = p.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_xcd = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int t_end = p.xcd_map ? (((p.ngroups - t_xcd + 7) >> 3) << p.tpi_log) : p.ntiles;
  for (int ts = t_first; ts < t_end; ts += t_step) {
    const int t = p.xcd_map ? (((((ts >> p.tpi_log) << 3) + t_xcd) << p.tpi_log) | (ts & (p.tiles_per_img - 1))) : ts;
    if (KIND == 4) { par = 0; py = 0; px = 0; }  // (advanced by the previous tile's K loop)
    const int img0 = (t >> p.tpi_log) * p.ni;
    const int tin = t & (p.tiles_per_img - 1);
    const int y0 = (tin >> p.tx_log) << p.th_log;
    const int x0 = (tin & (p.tiles_x - 1)) << p.tw_log;
    DM_TL(1);
    float ascale = 1.f, ainv = 1.f;  // F16 on raw inputs: the image's power-of-two scale
    if constexpr (F16) if (!tf) ascale = dm_amax_scale(p.amax1 + (size_t)img0 * p.namax1, p.namax1, p.amax2 ? p.amax2 + (size_t)img0 * p.namax2 : nullptr, p.namax2, lane, ainv);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const int cch = (KIND == 1) ? chunk % cchunks : chunk;   // channel chunk
      const int ph = (KIND == 1) ? chunk / cchunks : 0;        // KIND 1: input phase (row parity a, column parity b)
      const int pa = ph >> 1, pb = ph & 1;
      // KIND 1: plane pixel (Y, X) of phase (a, b) is input pixel (2 y0 - a + 2 Y, 2 x0 - b + 2 X); its taps are
      // ky = 2 ty + 1 - a, kx = 2 tx + 1 - b  (output (y, x) reads input rows 2y - 1 + ky)
      const int hy0 = (KIND == 1) ? 2 * y0 - pa : (KIND == 2) ? y0 - 1 + py : y0 - 1;
      const int hx0 = (KIND == 1) ? 2 * x0 - pb : (KIND == 2) ? x0 - 1 + px : x0 - 1;
      const int cg0 = cch * KC;
      const bool from2 = cg0 >= p.C1;
      const float* src = from2 ? p.x2 : p.x1;
      const int cs_log = from2 ? p.c2_log : p.c1_log;
      const int coff = from2 ? cg0 - p.C1 : cg0;
      // wave-uniform weight pointer (scalar registers) + one 32-bit lane byte offset: saddr-form loads, no 64-bit vector adds
      const int wn_s = __builtin_amdgcn_readfirstlane(wn);
      const bf16_t* wbase = wpk + (size_t)((KIND == 2) ? par * NT : 0) * tapstride + (size_t)cch * NJ * jstride + (size_t)(blockIdx.y * 2 + wn_s) * 512;
      const unsigned lane16 = (unsigned)lane * 16u;
      if (chunk > 0) {
        DM_TL(5);
        __builtin_amdgcn_s_setprio(DM_PRIO_STAGE);
        __syncthreads();  // previous chunk's LDS reads are done
      }
      // ---- stage: global fp32 -> GroupNorm + SiLU + temb -> three bf16 planes in LDS
      {
        const int imax = p.B - 1 - img0;
        const float* srcb = src + ((size_t)img0 << (p.h_log + p.w_log + cs_log)) + coff;
        for (int base = 0; base < items; base += 256 * U) {  // uniform trip count (barrier inside)
          int tq = threadIdx.x;  // opaque copy: index math is redone here, not hoisted out of the tile loop and kept live
          asm volatile("" : "+v"(tq));
          float4 v0[U], v1[U];
          unsigned inbm = 0;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256 + tq;
            const int c8 = idx & 7;
            int hx, hy, i;
            if (base == 0) {
              hx = hyx0[u] & 255; hy = (hyx0[u] >> 8) & 255; i = hyx0[u] >> 16;
            } else {
              const int pix = idx >> 3;
              const int r = __umulhi((unsigned)pix, p.mg_htw);
              hx = pix - r * htw;
              i = __umulhi((unsigned)r, p.mg_hth);
              hy = r - i * hth;
            }
            const int iy = hy0 + SS * hy, ix = hx0 + SS * hx;
            const bool inb = idx < items && i <= imax && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            // unconditional loads from a clamped (always valid) address; padding is a select afterwards
            const int cn = min(i, imax), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
            const unsigned off = ((unsigned)(((cn << p.h_log) + cy) << p.w_log) + cx) << cs_log;
            const float* q = srcb + off + c8 * 8;
            v0[u] = *reinterpret_cast<const float4*>(q);
            v1[u] = *reinterpret_cast<const float4*>(q + 4);
            inbm |= inb ? (1u << u) : 0u;
          }
          asm volatile("" : "+v"(inbm));
          if (tf && chunk == 0 && base == 0) {
            int tg = threadIdx.x;
            asm volatile("" : "+v"(tg));
            gn_tables(p, step_now, img0, tg, s_scale, s_shift, s_temb);  // while the tile loads are in flight
            __syncthreads();
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256 + tq;
            if (idx < items) {
              const int pix = idx >> 3, c8 = idx & 7;
              float e[8] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w, v1[u].x, v1[u].y, v1[u].z, v1[u].w};
              if (!((inbm >> u) & 1)) {
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = 0.f;  // zero padding / images past the batch stay exactly 0
              } else if (tf) {
                const int i = (p.ni == 1) ? 0 : (base == 0) ? (hyx0[u] >> 16) : (int)__umulhi(__umulhi((unsigned)pix, p.mg_htw), p.mg_hth);
                const int tofs = (i << p.c1_log) + coff + c8 * 8;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                  const float4 sc = *reinterpret_cast<const float4*>(s_scale + tofs + 4 * q);
                  const float4 sh = *reinterpret_cast<const float4*>(s_shift + tofs + 4 * q);
                  const float4 te = *reinterpret_cast<const float4*>(s_temb + tofs + 4 * q);
                  float4 w = make_float4(fmaf(e[4 * q], sc.x, sh.x), fmaf(e[4 * q + 1], sc.y, sh.y),
                                         fmaf(e[4 * q + 2], sc.z, sh.z), fmaf(e[4 * q + 3], sc.w, sh.w));
                  if (p.tf_silu) w = dm_silu_add4(w, te);
                  else { w.x += te.x; w.y += te.y; w.z += te.z; w.w += te.w; }
                  e[4 * q] = w.x; e[4 * q + 1] = w.y; e[4 * q + 2] = w.z; e[4 * q + 3] = w.w;
                }
              }
              uint4 s1, s2, s3;
              if constexpr (F16) {
                if (!tf) {
#pragma unroll
                  for (int k = 0; k < 8; ++k) e[k] *= ascale;
                }
                dm_split_h2(e[0], e[1], s1.x, s2.x);
                dm_split_h2(e[2], e[3], s1.y, s2.y);
                dm_split_h2(e[4], e[5], s1.z, s2.z);
                dm_split_h2(e[6], e[7], s1.w, s2.w);
              } else {
                dm_split3(e[0], e[1], s1.x, s2.x, s3.x);
                dm_split3(e[2], e[3], s1.y, s2.y, s3.y);
                dm_split3(e[4], e[5], s1.z, s2.z, s3.z);
                dm_split3(e[6], e[7], s1.w, s2.w, s3.w);
              }
              bf16_t* q = tile + pix * RSB + c8 * 8;
              *reinterpret_cast<uint4*>(q) = s1;
              *reinterpret_cast<uint4*>(q + PL) = s2;
              if (NS == 3) *reinterpret_cast<uint4*>(q + 2 * PL) = s3;
            }
            __builtin_amdgcn_sched_barrier(0);  // one item's chain at a time (register peak)
          }
        }
      }
      // weight tap of (this chunk's) tap index tp: KIND 1 maps the phase's 2x2 taps into the 4x4 kernel
      auto wtap = [&](int tp) {
        const int wt = (KIND == 1) ? ((2 * (tp >> 1) + 1 - pa) * 4 + 2 * (tp & 1) + 1 - pb) : tp;
        return wbase + (size_t)wt * tapstride;
      };
      uint4 bq[NJ][NS];
      {
        const bf16_t* w0 = wtap(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int s = 0; s < NS; ++s) bq[j][s] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(w0 + s * split_off + j * jstride) + lane16);
      }
      __syncthreads();
      __builtin_amdgcn_s_setprio(DM_PRIO_K);
      DM_TL(2);
      if (chunk == 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
      }
#ifdef DM_X6_FILLER
      float fill0 = 0.5f, fill1 = 0.25f;
#endif
      uint4 a0[NS][MB], a1[NS][MB];
      const bf16_t* ap[NS][MB];  // this lane's fragment address per plane at tap (0,0): per step only (tap offset) + immediate
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          ap[s][mb] = tile + s * PL + abase[mb];
          a0[s][mb] = *reinterpret_cast<const uint4*>(ap[s][mb]);
        }
      int ty = 0, tx = 0;
#pragma unroll 1
      for (int tap = 0; tap < NTOT; ++tap) {
        // KIND 4: tap = parity*4 + (ty*2+tx); parity (py,px) reads halo rows py+ty, columns px+tx
        const int tapoff = (KIND == 4) ? (((tap >> 3) + ((tap >> 1) & 1)) * htw + ((tap >> 2) & 1) + (tap & 1)) * RSB
                                       : (ty * htw + tx) * RSB;
        int ntx = tx + 1, nty = ty;
        if (ntx == NTY) { ntx = 0; ++nty; }
        const bool last = tap + 1 == NTOT;
        const int nt = tap + 1;
        const int ntapoff = last ? tapoff
                          : (KIND == 4) ? (((nt >> 3) + ((nt >> 1) & 1)) * htw + ((nt >> 2) & 1) + (nt & 1)) * RSB
                                        : (nty * htw + ntx) * RSB;
        const bf16_t* wnext = wtap(last ? tap : tap + 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          uint4 (*acur)[MB] = (j & 1) ? a1 : a0;
          uint4 (*anxt)[MB] = (j & 1) ? a0 : a1;
          const int noff = (j + 1 < NJ) ? tapoff + (j + 1) * 16 : ntapoff;
          // One memory instruction behind each MFMA: a batch of nine (six LDS reads, three weight loads) takes ~170
          // cycles to issue, during which this wave feeds the matrix pipe nothing; one at a time they issue while
          // the MFMA in front of them executes (32 cycles).  Products in the order a1b3, a2b2, a1b2, a3b1, a2b1, a1b1
          // (small terms first); a weight term is reloaded for the next tap right after its last use.
          const uint4 b1 = bq[j][0];
          const uint4 b2 = bq[j][1];
          const uint4 b3 = bq[j][NS - 1];  // (NS = 2: unused)
          __builtin_amdgcn_sched_barrier(0);
          constexpr int NTERM = (NS == 3) ? 6 : 3;  // NS = 2: a1b2, a2b1, a1b1
#pragma unroll
          for (int t = 0; t < NTERM; ++t) {
            const int as = (NS == 3) ? ((t == 1 || t == 4) ? 1 : (t == 3) ? 2 : 0) : (t == 1 ? 1 : 0);
            const uint4 bv = (NS == 3) ? ((t == 0) ? b3 : (t == 1 || t == 2) ? b2 : b1) : (t == 0 ? b2 : b1);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const uint4 av = acur[as][mb];
              acc[mb] = SW ? dm_mfma16<F16>(bv, av, acc[mb]) : dm_mfma16<F16>(av, bv, acc[mb]);
              if (t < NS) anxt[t][mb] = *reinterpret_cast<const uint4*>(ap[t][mb] + noff);
              if (NS == 3) {
                if (t == 3 && mb == 0) bq[j][2] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + 2 * split_off + j * jstride) + lane16);
                if (t == 4 && mb == 0) bq[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + split_off + j * jstride) + lane16);
              } else {
                if (t == 1 && mb == MB - 1) bq[j][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + split_off + j * jstride) + lane16);
              }
              if (t == NTERM - 1 && mb == MB - 1) bq[j][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wnext + j * jstride) + lane16);
#ifdef DM_X6_FILLER  // timing experiment: vector instructions of the wave's own in the shadow of each MFMA
#pragma unroll
              for (int f = 0; f < DM_X6_FILLER; ++f) {
                if (f & 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fill1));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fill0));
              }
#endif
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        tx = ntx; ty = nty;
        if constexpr (KIND == 4) if ((tap & 3) == 3) {
          // one output parity is complete: registers -> HBM (the input tile stays in LDS for the next parity)
          par = tap >> 2; py = par >> 1; px = par & 1;
          if constexpr (SW) x6_epilogue_sw<MB, 4>(p, acc, reinterpret_cast<float*>(tile + NS * PL), par, py, px, img0, tin, y0, x0, wm, wn, lane, h, tid, wave, oscale * ainv);
          else conv_epilogue<MB, 4>(p, acc, reinterpret_cast<float*>(tile + NS * PL), nullptr, par, py, px, img0, tin, y0, x0,
                                    wm, wn, lane, h, tid, wave, bias_lane, true);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
        }
      }
    }
    DM_TL(3);
    if constexpr (KIND != 4) {
      __builtin_amdgcn_s_setprio(DM_PRIO_STAGE);
      if constexpr (SW) {  // registers -> HBM; the reduction words sit behind the planes (no overlay, no barrier)
        x6_epilogue_sw<MB, KIND>(p, acc, reinterpret_cast<float*>(tile + NS * PL), par, py, px, img0, tin, y0, x0, wm, wn, lane, h, tid, wave, oscale * ainv);
      } else {
        __syncthreads();  // every wave is done reading the input tile: LDS is reused from here
        conv_epilogue<MB, KIND>(p, acc, smem + MT * 68, smem, par, py, px, img0, tin, y0, x0, wm, wn, lane, h, tid, wave, bias_lane, true);
      }
    }
    __syncthreads();  // output staging / reduction words are free again for the next tile
    DM_TL(4);
  }
}

// Three-way bf16 split of a conv weight in B-fragment order (see conv_x6_kernel):
//   out[s][par][tap][cin/16][cout/32][lane][8],  s = 0, 1, 2:  w = out[0] + out[1] + out[2] to 2^-27 |w|
__global__ void pack_conv_weight_x6_kernel(int kind, const float* __restrict__ w, int cin, int cout,
                                           bf16_t* __restrict__ out, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nty = kind == 0 ? 3 : kind == 1 ? 4 : 2;
  const int nt = nty * nty;
  const int ks = kind == 0 ? 3 : 4;
  size_t r = idx;
  const int e = r & 7; r >>= 3;
  const int lane = r & 63; r >>= 6;
  const int nb = r % (cout >> 5); r /= (cout >> 5);
  const int c16 = r % (cin >> 4); r /= (cin >> 4);
  const int tap = r % nt; r /= nt;
  const int par = (int)r;
  const int ty = tap / nty, tx = tap % nty;
  int ky = ty, kx = tx;
  if (kind == 2) { ky = 3 - (par >> 1) - 2 * ty; kx = 3 - (par & 1) - 2 * tx; }
  const int co = nb * 32 + (lane & 31);
  const int ci = c16 * 16 + 8 * (lane >> 5) + e;
  const size_t src = (kind == 2) ? (((size_t)ci * cout + co) * ks + ky) * ks + kx
                                 : (((size_t)co * cin + ci) * ks + ky) * ks + kx;
  float wv = w[src];
  const bf16_t s1 = f2bf(wv);
  wv -= bf2f(s1);
  const bf16_t s2 = f2bf(wv);
  wv -= bf2f(s2);
  out[idx] = s1;
  out[total + idx] = s2;
  out[2 * total + idx] = f2bf(wv);
}

// Two-way fp16 split of a conv weight in the same B-fragment order, times 2^k with max |w| 2^k in [2^10, 2^11) so that both
// terms of every weight of consequence are normal fp16 numbers; scale_inv (= 2^-k, written by weight_scale_kernel) sits
// behind the two planes and is applied to the accumulators by the epilogue (exact).
__global__ void weight_scale_kernel(const float* __restrict__ w, size_t n, float* __restrict__ scale_inv) {
  __shared__ float red[1024];
  float m = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0;
    if (red[0] > 0.f && red[0] < 3.0e38f) (void)frexpf(red[0], &e);  // max = f 2^e, f in [0.5, 1)
    *scale_inv = ldexpf(1.f, e - 11);                               // weights are packed times 2^(11 - e)
  }
}
__global__ void pack_conv_weight_h2_kernel(int kind, const float* __restrict__ w, int cin, int cout,
                                           bf16_t* __restrict__ out, size_t total, const float* __restrict__ scale_inv) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nty = kind == 0 ? 3 : kind == 1 ? 4 : 2;
  const int nt = nty * nty;
  const int ks = kind == 0 ? 3 : 4;
  size_t r = idx;
  const int e = r & 7; r >>= 3;
  const int lane = r & 63; r >>= 6;
  const int nb = r % (cout >> 5); r /= (cout >> 5);
  const int c16 = r % (cin >> 4); r /= (cin >> 4);
  const int tap = r % nt; r /= nt;
  const int par = (int)r;
  const int ty = tap / nty, tx = tap % nty;
  int ky = ty, kx = tx;
  if (kind == 2) { ky = 3 - (par >> 1) - 2 * ty; kx = 3 - (par & 1) - 2 * tx; }
  const int co = nb * 32 + (lane & 31);
  const int ci = c16 * 16 + 8 * (lane >> 5) + e;
  const size_t src = (kind == 2) ? (((size_t)ci * cout + co) * ks + ky) * ks + kx
                                 : (((size_t)co * cin + ci) * ks + ky) * ks + kx;
  float wv = w[src] * (1.0f / *scale_inv);   // power of two: exact
  const _Float16 h1 = (_Float16)wv;            // (weights: both terms to nearest)
  wv -= (float)h1;
  out[idx] = __builtin_bit_cast(bf16_t, h1);
  out[total + idx] = __builtin_bit_cast(bf16_t, (_Float16)wv);
}
