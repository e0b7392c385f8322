// dm_kernels_init5.hip — init_conv followed by down1.conv1 as ONE 5x5 convolution from the image.
//
// model.py:135 `x = self.init_conv(x)` feeds ONLY down1.conv1 (model.py:54, Block.forward's first op) and nothing stands
// between them: two affine maps, Conv2d(Cin -> C, 3x3, pad 1) then Conv2d(C -> C, 3x3, pad 1).  Their composition is a
// 5x5 convolution from the Cin image channels (K = 25 Cin instead of 9 C = 576: for Cin = 1 the layer that carried 20.7 %
// of the denoiser's FLOPs becomes a K = 25 vector kernel bound by the HBM write of its output) — EXCEPT at the image
// border, where conv1 sees zero padding of the intermediate, not init_conv evaluated outside the image: a pixel in the
// top row must not receive conv1's taps a = (-1, .), and so on.  Hence NINE weight sets, one per (row class, column
// class) in {interior, first, last}^2:
//   W[rc][cc][co][ci][d] = sum over conv1 taps a valid for (rc, cc), init taps b with a + b = d, channels c of
//                          w1[co][c][a] w0[c][ci][b]                                   (d: 5x5 offsets)
//   bias[rc][cc][co]     = b1[co] + sum over valid a, c of w1[co][c][a] b0[c]
// built once per weight pack with float64 sums (pack_init5_kernel).  y[co](p) = bias + sum_{ci,d} W x[ci](p + d) with the
// set of p's classes and x zero outside the image is then EXACTLY the two-layer result; it differs from the two-kernel
// form only by fp32 rounding / association (parity tests: the same bounds as every other layer).
// Output: NHWC fp32 (IEEE half in the bf16 path, where the consumer normalises it) + GroupNorm (mean, M2) partials per
// 8x16-pixel tile — the tile geometry of the 3x3 kernels this replaces, so consumers see the same partial count.
#pragma once
#include "dm_kernels.hip"

// acc.xy += w.xy * (x, x): a PLAIN v_pk_fma_f32 on a pair that holds the image value twice (one ds_read2_b32 with both offsets
// equal delivers it).  No op_sel: measured on gfx950 (tools/diag/corun_forward.py), v_pk_fma_f32 with
// op_sel:[0,1,0] (low half of the result from the HIGH half of src1) returns the low half computed from src1's LOW half in lanes
// 48-63 now and then while ANOTHER wave of the SIMD streams bf16 MFMAs - i.e. whenever a split-bf16 kernel of another stream
// shares the CU.  Plain packed FMAs, like scalar ones, are not affected.
typedef float i5_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void i5_fma(i5_f2& acc, i5_f2 w, i5_f2 xx) { acc = __builtin_elementwise_fma(w, xx, acc); }
// (v, v) from ONE LDS word: ds_read2_b32 with both offsets equal (offsets in dwords, 0..255).  hipcc does not track the LDS
// counter of inline asm: the reads of a column are followed by i5_lds_wait, which also ties the values to the wait.
#define I5_LDS_PAIR(dst, addr, off) asm volatile("ds_read2_b32 %0, %1 offset0:" #off " offset1:" #off : "=v"(dst) : "v"(addr) : "memory")
__device__ __forceinline__ void i5_lds_wait(i5_f2 (&x)[12]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
               "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
}

struct Init5Args {
  const float* x;        // [B, Cin, H, W]
  const float* wc;       // [3 rc][3 cc][25 Cin + 1][C]  (row 25 Cin = the bias)
  float* y;              // [B, H, W, C] fp32, or half when out_half
  float* stats_out;      // [B][tiles_per_img][8][2]
  int B, Cin, H, W, C;
  int tiles_x, tiles_per_img;  // 8x16-pixel tiles
  int g_log;             // log2(C / 8)
  int out_half;
};

// Block = 256 threads = 16 pixel columns x 16 channel quads; a thread owns one column (8 pixels) x 4 channels of a tile.
// A workgroup keeps ONE tile position (blockIdx.x % tiles_per_img) and walks the images: the weight sets it needs — interior
// plus at most one special row class and the special column classes of that position — are loaded into LDS once.
__global__ __launch_bounds__(256, 3) void init5_kernel(const Init5Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PH = 12, PW = 20;               // image patch: (8 + 4) x (16 + 4)
  const int Cin = p.Cin;
  const int SET = (25 * Cin + 1) * 64;          // floats per weight set (64 of the C output channels)
  const int tid = threadIdx.x, c4 = tid & 15, col = tid >> 4, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.y * 64;
  const int tin = blockIdx.x % p.tiles_per_img;
  const int y0 = (tin / p.tiles_x) * 8, x0 = (tin % p.tiles_x) * 16;
  const int rsp = (y0 == 0) ? 1 : (y0 + 8 == p.H) ? 2 : 0;          // this position's special row class (H >= 16: at most one)
  const bool cl = x0 == 0, cr = x0 + 16 == p.W;                      // special column classes present
  const int ncs = (cl ? 1 : 0) + (cr ? 1 : 0);
  float* xs = smem;                                                  // [Cin][PH][PW]
  float* ws = xs + ((Cin * PH * PW + 3) & ~3);                       // [(1 + (rsp != 0)) x (1 + ncs)] sets
  // local set index: lr * (1 + ncs) + lc;  lc: 0 interior, then left (if present), then right (if present)
  for (int lr = 0; lr < (rsp ? 2 : 1); ++lr)
    for (int lc = 0; lc <= ncs; ++lc) {
      const int rc = lr ? rsp : 0;
      const int cc = (lc == 0) ? 0 : (lc == 1 && cl) ? 1 : 2;
      const float* src = p.wc + (size_t)(rc * 3 + cc) * (25 * Cin + 1) * p.C + cb;
      float* dst = ws + (lr * (1 + ncs) + lc) * SET;
      for (int i = tid; i < (25 * Cin + 1) * 16; i += 256) {
        const int q = i & 15, r = i >> 4;
        *reinterpret_cast<float4*>(dst + r * 64 + q * 4) = *reinterpret_cast<const float4*>(src + (size_t)r * p.C + q * 4);
      }
    }
  float* red = ws + (rsp ? 2 : 1) * (1 + ncs) * SET;                 // [4 waves][8] x 2 + [8]
  const int gx = x0 + col;
  const int lc_t = (gx == 0) ? 1 : (gx == p.W - 1) ? ncs : 0;        // this thread's column class (local index)
  // (the interior row class' set of this thread's column class is ws + lc_t * SET; the special row class' follows (1 + ncs) sets later)
  const int gpt_log = 6 - p.g_log, gpt = 1 << gpt_log;               // GroupNorm groups in these 64 channels
  const int qg_log = p.g_log - 2;                                    // log2(channel quads per group): 1 .. 4
  const float inv_cnt = 1.0f / (float)(128 << p.g_log);
  const int n_step = gridDim.x / p.tiles_per_img;

  // the patch of the NEXT image is fetched into registers while the current one is computed (<= 4 values per thread)
  constexpr int NPF = 4;
  float pf[NPF];
  auto fetch = [&](int n) {
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
      const int i = tid + q * 256;
      const int px = i % PW, r = i / PW, py = r % PH, ci = r / PH;
      const int iy = y0 - 2 + py, ix = x0 - 2 + px;
      float v = 0.f;
      if (i < Cin * PH * PW && n < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = p.x[(((size_t)n * Cin + ci) * p.H + iy) * p.W + ix];
      pf[q] = v;
    }
  };
  fetch(blockIdx.x / p.tiles_per_img);
  for (int n = blockIdx.x / p.tiles_per_img; n < p.B; n += n_step) {
    __syncthreads();  // previous image's readers of xs are done (first pass: the weight sets are in LDS)
#pragma unroll
    for (int q = 0; q < NPF; ++q)
      if (tid + q * 256 < Cin * PH * PW) xs[tid + q * 256] = pf[q];
    __syncthreads();
    fetch(n + n_step);
    // (opaque copies: the weights are read from LDS per image; hoisted out of the image loop they would take 200 registers)
    int oi = lc_t * SET, osp = ((1 + ncs) + lc_t) * SET;  // (integer offsets: an opaque POINTER would become a flat address)
    asm volatile("" : "+v"(oi), "+v"(osp));
    const float* wi = ws + oi;
    const float* wsp = ws + osp;
    const float4 bf = *reinterpret_cast<const float4*>(wi + 25 * Cin * 64 + c4 * 4);
    i5_f2 a01[8], a23[8];  // channels (0,1) and (2,3) of the thread's quad, per pixel
#pragma unroll
    for (int k = 0; k < 8; ++k) { a01[k] = i5_f2{bf.x, bf.y}; a23[k] = i5_f2{bf.z, bf.w}; }
    float4 aspv = rsp ? *reinterpret_cast<const float4*>(wsp + 25 * Cin * 64 + c4 * 4) : bf;  // the special row's pixel, recomputed
    i5_f2 s01 = {aspv.x, aspv.y}, s23 = {aspv.z, aspv.w};
    for (int ci = 0; ci < Cin; ++ci)
#pragma unroll 1
      for (int dx = 0; dx < 5; ++dx) {  // (rolled: unrolled, hipcc loads several columns ahead and spills 165 registers)
        i5_f2 xr[PH];  // column col + dx of the patch, rows 0..11, every value as a pair (v, v)
        static_assert(PH == 12 && PW == 20, "row offsets below");
        const unsigned xa = (unsigned)(unsigned long long)(xs + ci * PH * PW + col + dx);  // (LDS aperture: the low 32 bits are the LDS address)
        I5_LDS_PAIR(xr[0], xa, 0); I5_LDS_PAIR(xr[1], xa, 20); I5_LDS_PAIR(xr[2], xa, 40); I5_LDS_PAIR(xr[3], xa, 60);
        I5_LDS_PAIR(xr[4], xa, 80); I5_LDS_PAIR(xr[5], xa, 100); I5_LDS_PAIR(xr[6], xa, 120); I5_LDS_PAIR(xr[7], xa, 140);
        I5_LDS_PAIR(xr[8], xa, 160); I5_LDS_PAIR(xr[9], xa, 180); I5_LDS_PAIR(xr[10], xa, 200); I5_LDS_PAIR(xr[11], xa, 220);
        i5_lds_wait(xr);
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
          const float4 w = *reinterpret_cast<const float4*>(wi + ((ci * 25 + dy * 5 + dx) * 64) + c4 * 4);
          const i5_f2 w01 = {w.x, w.y}, w23 = {w.z, w.w};
#pragma unroll
          for (int k = 0; k < 8; ++k) { i5_fma(a01[k], w01, xr[k + dy]); i5_fma(a23[k], w23, xr[k + dy]); }
        }
        if (rsp) {  // (uniform) first row of a top tile / last row of a bottom tile: its own weight set
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) {
            const float4 w = *reinterpret_cast<const float4*>(wsp + ((ci * 25 + dy * 5 + dx) * 64) + c4 * 4);
            const i5_f2 v = (rsp == 1) ? xr[dy] : xr[7 + dy];
            i5_fma(s01, i5_f2{w.x, w.y}, v); i5_fma(s23, i5_f2{w.z, w.w}, v);
          }
        }
      }
    float4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = make_float4(a01[k].x, a01[k].y, a23[k].x, a23[k].y);
    const float4 asp = make_float4(s01.x, s01.y, s23.x, s23.y);
    if (rsp == 1) acc[0] = asp;
    if (rsp == 2) acc[7] = asp;
    // GroupNorm (mean, M2) partial of this tile: two passes; lanes of a group meet by shuffles, the four waves through LDS
    // (fixed order: the result depends neither on the batch size nor on the launch shape)
    if (p.stats_out) {
      float mu = 0.f;
      for (int pass = 0; pass < 2; ++pass) {
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dx = acc[k].x - mu, dy = acc[k].y - mu, dz = acc[k].z - mu, dw = acc[k].w - mu;
          sm += pass ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : (dx + dy) + (dz + dw);
        }
        for (int o = 1; o < (1 << qg_log); o <<= 1) sm += __shfl_xor(sm, o, 64);  // the group's channel quads
        sm += __shfl_xor(sm, 16, 64);                                             // the wave's four columns
        sm += __shfl_xor(sm, 32, 64);
        float* rr = red + pass * 32;
        if (lane < 16 && (c4 & ((1 << qg_log) - 1)) == 0) rr[wave * 8 + (c4 >> qg_log)] = sm;
        __syncthreads();
        const float tot = (rr[c4 >> qg_log] + rr[8 + (c4 >> qg_log)]) + (rr[16 + (c4 >> qg_log)] + rr[24 + (c4 >> qg_log)]);
        if (pass == 0) mu = tot * inv_cnt;
        else if (tid < gpt) {
          const float t2 = (rr[tid] + rr[8 + tid]) + (rr[16 + tid] + rr[24 + tid]);
          float* so = p.stats_out + (((size_t)n * p.tiles_per_img + tin) * DM_GROUPS + blockIdx.y * gpt + tid) * 2;
          so[0] = ((red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid])) * inv_cnt;
          so[1] = t2;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t o = (((size_t)n * p.H + y0 + k) * p.W + gx) * p.C + cb + c4 * 4;
      if (p.out_half) {
        uint2 h2;
        h2.x = (unsigned)f2h(acc[k].x) | ((unsigned)f2h(acc[k].y) << 16);
        h2.y = (unsigned)f2h(acc[k].z) | ((unsigned)f2h(acc[k].w) << 16);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.y) + o) = h2;
      } else {
        *reinterpret_cast<float4*>(p.y + o) = acc[k];
      }
    }
  }
}

// The nine weight sets of init5_kernel from init_conv (w0 [C][Cin][3][3], b0 [C]) and down1.conv1 (w1 [C][C][3][3], b1 [C]):
//   wc[rc][cc][ci * 25 + d][co]  and the bias in row 25 Cin;  float64 sums.  rc / cc: 0 interior, 1 first row / column
//   (conv1 taps with ay = -1 / ax = -1 fall on the zero padding), 2 last row / column (ay = +1 / ax = +1 do).
__global__ void pack_init5_kernel(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                                  const float* __restrict__ b1, int Cin, int C, float* __restrict__ wc) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = 25 * Cin + 1;
  if (idx >= (size_t)9 * rows * C) return;
  size_t r = idx;
  const int co = r % C; r /= C;
  const int row = r % rows; r /= rows;
  const int cc = r % 3, rc = (int)(r / 3);
  auto valid = [&](int a) {
    const int ay = a / 3 - 1, ax = a % 3 - 1;
    return !((rc == 1 && ay < 0) || (rc == 2 && ay > 0) || (cc == 1 && ax < 0) || (cc == 2 && ax > 0));
  };
  double s = 0.0;
  if (row == 25 * Cin) {
    s = (double)b1[co];
    for (int a = 0; a < 9; ++a)
      if (valid(a))
        for (int c = 0; c < C; ++c) s += (double)w1[((size_t)co * C + c) * 9 + a] * (double)b0[c];
  } else {
    const int ci = row / 25, d = row % 25;
    const int dy = d / 5 - 2, dx = d % 5 - 2;
    for (int a = 0; a < 9; ++a) {
      if (!valid(a)) continue;
      const int by = dy - (a / 3 - 1), bx = dx - (a % 3 - 1);
      if (by < -1 || by > 1 || bx < -1 || bx > 1) continue;
      const int b = (by + 1) * 3 + bx + 1;
      for (int c = 0; c < C; ++c) s += (double)w1[((size_t)co * C + c) * 9 + a] * (double)w0[((size_t)c * Cin + ci) * 9 + b];
    }
  }
  wc[idx] = (float)s;
}
