// dm_kernels_init5.hip — init_conv followed by down1.conv1 as ONE 5x5 convolution from the image.
//
// model.py:135 `x = self.init_conv(x)` feeds ONLY down1.conv1 (model.py:54, Block.forward's first op) and nothing stands
// between them: two affine maps, Conv2d(Cin -> C, 3x3, pad 1) then Conv2d(C -> C, 3x3, pad 1).  Their composition is a
// 5x5 convolution from the Cin image channels (K = 25 Cin instead of 9 C = 576: for Cin = 1 the layer that carried 20.7 %
// of the denoiser's FLOPs becomes a K = 25 vector kernel bound by the HBM write of its output), EXCEPT at the image
// border, where conv1 sees zero padding of the intermediate, not init_conv evaluated outside the image.  With
//   M[co][a][ci][b] = sum_c w1[co][c][a] w0[c][ci][b],   Bq[co][a] = sum_c w1[co][c][a] b0[c]        (a, b: 3x3 taps)
//   W25[co][ci][d]  = sum_{a+b=d} M[co][a][ci][b],        bfull[co] = b1[co] + sum_a Bq[co][a]
// the exact result is
//   y[co](p) = bfull[co] + sum_{ci,d} W25[co][ci][d] x[ci](p+d)  -  sum_{a: p+a outside} ( Bq[co][a] + sum_{ci,b} M[co][a][ci][b] x[ci](p+a+b) )
// (x zero outside the image).  The tables are built once per weight pack in float64 (pack_init5_*); the result differs
// from the two-kernel form only by fp32 rounding / association (parity tests: same bounds as every other layer).
// Output: NHWC fp32 (or IEEE half in the bf16 path, where the consumer normalises it) + GroupNorm (mean, M2) partials per
// 8x16-pixel tile — the tile geometry of the 3x3 kernels this replaces, so consumers see the same partial count.
#pragma once
#include "dm_kernels.hip"

struct Init5Args {
  const float* x;        // [B, Cin, H, W]
  const float* w25;      // [Cin][25][C]
  const float* bfull;    // [C]
  const float* m;        // [9 a][Cin][9 b][C]
  const float* bq;       // [9 a][C]
  float* y;              // [B, H, W, C] fp32, or half when out_half
  float* stats_out;      // [B][pout][8][2]
  int B, Cin, H, W, C;
  int tiles_x, tiles_per_img, ntiles;  // 8x16-pixel tiles
  int g_log;             // log2(C / 8)
  int out_half;
};

// Block = 256 threads = 16 channel quads x 16 pixel columns; a thread owns one column (8 pixels) x 4 channels of the tile.
__global__ __launch_bounds__(256, 2) void init5_kernel(const Init5Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PH = 12, PW = 20;               // image patch: (8 + 4) x (16 + 4)
  const int Cin = p.Cin;
  float* xs = smem;                              // [Cin][PH][PW]
  float* w25 = xs + ((Cin * PH * PW + 3) & ~3);  // [Cin][25][64]
  float* ms = w25 + Cin * 25 * 64;               // [9][Cin][9][64]
  float* bqs = ms + 81 * Cin * 64;               // [9][64]
  float* bfs = bqs + 9 * 64;                     // [64]
  float* red = bfs + 64;                         // [256] + [8]
  const int tid = threadIdx.x, c4 = tid & 15, col = tid >> 4;
  const int cb = blockIdx.y * 64;
  for (int i = tid; i < Cin * 25 * 16; i += 256) {
    const int q = i & 15, r = i >> 4;
    *reinterpret_cast<float4*>(w25 + r * 64 + q * 4) = *reinterpret_cast<const float4*>(p.w25 + (size_t)r * p.C + cb + q * 4);
  }
  for (int i = tid; i < 81 * Cin * 16; i += 256) {
    const int q = i & 15, r = i >> 4;
    *reinterpret_cast<float4*>(ms + r * 64 + q * 4) = *reinterpret_cast<const float4*>(p.m + (size_t)r * p.C + cb + q * 4);
  }
  for (int i = tid; i < 9 * 16; i += 256) {
    const int q = i & 15, r = i >> 4;
    *reinterpret_cast<float4*>(bqs + r * 64 + q * 4) = *reinterpret_cast<const float4*>(p.bq + (size_t)r * p.C + cb + q * 4);
  }
  if (tid < 16) *reinterpret_cast<float4*>(bfs + tid * 4) = *reinterpret_cast<const float4*>(p.bfull + cb + tid * 4);
  const int G = 1 << p.g_log, gpt_log = 6 - p.g_log, gpt = 1 << gpt_log;  // channels per group; groups in these 64 channels
  const int qg_log = p.g_log - 2;                                         // log2(channel quads per group)

  for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
    const int n = t / p.tiles_per_img, tin = t - n * p.tiles_per_img;
    const int y0 = (tin / p.tiles_x) * 8, x0 = (tin % p.tiles_x) * 16;
    __syncthreads();  // previous tile's readers of xs / red are done (first tile: the tables are in LDS)
    for (int i = tid; i < Cin * PH * PW; i += 256) {
      const int px = i % PW, r = i / PW, py = r % PH, ci = r / PH;
      const int gy = y0 - 2 + py, gx = x0 - 2 + px;
      float v = 0.f;
      if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = p.x[(((size_t)n * Cin + ci) * p.H + gy) * p.W + gx];
      xs[i] = v;
    }
    __syncthreads();
    const float4 bf = *reinterpret_cast<const float4*>(bfs + c4 * 4);
    float4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = bf;
    for (int ci = 0; ci < Cin; ++ci) {
      float xv[PH][5];  // this column's window: rows 0..11, columns col..col+4
#pragma unroll
      for (int r = 0; r < PH; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) xv[r][c] = xs[(ci * PH + r) * PW + col + c];
#pragma unroll
      for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const float4 w = *reinterpret_cast<const float4*>(w25 + ((ci * 25 + dy * 5 + dx) * 64) + c4 * 4);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float v = xv[k + dy][dx];
            acc[k].x = fmaf(w.x, v, acc[k].x); acc[k].y = fmaf(w.y, v, acc[k].y);
            acc[k].z = fmaf(w.z, v, acc[k].z); acc[k].w = fmaf(w.w, v, acc[k].w);
          }
        }
    }
    // border: take back what conv1's zero padding of the intermediate removes (taps a whose position p + a is outside)
    const int gx = x0 + col;
    const bool colb = gx == 0 || gx == p.W - 1;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      const int gy = y0 + k;
      if (!(colb || gy == 0 || gy == p.H - 1)) continue;
      float4 sub = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int a = 0; a < 9; ++a) {
        const int ay = a / 3 - 1, ax = a % 3 - 1;
        if ((unsigned)(gy + ay) < (unsigned)p.H && (unsigned)(gx + ax) < (unsigned)p.W) continue;
        const float4 b = *reinterpret_cast<const float4*>(bqs + a * 64 + c4 * 4);
        sub.x += b.x; sub.y += b.y; sub.z += b.z; sub.w += b.w;
        for (int ci = 0; ci < Cin; ++ci)
          for (int bb = 0; bb < 9; ++bb) {
            const int by = bb / 3 - 1, bx = bb % 3 - 1;
            const float v = xs[(ci * PH + (k + 2 + ay + by)) * PW + col + 2 + ax + bx];
            const float4 mw = *reinterpret_cast<const float4*>(ms + (((a * Cin + ci) * 9 + bb) * 64) + c4 * 4);
            sub.x = fmaf(mw.x, v, sub.x); sub.y = fmaf(mw.y, v, sub.y); sub.z = fmaf(mw.z, v, sub.z); sub.w = fmaf(mw.w, v, sub.w);
          }
      }
      // (dynamic k: select instead of indexing the register array)
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q == k) { acc[q].x -= sub.x; acc[q].y -= sub.y; acc[q].z -= sub.z; acc[q].w -= sub.w; }
    }
    // GroupNorm (mean, M2) partial of this tile: two passes, fixed summation order
    if (p.stats_out) {
      const float inv_cnt = 1.0f / (float)(128 << p.g_log);
      float mu = 0.f;
      for (int pass = 0; pass < 2; ++pass) {
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dx = acc[k].x - mu, dy = acc[k].y - mu, dz = acc[k].z - mu, dw = acc[k].w - mu;
          sm += pass ? (dx * dx + dy * dy) + (dz * dz + dw * dw) : (dx + dy) + (dz + dw);
        }
        red[tid] = sm;
        __syncthreads();
        if (tid < gpt) {
          float tot = 0.f;
          const int q0 = tid << qg_log, nq = 1 << qg_log;
          for (int cc = 0; cc < 16; ++cc)
            for (int q = 0; q < nq; ++q) tot += red[cc * 16 + q0 + q];
          if (pass == 0) red[256 + tid] = tot * inv_cnt;
          else {
            float* so = p.stats_out + (((size_t)n * p.tiles_per_img + tin) * DM_GROUPS + blockIdx.y * gpt + tid) * 2;
            so[0] = red[256 + tid];
            so[1] = tot;
          }
        }
        __syncthreads();
        if (pass == 0) mu = red[256 + (c4 >> qg_log)];
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

// Tables of init5_kernel from init_conv (w0 [C][Cin][3][3], b0 [C]) and down1.conv1 (w1 [C][C][3][3], b1 [C]); float64 sums.
//   m  [9 a][Cin][9 b][C],  bq [9 a][C]
__global__ void pack_init5_m_kernel(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                                    int Cin, int C, float* __restrict__ m, float* __restrict__ bq) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nm = (size_t)81 * Cin * C;
  if (idx < nm) {
    size_t r = idx;
    const int co = r % C; r /= C;
    const int b = r % 9; r /= 9;
    const int ci = r % Cin; r /= Cin;
    const int a = (int)r;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += (double)w1[((size_t)co * C + c) * 9 + a] * (double)w0[((size_t)c * Cin + ci) * 9 + b];
    m[idx] = (float)s;
  } else if (idx < nm + (size_t)9 * C) {
    const size_t r = idx - nm;
    const int co = r % C, a = (int)(r / C);
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += (double)w1[((size_t)co * C + c) * 9 + a] * (double)b0[c];
    bq[r] = (float)s;
  }
}
//   w25 [Cin][25][C] = sum_{a+b=d} (float64 sums of the float64 products again, not of the rounded m),  bfull [C]
__global__ void pack_init5_w_kernel(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1,
                                    const float* __restrict__ b1, int Cin, int C, float* __restrict__ w25, float* __restrict__ bfull) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nw = (size_t)25 * Cin * C;
  if (idx < nw) {
    size_t r = idx;
    const int co = r % C; r /= C;
    const int d = r % 25; r /= 25;
    const int ci = (int)r;
    const int dy = d / 5 - 2, dx = d % 5 - 2;
    double s = 0.0;
    for (int a = 0; a < 9; ++a) {
      const int by = dy - (a / 3 - 1), bx = dx - (a % 3 - 1);
      if (by < -1 || by > 1 || bx < -1 || bx > 1) continue;
      const int b = (by + 1) * 3 + bx + 1;
      for (int c = 0; c < C; ++c) s += (double)w1[((size_t)co * C + c) * 9 + a] * (double)w0[((size_t)c * Cin + ci) * 9 + b];
    }
    w25[idx] = (float)s;
  } else if (idx < nw + (size_t)C) {
    const int co = (int)(idx - nw);
    double s = (double)b1[co];
    for (int a = 0; a < 9; ++a)
      for (int c = 0; c < C; ++c) s += (double)w1[((size_t)co * C + c) * 9 + a] * (double)b0[c];
    bfull[co] = (float)s;
  }
}
