// dm_kernels_generic.hip — the denoiser for EVERY geometry the reference accepts (diffusion/model.py:72-73,120-151):
// any H, W divisible by 8 (24, 40, 48, 56, 96 ...: the levels then are 3x3, 5x5, 6x6 ... pixels), any model_channels
// divisible by 8 (GroupNorm(8, C): 32, 96, 192 ...), any in/out channel count.  The tuned kernels of dm_kernels*.hip
// assume power-of-two image sides and C in {64,128,256,512} (shifts instead of divisions, 8x16-pixel tiles, Winograd /
// split-bf16 weight forms); everything outside that set runs here: the same NHWC fp32 layout, the same exact-fp32 matrix
// instruction (v_mfma_f32_32x32x2_f32), plain index arithmetic, ragged tiles, and GroupNorm as two small stand-alone
// kernels (statistics, then normalise + SiLU + time embedding in place) instead of the fused prologue/epilogue forms.
// Slower than the tuned path (about 2 extra passes over every activation), never a fallback to anything but HIP.
#pragma once
#include "dm_kernels.hip"

struct GConvArgs {
  const float* x1;   // [B,H,W,C1]
  const float* x2;   // [B,H,W,C2] or null: channel concat [x1, x2] (model.py:146-148)
  const float* wp;   // B-fragment order [parity][tap][Cin/8][ceil(Cout/32)][64 lanes][4], zero-padded past Cout
  const float* bias; // [Cout]
  float* y;          // [B,OH,OW,Cout]
  int B, H, W;       // input dims
  int C1, C2, Cout;
  int OH, OW;        // output dims
  int Hb, Wb;        // base grid of the 8x8-pixel tiles: the output grid (KIND 0, 1) or the input grid (KIND 2)
  int tiles_x, tiles_y;
  int nb32;          // ceil(Cout / 32)
};

// KIND 0: Conv2d 3x3 s1 p1   KIND 1: Conv2d 4x4 s2 p1   KIND 2: ConvTranspose2d 4x4 s2 p1 (one output parity per blockIdx.z)
// Workgroup = 256 threads = 4 waves as 2 (32 pixels each of an 8x8 tile) x 2 (32 output channels each); blockIdx.y walks
// 64-channel blocks of Cout.  K = taps x input channels in chunks of 32 channels staged through LDS with the halo; the
// weights come from L2 in B-fragment order (one coalesced 1 KiB read per wave, tap and 8 channels).
template <int KIND>
__global__ __launch_bounds__(256) void gconv_kernel(const GConvArgs p) {
  constexpr int KC = 32, RS = KC + 4;
  constexpr int S = (KIND == 1) ? 2 : 1;
  constexpr int NTY = (KIND == 0) ? 3 : (KIND == 1) ? 4 : 2;
  constexpr int NT = NTY * NTY;
  constexpr int HT = (KIND == 0) ? 10 : (KIND == 1) ? 18 : 9;  // halo tile edge (input pixels)
  __shared__ __attribute__((aligned(16))) float tile[HT * HT * RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int par = (KIND == 2) ? (int)blockIdx.z : 0, py = par >> 1, px = par & 1;
  const int tpi = p.tiles_x * p.tiles_y;
  const int n = blockIdx.x / tpi;
  const int tin = blockIdx.x - n * tpi;
  const int tyi = tin / p.tiles_x;
  const int y0 = tyi * 8, x0 = (tin - tyi * p.tiles_x) * 8;
  const int hy0 = (KIND == 0) ? y0 - 1 : (KIND == 1) ? 2 * y0 - 1 : y0 - 1 + py;
  const int hx0 = (KIND == 0) ? x0 - 1 : (KIND == 1) ? 2 * x0 - 1 : x0 - 1 + px;
  const int Cin = p.C1 + p.C2;
  const int nbw = blockIdx.y * 2 + wn;       // this wave's 32-channel block of the output
  const bool wave_on = nbw < p.nb32;         // (Cout <= 32: the second half-block does not exist)
  const int m = wm * 32 + (lane & 31);       // this lane's pixel of the 8x8 tile (A-operand row)
  const int abase = (((m >> 3) * S) * HT + (m & 7) * S) * RS + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const size_t jstride = (size_t)p.nb32 * 256;
  const size_t tapstride = (size_t)(Cin >> 3) * jstride;
  const float* wbase = p.wp + (size_t)par * NT * tapstride + (size_t)nbw * 256 + lane * 4;

  for (int c0 = 0; c0 < Cin; c0 += KC) {
    if (c0) __syncthreads();  // the previous chunk's LDS reads are done
    for (int idx = tid; idx < HT * HT * (KC / 4); idx += 256) {
      const int pix = idx >> 3, q = idx & 7;
      const int hy = pix / HT, hx = pix - hy * HT;
      const int iy = hy0 + hy, ix = hx0 + hx;
      const int c = c0 + 4 * q;  // (C1 is a multiple of 8: a channel quad never straddles the two inputs)
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < Cin && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        const size_t pixel = ((size_t)n * p.H + iy) * p.W + ix;
        v = (c < p.C1) ? *reinterpret_cast<const float4*>(p.x1 + pixel * p.C1 + c)
                       : *reinterpret_cast<const float4*>(p.x2 + pixel * p.C2 + (c - p.C1));
      }
      *reinterpret_cast<float4*>(tile + pix * RS + q * 4) = v;
    }
    __syncthreads();
    if (wave_on) {
      const int nj = min(KC, Cin - c0) >> 3;  // 8-channel groups of this chunk
      for (int tap = 0; tap < NT; ++tap) {
        const int ty = tap / NTY, tx = tap - ty * NTY;
        const int toff = (ty * HT + tx) * RS;
        const float* wt = wbase + (size_t)tap * tapstride + (size_t)(c0 >> 3) * jstride;
        for (int j = 0; j < nj; ++j) {
          const float4 a = *reinterpret_cast<const float4*>(tile + abase + toff + j * 8);
          const float4 b = *reinterpret_cast<const float4*>(wt + (size_t)j * jstride);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
      }
    }
  }
  const int co = nbw * 32 + (lane & 31);
  if (!wave_on || co >= p.Cout) return;
  const float bv = p.bias[co];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator row -> tile pixel
    const int y = y0 + (mr >> 3), x = x0 + (mr & 7);
    if (y < p.Hb && x < p.Wb) {
      const int oy = (KIND == 2) ? 2 * y + py : y, ox = (KIND == 2) ? 2 * x + px : x;
      p.y[(((size_t)n * p.OH + oy) * p.OW + ox) * p.Cout + co] = acc[r] + bv;
    }
  }
}

// Pack a conv weight into B-fragment order with the output channels padded to a multiple of 32 (zeros):
// the any-geometry form of pack_conv_weight_kernel (identical to it when cout % 32 == 0).
__global__ void gpack_conv_weight_kernel(int kind, const float* __restrict__ w, int cin, int cout,
                                         float* __restrict__ out, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int nty = kind == 0 ? 3 : kind == 1 ? 4 : kind == 2 ? 2 : 1;
  const int nt = nty * nty;
  const int ks = kind == 0 ? 3 : kind == 3 ? 1 : 4;
  const int nb32 = (cout + 31) >> 5;
  size_t r = idx;
  const int e = r & 3; r >>= 2;
  const int lane = r & 63; r >>= 6;
  const int nb = r % nb32; r /= nb32;
  const int c8 = r % (cin >> 3); r /= (cin >> 3);
  const int tap = r % nt; r /= nt;
  const int par = (int)r;
  const int ty = tap / nty, tx = tap % nty;
  int ky = ty, kx = tx;
  if (kind == 2) { ky = 3 - (par >> 1) - 2 * ty; kx = 3 - (par & 1) - 2 * tx; }
  const int co = nb * 32 + (lane & 31);
  const int ci = c8 * 8 + 4 * (lane >> 5) + e;
  float v = 0.f;
  if (co < cout)
    v = (kind == 2) ? w[(((size_t)ci * cout + co) * ks + ky) * ks + kx] : w[(((size_t)co * cin + ci) * ks + ky) * ks + kx];
  out[idx] = v;
}

// Sum over the workgroup in a fixed order (lanes by shuffles, the four waves through LDS): every thread gets the total.
__device__ __forceinline__ float g_block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();  // red is free again
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// GroupNorm(8, C) statistics (model.py:48-49,100,112): one workgroup per (image, group); mean, then the biased variance
// about that mean (two passes: no cancellation), stats[n][g] = (mean, 1/sqrt(var + eps)).
__global__ __launch_bounds__(256) void gstats_kernel(const float* __restrict__ x, float* __restrict__ stats, int HW, int C) {
  __shared__ float red[4];
  const int n = blockIdx.x >> 3, g = blockIdx.x & 7;
  const int G = C / DM_GROUPS;
  const float* xb = x + (size_t)n * HW * C + g * G;
  const int total = HW * G;
  float s = 0.f;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int pp = idx / G, c = idx - pp * G;
    s += xb[(size_t)pp * C + c];
  }
  const float mean = g_block_sum(s, red) / (float)total;
  float q = 0.f;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int pp = idx / G, c = idx - pp * G;
    const float dlt = xb[(size_t)pp * C + c] - mean;
    q += dlt * dlt;
  }
  const float var = g_block_sum(q, red) / (float)total;
  if (threadIdx.x == 0) {
    stats[(size_t)blockIdx.x * 2] = mean;
    stats[(size_t)blockIdx.x * 2 + 1] = 1.0f / sqrtf(var + DM_GN_EPS);
  }
}

// Per-image tables scale[c] = rstd * gamma[c], shift[c] = beta[c] - mean * scale[c], temb[c] (0 without time embedding).
__device__ __forceinline__ void g_norm_tables(const float* stats, const float* gamma, const float* beta, const float* temb,
                                              int n, int C, float* s_scale, float* s_shift, float* s_temb) {
  const int G = C / DM_GROUPS;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* st = stats + ((size_t)n * DM_GROUPS + c / G) * 2;
    const float sc = st[1] * gamma[c];
    s_scale[c] = sc;
    s_shift[c] = beta[c] - st[0] * sc;
    if (s_temb) s_temb[c] = temb ? temb[c] : 0.f;
  }
  __syncthreads();
}

__device__ __forceinline__ float g_silu(float v) { return v / (1.0f + expf(-v)); }

// x <- SiLU(GroupNorm(x)) + temb, in place (Block.forward, model.py:54-62; bottleneck.1-2, model.py:100-101).
// grid = B x (pixel chunks of ppb pixels); temb: null, or + n * temb_stride (0: one row shared by the batch) indexed by the step counter.
struct GNormArgs {
  float* x; const float* stats; const float* gamma; const float* beta;
  const float* temb; const int* step_ptr; int step_off, temb_stride, temb_step_stride;
  int HW, C, ppb, chunks;  // pixels per block, blocks per image
};
__global__ __launch_bounds__(256) void gnorm_kernel(const GNormArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_scale = smem;
  float* s_shift = smem + a.C;
  float* s_temb = smem + 2 * a.C;
  const int n = blockIdx.x / a.chunks;
  const int step = (a.step_ptr ? *a.step_ptr : 0) + a.step_off;
  const float* te = a.temb ? a.temb + (size_t)step * a.temb_step_stride + (size_t)n * a.temb_stride : nullptr;
  g_norm_tables(a.stats, a.gamma, a.beta, te, n, a.C, s_scale, s_shift, s_temb);
  const int p0 = (blockIdx.x - n * a.chunks) * a.ppb;
  const int np = min(a.ppb, a.HW - p0);
  const int V = a.C >> 2;
  float* xb = a.x + ((size_t)n * a.HW + p0) * a.C;
  for (int idx = threadIdx.x; idx < np * V; idx += 256) {
    const int pp = idx / V, c = (idx - pp * V) * 4;
    float4 v = *reinterpret_cast<float4*>(xb + (size_t)pp * a.C + c);
    v.x = g_silu(fmaf(v.x, s_scale[c], s_shift[c])) + s_temb[c];
    v.y = g_silu(fmaf(v.y, s_scale[c + 1], s_shift[c + 1])) + s_temb[c + 1];
    v.z = g_silu(fmaf(v.z, s_scale[c + 2], s_shift[c + 2])) + s_temb[c + 2];
    v.w = g_silu(fmaf(v.w, s_scale[c + 3], s_shift[c + 3])) + s_temb[c + 3];
    *reinterpret_cast<float4*>(xb + (size_t)pp * a.C + c) = v;
  }
}

// final_conv (model.py:111-115,151): GroupNorm -> SiLU -> Conv2d 1x1 (C -> Cout), NHWC in, NCHW out; any C % 8 == 0, any Cout.
// grid = B x (pixel chunks of ppb pixels); 16 lanes cooperate on a pixel, output channels in groups of 4.
__global__ __launch_bounds__(256) void gfinal_kernel(const float* __restrict__ u, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ out, int HW, int C, int Cout, int ppb, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_scale = smem;
  float* s_shift = smem + C;
  float* s_w = smem + 2 * C;  // [Cout][C]
  const int n = blockIdx.x / chunks;
  for (int i = threadIdx.x; i < Cout * C; i += 256) s_w[i] = w[i];
  g_norm_tables(stats, gamma, beta, nullptr, n, C, s_scale, s_shift, nullptr);
  const int p0 = (blockIdx.x - n * chunks) * ppb;
  const int np = min(ppb, HW - p0);
  const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int V = C >> 2;
  for (int pb = 0; pb < np; pb += 16) {  // uniform trip count (shuffles inside)
    const int pp = pb + pl;
    const bool on = pp < np;
    const float* up = u + ((size_t)n * HW + p0 + (on ? pp : 0)) * C;
    for (int co0 = 0; co0 < Cout; co0 += 4) {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c4 = q; c4 < V; c4 += 16) {
        const int c = c4 * 4;
        float4 v = *reinterpret_cast<const float4*>(up + c);
        v.x = g_silu(fmaf(v.x, s_scale[c], s_shift[c]));
        v.y = g_silu(fmaf(v.y, s_scale[c + 1], s_shift[c + 1]));
        v.z = g_silu(fmaf(v.z, s_scale[c + 2], s_shift[c + 2]));
        v.w = g_silu(fmaf(v.w, s_scale[c + 3], s_shift[c + 3]));
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (co0 + k < Cout) {
            const float* wr = s_w + (co0 + k) * C + c;
            part[k] += v.x * wr[0] + v.y * wr[1] + v.z * wr[2] + v.w * wr[3];
          }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s = part[k];
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (on && q == 0 && co0 + k < Cout) out[((size_t)n * Cout + co0 + k) * HW + p0 + pp] = s + bias[co0 + k];
      }
    }
  }
}
