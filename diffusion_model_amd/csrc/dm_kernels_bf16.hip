// dm_kernels_bf16.hip — bf16 variant of the conv kernel (BASELINE configs[3]: bf16 activations
// and weights, fp32 accumulate and fp32 GroupNorm statistics/transform).  Same structure as
// conv_mfma_kernel (dm_kernels.hip) on v_mfma_f32_32x32x16_bf16, which is 16x the fp32 MFMA
// rate: the kernel is then bound by staging (HBM + the SiLU transform) and the epilogue, so
// activations are stored as bf16 NHWC (128 B per pixel at C = 64) to halve the HBM traffic.
//   A operand: lane l holds channels 16*kk + 8*(l>>5) .. +7 of pixel (l&31): one ds_read_b128.
//   B operand: packed [parity][tap][Cin/16][Cout/32][64 lanes][8 bf16]: one 1 KiB load per wave.
// KIND 0 / 1 / 2 as the fp32 kernel (2 = one output parity per workgroup), KC = 64.
#pragma once
#include "dm_kernels.hip"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int MB, int KIND>
__global__ __launch_bounds__(256, 3) void conv_bf16_kernel(const ConvArgs p) {
  constexpr int KC = 64;
  constexpr int MT = 64 * MB;
  constexpr int RSB = KC + 8;  // LDS row stride in bf16 (144 B: 16-B skew between pixels)
  constexpr int S = (KIND == 1) ? 2 : 1;
  constexpr int NTY = (KIND == 0) ? 3 : (KIND == 1) ? 4 : 2;
  constexpr int NT = NTY * NTY;
  constexpr int NJ = KC / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int Cin = p.C1 + p.C2;
  const int par = (KIND == 2) ? blockIdx.z : 0;
  const int py = par >> 1, px = par & 1;
  const int th = 1 << p.th_log, tw = 1 << p.tw_log;
  const int t = blockIdx.x;
  const int img0 = (t / p.tiles_per_img) * p.ni;
  const int tin = t % p.tiles_per_img;
  const int y0 = (tin / p.tiles_x) << p.th_log;
  const int x0 = (tin % p.tiles_x) << p.tw_log;
  const int hy0 = (KIND == 0) ? y0 - 1 : (KIND == 1) ? 2 * y0 - 1 : y0 - 1 + py;
  const int hx0 = (KIND == 0) ? x0 - 1 : (KIND == 1) ? 2 * x0 - 1 : x0 - 1 + px;
  const int hth = p.hth, htw = p.htw;
  const int NP = p.ni * hth * htw;

  const bool tf = p.stats_in != nullptr;
  float* s_mean = smem;
  float* s_scale = s_mean + (tf ? p.ni * p.C1 : 0);
  float* s_beta = s_scale + (tf ? p.ni * p.C1 : 0);
  float* s_temb = s_beta + (tf ? p.C1 : 0);
  bf16_t* tile = reinterpret_cast<bf16_t*>(s_temb + (tf ? p.ni * p.C1 : 0));

  int abase[MB];  // bf16 offsets
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    const int x = m & (tw - 1);
    const int y = (m >> p.tw_log) & (th - 1);
    const int i = m >> (p.tw_log + p.th_log);
    abase[mb] = ((i * hth + y * S) * htw + x * S) * RSB + 8 * h;
  }

  f32x16 acc[MB];
  const int nb32 = p.Cout >> 5;
  const size_t jstride = (size_t)nb32 * 512;               // bf16 elements per (16 ch, all couts)
  const size_t tapstride = (size_t)(Cin >> 4) * jstride;
  const int nchunks = Cin / KC;
  constexpr int V = KC / 8;
  constexpr int U = 6;
  const int items = NP * V;
  const bf16_t* wpk = reinterpret_cast<const bf16_t*>(p.wp);

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cg0 = chunk * KC;
    const bool from2 = cg0 >= p.C1;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(from2 ? p.x2 : p.x1);
    const int Cs = from2 ? p.C2 : p.C1;
    const int coff = from2 ? cg0 - p.C1 : cg0;
    const bf16_t* wbase = wpk + ((size_t)par * NT) * tapstride + (size_t)chunk * NJ * jstride +
                          (size_t)(blockIdx.y * 2 + wn) * 512 + lane * 8;
    if (chunk > 0) {
      __builtin_amdgcn_s_setprio(2);
      __syncthreads();
    }
    if (KIND == 0 && p.init_x) {
      stage_init_tile<true, RSB>(p, tile, reinterpret_cast<float*>(tile + ((NP * RSB + 7) & ~7)), cg0, img0, hy0, hx0, hth, htw, tid);
    } else
    for (int base = 0; base < items; base += 256 * U) {
      uint4 v[U];
      int meta[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256 + tid;
        const int pix = idx / V, c8 = idx - pix * V;
        const int r = __umulhi((unsigned)pix, p.mg_htw);
        const int hx = pix - r * htw;
        const int i = __umulhi((unsigned)r, p.mg_hth);
        const int hy = r - i * hth;
        const int iy = hy0 + hy, ix = hx0 + hx, n = img0 + i;
        meta[u] = (idx < items) ? pix * RSB + c8 * 8 : -1;
        const bool inb = idx < items && n < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int cn = min(n, p.B - 1), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
        const int cc = (idx < items) ? c8 * 8 : 0;
        v[u] = *reinterpret_cast<const uint4*>(src + (((size_t)cn * p.H + cy) * p.W + cx) * Cs + coff + cc);
        if (inb) meta[u] |= (i * p.C1 + coff + c8 * 8 + 1) << 16;
      }
      if (tf && chunk == 0 && base == 0) {
        const int G = p.C1 / DM_GROUPS;
        const int step = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;
        const float* te = p.temb ? p.temb + (size_t)step * p.temb_step_stride : nullptr;
        for (int idx = tid; idx < p.ni * p.C1; idx += 256) {
          const int i = idx / p.C1, c = idx - i * p.C1;
          const int n = img0 + i;
          float mean = 0.f, scale = 0.f, tv = 0.f;
          if (n < p.B) {
            const float* st = p.stats_in + ((size_t)n * p.pin * DM_GROUPS + (c / G)) * 2;
            float msum = 0.f, m2 = 0.f, dev = 0.f;
            if (p.pin <= 8) {  // all partials in one round of independent loads (same arithmetic)
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
            if (te) tv = te[(size_t)n * p.temb_stride + c];
          }
          s_mean[idx] = mean;
          s_scale[idx] = scale;
          s_temb[idx] = tv;
        }
        for (int c = tid; c < p.C1; c += 256) s_beta[c] = p.beta[c];
        __syncthreads();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (meta[u] >= 0) {
          uint4 w = v[u];
          const int lofs = meta[u] & 0xffff, tofs = (meta[u] >> 16) - 1;
          if (tofs < 0) w = make_uint4(0u, 0u, 0u, 0u);
          if (tf && tofs >= 0) {
            const int c = coff + (lofs % RSB);
            unsigned wi[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float lo = __uint_as_float(wi[q] << 16), hi = __uint_as_float(wi[q] & 0xffff0000u);
              const int e = 2 * q;
              lo = (lo - s_mean[tofs + e]) * s_scale[tofs + e] + s_beta[c + e];
              hi = (hi - s_mean[tofs + e + 1]) * s_scale[tofs + e + 1] + s_beta[c + e + 1];
              if (p.tf_silu) { lo = dm_silu(lo); hi = dm_silu(hi); }
              lo += s_temb[tofs + e];
              hi += s_temb[tofs + e + 1];
              wi[q] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
            }
            w = make_uint4(wi[0], wi[1], wi[2], wi[3]);
          }
          *reinterpret_cast<uint4*>(tile + lofs) = w;
        }
      }
    }
    uint4 bq[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bq[j] = *reinterpret_cast<const uint4*>(wbase + j * jstride);
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    if (chunk == 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    }
    uint4 a0[MB], a1[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a0[mb] = *reinterpret_cast<const uint4*>(tile + abase[mb]);
    int ty = 0, tx = 0;
    for (int tap = 0; tap < NT; ++tap) {
      const int tapoff = (ty * htw + tx) * RSB;
      int ntx = tx + 1, nty = ty;
      if (ntx == NTY) { ntx = 0; ++nty; }
      const bool last = tap + 1 == NT;
      const int ntapoff = last ? tapoff : (nty * htw + ntx) * RSB;
      const bf16_t* wnext = wbase + (size_t)(last ? tap : tap + 1) * tapstride;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        uint4* acur = (j & 1) ? a1 : a0;
        uint4* anxt = (j & 1) ? a0 : a1;
        const int noff = (j + 1 < NJ) ? tapoff + (j + 1) * 16 : ntapoff;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) anxt[mb] = *reinterpret_cast<const uint4*>(tile + abase[mb] + noff);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, bq[j]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, acur[mb]), b, acc[mb], 0, 0, 0);
        bq[j] = *reinterpret_cast<const uint4*>(wnext + j * jstride);
        __builtin_amdgcn_sched_barrier(0);
      }
      tx = ntx; ty = nty;
    }
  }
  __builtin_amdgcn_s_setprio(2);
  __syncthreads();
  conv_epilogue<MB, KIND, true>(p, acc, smem + MT * 36, smem, par, py, px, img0, tin, y0, x0, wm, wn, lane, h, tid, wave);
}

// bf16 B-fragment packing: out[par][tap][cin/16][cout/32][lane][8] =
//   bf16( w(cout = nb*32 + (lane&31), cin = 16*c16 + 8*(lane>>5) + e, ky, kx) )
__global__ void pack_conv_weight_bf16_kernel(int kind, const float* __restrict__ w, int cin, int cout,
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
  out[idx] = f2bf(w[src]);
}

// init_conv with bf16 NHWC output
__global__ __launch_bounds__(256) void init_conv_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                             int B, int Cin, int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;
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
    uint2 o;
    o.x = (unsigned)f2bf(a.x) | ((unsigned)f2bf(a.y) << 16);
    o.y = (unsigned)f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
    *reinterpret_cast<uint2*>(y + pix * C + c4 * 4) = o;
  }
}

// final_conv reading bf16 NHWC (GN + SiLU + 1x1 in fp32)
__global__ __launch_bounds__(256) void final_conv_bf16_kernel(const bf16_t* __restrict__ u, const float* __restrict__ stats,
                                                              int pin, float cnt_in, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int B, int HW, int C, int Cout, int slabs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_mean = smem;
  float* s_scale = s_mean + C;
  float* s_beta = s_scale + C;
  float* s_w = s_beta + C;
  const int blocks_img = HW / (64 * slabs);  // `slabs` 64-pixel slabs per block (see final_conv_kernel)
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
  const int LP = V < 64 ? V : 64;
  const int ppp = 256 / LP;
  const int q = threadIdx.x % LP;
  for (int sl = 0; sl < slabs; ++sl) {
  const int pix0 = pixb + sl * 64;
  for (int pp = threadIdx.x / LP; pp < 64; pp += ppp) {
    const bf16_t* up = u + ((size_t)n * HW + pix0 + pp) * C;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = q; c4 < V; c4 += LP) {
      const int c = c4 * 4;
      const uint2 raw = *reinterpret_cast<const uint2*>(up + c);
      float v[4] = {__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                    __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u)};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = dm_silu((v[e] - s_mean[c + e]) * s_scale[c + e] + s_beta[c + e]);
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < Cout) {
          const float* wv = s_w + co * C + c;
          part[co] += v[0] * wv[0] + v[1] * wv[1] + v[2] * wv[2] + v[3] * wv[3];
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co)
      if (co < Cout) {
        float sm = part[co];
        for (int o = LP >> 1; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        if (q == 0) out[((size_t)n * Cout + co) * HW + pix0 + pp] = sm + bias[co];
      }
  }
  }  // slabs
}
