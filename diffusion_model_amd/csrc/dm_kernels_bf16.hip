// dm_kernels_bf16.hip — bf16 variant of the conv kernel (BASELINE configs[3]: bf16 activations
// and weights, fp32 accumulate and fp32 GroupNorm statistics/transform).  16-bit storage between
// kernels: bf16 for tensors that feed an MFMA as they are, IEEE half for the raw conv outputs a
// consumer normalises first (see f2h in dm_kernels.hip); the MFMA operands are always bf16.  Same structure as
// conv_mfma_kernel (dm_kernels.hip) on v_mfma_f32_32x32x16_bf16, which is 16x the fp32 MFMA
// rate: the kernel is then bound by staging (HBM + the SiLU transform) and the epilogue, so
// activations are stored as bf16 NHWC (128 B per pixel at C = 64) to halve the HBM traffic.
//   A operand: lane l holds channels 16*kk + 8*(l>>5) .. +7 of pixel (l&31): one ds_read_b128.
//   B operand: packed [parity][tap][Cin/16][Cout/32][64 lanes][8 bf16]: one 1 KiB load per wave.
// KIND 0 / 1 / 2 as the fp32 kernel (2 = one output parity per workgroup), KC = 64.
#pragma once
#include "dm_kernels.hip"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// Weights are a PAIR of bf16 values per element, w = hi + lo (lo = bf16(w - hi), packed behind the hi
// block): two MFMAs per A fragment, no extra vector work.  bf16 weights alone cost ~0.9e-2 of the 2e-2
// per-step eps budget (as much as the bf16 activations); the pair removes that share at the price of MFMA
// slots, which this kernel has to spare (it is bound by staging latency and the SiLU transform's VALU).
//
// Workgroups are persistent (<= 3 per CU, XCD-aware tile order as conv_mfma_kernel).  Measured and NOT
// adopted: issuing the next tile's loads before the current K loop (two workgroups per CU for the registers it
// takes: 309 -> 326 us per 3x3 launch at B = 2048).  With three waves per SIMD the kernel is bound by instruction
// issue (SiLU transform + MFMA), not by load latency.
#ifndef DM_BF16_WG
#define DM_BF16_WG 3
#endif
#ifndef DM_BF16_SPLITW
#define DM_BF16_SPLITW 1
#endif
// F16 (layers behind a GroupNorm only: conv2, the stride-2 / transposed transforms of the blocks at the two highest resolutions):
// the MFMA operands are IEEE half instead of bf16 - SiLU(GroupNorm(.)) + temb is of order 1, where half's 11 bits cost what
// bf16's 8 bits cost and round 8x finer; the activation operand's bf16 rounding in down1 / up3 was the largest share of the
// path's eps error.  F16 = 1: weights as a (hi, lo) half pair, two MFMAs per product; F16 = 2: ONE half
// weight (rounded to nearest, 2^-12 relative: the same size as the activation's rounding), one v_mfma_f32_32x32x16_f16 per product.
typedef _Float16 bf_h2v __attribute__((ext_vector_type(2)));
typedef _Float16 bf_h8v __attribute__((ext_vector_type(8)));
typedef float bf_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pack_h2(float a, float b) {  // two floats -> packed half, to nearest even; clamped like f2h
  const bf_f2v v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf_h2v));
}
template <int MB, int KIND, bool INIT = false, int F16 = 0>  // INIT: down1.conv1 with init_conv folded into its staging
__global__ __launch_bounds__(256, DM_BF16_WG) void conv_bf16_kernel(const ConvArgs p) {
  constexpr int KC = 64;
  constexpr int MT = 64 * MB;
  constexpr int RSB = KC + 8;  // LDS row stride in bf16 (144 B: 16-B skew between pixels)
  constexpr int S = (KIND == 1) ? 2 : 1;
  constexpr int NTY = (KIND == 0) ? 3 : (KIND == 1) ? 4 : 2;
  constexpr int NT = NTY * NTY;
  constexpr int NTOT = (KIND == 4) ? 16 : NT;  // KIND 4: the four output parities of the transposed conv from ONE staged tile
  constexpr int NJ = KC / 16;
  constexpr int V = KC / 8;  // 16-byte loads per pixel and chunk
  constexpr int U = 6;       // loads per thread and staging batch (one batch covers a 10x18 halo tile)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int Cin = p.C1 + p.C2;
  int par = (KIND == 2) ? blockIdx.z : 0;  // KIND 4: advances inside the K loop
  int py = par >> 1, px = par & 1;
  const int th = 1 << p.th_log, tw = 1 << p.tw_log;
  const int hth = p.hth, htw = p.htw;
  const int NP = p.ni * hth * htw;
  const int items = NP * V;
  const int step_now = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;

  const bool tf = p.stats_in != nullptr;
  float* s_scale = smem;
  float* s_shift = s_scale + (tf ? p.ni * p.C1 : 0);
  float* s_temb = s_shift + (tf ? p.ni * p.C1 : 0);
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
  const size_t lo_off = (size_t)((KIND == 2 || KIND == 4) ? 4 * NT : NT) * tapstride;  // the lo halves follow all hi halves
  const int nchunks = Cin / KC;
  const bf16_t* wpk = reinterpret_cast<const bf16_t*>(p.wp);

  // tile index -> (first image, tile inside the image, output origin, halo origin)
  struct TileAt { int img0, tin, y0, x0, hy0, hx0; };
  auto tile_at = [&](int t) {
    TileAt a;
    a.img0 = (t >> p.tpi_log) * p.ni;
    a.tin = t & (p.tiles_per_img - 1);
    a.y0 = (a.tin >> p.tx_log) << p.th_log;
    a.x0 = (a.tin & (p.tiles_x - 1)) << p.tw_log;
    a.hy0 = (KIND == 0 || KIND == 4) ? a.y0 - 1 : (KIND == 1) ? 2 * a.y0 - 1 : a.y0 - 1 + py;
    a.hx0 = (KIND == 0 || KIND == 4) ? a.x0 - 1 : (KIND == 1) ? 2 * a.x0 - 1 : a.x0 - 1 + px;
    return a;
  };
  // Halo coordinates of this thread's first-batch loads do not depend on the tile: (image << 16 | hy << 8 | hx),
  // computed once per workgroup (two multiply-high divisions per load otherwise, at a quarter of the vector rate).
  int hyx0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int pix = (u * 256 + tid) >> 3;
    const int r = __umulhi((unsigned)pix, p.mg_htw);
    const int i = __umulhi((unsigned)r, p.mg_hth);
    hyx0[u] = (i << 16) | ((r - i * hth) << 8) | (pix - r * htw);
  }
  // Issue one batch of staging loads (unconditional, from clamped addresses; padding is decided by `inbm`).
  auto issue = [&](const TileAt& a, int chunk, int base, uint4 (&v)[U], unsigned& inbm) {
    const int cg0 = chunk * KC;
    const bool from2 = cg0 >= p.C1;
    const bf16_t* src = reinterpret_cast<const bf16_t*>(from2 ? p.x2 : p.x1);
    const int cs_log = from2 ? p.c2_log : p.c1_log;
    const int coff = from2 ? cg0 - p.C1 : cg0;
    const int imax = p.B - 1 - a.img0;
    int tq = threadIdx.x;  // opaque copy: the index math below is redone per call, not hoisted and kept live (spilled)
    asm volatile("" : "+v"(tq));
    // 64-bit base of this tile's first image + 32-bit offsets inside the tile's image group
    const int simg = (p.in_wrap && a.img0 >= p.in_wrap) ? a.img0 - p.in_wrap : a.img0;
    const bf16_t* srcb = src + ((size_t)simg << (p.h_log + p.w_log + cs_log)) + coff;
    inbm = 0;
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
      const int iy = a.hy0 + hy, ix = a.hx0 + hx;
      const bool inb = idx < items && i <= imax && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int cn = min(i, imax), cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      const unsigned off = ((unsigned)(((cn << p.h_log) + cy) << p.w_log) + cx) << cs_log;
      v[u] = *reinterpret_cast<const uint4*>(srcb + off + c8 * 8);
      inbm |= inb ? (1u << u) : 0u;
    }
  };
  // Transform (GroupNorm + SiLU + time embedding, fp32) and store one batch into the LDS tile as bf16.
  auto consume = [&](int chunk, int base, uint4 (&v)[U], unsigned inbm) {
    const int cg0 = chunk * KC;
    const int coff = (cg0 >= p.C1) ? cg0 - p.C1 : cg0;
    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256 + tq;
      if (idx < items) {
        const int pix = idx >> 3, c8 = idx & 7;
        uint4 w = v[u];
        if (!((inbm >> u) & 1)) {
          w = make_uint4(0u, 0u, 0u, 0u);  // zero padding / images past the batch stay exactly 0
        } else if (tf) {
          const int i = (p.ni == 1) ? 0 : (base == 0) ? (hyx0[u] >> 16) : (int)__umulhi(__umulhi((unsigned)pix, p.mg_htw), p.mg_hth);
          const int tofs = (i << p.c1_log) + coff + c8 * 8;
          unsigned wi[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int q = 0; q < 4; q += 2) {  // four elements per round: one 16-byte read of each table
            const float4 sc = *reinterpret_cast<const float4*>(s_scale + tofs + 2 * q);
            const float4 sh = *reinterpret_cast<const float4*>(s_shift + tofs + 2 * q);
            const float4 te = *reinterpret_cast<const float4*>(s_temb + tofs + 2 * q);
            // normalised inputs are stored as half (see f2h)
            float e0 = fmaf(h2f((bf16_t)(wi[q] & 0xffffu)), sc.x, sh.x), e1 = fmaf(h2f((bf16_t)(wi[q] >> 16)), sc.y, sh.y);
            float e2 = fmaf(h2f((bf16_t)(wi[q + 1] & 0xffffu)), sc.z, sh.z), e3 = fmaf(h2f((bf16_t)(wi[q + 1] >> 16)), sc.w, sh.w);
            if (p.tf_silu) {
              e0 = dm_silu_add(e0, te.x); e1 = dm_silu_add(e1, te.y); e2 = dm_silu_add(e2, te.z); e3 = dm_silu_add(e3, te.w);
            } else {
              e0 += te.x; e1 += te.y; e2 += te.z; e3 += te.w;
            }
            if constexpr (F16 != 0) {
              wi[q] = bf_pack_h2(e0, e1);
              wi[q + 1] = bf_pack_h2(e2, e3);
            } else {
              wi[q] = (unsigned)f2bf(e0) | ((unsigned)f2bf(e1) << 16);
              wi[q + 1] = (unsigned)f2bf(e2) | ((unsigned)f2bf(e3) << 16);
            }
          }
          w = make_uint4(wi[0], wi[1], wi[2], wi[3]);
        }
        *reinterpret_cast<uint4*>(tile + pix * RSB + c8 * 8) = w;
      }
      __builtin_amdgcn_sched_barrier(0);  // one load's chain at a time: the other two waves of the SIMD fill the gaps,
    }                                     // and six interleaved chains do not fit 168 registers
  };

  // persistent tile walk (see conv_mfma_kernel for the XCD-aware order)
  const int t_first = p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = p.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_xcd = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int t_end = p.xcd_map ? (((p.ngroups - t_xcd + 7) >> 3) << p.tpi_log) : p.ntiles;
  auto tile_of = [&](int ts) {
    return p.xcd_map ? (((((ts >> p.tpi_log) << 3) + t_xcd) << p.tpi_log) | (ts & (p.tiles_per_img - 1))) : ts;
  };
  for (int ts = t_first; ts < t_end; ts += t_step) {
    if (KIND == 4) { par = 0; py = 0; px = 0; }  // (advanced by the previous tile's K loop)
    const TileAt at = tile_at(tile_of(ts));
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bf16_t* wbase = wpk + ((size_t)par * NT) * tapstride + (size_t)chunk * NJ * jstride +
                            (size_t)(blockIdx.y * 2 + wn) * 512 + lane * 8;
      if (chunk > 0) {
        __builtin_amdgcn_s_setprio(2);
        __syncthreads();  // previous chunk's LDS reads are done
      }
      if constexpr (INIT) {
        stage_init_tile<true, RSB>(p, tile, reinterpret_cast<float*>(tile + ((NP * RSB + 7) & ~7)), chunk * KC, at.img0, at.hy0, at.hx0, hth, htw, tid);
      } else {
        for (int base = 0; base < items; base += 256 * U) {  // uniform trip count (barrier inside)
          uint4 v[U];
          unsigned inbm;
          issue(at, chunk, base, v, inbm);
          if (tf && chunk == 0 && base == 0) {
            gn_tables(p, step_now, at.img0, tid, s_scale, s_shift, s_temb);  // while the tile loads are in flight
            __syncthreads();
          }
          consume(chunk, base, v, inbm);
        }
      }
      uint4 bq[NJ], bl[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        bq[j] = *reinterpret_cast<const uint4*>(wbase + j * jstride);
        if constexpr (F16 != 2) bl[j] = *reinterpret_cast<const uint4*>(wbase + lo_off + j * jstride);
      }
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
        const bf16_t* wnext = wbase + (size_t)(last ? tap : tap + 1) * tapstride;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          uint4* acur = (j & 1) ? a1 : a0;
          uint4* anxt = (j & 1) ? a0 : a1;
          const int noff = (j + 1 < NJ) ? tapoff + (j + 1) * 16 : ntapoff;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) anxt[mb] = *reinterpret_cast<const uint4*>(tile + abase[mb] + noff);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (F16 != 0) {
            const bf_h8v bh = __builtin_bit_cast(bf_h8v, bq[j]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const bf_h8v av = __builtin_bit_cast(bf_h8v, acur[mb]);
              if constexpr (F16 == 1) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(bf_h8v, bl[j]), acc[mb], 0, 0, 0);
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bh, acc[mb], 0, 0, 0);
            }
          } else {
            const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, bq[j]);
            const bf16x8_t bw = __builtin_bit_cast(bf16x8_t, bl[j]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const bf16x8_t av = __builtin_bit_cast(bf16x8_t, acur[mb]);
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, acc[mb], 0, 0, 0);
              if (DM_BF16_SPLITW) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc[mb], 0, 0, 0);
            }
          }
          bq[j] = *reinterpret_cast<const uint4*>(wnext + j * jstride);
          if constexpr (F16 != 2) bl[j] = *reinterpret_cast<const uint4*>(wnext + lo_off + j * jstride);
          __builtin_amdgcn_sched_barrier(0);
        }
        tx = ntx; ty = nty;
        if constexpr (KIND == 4) if ((tap & 3) == 3) {
          // one output parity is complete: through its OWN staging region (the input tile stays for the next parity)
          par = tap >> 2; py = par >> 1; px = par & 1;
          float* ot = reinterpret_cast<float*>(tile + ((NP * RSB + 7) & ~7));
          conv_epilogue<MB, KIND, true>(p, acc, ot + MT * 36, ot, par, py, px, at.img0, at.tin, at.y0, at.x0, wm, wn, lane, h, tid, wave);
          __syncthreads();  // the staging region is read out before the next parity writes it
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
        }
      }
    }
    if constexpr (KIND != 4) {
      __builtin_amdgcn_s_setprio(2);
      __syncthreads();  // every wave is done reading the input tile: LDS is reused from here
      conv_epilogue<MB, KIND, true>(p, acc, smem + MT * 36, smem, par, py, px, at.img0, at.tin, at.y0, at.x0, wm, wn, lane, h, tid, wave);
    }
    __syncthreads();  // output staging / reduction words (KIND 4: the input tile) are free again for the next tile
  }
}

// bf16 B-fragment packing, each weight as a (hi, lo) bf16 pair with hi + lo ~ w to 16 mantissa bits:
//   out[par][tap][cin/16][cout/32][lane][8] = hi = bf16( w(cout = nb*32 + (lane&31), cin = 16*c16 + 8*(lane>>5) + e, ky, kx) )
//   out[total + the same index]             = lo = bf16( w - hi )
// half != 0 (conv_bf16_kernel's F16 layers): the pair as IEEE half, hi to nearest - alone it is the single-weight form.
__global__ void pack_conv_weight_bf16_kernel(int kind, const float* __restrict__ w, int cin, int cout,
                                             bf16_t* __restrict__ out, size_t total, int half) {
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
  const float wv = w[src];
  if (half) {
    const bf16_t hi = f2h(wv);
    out[idx] = hi;
    out[total + idx] = f2h(wv - h2f(hi));
    return;
  }
  const bf16_t hi = f2bf(wv);
  out[idx] = hi;
  out[total + idx] = f2bf(wv - bf2f(hi));
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
  if (C == 64) {
    // C = 64 (as final_conv_kernel): 16 lanes per pixel, 4 pixels per thread and slab, a slab's four loads issued together
    // and the next slab's before the current one is reduced; table entries and weights in registers
    const int c = q * 4, p0 = threadIdx.x >> 4;
    const float4 mu = *reinterpret_cast<const float4*>(s_mean + c);
    const float4 sc = *reinterpret_cast<const float4*>(s_scale + c);
    const float4 be = *reinterpret_cast<const float4*>(s_beta + c);
    float4 wv[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) wv[co] = (co < Cout) ? *reinterpret_cast<const float4*>(s_w + co * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bf16_t* ub = u + ((size_t)n * HW + pixb + p0) * C + c;
    uint2 cur[4], nxt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = *reinterpret_cast<const uint2*>(ub + (size_t)(16 * k) * C);
    for (int sl = 0; sl < slabs; ++sl) {
      const int sn = (sl + 1 < slabs) ? sl + 1 : sl;  // (clamped: the last prefetch re-reads valid data, unused)
#pragma unroll
      for (int k = 0; k < 4; ++k) nxt[k] = *reinterpret_cast<const uint2*>(ub + (size_t)(sn * 64 + 16 * k) * C);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 v;  // u3 is stored as half (normalised here)
        v.x = dm_silu((h2f((bf16_t)(cur[k].x & 0xffffu)) - mu.x) * sc.x + be.x);
        v.y = dm_silu((h2f((bf16_t)(cur[k].x >> 16)) - mu.y) * sc.y + be.y);
        v.z = dm_silu((h2f((bf16_t)(cur[k].y & 0xffffu)) - mu.z) * sc.z + be.z);
        v.w = dm_silu((h2f((bf16_t)(cur[k].y >> 16)) - mu.w) * sc.w + be.w);
#pragma unroll
        for (int co = 0; co < 4; ++co) {
          if (co < Cout) {
            float sm = v.x * wv[co].x + v.y * wv[co].y + v.z * wv[co].z + v.w * wv[co].w;
            sm = dm_row16_sum(sm);  // the pixel's 16 lanes are one DPP row
            if (q == 0) out[((size_t)n * Cout + co) * HW + pixb + sl * 64 + p0 + 16 * k] = sm + bias[co];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
    return;
  }
  for (int sl = 0; sl < slabs; ++sl) {
  const int pix0 = pixb + sl * 64;
  for (int pp = threadIdx.x / LP; pp < 64; pp += ppp) {
    const bf16_t* up = u + ((size_t)n * HW + pix0 + pp) * C;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c4 = q; c4 < V; c4 += LP) {
      const int c = c4 * 4;
      const uint2 raw = *reinterpret_cast<const uint2*>(up + c);
      float v[4] = {h2f((bf16_t)(raw.x & 0xffffu)), h2f((bf16_t)(raw.x >> 16)),
                    h2f((bf16_t)(raw.y & 0xffffu)), h2f((bf16_t)(raw.y >> 16))};  // u3 is stored as half (normalised here)
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
