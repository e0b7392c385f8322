// dm_kernels_x6c3.hip — conv_x6_c3_kernel: Conv2d 3x3 pad 1 (model.py:42,45,47) on 8x16-pixel tiles with every operand as two
// fp16 terms (three v_mfma_f32_32x32x16_f16 per product, dm_kernels_x6.hip) and its STAGING SOFTWARE-PIPELINED INTO THE K LOOP:
// the wave that issues the MFMAs of K chunk g of the chunk stream also transforms, splits and writes chunk g+1 into the other
// half of a double-buffered LDS tile, a few vector instructions behind each MFMA (they ride in the matrix pipe's shadow), and
// issues the global loads of chunk g+3.  In conv_x6_kernel the staging of one workgroup runs beside the K loop of the other
// workgroup of its CU and is starved by it (an MFMA stream starves the vector instructions of the OTHER
// waves of its SIMD, its own cost it almost nothing): 16.6 k of the 36 k cycles of a tile.
//
// A tile has only Cin/32 = 2 chunks (4 with the channel concat), so the stream runs ACROSS tiles: the next tile's first chunk
// is staged under this tile's last K loop, its loads were issued one tile earlier, and what that staging needs of the next
// image - the GroupNorm tables (TF: conv2 reads SiLU(GroupNorm(conv1)) + temb) or the power-of-two scale from the producers'
// maxima (!TF: conv1 reads its inputs raw) - is fetched at the start of the K loop before and finished at its end.
//   chunk  = 32 input channels of the 10x18 halo tile, 9 taps x 2 x 16 channels, 108 MFMAs per wave
//   LDS    = 2 x GroupNorm tables | reduction words | 2 buffers x 2 planes x [halo pixel][32 + 8] fp16   (57.6 KB + tables)
// The K loop is kept lean in scalar registers and vector instructions (readlane/writelane traffic of spilled scalars cost the
// first version 3 vector instructions per MFMA): image side HL = log2(H) = log2(W) at compile time, one scalar descriptor per
// loaded chunk (base pointer, byte offsets of the tile's halo origin and of its first pixel, border flags), per-piece constants
// (byte offset inside the halo, border code) in one vector register each, weight addresses as a scalar pointer per tap plus
// four fixed lane offsets.
#pragma once
#include "dm_kernels_x6.hip"

// compile-time loop: f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{}) (the unroller gives up on bodies this size)
template <class F, size_t... I>
__device__ __forceinline__ void dm_static_for_impl(F&& f, std::index_sequence<I...>) { (f(std::integral_constant<int, (int)I>{}), ...); }
#ifndef DM_C3_EXP
#define DM_C3_EXP 0
#endif
template <int N, class F>
__device__ __forceinline__ void dm_static_for(F&& f) { dm_static_for_impl(f, std::make_index_sequence<N>{}); }

template <bool TF, int HL>
__global__ __launch_bounds__(256, 2) void conv_x6_c3_kernel(const ConvArgs p) {
  constexpr int MB = 2, NS = 2, KC = 32, RSB = KC + 8;
  constexpr int H = 1 << HL, W = 1 << HL;                 // square power-of-two images (host-checked)
  constexpr int tw = 16, htw = tw + 2, hth = 10;          // tile = 8 x 16 output pixels, halo 10 x 18
  constexpr int NP = hth * htw, items = NP * 8;           // 1440 pieces of 4 channels per chunk
  constexpr int U = 6;                                    // pieces per thread and chunk (1536 slots)
  constexpr int NSTEP = 10;
  constexpr int NMFMA = 9 * 2 * 3 * MB;                   // 108
  constexpr int PL = NP * RSB, BUF = NS * PL;
  constexpr int TX_LOG = HL - 4, TPI_LOG = 2 * HL - 7;    // tiles per row / per image (logs)
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef DM_TIMELINE
  int tl_n = 0;
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 128] = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int C1 = p.C1;
  const int cl2 = p.c1_log + 2;                 // log2 of a pixel's bytes (both inputs: C1 == C2, host-checked)
  float* tabs = smem;                           // tables [2 tile parities][scale | shift | temb][C1]
  float* scr = tabs + (TF ? 6 * C1 : 0);        // 64 reduction words (x6_epilogue_sw)
  bf16_t* tile = reinterpret_cast<bf16_t*>(scr + 64);

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    abase[mb] = ((m >> 4) * htw + (m & (tw - 1))) * RSB + 8 * h;
  }
  // per-piece constants: byte offset of the piece inside the halo tile (bits 0..26) | border code (bit 27: no such piece,
  // 28: halo row 0, 29: halo row 9, 30: halo column 0, 31: halo column 17)
  unsigned pcode[U];
  const int ldso0 = (tid >> 3) * RSB + (tid & 7) * 4;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = u * 256 + tid;
    const int pix = idx >> 3;
    const int hy = pix / htw, hx = pix - hy * htw;
    const unsigned off = ((unsigned)(hy * W + hx) << cl2) + (unsigned)(tid & 7) * 16u;
    pcode[u] = (idx < items) ? (off | (hy == 0 ? 1u << 28 : 0u) | (hy == hth - 1 ? 1u << 29 : 0u) | (hx == 0 ? 1u << 30 : 0u) | (hx == htw - 1 ? 1u << 31 : 0u))
                             : (1u << 27);
  }
  const int ldso_last = (pcode[U - 1] & (1u << 27)) ? KC : ldso0 + (U - 1) * 32 * RSB;   // (KC: the 8 unused fp16 behind a pixel's 32)
  const int c4o = (tid & 7) * 4;

  const int nchunks = (p.C1 + p.C2) >> 5;       // even: channel counts are multiples of 64
  const unsigned jstride = (unsigned)(p.Cout >> 5) * 1024u;         // bytes per (16 input channels, all output channels)
  const unsigned tapstride = (unsigned)((p.C1 + p.C2) >> 4) * jstride;
  const unsigned split_off = 9u * tapstride;
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  const char* wcol = reinterpret_cast<const char*>(p.wp) + (size_t)(blockIdx.y * 2 + wn_s) * 1024;
  unsigned wv[2][NS];   // lane offsets of the four fragments of a tap: [j][plane]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < NS; ++s) wv[j][s] = (unsigned)lane * 16u + j * jstride + s * split_off;
  auto wtap = [&](int q, int tp) { return wcol + ((unsigned)tp * tapstride + (unsigned)(2 * q) * jstride); };   // scalar
  auto wload = [&](const char* wp_, unsigned voff) { return *reinterpret_cast<const uint4*>(wp_ + voff); };
  const float wscale = p.wp[split_off / 2];     // the float 2^-k behind the two fp16 planes

  const int t_first = p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = p.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_xcd = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int t_end = p.xcd_map ? (((p.ngroups - t_xcd + 7) >> 3) << TPI_LOG) : p.ntiles;
  auto tile_id = [&](int ts) { return p.xcd_map ? (((((ts >> TPI_LOG) << 3) + t_xcd) << TPI_LOG) | (ts & ((1 << TPI_LOG) - 1))) : ts; };

  // scalar descriptor of the chunk whose loads are being issued
  struct LD { const char* ptr; unsigned lbase, cbase, flags; };
  auto describe = [&](int ts, int q) {
    const int t = tile_id(ts);
    const int img = t >> TPI_LOG, tin = t & ((1 << TPI_LOG) - 1);
    const int y0 = (tin >> TX_LOG) << 3, x0 = (tin & ((1 << TX_LOG) - 1)) << 4;
    const int cg0 = q * KC;
    const bool from2 = cg0 >= C1;
    LD d;
    d.ptr = reinterpret_cast<const char*>(from2 ? p.x2 : p.x1) + ((size_t)img << (2 * HL + cl2));   // the image (64-bit; offsets inside it: 32 bits)
    const unsigned chb = (unsigned)(from2 ? cg0 - C1 : cg0) * 4u;
    d.cbase = ((unsigned)(y0 * W + x0) << cl2) + chb;                            // the tile's first pixel: always inside
    d.lbase = d.cbase - ((unsigned)(W + 1) << cl2);                              // its halo origin (y0 - 1, x0 - 1): may wrap
    d.flags = (1u << 27) | (y0 == 0 ? 1u << 28 : 0u) | (y0 + 8 == H ? 1u << 29 : 0u) | (x0 == 0 ? 1u << 30 : 0u) | (x0 + 16 == W ? 1u << 31 : 0u);
    return d;
  };

  float4 RA[U], RB[U];         // raw pieces of the chunks in flight: even chunks in RA, odd chunks in RB
  unsigned mA = 0, mB = 0;     // bit u: piece u of the set is inside the image
  auto issue = [&](float4& dst, unsigned& msk, int u, const LD& d) {
    const bool ok = (pcode[u] & d.flags) == 0u;
    const unsigned off = ok ? (pcode[u] & 0x07ffffffu) + d.lbase : d.cbase + (unsigned)c4o * 4u;   // (outside: any valid address)
    dst = *reinterpret_cast<const float4*>(d.ptr + off);
    msk = ok ? (msk | (1u << u)) : (msk & ~(1u << u));
  };

  // ---- staging of one piece in NSTEP steps
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
  dm_f2 ex = {0.f, 0.f};
  unsigned q1a = 0, q2a = 0, q1b = 0, q2b = 0;
  unsigned okm = 0u;
  float4 tte = make_float4(0.f, 0.f, 0.f, 0.f), tsc = tte, tsh = tte;   // table rows (TF) of this thread's 4 channels of the chunk being staged
  float sc_cur = 1.f;              // power-of-two scale (!TF) of the chunk being staged
  const dm_f2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one2 = {1.f, 1.f};
  auto stage_step = [&](float4 (&R)[U], unsigned& msk, auto uc, auto sc_, bf16_t* dstb, const LD& dl, bool do_load) {
    constexpr int u = decltype(uc)::value, s = decltype(sc_)::value;
    if constexpr (s == 0) {
      okm = 0u - ((msk >> u) & 1u);
      const float4 v = R[u];
      if (TF) { w0 = fmaf(v.x, tsc.x, tsh.x); w1 = fmaf(v.y, tsc.y, tsh.y); w2 = fmaf(v.z, tsc.z, tsh.z); w3 = fmaf(v.w, tsc.w, tsh.w); }
      else { w0 = v.x * sc_cur; w1 = v.y * sc_cur; w2 = v.z * sc_cur; w3 = v.w * sc_cur; }
    } else if constexpr (s == 1) {   // (the refill load of R[u] sat here; now one batch per chunk, see run_chunk)
    } else if constexpr (s == 2) {
      if (TF) { const dm_f2 a = dm_f2{w0, w1} * nl2e; ex = dm_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
    } else if constexpr (s == 3) {
      if (TF) {
        const dm_f2 d = ex + one2;
        const dm_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const dm_f2 o = __builtin_elementwise_fma(dm_f2{w0, w1}, r, dm_f2{tte.x, tte.y});
        w0 = o.x; w1 = o.y;
      }
    } else if constexpr (s == 4) {
      if (TF) { const dm_f2 a = dm_f2{w2, w3} * nl2e; ex = dm_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
    } else if constexpr (s == 5) {
      if (TF) {
        const dm_f2 d = ex + one2;
        const dm_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const dm_f2 o = __builtin_elementwise_fma(dm_f2{w2, w3}, r, dm_f2{tte.z, tte.w});
        w2 = o.x; w3 = o.y;
      }
    } else if constexpr (s == 6) {  // zero padding stays exactly 0
      w0 = __uint_as_float(__float_as_uint(w0) & okm); w1 = __uint_as_float(__float_as_uint(w1) & okm);
      w2 = __uint_as_float(__float_as_uint(w2) & okm); w3 = __uint_as_float(__float_as_uint(w3) & okm);
    } else if constexpr (s == 7) dm_split_h2(w0, w1, q1a, q2a);
    else if constexpr (s == 8) dm_split_h2(w2, w3, q1b, q2b);
    else {
      // (no branch: behind a condition the compiler sinks the whole piece's arithmetic into the conditional block, one serial blob
      //  behind one MFMA; a piece past the halo - last u only - goes to the unused tail of pixel row 0)
      bf16_t* q = dstb + (u == U - 1 ? ldso_last : ldso0 + u * 32 * RSB);
      *reinterpret_cast<uint2*>(q) = make_uint2(q1a, q1b);
      *reinterpret_cast<uint2*>(q + PL) = make_uint2(q2a, q2b);
    }
  };
  auto chunk_setup = [&](int q, int ts, float sc) {
    if (TF) {
      const float* tb = tabs + ts * 3 * C1 + q * KC + c4o;
      tsc = *reinterpret_cast<const float4*>(tb);
      tsh = *reinterpret_cast<const float4*>(tb + C1);
      tte = *reinterpret_cast<const float4*>(tb + 2 * C1);
    } else sc_cur = sc;
  };

  // ---- what the staging of an image needs, in two halves (fetch early, finish late): TF: GroupNorm table entries of channel tid
  //      (pin <= 8 partials: one round of independent loads); !TF: the producers' maxima (namax <= 64 per input)
  const int step_now = (TF && p.step_ptr ? *p.step_ptr : 0) + p.step_off;   // (read once: inside the K loop its wait would drain every load in flight)
  float2 pr[TF ? 8 : 1];
  float pgam = 0.f, pbet = 0.f, ptmb = 0.f, pmx = 0.f;
  auto next_fetch = [&](int ts) {
    const int img = tile_id(ts) >> TPI_LOG;
    if (TF) {
      if (tid < C1) {
        const float* st = p.stats_in + ((size_t)img * p.pin * DM_GROUPS + (tid >> p.g1_log)) * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) pr[TF ? q : 0] = *reinterpret_cast<const float2*>(st + (q < p.pin ? q : 0) * DM_GROUPS * 2);
        pgam = p.gamma[tid]; pbet = p.beta[tid];
        ptmb = p.temb ? p.temb[(size_t)step_now * p.temb_step_stride + (size_t)img * p.temb_stride + tid] : 0.f;
      }
    } else {
      pmx = lane < p.namax1 ? p.amax1[(size_t)img * p.namax1 + lane] : 0.f;
      if (p.amax2 && lane < p.namax2) pmx = fmaxf(pmx, p.amax2[(size_t)img * p.namax2 + lane]);
    }
  };
  auto next_finish = [&](int ts, float& sc, float& inv) {
    if (TF) {
      if (tid < C1) {  // gn_tables' arithmetic (Chan merge of the partials in a fixed order)
        float msum = 0.f, m2 = 0.f, dev = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < p.pin) msum += pr[TF ? q : 0].x;
        const float mean = msum / (float)p.pin;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < p.pin) {
          const float d = pr[TF ? q : 0].x - mean;
          m2 += pr[TF ? q : 0].y;
          dev += d * d;
        }
        const float var = (m2 + p.cnt_in * dev) / (p.cnt_in * (float)p.pin);
        const float scale = rsqrtf(var + DM_GN_EPS) * pgam;
        float* tb = tabs + ts * 3 * C1;
        tb[tid] = scale;
        tb[C1 + tid] = pbet - mean * scale;
        tb[2 * C1 + tid] = ptmb;
      }
      sc = 1.f; inv = 1.f;
    } else {
      float mx = pmx;
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      int eb = (int)((__float_as_uint(mx) >> 23) & 0xffu);
      eb = min(max(eb, 20), 240);
      inv = __uint_as_float((unsigned)(eb - 10) << 23);
      sc = __uint_as_float((unsigned)(264 - eb) << 23);
    }
  };

  f32x16 acc[MB];
  // weight fragments of the next TWO taps ([tap buffer][j][plane]): a fragment is requested four (tap, j) groups = 24 MFMAs before
  // its first use (two groups of six MFMAs do not cover an L2 hit).  Nine taps per chunk: the buffer parity flips with the chunk.
  uint4 bq[2][2][NS];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < NS; ++s) bq[tb][j][s] = wload(wtap(0, tb), wv[j][s]);

  // K loop of chunk q (buffer PB = q & 1, compile time) while the stream's next chunk is staged from R into the other buffer;
  // dl: descriptor of the chunk whose loads refill R (do_load).  pre_ts >= 0: this K loop also prepares the tables / scale of the
  // image of tile pre_ts (fetch at its start, finish at its end) into table set ts_pre.
  auto run_chunk = [&](auto pbc, float4 (&R)[U], unsigned& msk, int q, bool do_stage, const LD& dl, bool do_load,
                       int pre_ts, int ts_pre, float& sc_pre, float& inv_pre) {
    constexpr int PB = decltype(pbc)::value;
    const bf16_t* kb = tile + PB * BUF;
    bf16_t* sb = tile + (PB ^ 1) * BUF;
    uint4 a0[NS][MB], a1[NS][MB];
    int ap[NS][MB];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        ap[s][mb] = s * PL + abase[mb];
        a0[s][mb] = *reinterpret_cast<const uint4*>(kb + ap[s][mb]);
      }
    if (pre_ts >= 0) next_fetch(pre_ts);
    const int qn = q + 1 < nchunks ? q + 1 : 0;
    DM_TL(2);
    dm_static_for<NMFMA>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int mb = i % MB, t = (i / MB) % 3, j = (i / (3 * MB)) % 2, tp = i / (6 * MB);
      constexpr int tapoff = ((tp / 3) * htw + tp % 3) * RSB;
      constexpr int ntapoff = (((tp + 1) / 3) * htw + (tp + 1) % 3) * RSB;
      constexpr int tb = (tp + PB) & 1;       // tap buffer (PB: that of this chunk's tap 0)
      constexpr bool odd = ((tp * 2 + j) & 1) != 0;
      constexpr bool more = !(tp == 8 && j == 1);
      constexpr int noff = (j == 0) ? tapoff + 16 : ntapoff;
      constexpr int as = (t == 1) ? 1 : 0;    // products h1 g2, h2 g1, h1 g1
      uint4 (&acur)[NS][MB] = odd ? a1 : a0;
      uint4 (&anxt)[NS][MB] = odd ? a0 : a1;
      if constexpr (t == 0 && mb == 0) __builtin_amdgcn_sched_barrier(0);
      acc[mb] = dm_mfma16<true>(t == 0 ? bq[tb][j][1] : bq[tb][j][0], acur[as][mb], acc[mb]);
#if !(DM_C3_EXP & 1)   // timing experiment (tools/diag/r3_c3_exp.sh; wrong results): bit 0 = no pixel-operand re-reads from LDS,
      if constexpr (t < NS && more) anxt[t][mb] = *reinterpret_cast<const uint4*>(kb + ap[t][mb] + noff);
#endif                 //                                                                  bit 1 = no weight-fragment loads in the K loop
      if constexpr (!(DM_C3_EXP & 2) && mb == MB - 1 && (t == 1 || t == 2)) {
        const char* wnext = (tp < 7) ? wtap(q, tp + 2) : wtap(qn, tp - 7);   // two taps ahead (scalar)
        if constexpr (t == 1) bq[tb][j][1] = wload(wnext, wv[j][1]);
        else bq[tb][j][0] = wload(wnext, wv[j][0]);
      }
      constexpr int TOT = U * NSTEP;
      constexpr int k0 = i * TOT / NMFMA, k1 = (i + 1) * TOT / NMFMA;   // at most one staging step behind an MFMA (TOT <= NMFMA)
      static_assert(k1 - k0 <= 1, "one staging step per MFMA");
      if constexpr (k1 > k0) {
        if (do_stage) stage_step(R, msk, std::integral_constant<int, k0 / NSTEP>{}, std::integral_constant<int, k0 % NSTEP>{}, sb, dl, do_load);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // the refill of R in ONE batch behind the chunk's last staging step: loads return in order, so a wait for a weight fragment
    // requested after a refill load also waits for that load (HBM latency, not an L2 hit); one batch per chunk is covered by the
    // fragments' 24-MFMA lead, a refill load behind every piece was not (3x3 layers 128.6 -> 125.0 us)
    if (do_stage && do_load) {
#pragma unroll
      for (int u = 0; u < U; ++u) issue(R[u], msk, u, dl);
    }
    DM_TL(3);
    if (pre_ts >= 0) next_finish(ts_pre, sc_pre, inv_pre);
  };

  if (t_first >= t_end) return;
  float sc0 = 1.f, inv0 = 1.f, sc1 = 1.f, inv1 = 1.f;   // power-of-two scales of this / the next tile's image (!TF)
  int tpar = 0;                                        // table set of this tile
  // prologue (once per workgroup): the first tile's tables / scale and chunk 0 in the open
  next_fetch(t_first);
  {
    const LD d0 = describe(t_first, 0), d1 = describe(t_first, 1);
#pragma unroll
    for (int u = 0; u < U; ++u) issue(RA[u], mA, u, d0);
#pragma unroll
    for (int u = 0; u < U; ++u) issue(RB[u], mB, u, d1);
  }
  next_finish(0, sc0, inv0);
  __syncthreads();
  chunk_setup(0, 0, sc0);
  {
    const bool two = nchunks > 2;   // loads of stream chunk 2: this tile's chunk 2, else the next tile's chunk 0
    const bool h1 = t_first + t_step < t_end;
    const LD d2 = describe(two || !h1 ? t_first : t_first + t_step, two ? 2 : 0);
    dm_static_for<U * NSTEP>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      stage_step(RA, mA, std::integral_constant<int, k / NSTEP>{}, std::integral_constant<int, k % NSTEP>{}, tile, d2, two || h1);
    });
    if (two || h1) {
#pragma unroll
      for (int u = 0; u < U; ++u) issue(RA[u], mA, u, d2);
    }
  }
  __syncthreads();

  for (int ts = t_first; ts < t_end; ts += t_step) {
    const bool has1 = ts + t_step < t_end, has2 = ts + 2 * t_step < t_end;
    DM_TL(1);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll 1
    for (int q = 0; q < nchunks; q += 2) {
      // stream positions relative to this tile: staged q + 1 (, q + 2), loaded q + 3 (, q + 4); tile = position / nchunks
      {
        const int pl = q + 3;
        const int tl = pl >= 2 * nchunks ? 2 : pl >= nchunks ? 1 : 0;
        const bool dl = tl == 0 || (tl == 1 ? has1 : has2);
        const LD d = describe(dl ? ts + tl * t_step : ts, pl - tl * nchunks);
        chunk_setup(q + 1, tpar, sc0);
        run_chunk(std::integral_constant<int, 0>{}, RB, mB, q, true, d, dl, (has1 && q == nchunks - 2) ? ts + t_step : -1, tpar ^ 1, sc1, inv1);
      }
      __syncthreads();
      {
        const int ps = q + 2, pl = q + 4;
        const int tsg = ps >= nchunks ? 1 : 0;
        const int tl = pl >= 2 * nchunks ? 2 : pl >= nchunks ? 1 : 0;
        const bool dstg = tsg == 0 || has1;
        const bool dl = dstg && (tl == 0 || (tl == 1 ? has1 : has2));
        const LD d = describe(dl ? ts + tl * t_step : ts, pl - tl * nchunks);
        if (tsg) chunk_setup(0, tpar ^ 1, sc1);
        else chunk_setup(ps, tpar, sc0);
        float dum0, dum1;
        run_chunk(std::integral_constant<int, 1>{}, RA, mA, q + 1, dstg, d, dl, -1, 0, dum0, dum1);
      }
      __syncthreads();
    }
    {
      const int t = tile_id(ts);
      const int img = t >> TPI_LOG, tin = t & ((1 << TPI_LOG) - 1);
      x6_epilogue_sw<MB, 0>(p, acc, scr, 0, 0, 0, img, tin, (tin >> TX_LOG) << 3, (tin & ((1 << TX_LOG) - 1)) << 4, wm, wn, lane, h, tid, wave, wscale * inv0);
    }
    DM_TL(4);
    sc0 = sc1; inv0 = inv1;
    tpar ^= 1;
  }
}
