// dm_kernels_x6c3.hip — conv_x6_c3_kernel: Conv2d 3x3 pad 1 (model.py:42,45,47) on 8x16-pixel tiles with every operand as two
// fp16 terms (three v_mfma_f32_32x32x16_f16 per product, dm_kernels_x6.hip) and its STAGING SOFTWARE-PIPELINED INTO THE K LOOP,
// as conv_x6_down_kernel (dm_kernels_x6p.hip) does for the stride-2 conv: the wave that issues the MFMAs of K chunk g of the
// chunk stream also transforms, splits and writes chunk g+1 into the other half of a double-buffered LDS tile, a few vector
// instructions behind each MFMA (they ride in the matrix pipe's shadow), and issues the global loads of chunk g+3.
//
// A tile has only Cin/32 = 2 chunks (4 with the channel concat), so the stream runs ACROSS tiles: the next tile's first chunk
// is staged under this tile's last K loop, its loads were issued one tile earlier, and what that staging needs of the next
// image - the GroupNorm tables (TF: conv2 reads SiLU(GroupNorm(conv1)) + temb) or the power-of-two scale from the producers'
// maxima (!TF: conv1 reads its inputs raw) - is fetched at the start of the K loop before and finished at its end.
//   chunk  = 32 input channels of the 10x18 halo tile, 9 taps x 2 x 16 channels, 108 MFMAs per wave
//   LDS    = 2 x GroupNorm tables | reduction words | 2 buffers x 2 planes x [halo pixel][32 + 8] fp16   (57.6 KB + tables)
// Same arithmetic and K order (chunk, tap, 16 channels, term) per output as conv_x6_kernel<2, 0, true, 2, true> up to the
// order of the 32-channel chunks inside a tap - not bit-identical to it; chosen by the layer geometry only.
#pragma once
#include "dm_kernels_x6.hip"

template <bool TF>
__global__ __launch_bounds__(256, 2) void conv_x6_c3_kernel(const ConvArgs p) {
  constexpr int MB = 2, NS = 2, KC = 32, RSB = KC + 8;
  constexpr int tw = 16, htw = tw + 2, hth = 10;          // tile = 8 x 16 output pixels (host-checked), halo 10 x 18
  constexpr int NP = hth * htw, items = NP * 8;           // 1440 pieces of 4 channels per chunk
  constexpr int U = 6;                                    // pieces per thread and chunk (1536 slots)
  constexpr int NSTEP = 12;
  constexpr int NMFMA = 9 * 2 * 3 * MB;                   // 108
  constexpr int PL = NP * RSB, BUF = NS * PL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int C1 = p.C1, Cin = p.C1 + p.C2;
  const int step_now = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;
  // tables [2 tile parities][scale | shift | temb][C1]
  float* tabs = smem;
  float* scr = tabs + (TF ? 6 * C1 : 0);        // 64 reduction words (x6_epilogue_sw)
  bf16_t* tile = reinterpret_cast<bf16_t*>(scr + 64);

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    abase[mb] = ((m >> 4) * htw + (m & (tw - 1))) * RSB + 8 * h;
  }
  int hyx[U];
  const int ldso0 = (tid >> 3) * RSB + (tid & 7) * 4;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = u * 256 + tid;
    const int pix = idx >> 3;
    const int r = pix / htw;
    hyx[u] = (idx < items) ? ((r << 8) | (pix - r * htw)) : -1;
  }
  const int c4o = (tid & 7) * 4;

  const int nchunks = Cin >> 5;                 // even: channel counts are multiples of 64
  const int nb32 = p.Cout >> 5;
  const unsigned jstride = (unsigned)nb32 * 1024u;                  // bytes per (16 input channels, all output channels)
  const unsigned tapstride = (unsigned)(Cin >> 4) * jstride;
  const unsigned split_off = 9u * tapstride;
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  const char* wcol = reinterpret_cast<const char*>(p.wp) + (size_t)(blockIdx.y * 2 + wn_s) * 1024;
  const unsigned lane16 = (unsigned)lane * 16u;
  auto wtap = [&](int q, int tp) { return (unsigned)tp * tapstride + (unsigned)(2 * q) * jstride; };
  auto wload = [&](unsigned off) { return *reinterpret_cast<const uint4*>(wcol + (off + lane16)); };
  const float wscale = p.wp[split_off / 2];     // the float 2^-k behind the two fp16 planes

  struct TG { int y0, x0, tin, img; };
  const int t_first = p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = p.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_xcd = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int t_end = p.xcd_map ? (((p.ngroups - t_xcd + 7) >> 3) << p.tpi_log) : p.ntiles;
  auto tile_of = [&](int ts) {
    const int t = p.xcd_map ? (((((ts >> p.tpi_log) << 3) + t_xcd) << p.tpi_log) | (ts & (p.tiles_per_img - 1))) : ts;
    TG g;
    g.img = t >> p.tpi_log;
    g.tin = t & (p.tiles_per_img - 1);
    g.y0 = (g.tin >> p.tx_log) << 3;
    g.x0 = (g.tin & (p.tiles_x - 1)) << 4;
    return g;
  };

  float4 RA[U], RB[U];         // raw pieces of the chunks in flight: even chunks in RA, odd chunks in RB
  unsigned mA = 0, mB = 0;     // bit u: piece u of the set is inside the image
  auto issue = [&](float4& dst, unsigned& msk, int u, const TG& g, int q) {
    const int cg0 = q * KC;
    const bool from2 = cg0 >= C1;
    const float* src = from2 ? p.x2 : p.x1;
    const int cs_log = from2 ? p.c2_log : p.c1_log;
    const int hy = hyx[u] >> 8, hx = hyx[u] & 255;
    const int iy = g.y0 - 1 + hy, ix = g.x0 - 1 + hx;
    const bool ok = hyx[u] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    const unsigned pixel = (((((unsigned)g.img << p.h_log) + (unsigned)cy) << p.w_log) + (unsigned)cx);
    const unsigned off = (pixel << cs_log) + (unsigned)((from2 ? cg0 - C1 : cg0) + c4o);
    dst = *reinterpret_cast<const float4*>(src + off);
    msk = ok ? (msk | (1u << u)) : (msk & ~(1u << u));
  };

  // ---- staging of one piece in NSTEP steps (see conv_x6_down_kernel)
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
  dm_f2 ex = {0.f, 0.f};
  unsigned q1a = 0, q2a = 0, q1b = 0, q2b = 0;
  unsigned okm = 0u;
  float4 tte = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* tb_cur = tabs;      // table set (TF) of the chunk being staged, offset to this thread's channels
  float sc_cur = 1.f;              // power-of-two scale (!TF) of the chunk being staged
  const dm_f2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one2 = {1.f, 1.f};
  auto stage_step = [&](float4 (&R)[U], unsigned& msk, int u, int s, bf16_t* dstb, const TG& gl, int ql, bool do_load) {
    switch (s) {
      case 0: {
        okm = 0u - ((msk >> u) & 1u);
        const float4 v = R[u];
        if (TF) {
          const float4 tsc = *reinterpret_cast<const float4*>(tb_cur), tsh = *reinterpret_cast<const float4*>(tb_cur + C1);
          w0 = fmaf(v.x, tsc.x, tsh.x); w1 = fmaf(v.y, tsc.y, tsh.y); w2 = fmaf(v.z, tsc.z, tsh.z); w3 = fmaf(v.w, tsc.w, tsh.w);
        } else { w0 = v.x * sc_cur; w1 = v.y * sc_cur; w2 = v.z * sc_cur; w3 = v.w * sc_cur; }
      } break;
      case 1: if (do_load) issue(R[u], msk, u, gl, ql); break;
      case 2: if (TF) { const dm_f2 a = dm_f2{w0, w1} * nl2e; ex = dm_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; } break;
      case 3: if (TF) {
        const dm_f2 d = ex + one2;
        const dm_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const dm_f2 o = __builtin_elementwise_fma(dm_f2{w0, w1}, r, dm_f2{tte.x, tte.y});
        w0 = o.x; w1 = o.y;
      } break;
      case 4: if (TF) { const dm_f2 a = dm_f2{w2, w3} * nl2e; ex = dm_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; } break;
      case 5: if (TF) {
        const dm_f2 d = ex + one2;
        const dm_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        const dm_f2 o = __builtin_elementwise_fma(dm_f2{w2, w3}, r, dm_f2{tte.z, tte.w});
        w2 = o.x; w3 = o.y;
      } break;
      case 6:  // zero padding stays exactly 0
        w0 = __uint_as_float(__float_as_uint(w0) & okm); w1 = __uint_as_float(__float_as_uint(w1) & okm);
        w2 = __uint_as_float(__float_as_uint(w2) & okm); w3 = __uint_as_float(__float_as_uint(w3) & okm);
        break;
      case 7: dm_split_h2(w0, w1, q1a, q2a); break;
      case 9: dm_split_h2(w2, w3, q1b, q2b); break;
      case 11: if (hyx[u] >= 0) {
        bf16_t* q = dstb + ldso0 + u * 32 * RSB;
        *reinterpret_cast<uint2*>(q) = make_uint2(q1a, q1b);
        *reinterpret_cast<uint2*>(q + PL) = make_uint2(q2a, q2b);
      } break;
      default: break;
    }
  };
  // this thread's table row / scale for staging chunk q of a tile whose tables sit in set `ts` (scale: sc)
  auto chunk_setup = [&](int q, int ts, float sc) {
    if (TF) {
      tb_cur = tabs + ts * 3 * C1 + q * KC + c4o;
      tte = *reinterpret_cast<const float4*>(tb_cur + 2 * C1);
    } else sc_cur = sc;
  };

  // ---- what the staging of an image needs, in two halves (fetch early, finish late): TF: GroupNorm tables of channels tid, tid + 256
  //      (pin <= 8 partials: one round of independent loads); !TF: the producers' maxima (namax <= 64 per input)
  float2 pr[8];
  float pgam = 0.f, pbet = 0.f, ptmb = 0.f, pmx = 0.f;
  auto next_fetch = [&](const TG& g) {
    if (TF) {
      if (tid < C1) {
        const float* st = p.stats_in + ((size_t)g.img * p.pin * DM_GROUPS + (tid >> p.g1_log)) * 2;
#pragma unroll
        for (int q = 0; q < 8; ++q) pr[q] = *reinterpret_cast<const float2*>(st + (q < p.pin ? q : 0) * DM_GROUPS * 2);
        pgam = p.gamma[tid]; pbet = p.beta[tid];
        ptmb = p.temb ? p.temb[(size_t)step_now * p.temb_step_stride + (size_t)g.img * p.temb_stride + tid] : 0.f;
      }
    } else {
      pmx = lane < p.namax1 ? p.amax1[(size_t)g.img * p.namax1 + lane] : 0.f;
      if (p.amax2 && lane < p.namax2) pmx = fmaxf(pmx, p.amax2[(size_t)g.img * p.namax2 + lane]);
    }
  };
  // finish: TF -> table set ts; !TF -> returns the scale (inv through the reference)
  auto next_finish = [&](int ts, float& sc, float& inv) {
    if (TF) {
      if (tid < C1) {  // gn_tables' arithmetic (Chan merge of the partials in a fixed order)
        float msum = 0.f, m2 = 0.f, dev = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < p.pin) msum += pr[q].x;
        const float mean = msum / (float)p.pin;
#pragma unroll
        for (int q = 0; q < 8; ++q) if (q < p.pin) {
          const float d = pr[q].x - mean;
          m2 += pr[q].y;
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
  uint4 bq[2][NS];
  {
    const unsigned w0p = wtap(0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < NS; ++s) bq[j][s] = wload(w0p + s * split_off + j * jstride);
  }

  // K loop of chunk q (buffer PB = q & 1, compile time) while the stream's next chunk is staged from R into the other buffer.
  // gs / qs: tile and chunk being staged (do_stage); gl / ql: tile and chunk whose loads refill R (do_load).
  // PRE: 0 none, 1 = this K loop also prepares the next image's tables / scale (fetch at its start, finish at its end).
  auto run_chunk = [&](auto pbc, bool PRE, float4 (&R)[U], unsigned& msk, int q, bool do_stage, const TG& gl, int ql, bool do_load,
                       const TG& gpre, int ts_pre, float& sc_pre, float& inv_pre) {
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
    if (PRE) next_fetch(gpre);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int tapoff = ((tp / 3) * htw + tp % 3) * RSB;
      const int ntapoff = (((tp + 1) / 3) * htw + (tp + 1) % 3) * RSB;
      const unsigned wnext = (tp < 8) ? wtap(q, tp + 1) : wtap(q + 1 < nchunks ? q + 1 : 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint4 (*acur)[MB] = ((tp * 2 + j) & 1) ? a1 : a0;
        uint4 (*anxt)[MB] = ((tp * 2 + j) & 1) ? a0 : a1;
        const bool more = !(tp == 8 && j == 1);
        const int noff = (j == 0) ? tapoff + 16 : ntapoff;
        const uint4 b1 = bq[j][0], b2 = bq[j][1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; ++t) {   // h1 g2, h2 g1, h1 g1
          const int as = (t == 1) ? 1 : 0;
          const uint4 bv = (t == 0) ? b2 : b1;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            acc[mb] = dm_mfma16<true>(bv, acur[as][mb], acc[mb]);
            if (t < NS && more) anxt[t][mb] = *reinterpret_cast<const uint4*>(kb + ap[t][mb] + noff);
            if (t == 1 && mb == MB - 1) bq[j][1] = wload(wnext + split_off + j * jstride);
            if (t == 2 && mb == MB - 1) bq[j][0] = wload(wnext + j * jstride);
            constexpr int TOT = U * NSTEP;
            const int i = ((tp * 2 + j) * 3 + t) * MB + mb;
            if (do_stage) {
#pragma unroll
              for (int k = i * TOT / NMFMA; k < (i + 1) * TOT / NMFMA; ++k) stage_step(R, msk, k / NSTEP, k % NSTEP, sb, gl, ql, do_load);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    if (PRE) next_finish(ts_pre, sc_pre, inv_pre);
  };

  if (t_first >= t_end) return;
  TG g0 = tile_of(t_first);
  float sc0 = 1.f, inv0 = 1.f, sc1 = 1.f, inv1 = 1.f;   // power-of-two scales of this / the next tile's image (!TF)
  int tpar = 0;                                        // table set of this tile
  // prologue (once per workgroup): the first tile's tables / scale and chunk 0 in the open
  next_fetch(g0);
#pragma unroll
  for (int u = 0; u < U; ++u) issue(RA[u], mA, u, g0, 0);
#pragma unroll
  for (int u = 0; u < U; ++u) issue(RB[u], mB, u, g0, 1);
  next_finish(0, sc0, inv0);
  __syncthreads();
  chunk_setup(0, 0, sc0);
  {
    const bool two = nchunks > 2;   // loads of stream chunk 2: this tile's chunk 2, else the next tile's chunk 0
    const bool h1 = t_first + t_step < t_end;
    const TG gn = tile_of(h1 ? t_first + t_step : t_first);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) stage_step(RA, mA, u, s, tile, two ? g0 : gn, two ? 2 : 0, two || h1);
  }
  __syncthreads();

  for (int ts = t_first; ts < t_end; ts += t_step) {
    const bool has1 = ts + t_step < t_end, has2 = ts + 2 * t_step < t_end;
    const TG g1 = tile_of(has1 ? ts + t_step : ts), g2 = tile_of(has2 ? ts + 2 * t_step : ts);
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
        chunk_setup(q + 1, tpar, sc0);
        run_chunk(std::integral_constant<int, 0>{}, has1 && q == nchunks - 2, RB, mB, q, true, tl == 2 ? g2 : tl == 1 ? g1 : g0, pl - tl * nchunks, dl, g1, tpar ^ 1, sc1, inv1);
      }
      __syncthreads();
      {
        const int ps = q + 2, pl = q + 4;
        const int tsg = ps >= nchunks ? 1 : 0;
        const int tl = pl >= 2 * nchunks ? 2 : pl >= nchunks ? 1 : 0;
        const bool dstg = tsg == 0 || has1;
        const bool dl = dstg && (tl == 0 || (tl == 1 ? has1 : has2));
        if (tsg) chunk_setup(0, tpar ^ 1, sc1);
        else chunk_setup(ps, tpar, sc0);
        run_chunk(std::integral_constant<int, 1>{}, false, RA, mA, q + 1, dstg, tl == 2 ? g2 : tl == 1 ? g1 : g0, pl - tl * nchunks, dl, g1, tpar ^ 1, sc1, inv1);
      }
      __syncthreads();
    }
    x6_epilogue_sw<MB, 0>(p, acc, scr, 0, 0, 0, g0.img, g0.tin, g0.y0, g0.x0, wm, wn, lane, h, tid, wave, wscale * inv0);
    g0 = g1;
    sc0 = sc1; inv0 = inv1;
    tpar ^= 1;
  }
}
