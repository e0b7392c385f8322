// dm_kernels_x6p.hip — conv_x6_down_kernel: the stride-2 4x4 convolution of conv_x6_kernel<2, 1, true> (fp32 operands as
// three bf16 terms on the bf16 matrix pipe, dm_kernels_x6.hip) with its STAGING SOFTWARE-PIPELINED INTO THE K LOOP.
//
// Why (tools/diag/r3_filler.sh, profiles/r3_diag_filler.log): a bf16 MFMA occupies the matrix pipe for 32 cycles and its wave's
// issue slot for 4; three more vector instructions of the SAME wave behind every MFMA cost the kernel 3 %, six cost 12 %.
// Vector work of ANOTHER wave of the SIMD is starved by an MFMA stream instead, so conv_x6_kernel's staging
// phase (GroupNorm + SiLU + three-way split: ~11 vector instructions per element, 2.6 per MFMA) is paid in full while the
// partner workgroup's K loop crawls around it.  Here the wave that issues the MFMAs of K chunk c also transforms, splits and
// writes chunk c+1 (from registers loaded two chunks earlier) into the other half of a double-buffered LDS tile, a few
// instructions behind each MFMA, and issues the global loads of chunk c+3 as registers become free.
//
// Same arithmetic as conv_x6_kernel<MB, 1, SW>: K is walked as (input phase, 32-channel half, tap, 16 channels) in both, so
// the two are bit-identical (the 64-pixel-tile variant serves small batches, dm_api.hip).
//   chunk  = (input phase (a, b), 32 input channels): a (th+1) x (tw+1) plane of ONE phase, 4 taps x 2 x 16 channels
//   LDS    = GroupNorm tables | reduction words | 2 buffers x 3 planes x [plane pixel][32 + 8] bf16   (73.4 KB + tables)
//   stream = chunks of consecutive tiles: the loads of the next tile's first two chunks are issued by this tile's last two
#pragma once
#include "dm_kernels_x6.hip"

template <int MB, int NS, bool TF, bool F16 = false>
__global__ __launch_bounds__(256, 2) void conv_x6_down_kernel(const ConvArgs p) {
  static_assert(!F16 || (NS == 2 && TF), "the two-way fp16 split (dm_split_h2) serves inputs behind a GroupNorm");
  constexpr int KC = 32, RSB = KC + 8;
  constexpr int U = (MB == 2) ? 5 : 3;      // pieces per thread and chunk: 17x9 / 9x9 plane pixels x 8 pieces = 1224 / 648 <= 256 U
  constexpr int NSTEP = 12;                 // staging steps per piece (4 channels of one plane pixel)
  constexpr int NMFMA = 4 * 2 * (NS == 3 ? 6 : 3) * MB;   // MFMAs of a chunk's K loop
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef DM_TIMELINE
  int tl_n = 0;
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 128] = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5;
  const int C1 = p.C1;
  constexpr int tw = 8, htw = tw + 1, hth = 8 * MB + 1;   // tile = 8 MB x 8 output pixels (host-checked): plane = (8 MB + 1) x 9
  constexpr int NP = hth * htw, items = NP * 8;
  const int step_now = (p.step_ptr ? *p.step_ptr : 0) + p.step_off;
  float* s_scale = smem;
  float* s_shift = s_scale + (TF ? C1 : 0);
  float* s_temb = s_shift + (TF ? C1 : 0);
  float* scr = s_temb + (TF ? C1 : 0);          // 64 reduction words (x6_epilogue_sw)
  bf16_t* tile = reinterpret_cast<bf16_t*>(scr + 64);
  constexpr int PL = NP * RSB;                  // plane stride (bf16 elements)
  constexpr int BUF = NS * PL;                  // buffer stride

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = wm * (32 * MB) + mb * 32 + (lane & 31);
    abase[mb] = ((m >> 3) * htw + (m & (tw - 1))) * RSB + 8 * h;
  }
  // staging pieces of this thread: piece u = 4 channels (c4 = tid & 7) of plane pixel (u * 256 + tid) >> 3, every chunk alike
  int hyx[U];
  const int ldso0 = (tid >> 3) * RSB + (tid & 7) * 4;   // LDS offset of piece 0; piece u: + u * 32 pixels
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = u * 256 + tid;
    const int pix = idx >> 3;
    const int r = pix / htw;
    const bool v = idx < items;
    hyx[u] = v ? ((r << 8) | (pix - r * htw)) : -1;
  }
  const int c4o = (tid & 7) * 4;

  const int cch_log = p.c1_log - 5;             // 32-channel chunks per phase
  const int nchunks = 4 << cch_log;
  const int nb32 = p.Cout >> 5;
  // packed weights [split][tap 16][C1/16][Cout/32][64 lanes][8 bf16]: BYTE offsets in 32 bits from one wave-uniform base
  const unsigned jstride = (unsigned)nb32 * 1024u;
  const unsigned tapstride = (unsigned)(C1 >> 4) * jstride;
  const unsigned split_off = 16u * tapstride;
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  const char* wcol = reinterpret_cast<const char*>(p.wp) + (size_t)(blockIdx.y * 2 + wn_s) * 1024;
  const unsigned lane16 = (unsigned)lane * 16u;
  // weights of (chunk q, tap tp): the phase's 2x2 taps inside the 4x4 kernel, 16-channel groups 2 cc, 2 cc + 1
  auto wtap = [&](int q, int tp) {
    const int ph = q >> cch_log, cc = q & ((1 << cch_log) - 1);
    const int wt = (2 * (tp >> 1) + 1 - (ph >> 1)) * 4 + 2 * (tp & 1) + 1 - (ph & 1);
    return (unsigned)wt * tapstride + (unsigned)(2 * cc) * jstride;
  };
  auto wload = [&](unsigned off) { return *reinterpret_cast<const uint4*>(wcol + (off + lane16)); };
  const float oscale = F16 ? p.wp[split_off / 2] : 1.f;   // fp16 form: the float 2^-k behind the two planes

  struct TG { const float* src; int y0, x0, tin, img; };
  const int t_first = p.xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = p.xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_xcd = p.xcd_map ? (int)(blockIdx.x & 7) : 0;
  const int t_end = p.xcd_map ? (((p.ngroups - t_xcd + 7) >> 3) << p.tpi_log) : p.ntiles;
  auto tile_of = [&](int ts) {
    const int t = p.xcd_map ? (((((ts >> p.tpi_log) << 3) + t_xcd) << p.tpi_log) | (ts & (p.tiles_per_img - 1))) : ts;
    TG g;
    g.img = t >> p.tpi_log;
    g.tin = t & (p.tiles_per_img - 1);
    g.y0 = (g.tin >> p.tx_log) << p.th_log;
    g.x0 = (g.tin & (p.tiles_x - 1)) << p.tw_log;
    g.src = p.x1 + ((size_t)g.img << (p.h_log + p.w_log + p.c1_log)) + c4o;
    return g;
  };

  float4 RA[U], RB[U];         // raw pieces of the chunks in flight: even chunks in RA, odd chunks in RB
  unsigned mA = 0, mB = 0;     // bit u: piece u of the set is inside the image (zero padding otherwise)
  // issue the load of piece u of chunk q of tile g
  auto issue = [&](float4& dst, unsigned& msk, int u, const TG& g, int q) {
    const int ph = q >> cch_log, cc = q & ((1 << cch_log) - 1);
    const int hy = hyx[u] >> 8, hx = hyx[u] & 255;
    const int iy = 2 * g.y0 - (ph >> 1) + 2 * hy, ix = 2 * g.x0 - (ph & 1) + 2 * hx;
    const bool ok = hyx[u] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
    const unsigned off = (((unsigned)(cy << p.w_log) + cx) << p.c1_log) + (unsigned)(cc * KC);
    dst = *reinterpret_cast<const float4*>(g.src + off);   // unconditional, clamped address; padding is a select later
    msk = ok ? (msk | (1u << u)) : (msk & ~(1u << u));
  };

  // ---- staging of one piece in NSTEP small steps (working registers shared by all pieces: one piece at a time)
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
  dm_f2 ex = {0.f, 0.f};
  unsigned q1a = 0, q2a = 0, q3a = 0, q1b = 0, q2b = 0, q3b = 0;
  dm_f2 rs = {0.f, 0.f};
  unsigned okm = 0u;   // all ones: the piece is inside the image
  float4 tte = make_float4(0.f, 0.f, 0.f, 0.f);  // time embedding of this thread's 4 channels of the chunk being staged
  int tch = 0;                                   // ... and their table offset
  const dm_f2 nl2e = {-1.4426950408889634f, -1.4426950408889634f}, one2 = {1.f, 1.f};
  // R: the register set of the chunk being staged; q: that chunk; dstb: its LDS buffer; (gl, ql, do_load): the load that refills R[u]
  auto stage_step = [&](float4 (&R)[U], unsigned& msk, int u, int s, bf16_t* dstb, const TG& gl, int ql, bool do_load) {
    switch (s) {
      case 0: {
        okm = 0u - ((msk >> u) & 1u);
        const float4 v = R[u];
        const float4 tsc = *reinterpret_cast<const float4*>(s_scale + tch), tsh = *reinterpret_cast<const float4*>(s_shift + tch);
        if (TF) { w0 = fmaf(v.x, tsc.x, tsh.x); w1 = fmaf(v.y, tsc.y, tsh.y); w2 = fmaf(v.z, tsc.z, tsh.z); w3 = fmaf(v.w, tsc.w, tsh.w); }
        else { w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w; }
      } break;
      case 1: break;   // (the refill load of R[u] sat here; now one batch per chunk, see run_chunk)
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
      case 7: {  // three-way bf16 split (dm_split3) / two-way fp16 split (dm_split_h2), pair (w0, w1), first half
        if constexpr (F16) dm_split_h2(w0, w1, q1a, q2a);
        else {
          rs = dm_f2{w0, w1};
          q1a = __builtin_bit_cast(unsigned, __builtin_convertvector(rs, dm_bf2));
          rs -= dm_f2{__uint_as_float(q1a << 16), __uint_as_float(q1a & 0xffff0000u)};
          q2a = __builtin_bit_cast(unsigned, __builtin_convertvector(rs, dm_bf2));
        }
      } break;
      case 8: if constexpr (NS == 3) {
        rs -= dm_f2{__uint_as_float(q2a << 16), __uint_as_float(q2a & 0xffff0000u)};
        q3a = __builtin_bit_cast(unsigned, __builtin_convertvector(rs, dm_bf2));
      } break;
      case 9: {
        if constexpr (F16) dm_split_h2(w2, w3, q1b, q2b);
        else {
          rs = dm_f2{w2, w3};
          q1b = __builtin_bit_cast(unsigned, __builtin_convertvector(rs, dm_bf2));
          rs -= dm_f2{__uint_as_float(q1b << 16), __uint_as_float(q1b & 0xffff0000u)};
          q2b = __builtin_bit_cast(unsigned, __builtin_convertvector(rs, dm_bf2));
        }
      } break;
      case 10: if constexpr (NS == 3) {
        rs -= dm_f2{__uint_as_float(q2b << 16), __uint_as_float(q2b & 0xffff0000u)};
        q3b = __builtin_bit_cast(unsigned, __builtin_convertvector(rs, dm_bf2));
      } break;
      default: {
        // (no branch: behind a condition the compiler sinks the whole piece's arithmetic into the conditional block, one serial blob
        //  behind one MFMA; a piece outside the plane goes to the 8 unused elements behind pixel 0's 32)
        bf16_t* q = dstb + (hyx[u] >= 0 ? ldso0 + u * 32 * RSB : KC);
        *reinterpret_cast<uint2*>(q) = make_uint2(q1a, q1b);
        *reinterpret_cast<uint2*>(q + PL) = make_uint2(q2a, q2b);
        if (NS == 3) *reinterpret_cast<uint2*>(q + 2 * PL) = make_uint2(q3a, q3b);
      } break;
    }
  };
  // this thread's table entries for the chunk to be staged (its 4 channels never change within a 32-channel half)
  auto chunk_tables = [&](int q) {
    if (TF) {
      tch = (q & ((1 << cch_log) - 1)) * KC + c4o;
      tte = *reinterpret_cast<const float4*>(s_temb + tch);
    }
  };

  f32x16 acc[MB];
  // weight fragments of the upcoming taps, [tap buffer][j][plane].  WD = taps of look-ahead: with two planes (three MFMAs per product)
  // a (tap, j) group is six MFMAs and one tap ahead (12 MFMAs) does not cover an L2 hit - two taps ahead (24 MFMAs) do
  constexpr int WD = (NS == 2) ? 2 : 1;
  uint4 bq[WD][2][NS];
  {
    const unsigned w0p = wtap(0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bq[0][j][s] = wload(w0p + s * split_off + j * jstride);
        if (WD == 2) bq[WD - 1][j][s] = wload(wtap(0, 1) + s * split_off + j * jstride);
      }
  }

  // K loop of chunk q on buffer PB (compile-time) while chunk q + 1 is staged from R into the other buffer
  auto run_chunk = [&](auto pbc, auto stc, float4 (&R)[U], unsigned& msk, int q, const TG& gcur, const TG& gnext, bool has_next) {
    constexpr int PB = decltype(pbc)::value;
    constexpr bool do_stage = decltype(stc)::value;   // false: the tile's last chunk
    const bf16_t* kb = tile + PB * BUF;
    bf16_t* sb = tile + (PB ^ 1) * BUF;
    const int ql_raw = q + 3;                     // staging chunk q + 1 frees its registers for chunk q + 3
    const bool next_tile = ql_raw >= nchunks;
    const bool do_load = do_stage && (!next_tile || has_next);
    const int ql = next_tile ? ql_raw - nchunks : ql_raw;
    const TG& gl = next_tile ? gnext : gcur;
    if (do_stage) chunk_tables(q + 1);
    uint4 a0[NS][MB], a1[NS][MB];
    int ap[NS][MB];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        ap[s][mb] = s * PL + abase[mb];
        a0[s][mb] = *reinterpret_cast<const uint4*>(kb + ap[s][mb]);
      }
    DM_TL(2);
    constexpr int NTERM = (NS == 3) ? 6 : 3;
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
      const int tapoff = ((tp >> 1) * htw + (tp & 1)) * RSB;
      const int ntapoff = (((tp + 1) >> 1) * htw + ((tp + 1) & 1)) * RSB;
      const int qn_ = q + 1 < nchunks ? q + 1 : 0;
      const unsigned wnext = (tp + WD < 4) ? wtap(q, tp + WD) : wtap(qn_, tp + WD - 4);
      const int tb = (WD == 2) ? (tp & 1) : 0;   // (four taps per chunk: the buffer parity is the tap's)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint4 (*acur)[MB] = ((tp * 2 + j) & 1) ? a1 : a0;
        uint4 (*anxt)[MB] = ((tp * 2 + j) & 1) ? a0 : a1;
        const bool more = !(tp == 3 && j == 1);
        const int noff = (j == 0) ? tapoff + 16 : ntapoff;
        const uint4 b1 = bq[tb][j][0];
        const uint4 b2 = bq[tb][j][1];
        const uint4 b3 = bq[tb][j][NS - 1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NTERM; ++t) {
          const int as = (NS == 3) ? ((t == 1 || t == 4) ? 1 : (t == 3) ? 2 : 0) : (t == 1 ? 1 : 0);
          const uint4 bv = (NS == 3) ? ((t == 0) ? b3 : (t == 1 || t == 2) ? b2 : b1) : (t == 0 ? b2 : b1);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            acc[mb] = dm_mfma16<F16>(bv, acur[as][mb], acc[mb]);
            if (t < NS && more) anxt[t][mb] = *reinterpret_cast<const uint4*>(kb + ap[t][mb] + noff);
            if (NS == 3) {
              if (t == 3 && mb == 0) bq[tb][j][2] = wload(wnext + 2 * split_off + j * jstride);
              if (t == 4 && mb == 0) bq[tb][j][1] = wload(wnext + split_off + j * jstride);
            } else {
              if (t == 1 && mb == MB - 1) bq[tb][j][1] = wload(wnext + split_off + j * jstride);
            }
            if (t == NTERM - 1 && mb == MB - 1) bq[tb][j][0] = wload(wnext + j * jstride);
            // the staging steps that ride behind this MFMA
            constexpr int TOT = U * NSTEP;
            const int i = ((tp * 2 + j) * NTERM + t) * MB + mb;
            if (do_stage) {
#pragma unroll
              for (int k = i * TOT / NMFMA; k < (i + 1) * TOT / NMFMA; ++k) stage_step(R, msk, k / NSTEP, k % NSTEP, sb, gl, ql, do_load);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    // the refill of R in ONE batch behind the chunk's last staging step: loads return in order, so a wait for a weight fragment
    // requested after a refill load also waits for that load (HBM latency, not an L2 hit); one batch per chunk is covered by the
    // fragments' lead, a refill load behind every piece was not
    if (do_load) {
#pragma unroll
      for (int u = 0; u < U; ++u) issue(R[u], msk, u, gl, ql);
    }
    DM_TL(3);
  };

  TG gcur = tile_of(t_first < t_end ? t_first : 0);
  if (t_first < t_end) {
#pragma unroll
    for (int u = 0; u < U; ++u) issue(RA[u], mA, u, gcur, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) issue(RB[u], mB, u, gcur, 1);
  }
  for (int ts = t_first; ts < t_end; ts += t_step) {
    const bool has_next = ts + t_step < t_end;
    const TG gnext = tile_of(has_next ? ts + t_step : ts);
    DM_TL(1);
    if (TF) {
      int tg = threadIdx.x;
      asm volatile("" : "+v"(tg));
      gn_tables(p, step_now, gcur.img, tg, s_scale, s_shift, s_temb);
      __syncthreads();
    }
    // chunk 0: staged in the open (the only staging of a tile outside a K loop); its registers then take chunk 2
    chunk_tables(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) stage_step(RA, mA, u, s, tile, gcur, 2, true);
#pragma unroll
    for (int u = 0; u < U; ++u) issue(RA[u], mA, u, gcur, 2);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < nchunks; q += 2) {
      run_chunk(std::integral_constant<int, 0>{}, std::true_type{}, RB, mB, q, gcur, gnext, has_next);
      __syncthreads();
      if (q + 2 < nchunks) run_chunk(std::integral_constant<int, 1>{}, std::true_type{}, RA, mA, q + 1, gcur, gnext, has_next);
      else run_chunk(std::integral_constant<int, 1>{}, std::false_type{}, RA, mA, q + 1, gcur, gnext, has_next);
      __syncthreads();
    }
    x6_epilogue_sw<MB, 1>(p, acc, scr, 0, 0, 0, gcur.img, gcur.tin, gcur.y0, gcur.x0, wm, wn, lane, h, tid, wave, oscale);
    DM_TL(4);
    gcur = gnext;
  }
}
