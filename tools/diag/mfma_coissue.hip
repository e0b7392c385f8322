// Which instruction classes of a "staging-like" wave are held up by a co-resident wave that streams fp32 MFMAs
// (32x32x2, 8 accumulators, four dependent MFMAs in a row: the Winograd K loop's pattern)?
//   blocks 0..255: the instruction class under test (timed);  blocks 256..511: MFMA stream
// hipcc --offload-arch=gfx950 -O3 mfma_coissue.hip -o mfma_coissue && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// -DBF16_STREAM: the MFMA stream is v_mfma_f32_32x32x16_bf16 (32 cycles each) instead of v_mfma_f32_32x32x2_f32 (64 cycles)
#ifdef BF16_STREAM
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_float4(a_, a_, b_, b_)), __builtin_bit_cast(bf16x8, make_float4(b_, a_, b_, a_)), c_, 0, 0, 0)
#else
#define MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, 0, 0, 0)
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MF: 0 no MFMA neighbour, 1 pure MFMA stream, 2 MFMA stream with 2 v_pk_fma per 4 MFMAs (the K loop's own VALU),
//     3 MFMA stream with an LDS b128 read per 4 MFMAs as well
template <int VK, int MF>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters, unsigned long long* clk, int reps, int swap) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int role = (blockIdx.x / 256) ^ swap;  // swap = 1: the MFMA workgroups are dispatched FIRST (older waves)
  __shared__ float4 lbuf[1024];
  if (role == 0) {
    __builtin_amdgcn_s_setprio(2);
    __builtin_amdgcn_s_sleep(100);
    float x[8];
    for (int q = 0; q < 8; ++q) x[q] = src[tid + q * 256];
    int ix[8];
    for (int q = 0; q < 8; ++q) ix[q] = tid * 7 + q;
    lbuf[tid] = make_float4(0, 0, 0, 0); lbuf[tid + 256] = lbuf[tid]; lbuf[tid + 512] = lbuf[tid]; lbuf[tid + 768] = lbuf[tid];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int rep = 0; rep < reps; ++rep) {
      if (VK == 0) {            // 40 v_fma_f32
#pragma unroll
        for (int q = 0; q < 40; ++q) x[q & 7] = x[q & 7] * 1.0001f + 0.5f;
      } else if (VK == 1) {     // 40 x (v_cmp + v_cndmask): VCC traffic
#pragma unroll
        for (int q = 0; q < 40; ++q) x[q & 7] = (x[q & 7] > 0.25f * q) ? x[(q + 1) & 7] : x[(q + 3) & 7];
      } else if (VK == 2) {     // 40 integer address-math VALU (shift, and, add, mul_lo)
#pragma unroll
        for (int q = 0; q < 10; ++q) { int& v = ix[q & 7]; v = ((v << 3) & 0xfff0) + ix[(q + 1) & 7]; v = v * 18 + q; }
      } else if (VK == 3) {     // 40 SALU
        int sv = __builtin_amdgcn_readfirstlane(ix[0]);
#pragma unroll
        for (int q = 0; q < 40; ++q) asm volatile("s_add_u32 %0, %0, 0x1234\n" : "+s"(sv));
        ix[0] += sv;
      } else if (VK == 4) {     // 8 dependent LDS b128 reads (value -> address)
        int a = ix[0] & 255;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const float4 r = lbuf[a]; a = (__float_as_int(r.x) + tid + q) & 1023; }
        ix[0] += a;
      } else if (VK == 5) {     // 13 independent global loads (L2-resident) + wait
        float4 v[13];
#pragma unroll
        for (int q = 0; q < 13; ++q) v[q] = *reinterpret_cast<const float4*>(src + (((ix[0] + q * 4096 + rep * 64) & 0x3fff) * 4));
#pragma unroll
        for (int q = 0; q < 13; ++q) x[q & 7] += v[q].x;
      } else if (VK == 6) {     // 8 barriers
#pragma unroll
        for (int q = 0; q < 8; ++q) { x[q] += 1.f; __syncthreads(); }
      } else if (VK == 7) {     // 16 cross-lane shuffles (ds_bpermute / dpp)
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q & 7] += __shfl_xor(x[q & 7], 1 << (q % 6), 64);
      } else if (VK == 8) {     // 13 LDS b128 writes + barrier + 13 reads (staging's LDS traffic)
#pragma unroll
        for (int q = 0; q < 13; ++q) lbuf[(tid + q * 64) & 1023] = make_float4(x[q & 7], x[1], x[2], x[3]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 13; ++q) x[q & 7] += lbuf[(tid * 3 + q * 64) & 1023].y;
        __syncthreads();
      } else if (VK == 9) {     // 8 global float4 stores
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(out + ((size_t)(blockIdx.x * 8 + q) * 256 + tid) * 4 + 1048576) = make_float4(x[q], x[1], x[2], x[3]);
      } else if (VK == 10) {    // 40 v_pk_fma_f32
#pragma unroll
        for (int q = 0; q < 20; ++q) {
          f32x2 v = {x[(2 * q) & 7], x[(2 * q + 1) & 7]};
          v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
          v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
          x[(2 * q) & 7] = v.x; x[(2 * q + 1) & 7] = v.y;
        }
      } else if (VK == 11) {    // 20 exp + 20 rcp
#pragma unroll
        for (int q = 0; q < 20; ++q) x[q & 7] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x[q & 7]));
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += x[q] + (float)ix[q];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[role * 256 + (blockIdx.x & 255)] = t1 - t0;
  } else if (MF != 0) {
    float a0 = src[lane], a1 = src[lane + 64], b = src[lane + 128];
    f32x2 e0 = {src[lane + 192], src[lane + 256]}, e1 = {src[lane + 320], src[lane + 384]};
    lbuf[tid] = make_float4(a0, a1, b, 0.f);
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    int sv = __builtin_amdgcn_readfirstlane(tid);
    float4 gl = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pl[8], pg[8];
    for (int q = 0; q < 8; ++q) { pl[q] = gl; pg[q] = gl; }
    for (int it = 0; it < iters; ++it) {
      if (MF == 12) { __builtin_amdgcn_sched_barrier(0); e0 = __builtin_elementwise_fma(e0, f32x2{1.0001f, 1.0001f}, e1); asm volatile("" : "+v"(e0)); __builtin_amdgcn_sched_barrier(0); }  // ONE v_pk_fma per 32 MFMAs
      if (MF == 13) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop"); __builtin_amdgcn_sched_barrier(0); }  // ONE v_nop
      if (MF == 18) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0"); __builtin_amdgcn_sched_barrier(0); }  // ONE s_nop
      if (MF == 19 && (it & 7) == 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop"); __builtin_amdgcn_sched_barrier(0); }  // ONE v_nop per 256 MFMAs
      if (MF == 20) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15"); __builtin_amdgcn_sched_barrier(0); }  // v_nop, then 64 idle cycles
      if (MF == 21) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n v_nop"); __builtin_amdgcn_sched_barrier(0); }  // MFMA pipe drained first, then v_nop
      if (MF == 30) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_barrier"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 31) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_sleep 1"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 32) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 33) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 34 && it == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop"); __builtin_amdgcn_sched_barrier(0); }  // ONE v_nop in the whole run, after the first 32 MFMAs
      if (MF == 35) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_branch 1f\n s_nop 0\n1:"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 36) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop\n s_sleep 8"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 22) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_setprio 0\n v_nop\n s_setprio 3"); __builtin_amdgcn_sched_barrier(0); }
      if (MF == 14) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_mov_b32 %0, %0" : "+v"(e0.x)); __builtin_amdgcn_sched_barrier(0); }  // ONE v_mov
      if (MF == 15) { __builtin_amdgcn_sched_barrier(0); e1.y += pl[0].x; pl[0] = lbuf[(tid + it) & 255]; __builtin_amdgcn_sched_barrier(0); }  // ONE LDS read, consumed an iteration later
      if (MF == 16 || MF == 17) { __builtin_amdgcn_sched_barrier(0); e0 = __builtin_elementwise_fma(e0, f32x2{1.0001f, 1.0001f}, e1); e1 = __builtin_elementwise_fma(e1, f32x2{0.9999f, 0.9999f}, e0); asm volatile("" : "+v"(e0), "+v"(e1)); __builtin_amdgcn_sched_barrier(0); }
      if (MF >= 9 && MF <= 11) {  // everything that is not an MFMA in ONE fenced batch in front of the 32 MFMAs
        // MF 9: 16 v_pk_fma; MF 10: + 8 LDS b128 reads consumed one iteration later; MF 11: + 8 global loads as well
        __builtin_amdgcn_sched_barrier(0);
        if (MF >= 10) {
          float sacc = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) { sacc += pl[q].x; pl[q] = lbuf[(tid + q * 16 + it) & 255]; }
          e1.y += sacc;
        }
        if (MF >= 11) {
          float sacc = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) { sacc += pg[q].x; pg[q] = *reinterpret_cast<const float4*>(src + ((tid + q * 256 + it * 2048) & 0xffff) * 4); }
          e0.y += sacc;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          e0 = __builtin_elementwise_fma(e0, f32x2{1.0001f, 1.0001f}, e1);
          e1 = __builtin_elementwise_fma(e1, f32x2{0.9999f, 0.9999f}, e0);
          asm volatile("" : "+v"(e0), "+v"(e1));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (MF == 2 || MF == 3 || MF == 4) {  // 2 v_pk_fma per 4 MFMAs; MF 4: results NOT consumed by the MFMAs
          e0 = __builtin_elementwise_fma(e0, f32x2{1.0001f, 1.0001f}, e1);
          e1 = __builtin_elementwise_fma(e1, f32x2{0.9999f, 0.9999f}, e0);
          asm volatile("" : "+v"(e0), "+v"(e1));
        }
        if (MF == 5) {  // 2 plain v_fma_f32 per 4 MFMAs, not consumed
          e0.x = fmaf(e0.x, 1.0001f, e1.x); e1.x = fmaf(e1.x, 0.9999f, e0.x);
          asm volatile("" : "+v"(e0), "+v"(e1));
        }
        if (MF == 3 || MF == 6) { const float4 r = lbuf[(tid + q * 16 + it) & 255]; asm volatile("" :: "v"(r.x), "v"(r.w)); }  // LDS b128 read per 4 MFMAs
        if (MF == 7) { gl = *reinterpret_cast<const float4*>(src + ((tid + q * 256 + it * 2048) & 0xffff) * 4); asm volatile("" :: "v"(gl.x), "v"(gl.w)); }
        if (MF == 8) { asm volatile("s_add_u32 %0, %0, 0x1234\ns_add_u32 %0, %0, 0x1234\n" : "+s"(sv)); }
        const bool use = (MF == 2 || MF == 3 || (MF >= 9 && MF <= 11));
        if (MF == 17) {
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n v_mfma_f32_32x32x2_f32 %0, %3, %2, %0\n v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n v_mfma_f32_32x32x2_f32 %0, %3, %2, %0" : "+a"(acc[q]) : "v"(a0), "v"(b), "v"(a1));
        } else {
        acc[q] = MFMA(use ? e0.x : a0, b, acc[q]);
        acc[q] = MFMA(use ? e0.y : a1, b, acc[q]);
        acc[q] = MFMA(use ? e1.x : a0, b, acc[q]);
        acc[q] = MFMA(use ? e1.y : a1, b, acc[q]);
        }
      }
    }
    if (MF >= 9) __builtin_amdgcn_sched_barrier(0);
    a0 += e0.x + e1.x + gl.x + (float)sv + pl[0].y + pg[0].y;
    out[blockIdx.x * 256 + tid + 4096] = a0;
    float s = 0.f;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[role * 256 + (blockIdx.x & 255)] = t1 - t0;
  }
}

template <int VK, int MF> unsigned long long run1(const float* src, float* out, unsigned long long* clk, int iters, int reps, unsigned long long* mf, int swap = 0) {
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(clk, 0, 512 * 8));
    hipLaunchKernelGGL((k<VK, MF>), dim3(512), dim3(256), 0, 0, src, out, iters, clk, reps, swap);
    CK(hipDeviceSynchronize());
  }
  unsigned long long h[512];
  CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> a(h, h + 256), b(h + 256, h + 512);
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  *mf = b[128];
  return a[128];
}
template <int VK> void run(const char* name, const float* src, float* out, unsigned long long* clk) {
  const int iters = 60, reps = 16;  // MFMA stream: 60 x 32 x 64 = 122,880 cycles
  unsigned long long m[12], t[12];
  t[0] = run1<VK, 0>(src, out, clk, iters, reps, &m[0]);
  t[1] = run1<VK, 1>(src, out, clk, iters, reps, &m[1]);
  t[2] = run1<VK, 13>(src, out, clk, iters, reps, &m[2]);
  t[3] = run1<VK, 4>(src, out, clk, iters, reps, &m[3]);
  t[4] = run1<VK, 6>(src, out, clk, iters, reps, &m[4]);
  t[5] = run1<VK, 7>(src, out, clk, iters, reps, &m[5]);
  printf("%-30s alone %6llu |", name, t[0]);
  const char* nm[] = {"", "pure MFMA", "1 v_nop / 32 MFMA", "2 pk_fma / 4 MFMA", "1 LDS read / 4 MFMA", "1 global load / 4 MFMA"};
  for (int i = 1; i < 6; ++i) printf(" %s x%.2f (mfma %llu) |", nm[i], (double)t[i] / t[0], m[i]);
  printf("\n");
}

int main() {
  float *src, *out; unsigned long long* clk;
  CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&clk, 512 * 8));
  CK(hipMemset(src, 0, 1 << 20));
  printf("cycles for 16 repetitions of the block under test (median over 256 workgroups); MFMA stream alone = 122,880\n");
  run<0>("40 v_fma_f32", src, out, clk);
  run<3>("40 SALU s_add_u32", src, out, clk);
  run<4>("8 dependent LDS b128 reads", src, out, clk);
  return 0;
}
