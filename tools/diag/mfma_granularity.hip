// Does the fp32 MFMA's issue granularity (32x32x2: 64 cycles, 16x16x4: 32 cycles) change how fast a co-resident
// VALU ("staging-like") wave on the same SIMD gets through its instructions, and what it costs the MFMA wave?
//   blocks 0..255: staging-like wave set (VALU chain, timed);  blocks 256..511: MFMA loop (timed)
// hipcc --offload-arch=gfx950 -O3 mfma_granularity.hip -o mfma_granularity && ./mfma_granularity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int SHAPE, int PRIO_STAGE, int VK = 0>  // VK: the staging wave's instruction: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_rcp_f32, 4 LDS b128 write+read
                                        // SHAPE 0: no MFMA neighbour, 1: 32x32x2 (2 accumulators alternating), 2: 16x16x4, 3-5: see below
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int role = blockIdx.x / 256;
  if (role == 0) {
    __builtin_amdgcn_s_setprio(PRIO_STAGE);
    __builtin_amdgcn_s_sleep(100);  // let the neighbour get going
    float x[8];
    for (int q = 0; q < 8; ++q) x[q] = src[tid + q * 256];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __shared__ float4 lbuf[256];
#pragma unroll 1
    for (int rep = 0; rep < 16; ++rep) {
      if (VK == 0) {
#pragma unroll
        for (int q = 0; q < 40; ++q) x[q & 7] = x[q & 7] * 1.0001f + 0.5f;  // 640 VALU, 8 independent chains
      } else if (VK == 1) {
#pragma unroll
        for (int q = 0; q < 40; ++q) {
          f32x2 v = {x[(2 * q) & 7], x[(2 * q + 1) & 7]};
          v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
          x[(2 * q) & 7] = v.x; x[(2 * q + 1) & 7] = v.y;
        }
      } else if (VK == 2) {
#pragma unroll
        for (int q = 0; q < 40; ++q) x[q & 7] = __builtin_amdgcn_exp2f(x[q & 7]);
      } else if (VK == 3) {
#pragma unroll
        for (int q = 0; q < 40; ++q) x[q & 7] = __builtin_amdgcn_rcpf(x[q & 7]);
      } else {
#pragma unroll
        for (int q = 0; q < 20; ++q) {
          lbuf[tid] = make_float4(x[0], x[1], x[2], x[3]);
          const float4 r = lbuf[tid ^ 1];
          x[q & 7] += r.x;
        }
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += x[q];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
  } else if (SHAPE != 0) {
    float a0 = src[lane], a1 = src[lane + 64], b = src[lane + 128];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (SHAPE == 1) {
      f32x16 acc[2];
      for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
        }
      }
      for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    } else if (SHAPE >= 3) {
      // 32x32x2 with 8 accumulators: SHAPE 3 = four dependent MFMAs per accumulator in a row (the Winograd K loop's
      // pattern), 4 = round-robin over the 8 accumulators (always an independent MFMA next), 5 = ONE dependent chain
      f32x16 acc[8];
      for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
        if (SHAPE == 3) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(e & 1 ? a1 : a0, b, acc[q], 0, 0, 0);
        } else if (SHAPE == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(e & 1 ? a1 : a0, b, acc[q], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 32; ++e) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e & 1 ? a1 : a0, b, acc[0], 0, 0, 0);
        }
      }
      for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    } else {
      f32x4 acc[4];
      for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {  // same FLOPs per iteration: 64 x 16x16x4 = 32 x 32x32x2
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[3], 0, 0, 0);
        }
      }
      for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
  }
}

template <int SHAPE, int PS, int VK = 0> void run(const char* name, const float* src, float* out, unsigned long long* clk, int iters) {
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(clk, 0, 512 * 8));
    hipLaunchKernelGGL((k<SHAPE, PS, VK>), dim3(512), dim3(256), 0, 0, src, out, iters, clk);
    CK(hipDeviceSynchronize());
  }
  unsigned long long h[512];
  CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> a(h, h + 256), b(h + 256, h + 512);
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("%-46s 640 VALU: median %7llu cyc (p90 %7llu)   MFMA loop (%d x 32 x 4096 FLOP): median %8llu cyc\n", name, a[128], a[230], iters, b[128]);
}

int main() {
  float *src, *out; unsigned long long* clk;
  CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&clk, 512 * 8));
  CK(hipMemset(src, 0, 1 << 20));
  const int iters = 40;  // 40 x 32 MFMAs(32x32x2) = 81,920 cycles of MFMA per wave
  run<0, 0>("alone", src, out, clk, iters);
  run<1, 0>("beside 32x32x2 loop, staging prio 0", src, out, clk, iters);
  run<1, 2>("beside 32x32x2 loop, staging prio 2", src, out, clk, iters);
  run<2, 0>("beside 16x16x4 loop, staging prio 0", src, out, clk, iters);
  run<2, 2>("beside 16x16x4 loop, staging prio 2", src, out, clk, iters);
  run<3, 2>("beside 32x32x2, 8 acc, 4 in a row each, prio 2", src, out, clk, iters);
  run<4, 2>("beside 32x32x2, 8 acc round-robin, prio 2", src, out, clk, iters);
  run<5, 2>("beside 32x32x2, ONE dependent chain, prio 2", src, out, clk, iters);
  run<0, 2, 1>("v_pk_fma_f32 alone", src, out, clk, iters);
  run<3, 2, 1>("v_pk_fma_f32 beside 32x32x2 (8 acc)", src, out, clk, iters);
  run<0, 2, 2>("v_exp_f32 alone", src, out, clk, iters);
  run<3, 2, 2>("v_exp_f32 beside 32x32x2 (8 acc)", src, out, clk, iters);
  run<0, 2, 3>("v_rcp_f32 alone", src, out, clk, iters);
  run<3, 2, 3>("v_rcp_f32 beside 32x32x2 (8 acc)", src, out, clk, iters);
  run<0, 2, 4>("LDS b128 write+read alone (320 pairs)", src, out, clk, iters);
  run<3, 2, 4>("LDS b128 write+read beside 32x32x2 (8 acc)", src, out, clk, iters);
  return 0;
}
