// Sustained clock and rate of bare MFMA streams with random operands on all CUs:
//   fp32 32x32x2, bf16 32x32x16 (dense), bf16 with small-magnitude operands (split residual terms), at 1-2 waves per SIMD.
// clock = delta s_memtime / delta s_memrealtime (100 MHz).   hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int V, int NACC = 4>  // 0: fp32 32x32x2, 1: bf16 32x32x16; NACC independent accumulators (dependency distance)
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x;
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const uint4 ua = *reinterpret_cast<const uint4*>(src + tid * 4), ub = *reinterpret_cast<const uint4*>(src + 4096 + tid * 4);
  const float fa = src[tid], fb = src[tid + 256];
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 32 / NACC; ++e)
#pragma unroll
      for (int q = 0; q < NACC; ++q) {
        if (V == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[q], 0, 0, 0);
        else acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[q], 0, 0, 0);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int V, int NACC = 4> void run(const char* name, const float* src, float* out, unsigned long long* clk, int blocks, int iters, double flop_per_mfma) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, NACC>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * 2);
  CK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
  std::vector<double> ghz, cyc;
  for (int b = 0; b < blocks; ++b) { ghz.push_back((double)h[2 * b] / h[2 * b + 1] * 0.1); cyc.push_back((double)h[2 * b]); }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
  const double mf = (double)blocks * 4 * iters * 32;
  printf("%-46s %d workgroups: %8.1f us, clock median %.2f GHz, %.0f cycles per MFMA per SIMD-wave, %.1f TFLOP/s\n", name, blocks, ms * 1e3,
         ghz[blocks / 2], cyc[blocks / 2] / (iters * 32.0), mf * flop_per_mfma / (ms * 1e-3) / 1e12);
}
int main() {
  float *src, *out; unsigned long long* clk;
  CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&clk, 1024 * 16));
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 2.f;
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const int iters = 4000;
  run<0>("fp32 32x32x2, random operands, 1 wave/SIMD", src, out, clk, 256, iters / 2, 4096.0);
  run<0>("fp32 32x32x2, random operands, 2 waves/SIMD", src, out, clk, 512, iters / 2, 4096.0);
  run<1>("bf16 32x32x16, random operands, 1 wave/SIMD", src, out, clk, 256, iters, 32768.0);
  run<1>("bf16 32x32x16, random operands, 2 waves/SIMD", src, out, clk, 512, iters, 32768.0);
  run<1, 2>("bf16 32x32x16, 2 accumulators (distance 2), 1 wave/SIMD", src, out, clk, 256, iters, 32768.0);
  run<1, 1>("bf16 32x32x16, 1 accumulator (distance 1), 1 wave/SIMD", src, out, clk, 256, iters, 32768.0);
  run<1, 3>("bf16 32x32x16, 3 accumulators, 1 wave/SIMD", src, out, clk, 256, iters, 32768.0 * 30 / 32);
  run<0, 2>("fp32 32x32x2, 2 accumulators, 1 wave/SIMD", src, out, clk, 256, iters / 2, 4096.0);
  run<0, 1>("fp32 32x32x2, 1 accumulator, 1 wave/SIMD", src, out, clk, 256, iters / 2, 4096.0);
  CK(hipMemset(src, 0, 1 << 20));
  run<1>("bf16 32x32x16, zero operands, 2 waves/SIMD", src, out, clk, 512, iters, 32768.0);
  run<0>("fp32 32x32x2, zero operands, 2 waves/SIMD", src, out, clk, 512, iters / 2, 4096.0);
  return 0;
}
