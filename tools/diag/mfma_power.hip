// Microbenchmark: what lowers the clock in the fp32-MFMA conv loop?  Variants of a K loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// V: 0 bare 32x32x2 (register operands), 1 bare 16x16x4, 2 = 0 + LDS A reads (b128), 3 = 2 + global B loads (L2)
template <int V>
__global__ __launch_bounds__(256, 3) void k(const float* __restrict__ w, float* out, int iters, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 12288; i += 256) lds[i] = (float)((i * 2654435761u) >> 8) * 1e-8f - 0.08f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float4 a0 = *reinterpret_cast<const float4*>(lds + lane * 68), a1 = *reinterpret_cast<const float4*>(lds + 4352 + lane * 68);
  float4 b = *reinterpret_cast<const float4*>(w + lane * 4);
  if (V == 1) {
    f32x4 acc[8];
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // 8 channels: 2 k-steps of 4 x 8 blocks = 16 MFMAs of 32 cycles = the 8 MFMAs of 64 cycles below
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x + q, b.x, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y + q, b.y, acc[q], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
  } else {
    f32x16 acc[2];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float* wp = w + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float4 n0 = a0, n1 = a1, nb = b;
        if (V >= 2) {
          n0 = *reinterpret_cast<const float4*>(lds + ((it * 8 + j) & 31) * 8 + (lane & 31) * 68 + 4 * (lane >> 5));
          n1 = *reinterpret_cast<const float4*>(lds + 4352 + ((it * 8 + j) & 31) * 8 + (lane & 31) * 68 + 4 * (lane >> 5));
        }
        if (V >= 3) nb = *reinterpret_cast<const float4*>(wp + (size_t)(((it * 8 + j) * 2 + (tid >> 6 & 1)) & 1023) * 256);
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a0 = n0; a1 = n1; b = nb;
      }
    }
    float s = 0.f;
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int V> void run(const char* name, const float* w, float* out, unsigned long long* clk, int blocks) {
  const int iters = 600;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 49152, 0, w, out, iters, clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 49152, 0, w, out, iters, clk);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // FLOPs per wave per iter: 8 j x 8 MFMA x 4096 (V==1: 8 x 16 x 2048): the same
  double fl = (double)blocks * 4 * iters * 8 * 8 * 4096.0 * reps;
  unsigned long long h[2048];
  CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
  double c = 0; for (int i = 0; i < 1024; ++i) c += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
  printf("%-34s %7.1f TFLOP/s  in-kernel clock %.0f MHz\n", name, fl / (ms * 1e-3) / 1e12, c / 1024);
}

int main(int argc, char** argv) {
  float *w, *out; unsigned long long* clk;
  CK(hipMalloc(&w, 1024 * 256 * 4 + 4096)); CK(hipMalloc(&out, 8192 * 256 * 4)); CK(hipMalloc(&clk, 8192 * 16));
  float* hw = (float*)malloc(1024 * 256 * 4 + 4096);
  for (int i = 0; i < 1024 * 256 + 1024; ++i) hw[i] = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(w, hw, 1024 * 256 * 4 + 4096, hipMemcpyHostToDevice));
  const int blocks = (argc > 1) ? atoi(argv[1]) : 768 * 4;
  run<0>("bare 32x32x2 f32", w, out, clk, blocks);
  run<1>("bare 16x16x4 f32", w, out, clk, blocks);
  run<2>("32x32x2 + LDS A reads", w, out, clk, blocks);
  run<3>("32x32x2 + LDS A + L2 B loads", w, out, clk, blocks);
  run<0>("bare 32x32x2 f32 (again)", w, out, clk, blocks);
  return 0;
}
