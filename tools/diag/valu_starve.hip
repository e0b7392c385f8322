// Does a co-resident fp32-MFMA loop starve another wave's VALU / VMEM issue on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// role 0 (blocks 0..255): "staging-like" work: NV dependent-free integer/float VALU ops + 12 global loads, timed.
// roles 1,2: MFMA loop (if mfma_on) for `iters` iterations.
template <int PRIO_STAGE, int PRIO_MFMA>
__global__ __launch_bounds__(256, 3) void k(const float* __restrict__ src, float* out, int iters, int mfma_on,
                                            unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int role = (blockIdx.x / 256) % 3;
  if (role == 0) {
    __builtin_amdgcn_s_setprio(PRIO_STAGE);
    // let the neighbours get going
    __builtin_amdgcn_s_sleep(100);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    unsigned idx = tid + blockIdx.x * 256;
    float4 v[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      // ~40 integer VALU ops of address math per load
      unsigned a = idx + u * 256;
#pragma unroll
      for (int q = 0; q < 10; ++q) { a = a * 1664525u + 1013904223u; a ^= a >> 7; a += (a << 3); }
      v[u] = *reinterpret_cast<const float4*>(src + (size_t)(a & 0xfffff) * 4);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      float x = v[u].x + v[u].y + v[u].z + v[u].w;
#pragma unroll
      for (int q = 0; q < 12; ++q) x = x * 1.0001f + 0.5f;  // transform-like VALU chain (48 ops / load)
      acc += x;
    }
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = t2 - t1; }
  } else if (mfma_on) {
    __builtin_amdgcn_s_setprio(PRIO_MFMA);
    f32x16 acc[2];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a0 = src[lane], a1 = src[lane + 64], b = src[lane + 128];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
  }
}

template <int PS, int PM> void run(const char* name, const float* src, float* out, unsigned long long* clk, int mfma_on) {
  hipLaunchKernelGGL((k<PS, PM>), dim3(768), dim3(256), 0, 0, src, out, 300, mfma_on, clk);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<PS, PM>), dim3(768), dim3(256), 0, 0, src, out, 300, mfma_on, clk);
  CK(hipDeviceSynchronize());
  unsigned long long h[512];
  CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> a, b;
  for (int i = 0; i < 256; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  printf("%-44s issue+addr (480 VALU + 12 loads): median %6llu cyc   loads-wait + 576 VALU: median %6llu cyc\n", name, a[128], b[128]);
}

int main() {
  float *src, *out; unsigned long long* clk;
  CK(hipMalloc(&src, (1 << 22) * 4 + 64)); CK(hipMalloc(&out, 768 * 256 * 4)); CK(hipMalloc(&clk, 8192));
  CK(hipMemset(src, 0, (1 << 22) * 4 + 64));
  run<0, 0>("alone (no MFMA neighbours)", src, out, clk, 0);
  run<0, 0>("2 MFMA neighbours, all prio 0", src, out, clk, 1);
  run<2, 0>("2 MFMA neighbours, staging prio 2", src, out, clk, 1);
  run<3, 0>("2 MFMA neighbours, staging prio 3", src, out, clk, 1);
  run<0, 0>("alone again", src, out, clk, 0);
  return 0;
}
