// When does a VALU wave get to issue beside an MFMA wave that executes ONE v_nop per 32 MFMAs (see mfma_coissue.hip)?
// The VALU wave stamps s_memtime every 8 v_fma; the MFMA wave stamps once per iteration (SMEM) in variant 1, never in 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int STAMP_A, int PB, int PA>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int role = blockIdx.x / 256;
  if (role == 0) {
    __builtin_amdgcn_s_setprio(PB);
    __builtin_amdgcn_s_sleep(100);
    float x[8];
    for (int q = 0; q < 8; ++q) x[q] = src[tid + q * 256];
    unsigned long long st[80];
#pragma unroll
    for (int rep = 0; rep < 80; ++rep) {
      st[rep] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = x[q] * 1.0001f + 0.5f;
      asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += x[q];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) for (int rep = 0; rep < 80; ++rep) clk[(size_t)blockIdx.x * 128 + rep] = st[rep];
  } else {
    float a0 = src[lane], a1 = src[lane + 64], b = src[lane + 128];
    __builtin_amdgcn_s_setprio(PA);
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    unsigned long long* myclk = clk + (size_t)blockIdx.x * 128;
    for (int it = 0; it < iters; ++it) {
      __builtin_amdgcn_sched_barrier(0);
      if (STAMP_A && tid == 0 && it < 100) myclk[it] = __builtin_amdgcn_s_memtime();
      asm volatile("v_nop");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(e & 1 ? a1 : a0, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
    if (!STAMP_A && tid == 0) { myclk[0] = 0; myclk[1] = __builtin_amdgcn_s_memtime(); }
  }
}
int main() {
  float *src, *out; unsigned long long* clk;
  CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 64 << 20)); CK(hipMalloc(&clk, 512 * 128 * 8));
  CK(hipMemset(src, 0, 1 << 20));
  static unsigned long long h[512 * 128];
  for (int variant = 0; variant < 4; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(clk, 0, 512 * 128 * 8));
      if (variant == 0) hipLaunchKernelGGL((k<0, 2, 0>), dim3(512), dim3(256), 0, 0, src, out, 60, clk);
      if (variant == 1) hipLaunchKernelGGL((k<0, 0, 0>), dim3(512), dim3(256), 0, 0, src, out, 60, clk);
      if (variant == 2) hipLaunchKernelGGL((k<0, 0, 3>), dim3(512), dim3(256), 0, 0, src, out, 60, clk);
      if (variant == 3) hipLaunchKernelGGL((k<0, 3, 0>), dim3(512), dim3(256), 0, 0, src, out, 60, clk);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
    const char* nm[] = {"VALU wave prio 2, MFMA wave prio 0", "both prio 0", "VALU 0, MFMA 3", "VALU 3, MFMA 0"};
    printf("variant %d (%s)\n", variant, nm[variant]);
    for (int wg = 0; wg < 256; wg += 97) {
      const unsigned long long* b = h + (size_t)wg * 128;
      printf(" VALU workgroup %d: 80 x 8 v_fma took %llu cycles; first steps:", wg, b[79] - b[0]);
      for (int r = 1; r < 12; ++r) printf(" %llu", b[r] - b[r - 1]);
      printf("\n");
    }
  }
  return 0;
}
