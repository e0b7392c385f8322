// Are LDS reads (and the VALU instructions that consume them) correct while ANOTHER wave on the SIMD streams bf16 MFMAs?
// LDS holds lds[i] = i (as float); every read is checked against its index.  hipcc --offload-arch=gfx950 -O3 lds_hazard.hip -o lds_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256, 2) void burner(int iters, float* sink) {
  extern __shared__ float burner_lds[];  // 64 KB of dynamic LDS: at most two burner workgroups per CU, so the kernel under test is co-resident
  if (iters < 0) burner_lds[threadIdx.x] = 0.f;
  f32x16 acc[4]; for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a = __builtin_bit_cast(bf16x8, make_float4(1.37f * threadIdx.x, 0.11f, 3.3f, 0.7f)), b = __builtin_bit_cast(bf16x8, make_float4(0.37f, 2.11f, 1.3f, 0.9f));
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 3], 0, 0, 0);
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.678f) sink[0] = 1.f;
}
// MODE 0: ds_read_b128, compare   1: ds_read2_b32, compare   2: ds_read_b128 + ds_read2_b32 feeding v_pk_fma_f32 (sum compared with the
// sum of the expected values)   3: as 2 with 12 reads in flight (partial s_waitcnt lgkmcnt(N) like init5_kernel)
template <int MODE>
__global__ __launch_bounds__(256, 3) void test(int iters, unsigned* bad) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i4 = (tid * 7 + it * 13 + q * 257) & 2047;
        const float4 v = *reinterpret_cast<const float4*>(lds + i4 * 4);
        const float e = (float)(i4 * 4);
        nbad += (v.x != e) + (v.y != e + 1.f) + (v.z != e + 2.f) + (v.w != e + 3.f);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = (tid * 5 + it * 11 + q * 131) & 8063;
        const float a = lds[i], b = lds[i + 20];   // ds_read2_b32 offset0:0 offset1:20
        nbad += (a != (float)i) + (b != (float)(i + 20));
      }
    } else {
      f2 acc[8]; for (int k = 0; k < 8; ++k) acc[k] = f2{0.f, 0.f};
      float e0 = 0.f, e1 = 0.f;
      const int col = tid >> 4, c4 = tid & 15;
      const int xb = ((it * 3) & 63) + col;
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const int wi = ((it * 5 + d) & 127) * 64 + c4 * 4;
        const float4 w = *reinterpret_cast<const float4*>(lds + wi);
        f2 xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float* q = lds + xb + k * 20 + 2 * (d >> 1); xv[k] = f2{q[0], q[1]}; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xs = (d & 1) ? xv[k].y : xv[k].x;
          acc[k] = __builtin_elementwise_fma(f2{w.x, w.y}, f2{xs, xs}, acc[k]);
          const float ex = (float)(xb + k * 20 + 2 * (d >> 1) + (d & 1));
          if (k == 0) { e0 += 0.f; }
          (void)ex;
        }
      }
      // expected: acc[k].x = sum_d (wi_d) * x_{k,d},  acc[k].y = sum_d (wi_d + 1) * x_{k,d}  in the same fma order
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          const float wv = (float)(((it * 5 + d) & 127) * 64 + c4 * 4);
          const float xs = (float)(xb + k * 20 + 2 * (d >> 1) + (d & 1));
          r0 = __builtin_fmaf(wv, xs, r0); r1 = __builtin_fmaf(wv + 1.f, xs, r1);
        }
        nbad += (acc[k].x != r0) + ((acc[k].y != r1) << 16);
      }
      (void)e0; (void)e1;
    }
  }
  if (nbad) { atomicAdd(&bad[0], nbad & 0xffff); atomicAdd(&bad[1], nbad >> 16); atomicAdd(&bad[2], 1u); }
}
template <int MODE> void run(const char* name, hipStream_t s1, hipStream_t s2, unsigned* bad, float* sink) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&test<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  for (int co = 0; co < 2; ++co) {
    CK(hipMemset(bad, 0, 16));
    CK(hipDeviceSynchronize());
    if (co == 1) hipLaunchKernelGGL(burner, dim3(2048), dim3(256), 64 * 1024, s2, 3000, sink);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(test<MODE>, dim3(1024), dim3(256), 32768, s1, 2000, bad);
    CK(hipDeviceSynchronize());
    unsigned h[4]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("%-44s %-18s: mismatches %u (high halves %u), threads affected %u\n", name, co == 0 ? "alone" : "beside bf16 MFMA", h[0], h[1], h[2]);
  }
}
int main() {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&burner), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned* bad; float* sink; CK(hipMalloc(&bad, 16)); CK(hipMalloc(&sink, 16));
  run<0>("ds_read_b128 checked", s1, s2, bad, sink);
  run<1>("ds_read2_b32 checked", s1, s2, bad, sink);
  run<2>("LDS-fed v_pk_fma_f32 (init5-like)", s1, s2, bad, sink);
  return 0;
}
