// Does v_pk_fma_f32 give wrong results while ANOTHER wave on the same SIMD streams bf16 MFMAs?
// Test kernel: chains of packed FMAs (plain / op_sel broadcast of the low / high half) checked lane by lane against scalar v_fma_f32
// chains on the same data.  Burner: v_mfma_f32_32x32x16_bf16 back to back.  hipcc --offload-arch=gfx950 -O3 pkfma_hazard.hip -o pkfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void burner(int iters, float* sink) {
  f32x16 acc[4]; for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a = __builtin_bit_cast(bf16x8, make_float4(1.37f * threadIdx.x, 0.11f, 3.3f, 0.7f)), b = __builtin_bit_cast(bf16x8, make_float4(0.37f, 2.11f, 1.3f, 0.9f));
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 3], 0, 0, 0);
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.678f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256, 2) void burner_f32(int iters, float* sink) {
  f32x16 acc[4]; for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float a = 1.37f * threadIdx.x, b = 0.37f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 3], 0, 0, 0);
  if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.678f) sink[0] = 1.f;
}

// MODE 0: v_pk_fma_f32 plain   1: op_sel_hi:[1,0,1] (src1 low half for both)   2: op_sel:[0,1,0] op_sel_hi:[1,1,1] (src1 high half for both)
// 3: v_pk_mul_f32 + v_pk_add_f32 plain   4: scalar v_fma only (control)
template <int MODE>
__global__ __launch_bounds__(256, 2) void test(int iters, unsigned* bad, unsigned seed) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  auto rnd = [&](unsigned k) { unsigned h = (t * 2654435761u) ^ (k * 40503u + seed); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return (float)(h & 0xffff) / 65536.0f + 0.5f; };
  f2 acc[8], w[4], x[4]; float r0[8], r1[8];
  for (int k = 0; k < 8; ++k) { acc[k] = f2{rnd(k), rnd(k + 8)}; r0[k] = acc[k].x; r1[k] = acc[k].y; }
  for (int k = 0; k < 4; ++k) { w[k] = f2{rnd(20 + k) * 0.01f - 0.01f, rnd(30 + k) * 0.01f - 0.01f}; x[k] = f2{rnd(40 + k), rnd(50 + k)}; }
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f2 ww = w[k & 3], xx = x[(k + it) & 3];
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(ww), "v"(xx));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(ww), "v"(xx));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[k]) : "v"(ww), "v"(xx));
      if (MODE == 3) { f2 tmp; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(tmp) : "v"(ww), "v"(xx)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[k]) : "v"(tmp)); }
      if (MODE == 5 || MODE == 6) {  // alternating selectors: per group of 8 (MODE 5, as init5_kernel) or per instruction (MODE 6)
        const bool hi = (MODE == 5) ? (it & 1) : (k & 1);
        if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[k]) : "v"(ww), "v"(xx));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(ww), "v"(xx));
      }
      if (MODE == 4) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k].x) : "v"(ww.x), "v"(xx.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k].y) : "v"(ww.y), "v"(xx.y)); }
    }
    // the same arithmetic with scalar instructions, a little later (different co-issue alignment)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f2 ww = w[k & 3], xx = x[(k + it) & 3];
      const bool hi56 = (MODE == 5) ? (it & 1) : (k & 1);
      const float x0 = (MODE == 2 || ((MODE == 5 || MODE == 6) && hi56)) ? xx.y : xx.x, x1 = (MODE == 1 || ((MODE == 5 || MODE == 6) && !hi56)) ? xx.x : xx.y;
      if (MODE == 3) { float m0, m1; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(ww.x), "v"(x0)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(ww.y), "v"(x1));
                       asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0[k]) : "v"(m0)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1[k]) : "v"(m1)); }
      else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0[k]) : "v"(ww.x), "v"(x0)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1[k]) : "v"(ww.y), "v"(x1)); }
    }
    if ((it & 63) == 63) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (__float_as_uint(acc[k].x) != __float_as_uint(r0[k])) { nbad += 1; acc[k].x = r0[k]; }
        if (__float_as_uint(acc[k].y) != __float_as_uint(r1[k])) { nbad += 0x10000; acc[k].y = r1[k]; }
      }
    }
  }
  if (nbad) { atomicAdd(&bad[0], nbad & 0xffff); atomicAdd(&bad[1], nbad >> 16); atomicAdd(&bad[2], 1u); }
}

template <int MODE> void run(const char* name, hipStream_t s1, hipStream_t s2, unsigned* bad, float* sink) {
  for (int co = 0; co < 3; ++co) {
    CK(hipMemset(bad, 0, 16));
    CK(hipDeviceSynchronize());
    if (co == 1) hipLaunchKernelGGL(burner, dim3(512), dim3(256), 0, s2, 40000, sink);
    if (co == 2) hipLaunchKernelGGL(burner_f32, dim3(512), dim3(256), 0, s2, 20000, sink);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(test<MODE>, dim3(1024), dim3(256), 0, s1, 4096, bad, 17u + rep);
    CK(hipDeviceSynchronize());
    unsigned h[4]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("%-44s %-22s: wrong low halves %u, wrong high halves %u, threads affected %u\n", name, co == 0 ? "alone" : co == 1 ? "beside bf16 MFMA" : "beside fp32 MFMA", h[0], h[1], h[2]);
  }
}
int main() {
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned* bad; float* sink; CK(hipMalloc(&bad, 16)); CK(hipMalloc(&sink, 16));
  run<4>("scalar v_fma_f32 (control)", s1, s2, bad, sink);
  run<0>("v_pk_fma_f32", s1, s2, bad, sink);
  run<1>("v_pk_fma_f32 op_sel_hi:[1,0,1]", s1, s2, bad, sink);
  run<2>("v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,1,1]", s1, s2, bad, sink);
  run<3>("v_pk_mul_f32 + v_pk_add_f32", s1, s2, bad, sink);
  run<5>("v_pk_fma_f32 lo x8 / hi x8 alternating", s1, s2, bad, sink);
  run<6>("v_pk_fma_f32 lo / hi every instruction", s1, s2, bad, sink);
  return 0;
}
