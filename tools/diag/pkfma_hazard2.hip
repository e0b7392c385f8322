// Second attempt to isolate the packed-fp32 op_sel problem of init5_kernel: a test kernel shaped like it - 3 workgroups per CU,
// ~150 live VGPRs, a 12x3 window of pairs re-read from LDS with ds_read2_b32 per image, 25 taps x 16 packed FMAs alternating
// op_sel_hi:[1,0,1] / op_sel:[0,1,0] op_sel_hi:[1,1,1], weights from LDS (ds_read_b128) - checked against the same sums with scalar FMAs,
// alone and beside a bf16 MFMA burner on another stream.   hipcc --offload-arch=gfx950 -O3 pkfma_hazard2.hip -o pkfma_hazard2
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
__device__ __forceinline__ void fma_lo(f2& acc, f2 w, f2 x) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(x)); }
__device__ __forceinline__ void fma_hi(f2& acc, f2 w, f2 x) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(x)); }
// SCALAR = 1: the same arithmetic with v_fma_f32 (reference run)
template <int SCALAR>
__global__ __launch_bounds__(256, 3) void test(int images, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;               // [12][20] patch
  float* ws = lds + 240;         // [25][64] weights
  const int tid = threadIdx.x, c4 = tid & 15, col = tid >> 4;
  for (int i = tid; i < 25 * 64; i += 256) ws[i] = 0.01f * (float)((i * 37) % 101) - 0.5f;
  f2 a01[8], a23[8];
  for (int k = 0; k < 8; ++k) { a01[k] = f2{0.f, 0.f}; a23[k] = f2{0.f, 0.f}; }
  for (int n = 0; n < images; ++n) {
    __syncthreads();
    if (tid < 240) xs[tid] = 0.001f * (float)(((tid + 1) * (n * 7 + blockIdx.x + 3) * 2654435761u >> 20) & 1023) - 0.5f;
    __syncthreads();
    f2 xv[12][3];
#pragma unroll
    for (int r = 0; r < 12; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) { const float* q = xs + r * 20 + col + 2 * c; xv[r][c] = f2{q[0], c < 2 ? q[1] : 0.f}; }
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const float4 w = *reinterpret_cast<const float4*>(ws + (dy * 5 + dx) * 64 + c4 * 4);
        const f2 w01 = {w.x, w.y}, w23 = {w.z, w.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f2 x = xv[k + dy][dx >> 1];
          if (SCALAR) {
            const float xs_ = (dx & 1) ? x.y : x.x;
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a01[k].x) : "v"(w01.x), "v"(xs_)); asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a01[k].y) : "v"(w01.y), "v"(xs_));
            asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a23[k].x) : "v"(w23.x), "v"(xs_)); asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a23[k].y) : "v"(w23.y), "v"(xs_));
          } else if (dx & 1) { fma_hi(a01[k], w01, x); fma_hi(a23[k], w23, x); }
          else { fma_lo(a01[k], w01, x); fma_lo(a23[k], w23, x); }
        }
      }
  }
  float* o = out + ((size_t)blockIdx.x * 256 + tid) * 32;
  for (int k = 0; k < 8; ++k) { o[4 * k] = a01[k].x; o[4 * k + 1] = a01[k].y; o[4 * k + 2] = a23[k].x; o[4 * k + 3] = a23[k].y; }
}
int main() {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&burner), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int blocks = 1536, images = 64; const size_t n = (size_t)blocks * 256 * 32;
  float *ref, *out, *sink; CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sink, 16));
  const size_t ldsb = (240 + 25 * 64) * 4;
  hipLaunchKernelGGL(test<1>, dim3(blocks), dim3(256), ldsb, s1, images, ref);
  CK(hipDeviceSynchronize());
  float* href = (float*)malloc(n * 4); float* hout = (float*)malloc(n * 4);
  CK(hipMemcpy(href, ref, n * 4, hipMemcpyDeviceToHost));
  for (int co = 0; co < 2; ++co)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(out, 0, n * 4)); CK(hipDeviceSynchronize());
      if (co) hipLaunchKernelGGL(burner, dim3(2048), dim3(256), 64 * 1024, s2, 1500, sink);
      hipLaunchKernelGGL(test<0>, dim3(blocks), dim3(256), ldsb, s1, images, out);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hout, out, n * 4, hipMemcpyDeviceToHost));
      size_t bad = 0, lo = 0, lanes48 = 0;
      for (size_t i = 0; i < n; ++i) if (hout[i] != href[i]) { ++bad; if ((i & 1) == 0) ++lo; if ((((i / 32) % 256) & 63) >= 48) ++lanes48; }
      printf("packed op_sel FMAs (init5-shaped) %-18s run %d: %zu of %zu values differ from the scalar run (%zu low halves, %zu in lanes 48-63)\n",
             co ? "beside bf16 MFMA" : "alone", rep, bad, n, lo, lanes48);
    }
  return 0;
}
