// synthetic co-runners for a kernel under test (another stream): which shared resource makes it misbehave?
//   0: LDS hammer (64 KB, back-to-back ds_read_b128 / ds_write_b128)   1: bf16 MFMA burner (no LDS, no memory)
//   2: fp32 v_pk_fma burner   3: global-memory streamer   4: LDS hammer + MFMA
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC aggressors.hip -o libaggressors.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 2) void aggr(int mode, int iters, float* buf, size_t nbuf) {
  extern __shared__ float4 lds[];
  const int tid = threadIdx.x;
  float4 v = make_float4(tid, 1.f, 2.f, 3.f);
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a = __builtin_bit_cast(bf16x8, make_float4(1.37f * tid, 0.11f, 3.3f, 0.7f)), b = __builtin_bit_cast(bf16x8, make_float4(0.37f, 2.11f * tid, 1.3f, 0.9f));
  f2 p0 = {1.f, 2.f}, p1 = {1.0001f, 0.9999f}, p2 = {0.5f, 0.25f};
  if (mode == 0 || mode == 4) { for (int i = tid; i < 4096; i += 256) lds[i] = v; __syncthreads(); }
  for (int it = 0; it < iters; ++it) {
    if (mode == 0 || mode == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { float4 r = lds[(tid * 5 + q * 256 + it) & 4095]; v.x += r.y; lds[(tid + q * 256) & 4095] = v; }
    }
    if (mode == 1 || mode == 4) {
#pragma unroll
      for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    if (mode == 2) {
#pragma unroll
      for (int q = 0; q < 32; ++q) { p0 = __builtin_elementwise_fma(p0, p1, p2); p2 = __builtin_elementwise_fma(p2, p1, p0); }
    }
    if (mode == 3) {
      const size_t i = ((size_t)blockIdx.x * 256 + tid + (size_t)it * 65536 * 4) % (nbuf / 4);
      float4 r = reinterpret_cast<float4*>(buf)[i]; r.x += 1.f; reinterpret_cast<float4*>(buf)[i] = r;
    }
  }
  if (v.x + acc[0] + p0.x == 12345.678f) buf[0] = v.x;
}
extern "C" int aggressor(int mode, int blocks, int iters, float* buf, size_t nbuf, void* stream) {
  static bool done = false;
  if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&aggr), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1; done = true; }
  hipLaunchKernelGGL(aggr, dim3(blocks), dim3(256), 64 * 1024, (hipStream_t)stream, mode, iters, buf, nbuf);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
