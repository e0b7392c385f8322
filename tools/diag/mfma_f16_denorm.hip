// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs?  A = 2^-20 (subnormal, bits 0x0010), B = 2^10: every output is
// 16 * 2^-10 = 0.015625 if subnormals are honoured, 0 if they are flushed.  Same for a subnormal in B, and for v_cvt results.
// hipcc --offload-arch=gfx950 -O2 tools/diag/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short abits, unsigned short bbits) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, abits); b[i] = __builtin_bit_cast(_Float16, bbits); }
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  out[threadIdx.x] = acc[0];
  // conversion: fp32 -> fp16 of a value in fp16's subnormal range, and back
  float x = 3.1e-6f * (1 + threadIdx.x * 0.f);
  _Float16 hx = (_Float16)x;
  out[64 + threadIdx.x] = (float)hx;
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  float h[128];
  k<<<1, 64>>>(d, 0x0010, 0x6400); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("A subnormal (2^-20) x B 2^10: %g (expect 0.015625)\n", h[0]);
  k<<<1, 64>>>(d, 0x6400, 0x0010); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("A 2^10 x B subnormal (2^-20): %g (expect 0.015625)\n", h[0]);
  k<<<1, 64>>>(d, 0x0001, 0x7800); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("A min subnormal (2^-24) x B 2^15: %g (expect %g)\n", h[0], 16 * 32768.0 / 16777216.0);
  printf("fp32 3.1e-6 -> fp16 -> fp32: %g (fp16 subnormal spacing 5.96e-8)\n", h[64]);
  return 0;
}
