// LDS poison: every workgroup fills its whole dynamic LDS allocation with a NaN pattern and leaves.  Run beside (another
// stream) or before a kernel under test: a kernel that USES LDS words it never wrote now produces NaNs instead of whatever
// its own previous launch left there.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC lds_poison.hip -o liblds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void poison(int words, unsigned pattern, int spin) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
  __syncthreads();
  for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(64);
}
extern "C" int lds_poison(int blocks, int bytes, unsigned pattern, int spin, void* stream) {
  static bool done = false;
  if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1; done = true; }
  hipLaunchKernelGGL(poison, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, bytes / 4, pattern, spin);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
