// Feasibility of a Winograd F(2x2,3x3) K loop on the bf16 pipe with split operands: per 16 input channels a wave reads
// 2 rows x 4 columns x 8 channels of raw fp32 input from LDS (16 ds_read_b128), forms (B^T d B)[i][0..3] (32 packed
// vector instructions), splits the 32 values into three bf16 terms (144 vector instructions), loads 24 weight fragments
// (4 j x 2 nb x 3 splits) from L2 and issues 48 v_mfma_f32_32x32x16_bf16 into 8 accumulators (128 registers).
// Reports cycles per group, the clock, and the MFMA-only time (48 x 32 = 1536 cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ void split3(float a, float b, unsigned& s1, unsigned& s2, unsigned& s3) {
  f32x2 v = {a, b};
  s1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
  v -= f32x2{__uint_as_float(s1 << 16), __uint_as_float(s1 & 0xffff0000u)};
  s2 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
  v -= f32x2{__uint_as_float(s2 << 16), __uint_as_float(s2 & 0xffff0000u)};
  s3 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
template <int MODE>  // 0: everything; 1: no split (one bf16 term, 8 MFMAs per j... still 48 MFMAs on the same operand); 2: MFMAs only
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, const uint4* __restrict__ wts, float* out, int groups, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  for (int i = tid; i < 180 * 68; i += 256) lds[i] = src[i];
  __syncthreads();
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const float sg = (wave == 1) ? 1.f : -1.f;
  const int rA = wave == 0 ? 0 : wave == 2 ? 2 : 1, rB = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const int abase = ((lane & 31) >> 3) * 2 * 18 * 68 + ((lane & 7) * 2) * 68 + 8 * h;   // tile (ty, tx) -> pixel (2ty, 2tx)
  const uint4* wp = wts + (size_t)wave * 4 * 2 * 3 * 64 + lane;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint4 bq[2][3];
  for (int g = 0; g < groups; ++g) {
    const int coff = (g & 3) * 16;
    // raw rows rA, rB x 4 columns x 8 channels
    float4 dA[4][2], dB[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        dA[c][q] = *reinterpret_cast<const float4*>(lds + abase + (rA * 18 + c) * 68 + coff + 4 * q);
        dB[c][q] = *reinterpret_cast<const float4*>(lds + abase + (rB * 18 + c) * 68 + coff + 4 * q);
      }
    float e[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        e[c][4 * q] = fmaf(sg, dB[c][q].x, dA[c][q].x); e[c][4 * q + 1] = fmaf(sg, dB[c][q].y, dA[c][q].y);
        e[c][4 * q + 2] = fmaf(sg, dB[c][q].z, dA[c][q].z); e[c][4 * q + 3] = fmaf(sg, dB[c][q].w, dA[c][q].w);
      }
    const uint4* wg = wp + (size_t)(g & 3) * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int s = 0; s < 3; ++s) bq[nb][s] = wg[((j * 2 + nb) * 3 + s) * 64];
      uint4 v1, v2, v3;
      unsigned* p1 = reinterpret_cast<unsigned*>(&v1); unsigned* p2 = reinterpret_cast<unsigned*>(&v2); unsigned* p3 = reinterpret_cast<unsigned*>(&v3);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 2 * q, c1 = 2 * q + 1;
        const float va = (j == 0) ? e[0][c0] - e[2][c0] : (j == 1) ? e[1][c0] + e[2][c0] : (j == 2) ? e[2][c0] - e[1][c0] : e[1][c0] - e[3][c0];
        const float vb = (j == 0) ? e[0][c1] - e[2][c1] : (j == 1) ? e[1][c1] + e[2][c1] : (j == 2) ? e[2][c1] - e[1][c1] : e[1][c1] - e[3][c1];
        if (MODE == 0) split3(va, vb, p1[q], p2[q], p3[q]);
        else { p1[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{va, vb}), bf2)); p2[q] = p1[q]; p3[q] = p1[q]; }
      }
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, v1), a2 = __builtin_bit_cast(bf16x8, v2), a3 = __builtin_bit_cast(bf16x8, v3);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, bq[nb][0]), b2 = __builtin_bit_cast(bf16x8, bq[nb][1]), b3 = __builtin_bit_cast(bf16x8, bq[nb][2]);
        f32x16& c = acc[j * 2 + nb];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a1, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a3, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a2, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, c, 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int MODE> void run(const char* name, const float* src, const uint4* w, float* out, unsigned long long* clk, int blocks, int groups) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 180 * 68 * 4, 0, src, w, out, groups, clk); CK(hipDeviceSynchronize()); }
  std::vector<unsigned long long> h(blocks * 2);
  CK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
  std::vector<double> cyc, ghz;
  for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[2 * b] / groups); ghz.push_back((double)h[2 * b] / h[2 * b + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  printf("%-44s %d workgroups: %6.0f cycles per 16-channel group (MFMA alone: 1536), clock %.2f GHz -> %.2f us per 128-pixel tile K loop\n", name, blocks, cyc[blocks / 2],
         ghz[blocks / 2], cyc[blocks / 2] * 4 / ghz[blocks / 2] / 1e3);
}
int main() {
  float *src, *out; uint4* w; unsigned long long* clk;
  CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&clk, 1024 * 16)); CK(hipMalloc(&w, 4 << 20));
  std::vector<float> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((float)rand() / (float)RAND_MAX - 0.5f) * 2.f;
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<unsigned short> hw(2 << 20);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  const int groups = 400;
  run<0>("transform + 3-way split + 48 MFMAs", src, w, out, clk, 256, groups);
  run<0>("the same, two workgroups per CU", src, w, out, clk, 512, groups);
  run<1>("transform, one bf16 term (no split VALU)", src, w, out, clk, 512, groups);
  return 0;
}
