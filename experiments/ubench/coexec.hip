// Can the matrix pipe and the VALU of one SIMD work at the same time on gfx950, and does it matter whether the two instruction
// streams come from the same wave?  Wall-clock only (hipEvents), grid = 2048 workgroups of 256 threads (every SIMD holds
// several waves), the per-tile instruction mix of the memory-read kernel: 8 MFMA 32x32x16 + 32 v_exp_f32 + 16 v_cvt_pk + 16 v_dot2c.
//   mode 0: MFMAs only      mode 1: VALU only      mode 2: both, interleaved in every wave
//   mode 3: even waves MFMAs only, odd waves VALU only (twice the iterations each: same total work as mode 2)
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

__device__ __forceinline__ void valu_part(float (&x)[32], float& l0, int i) {
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[8 * i + j]));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[8 * i + 2 * j]) : "v"(x[8 * i + 2 * j + 1]));
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(l0) : "v"(x[8 * i + 2 * j]), "v"(x[31]));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  bf8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * i); b[i] = (__bf16)(0.002f * i); }
  f16v acc0 = {}, acc1 = {};
  float l0 = 0.f;
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  const int n = MODE == 3 ? 2 * iters : iters;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (do_m) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
      if (do_v) valu_part(x, l0, i);
    }
  }
  float s = l0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int blocks = 2048, iters = 1000;
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  const double tiles_per_simd = (double)blocks * 4 * iters / 1024.0;     // wave-tiles per SIMD
  for (int rep = 0; rep < 2; ++rep) {
    const float m = run<0>(out, blocks, iters), v = run<1>(out, blocks, iters), b = run<2>(out, blocks, iters), s = run<3>(out, blocks, iters);
    printf("per tile and SIMD: MFMA only %.1f ns   VALU only %.1f ns   both in every wave %.1f ns   split by wave %.1f ns   (sum %.1f, max %.1f)\n",
           m * 1e6 / tiles_per_simd, v * 1e6 / tiles_per_simd, b * 1e6 / tiles_per_simd, s * 1e6 / tiles_per_simd,
           (m + v) * 1e6 / tiles_per_simd, (m > v ? m : v) * 1e6 / tiles_per_simd);
  }
  return 0;
}
