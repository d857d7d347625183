// Which VALU instructions co-execute with the matrix pipe on gfx950?  Per SIMD and iteration: 8 MFMA 32x32x16 (or none) plus
// N copies of one VALU instruction, interleaved in every wave; wall clock, 2048 workgroups of 256 threads.
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec2 coexec2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

template <int OP>
__device__ __forceinline__ void valu(float (&x)[32], int j) {
  if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
  if (OP == 1) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x[j]));
  if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[j]) : "v"(x[(j + 1) & 31]));
  if (OP == 3) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[j]) : "v"(x[(j + 1) & 31]), "v"(x[(j + 2) & 31]));
  if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j]));
  if (OP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(x[(j + 1) & 31]), "v"(x[(j + 2) & 31]));
  if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&x[j & 30]));
  if (OP == 7) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[j]) : "v"(x[(j + 1) & 31]));
  if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
  if (OP == 9) asm volatile("v_exp_f16 %0, %0" : "+v"(x[j]));
}

// MM: 0 none, 1 32x32x16 bf16 (8 per iteration), 2 16x16x32 bf16 (16 per iteration: same FLOPs)
template <int OP, int MM>
__global__ __launch_bounds__(256) void k(float* out, int iters, int nv) {
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f + 0.5f;
  bf8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * i); b[i] = (__bf16)(0.002f * i); }
  f16v acc0 = {}, acc1 = {};
  f4v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MM == 1) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
      if (MM == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) valu<OP>(x, 16 * (i & 1) + j);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  s += c0[0] + c1[1] + c2[2] + c3[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int MM>
float run(float* out) {
  const int blocks = 2048, iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<OP, MM><<<blocks, 256>>>(out, 10, 0);
  (void)hipEventRecord(e0);
  k<OP, MM><<<blocks, 256>>>(out, iters, 0);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)blocks * 4 * iters / 1024.0);     // ns per iteration and SIMD
}

template <int OP>
void line(const char* name, float* out, float mf1, float mf2) {
  const float v = run<OP, 0>(out), b1 = run<OP, 1>(out), b2 = run<OP, 2>(out);
  printf("%-20s x64: alone %6.1f ns (%.2f ns each) | with 8 mfma32 %6.1f ns (sum %6.1f, hidden %4.0f%%) | with 16 mfma16 %6.1f ns (sum %6.1f, hidden %4.0f%%)\n",
         name, v, v / 64, b1, v + mf1, 100 * (v + mf1 - b1) / (v < mf1 ? v : mf1), b2, v + mf2, 100 * (v + mf2 - b2) / (v < mf2 ? v : mf2));
}

int main() {
  float* out; (void)hipMalloc(&out, sizeof(float) * 2048 * 256);
  // MFMA alone: OP 99 = no VALU
  const float mf1 = run<99, 1>(out), mf2 = run<99, 2>(out);
  printf("8 x mfma_32x32x16 alone %.1f ns, 16 x mfma_16x16x32 alone %.1f ns (same FLOPs)\n", mf1, mf2);
  line<0>("v_exp_f32", out, mf1, mf2);
  line<9>("v_exp_f16", out, mf1, mf2);
  line<8>("v_rcp_f32", out, mf1, mf2);
  line<1>("v_mul_f32", out, mf1, mf2);
  line<4>("v_fma_f32", out, mf1, mf2);
  line<6>("v_pk_mul_f32", out, mf1, mf2);
  line<2>("v_cvt_pk_bf16_f32", out, mf1, mf2);
  line<3>("v_dot2c_f32_bf16", out, mf1, mf2);
  line<5>("v_perm_b32", out, mf1, mf2);
  line<7>("v_ldexp_f32", out, mf1, mf2);
  return 0;
}
