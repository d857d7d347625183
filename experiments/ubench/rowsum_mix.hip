// Which row-sum form costs the tile loop least on gfx950?  Per iteration and wave: 8 MFMA 32x32x16 + 32 v_exp_f32 + 16 v_cvt_pk_bf16
// (the memory-read tile without row sums) plus one of
//   mode 0: nothing            mode 1: 16 v_dot2c_f32_bf16        mode 2: 32 v_add_f32
//   mode 3: 8 v_mfma_f32_4x4x4_16b_bf16 with a ones B operand     mode 4: 4 v_mfma_f32_32x32x16 with a ones A operand (round 2, first half)
// Wall clock per tile and SIMD; 2048 workgroups x 256 threads and 768 x 256 (3 waves per SIMD, as the kernel runs).
// Build: hipcc --offload-arch=gfx950 -O3 -o rowsum_mix rowsum_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(4))) short s4v;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  bf8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * i); b[i] = (__bf16)(0.002f * i); }
  const bf4 ones4 = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
  f16v acc0 = {}, acc1 = {}, lacc = {};
  f4v s0 = {}, s1 = {};
  float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[8 * i + j]));
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(l0) : "v"(x[8 * i + j]));
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(l1) : "v"(x[8 * i + j + 1]));
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(l2) : "v"(x[8 * i + j + 2]));
          asm volatile("v_add_f32 %0, %1, %0" : "+v"(l3) : "v"(x[8 * i + j + 3]));
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[8 * i + 2 * j]) : "v"(x[8 * i + 2 * j + 1]));
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(l0) : "v"(x[8 * i + 2 * j]), "v"(x[31]));
          asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(l1) : "v"(x[8 * i + 2 * j + 2]), "v"(x[31]));
        }
      }
      if (MODE == 3) {
        bf4 p0, p1;
        __builtin_memcpy(&p0, &x[8 * i], 8);
        __builtin_memcpy(&p1, &x[8 * i + 4], 8);
        s0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4v, p0), __builtin_bit_cast(s4v, ones4), s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s4v, p1), __builtin_bit_cast(s4v, ones4), s1, 0, 0, 0);
      }
      if (MODE == 4) {
        bf8 p;
        __builtin_memcpy(&p, &x[8 * i], 16);
        lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, p, lacc, 0, 0, 0);
      }
    }
  }
  float s = l0 + l1 + l2 + l3 + s0[0] + s0[1] + s0[2] + s0[3] + s1[0] + s1[1] + s1[2] + s1[3];
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + lacc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int blocks) {
  const int iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 10);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)blocks * 4 * iters / 1024.0);
}

int main() {
  float* out; (void)hipMalloc(&out, sizeof(float) * 2048 * 256);
  for (int blocks : {2048, 768}) {
    for (int rep = 0; rep < 2; ++rep)
      printf("%4d workgroups: no row sum %.1f ns | 16 v_dot2c %.1f | 32 v_add_f32 %.1f | 8 mfma_4x4x4 %.1f | 4 mfma_32x32x16 (ones) %.1f   per tile and SIMD\n",
             blocks, run<0>(out, blocks), run<1>(out, blocks), run<2>(out, blocks), run<3>(out, blocks), run<4>(out, blocks));
  }
  return 0;
}
