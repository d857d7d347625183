// Issue-rate micro-benchmark for the instructions of the memory-read tile loop on gfx950: cycles per wave-instruction
// with W waves per SIMD, measured with s_memtime around a long unrolled loop (one workgroup per CU, 256 * W threads).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
  bf8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * i); b[i] = (__bf16)(0.002f * i); }
  f16v acc0 = {}, acc1 = {};
  float l0 = 0.f, l1 = 0.f;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 32 v_exp_f32
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    } else if (MODE == 1) {   // 32 v_mul_f32
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x[i]));
    } else if (MODE == 2) {   // 16 v_cvt_pk_bf16_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[2 * i]) : "v"(x[2 * i + 1]));
    } else if (MODE == 3) {   // 16 v_dot2c_f32_bf16 (two chains)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(l0) : "v"(x[2 * i]), "v"(x[31]));
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(l1) : "v"(x[2 * i + 1]), "v"(x[31]));
      }
    } else if (MODE == 4) {   // 8 MFMA 32x32x16 bf16, two accumulators
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
    } else if (MODE == 5) {   // the tile's mix: 8 MFMA + 32 exp + 16 cvt + 16 dot2, independent streams
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[8 * i + j]));
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int j = 4; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[8 * i + j]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[8 * i + 2 * j]) : "v"(x[8 * i + 2 * j + 1]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(l0) : "v"(x[8 * i + 2 * j]), "v"(x[31]));
      }
    } else if (MODE == 6) {   // 32 v_exp_f16 (packed halves? no: scalar half exp)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
    } else if (MODE == 7) {   // 16 v_pk_mul_f32
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&x[2 * i]));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = l0 + l1;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ninstr, int waves) {
  const int blocks = 256, threads = 256 * waves, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  k<MODE><<<blocks, threads>>>(out, cyc, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += (double)v; c /= blocks;
  // s_memtime counts at a fixed 100 MHz on gfx9: report wall time per wave-instruction per SIMD instead, and the implied
  // cycles at the clock measured by a known full-rate instruction (v_mul_f32 = 4 cycles)
  const double ns_per_instr = ms * 1e6 / ((double)iters * ninstr * waves);
  printf("%-28s waves/SIMD %d: %8.3f ms  %7.3f ns per wave-instruction per SIMD  (timer ticks/iter %.1f)\n", name, waves, ms, ns_per_instr,
         c / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 3}) {
    run<1>("v_mul_f32 x32", 32, w);
    run<0>("v_exp_f32 x32", 32, w);
    run<6>("v_exp_f16 x32", 32, w);
    run<2>("v_cvt_pk_bf16_f32 x16", 16, w);
    run<3>("v_dot2c_f32_bf16 x16", 16, w);
    run<7>("v_pk_mul_f32 x16", 16, w);
    run<4>("mfma_32x32x16_bf16 x8", 8, w);
    run<5>("tile mix (8 mfma+64 valu)", 72, w);
  }
  return 0;
}
