// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of librmem_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RMEM_WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// error plumbing shared by the C-ABI translation units (api.cpp owns the storage)
extern "C" void rmem_set_error(const char* msg);
int rmem_check_launch(const char* what);

#define RMEM_REQUIRE(cond, msg)        \
  do {                                 \
    if (!(cond)) {                     \
      rmem_set_error(msg);             \
      return -1;                       \
    }                                  \
  } while (0)
