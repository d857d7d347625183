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

// source coordinate of destination index d of a bilinear resize (PyTorch upsample_bilinear2d semantics)
// (explicit fmaf: every caller must get the same coordinates and blend whatever -ffp-contract chooses around the call)
__device__ __forceinline__ void rmem_src_coord(int d, int in, int out, int align, int& i0, int& i1, float& w1) {
  float s;
  if (align) s = out > 1 ? (float)d * ((float)(in - 1) / (float)(out - 1)) : 0.f;
  else s = fmaxf(__builtin_fmaf((float)d + 0.5f, (float)in / (float)out, -0.5f), 0.f);
  i0 = min((int)s, in - 1);
  i1 = min(i0 + 1, in - 1);
  w1 = s - (float)i0;
}
// bilinear blend of the four taps, one fixed operation sequence (rmem_bilinear_nhwc and the GEMM's resized residual)
__device__ __forceinline__ float rmem_bilerp(float a, float b, float c, float d, float wx, float wy) {
  const float ux = 1.f - wx, uy = 1.f - wy;
  const float top = __builtin_fmaf(b, wx, a * ux);
  const float bot = __builtin_fmaf(d, wx, c * ux);
  return __builtin_fmaf(bot, wy, top * uy);
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
