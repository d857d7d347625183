// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of librmem_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit element type of activations, weights and the memory bank.  Every kernel translation unit is compiled twice:
// once with e16 = bfloat16 (the entry points of include/rmem.h under their plain names) and once with -DRMEM_F16, e16 = IEEE
// half, exported as <name>_f16 -- the operand type of the reference's --amp path (tools/eval.py:45-47, torch autocast) and of
// BASELINE cfg 5.  Accumulation, residual streams, softmax statistics and normalisation statistics are fp32 in both.
#ifdef RMEM_F16
typedef _Float16 e16;
#define RMEM_API(name) name##_f16
#define RMEM_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define RMEM_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define RMEM_MFMA_32x32x16_ASM "v_mfma_f32_32x32x16_f16"
#else
typedef __bf16 e16;
#define RMEM_API(name) name
#define RMEM_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define RMEM_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define RMEM_MFMA_32x32x16_ASM "v_mfma_f32_32x32x16_bf16"
#endif
typedef __attribute__((ext_vector_type(2))) e16 e16x2;
typedef __attribute__((ext_vector_type(4))) e16 e16x4;
typedef __attribute__((ext_vector_type(8))) e16 e16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// a.x * b.x + a.y * b.y + c in fp32 (v_dot2c_f32_bf16 / v_dot2c_f32_f16): one full-rate VALU instruction per 16-bit pair
__device__ __forceinline__ float rmem_dot2(e16x2 a, e16x2 b, float c) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return c;
#elif defined(RMEM_F16)
  return __builtin_amdgcn_fdot2(a, b, c, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
#endif
}
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

// LayerNorm over a 256-channel row held by one wave, 4 consecutive channels per lane (rmem_layernorm256 and the LSTT chain
// kernels share this one operation sequence -- every product / sum spelled out, so that -ffp-contract cannot contract the two
// call sites differently -- and fused and unfused routes round identically)
// (#pragma clang fp contract(off): with -ffp-contract=fast the compiler decides per call site whether a * b + c becomes an fma;
// two sites that must round identically may not leave that to it -- the IEEE-half build of the chain kernels differed from
// rmem_layernorm256 by one fp32 ulp in `v - sum * (1 / 256)` until every fma here was spelled out)
__device__ __forceinline__ float rmem_sum4(f32x4 v) {
#pragma clang fp contract(off)
  return ((v[0] + v[1]) + v[2]) + v[3];
}
__device__ __forceinline__ float rmem_sumsq4(f32x4 d) {
#pragma clang fp contract(off)
  return __builtin_fmaf(d[3], d[3], __builtin_fmaf(d[2], d[2], __builtin_fmaf(d[1], d[1], d[0] * d[0])));
}
__device__ __forceinline__ f32x4 rmem_ln_center(f32x4 v, float sum) {
#pragma clang fp contract(off)
  const float mean = sum * (1.f / 256.f);
  return f32x4{v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
}
__device__ __forceinline__ f32x4 rmem_ln_apply(f32x4 dv, float sumsq, float eps, f32x4 g, f32x4 bt) {
#pragma clang fp contract(off)
  const float var = sumsq * (1.f / 256.f);
  const float rstd = rsqrtf(var + eps);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = dv[j] * rstd;
    o[j] = __builtin_fmaf(t, g[j], bt[j]);
  }
  return o;
}
__device__ __forceinline__ f32x4 rmem_ln256_row(f32x4 v, f32x4 g, f32x4 bt, float eps) {
  const f32x4 dv = rmem_ln_center(v, wave_sum(rmem_sum4(v)));
  return rmem_ln_apply(dv, wave_sum(rmem_sumsq4(dv)), eps, g, bt);
}
// the same for NR rows at once (lane = the same 4 channels of every row): per row exactly rmem_ln256_row's operations, the NR
// butterfly reductions interleaved so that their cross-lane latencies overlap instead of adding up
template <int NR>
__device__ __forceinline__ void rmem_ln256_rows(f32x4 (&v)[NR], f32x4 g, f32x4 bt, float eps) {
  float s[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) s[r] = rmem_sum4(v[r]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < NR; ++r) s[r] += __shfl_xor(s[r], o, 64);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    v[r] = rmem_ln_center(v[r], s[r]);
    s[r] = rmem_sumsq4(v[r]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < NR; ++r) s[r] += __shfl_xor(s[r], o, 64);
#pragma unroll
  for (int r = 0; r < NR; ++r) v[r] = rmem_ln_apply(v[r], s[r], eps, g, bt);
}

// source coordinate of destination index d of a bilinear resize (PyTorch upsample_bilinear2d semantics)
// (explicit fmaf: every caller must get the same coordinates and blend whatever -ffp-contract chooses around the call)
__host__ __device__ __forceinline__ void rmem_src_coord(int d, int in, int out, int align, int& i0, int& i1, float& w1) {
  float s;
  if (align) s = out > 1 ? (float)d * ((float)(in - 1) / (float)(out - 1)) : 0.f;
  else s = __builtin_fmaxf(__builtin_fmaf((float)d + 0.5f, (float)in / (float)out, -0.5f), 0.f);
  i0 = (int)s < in - 1 ? (int)s : in - 1;
  i1 = i0 + 1 < in - 1 ? i0 + 1 : in - 1;
  w1 = s - (float)i0;
}
// bilinear blend of the four taps, one fixed operation sequence (rmem_bilinear_nhwc and the GEMM's resized residual)
__device__ __forceinline__ float rmem_bilerp(float a, float b, float c, float d, float wx, float wy) {
  const float ux = 1.f - wx, uy = 1.f - wy;
  const float top = __builtin_fmaf(b, wx, a * ux);
  const float bot = __builtin_fmaf(d, wx, c * ux);
  return __builtin_fmaf(bot, wy, top * uy);
}

// error plumbing shared by the C-ABI translation units (api.hip owns the storage)
extern "C" void rmem_set_error(const char* msg);
int rmem_check_launch(const char* what);
// opt-in launch timers (api.hip): channel 0 = memory-read attention, 1 = gated attention's value GEMM.  begin() returns a slot
// (>= 0) after recording the start event on the stream, or -1 when the channel is off, full, or the stream is being captured.
enum { RMEM_PROF_MEM_READ = 0, RMEM_PROF_GATED_PV = 1 };
int rmem_prof_begin(int channel, void* stream, double flops);
void rmem_prof_end(int channel, int slot, void* stream);

#define RMEM_REQUIRE(cond, msg)        \
  do {                                 \
    if (!(cond)) {                     \
      rmem_set_error(msg);             \
      return -1;                       \
    }                                  \
  } while (0)
