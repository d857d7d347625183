// Normalisation / activation glue of the LSTT block and the FPN head (HBM-bound
// streaming kernels, 8-16 bytes per lane):
//   LayerNorm(256) with optional summed second input and optional "+ positional
//   embedding" second output      (layers/transformer.py:566-568, 574, 659-660, 683, 250-259)
//   GroupNorm + {none, ReLU, GELU} on NHWC   (layers/basic.py:27-35, 60-70; decoders/fpn.py:44-64)
//   depth-wise 5x5 convolution on NHWC       (layers/basic.py:19-25, 34)
//   e16 elementwise add                      (layers/transformer.py:279-285: curr_V + id_emb)
#include "common.h"
#include "../../include/rmem.h"

namespace {

// ------------------------------------------------------------------ LayerNorm
struct LnParams {
  const void* a; int a_f32; int lda;
  const void* b; int b_f32; int ldb;
  const float* gamma; const float* beta; float eps; int M;
  e16* y; int ldy; const float* pos; e16* ypos; int ldyp;
  float* yf; int ldyf;
};

__device__ __forceinline__ f32x4 load4(const void* base, int is_f32, long off) {
  if (is_f32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
  const e16x4 v = *reinterpret_cast<const e16x4*>(reinterpret_cast<const e16*>(base) + off);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// one wave per row of 256 channels, 4 channels per lane
__device__ __forceinline__ void layernorm256_body(const LnParams& p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int c0 = lane * 4;
  f32x4 v = load4(p.a, p.a_f32, (long)row * p.lda + c0);
  if (p.b) {
    const f32x4 w = load4(p.b, p.b_f32, (long)row * p.ldb + c0);
    v += w;
  }
  const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c0);
  const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c0);
  const f32x4 o = rmem_ln256_row(v, g, bt, p.eps);
  if (p.y) *reinterpret_cast<e16x4*>(p.y + (long)row * p.ldy + c0) = e16x4{(e16)o[0], (e16)o[1], (e16)o[2], (e16)o[3]};
  if (p.yf) *reinterpret_cast<f32x4*>(p.yf + (long)row * p.ldyf + c0) = o;
  if (p.ypos) {
    const f32x4 ps = *reinterpret_cast<const f32x4*>(p.pos + (long)row * 256 + c0);
    const f32x4 q = o + ps;
    *reinterpret_cast<e16x4*>(p.ypos + (long)row * p.ldyp + c0) = e16x4{(e16)q[0], (e16)q[1], (e16)q[2], (e16)q[3]};
  }
}

__global__ __launch_bounds__(256) void k_layernorm256(LnParams p) { layernorm256_body(p); }

// two independent LayerNorms of equal row count as one launch (blockIdx.y selects): norm4 of the key and of the value sum
// (layers/transformer.py:659-660)
struct LnPair { LnParams p[2]; };
__global__ __launch_bounds__(256) void k_layernorm256_pair(LnPair pp) { layernorm256_body(pp.p[blockIdx.y]); }

// generic channel count (Swin-B stages: 128 / 256 / 512 / 1024): one wave per row, C / 64 consecutive channels per lane, moved
// as 8- or 16-byte vectors (element-wise accesses ran at ~1 TB/s on the 512-wide rows of Swin stage 3)
template <int C>
__global__ __launch_bounds__(256) void k_layernorm_c(const void* a, int a_f32, int lda, const float* gamma, const float* beta, float eps,
                                                     int M, e16* y, int ldy, float* yf, int ldyf) {
  constexpr int NV = C / 64;                 // 2, 4, 8 or 16
  constexpr int FV = NV >= 4 ? 4 : 2;        // floats per fp32 vector access
  typedef __attribute__((ext_vector_type(FV))) float fvec;
  typedef __attribute__((ext_vector_type(FV))) e16 evec;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int c0 = lane * NV;
  float v[NV];
  if (a_f32) {
    const float* pa = reinterpret_cast<const float*>(a) + (long)row * lda + c0;
#pragma unroll
    for (int j = 0; j < NV; j += FV) {
      const fvec t = *reinterpret_cast<const fvec*>(pa + j);
#pragma unroll
      for (int i = 0; i < FV; ++i) v[j + i] = t[i];
    }
  } else {
    const e16* pa = reinterpret_cast<const e16*>(a) + (long)row * lda + c0;
#pragma unroll
    for (int j = 0; j < NV; j += FV) {
      const evec t = *reinterpret_cast<const evec*>(pa + j);
#pragma unroll
      for (int i = 0; i < FV; ++i) v[j + i] = (float)t[i];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += v[j];
  const float mean = wave_sum(s) * (1.f / C);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) { v[j] -= mean; ss += v[j] * v[j]; }
  const float rstd = rsqrtf(wave_sum(ss) * (1.f / C) + eps);
#pragma unroll
  for (int j = 0; j < NV; j += FV) {
    const fvec g = *reinterpret_cast<const fvec*>(gamma + c0 + j), bb = *reinterpret_cast<const fvec*>(beta + c0 + j);
    fvec o;
    evec oe;
#pragma unroll
    for (int i = 0; i < FV; ++i) { o[i] = v[j + i] * rstd * g[i] + bb[i]; oe[i] = (e16)o[i]; }
    if (y) *reinterpret_cast<evec*>(y + (long)row * ldy + c0 + j) = oe;
    if (yf) *reinterpret_cast<fvec*>(yf + (long)row * ldyf + c0 + j) = o;
  }
}

// Swin patch merging (encoders/swin/swin_transformer.py:336-356): gather the 2x2 neighbourhood of every output token
// ([even,even], [odd,even], [even,odd], [odd,odd] row/col order, zero beyond an odd border), LayerNorm over 4C, e16 out.
// One wave per output token.
template <int C>
__global__ __launch_bounds__(256) void k_patch_merge_ln(const float* x, int H, int W, const float* gamma, const float* beta, float eps, e16* y) {
  constexpr int C4 = 4 * C, NV = C4 / 64;
  const int lane = threadIdx.x & 63;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= Ho * Wo) return;
  x += (long)blockIdx.y * H * W * C;                      // blockIdx.y = image of a batch
  y += (long)blockIdx.y * Ho * Wo * C4;
  const int oy = tok / Wo, ox = tok - oy * Wo;
  const int c0 = lane * NV;                  // NV consecutive channels of the 4C vector: one source pixel (NV divides C)
  const int part = c0 / C, cc = c0 - part * C;
  const int sy = 2 * oy + (part & 1), sx = 2 * ox + (part >> 1);
  float v[NV];
  const bool ok = sy < H && sx < W;
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = ok ? x[((long)sy * W + sx) * C + cc + j] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += v[j];
  const float mean = wave_sum(s) * (1.f / C4);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) { v[j] -= mean; ss += v[j] * v[j]; }
  const float rstd = rsqrtf(wave_sum(ss) * (1.f / C4) + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) y[(long)tok * C4 + c0 + j] = (e16)(v[j] * rstd * gamma[c0 + j] + beta[c0 + j]);
}

// ------------------------------------------------------------------ add
__global__ __launch_bounds__(256) void k_add16(const e16* a, const e16* b, e16* y, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const e16x8 x = reinterpret_cast<const e16x8*>(a)[i];
    const e16x8 z = reinterpret_cast<const e16x8*>(b)[i];
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (e16)((float)x[j] + (float)z[j]);
    reinterpret_cast<e16x8*>(y)[i] = o;
  }
}

struct AddGroup { const e16* a[8]; const e16* b[8]; e16* y[8]; };
__global__ __launch_bounds__(256) void k_add16_grouped(AddGroup g, long n8) {
  const e16* a = g.a[blockIdx.y];
  const e16* b = g.b[blockIdx.y];
  e16* y = g.y[blockIdx.y];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const e16x8 x = reinterpret_cast<const e16x8*>(a)[i];
    const e16x8 z = reinterpret_cast<const e16x8*>(b)[i];
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (e16)((float)x[j] + (float)z[j]);
    reinterpret_cast<e16x8*>(y)[i] = o;
  }
}

// ------------------------------------------------------------------ GroupNorm (NHWC)
constexpr int GN_SPLITS = 64;

__device__ __forceinline__ void gn_load8(const e16* p, float (&f)[8]) {
  const e16x8 d = *reinterpret_cast<const e16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)d[j];
}
__device__ __forceinline__ void gn_load8(const float* p, float (&f)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
}

// partial (sum, sumsq) per (group, split); deterministic two-stage reduction
// blockIdx.z = image of a batch ([images][M][C]; statistics are per image and group)
template <typename TI>
__global__ __launch_bounds__(256) void k_gn_stats(const TI* x, int M, int C, int cpg, float* ws) {
  x += (long)blockIdx.z * M * C;
  ws += (long)blockIdx.z * gridDim.x * GN_SPLITS * 2;
  const int g = blockIdx.x, sp = blockIdx.y;
  const int vec = cpg / 8;                       // 16-byte pieces per pixel in this group
  const int rows_per = (M + GN_SPLITS - 1) / GN_SPLITS;
  const int r0 = sp * rows_per, r1 = min(M, r0 + rows_per);
  float s = 0.f, ss = 0.f;
  const long total = (long)(r1 - r0) * vec;
  for (long i = threadIdx.x; i < total; i += 256) {
    const int r = r0 + (int)(i / vec), v = (int)(i % vec);
    float d[8];
    gn_load8(x + (long)r * C + g * cpg + v * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += d[j]; ss += d[j] * d[j]; }
  }
  s = wave_sum(s); ss = wave_sum(ss);
  __shared__ float red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[(g * GN_SPLITS + sp) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ws[(g * GN_SPLITS + sp) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// (mean, rstd) of `groups` (<= 64) groups from the per-split partials, by 4 threads per group in a fixed summation order
// (every consumer of a workspace gets bit-identical statistics).  Call with all 256 threads, then __syncthreads().
__device__ __forceinline__ void gn_finalize(const float* ws, int groups, float n, float eps, float* s_mean, float* s_rstd) {
  const int g = threadIdx.x >> 2, q = threadIdx.x & 3;
  float s = 0.f, ss = 0.f;
  if (g < groups) {
    const float* w = ws + ((long)g * GN_SPLITS + q * (GN_SPLITS / 4)) * 2;
#pragma unroll
    for (int i = 0; i < GN_SPLITS / 4; ++i) { s += w[2 * i]; ss += w[2 * i + 1]; }
  }
  s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
  s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
  if (g < groups && q == 0) {
    const float mean = s / n;
    const float var = fmaxf(ss / n - mean * mean, 0.f);
    s_mean[g] = mean;
    s_rstd[g] = rsqrtf(var + eps);
  }
}

// Row-coalesced statistics for C = 128 .. 2048 (256 % (C/8) == 0): thread = (fixed 8 channels, every RPI-th row), so a wave
// reads whole contiguous rows; 4 loads in flight per thread; per-group sums through LDS in a fixed order.
// grid (GN_SPLITS, 1, images)
template <typename TI>
__global__ __launch_bounds__(256) void k_gn_stats_rows(const TI* x, int M, int C, int cpg, float* ws) {
  const int groups = C / cpg;
  x += (long)blockIdx.z * M * C;
  ws += (long)blockIdx.z * groups * GN_SPLITS * 2;
  const int sp = blockIdx.x;
  const int vpr = C / 8, rpi = 256 / vpr;
  const int c = threadIdx.x % vpr, rl = threadIdx.x / vpr;
  const int rows_per = (M + GN_SPLITS - 1) / GN_SPLITS;
  const int r0 = sp * rows_per, r1 = min(M, r0 + rows_per);
  float s = 0.f, ss = 0.f;
  for (int r = r0 + rl; r < r1; r += 4 * rpi) {
    float d[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + u * rpi;
      if (rr < r1) gn_load8(x + (long)rr * C + c * 8, d[u]);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[u][j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += d[u][j]; ss += d[u][j] * d[u][j]; }
  }
  __shared__ float sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < groups) {
    const int g = threadIdx.x, cpv = cpg / 8;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < rpi; ++l)
      for (int v = 0; v < cpv; ++v) { a += sh[0][l * vpr + g * cpv + v]; b += sh[1][l * vpr + g * cpv + v]; }
    ws[(g * GN_SPLITS + sp) * 2 + 0] = a;
    ws[(g * GN_SPLITS + sp) * 2 + 1] = b;
  }
}

// exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the e16 the
// callers round to): 2 transcendentals + ~12 plain VALU ops instead of the ~40 of erff.  For x < 0 the factor 1 + erf is formed
// directly as poly * exp(-z^2) (= erfc), without the cancellation of 1 - 0.9999...
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float erfc_z = q * t * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  return 0.5f * x * (x < 0.f ? erfc_z : 2.f - erfc_z);
}

template <typename TI>
__global__ __launch_bounds__(256) void k_gn_apply(const TI* x, int M, int C, int cpg, int groups, const float* ws,
                                                  const float* gamma, const float* beta, float eps, int act, e16* y) {
  x += (long)blockIdx.y * M * C;
  y += (long)blockIdx.y * M * C;
  ws += (long)blockIdx.y * groups * GN_SPLITS * 2;
  __shared__ float s_mean[64], s_rstd[64];
  gn_finalize(ws, groups, (float)M * (float)cpg, eps, s_mean, s_rstd);
  __syncthreads();
  const int vpr = C / 8;
  const long total = (long)M * vpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c0 = (int)(i % vpr) * 8;
    const int g = c0 / cpg;
    float d[8];
    gn_load8(x + i * 8, d);
    const float mean = s_mean[g], rstd = s_rstd[g];
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (d[j] - mean) * rstd * gamma[c0 + j] + beta[c0 + j];
      if (act == 1) f = fmaxf(f, 0.f);
      else if (act == 2) f = gelu_erf(f);
      o[j] = (e16)f;
    }
    reinterpret_cast<e16x8*>(y)[i] = o;
  }
}

// Row-coalesced apply for C = 128 .. 2048: thread = (fixed 8 channels, 8 rows); gamma / beta / statistics in registers, all 8
// loads issued before the first use.  grid (ceil(M / (8 * rpi)), images).  Same arithmetic as k_gn_apply (bit-identical).
template <typename TI>
__global__ __launch_bounds__(256) void k_gn_apply_rows(const TI* x, int M, int C, int cpg, int groups, const float* ws,
                                                       const float* gamma, const float* beta, float eps, int act, e16* y) {
  x += (long)blockIdx.y * M * C;
  y += (long)blockIdx.y * M * C;
  ws += (long)blockIdx.y * groups * GN_SPLITS * 2;
  __shared__ float s_mean[64], s_rstd[64];
  gn_finalize(ws, groups, (float)M * (float)cpg, eps, s_mean, s_rstd);
  __syncthreads();
  constexpr int UN = 8;
  const int vpr = C / 8, rpi = 256 / vpr;
  const int c = threadIdx.x % vpr, rl = threadIdx.x / vpr, c0 = c * 8;
  const float mean = s_mean[c0 / cpg], rstd = s_rstd[c0 / cpg];
  float gm[8], bt[8];
  {
    const f32x4 g0v = *reinterpret_cast<const f32x4*>(gamma + c0), g1v = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
    const f32x4 b0v = *reinterpret_cast<const f32x4*>(beta + c0), b1v = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[j] = g0v[j]; gm[4 + j] = g1v[j]; bt[j] = b0v[j]; bt[4 + j] = b1v[j]; }
  }
  const int rb = blockIdx.x * (UN * rpi) + rl;
  float d[UN][8];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int r = rb + u * rpi;
    if (r < M) gn_load8(x + (long)r * C + c0, d[u]);
  }
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int r = rb + u * rpi;
    if (r >= M) continue;
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (d[u][j] - mean) * rstd * gm[j] + bt[j];
      if (act == 1) f = fmaxf(f, 0.f);
      else if (act == 2) f = gelu_erf(f);
      o[j] = (e16)f;
    }
    *reinterpret_cast<e16x8*>(y + (long)r * C + c0) = o;
  }
}

// GroupNorm apply + activation + a narrow 1x1 convolution (N <= 16 outputs) in one pass: the segmentation head's last
// `conv_out(relu(gn(x)))` (decoders/fpn.py:62-66) without writing the normalised 128-channel map.  A workgroup takes HPASS x 64
// pixels: phase 1 normalises 64 of them into LDS (e16, the rounding the two-kernel route stores); phase 2 is a [16 pixels] x [16
// outputs] x [K = 128] product per wave on the matrix pipe (four v_mfma_f32_16x16x32; the weights are this wave's B fragments,
// read from global once) -- as 128-long VALU dot products it was 6 x off the kernel's byte count.  C = 128 only (the path's head);
// y is fp32 [M][ldy].
constexpr int HC = 128, HPIX = 64, HROW = HC + 8, HPASS = 2;
__global__ __launch_bounds__(256) void k_gn_apply_head(const e16* x, int M, int cpg, int groups, const float* ws, const float* gamma,
                                                       const float* beta, float eps, int act, const e16* w, const float* bias, int N,
                                                       float* y, int ldy) {
  x += (long)blockIdx.y * M * HC;
  y += (long)blockIdx.y * M * ldy;
  ws += (long)blockIdx.y * groups * GN_SPLITS * 2;
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ __attribute__((aligned(16))) e16 tile[HPIX * HROW];
  gn_finalize(ws, groups, (float)M * (float)cpg, eps, s_mean, s_rstd);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // B fragments: lane -> output n = lane & 15 (zero rows for n >= N), k = 32 ks + 8 (lane >> 4) .. + 7; w is [N][128] e16
  const int n = lane & 15, kq = lane >> 4;
  e16x8 bf[HC / 32];
#pragma unroll
  for (int ks = 0; ks < HC / 32; ++ks) {
    bf[ks] = e16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (n < N) bf[ks] = *reinterpret_cast<const e16x8*>(w + n * HC + 32 * ks + 8 * kq);
  }
  const float bn = (bias && n < N) ? bias[n] : 0.f;
  __syncthreads();
  const int c = tid & 15, rl = tid >> 4, c0 = c * 8;          // phase 1: 16 chunks of 8 channels per pixel, 16 pixels per pass
  const float mean = s_mean[c0 / cpg], rstd = s_rstd[c0 / cpg];
  float gm[8], bt[8];
  {
    const f32x4 g0v = *reinterpret_cast<const f32x4*>(gamma + c0), g1v = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
    const f32x4 b0v = *reinterpret_cast<const f32x4*>(beta + c0), b1v = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { gm[j] = g0v[j]; gm[4 + j] = g1v[j]; bt[j] = b0v[j]; bt[4 + j] = b1v[j]; }
  }
  for (int pass = 0; pass < HPASS; ++pass) {
    const int m0 = (blockIdx.x * HPASS + pass) * HPIX;
    if (m0 >= M) break;                                        // (workgroup-uniform)
    if (pass > 0) __syncthreads();                             // everybody has read the previous tile
    float d[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = m0 + rl + 16 * u;
      if (r < M) gn_load8(x + (long)r * HC + c0, d[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = m0 + rl + 16 * u;
      e16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
      if (r < M) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = (d[u][j] - mean) * rstd * gm[j] + bt[j];
          if (act == 1) f = fmaxf(f, 0.f);
          else if (act == 2) f = gelu_erf(f);
          o[j] = (e16)f;
        }
      }
      *reinterpret_cast<e16x8*>(&tile[(rl + 16 * u) * HROW + c0]) = o;
    }
    __syncthreads();
    // phase 2: wave -> pixels 16 wave .. + 15; A fragment: lane -> pixel row (lane & 15), k = 32 ks + 8 (lane >> 4) .. + 7
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < HC / 32; ++ks) {
      const e16x8 a = *reinterpret_cast<const e16x8*>(&tile[(16 * wave + n) * HROW + 32 * ks + 8 * kq]);
      acc = RMEM_MFMA_16x16x32(a, bf[ks], acc, 0, 0, 0);
    }
    // D: lane -> output column n, pixel rows 4 (lane >> 4) + i
    if (n < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * wave + 4 * kq + i;
        if (m < M) y[(long)m * ldy + n] = acc[i] + bn;
      }
    }
  }
}

// ------------------------------------------------------------------ depth-wise 5x5 (NHWC, pad 2)
// thread = (pixel, 8 channels); weights pre-transposed to [25][C] fp32
__global__ __launch_bounds__(256) void k_dwconv5(const e16* x, const float* w, e16* y, int H, int W, int C) {
  const int vpr = C / 8;
  const long total = (long)H * W * vpr;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c0 = (int)(i % vpr) * 8;
  const int pix = (int)(i / vpr);
  const int py = pix / W, px = pix - py * W;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int dx = 0; dx < 5; ++dx) {             // dx outer, dy inner: the summation order of the fused k_gn_dwconv5 (bit-identical)
    const int xx = px + dx - 2;
    if ((unsigned)xx >= (unsigned)W) continue;
    for (int dy = 0; dy < 5; ++dy) {
      const int yy = py + dy - 2;
      if ((unsigned)yy >= (unsigned)H) continue;
      const e16x8 d = *reinterpret_cast<const e16x8*>(x + ((long)yy * W + xx) * C + c0);
      const float* wt = w + (dy * 5 + dx) * C + c0;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] += (float)d[j] * w0[j]; acc[4 + j] += (float)d[4 + j] * w1[j]; }
    }
  }
  e16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (e16)acc[j];
  reinterpret_cast<e16x8*>(y)[i] = o;
}

// GroupNorm apply + activation + depth-wise 5x5 in one pass (basic.py:31-33: gn -> GELU -> conv of GNActDWConv2d): a tile of
// 8 x 16 pixels x 64 channels with its 2-pixel halo is normalised + activated ONCE into LDS (e16, the same rounding the
// two-kernel path stores), then the 25 taps read LDS.  ws holds the (sum, sumsq) partials of k_gn_stats.
constexpr int DT_H = 8, DT_W = 16, DT_C = 64, DT_HW = (DT_H + 4) * (DT_W + 4);
__global__ __launch_bounds__(256) void k_gn_dwconv5(const e16* x, const float* ws, const float* gamma, const float* beta, float eps,
                                                    int cpg, int act, const float* w, e16* y, int H, int W, int C, int M) {
  __shared__ __attribute__((aligned(16))) e16 tile[DT_HW * DT_C];
  __shared__ __attribute__((aligned(16))) float wl[25 * DT_C];
  __shared__ float s_mean[8], s_rstd[8];
  const int tid = threadIdx.x;
  x += (long)blockIdx.z * M * C;
  y += (long)blockIdx.z * M * C;
  ws += (long)blockIdx.z * (C / cpg) * GN_SPLITS * 2;
  const int tiles_x = (W + DT_W - 1) / DT_W;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int c0 = blockIdx.y * DT_C;
  const int ng = DT_C / cpg, g0 = c0 / cpg;
  gn_finalize(ws + (long)g0 * GN_SPLITS * 2, ng, (float)M * (float)cpg, eps, s_mean, s_rstd);
  for (int i = tid; i < 25 * DT_C; i += 256) wl[i] = w[(i / DT_C) * C + c0 + (i % DT_C)];
  __syncthreads();
  {
    // phase 1: thread = (8 channels, every 32nd halo pixel).  All global loads of a thread are issued before the first use
    // (a loop with the LDS store inside serialises one memory latency per pixel), gamma / beta live in registers.
    constexpr int NIT = (DT_HW * (DT_C / 8) + 255) / 256;
    const int c8 = tid & 7, cc = c0 + c8 * 8;
    float gm[8], bt[8];
    {
      const f32x4 g0v = *reinterpret_cast<const f32x4*>(gamma + cc), g1v = *reinterpret_cast<const f32x4*>(gamma + cc + 4);
      const f32x4 b0v = *reinterpret_cast<const f32x4*>(beta + cc), b1v = *reinterpret_cast<const f32x4*>(beta + cc + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { gm[j] = g0v[j]; gm[4 + j] = g1v[j]; bt[j] = b0v[j]; bt[4 + j] = b1v[j]; }
    }
    const float mean = s_mean[(c8 * 8) / cpg], rstd = s_rstd[(c8 * 8) / cpg];
    e16x8 raw[NIT];
    bool ok[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int pix = (tid >> 3) + 32 * k;
      const int hy = pix / (DT_W + 4), hx = pix - hy * (DT_W + 4);
      const int gy = ty * DT_H + hy - 2, gx = tx * DT_W + hx - 2;
      ok[k] = pix < DT_HW && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      raw[k] = e16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (ok[k]) raw[k] = *reinterpret_cast<const e16x8*>(x + ((long)gy * W + gx) * C + cc);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int pix = (tid >> 3) + 32 * k;
      e16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok[k]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = ((float)raw[k][j] - mean) * rstd * gm[j] + bt[j];
          if (act == 1) f = fmaxf(f, 0.f);
          else if (act == 2) f = gelu_erf(f);
          o[j] = (e16)f;
        }
      }
      if (pix < DT_HW) *reinterpret_cast<e16x8*>(&tile[pix * DT_C + c8 * 8]) = o;
    }
  }
  __syncthreads();
  // thread = (8 channels, one tile column, half of the tile rows): the 4 outputs of its column slide over 8 input rows, so each
  // LDS pixel is read once per dx and feeds up to 4 outputs (40 data + 50 weight reads per thread instead of 100 + 200)
  static_assert(DT_H == 8 && DT_W == 16 && DT_C == 64, "thread mapping below");
  const int ch8 = tid & 7, ox = (tid >> 3) & 15, half = tid >> 7;
  float acc[4][8];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
#pragma unroll 1
  for (int dx = 0; dx < 5; ++dx) {
    float wv[5][8];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      const float* wt = &wl[(dy * 5 + dx) * DT_C + ch8 * 8];
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { wv[dy][j] = w0[j]; wv[dy][4 + j] = w1[j]; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const e16x8 d = *reinterpret_cast<const e16x8*>(&tile[((4 * half + r) * (DT_W + 4) + ox + dx) * DT_C + ch8 * 8]);
      float df[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) df[j] = (float)d[j];
#pragma unroll
      for (int dy = 0; dy < 5; ++dy) {
        const int o = r - dy;
        if (o >= 0 && o < 4) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[o][j] += df[j] * wv[dy][j];
        }
      }
    }
  }
  const int gx = tx * DT_W + ox;
  if (gx < W) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gy = ty * DT_H + 4 * half + o;
      if (gy < H) {
        e16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (e16)acc[o][j];
        *reinterpret_cast<e16x8*>(y + ((long)gy * W + gx) * C + c0 + ch8 * 8) = ov;
      }
    }
  }
}

}  // namespace

extern "C" int RMEM_API(rmem_layernorm256)(const void* a, int a_is_f32, int lda, const void* b, int b_is_f32, int ldb,
                                 const float* gamma, const float* beta, float eps, int M, void* y_bf16, int ldy,
                                 const float* pos, void* ypos_bf16, int ldyp, float* y_f32, int ldyf, void* stream) {
  RMEM_REQUIRE(a && gamma && beta && M > 0, "rmem_layernorm256: null argument");
  RMEM_REQUIRE(y_bf16 || y_f32 || ypos_bf16, "rmem_layernorm256: no output requested");
  RMEM_REQUIRE(lda % 4 == 0 && (!b || ldb % 4 == 0) && (!y_bf16 || ldy % 4 == 0) && (!ypos_bf16 || ldyp % 4 == 0) &&
                   (!y_f32 || ldyf % 4 == 0), "rmem_layernorm256: leading dimensions must be multiples of 4");
  RMEM_REQUIRE(!ypos_bf16 || pos, "rmem_layernorm256: ypos needs pos");
  LnParams p{a, a_is_f32, lda, b, b_is_f32, ldb, gamma, beta, eps, M, (e16*)y_bf16, ldy, pos, (e16*)ypos_bf16, ldyp, y_f32, ldyf};
  hipLaunchKernelGGL(k_layernorm256, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_layernorm256");
}

extern "C" int RMEM_API(rmem_layernorm256_pair)(const void* a0, const void* b0, void* y0, const void* a1, const void* b1, void* y1,
                                      const float* gamma, const float* beta, float eps, int M, void* stream) {
  RMEM_REQUIRE(a0 && b0 && y0 && a1 && b1 && y1 && gamma && beta && M > 0, "rmem_layernorm256_pair: null argument");
  LnPair pp;
  pp.p[0] = LnParams{a0, 0, 256, b0, 0, 256, gamma, beta, eps, M, (e16*)y0, 256, nullptr, nullptr, 0, nullptr, 0};
  pp.p[1] = LnParams{a1, 0, 256, b1, 0, 256, gamma, beta, eps, M, (e16*)y1, 256, nullptr, nullptr, 0, nullptr, 0};
  hipLaunchKernelGGL(k_layernorm256_pair, dim3((M + 3) / 4, 2), dim3(256), 0, (hipStream_t)stream, pp);
  return rmem_check_launch("rmem_layernorm256_pair");
}

extern "C" int RMEM_API(rmem_layernorm)(const void* a, int a_is_f32, int lda, const float* gamma, const float* beta, float eps, int M, int C,
                              void* y_bf16, int ldy, float* y_f32, int ldyf, void* stream) {
  RMEM_REQUIRE(a && gamma && beta && M > 0 && (y_bf16 || y_f32), "rmem_layernorm: bad argument");
  RMEM_REQUIRE(lda % 4 == 0 && ldy % 4 == 0 && ldyf % 4 == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)y_bf16 % 8 == 0 &&
               (uintptr_t)y_f32 % 16 == 0 && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0,
               "rmem_layernorm: rows are moved as 8 / 16-byte vectors: leading dimensions must be multiples of 4, pointers 16-byte aligned");
  const dim3 g((M + 3) / 4), b(256);
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 128: hipLaunchKernelGGL(k_layernorm_c<128>, g, b, 0, s, a, a_is_f32, lda, gamma, beta, eps, M, (e16*)y_bf16, ldy, y_f32, ldyf); break;
    case 256: hipLaunchKernelGGL(k_layernorm_c<256>, g, b, 0, s, a, a_is_f32, lda, gamma, beta, eps, M, (e16*)y_bf16, ldy, y_f32, ldyf); break;
    case 512: hipLaunchKernelGGL(k_layernorm_c<512>, g, b, 0, s, a, a_is_f32, lda, gamma, beta, eps, M, (e16*)y_bf16, ldy, y_f32, ldyf); break;
    case 1024: hipLaunchKernelGGL(k_layernorm_c<1024>, g, b, 0, s, a, a_is_f32, lda, gamma, beta, eps, M, (e16*)y_bf16, ldy, y_f32, ldyf); break;
    default: rmem_set_error("rmem_layernorm: C must be 128, 256, 512 or 1024"); return -1;
  }
  return rmem_check_launch("rmem_layernorm");
}

extern "C" int RMEM_API(rmem_patch_merge_ln_images)(const float* x, int images, int H, int W, int C, const float* gamma, const float* beta, float eps,
                                          void* y_bf16, void* stream);
extern "C" int RMEM_API(rmem_patch_merge_ln)(const float* x, int H, int W, int C, const float* gamma, const float* beta, float eps, void* y_bf16, void* stream) {
  return RMEM_API(rmem_patch_merge_ln_images)(x, 1, H, W, C, gamma, beta, eps, y_bf16, stream);
}

extern "C" int RMEM_API(rmem_patch_merge_ln_images)(const float* x, int images, int H, int W, int C, const float* gamma, const float* beta, float eps,
                                          void* y_bf16, void* stream) {
  RMEM_REQUIRE(x && gamma && beta && y_bf16 && H > 0 && W > 0 && images >= 1, "rmem_patch_merge_ln: bad argument");
  const int M = ((H + 1) / 2) * ((W + 1) / 2);
  const dim3 g((M + 3) / 4, images), b(256);
  hipStream_t s = (hipStream_t)stream;
  if (C == 128) hipLaunchKernelGGL(k_patch_merge_ln<128>, g, b, 0, s, x, H, W, gamma, beta, eps, (e16*)y_bf16);
  else if (C == 256) hipLaunchKernelGGL(k_patch_merge_ln<256>, g, b, 0, s, x, H, W, gamma, beta, eps, (e16*)y_bf16);
  else { rmem_set_error("rmem_patch_merge_ln: C must be 128 or 256"); return -1; }
  return rmem_check_launch("rmem_patch_merge_ln");
}

extern "C" int RMEM_API(rmem_add16)(const void* a, const void* b, void* y, long long n, void* stream) {
  RMEM_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "rmem_add16: n must be a positive multiple of 8");
  const long n8 = n / 8;
  const int blocks = (int)min((long)2048, (n8 + 255) / 256);
  hipLaunchKernelGGL(k_add16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const e16*)a, (const e16*)b, (e16*)y, n8);
  return rmem_check_launch("rmem_add16");
}

extern "C" int RMEM_API(rmem_add16_grouped)(int n, const void* const* a, const void* const* b, void* const* y, long long count, void* stream) {
  RMEM_REQUIRE(n >= 1 && n <= 8 && a && b && y && count > 0 && count % 8 == 0, "rmem_add16_grouped: 1..8 problems, count a positive multiple of 8");
  AddGroup g = {};
  for (int i = 0; i < n; ++i) {
    RMEM_REQUIRE(a[i] && b[i] && y[i], "rmem_add16_grouped: null operand");
    g.a[i] = (const e16*)a[i]; g.b[i] = (const e16*)b[i]; g.y[i] = (e16*)y[i];
  }
  const long n8 = count / 8;
  const int blocks = (int)min((long)1024, (n8 + 255) / 256);
  hipLaunchKernelGGL(k_add16_grouped, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, g, n8);
  return rmem_check_launch("rmem_add16_grouped");
}

namespace {
template <typename TI>
void gn_launch_stats(const TI* x, int images, int M, int C, int cpg, float* ws, hipStream_t s) {
  const int vpr = C / 8;
  if (vpr <= 256 && 256 % vpr == 0)
    hipLaunchKernelGGL(k_gn_stats_rows<TI>, dim3(GN_SPLITS, 1, images), dim3(256), 0, s, x, M, C, cpg, ws);
  else
    hipLaunchKernelGGL(k_gn_stats<TI>, dim3(C / cpg, GN_SPLITS, images), dim3(256), 0, s, x, M, C, cpg, ws);
}
template <typename TI>
void gn_launch_apply(const TI* x, int images, int M, int C, int cpg, const float* ws, const float* gamma, const float* beta, float eps,
                     int act, e16* y, hipStream_t s) {
  const int vpr = C / 8, groups = C / cpg;
  if (vpr <= 256 && 256 % vpr == 0) {
    const int rows_per_wg = 8 * (256 / vpr);
    hipLaunchKernelGGL(k_gn_apply_rows<TI>, dim3((M + rows_per_wg - 1) / rows_per_wg, images), dim3(256), 0, s, x, M, C, cpg, groups, ws,
                       gamma, beta, eps, act, y);
  } else {
    const long total = (long)M * vpr;
    const int blocks = (int)min((long)2048, (total + 255) / 256);
    hipLaunchKernelGGL(k_gn_apply<TI>, dim3(blocks, images), dim3(256), 0, s, x, M, C, cpg, groups, ws, gamma, beta, eps, act, y);
  }
}
}  // namespace

#ifndef RMEM_F16
extern "C" size_t rmem_groupnorm_workspace_bytes(int groups) { return (size_t)groups * GN_SPLITS * 2 * sizeof(float); }
#endif

static int gn_check(const void* x, const void* y, const float* gamma, const float* beta, const float* ws, int groups, int C, int act,
                    int M, int images) {
  RMEM_REQUIRE(x && y && gamma && beta && ws, "rmem_groupnorm: null argument");
  RMEM_REQUIRE(groups >= 1 && groups <= 64 && C % groups == 0 && (C / groups) % 8 == 0,
               "rmem_groupnorm: channels per group must be a multiple of 8 and groups <= 64");
  RMEM_REQUIRE(act >= 0 && act <= 2 && M > 0 && images >= 1, "rmem_groupnorm: bad act / M / images");
  return 0;
}

extern "C" int RMEM_API(rmem_groupnorm_nhwc_images)(const void* x, int images, int M, int C, int groups, const float* gamma, const float* beta,
                                          float eps, int act, void* y, float* workspace, void* stream) {
  if (gn_check(x, y, gamma, beta, workspace, groups, C, act, M, images)) return -1;
  const int cpg = C / groups;
  hipStream_t s = (hipStream_t)stream;
  gn_launch_stats((const e16*)x, images, M, C, cpg, workspace, s);
  gn_launch_apply((const e16*)x, images, M, C, cpg, workspace, gamma, beta, eps, act, (e16*)y, s);
  return rmem_check_launch("rmem_groupnorm_nhwc_images");
}

extern "C" int RMEM_API(rmem_groupnorm_head_nhwc_images)(const void* x, int images, int M, int C, int groups, const float* gamma, const float* beta,
                                               float eps, int act, const void* w, const float* bias, int N, float* y, int ldy,
                                               float* workspace, void* stream) {
  if (gn_check(x, y, gamma, beta, workspace, groups, C, act, M, images)) return -1;
  RMEM_REQUIRE(C == 128 && w && N >= 1 && N <= 16 && ldy >= N, "rmem_groupnorm_head_nhwc: C must be 128, 1 <= N <= 16 <= ... ldy >= N");
  RMEM_REQUIRE((uintptr_t)w % 16 == 0, "rmem_groupnorm_head_nhwc: w must be 16-byte aligned");
  const int cpg = C / groups;
  hipStream_t s = (hipStream_t)stream;
  gn_launch_stats((const e16*)x, images, M, C, cpg, workspace, s);
  hipLaunchKernelGGL(k_gn_apply_head, dim3((M + HPIX * HPASS - 1) / (HPIX * HPASS), images), dim3(256), 0, s, (const e16*)x, M, cpg, groups, workspace, gamma,
                     beta, eps, act, (const e16*)w, bias, N, y, ldy);
  return rmem_check_launch("rmem_groupnorm_head_nhwc_images");
}

extern "C" int RMEM_API(rmem_groupnorm_nhwc)(const void* x, int M, int C, int groups, const float* gamma, const float* beta, float eps,
                                   int act, void* y, float* workspace, void* stream) {
  RMEM_REQUIRE(x && y && gamma && beta && workspace, "rmem_groupnorm_nhwc: null argument");
  RMEM_REQUIRE(groups >= 1 && groups <= 64 && C % groups == 0 && (C / groups) % 8 == 0,
               "rmem_groupnorm_nhwc: channels per group must be a multiple of 8 and groups <= 64");
  RMEM_REQUIRE(act >= 0 && act <= 2 && M > 0, "rmem_groupnorm_nhwc: bad act / M");
  const int cpg = C / groups;
  hipStream_t s = (hipStream_t)stream;
  gn_launch_stats((const e16*)x, 1, M, C, cpg, workspace, s);
  gn_launch_apply((const e16*)x, 1, M, C, cpg, workspace, gamma, beta, eps, act, (e16*)y, s);
  return rmem_check_launch("rmem_groupnorm_nhwc");
}

extern "C" int RMEM_API(rmem_groupnorm_f32_nhwc)(const float* x, int M, int C, int groups, const float* gamma, const float* beta, float eps,
                                       int act, void* y, float* workspace, void* stream) {
  RMEM_REQUIRE(x && y && gamma && beta && workspace, "rmem_groupnorm_f32_nhwc: null argument");
  RMEM_REQUIRE(groups >= 1 && groups <= 64 && C % groups == 0 && (C / groups) % 8 == 0,
               "rmem_groupnorm_f32_nhwc: groups must divide C into multiples of 8 channels (<= 64 groups)");
  RMEM_REQUIRE(act >= 0 && act <= 2 && M > 0, "rmem_groupnorm_f32_nhwc: bad act / M");
  const int cpg = C / groups;
  hipStream_t s = (hipStream_t)stream;
  gn_launch_stats(x, 1, M, C, cpg, workspace, s);
  gn_launch_apply(x, 1, M, C, cpg, workspace, gamma, beta, eps, act, (e16*)y, s);
  return rmem_check_launch("rmem_groupnorm_f32_nhwc");
}

extern "C" int RMEM_API(rmem_gn_act_dwconv5x5_nhwc_images)(const void* x, int images, int H, int W, int C, int groups, const float* gamma,
                                                 const float* beta, float eps, int act, const float* w_t, void* y, float* workspace,
                                                 void* stream);
extern "C" int RMEM_API(rmem_gn_act_dwconv5x5_nhwc)(const void* x, int H, int W, int C, int groups, const float* gamma, const float* beta,
                                          float eps, int act, const float* w_t, void* y, float* workspace, void* stream) {
  return RMEM_API(rmem_gn_act_dwconv5x5_nhwc_images)(x, 1, H, W, C, groups, gamma, beta, eps, act, w_t, y, workspace, stream);
}

static int gn_act_dwconv_launch(const void* x, int images, int H, int W, int C, int groups, const float* gamma, const float* beta, float eps,
                                int act, const float* w_t, void* y, float* workspace, void* stream, bool with_stats);
extern "C" int RMEM_API(rmem_gn_act_dwconv5x5_nhwc_images)(const void* x, int images, int H, int W, int C, int groups, const float* gamma,
                                                 const float* beta, float eps, int act, const float* w_t, void* y, float* workspace,
                                                 void* stream) {
  return gn_act_dwconv_launch(x, images, H, W, C, groups, gamma, beta, eps, act, w_t, y, workspace, stream, true);
}
// the same with the statistics partials ALREADY in the workspace ([image][group][64 splits][2]: (sum, sum of squares) of up to 64
// row ranges per group, unused splits zero) -- written by the producer of x (rmem_lstt_chain_b's gn_partial)
extern "C" int RMEM_API(rmem_gn_act_dwconv5x5_prestats_nhwc_images)(const void* x, int images, int H, int W, int C, int groups, const float* gamma,
                                                          const float* beta, float eps, int act, const float* w_t, void* y,
                                                          const float* stats, void* stream) {
  return gn_act_dwconv_launch(x, images, H, W, C, groups, gamma, beta, eps, act, w_t, y, const_cast<float*>(stats), stream, false);
}
static int gn_act_dwconv_launch(const void* x, int images, int H, int W, int C, int groups, const float* gamma, const float* beta, float eps,
                                int act, const float* w_t, void* y, float* workspace, void* stream, bool with_stats) {
  RMEM_REQUIRE(images >= 1, "rmem_gn_act_dwconv5x5_nhwc: images must be >= 1");
  RMEM_REQUIRE(x && y && gamma && beta && w_t && workspace && H > 0 && W > 0, "rmem_gn_act_dwconv5x5_nhwc: bad argument");
  RMEM_REQUIRE(groups >= 1 && groups <= 64 && C % groups == 0 && C % DT_C == 0, "rmem_gn_act_dwconv5x5_nhwc: C must be a multiple of 64 and of groups (<= 64)");
  const int cpg = C / groups;
  RMEM_REQUIRE(cpg % 8 == 0 && DT_C % cpg == 0, "rmem_gn_act_dwconv5x5_nhwc: channels per group must be 8, 16, 32 or 64");
  RMEM_REQUIRE(act >= 0 && act <= 2, "rmem_gn_act_dwconv5x5_nhwc: bad act");
  const int M = H * W;
  hipStream_t s = (hipStream_t)stream;
  if (with_stats) gn_launch_stats((const e16*)x, images, M, C, cpg, workspace, s);
  const dim3 grid(((W + DT_W - 1) / DT_W) * ((H + DT_H - 1) / DT_H), C / DT_C, images);
  hipLaunchKernelGGL(k_gn_dwconv5, grid, dim3(256), 0, s, (const e16*)x, workspace, gamma, beta, eps, cpg, act, w_t, (e16*)y, H, W, C, M);
  return rmem_check_launch("rmem_gn_act_dwconv5x5_nhwc");
}

extern "C" int RMEM_API(rmem_dwconv5x5_nhwc)(const void* x, const float* w_t, void* y, int H, int W, int C, void* stream) {
  RMEM_REQUIRE(x && w_t && y && H > 0 && W > 0 && C % 8 == 0, "rmem_dwconv5x5_nhwc: bad argument");
  const long total = (long)H * W * (C / 8);
  hipLaunchKernelGGL(k_dwconv5, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const e16*)x, w_t, (e16*)y, H, W, C);
  return rmem_check_launch("rmem_dwconv5x5_nhwc");
}
