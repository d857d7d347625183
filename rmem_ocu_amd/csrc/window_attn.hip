// Swin window attention (W-MSA / SW-MSA) for gfx950, head dim 32, 7x7 windows.
//
// Replaces encoders/swin/swin_transformer.py:156-195 (WindowAttention.forward, minus the qkv / proj linears, which are
// rmem_conv2d_nhwc calls) together with the token plumbing of SwinTransformerBlock.forward (263-305): zero padding to a
// multiple of the window, cyclic shift, window partition, the shifted-window mask, window reverse, un-shift and crop are
// all index arithmetic here -- no padded / rolled / partitioned copy of the feature map is ever written.
//
// One workgroup (2 waves) per (window, head).  The 49 tokens of a window are padded to 64 keys / 2 x 32 queries:
//   S^T = K . Q^T        v_mfma_f32_32x32x16_bf16, query on the lane; the accumulator starts from the pre-scaled
//                        (relative-position bias + shift mask) tile, so bias and mask cost no VALU work;
//   softmax over 64 keys  in registers (keys >= 49 start at -1e30);
//   O^T = V^T . P^T      V staged row-major in LDS, read transposed (ds_read_b64_tr_b16); P never leaves registers.
// Tokens outside the feature map (the padding the reference adds AFTER norm1) carry q = k = v = qkv bias, exactly as a
// Linear applied to a zero token does.
#include "common.h"
#include "../../include/rmem.h"

namespace {

constexpr int D = 32, WS = 7, NT = 49;
constexpr float NEG_BIG = -1.0e30f;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WinParams {
  const e16* qkv;        // [H*W][3C]: q | k | v
  const float* qkv_bias;  // [3C]
  const float* table;     // [4 window types][heads][49][49], (rel-pos bias + mask) * log2(e)
  e16* out;              // [H*W][C]
  int H, W, C, heads, shift, nwx, nwy;
  float qscale;           // log2(e) / sqrt(32)
};

__device__ __forceinline__ int kswz(int row, int chunk) { return row * D + ((chunk ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ float pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// token index of window position i (0..48) in window (wy, wx), or -1 if it lies in the zero padding
__device__ __forceinline__ int win_token(const WinParams& p, int wy, int wx, int i) {
  const int py = i / WS, px = i - py * WS;
  const int Hp = p.nwy * WS, Wp = p.nwx * WS;
  int y = wy * WS + py + p.shift, x = wx * WS + px + p.shift;     // shifted[i] = x[(i + shift) mod Hp]
  if (y >= Hp) y -= Hp;
  if (x >= Wp) x -= Wp;
  return (y < p.H && x < p.W) ? y * p.W + x : -1;
}

__device__ __forceinline__ e16x8 load8(const WinParams& p, int tok, int col) {
  if (tok >= 0) return *reinterpret_cast<const e16x8*>(p.qkv + (long)tok * 3 * p.C + col);
  e16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (e16)p.qkv_bias[col + j];
  return v;
}

__global__ __launch_bounds__(128) void k_window_attn(WinParams pin) {
  WinParams p = pin;
  __shared__ __attribute__((aligned(16))) e16 Ks[64 * D];
  __shared__ __attribute__((aligned(16))) e16 Vs[64 * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int win = blockIdx.x, head = blockIdx.y;
  p.qkv += (long)blockIdx.z * p.H * p.W * 3 * p.C;       // blockIdx.z = image of a batch (rows [image][token])
  p.out += (long)blockIdx.z * p.H * p.W * p.C;
  const int wy = win / p.nwx, wx = win - wy * p.nwx;
  const int type = p.shift > 0 ? ((wy == p.nwy - 1) ? 2 : 0) + ((wx == p.nwx - 1) ? 1 : 0) : 0;

  // ---- stage K and V of the window's 64 (49 real) keys ----
  const e16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + i * 128;
    const int key = id >> 2, chunk = id & 3;
    e16x8 kv = zero8, vv = zero8;
    if (key < NT) {
      const int tok = win_token(p, wy, wx, key);
      kv = load8(p, tok, p.C + head * D + chunk * 8);
      vv = load8(p, tok, 2 * p.C + head * D + chunk * 8);
    }
    *reinterpret_cast<e16x8*>(&Ks[kswz(key, chunk)]) = kv;
    *reinterpret_cast<e16x8*>(&Vs[key * D + chunk * 8]) = vv;
  }

  // ---- Q^T fragment of this wave's 32 queries ----
  const int qi = wave * 32 + lq;
  const int qtok = qi < NT ? win_token(p, wy, wx, qi) : -1;
  e16x8 qf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const e16x8 raw = qi < NT ? load8(p, qtok, head * D + 16 * s + 8 * lh) : zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[s][j] = (e16)((float)raw[j] * p.qscale);
  }
  __syncthreads();

  // ---- S^T for the two 32-key blocks, starting from the bias + mask tile ----
  const float* trow = p.table + (((long)type * p.heads + head) * NT + min(qi, NT - 1)) * NT;
  f32x16 sacc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      sacc[b][r] = k < NT ? trow[k] : NEG_BIG;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const e16x8 a = *reinterpret_cast<const e16x8*>(&Ks[kswz(b * 32 + lq, 2 * s + lh)]);
      sacc[b] = RMEM_MFMA_32x32x16(a, qf[s], sacc[b], 0, 0, 0);
    }
  }
  float m = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
  for (int r = 1; r < 16; ++r) m = fmaxf(fmaxf(m, sacc[0][r]), sacc[1][r]);
  m = pair_max(m);
  float l = 0.f;
  e16x8 pb[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sacc[b][r] - m);
      l += e;
      pb[b][r >> 3][r & 7] = (e16)e;
    }
  l += __shfl_xor(l, 32, 64);

  // ---- O^T = V^T . P^T ----
  const int tr_off = ((4 * lh + ((lane & 15) >> 2)) * D) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const e16* vb = &Vs[(b * 32 + 16 * s) * D + tr_off];
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)vb);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + 8 * D));
      const __attribute__((ext_vector_type(8))) short a16 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc = RMEM_MFMA_32x32x16(__builtin_bit_cast(e16x8, a16), pb[b][s], oacc, 0, 0, 0);
    }

  if (qtok >= 0) {
    const float inv = 1.f / l;
    e16* o = p.out + (long)qtok * p.C + head * D + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g)    // C/D rows (r&3) + 8(r>>2) + 4h -> d = 8g + 4h + (0..3)
      *reinterpret_cast<e16x4*>(o + 8 * g) = e16x4{(e16)(oacc[4 * g] * inv), (e16)(oacc[4 * g + 1] * inv),
                                                     (e16)(oacc[4 * g + 2] * inv), (e16)(oacc[4 * g + 3] * inv)};
  }
}

}  // namespace

extern "C" int RMEM_API(rmem_window_attn_images)(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int images,
                                       int H, int W, int C, int heads, int shift, void* stream);
extern "C" int RMEM_API(rmem_window_attn)(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int H, int W,
                                int C, int heads, int shift, void* stream) {
  return RMEM_API(rmem_window_attn_images)(qkv, qkv_bias, bias_mask_table, out, 1, H, W, C, heads, shift, stream);
}

extern "C" int RMEM_API(rmem_window_attn_images)(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int images,
                                       int H, int W, int C, int heads, int shift, void* stream) {
  RMEM_REQUIRE(qkv && qkv_bias && bias_mask_table && out && images >= 1, "rmem_window_attn: null argument");
  RMEM_REQUIRE(H > 0 && W > 0 && heads >= 1 && C == heads * D, "rmem_window_attn: C must equal heads * 32");
  RMEM_REQUIRE(shift == 0 || shift == WS / 2, "rmem_window_attn: shift must be 0 or 3");
  WinParams p;
  p.qkv = (const e16*)qkv; p.qkv_bias = qkv_bias; p.table = bias_mask_table; p.out = (e16*)out;
  p.H = H; p.W = W; p.C = C; p.heads = heads; p.shift = shift;
  p.nwy = (H + WS - 1) / WS; p.nwx = (W + WS - 1) / WS;
  p.qscale = 1.4426950408889634f / sqrtf((float)D);
  hipLaunchKernelGGL(k_window_attn, dim3(p.nwy * p.nwx, heads, images), dim3(128), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_window_attn");
}
