// ResNet bottleneck tail chained into the NEXT block's first convolution, one launch (gfx950, wave64).
//
//   y  = relu(b . W3^T + bias3 + res)        last 1x1 conv of a bottleneck + identity shortcut      (encoders/resnet.py:62-68)
//   a2 = relu(y . W1'^T + bias1')            first 1x1 conv of the following block                  (encoders/resnet.py:48-50)
//
// Why: at 16 images per launch the 256-channel stride-4 activations are 211 MB each, more than the Infinity Cache holds, and the
// layer-1 convolutions run at the HBM rate (conv3 writes y, the next conv1 reads all of it back: 211 of the 739 MB the two
// launches move).  Here a workgroup owns 64 output pixels with ALL 256 channels of y: the y tile is stored to HBM (the block after
// next needs it as its shortcut) and, rounded exactly as stored, stays in LDS as the A operand of the next conv1, whose
// weights (N2 x 256) stream through a small ring.  y is never read back.  The dual form (x2 != NULL) is the block with the
// strided 1x1 shortcut: y = relu([b | x2 sampled at stride2] . W3cat^T + bias3), K = K1 + Cin2 (rmem_conv1x1_dual_nhwc).
//
// Results are BIT-IDENTICAL to rmem_conv2d_nhwc(+residual) followed by rmem_conv2d_nhwc: the same 16x16x32 MFMA chains in
// the same k order, the same epilogue arithmetic (accumulator + bias, + residual, ReLU, one rounding) -- tests/test_hip_ops.py.
//
// LDS (73,984 B -> two workgroups per CU), 256 threads = 2 x 2 waves:
//   R1 32 KB  W3 panel of one k-step [256 rows][64 k]; after the first GEMM the y tile as four k-step panels [64 rows][64 k]
//   R2 8.25 KB A panel of one k-step [64 rows][64 k]; afterwards the fp32 staging of the epilogues, 16 rows x 128 columns a pass
//   R3 32 KB  W1' k-step panels [N2 rows][64 k]: all four (N2 = 64) or a ring of two (N2 = 128)
// Everything a workgroup needs from memory is requested before it computes anything: both operand panels, the first W1' panels
// and (into registers) its 64 x 256 shortcut tile -- ~90 KB in flight per workgroup, which is what an HBM-bound kernel wants.
#include "common.h"
#include "../../include/rmem.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(rsrc_t r, lptr_t dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}
#else
struct rsrc_t {};
__device__ inline rsrc_t make_rsrc(const void*, long) { return {}; }
__device__ inline void buf_load_lds16(rsrc_t, lptr_t, int, int) {}
#endif

constexpr int BM = 64, N1 = 256, BK = 64;
constexpr int OOB = (int)0x80000000;
constexpr int R1_BYTES = N1 * BK * 2, CP = 132, R2_BYTES = 16 * CP * 4, R3_BYTES = 32768;
static_assert(R2_BYTES >= BM * BK * 2, "the A panel shares the staging region");
constexpr int LDS_BYTES = R1_BYTES + R2_BYTES + R3_BYTES;

struct BnParams {
  const e16* b; const e16* x2; const e16* w3; const float* b3; const e16* res; e16* y;
  const e16* w1; const float* b1; e16* a2;
  int M, K1, KT, N2;
  int Ho, Wo, HoWo, H2, W2, Cin2, stride2;
  long b_bytes, x2_bytes;
};

// element index of (row, 16-byte chunk) in a [rows][64] e16 panel: the chunk is XOR-swizzled with (row >> 1) & 7 (gemm_conv.hip)
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

template <int N2, bool DUAL>
__global__ __launch_bounds__(256) void k_bneck_chain(BnParams p) {
  constexpr int NB2 = N2 / 32;                 // DMA pieces (8 rows x 128 B) per wave and W1' k-step panel
  constexpr int NST = N2 == 64 ? 4 : 2;        // W1' panels resident at once
  constexpr int TN2 = N2 / 32;                 // 16-column tiles per wave in the second GEMM
  constexpr int NRES = DUAL ? 0 : 8;           // shortcut vectors (8 channels) per thread
  static_assert(NST * N2 * BK * 2 <= R3_BYTES, "W1' ring");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  char* const R1 = smem;
  char* const R2 = smem + R1_BYTES;
  char* const R3 = smem + R1_BYTES + R2_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fc = lane >> 4;
  const int m0 = blockIdx.x * BM;

  // ---- per-lane source offsets of the DMA pieces (piece = 8 rows x 128 B; lane = (row, physical chunk)) ----
  int a_off[2], a_off2[2], w3_off[8], w1_off[NB2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 16 * wave + 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int m = m0 + r;
    a_off[i] = m < p.M ? (m * p.K1 + c * 8) * 2 : OOB;
    a_off2[i] = OOB;
    if (DUAL && m < p.M) {
      const int img = m / p.HoWo, rem = m - img * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_off2[i] = (((img * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + c * 8) * 2;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 64 * wave + 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w3_off[i] = (r * p.KT + c * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < NB2; ++i) {
    const int r = (N2 / 4) * wave + 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w1_off[i] = (r * N1 + c * 8) * 2;
  }
  const rsrc_t rs_b = make_rsrc(p.b, p.b_bytes), rs_x2 = make_rsrc(DUAL ? p.x2 : p.b, DUAL ? p.x2_bytes : 0);
  const rsrc_t rs_w3 = make_rsrc(p.w3, (long)N1 * p.KT * 2), rs_w1 = make_rsrc(p.w1, (long)N2 * N1 * 2);
  const int nk1 = p.K1 / BK, nk = p.KT / BK;

  auto issue1 = [&](int kt) {                  // A and W3 panels of k-step kt
    if (DUAL && kt >= nk1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) buf_load_lds16(rs_x2, (lptr_t)(R2 + (16 * wave + 8 * i) * 128), a_off2[i], (kt - nk1) * (BK * 2));
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) buf_load_lds16(rs_b, (lptr_t)(R2 + (16 * wave + 8 * i) * 128), a_off[i], kt * (BK * 2));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) buf_load_lds16(rs_w3, (lptr_t)(R1 + (64 * wave + 8 * i) * 128), w3_off[i], kt * (BK * 2));
  };
  auto issue2 = [&](int kt, int stage) {       // W1' panel of k-step kt
#pragma unroll
    for (int i = 0; i < NB2; ++i)
      buf_load_lds16(rs_w1, (lptr_t)(R3 + stage * (N2 * BK * 2) + ((N2 / 4) * wave + 8 * i) * 128), w1_off[i], kt * (BK * 2));
  };

  issue1(0);
#pragma unroll
  for (int s = 0; s < NST; ++s) issue2(s, s);
  __builtin_amdgcn_sched_barrier(0);           // (the counted wait below relies on this issue order)
  // the shortcut tile, in the layout the first epilogue finishes rows in: pass q = (16-row group q >> 1, column half q & 1), this
  // thread's row = tid >> 4 of the group, channels (q & 1) * 128 + (tid & 15) * 8 .. + 7.  Rows past M read row M - 1 (never stored).
  e16x8 resv[NRES > 0 ? NRES : 1];
  const int erow = tid >> 4, ecv = tid & 15;
  if constexpr (!DUAL) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int m = min(m0 + (q >> 1) * 16 + erow, p.M - 1);
      resv[q] = *reinterpret_cast<const e16x8*>(p.res + (long)m * N1 + (q & 1) * 128 + ecv * 8);
    }
  }
  // the bias vectors this thread adds in the two epilogues, requested now as well: a load inside an epilogue pass would have to
  // be waited for with a vmcnt that also counts the previous pass's store of y
  f32x4 bias3[2][2], bias1[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    bias3[h][0] = *reinterpret_cast<const f32x4*>(p.b3 + h * 128 + ecv * 8);
    bias3[h][1] = *reinterpret_cast<const f32x4*>(p.b3 + h * 128 + ecv * 8 + 4);
  }
  constexpr int VPR2 = N2 / 8;                 // second epilogue: this thread finishes channels (tid % VPR2) * 8 .. + 7 of row tid / VPR2
  bias1[0] = *reinterpret_cast<const f32x4*>(p.b1 + (tid % VPR2) * 8);
  bias1[1] = *reinterpret_cast<const f32x4*>(p.b1 + (tid % VPR2) * 8 + 4);
  __builtin_amdgcn_sched_barrier(0);

  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kt = 0; kt < nk; ++kt) {
    if (kt == 0) {
      // the 10 pieces of k-step 0 have landed once only the younger requests (W1' panels, shortcut and bias vectors) are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST * NB2 + NRES + 6) : "memory");
    } else {
      __builtin_amdgcn_s_barrier();            // everyone finished reading the panels of k-step kt - 1
      issue1(kt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const e16* As = reinterpret_cast<const e16*>(R2);
    const e16* Bs = reinterpret_cast<const e16*>(R1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      e16x8 af[2], bfr[8];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const e16x8*>(&As[swz(wm * 32 + i * 16 + fr, 4 * ks + fc)]);
#pragma unroll
      for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const e16x8*>(&Bs[swz(wn * 128 + j * 16 + fr, 4 * ks + fc)]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = RMEM_MFMA_16x16x32(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();                             // both panels consumed: R1 becomes the y tile, R2 the staging buffer

  // ---- epilogue 1: y = relu(acc + bias3 (+ res)), stored and kept (rounded as stored) as the A operand of the second GEMM ----
  float* Cs = reinterpret_cast<float*>(R2);
  e16* Yt = reinterpret_cast<e16*>(R1);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int g = q >> 1, h = q & 1;           // 16-row group, column half
    if (wm == (g >> 1) && wn == h) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(fc * 4 + r) * CP + j * 16 + fr] = acc[g & 1][j][r];
    }
    __syncthreads();
    {
      const int rt = g * 16 + erow, m = m0 + rt, n = h * 128 + ecv * 8;
      const float* c = Cs + erow * CP + ecv * 8;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
      const f32x4 b0 = bias3[h][0], b1 = bias3[h][1];
      float v[8] = {c0[0] + b0[0], c0[1] + b0[1], c0[2] + b0[2], c0[3] + b0[3], c1[0] + b1[0], c1[1] + b1[1], c1[2] + b1[2], c1[3] + b1[3]};
      e16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (!DUAL) v[j] += (float)resv[q][j];
        o[j] = (e16)fmaxf(v[j], 0.f);
      }
      if (m < p.M) *reinterpret_cast<e16x8*>(p.y + (long)m * N1 + n) = o;
      *reinterpret_cast<e16x8*>(&Yt[(n >> 6) * (BM * BK) + swz(rt, (n >> 3) & 7)]) = o;
    }
    __syncthreads();
  }

  // ---- second GEMM: a2 tile [64][N2] = y tile [64][256] . W1'^T, four k-steps ----
  f32x4 acc2[2][TN2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (the first NST panels were requested before the shortcut vectors, which the epilogue above has consumed: they have landed, and
  // the barriers of the epilogue made every wave's pieces visible)
  if constexpr (DUAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (DUAL) __syncthreads();
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int stage = kt % NST;
    if (kt >= NST) {
      // panel kt was requested after the stores of y: all but the pieces of a later panel must be complete
      if (kt + 1 < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB2) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const e16* As = Yt + kt * (BM * BK);
    const e16* Bs = reinterpret_cast<const e16*>(R3 + stage * (N2 * BK * 2));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      e16x8 af[2], bfr[TN2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const e16x8*>(&As[swz(wm * 32 + i * 16 + fr, 4 * ks + fc)]);
#pragma unroll
      for (int j = 0; j < TN2; ++j) bfr[j] = *reinterpret_cast<const e16x8*>(&Bs[swz(wn * (N2 / 2) + j * 16 + fr, 4 * ks + fc)]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN2; ++j) acc2[i][j] = RMEM_MFMA_16x16x32(af[i], bfr[j], acc2[i][j], 0, 0, 0);
    }
    if (kt + NST < 4) {
      __syncthreads();                         // everyone finished reading this stage (fragments are in registers)
      issue2(kt + NST, stage);
    }
  }
  __syncthreads();

  // ---- epilogue 2: a2 = relu(acc2 + bias1'), 16 rows a pass ----
  constexpr int CP2 = N2 + 4;
  static_assert(16 * CP2 * 4 <= R2_BYTES, "second staging buffer");
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (wm == (g >> 1)) {
#pragma unroll
      for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(fc * 4 + r) * CP2 + wn * (N2 / 2) + j * 16 + fr] = acc2[g & 1][j][r];
    }
    __syncthreads();
    if (tid < 16 * VPR2) {
      const int row = tid / VPR2, cv = tid % VPR2;
      const int m = m0 + g * 16 + row, n = cv * 8;
      const float* c = Cs + row * CP2 + n;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
      const f32x4 b0 = bias1[0], b1 = bias1[1];
      const float v[8] = {c0[0] + b0[0], c0[1] + b0[1], c0[2] + b0[2], c0[3] + b0[3], c1[0] + b1[0], c1[1] + b1[1], c1[2] + b1[2], c1[3] + b1[3]};
      e16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (e16)fmaxf(v[j], 0.f);
      if (m < p.M) *reinterpret_cast<e16x8*>(p.a2 + (long)m * N2 + n) = o;
    }
    __syncthreads();
  }
}

bool al16(const void* q) { return ((uintptr_t)q % 16) == 0; }

}  // namespace

extern "C" int RMEM_API(rmem_bneck_chain)(const rmem_bneck_chain_desc* d, const void* b, const void* x2, const void* w3, const float* bias3,
                                          const void* res, void* y, const void* w1, const float* bias1, void* a2, void* stream) {
  RMEM_REQUIRE(d && b && w3 && bias3 && y && w1 && bias1 && a2, "rmem_bneck_chain: null argument");
  RMEM_REQUIRE(d->batch >= 1 && d->Ho > 0 && d->Wo > 0 && d->K1 > 0 && d->K1 % 64 == 0 && d->Cout == 256 && (d->N2 == 64 || d->N2 == 128),
               "rmem_bneck_chain: K1 must be a multiple of 64, Cout 256, N2 64 or 128");
  RMEM_REQUIRE((x2 != nullptr) != (res != nullptr), "rmem_bneck_chain: exactly one of res (identity shortcut) and x2 (1x1 shortcut) must be given");
  RMEM_REQUIRE(!x2 || (d->Cin2 > 0 && d->Cin2 % 64 == 0 && d->stride2 >= 1 && d->H2 >= (d->Ho - 1) * d->stride2 + 1 && d->W2 >= (d->Wo - 1) * d->stride2 + 1),
               "rmem_bneck_chain: the 1x1 shortcut needs Cin2 % 64 == 0 and an H2 x W2 map that covers the strided samples");
  RMEM_REQUIRE(al16(b) && al16(x2) && al16(w3) && al16(bias3) && al16(res) && al16(y) && al16(w1) && al16(bias1) && al16(a2),
               "rmem_bneck_chain: operands must be 16-byte aligned");
  BnParams p;
  p.b = (const e16*)b; p.x2 = (const e16*)x2; p.w3 = (const e16*)w3; p.b3 = bias3; p.res = (const e16*)res; p.y = (e16*)y;
  p.w1 = (const e16*)w1; p.b1 = bias1; p.a2 = (e16*)a2;
  p.Ho = d->Ho; p.Wo = d->Wo; p.HoWo = d->Ho * d->Wo;
  const long M = (long)d->batch * p.HoWo;
  p.K1 = d->K1; p.KT = d->K1 + (x2 ? d->Cin2 : 0); p.N2 = d->N2;
  p.H2 = d->H2; p.W2 = d->W2; p.Cin2 = d->Cin2; p.stride2 = d->stride2;
  p.b_bytes = M * d->K1 * 2;
  p.x2_bytes = x2 ? (long)d->batch * d->H2 * d->W2 * d->Cin2 * 2 : 0;
  const long lim = (1L << 31) - (1L << 22);
  RMEM_REQUIRE(M < (1L << 30) && p.b_bytes < lim && p.x2_bytes < lim, "rmem_bneck_chain: an operand exceeds the 2 GB a buffer descriptor addresses");
  p.M = (int)M;
  const dim3 grid((unsigned)((M + BM - 1) / BM));
  hipStream_t s = (hipStream_t)stream;
  if (x2) {
    if (d->N2 == 64) hipLaunchKernelGGL((k_bneck_chain<64, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_bneck_chain<128, true>), grid, dim3(256), 0, s, p);
  } else {
    if (d->N2 == 64) hipLaunchKernelGGL((k_bneck_chain<64, false>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_bneck_chain<128, false>), grid, dim3(256), 0, s, p);
  }
  return rmem_check_launch("rmem_bneck_chain");
}
