// Identity-bank embedding of a label map without the one-hot tensor (gfx950, wave64).
//
//   id_emb[pos, :] = bias + sum over the K x K window of pos:  W[:, ky, kx, label(y, x)]        (networks/models/aot.py:139-147,
//   patch_wise_id_bank = Conv2d(max_obj + 1, 256, 17, stride 16, padding 8) applied to one_hot(mask); engines/aot_engine.py:208-232)
//
// As a GEMM this is [positions] x [K*K*16] . [K*K*16] x [256] with an A operand that is 1/16 dense.  The implicit-GEMM form wrote
// the one-hot map first (104 MB per 8 clips at 481 x 849: rmem_label_to_onehot16) and copied it through LDS-DMA once per filter tap
// beside the weights (31 + 81..108 us per group step).  Here the one-hot operand never exists in memory: the
// weights are the MFMA A operand (rows = output channels), streamed through an LDS-DMA ring in k-steps of 64 = 4 taps x 16
// channels, and the B fragment of a (position, tap pair) is built in REGISTERS from one label byte per fragment: lane (position
// l & 15, chunk l >> 4) covers tap 2 ks + (chunk >> 1), channels 8 (chunk & 1) .. + 7, i.e. the constant 1.0 shifted to element
// label & 7 if label >> 3 equals its half, else zeros.  Label bytes are requested two k-steps ahead, like the weight panels.
// K runs over the real taps only (17 x 17 x 16 = 4624 = 72.25 k-steps instead of 85 row-padded ones).
// A first tiny kernel resizes the delivered label map (output size) to the network size, nearest, exactly as the one-hot kernel did.
// 256 threads = 2 (channel halves) x 2 (position halves of 32) waves; tile = 64 positions x 128 channels, a 3-deep weight ring
// (measured: 64 channels x 8 stages 113 us against 72 us -- every workgroup builds the one-hot fragments of its positions again).
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

#ifndef RMEM_IDB_TCH
#define RMEM_IDB_TCH 128
#endif
#ifndef RMEM_IDB_ST
#define RMEM_IDB_ST 3
#endif
constexpr int TPOS = 64, TCH = RMEM_IDB_TCH, BK = 64, ST = RMEM_IDB_ST, AHEAD = ST - 1;   // weights and labels are requested AHEAD k-steps ahead
constexpr int PANEL = TCH * BK * 2;             // one weight k-step panel: [TCH channels][64 k] e16, 128-byte rows, XOR-swizzled chunks
constexpr int NP = TCH / 32;                    // LDS-DMA pieces (8 rows x 128 B) per wave and panel
constexpr int CT = TCH / 32;                    // 16-channel tiles per wave
constexpr int SROW = TCH * 2 + 16;              // staging row of one position

struct IdParams {
  const uint8_t* lab;                           // [images][Hpd][Wpd]: labels at the network size inside a border of 255s (pad all round, one more row below)
  const e16* w; const float* bias; e16* out;
  int images, Hpd, Wpd, Ho, Wo, KH, KW, stride, ncls, M, K, nk;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// the 8-element one-hot fragment of label L for channel half `half` (channels 8 half .. + 7): 1.0 at element L & 7, or zeros
__device__ __forceinline__ e16x8 onehot_frag(int L, int half, int ncls) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const unsigned short one16 = __builtin_bit_cast(unsigned short, (e16)1.0f);
  const bool hit = L < ncls && (L >> 3) == half;
  const unsigned word = hit ? ((unsigned)one16 << ((L & 1) * 16)) : 0u;
  const int d = (L & 7) >> 1;
  const u32x4 v = {d == 0 ? word : 0u, d == 1 ? word : 0u, d == 2 ? word : 0u, d == 3 ? word : 0u};
  return __builtin_bit_cast(e16x8, v);
}

__global__ __launch_bounds__(256) void k_idbank_labels(IdParams p) {
  __shared__ __attribute__((aligned(16))) char smem[ST * PANEL + TPOS * SROW + 34 * 16];      // ONE shared object (see stem.hip)
  char* const stage = smem + ST * PANEL;
  char* const lut = stage + TPOS * SROW;        // one-hot fragments by (label 0 .. 15 | none, channel half): building one costs ~12 VALU, reading one 1 ds_read
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;      // channel half (32 of the tile's 64), position half (32 of 64)
  const int fr = lane & 15, fc = lane >> 4;
  const int m0 = blockIdx.x * TPOS, n0 = blockIdx.y * TCH;
  const rsrc_t rs_w = make_rsrc(p.w, (long)256 * p.K * 2);

  // weight panel DMA: 64 rows x 128 B per k-step = NP pieces (8 rows x 128 B) per wave; source-side swizzle as gemm_conv.hip
  int w_off[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = (TCH / 4) * wave + 8 * i + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    w_off[i] = ((n0 + r) * p.K + c * 8) * 2;
  }
  auto issue = [&](int kt, int stg) {
#pragma unroll
    for (int i = 0; i < NP; ++i) buf_load_lds16(rs_w, (lptr_t)(smem + stg * PANEL + ((TCH / 4) * wave + 8 * i) * 128), w_off[i], kt * (BK * 2));
  };

  // this lane's two positions (pt = 0, 1): offset of their window's top-left corner in the bordered label map
  int poff[2];
  [[maybe_unused]] const uint8_t* lab = p.lab;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int m = min(m0 + wp * 32 + pt * 16 + fr, p.M - 1);
    const int img = m / (p.Ho * p.Wo), rem = m - img * (p.Ho * p.Wo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    poff[pt] = (img * p.Hpd + oy * p.stride) * p.Wpd + ox * p.stride;
  }
  const int half = fc & 1, tsel = fc >> 1;      // this lane's channel half and which tap of a slice's pair it covers
  const int ntaps = p.KH * p.KW;
  // this lane's taps of the NEXT fetch: (ky, kx) of tap 4 kt + 2 ks + tsel, advanced by 4 taps per k-step (no division in the loop)
  int tky[2], tkx[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) { tky[ks] = (2 * ks + tsel) / p.KW; tkx[ks] = (2 * ks + tsel) - tky[ks] * p.KW; }
  // label bytes of the next k-step: positions pt = 0, 1 under taps ks = 0, 1.  Every tap of every window lies inside the bordered map
  // (255 = no class outside the image; the padding taps of the last k-step read the row below the window and are discarded), so
  // the loads need no predicate -- a predicated load is put behind a branch, and the compiler drains vmcnt at its join.
  auto fetch = [&](int (&L)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      [[maybe_unused]] const int toff = min(tky[ks], p.KH) * p.Wpd + tkx[ks];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        // (inline asm: a compiler-visible load inside the ring loop makes the compiler drain vmcnt -- the next weight panel with it -- at
        // the loop header before the label registers are read.  The loop's own counted wait covers these loads: they are requested
        // before the panel of the same iteration and read only after the next iteration's wait.)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("global_load_ubyte %0, %1, %2" : "=v"(L[ks][pt]) : "v"(poff[pt] + toff), "s"(lab) : "memory");
#else
        L[ks][pt] = 0;
#endif
      }
      tkx[ks] += 4;                             // (KW >= 4: at most one wrap)
      if (tkx[ks] >= p.KW) { tkx[ks] -= p.KW; ++tky[ks]; }
    }
  };

  if (tid < 34) *reinterpret_cast<e16x8*>(lut + tid * 16) = onehot_frag((tid >> 1) < 16 ? (tid >> 1) : 255, tid & 1, p.ncls);
  __syncthreads();
  // labels and weight panels are both requested AHEAD k-steps ahead; labels live in a register ring of ST sets (set = stage = kt % ST).
  int Lr[ST][2][2];
  // (request order of the prologue = the order the loop keeps: labels kt + AHEAD | panel kt + AHEAD per step)
#pragma unroll
  for (int j = 0; j < AHEAD; ++j) {
    fetch(Lr[j]);
    __builtin_amdgcn_sched_barrier(0);
    issue(min(j, p.nk - 1), j);
    __builtin_amdgcn_sched_barrier(0);
  }
  f32x4 acc[CT][2];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // k-steps in groups of ST (label set and ring stage = kt % ST are compile-time: the inner loop is unrolled); the surplus steps of the
  // last group multiply zero one-hot fragments
  for (int kt0 = 0; kt0 < p.nk; kt0 += ST) {
#pragma unroll
    for (int set = 0; set < ST; ++set) {
      const int kt = kt0 + set;
      // panel kt and the labels of k-step kt were requested AHEAD steps ago; each step since has requested 4 labels + NP pieces: once at
      // most (AHEAD - 1) (4 + NP) requests are outstanding, both have landed.  The label registers are written by inline asm and read
      // only here, behind this wait.
      // (the label registers are operands of the wait: nothing that reads them may be scheduled above it)
      asm volatile("s_waitcnt vmcnt(%4)"
                   : "+v"(Lr[set][0][0]), "+v"(Lr[set][0][1]), "+v"(Lr[set][1][0]), "+v"(Lr[set][1][1])
                   : "n"((AHEAD - 1) * (4 + NP))
                   : "memory");
      __builtin_amdgcn_s_barrier();             // everyone is done with stage (set - 1) % ST as well: it takes panel kt + AHEAD below
      e16x8 bf[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)          // (taps past the window -- the tail of the last k-step and the surplus k-steps -- read some label: no class)
          bf[ks][pt] = *reinterpret_cast<const e16x8*>(lut + ((4 * kt + 2 * ks + tsel < ntaps ? min(Lr[set][ks][pt], 16) : 16) * 2 + half) * 16);
      __builtin_amdgcn_sched_barrier(0);
      fetch(Lr[(set + AHEAD) % ST]);            // labels of k-step kt + AHEAD (the set this lane read one step ago)
      __builtin_amdgcn_sched_barrier(0);
      issue(min(kt + AHEAD, p.nk - 1), (set + AHEAD) % ST);     // unconditionally (past the end: the last panel again, into a stage nobody reads)
      __builtin_amdgcn_sched_barrier(0);
      const e16* Ws = reinterpret_cast<const e16*>(smem + set * PANEL);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        e16x8 af[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) af[ct] = *reinterpret_cast<const e16x8*>(&Ws[swz(wc * (TCH / 2) + ct * 16 + fr, 4 * ks + fc)]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = RMEM_MFMA_16x16x32(af[ct], bf[ks][pt], acc[ct][pt], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the tail's surplus panel requests
  __syncthreads();

  // epilogue: + bias, round; lane holds channels 4 fc .. + 3 of position fr of each (ct, pt) tile -> staging rows of whole positions
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + wc * (TCH / 2) + ct * 16 + fc * 4);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      e16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (e16)(acc[ct][pt][r] + bv[r]);
      *reinterpret_cast<e16x4*>(stage + (wp * 32 + pt * 16 + fr) * SROW + (wc * (TCH / 2) + ct * 16 + fc * 4) * 2) = o;
    }
  }
  __syncthreads();
  constexpr int VPR = TCH / 8;                  // 16-byte vectors per staged position
#pragma unroll
  for (int j = 0; j < TPOS * VPR / 256; ++j) {
    const int v = tid + 256 * j, pos = v / VPR, c16 = v % VPR;
    if (m0 + pos < p.M)
      *reinterpret_cast<e16x8*>(p.out + (long)(m0 + pos) * 256 + n0 + c16 * 8) = *reinterpret_cast<const e16x8*>(stage + pos * SROW + c16 * 16);
  }
}

// delivered labels (uint8 or fp32, [images][Hs][Ws]) -> uint8 at the network size, nearest (F.interpolate(mode='nearest'), the mapping
// of k_label_onehot in resample.hip)
__global__ __launch_bounds__(256) void k_label_resize_u8(const void* lab, int lab_f32, int Hs, int Ws, int Hd, int Wd, int pad, int Hpd, int Wpd,
                                                         uint8_t* out) {
  const long total = (long)Hd * Wd;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  lab = reinterpret_cast<const char*>(lab) + (long)blockIdx.y * Hs * Ws * (lab_f32 ? 4 : 1);
  const int y = (int)(i / Wd), x = (int)(i - (long)y * Wd);
  const int sy = min((int)floorf((float)y * ((float)Hs / (float)Hd)), Hs - 1);
  const int sx = min((int)floorf((float)x * ((float)Ws / (float)Wd)), Ws - 1);
  const long si = (long)sy * Ws + sx;
  const int v = lab_f32 ? (int)reinterpret_cast<const float*>(lab)[si] : (int)reinterpret_cast<const uint8_t*>(lab)[si];
  out[((long)blockIdx.y * Hpd + y + pad) * Wpd + x + pad] = (uint8_t)v;
}

}  // namespace

#ifndef RMEM_F16
extern "C" int rmem_label_id_embed_scratch_size(int H, int W, int pad, int* Hpd, int* Wpd) {
  if (H <= 0 || W <= 0 || pad < 0 || !Hpd || !Wpd) return -1;
  *Hpd = H + 2 * pad + 1;
  *Wpd = W + 2 * pad;
  return 0;
}
#endif

extern "C" int RMEM_API(rmem_label_id_embed)(const void* label, int label_is_f32, int images, int Hs, int Ws, int H, int W, int KH, int KW,
                                             int stride, int pad, int num_classes, const void* w, const float* bias, void* label_scratch_u8,
                                             void* out, void* stream) {
  RMEM_REQUIRE(label && w && bias && label_scratch_u8 && out && images >= 1 && Hs > 0 && Ws > 0 && H > 0 && W > 0,
               "rmem_label_id_embed: bad argument");
  RMEM_REQUIRE(KH >= 1 && KW >= 4 && KH * KW <= 1024 && stride >= 1 && pad >= 0 && num_classes >= 1 && num_classes <= 16,
               "rmem_label_id_embed: bad kernel geometry / class count (KW >= 4)");
  RMEM_REQUIRE(((uintptr_t)w % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)out % 16) == 0, "rmem_label_id_embed: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  IdParams p;
  rmem_label_id_embed_scratch_size(H, W, pad, &p.Hpd, &p.Wpd);
  hipLaunchKernelGGL(k_label_resize_u8, dim3((unsigned)(((long)H * W + 255) / 256), images), dim3(256), 0, s, label, label_is_f32, Hs, Ws, H, W,
                     pad, p.Hpd, p.Wpd, (uint8_t*)label_scratch_u8);
  p.lab = (const uint8_t*)label_scratch_u8; p.w = (const e16*)w; p.bias = bias; p.out = (e16*)out;
  p.images = images; p.KH = KH; p.KW = KW; p.stride = stride; p.ncls = num_classes;
  p.Ho = (H + 2 * pad - KH) / stride + 1; p.Wo = (W + 2 * pad - KW) / stride + 1;
  RMEM_REQUIRE(p.Ho > 0 && p.Wo > 0, "rmem_label_id_embed: empty output");
  const long M = (long)images * p.Ho * p.Wo;
  p.K = KH * KW * 16;
  p.nk = (p.K + BK - 1) / BK;
  RMEM_REQUIRE(M < (1L << 30) && (long)images * p.Hpd * p.Wpd < (1L << 31) && (long)256 * p.K * 2 < (1L << 31) - (1L << 22), "rmem_label_id_embed: problem too large");
  p.M = (int)M;
  hipLaunchKernelGGL(k_idbank_labels, dim3((unsigned)((M + TPOS - 1) / TPOS), 256 / TCH), dim3(256), 0, s, p);
  return rmem_check_launch("rmem_label_id_embed");
}
