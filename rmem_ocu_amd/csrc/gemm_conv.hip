// Implicit-GEMM convolution / linear layer for gfx950 (CDNA4).
//
//   Y[m, n] = act( sum_k A[m, k] * Wt[n, k] + bias[n] (+ R[m, n]) )
//
// A is never materialised (im2col-free): m = (ho, wo) of an NHWC e16 feature map
// and k = (kh, kw, ci) with ci fastest, gathered straight from HBM in 16-byte
// (8 x e16) pieces with zero fill outside the image.  A linear layer is the
// 1x1 case (H = rows, W = 1).  Replaces the cuDNN/MIOpen conv + addmm call sites
// of the reference (encoders/resnet.py:48-68,
// decoders/fpn.py:36-68, layers/transformer.py:576, 675, 685, models/aot.py:112, 133).
//
// Tiling: 256 threads = 4 waves in a 2x2 grid over a BM x BN block tile, BK = 64,
// v_mfma_f32_16x16x32_bf16 (two MFMAs per k-step and 16x16 sub-tile), fp32 accumulation.
// Global -> register -> LDS staging: a ring of PF register stages keeps PF k-tiles in flight
// from HBM/L2 (these problems run at ~1 workgroup per CU, so latency is hidden inside the
// workgroup, not by occupancy); LDS is double-buffered with one barrier per k-step.  LDS rows
// are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 so the four 16-lane
// groups of a ds_read_b128 fragment read hit 16 distinct 16-byte slots of the 256-byte bank row.
//
// Epilogue: the accumulators go through LDS (fp32, half a tile at a time) so that every
// thread finishes 8 consecutive channels of one output row: bias / residual / second
// output / store are 16- or 32-byte accesses instead of 2-byte scatters.
//
// Split-K: problems with few output tiles (M = 1674 token GEMMs with K up to 4624) are
// cut along K over gridDim.z; each slice stores an fp32 slab with plain 16-byte stores
// and k_splitk_epilogue sums the slabs in slice order (bitwise reproducible, no atomics)
// while applying the same fused epilogue.
#include "common.h"
#include "../../include/rmem.h"
#include <stdlib.h>

namespace {

struct ConvParams {
  const e16* x;
  const e16* w;
  const float* bias;
  const void* res;
  void* y;
  e16* y2;
  float* slabs;          // split-K partials [splits][M][Cout] (fp32) or null
  int H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int M, K, HoWo;
  int ldo, ldr, ld2;
  int relu, out_f32, res_f32;
  int ldx;               // input row stride of the 1x1 (GEMM) case
  int act_begin;         // first output channel the activation applies to (multiple of 8)
  int up_h, up_w, up_align;   // > 0: the (e16) residual is a [batch][up_h][up_w] map, bilinearly resized to (Ho, Wo) on the fly
  int steps_per_split;   // k-steps (of 32) per gridDim.z slice
  int vec_ok;            // all leading dimensions / pointers allow 8-wide vector access
  long x_elems;          // addressable span of x in elements (fast path: buffer descriptor range)
  const e16* x2;        // dual form: second A source, NHWC [batch][H2][W2][Cin2] sampled at stride2; k >= Cin comes from it
  int H2, W2, Cin2, stride2;
  long x2_elems;
  int fast_ok;           // 1: Cin % 64 == 0, <= 32 taps, x and w below 2 GB: scalar k-walk + hardware zero fill; 2: row-run form
  int xcd_ny;            // > 0: 1-D grid in XCD-aware order, xcd_ny = column tiles per row tile (see tile_of_block)
  int debug;             // timing experiments only (RMEM_GEMM_DEBUG; results are then wrong by construction): 1 = no MFMA, 2 = no DMA
                         // after the first k-step, 4 = no epilogue (nothing is stored), 8 = no global stores / residual reads in it
};

// Which output tile a workgroup owns.  Plain 2-D grid (xcd_ny == 0): (blockIdx.x, blockIdx.y) -- all row tiles of column tile 0 are
// dispatched before any of column tile 1, so by the time a row tile's A operand is wanted again it has left every L2 and is
// fetched from the Infinity Cache / HBM once per COLUMN tile.  XCD-aware 1-D grid (xcd_ny = Ny > 0): hardware deals consecutive
// block ids round-robin over the 8 XCDs (a private 4 MiB L2 each), so ids that are congruent mod 8 walk the (row tile, column
// tile) pairs with the column tile fastest: the Ny workgroups that read the same A rows run back to back on ONE XCD and all
// but the first find them in its L2.  Row tile = (j / Ny) * 8 + xcd for the j-th block of an XCD; the grid is rounded up to a
// whole number of row tiles per XCD, the surplus workgroups exit at once.  Placement is a speed matter only.
__device__ __forceinline__ bool tile_of_block(const ConvParams& p, int BM, int& bx, int& by) {
  if (p.xcd_ny <= 0) { bx = blockIdx.x; by = blockIdx.y; return true; }
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int xl = j / p.xcd_ny;
  bx = xl * 8 + xcd;
  by = j - xl * p.xcd_ny;
  return bx * BM < p.M;
}

// LDS tile rows are 128 B (BK = 64 e16 = 8 chunks of 16 B).  Physical chunk = chunk ^ ((row >> 1) & 7): the 16 lanes of
// every ds_read_b128 lane group (rows r..r+3, r+12..r+15 at chunk c and rows r+4..r+11 at chunk c+1) then land on 16
// distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// finish 8 consecutive channels n..n+7 of output row m (v = accumulator + nothing yet)
__device__ __forceinline__ void finish8(const ConvParams& p, int m, int n, float (&v)[8]) {
  if (p.bias) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
  }
  if (p.y2) {
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (e16)v[j];
    *reinterpret_cast<e16x8*>(p.y2 + (long)m * p.ld2 + n) = o;
  }
  if (p.res && p.up_h > 0) {
    // same arithmetic and the same e16 rounding as k_bilinear_nhwc followed by a plain residual add
    const int img = m / p.HoWo, rem = m - img * p.HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    int y0, y1, x0, x1; float wy, wx;
    rmem_src_coord(oy, p.up_h, p.Ho, p.up_align, y0, y1, wy);
    rmem_src_coord(ox, p.up_w, p.Wo, p.up_align, x0, x1, wx);
    const e16* rb = reinterpret_cast<const e16*>(p.res) + (long)img * p.up_h * p.up_w * p.ldr + n;
    const e16x8 a = *reinterpret_cast<const e16x8*>(rb + ((long)y0 * p.up_w + x0) * p.ldr);
    const e16x8 b = *reinterpret_cast<const e16x8*>(rb + ((long)y0 * p.up_w + x1) * p.ldr);
    const e16x8 c = *reinterpret_cast<const e16x8*>(rb + ((long)y1 * p.up_w + x0) * p.ldr);
    const e16x8 d = *reinterpret_cast<const e16x8*>(rb + ((long)y1 * p.up_w + x1) * p.ldr);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += (float)(e16)rmem_bilerp((float)a[j], (float)b[j], (float)c[j], (float)d[j], wx, wy);
  } else if (p.res) {
    if (p.res_f32) {
      const float* r = reinterpret_cast<const float*>(p.res) + (long)m * p.ldr + n;
      const f32x4 r0 = *reinterpret_cast<const f32x4*>(r), r1 = *reinterpret_cast<const f32x4*>(r + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
    } else {
      const e16x8 r = *reinterpret_cast<const e16x8*>(reinterpret_cast<const e16*>(p.res) + (long)m * p.ldr + n);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += (float)r[j];
    }
  }
  if (n >= p.act_begin) {
    if (p.relu == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    } else if (p.relu == 2) {               // exact (erf) GELU: Swin MLP, encoders/swin/swin_transformer.py:55-57
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
    } else if (p.relu == 3) {               // SiLU: layers/attention.py:89-90
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.f + expf(-v[j]));
    }
  }
  if (p.out_f32) {
    float* y = reinterpret_cast<float*>(p.y) + (long)m * p.ldo + n;
    *reinterpret_cast<f32x4*>(y) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(y + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    e16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (e16)v[j];
    *reinterpret_cast<e16x8*>(reinterpret_cast<e16*>(p.y) + (long)m * p.ldo + n) = o;
  }
}

// scalar tail (Cout not a multiple of 8, or unaligned leading dimensions)
__device__ __forceinline__ void finish1(const ConvParams& p, int m, int n, float v) {
  if (p.bias) v += p.bias[n];
  if (p.y2) p.y2[(long)m * p.ld2 + n] = (e16)v;
  if (p.res)
    v += p.res_f32 ? reinterpret_cast<const float*>(p.res)[(long)m * p.ldr + n]
                   : (float)reinterpret_cast<const e16*>(p.res)[(long)m * p.ldr + n];
  if (n >= p.act_begin) {
    if (p.relu == 1) v = fmaxf(v, 0.f);
    else if (p.relu == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    else if (p.relu == 3) v = v / (1.f + expf(-v));
  }
  if (p.out_f32) reinterpret_cast<float*>(p.y)[(long)m * p.ldo + n] = v;
  else reinterpret_cast<e16*>(p.y)[(long)m * p.ldo + n] = (e16)v;
}

template <int BM, int BN, int PF, bool IS1X1, bool SPLITK>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvParams p) {
  constexpr int BK = 64;
  constexpr int NA = BM / 32;  // 16-byte A chunks per thread per k-step (BM rows x 8 chunks / 256 threads)
  constexpr int NB = BN / 32;
  constexpr int TM = BM / 32;  // 16x16 tiles per wave along M
  constexpr int TN = BN / 32;
  constexpr int CP = BN + 4;   // padded fp32 row of the epilogue staging tile
  constexpr int AB_BYTES = 2 * (BM + BN) * BK * 2;
  constexpr int C_BYTES = (BM / 2) * CP * 4;
  constexpr int SMEM = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  e16* As = reinterpret_cast<e16*>(smem);                 // [2][BM*64]
  e16* Bs = As + 2 * BM * BK;                              // [2][BN*64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nk_total = (p.K + BK - 1) / BK;
  const int kt0 = SPLITK ? blockIdx.z * p.steps_per_split : 0;
  const int kt1 = SPLITK ? min(nk_total, kt0 + p.steps_per_split) : nk_total;

  // ---- per-thread gather state for the A operand: chunk id = tid + i*256 -> (row = id >> 3, chunk = id & 7) ----
  const int a_chunk = tid & 7;             // same for every i (256 % 8 == 0)
  long a_base[NA];
  int a_hi0[NA], a_wi0[NA];
  bool a_ok[NA];
  int a_ci, a_kw, a_kh;                    // k position of this thread's chunk column (shared by its rows)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + (tid >> 3) + i * 32;
    a_ok[i] = m < p.M;
    if (IS1X1) {
      a_base[i] = (long)m * p.ldx;
      a_hi0[i] = a_wi0[i] = 0;
    } else {
      const int img = m / p.HoWo, rem = m - img * p.HoWo;      // batch of images: rows are [image][ho][wo]
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride - p.pad;
      a_wi0[i] = wo * p.stride - p.pad;
      a_base[i] = (long)img * p.H * p.W * p.Cin;
    }
  }
  if (IS1X1) {
    a_ci = a_kw = a_kh = 0;
  } else {
    const int kidx = kt0 * BK + a_chunk * 8;
    const int kk = kidx / p.Cin;
    a_ci = kidx - kk * p.Cin;
    a_kh = kk / p.KW;
    a_kw = kk - a_kh * p.KW;
  }
  const int b_chunk = tid & 7;

  const e16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // register ring: PF tiles of (A, B) pieces in flight from HBM/L2 while earlier tiles are being multiplied
  e16x8 ra[PF][NA], rb[PF][NB];

  auto load_tile = [&](e16x8 (&xa)[NA], e16x8 (&xb)[NB], int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      e16x8 v = zero8;
      if (IS1X1) {
        const int kidx = k0 + a_chunk * 8;
        if (a_ok[i] && kidx < p.K) v = *reinterpret_cast<const e16x8*>(p.x + a_base[i] + kidx);
      } else {
        const int hi = a_hi0[i] + a_kh, wi = a_wi0[i] + a_kw;
        if (a_ok[i] && a_kh < p.KH && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
          v = *reinterpret_cast<const e16x8*>(p.x + a_base[i] + ((long)hi * p.W + wi) * p.Cin + a_ci);
      }
      xa[i] = v;
    }
    if (!IS1X1) {   // advance this thread's (ci, kw, kh) by BK for the next k-step
      a_ci += BK;
      while (a_ci >= p.Cin) {
        a_ci -= p.Cin;
        if (++a_kw == p.KW) { a_kw = 0; ++a_kh; }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      e16x8 v = zero8;
      const int n = n0 + (tid >> 3) + i * 32;
      const int kidx = k0 + b_chunk * 8;
      if (n < p.Cout && kidx < p.K) v = *reinterpret_cast<const e16x8*>(p.w + (long)n * p.K + kidx);
      xb[i] = v;
    }
  };
  auto store_tile = [&](const e16x8 (&xa)[NA], const e16x8 (&xb)[NB], int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<e16x8*>(&As[buf * BM * BK + swz((tid >> 3) + i * 32, a_chunk)]) = xa[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<e16x8*>(&Bs[buf * BN * BK + swz((tid >> 3) + i * 32, b_chunk)]) = xb[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fc = lane >> 4;
  // prologue: PF tiles in flight, tile kt0 staged
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (kt0 + u < kt1) load_tile(ra[u], rb[u], (kt0 + u) * BK);
  if (kt0 < kt1) store_tile(ra[0], rb[0], 0);
  __syncthreads();

  for (int ktb = kt0; ktb < kt1; ktb += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {        // static ring index (runtime-indexed register arrays would go to scratch)
      const int kt = ktb + u;
      if (kt < kt1) {
        const int cur = (kt - kt0) & 1;
        if (kt + PF < kt1) load_tile(ra[u], rb[u], (kt + PF) * BK);     // slot u was drained into LDS one step ago
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          e16x8 af[TM], bfr[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            af[i] = *reinterpret_cast<const e16x8*>(&As[cur * BM * BK + swz(wm * (BM / 2) + i * 16 + fr, 4 * ks + fc)]);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bfr[j] = *reinterpret_cast<const e16x8*>(&Bs[cur * BN * BK + swz(wn * (BN / 2) + j * 16 + fr, 4 * ks + fc)]);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = RMEM_MFMA_16x16x32(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tile(ra[(u + 1) % PF], rb[(u + 1) % PF], cur ^ 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue through LDS: two passes of BM/2 rows ----
  // C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int VPR = BN / 8;                       // 8-wide vectors per staged row
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cs[(i * 16 + fc * 4 + r) * CP + wn * (BN / 2) + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();
    for (int vi = tid; vi < (BM / 2) * VPR; vi += 256) {
      const int row = vi / VPR, cv = vi - row * VPR;
      const int m = m0 + pass * (BM / 2) + row;
      const int n = n0 + cv * 8;
      if (m >= p.M || n >= p.Cout) continue;
      const float* c = Cs + row * CP + cv * 8;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
      float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      if (SPLITK) {
        float* s = p.slabs + ((long)blockIdx.z * p.M + m) * p.Cout + n;    // Cout % 8 == 0 is required for split-K
        *reinterpret_cast<f32x4*>(s) = c0;
        *reinterpret_cast<f32x4*>(s + 4) = c1;
      } else if (p.vec_ok && n + 8 <= p.Cout) {
        finish8(p, m, n, v);
      } else {
        for (int j = 0; j < 8 && n + j < p.Cout; ++j) finish1(p, m, n + j, v[j]);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 64x64 tile (the one every shape of this path uses).  Global -> LDS goes through
// global_load_lds_dwordx4 (no VGPR staging, no ds_write: the ~79 B/clk ds_write path was the per-k-step bottleneck of the
// register-staged kernel).  Each wave-instruction fills 8 rows x 128 B of the tile linearly, so the XOR swizzle is
// applied to the per-lane SOURCE address; padded / out-of-range pieces read a 16-byte zero buffer.  ST-deep LDS ring with
// counted vmcnt + raw s_barrier; the default is ST = 1 (a single 16 KB buffer, 8 workgroups per CU: see launch()).
static __device__ uint4 g_zero16[1];

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// buffer-descriptor LDS-DMA (buffer_load_dwordx4 ... offen lds).  The descriptor type and these builtins exist only in the
// device pass of hipcc; the host pass just needs the names to parse.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load_lds16(rsrc_t r, lptr_t dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}
#else
struct rsrc_t {};
__device__ inline rsrc_t make_rsrc(const void*, int) { return {}; }
__device__ inline void buf_load_lds16(rsrc_t, lptr_t, int, int) {}
#endif

//
// FAST (Cin % 64 == 0, which every 3x3 / 1x1 layer of the path satisfies): a k-step never leaves one filter tap, so the k-walk
// (tap, ci) is wave-uniform and lives in SGPRs; each lane keeps ONE 32-bit byte offset per piece plus a bit mask of the taps
// that fall inside the image.  Loads are buffer_load_dwordx4 ... lds through a buffer descriptor: the per-step uniform part
// goes in soffset, and a lane whose tap is padding (or whose row is >= M) passes an out-of-range voffset, which the hardware
// range check turns into a zero fill.  This removes ~90 of the ~110 instructions per k-step that the general form spends
// on 64-bit address arithmetic and divergent tap bookkeeping (PMC: VALU busy 55 % vs MFMA busy 15 % on the 3x3 layers).
//
// PC (producer / consumer, 512 threads): waves 4..7 only issue the DMA pieces and count them in, waves 0..3 only read fragments
// and run the MFMAs.  An LDS-DMA piece costs its issuing wave ~176 cycles of issue time in a mixed phase (idbank.hip ablation),
// 8 pieces per wave and k-step at 128x128 -- as much as the 512 MFMA cycles they feed; on separate waves the two overlap.
template <bool IS1X1, bool SPLITK, int ST = 1, int BM = 64, int BN = 64, int MODE = 0, bool PC = false>
__device__ __forceinline__ void conv_gemm_dma_body(const ConvParams& p, const int zslice) {
  constexpr bool FAST = MODE != 0;
  constexpr int NT = PC ? 512 : 256;
  static_assert(!PC || ST >= 2, "the producer / consumer form needs a ring");
  constexpr bool DUAL = MODE == 3;        // Y = act([x | x2 sampled] * Wcat^T + b): a bottleneck's conv3 and its strided 1x1 shortcut as one GEMM
  constexpr bool ROWRUN = MODE == 2;      // Cin % 64 != 0 (stem 7x7x8, id bank 17x17x16): the KW * Cin elements of one filter row are
                                          // contiguous in NHWC, so a filter row is walked as spr = ceil(KW * Cin / 64) k-steps
  constexpr int BK = 64, TM = BM / 32, TN = BN / 32;   // 2x2 waves, each (BM/2) x (BN/2) = TM x TN tiles of 16x16
  constexpr int NA = BM / 32, NB = BN / 32;             // DMA wave-instructions (8 rows x 128 B each) per wave and k-step
  constexpr int CP = BN + 4;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  static_assert(32 * CP * 4 <= ST * STAGE_BYTES, "epilogue staging must fit the ring");
  __shared__ __attribute__((aligned(16))) char smem[ST * STAGE_BYTES];   // the ONLY shared object (epilogue staging aliases it)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA destinations stay in SGPRs
  const bool loader = PC && wave_all >= 4;
  const int wave = PC ? (wave_all & 3) : wave_all;      // index inside the role: piece rows for a loader, tile for a consumer
  const int wm = wave >> 1, wn = wave & 1;
  int tbx, tby;
  if (!tile_of_block(p, BM, tbx, tby)) return;          // (workgroup-uniform: surplus block of the XCD-aware grid)
  const int m0 = tbx * BM, n0 = tby * BN;
  const int rr_len = p.KW * p.Cin;                        // ROWRUN: elements per filter row, k-steps per filter row
  const int rr_spr = (rr_len + BK - 1) / BK;
  const int nk_total = ROWRUN ? p.KH * rr_spr : (p.K + BK - 1) / BK;
  const int kt0 = SPLITK ? zslice * p.steps_per_split : 0;
  const int kt1 = SPLITK ? min(nk_total, kt0 + p.steps_per_split) : nk_total;

  // this lane's (row, chunk) pieces of the A tile and of the B tile: instruction i of a wave covers rows (BM/4)*wave + 8*i .. +7
  int a_c[NA], a_hi0[NA], a_wi0[NA], a_ci[NA], a_kw[NA], a_kh[NA];
  long a_base[NA];
  bool a_ok[NA], b_ok[NB];
  int b_c[NB];
  long b_base[NB];
  // FAST: per-lane byte offsets (31 bits) and tap masks; wave-uniform k-walk
  constexpr int OOB = (int)0x80000000;
  int f_aoff[NA], f_boff[NB];
  unsigned f_amask[NA], f_cmask[NA];
  int f_aoff2[NA];
  const int nk1 = p.Cin / BK;                             // DUAL: k-steps served by x
  int s_ci = 0, s_kw = 0, s_tap = 0, s_aoff = 0;         // FAST is never split along K: the walk starts at tap 0, channel 0
  int s_boff = 0;                                        // ROWRUN: s_tap = filter row, s_kw = step inside the row
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = (BM / 4) * wave + 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const int m = m0 + r;
      if (IS1X1) {
        f_aoff[i] = m < p.M ? (m * p.ldx + c * 8) * 2 : OOB;
        f_amask[i] = 1u;
        f_cmask[i] = 0u;
        if (DUAL) {
          const int img = m / p.HoWo, rem = m - img * p.HoWo;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          f_aoff2[i] = m < p.M ? (((img * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + c * 8) * 2 : OOB;
        }
      } else {
        const int img = m / p.HoWo, rem = m - img * p.HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        // offset relative to the (virtual) pixel (-pad, -pad) of image 0, so that it is never negative
        f_aoff[i] = (((img * p.H + hi0 + p.pad) * p.W + wi0 + p.pad) * p.Cin + c * 8) * 2;
        unsigned cm = 0, mask = 0;
        if (ROWRUN) {           // bit kh of mask: filter row inside the image; bit j of cm: this chunk of step j is a real tap inside it
          for (int kh = 0; kh < p.KH; ++kh) mask |= ((unsigned)(hi0 + kh) < (unsigned)p.H ? 1u : 0u) << kh;
          for (int j = 0; j < rr_spr; ++j) {
            const int e = j * BK + c * 8;
            if (e < rr_len && (unsigned)(wi0 + e / p.Cin) < (unsigned)p.W) cm |= 1u << j;
          }
          f_cmask[i] = cm;
        } else {
          for (int kw = 0; kw < p.KW; ++kw) cm |= ((unsigned)(wi0 + kw) < (unsigned)p.W ? 1u : 0u) << kw;
          for (int kh = 0; kh < p.KH; ++kh)
            if ((unsigned)(hi0 + kh) < (unsigned)p.H) mask |= cm << (kh * p.KW);
        }
        f_amask[i] = m < p.M ? mask : 0u;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = (BN / 4) * wave + 8 * i + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const int n = n0 + r;
      // ROWRUN: chunks past the end of a filter row read the next row's weights (or 0 past the buffer); A is zero there
      f_boff[i] = n < p.Cout ? (n * p.K + c * 8) * 2 : OOB;
    }
  } else {
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (BM / 4) * wave + 8 * i + (lane >> 3);
    a_c[i] = (lane & 7) ^ ((r >> 1) & 7);              // logical chunk that belongs in physical slot (lane & 7) of row r
    const int m = m0 + r;
    a_ok[i] = m < p.M;
    if (IS1X1) {
      a_base[i] = (long)m * p.ldx;
      a_hi0[i] = a_wi0[i] = a_ci[i] = a_kw[i] = a_kh[i] = 0;
    } else {
      const int img = m / p.HoWo, rem = m - img * p.HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride - p.pad;
      a_wi0[i] = wo * p.stride - p.pad;
      a_base[i] = (long)img * p.H * p.W * p.Cin;
      const int kidx = kt0 * BK + a_c[i] * 8;
      const int kk = kidx / p.Cin;
      a_ci[i] = kidx - kk * p.Cin;
      a_kh[i] = kk / p.KW;
      a_kw[i] = kk - a_kh[i] * p.KW;
    }
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (BN / 4) * wave + 8 * i + (lane >> 3);
    b_c[i] = (lane & 7) ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_ok[i] = n < p.Cout;
    b_base[i] = (long)n * p.K;
  }
  }
  const char* zero = reinterpret_cast<const char*>(g_zero16);
  // FAST: descriptors built from kernel arguments only (wave-uniform); A starts at the virtual pixel (-pad, -pad)
  const long a_shift = (FAST && !IS1X1) ? ((long)p.pad * p.W + p.pad) * p.Cin : 0;
  const rsrc_t rsrc_a = make_rsrc(p.x - a_shift, FAST ? (int)((p.x_elems + a_shift) * 2) : 0);
  const rsrc_t rsrc_b = make_rsrc(p.w, FAST ? p.Cout * p.K * 2 : 0);
  const rsrc_t rsrc_a2 = make_rsrc(DUAL ? p.x2 : p.x, DUAL ? (int)(p.x2_elems * 2) : 0);

  auto issue = [&](int kt, int stage) {
    char* As = smem + stage * STAGE_BYTES;
    char* Bs = As + BM * BK * 2;
    if constexpr (FAST) {
      const int soff_a = IS1X1 ? kt * (BK * 2) : s_aoff;
      if (DUAL && kt >= nk1) {  // wave-uniform: this k-step comes from the second source
#pragma unroll
        for (int i = 0; i < NA; ++i)
          buf_load_lds16(rsrc_a2, (lptr_t)(As + ((BM / 4) * wave + 8 * i) * 128), f_aoff2[i], (kt - nk1) * (BK * 2));
      } else
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int voff = IS1X1 ? f_aoff[i]
                         : ROWRUN ? ((((f_amask[i] >> s_tap) & (f_cmask[i] >> s_kw)) & 1u) ? f_aoff[i] : OOB)
                                  : (((f_amask[i] >> s_tap) & 1u) ? f_aoff[i] : OOB);
        buf_load_lds16(rsrc_a, (lptr_t)(As + ((BM / 4) * wave + 8 * i) * 128), voff, soff_a);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
        buf_load_lds16(rsrc_b, (lptr_t)(Bs + ((BN / 4) * wave + 8 * i) * 128), f_boff[i], ROWRUN ? s_boff : kt * (BK * 2));
      if (ROWRUN) {             // next 64 elements of this filter row, or the start of the next row
        s_aoff += BK * 2;
        s_boff += BK * 2;
        if (++s_kw == rr_spr) {
          s_kw = 0;
          ++s_tap;
          s_aoff += (p.W * p.Cin - rr_spr * BK) * 2;
          s_boff += (rr_len - rr_spr * BK) * 2;
        }
      } else if (!IS1X1) {      // advance the uniform k-walk by one step (64 channels of one tap)
        s_ci += BK;
        s_aoff += BK * 2;
        if (s_ci == p.Cin) {
          s_ci = 0;
          ++s_tap;
          // the next tap of the same filter row is the next pixel: its + Cin channels are already in s_aoff
          if (++s_kw == p.KW) { s_kw = 0; s_aoff += (p.W - p.KW) * p.Cin * 2; }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const char* src = zero;
      if (IS1X1) {
        const int kidx = kt * BK + a_c[i] * 8;
        if (a_ok[i] && kidx < p.K) src = reinterpret_cast<const char*>(p.x + a_base[i] + kidx);
      } else {
        const int hi = a_hi0[i] + a_kh[i], wi = a_wi0[i] + a_kw[i];
        if (a_ok[i] && a_kh[i] < p.KH && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
          src = reinterpret_cast<const char*>(p.x + a_base[i] + ((long)hi * p.W + wi) * p.Cin + a_ci[i]);
        a_ci[i] += BK;
        while (a_ci[i] >= p.Cin) {
          a_ci[i] -= p.Cin;
          if (++a_kw[i] == p.KW) { a_kw[i] = 0; ++a_kh[i]; }
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + ((BM / 4) * wave + 8 * i) * 128), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int kidx = kt * BK + b_c[i] * 8;
      const char* src = (b_ok[i] && kidx < p.K) ? reinterpret_cast<const char*>(p.w + b_base[i] + kidx) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + ((BN / 4) * wave + 8 * i) * 128), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fc = lane >> 4;
  // tile kt has landed once only the NA + NB DMA pieces per later tile (at most ST - 2 of them) are still outstanding for this wave
  auto wait_tile = [&](int kt) {
    static_assert((ST - 2) * (NA + NB) <= 63, "vmcnt is a 6-bit counter");
    switch (min(ST - 2, kt1 - 1 - kt)) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NB)) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST >= 5 ? 3 * (NA + NB) : 0) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST >= 6 ? 4 * (NA + NB) : 0) : "memory"); break;
    }
  };
  auto compute = [&](int stage) {
    const e16* As = reinterpret_cast<const e16*>(smem + stage * STAGE_BYTES);
    const e16* Bs = As + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      e16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const e16x8*>(&As[swz(wm * (BM / 2) + i * 16 + fr, 4 * ks + fc)]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const e16x8*>(&Bs[swz(wn * (BN / 2) + j * 16 + fr, 4 * ks + fc)]);
      if (p.debug & 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bfr[j]));
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = RMEM_MFMA_16x16x32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  if constexpr (PC) {
    // both roles pass the same kt1 - kt0 barriers: barrier kt = "tile kt landed, tile kt-1 is no longer read"
    if (loader) {
#pragma unroll
      for (int i = 0; i < ST - 1; ++i)
        if (kt0 + i < kt1) issue(kt0 + i, i);
      int stage = 0;
      for (int kt = kt0; kt < kt1; ++kt) {
        wait_tile(kt);
        __builtin_amdgcn_s_barrier();
        const int nxt = stage == 0 ? ST - 1 : stage - 1;
        if (kt + ST - 1 < kt1) issue(kt + ST - 1, nxt);
        stage = stage == ST - 1 ? 0 : stage + 1;
      }
    } else {
      int stage = 0;
      for (int kt = kt0; kt < kt1; ++kt) {
        __builtin_amdgcn_s_barrier();
        compute(stage);
        stage = stage == ST - 1 ? 0 : stage + 1;
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < ST - 1; ++i)
    if (kt0 + i < kt1) issue(kt0 + i, i);
  int stage = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    if (ST == 1) {                                    // single buffer: other resident workgroups hide the load
      if (kt > kt0) __builtin_amdgcn_s_barrier();     // everyone finished reading tile kt-1
      if (!(p.debug & 2) || kt == kt0) issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      wait_tile(kt);
      __builtin_amdgcn_s_barrier();                     // everyone's pieces of tile kt landed; everyone finished reading tile kt-1
      const int nxt = stage == 0 ? ST - 1 : stage - 1;  // slot of tile kt+ST-1 == slot of tile kt-1
      if (kt + ST - 1 < kt1 && !(p.debug & 2)) issue(kt + ST - 1, nxt);
    }
    compute(stage);
    stage = stage == ST - 1 ? 0 : stage + 1;
  }
  }
  __syncthreads();                                     // all DMA drained (vmcnt(0) above) and all fragment reads done
  if (p.debug & 4) {                                   // timing experiment: keep the accumulators alive, store nothing
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(acc[i][j]));
    return;
  }

  // epilogue through LDS, 32 output rows per pass: pass -> wave row wm = pass / (TM/2), its tiles i = 2*(pass % (TM/2)), +1
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int VPR = BN / 8;
  constexpr int PPW = TM / 2;                          // passes per wave row
#pragma unroll
  for (int pass = 0; pass < 2 * PPW; ++pass) {
    if (!loader && wm == pass / PPW) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[(ii * 16 + fc * 4 + r) * CP + wn * (BN / 2) + j * 16 + fr] = acc[2 * (pass % PPW) + ii][j][r];
    }
    __syncthreads();
    for (int vi = tid; vi < 32 * VPR; vi += NT) {
      const int row = vi / VPR, cv = vi - row * VPR;
      const int m = m0 + (pass / PPW) * (BM / 2) + (pass % PPW) * 32 + row;
      const int n = n0 + cv * 8;
      if (m >= p.M || n >= p.Cout) continue;
      const float* c = Cs + row * CP + cv * 8;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
      float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      if (p.debug & 8) {
        asm volatile("" :: "v"(v[0]), "v"(v[7]));
      } else if (SPLITK) {
        float* s = p.slabs + ((long)zslice * p.M + m) * p.Cout + n;
        *reinterpret_cast<f32x4*>(s) = c0;
        *reinterpret_cast<f32x4*>(s + 4) = c1;
      } else if (p.vec_ok && n + 8 <= p.Cout) {
        finish8(p, m, n, v);
      } else {
        for (int j = 0; j < 8 && n + j < p.Cout; ++j) finish1(p, m, n + j, v[j]);
      }
    }
    __syncthreads();
  }
}


template <bool IS1X1, bool SPLITK, int ST = 1, int MODE = 0>
__global__ __launch_bounds__(256) void k_conv_gemm_dma(ConvParams p) {
  conv_gemm_dma_body<IS1X1, SPLITK, ST, 64, 64, MODE>(p, blockIdx.z);
}

// larger block tiles for the many-row problems (batched encoder / decoder convs): a 64x64 tile moves 16 KB global -> LDS per
// 0.52 MFLOP, which caps a CU at its ~70 GB/s L2 -> LDS rate (MI355X_MICROARCH.md, 'Indexed rows: gather into LDS')
template <bool IS1X1, int ST, int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void k_conv_gemm_dma_big(ConvParams p) {
  conv_gemm_dma_body<IS1X1, false, ST, BM, BN, MODE>(p, 0);
}

template <bool IS1X1, int ST, int MODE>
__global__ __launch_bounds__(512) void k_conv_gemm_dma_pc(ConvParams p) {
  conv_gemm_dma_body<IS1X1, false, ST, 128, 128, MODE, true>(p, 0);
}

// up to 4 GEMMs of identical shape (different operands) as ONE launch: blockIdx.z selects the operand set.  The per-layer
// memory-update linears of a frame are independent of each other (layers/transformer.py:269-322), and at M = 1674 a launch
// costs more than its arithmetic.
struct GroupPtrs {
  const e16* x[4]; const e16* w[4]; const float* bias[4]; const void* res[4]; void* y[4]; e16* y2[4];
};
template <bool FAST, int ST = 1>
__global__ __launch_bounds__(256) void k_gemm_dma_grouped(ConvParams p, GroupPtrs g) {
  const int z = blockIdx.z;
  ConvParams q = p;
  q.x = g.x[z]; q.w = g.w[z]; q.bias = g.bias[z]; q.res = g.res[z]; q.y = g.y[z]; q.y2 = g.y2[z];
  conv_gemm_dma_body<true, false, ST, 64, 64, FAST ? 1 : 0>(q, 0);
}

// sum the split-K slabs in slice order and apply the fused epilogue; thread = 8 channels of one row
__global__ __launch_bounds__(256) void k_splitk_epilogue(ConvParams p, int splits) {
  const int vpr = p.Cout / 8;
  const long total = (long)p.M * vpr;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int m = (int)(i / vpr), n = (int)(i - (long)m * vpr) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < splits; ++z) {
    const float* s = p.slabs + ((long)z * p.M + m) * p.Cout + n;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] += a[j]; v[4 + j] += b[j]; }
  }
  if (p.vec_ok) finish8(p, m, n, v);
  else
    for (int j = 0; j < 8; ++j) finish1(p, m, n + j, v[j]);
}

template <int BM, int BN, int PF>
void launch(const ConvParams& p, bool is1x1, int splits, hipStream_t s) {
  dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, splits);
  static const bool use_dma = !(getenv("RMEM_GEMM_DMA") && atoi(getenv("RMEM_GEMM_DMA")) == 0);   // kernel experiments only
  if (BM == 64 && BN == 64 && use_dma) {
    if (splits > 1) {
      if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma<true, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((k_conv_gemm_dma<false, true>), grid, dim3(256), 0, s, p);
      const long total = (long)p.M * (p.Cout / 8);
      hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, splits);
    } else {
      ConvParams q = p;
      static const bool xcd_on = !(getenv("RMEM_GEMM_XCD") && atoi(getenv("RMEM_GEMM_XCD")) == 0);   // kernel experiments only
      if (xcd_on && grid.y > 1 && grid.x >= 16) {        // XCD-aware order: the column tiles of a row tile share one L2
        q.xcd_ny = (int)grid.y;
        grid = dim3(8 * ((grid.x + 7) / 8) * grid.y, 1, 1);
      }
      const ConvParams& p = q;
      // ring depth: measured with 4 clips per launch, 1 / 2 / 3 / 4 stages give 2146 / 2044 / 1978 / 1735 frames/s -- a
      // 16 KB single buffer lets 8 workgroups share a CU, and their DMA in flight beats any prefetch depth inside one
      static const int st = getenv("RMEM_GEMM_ST") ? atoi(getenv("RMEM_GEMM_ST")) : 1;     // kernel experiments only
      if (st == 2) {
        if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma<true, false, 2>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_conv_gemm_dma<false, false, 2>), grid, dim3(256), 0, s, p);
      } else if (st == 3) {
        if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma<true, false, 3>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_conv_gemm_dma<false, false, 3>), grid, dim3(256), 0, s, p);
      } else if (p.fast_ok) {
        // few-workgroup problems (token GEMMs of the LSTT, id bank): too few resident workgroups to hide the DMA latency behind
        // each other, so a 3-deep ring keeps two k-steps in flight inside the workgroup
        static const int deep_wgs = getenv("RMEM_GEMM_DEEP_WGS") ? atoi(getenv("RMEM_GEMM_DEEP_WGS")) : 1024;
        const bool deep = (long)((p.M + 63) / 64) * ((p.Cout + 63) / 64) <= deep_wgs && p.steps_per_split >= 3;
        if (p.fast_ok == 2) {
          if (deep) hipLaunchKernelGGL((k_conv_gemm_dma<false, false, 3, 2>), grid, dim3(256), 0, s, p);
          else hipLaunchKernelGGL((k_conv_gemm_dma<false, false, 1, 2>), grid, dim3(256), 0, s, p);
        } else if (is1x1) {
          if (deep) hipLaunchKernelGGL((k_conv_gemm_dma<true, false, 3, 1>), grid, dim3(256), 0, s, p);
          else hipLaunchKernelGGL((k_conv_gemm_dma<true, false, 1, 1>), grid, dim3(256), 0, s, p);
        } else {
          if (deep) hipLaunchKernelGGL((k_conv_gemm_dma<false, false, 3, 1>), grid, dim3(256), 0, s, p);
          else hipLaunchKernelGGL((k_conv_gemm_dma<false, false, 1, 1>), grid, dim3(256), 0, s, p);
        }
      } else {
        if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma<true, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_conv_gemm_dma<false, false>), grid, dim3(256), 0, s, p);
      }
    }
    return;
  }
  if (splits > 1) {
    if (is1x1) hipLaunchKernelGGL((k_conv_gemm<BM, BN, PF, true, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_conv_gemm<BM, BN, PF, false, true>), grid, dim3(256), 0, s, p);
    const long total = (long)p.M * (p.Cout / 8);
    hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, splits);
  } else {
    if (is1x1) hipLaunchKernelGGL((k_conv_gemm<BM, BN, PF, true, false>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_conv_gemm<BM, BN, PF, false, false>), grid, dim3(256), 0, s, p);
  }
}

template <int BM, int BN, int MODE = 1>
void launch_big(const ConvParams& pin, bool is1x1, int st, hipStream_t s) {
  dim3 grid((pin.M + BM - 1) / BM, (pin.Cout + BN - 1) / BN, 1);
  ConvParams p = pin;
  static const bool xcd_on = !(getenv("RMEM_GEMM_XCD") && atoi(getenv("RMEM_GEMM_XCD")) == 0);   // kernel experiments only
  if (xcd_on && grid.y > 1 && grid.x >= 16) {
    p.xcd_ny = (int)grid.y;
    grid = dim3(8 * ((grid.x + 7) / 8) * grid.y, 1, 1);
  }
  if constexpr (BM == 128 && BN == 128 && MODE == 1) {
    // producer / consumer form (loader waves + MFMA waves), ring depth RMEM_GEMM_PC (0 = off).  Measured, 16 images / 8 clips per
    // launch, bit-identical outputs: a 2-deep ring (64 KB, two workgroups = 16 waves per CU) takes the K >= 512 layers from
    // 70.3 / 46.4 / 58.0 / 66.1 / 50.7 / 27.5 / 49.9 us to 57.7 / 38.7 / 49.6 / 60.6 / 42.9 / 21.1 / 44.2 us alone and the whole
    // pipeline from 3371 to 3404 frames/s (three A/B pairs); 3- and 4-deep rings (one workgroup per CU) are as fast alone but
    // lose 3 % in the pipeline, where the other streams' kernels want the LDS
    static const int pc_env = getenv("RMEM_GEMM_PC") ? atoi(getenv("RMEM_GEMM_PC")) : 2;
    const int pc = pc_env == 1 ? (st >= 3 ? 3 : 2) : pc_env;     // 1 = by shape: the few-tile deep-K problems keep the 3-deep ring
    if (pc >= 2) {
      if (pc >= 4) {
        if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_pc<true, 4, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_conv_gemm_dma_pc<false, 4, 1>), grid, dim3(512), 0, s, p);
      } else if (pc == 3) {
        if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_pc<true, 3, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_conv_gemm_dma_pc<false, 3, 1>), grid, dim3(512), 0, s, p);
      } else {
        if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_pc<true, 2, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((k_conv_gemm_dma_pc<false, 2, 1>), grid, dim3(512), 0, s, p);
      }
      return;
    }
  }
  if constexpr (BM == 128 && BN == 128) {
    if (st == 5) {        // 160 KB of LDS: one workgroup per CU with four k-steps (128 KB) in flight
      if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_big<true, 5, BM, BN, MODE>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((k_conv_gemm_dma_big<false, 5, BM, BN, MODE>), grid, dim3(256), 0, s, p);
      return;
    }
    if (st == 4) {
      if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_big<true, 4, BM, BN, MODE>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((k_conv_gemm_dma_big<false, 4, BM, BN, MODE>), grid, dim3(256), 0, s, p);
      return;
    }
  }
  if (st == 3) {
    if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_big<true, 3, BM, BN, MODE>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_conv_gemm_dma_big<false, 3, BM, BN, MODE>), grid, dim3(256), 0, s, p);
  } else if (st == 2) {
    if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_big<true, 2, BM, BN, MODE>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_conv_gemm_dma_big<false, 2, BM, BN, MODE>), grid, dim3(256), 0, s, p);
  } else {
    if (is1x1) hipLaunchKernelGGL((k_conv_gemm_dma_big<true, 1, BM, BN, MODE>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_conv_gemm_dma_big<false, 1, BM, BN, MODE>), grid, dim3(256), 0, s, p);
  }
}

// split-K plan shared by rmem_conv_workspace_bytes and the launcher (64x64 tiles only)
int plan_splits(int M, int Cout, int K) {
  if (Cout % 8) return 1;
  const long tiles = (long)((M + 63) / 64) * ((Cout + 63) / 64);
  const int nk = (K + 63) / 64;
  if (tiles >= 192 || nk < 24) return 1;          // measured: K = 1024 (16 steps) is faster unsplit, K >= 2304 split
  int s = (int)((448 + tiles - 1) / tiles);      // aim at >= ~450 workgroups
  s = min(s, nk / 4);                            // keep >= 4 k-steps (of 64) per slice
  return max(1, min(s, 16));
}

// Measured on MI355X over every conv / linear shape of the path (M = 1674 .. 102425 rows, and 4 - 16 images or clips per
// launch, M up to 412 k): the LDS-DMA 64x64 tile at 4 workgroups per CU is never slower than the register-ring 128x64 /
// 128x128 kernels (1981 vs 1962 frames/s with 4 clips per launch), so it is used for everything; the larger tiles stay
// reachable through RMEM_GEMM_TILE for experiments.
bool use_small_tiles(int M, int Cout) {
  (void)M; (void)Cout;
  return true;
}

}  // namespace

#ifndef RMEM_F16
extern "C" size_t rmem_conv_workspace_bytes(const rmem_conv_desc* d) {
  if (!d) return 0;
  const int M = (d->batch > 0 ? d->batch : 1) * d->Ho * d->Wo, K = d->KH * d->KW * d->Cin;
  if (!use_small_tiles(M, d->Cout)) return 0;
  const int s = plan_splits(M, d->Cout, K);
  return s > 1 ? (size_t)s * M * d->Cout * sizeof(float) : 0;
}
#endif

static int conv_setup(const rmem_conv_desc* d, const void* x, const void* w, const float* bias, const void* residual, void* y,
                      void* y2, ConvParams& p, bool& is1x1) {
  RMEM_REQUIRE(d && x && w && y, "rmem_conv2d_nhwc: null argument");
  RMEM_REQUIRE(d->Cin > 0 && d->Cin % 8 == 0, "rmem_conv2d_nhwc: Cin must be a positive multiple of 8");
  RMEM_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "rmem_conv2d_nhwc: bad kernel geometry");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  RMEM_REQUIRE(Ho == d->Ho && Wo == d->Wo && Ho > 0 && Wo > 0, "rmem_conv2d_nhwc: Ho/Wo do not match the geometry");
  RMEM_REQUIRE(d->ldo >= d->Cout, "rmem_conv2d_nhwc: ldo < Cout");
  RMEM_REQUIRE(!residual || d->ldr >= d->Cout, "rmem_conv2d_nhwc: ldr < Cout");
  RMEM_REQUIRE(!y2 || d->ld2 >= d->Cout, "rmem_conv2d_nhwc: ld2 < Cout");
  RMEM_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, "rmem_conv2d_nhwc: x/w must be 16-byte aligned");
  p.x = (const e16*)x; p.w = (const e16*)w; p.bias = bias; p.res = residual; p.y = y; p.y2 = (e16*)y2;
  p.slabs = nullptr;
  p.xcd_ny = 0;
  { static const int dbg = getenv("RMEM_GEMM_DEBUG") ? atoi(getenv("RMEM_GEMM_DEBUG")) : 0; p.debug = dbg; }
  p.x2 = nullptr; p.H2 = p.W2 = p.Cin2 = p.stride2 = 0; p.x2_elems = 0;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
  const int nb = d->batch > 0 ? d->batch : 1;
  p.HoWo = Ho * Wo; p.M = nb * Ho * Wo; p.K = d->KH * d->KW * d->Cin;
  p.ldo = d->ldo; p.ldr = d->ldr; p.ld2 = d->ld2;
  p.relu = d->relu; p.out_f32 = d->out_f32; p.res_f32 = d->res_f32;
  is1x1 = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0;
  RMEM_REQUIRE(d->relu >= 0 && d->relu <= 3, "rmem_conv2d_nhwc: relu must be 0 (none), 1 (ReLU), 2 (GELU) or 3 (SiLU)");
  RMEM_REQUIRE(d->act_begin >= 0 && d->act_begin % 8 == 0, "rmem_conv2d_nhwc: act_begin must be a non-negative multiple of 8");
  RMEM_REQUIRE(d->ldx == 0 || (is1x1 && d->ldx >= d->Cin && d->ldx % 8 == 0),
               "rmem_conv2d_nhwc: ldx needs a 1x1 stride-1 problem, ldx >= Cin, ldx % 8 == 0");
  p.ldx = d->ldx ? d->ldx : d->Cin;
  p.act_begin = d->act_begin;
  p.up_h = d->res_up_h; p.up_w = d->res_up_w; p.up_align = d->res_up_align;
  RMEM_REQUIRE(p.up_h >= 0 && p.up_w >= 0 && (p.up_h > 0) == (p.up_w > 0), "rmem_conv2d_nhwc: res_up_h / res_up_w must both be set or both be 0");
  RMEM_REQUIRE(p.up_h == 0 || (residual && !d->res_f32 && p.Cout % 8 == 0 && p.ldr % 8 == 0 && ((uintptr_t)residual % 16) == 0 && p.ldo % 8 == 0),
               "rmem_conv2d_nhwc: the resized residual must be e16, 16-byte aligned, with Cout, ldr, ldo multiples of 8");
  p.steps_per_split = (p.K + 63) / 64;
  auto al = [](const void* q, int a) { return q == nullptr || ((uintptr_t)q % a) == 0; };
  {
    static const bool fast_on = !(getenv("RMEM_GEMM_FAST") && atoi(getenv("RMEM_GEMM_FAST")) == 0);   // kernel experiments only
    p.x_elems = is1x1 ? (long)(p.M - 1) * p.ldx + p.Cin : (long)nb * p.H * p.W * p.Cin;
    const long shift = is1x1 ? 0 : ((long)p.pad * p.W + p.pad) * p.Cin;
    const long lim = (1L << 31) - (1L << 22);          // every in-range byte offset stays below 2^31 (masked lanes may wrap: unused)
    const bool small = (p.x_elems + shift) * 2 < lim && (long)p.Cout * p.K * 2 < lim;
    p.fast_ok = fast_on && small && p.Cin % 64 == 0 && p.KH * p.KW <= 32;
    if (!p.fast_ok && fast_on && small && !is1x1 && p.KH <= 32 && (p.KW * p.Cin + 63) / 64 <= 32 && p.KW * p.Cin >= 48)
      p.fast_ok = 2;                                     // row-run form (7x7x8 stem, 17x17x16 id bank, 4x4x8 patch embedding)
  }
  p.vec_ok = p.Cout % 8 == 0 && p.ldo % 8 == 0 && al(y, 16) && al(bias, 16) &&
             (!residual || (p.ldr % 8 == 0 && al(residual, 16))) && (!y2 || (p.ld2 % 8 == 0 && al(y2, 16)));
  return 0;
}

extern "C" int RMEM_API(rmem_conv2d_nhwc)(const rmem_conv_desc* d, const void* x, const void* w, const float* bias,
                                const void* residual, void* y, void* y2, void* workspace, void* stream) {
  ConvParams p;
  bool is1x1 = false;
  if (conv_setup(d, x, w, bias, residual, y, y2, p, is1x1)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
  static const int force_tile = getenv("RMEM_GEMM_TILE") ? atoi(getenv("RMEM_GEMM_TILE")) : -1;   // kernel experiments only
  if (force_tile == 0) { launch<64, 64, 4>(p, is1x1, 1, s); return rmem_check_launch("rmem_conv2d_nhwc"); }
  if (force_tile == 1) { launch<128, 64, 3>(p, is1x1, 1, s); return rmem_check_launch("rmem_conv2d_nhwc"); }
  if (force_tile == 2) { launch<128, 128, 2>(p, is1x1, 1, s); return rmem_check_launch("rmem_conv2d_nhwc"); }
  if (!use_small_tiles(p.M, p.Cout)) {
    if (p.Cout >= 128 && t128 >= 384) launch<128, 128, 2>(p, is1x1, 1, s);
    else launch<128, 64, 3>(p, is1x1, 1, s);
  } else {
    int splits = workspace ? plan_splits(p.M, p.Cout, p.K) : 1;
    if (splits > 1) {
      const int nk = (p.K + 63) / 64;
      p.steps_per_split = (nk + splits - 1) / splits;
      splits = (nk + p.steps_per_split - 1) / p.steps_per_split;   // no empty slice
      p.slabs = (float*)workspace;
    }
    // 128x128 tiles halve the global -> LDS bytes per flop; they pay only where the k-loop dominates (K >= 512) and there are
    // enough tiles to balance 256 CUs.  Measured per layer with 8 images / 4 clips per launch: 121x213 3x3 128->128 563 -> 662
    // TFLOP/s, 512->1024 stride 2 393 -> 470; shallow-K 1x1 layers (64->256, 128->512) lose 20-40 % and stay on 64x64.
    static const int big_thr = getenv("RMEM_GEMM_BIG") ? atoi(getenv("RMEM_GEMM_BIG")) : 128;
    // (round 3, 16 images / 8 clips per launch, measured in the whole pipeline where other streams' kernels share the CUs: K = 256
    // layers -- layer-3 conv3, the decoder's 1x1s -- are better off on 64x64 tiles: 3276 -> 3315 frames/s, three A/B pairs)
    static const int big_k = getenv("RMEM_GEMM_BIG_K") ? atoi(getenv("RMEM_GEMM_BIG_K")) : 512;
    static const int big_st = getenv("RMEM_GEMM_BIG_ST") ? atoi(getenv("RMEM_GEMM_BIG_ST")) : 1;
    if (splits == 1 && big_thr > 0 && p.fast_ok == 1 && p.Cout >= 128 && p.K >= big_k &&
        (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) >= big_thr) {
      // at most ~1 workgroup per CU and a deep k-loop: nothing else hides the DMA latency, so keep two k-steps in flight
      static const int big_deep = getenv("RMEM_GEMM_BIG_DEEP") ? atoi(getenv("RMEM_GEMM_BIG_DEEP")) : 256;
      const long t128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
      static const int deep_st = getenv("RMEM_GEMM_BIG_DEEP_ST") ? atoi(getenv("RMEM_GEMM_BIG_DEEP_ST")) : 3;
      launch_big<128, 128>(p, is1x1, (t128 <= big_deep && p.K >= 512) ? deep_st : big_st, s);
      return rmem_check_launch("rmem_conv2d_nhwc");
    }
    // row-run problems on the larger tiles: their k-loop is all global -> LDS traffic as well.  Measured (16 images / 8 clips per
    // launch): the id bank (17x17x16 -> 256, 85 k-steps) 105.8 -> 81.2 us on 128x128 tiles with a 3-deep ring (bit 0, the default);
    // the stem (7x7x8 -> 64) on 128x64 tiles (bit 1) 213.8 -> 195.2 us alone with a single buffer but no faster end to end: off
    static const int rr_big = getenv("RMEM_GEMM_ROWRUN_BIG") ? atoi(getenv("RMEM_GEMM_ROWRUN_BIG")) : 1;
    static const int rr_st = getenv("RMEM_GEMM_ROWRUN_ST") ? atoi(getenv("RMEM_GEMM_ROWRUN_ST")) : 3;
    if (splits == 1 && p.fast_ok == 2 && (rr_big & 1) && p.Cout >= 128) {
      launch_big<128, 128, 2>(p, false, rr_st, s);
      return rmem_check_launch("rmem_conv2d_nhwc");
    }
    if (splits == 1 && p.fast_ok == 2 && (rr_big & 2) && p.Cout <= 64) {
      launch_big<128, 64, 2>(p, false, rr_st > 3 ? 3 : rr_st, s);
      return rmem_check_launch("rmem_conv2d_nhwc");
    }
    static const int big64 = getenv("RMEM_GEMM_BIG64") ? atoi(getenv("RMEM_GEMM_BIG64")) : 0;
    if (splits == 1 && big64 > 0 && p.fast_ok == 1 && p.Cout <= 64 && p.K >= big_k && (p.M + 127) / 128 >= big64) {
      launch_big<128, 64>(p, is1x1, big_st, s);
      return rmem_check_launch("rmem_conv2d_nhwc");
    }
    launch<64, 64, 4>(p, is1x1, splits, s);
  }
  return rmem_check_launch("rmem_conv2d_nhwc");
}

// Y = act([x | x2 sampled at stride2] * Wcat^T + bias): the last 1x1 conv of a ResNet bottleneck and its (strided) 1x1 shortcut
// (encoders/resnet.py:48-68, downsample branch) as ONE GEMM over K = Cin + Cin2 -- the shortcut tensor is never written or re-read.
extern "C" int RMEM_API(rmem_conv1x1_dual_nhwc)(const rmem_conv_desc* d, const void* x, const void* x2, int H2, int W2, int Cin2, int stride2,
                                      const void* w_cat, const float* bias, void* y, void* stream) {
  ConvParams p;
  bool is1x1 = false;
  RMEM_REQUIRE(d && x2 && H2 > 0 && W2 > 0 && stride2 >= 1, "rmem_conv1x1_dual_nhwc: bad second source");
  if (conv_setup(d, x, w_cat, bias, nullptr, y, nullptr, p, is1x1)) return -1;
  RMEM_REQUIRE(is1x1 && d->ldx == 0, "rmem_conv1x1_dual_nhwc: the main problem must be a dense 1x1 stride-1 convolution");
  RMEM_REQUIRE(p.Cin % 64 == 0 && Cin2 % 64 == 0, "rmem_conv1x1_dual_nhwc: Cin and Cin2 must be multiples of 64");
  RMEM_REQUIRE((H2 - 1) / stride2 + 1 == p.Ho && (W2 - 1) / stride2 + 1 == p.Wo, "rmem_conv1x1_dual_nhwc: x2 geometry does not match the output");
  RMEM_REQUIRE(((uintptr_t)x2 % 16) == 0, "rmem_conv1x1_dual_nhwc: x2 must be 16-byte aligned");
  const int nb = d->batch > 0 ? d->batch : 1;
  p.x2 = (const e16*)x2; p.H2 = H2; p.W2 = W2; p.Cin2 = Cin2; p.stride2 = stride2;
  p.x2_elems = (long)nb * H2 * W2 * Cin2;
  p.K = p.Cin + Cin2;                                   // Wcat is [Cout][Cin + Cin2]
  p.steps_per_split = p.K / 64;
  const long lim = (1L << 31) - (1L << 22);
  RMEM_REQUIRE(p.x_elems * 2 < lim && p.x2_elems * 2 < lim && (long)p.Cout * p.K * 2 < lim, "rmem_conv1x1_dual_nhwc: operands must stay below 2 GB");
  hipStream_t s = (hipStream_t)stream;
  static const int big_thr = getenv("RMEM_GEMM_BIG") ? atoi(getenv("RMEM_GEMM_BIG")) : 128;
  static const bool xcd_on = !(getenv("RMEM_GEMM_XCD") && atoi(getenv("RMEM_GEMM_XCD")) == 0);   // kernel experiments only
  if (big_thr > 0 && p.Cout >= 128 && p.K >= 256 && (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) >= big_thr) {
    dim3 grid((p.M + 127) / 128, (p.Cout + 127) / 128, 1);
    if (xcd_on && grid.y > 1 && grid.x >= 16) { p.xcd_ny = (int)grid.y; grid = dim3(8 * ((grid.x + 7) / 8) * grid.y, 1, 1); }
    static const int pc = getenv("RMEM_GEMM_PC_DUAL") ? atoi(getenv("RMEM_GEMM_PC_DUAL")) : 0;   // producer / consumer form, as launch_big
    if (pc >= 3) hipLaunchKernelGGL((k_conv_gemm_dma_pc<true, 3, 3>), grid, dim3(512), 0, s, p);
    else if (pc == 2) hipLaunchKernelGGL((k_conv_gemm_dma_pc<true, 2, 3>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((k_conv_gemm_dma_big<true, 1, 128, 128, 3>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid((p.M + 63) / 64, (p.Cout + 63) / 64, 1);
    if (xcd_on && grid.y > 1 && grid.x >= 16) { p.xcd_ny = (int)grid.y; grid = dim3(8 * ((grid.x + 7) / 8) * grid.y, 1, 1); }
    hipLaunchKernelGGL((k_conv_gemm_dma<true, false, 1, 3>), grid, dim3(256), 0, s, p);
  }
  return rmem_check_launch("rmem_conv1x1_dual_nhwc");
}

extern "C" int RMEM_API(rmem_linear_grouped)(const rmem_conv_desc* d, int n, const void* const* x, const void* const* w,
                                   const float* const* bias, const void* const* residual, void* const* y, void* stream) {
  RMEM_REQUIRE(d && x && w && y && n >= 1 && n <= 4, "rmem_linear_grouped: 1..4 problems");
  ConvParams p;
  bool is1x1 = false;
  GroupPtrs g = {};
  for (int i = 0; i < n; ++i) {
    ConvParams pi;
    if (conv_setup(d, x[i], w[i], bias ? bias[i] : nullptr, residual ? residual[i] : nullptr, y[i], nullptr, pi, is1x1)) return -1;
    RMEM_REQUIRE(is1x1, "rmem_linear_grouped: 1x1 stride-1 problems only");
    RMEM_REQUIRE(i == 0 || pi.vec_ok == p.vec_ok, "rmem_linear_grouped: operands of all problems must have the same alignment class");
    if (i == 0) p = pi;
    g.x[i] = pi.x; g.w[i] = pi.w; g.bias[i] = pi.bias; g.res[i] = pi.res; g.y[i] = pi.y; g.y2[i] = nullptr;
  }
  dim3 grid((p.M + 63) / 64, (p.Cout + 63) / 64, n);
  static const int deep_wgs = getenv("RMEM_GEMM_DEEP_WGS") ? atoi(getenv("RMEM_GEMM_DEEP_WGS")) : 1024;
  if (p.fast_ok == 1 && (long)grid.x * grid.y * n <= deep_wgs && p.steps_per_split >= 3)
    hipLaunchKernelGGL((k_gemm_dma_grouped<true, 3>), grid, dim3(256), 0, (hipStream_t)stream, p, g);
  else if (p.fast_ok == 1) hipLaunchKernelGGL(k_gemm_dma_grouped<true>, grid, dim3(256), 0, (hipStream_t)stream, p, g);
  else hipLaunchKernelGGL(k_gemm_dma_grouped<false>, grid, dim3(256), 0, (hipStream_t)stream, p, g);
  return rmem_check_launch("rmem_linear_grouped");
}
