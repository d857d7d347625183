// Row-block chains of the LSTT block for gfx950 (CDNA4).
//
// Between its three attentions an LSTT block (layers/transformer.py:553-692) is a sequence of ROW-LOCAL operations on the
// [tokens, 256] residual stream: Linear 256 -> 256 (+ residual), LayerNorm, Linear, LayerNorm of a sum, Linear 256 -> 1024 ...
// As separate launches (rmem_conv2d_nhwc, rmem_layernorm256, ...) every one of them is a 5 - 20 us kernel over 6696 rows whose
// arithmetic takes under a microsecond per CU: a group step spent ~54 launches and as many HBM round trips of the activations on
// them.  Here a workgroup owns 32 consecutive tokens of one clip and walks a whole chain with the rows resident in LDS:
//
//   chain A  (after the self attention)   x += att . Wp^T + bp ; curr_V = LN2(x) ; curr_Q = curr_V . Wq^T + bq ;
//                                         k4 = LN4(short_K + curr_Q) ; v4 = LN4(short_V + curr_V)         (transformer.py:571-576, 659-660)
//   chain B  (after the long- and short-term attention)
//                                         x += attL . Wl^T + bl ; tgt3 = attS . Ws^T + bs ; x += tgt3 ; h1 = LN3(x) . W1^T + b1
//                                         (+ the GroupNorm partial sums of h1)                             (transformer.py:635, 662, 681-685)
//   chain C  (after GroupNorm + GELU + depth-wise 5x5)
//                                         x += h3 . W2^T + b2 ; dec_in[:, slice] = LN_dec(x) ; and, unless this is the last
//                                         block, qkv' = LN1'(x) . Wqkv'^T + bqkv' + pos_qk' of the NEXT block      (transformer.py:685-687, 250-259, 565-569)
//
// Arithmetic is the unfused route's, operation for operation: the same v_mfma_f32_16x16x32 accumulation order over k, the same
// epilogue order (accumulator + bias, then residual), the same LayerNorm reduction (rmem_ln256_row in common.h), the same
// points of rounding to e16 -- the fused and unfused LSTT give bit-identical x / curr_Q / k4 / v4 / h1 / qkv
// (tests/test_hip_ops.py::test_lstt_chains_are_bit_identical); only the GroupNorm partial sums are added in another order.
//
// GEMM form.  M = 32 rows per workgroup is far too few to amortise a weight tile through LDS per workgroup; instead the A
// operand (the 32 rows, <= 1024 deep) lives in LDS for the whole chain and the WEIGHTS stream straight from L2 into registers,
// each wave taking its own 64 output columns: they are packed once at model load in fragment order (pack.py::pack_frag:
// [N / 256][wave][K / 32][column tiles][64 lanes][8]), so a wave-instruction reads 1 KiB of contiguous memory that is exactly
// one B fragment, no LDS, no barrier, no address arithmetic; a ring of 4 k-chunks (16 KiB per workgroup) stays in flight.
// All workgroups read the same weights (128 KiB - 512 KiB per matrix): they are L2 hits after the first workgroup of an XCD.
#include "common.h"
#include "../../include/rmem.h"

namespace {

#ifndef RMEM_CHAIN_WAVES
#define RMEM_CHAIN_WAVES 8
#endif
constexpr int NW = RMEM_CHAIN_WAVES;   // waves per workgroup (4 or 8): each takes 256 / NW of the 256 output columns of a block
constexpr int NT = 64 * NW;       // threads
constexpr int JT = 16 / NW;       // 16-column MFMA tiles per wave and column block (4 or 2)
constexpr int WC = 256 / NW;      // columns per wave and column block
constexpr int RW = 32 / NW;       // rows per wave in the row phases (8 or 4)
constexpr int BM = 32;            // rows per workgroup
constexpr int XS = 260;           // fp32 staging row stride (floats): 4 rows apart = 16 banks apart, conflict-free ds_write_b32
constexpr int PANEL = BM * 64;    // elements of one [32 rows][64 k] A panel (128-byte rows, XOR-swizzled 16-byte chunks)
constexpr int PF = 16 / JT;       // k-chunks (of 32) of B fragments in flight per wave (64 registers; PF <= 8 = the k-chunks of the shortest K)
constexpr int OOB = (int)0x80000000;   // byte offset beyond every buffer: the hardware range check drops the store

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ e16x4 cvt4(f32x4 v) { return e16x4{(e16)v[0], (e16)v[1], (e16)v[2], (e16)v[3]}; }
__device__ __forceinline__ f32x4 up4(e16x4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

// Stores go through buffer descriptors: a row beyond the block's last one passes the offset OOB and the hardware drops the
// store -- NO branch in the row loops.  (With `if (row < nrows) store` the compiler branches around every row, loses track of
// the loads in flight at each join and drains them all -- s_waitcnt vmcnt(0), previous row's store included -- once per row.)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st8(rsrc_t r, int off, e16x4 v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off, 0, 0); }
__device__ __forceinline__ void st16(rsrc_t r, int off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0); }
#else
struct rsrc_t {};
__device__ inline rsrc_t make_rsrc(const void*, long) { return {}; }
__device__ inline void st8(rsrc_t, int, e16x4) {}
__device__ inline void st16(rsrc_t, int, f32x4) {}
#endif

// [nrows <= 32][64 * NP] e16 rows of stride ld at src -> A panels p0 .. p0 + NP - 1
template <int NP>
__device__ __forceinline__ void stage_tile(e16* A16, int p0, const e16* src, long ld, int nrows) {
  // 256 threads cover one [32][64] panel (row = t >> 3, 16-byte chunk = t & 7); with 512 threads the two halves take alternate panels
  constexpr int PT = NT / 256;                      // panels per pass
  static_assert(NP % PT == 0, "panels must divide among the thread halves");
  const int t = threadIdx.x & 255, half = threadIdx.x >> 8;
  const int row = t >> 3, ch = t & 7;
  // rows beyond the block's last one re-read that last row (their results are never stored): an unconditional load
  const e16* sr = src + min(row, nrows - 1) * ld + ch * 8;
  e16x8 v[NP / PT];
#pragma unroll
  for (int p = 0; p < NP / PT; ++p) v[p] = *reinterpret_cast<const e16x8*>(sr + (p * PT + half) * 64);
#pragma unroll
  for (int p = 0; p < NP / PT; ++p) *reinterpret_cast<e16x8*>(&A16[(p0 + p * PT + half) * PANEL + swz(row, ch)]) = v[p];
}

// 4 consecutive channels c0 .. c0 + 3 of row `row` as e16 into the A panels p0 .. (the LayerNorm output that feeds the next GEMM)
__device__ __forceinline__ void put_a(e16* A16, int p0, int row, int lane, e16x4 v) {
  *reinterpret_cast<e16x4*>(&A16[(p0 + (lane >> 4)) * PANEL + swz(row, (lane & 15) >> 1) + (lane & 1) * 4]) = v;
}

struct BRing { e16x8 b[PF][JT]; };

// this wave's weight stream of column block nb: [K / 32][JT][64 lanes][8]
__device__ __forceinline__ const e16* wstream(const e16* w, int nb, int wave, int KC) { return w + (long)(nb * NW + wave) * KC * (JT * 512); }

// (sched_barrier: the machine scheduler otherwise SINKS these loads down to their first use to save registers, which turns the
// ring into load -> wait -> MFMA, one L2 round trip per k-chunk)
__device__ __forceinline__ void b_preload(BRing& r, const e16* wp, int lane) {
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int j = 0; j < JT; ++j) r.b[u][j] = *reinterpret_cast<const e16x8*>(wp + ((u * JT + j) * 64 + lane) * 8);
  __builtin_amdgcn_sched_barrier(0);
}

// acc[2][4] (32 rows x this wave's 64 columns) += A[32][64 * NP] (panels p0 ..) . W^T, k ascending in chunks of 32: the
// accumulation order of the 64x64-tile GEMM kernel (two MFMAs per 64-deep k-step).  The ring holds chunks 0 .. PF - 1 on entry.
template <int NP>
__device__ __forceinline__ void gemm32(const e16* A16, int p0, const e16* wp, BRing& r, f32x4 (&acc)[2][JT], int lane) {
  constexpr int KC = NP * 2;
  const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const e16* Ap = A16 + (p0 + (kc >> 1)) * PANEL;
    const e16x8 a0 = *reinterpret_cast<const e16x8*>(&Ap[swz(fr, 4 * (kc & 1) + fc)]);
    const e16x8 a1 = *reinterpret_cast<const e16x8*>(&Ap[swz(16 + fr, 4 * (kc & 1) + fc)]);
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      acc[0][j] = RMEM_MFMA_16x16x32(a0, r.b[kc % PF][j], acc[0][j], 0, 0, 0);
      acc[1][j] = RMEM_MFMA_16x16x32(a1, r.b[kc % PF][j], acc[1][j], 0, 0, 0);
    }
    if (kc + PF < KC) {
#pragma unroll
      for (int j = 0; j < JT; ++j) r.b[kc % PF][j] = *reinterpret_cast<const e16x8*>(wp + (((kc + PF) * JT + j) * 64 + lane) * 8);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[2][JT]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// raw accumulators -> fp32 staging tile [32][XS] (C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg)
__device__ __forceinline__ void dump_acc(float* X, const f32x4 (&acc)[2][JT], int wave, int lane) {
  const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) X[(i * 16 + fc * 4 + q) * XS + wave * WC + j * 16 + fr] = acc[i][j][q];
}

struct Blk { int nrows; long g0; };      // rows this workgroup owns: g0 .. g0 + nrows - 1 of the [clips * L] row space
__device__ __forceinline__ Blk my_rows(int L) {
  Blk b;
  const int r0 = blockIdx.x * BM;
  b.nrows = min(BM, L - r0);
  b.g0 = (long)blockIdx.y * L + r0;
  return b;
}

// Row phases.  A wave owns RW = 32 / NW consecutive rows of the block (wave * RW ..), a lane 4 consecutive channels of each.  Everything a phase
// reads from global memory is loaded into registers at the START of the kernel (x rows, short-term memory rows, biases, norm
// parameters) or before the GEMM whose epilogue needs it: a load issued inside the phase costs one full memory latency PER ROW
// (the phase is a dependent chain of cross-lane reductions; the first version of this file spent 12 us per phase that way).
template <typename T>
struct Rows8 { T r[RW]; };      // this wave's rows

__device__ __forceinline__ Rows8<f32x4> load_rows_f32(const float* base, int ld, const Blk& blk, int wave, int c0) {
  Rows8<f32x4> o;
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = min(wave * RW + rr, blk.nrows - 1);      // (rows beyond the block: the last row again, unconditionally; never stored)
    o.r[rr] = ld4(base + (blk.g0 + row) * ld + c0);
  }
  return o;
}
__device__ __forceinline__ Rows8<e16x4> load_rows_e16(const e16* base, int ld, const Blk& blk, int wave, int c0) {
  Rows8<e16x4> o;
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = min(wave * RW + rr, blk.nrows - 1);
    o.r[rr] = *reinterpret_cast<const e16x4*>(base + (blk.g0 + row) * ld + c0);
  }
  return o;
}
// byte offset of (row rr of this wave, channel c0) in a [rows][ld] array of `esz`-byte elements, or OOB beyond the block
__device__ __forceinline__ int row_off(const Blk& blk, int wave, int rr, int ld, int c0, int esz) {
  const int row = wave * RW + rr;
  return row < blk.nrows ? (int)(((blk.g0 + row) * ld + c0) * esz) : OOB;
}

// ------------------------------------------------------------------------------------------------------------ chain A
__global__ __launch_bounds__(NT) void k_chain_a(rmem_chain_a_desc d) {
  __shared__ __attribute__((aligned(16))) char smem[4 * PANEL * 2 + BM * XS * 4];
  e16* A16 = reinterpret_cast<e16*>(smem);
  float* X = reinterpret_cast<float*>(smem + 4 * PANEL * 2);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Blk blk = my_rows(d.L);
  const int c0 = lane * 4;
  const long rows = (long)d.clips * d.L;
  const rsrc_t rx = make_rsrc(d.x, rows * 1024), rcv = make_rsrc(d.curr_v, rows * 512), rcq = make_rsrc(d.curr_q, rows * 512),
               rk4 = make_rsrc(d.k4, rows * 512), rv4 = make_rsrc(d.v4, rows * 512);
  const f32x4 bp = ld4(d.b_proj + c0), g2 = ld4(d.ln2_g + c0), b2 = ld4(d.ln2_b + c0);
  const f32x4 bq = ld4(d.b_q + c0), g4 = ld4(d.ln4_g + c0), b4 = ld4(d.ln4_b + c0);
  const Rows8<f32x4> xo = load_rows_f32(d.x, 256, blk, wave, c0);
  const Rows8<e16x4> sk = load_rows_e16((const e16*)d.short_k, 256, blk, wave, c0);
  const Rows8<e16x4> sv = load_rows_e16((const e16*)d.short_v, 256, blk, wave, c0);
  BRing ring;
  f32x4 acc[2][JT];
  const e16* wp = wstream((const e16*)d.w_proj, 0, wave, 8);
  stage_tile<4>(A16, 0, (const e16*)d.att + blk.g0 * 256, 256, blk.nrows);
  b_preload(ring, wp, lane);                                     // weights do not depend on anything: in flight before the rows arrive
  __syncthreads();
  zero_acc(acc);
  gemm32<4>(A16, 0, wp, ring, acc, lane);
  dump_acc(X, acc, wave, lane);
  const e16* wq = wstream((const e16*)d.w_q, 0, wave, 8);
  b_preload(ring, wq, lane);
  __syncthreads();
  // x += self_proj(att) ; curr_V = LN2(x)
  e16x4 cv[RW];
  {
    f32x4 v[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      v[rr] = ld4(&X[(wave * RW + rr) * XS + c0]);
      v[rr] += bp;
      v[rr] += xo.r[rr];
      st16(rx, row_off(blk, wave, rr, 256, c0, 4), v[rr]);
    }
    rmem_ln256_rows<RW>(v, g2, b2, d.eps);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      cv[rr] = cvt4(v[rr]);
      st8(rcv, row_off(blk, wave, rr, 256, c0, 2), cv[rr]);
      put_a(A16, 0, wave * RW + rr, lane, cv[rr]);
    }
  }
  __syncthreads();
  zero_acc(acc);
  gemm32<4>(A16, 0, wq, ring, acc, lane);
  dump_acc(X, acc, wave, lane);
  __syncthreads();
  // curr_Q ; k4 = LN4(short_K + curr_Q) ; v4 = LN4(short_V + curr_V)
  {
    f32x4 kk[RW], vv[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      f32x4 q = ld4(&X[(wave * RW + rr) * XS + c0]);
      q += bq;
      const e16x4 cq = cvt4(q);
      st8(rcq, row_off(blk, wave, rr, 256, c0, 2), cq);
      kk[rr] = up4(sk.r[rr]);
      kk[rr] += up4(cq);
      vv[rr] = up4(sv.r[rr]);
      vv[rr] += up4(cv[rr]);
    }
    rmem_ln256_rows<RW>(kk, g4, b4, d.eps);
    rmem_ln256_rows<RW>(vv, g4, b4, d.eps);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int off = row_off(blk, wave, rr, 256, c0, 2);
      st8(rk4, off, cvt4(kk[rr]));
      st8(rv4, off, cvt4(vv[rr]));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ chain B
// STATS: also write the GroupNorm partial sums of h1 (d.gn_partial)
template <bool STATS>
__global__ __launch_bounds__(NT) void k_chain_b(rmem_chain_b_desc d) {
  __shared__ __attribute__((aligned(16))) char smem[8 * PANEL * 2 + 2 * BM * XS * 4 + 4 * NW * 8 * 2 * 4];
  e16* A16 = reinterpret_cast<e16*>(smem);
  float* Xa = reinterpret_cast<float*>(smem + 8 * PANEL * 2);
  float* Xb = Xa + BM * XS;
  float* gsum = Xb + BM * XS;               // [column block][wave][group of the block][2]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Blk blk = my_rows(d.L);
  const int c0 = lane * 4;
  const long rows = (long)d.clips * d.L;
  const rsrc_t rx = make_rsrc(d.x, rows * 1024), rt3 = make_rsrc(d.tgt3, rows * 512), rh1 = make_rsrc(d.h1, rows * 2048);
  const f32x4 bl = ld4(d.b_long + c0), bs = ld4(d.b_short + c0), g3 = ld4(d.ln3_g + c0), b3 = ld4(d.ln3_b + c0);
  f32x4 b1[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) b1[nb] = ld4(d.b1 + nb * 256 + c0);
  const Rows8<f32x4> xo = load_rows_f32(d.x, 256, blk, wave, c0);
  BRing ring;
  f32x4 acc[2][JT];
  const e16* wl = wstream((const e16*)d.w_long, 0, wave, 8);
  stage_tile<4>(A16, 0, (const e16*)d.att_long + blk.g0 * 256, 256, blk.nrows);
  stage_tile<4>(A16, 4, (const e16*)d.att_short + blk.g0 * 256, 256, blk.nrows);
  b_preload(ring, wl, lane);
  __syncthreads();
  zero_acc(acc);
  gemm32<4>(A16, 0, wl, ring, acc, lane);
  dump_acc(Xa, acc, wave, lane);
  const e16* ws = wstream((const e16*)d.w_short, 0, wave, 8);
  b_preload(ring, ws, lane);
  zero_acc(acc);
  gemm32<4>(A16, 4, ws, ring, acc, lane);
  dump_acc(Xb, acc, wave, lane);
  const e16* w1 = wstream((const e16*)d.w1, 0, wave, 8);
  b_preload(ring, w1, lane);
  __syncthreads();
  // x += long_proj(attL) ; tgt3 = short_proj(attS) ; x += tgt3 ; LN3(x) -> A panels 0..3
  {
    f32x4 t[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      f32x4 v1 = ld4(&Xa[(wave * RW + rr) * XS + c0]);
      v1 += bl;
      t[rr] = ld4(&Xb[(wave * RW + rr) * XS + c0]);
      t[rr] += bs;
      v1 += xo.r[rr];                                   // x after the long-term projection (what the unfused route stores)
      st8(rt3, row_off(blk, wave, rr, 256, c0, 2), cvt4(t[rr]));
      t[rr] += v1;
      st16(rx, row_off(blk, wave, rr, 256, c0, 4), t[rr]);
    }
    rmem_ln256_rows<RW>(t, g3, b3, d.eps);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) put_a(A16, 0, wave * RW + rr, lane, cvt4(t[rr]));
  }
  __syncthreads();
  // h1 = linear1(LN3(x)): four column blocks of 256, staged alternately through Xa / Xb (one barrier per block)
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    float* X = (nb & 1) ? Xb : Xa;
    zero_acc(acc);
    gemm32<4>(A16, 0, w1, ring, acc, lane);
    dump_acc(X, acc, wave, lane);
    if (nb < 3) { w1 = wstream((const e16*)d.w1, nb + 1, wave, 8); b_preload(ring, w1, lane); }
    __syncthreads();
    float gs = 0.f, gss = 0.f;          // this lane's channels belong to ONE GroupNorm group (32 channels = 8 lanes) per block
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      f32x4 v = ld4(&X[(wave * RW + rr) * XS + c0]);
      v += b1[nb];
      const e16x4 h = cvt4(v);
      st8(rh1, row_off(blk, wave, rr, 1024, nb * 256 + c0, 2), h);
      if (STATS) {
        const f32x4 hf = up4(h);
        const float okf = wave * RW + rr < blk.nrows ? 1.f : 0.f;      // (a select, not a branch)
        gs += okf * rmem_sum4(hf);
        gss += okf * rmem_sumsq4(hf);
      }
    }
    if (STATS) {
      // (sum, sum of squares) of this block's rows per group: 8 lanes -> wave; the waves are added after the last block
      gs += __shfl_xor(gs, 1); gss += __shfl_xor(gss, 1);
      gs += __shfl_xor(gs, 2); gss += __shfl_xor(gss, 2);
      gs += __shfl_xor(gs, 4); gss += __shfl_xor(gss, 4);
      if ((lane & 7) == 0) { gsum[((nb * NW + wave) * 8 + (lane >> 3)) * 2] = gs; gsum[((nb * NW + wave) * 8 + (lane >> 3)) * 2 + 1] = gss; }
    }
  }
  if (STATS) {
    __syncthreads();
    if (threadIdx.x < 32) {
      // workspace [clip][group 0..31][split][2] with d.gn_splits splits per group (this workgroup is split blockIdx.x)
      const int nb = threadIdx.x >> 3, g = threadIdx.x & 7;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a += gsum[((nb * NW + w) * 8 + g) * 2]; b += gsum[((nb * NW + w) * 8 + g) * 2 + 1]; }
      float* o = d.gn_partial + (((long)blockIdx.y * 32 + threadIdx.x) * d.gn_splits + blockIdx.x) * 2;
      o[0] = a; o[1] = b;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ chain C
// FFN2: x += linear2(h3) first (else the rows are read from x as they are: the first block of the stack);
// NEXT: LN1' + fused QKV projection of the next block
template <bool FFN2, bool NEXT>
__global__ __launch_bounds__(NT) void k_chain_c(rmem_chain_c_desc d) {
  constexpr int NPA = FFN2 ? 16 : 4;
  // with FFN2 the second staging buffer lives in A panels 4 .. 15: they are dead once linear2's GEMM is done (the barrier behind its
  // dump), and only panels 0 .. 3 are written again (LN1' of the next block) -- 97 KB instead of 130 KB, so the workgroup finds a CU sooner
  static_assert(!FFN2 || 12 * PANEL * 2 >= BM * XS * 4, "the second staging buffer must fit panels 4 .. 15");
  __shared__ __attribute__((aligned(16))) char smem[NPA * PANEL * 2 + (FFN2 ? 1 : 2) * BM * XS * 4];
  e16* A16 = reinterpret_cast<e16*>(smem);
  float* Xa = reinterpret_cast<float*>(smem + NPA * PANEL * 2);
  float* Xb = FFN2 ? reinterpret_cast<float*>(smem + 4 * PANEL * 2) : Xa + BM * XS;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Blk blk = my_rows(d.L);
  const int c0 = lane * 4;
  const long rows = (long)d.clips * d.L;
  const rsrc_t rx = make_rsrc(d.x, rows * 1024), rdec = make_rsrc(d.dec_out, FFN2 ? ((rows - 1) * d.ld_dec + 256) * 2 : 0),
               rqkv = make_rsrc(d.qkv, NEXT ? rows * 1536 : 0);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 b2 = FFN2 ? ld4(d.b2 + c0) : z4, gd = FFN2 ? ld4(d.dec_g + c0) : z4, bd = FFN2 ? ld4(d.dec_b + c0) : z4;
  const f32x4 g1 = NEXT ? ld4(d.ln1_g + c0) : z4, b1 = NEXT ? ld4(d.ln1_b + c0) : z4;
  f32x4 bq[3] = {z4, z4, z4};
  if (NEXT) {
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) bq[nb] = ld4(d.b_qkv + nb * 256 + c0);
  }
  const Rows8<f32x4> xo = load_rows_f32(d.x, 256, blk, wave, c0);
  BRing ring;
  f32x4 acc[2][JT];
  const e16* wq = NEXT ? wstream((const e16*)d.w_qkv, 0, wave, 8) : nullptr;
  if (FFN2) {
    const e16* w2 = wstream((const e16*)d.w2, 0, wave, 32);
    stage_tile<16>(A16, 0, (const e16*)d.h3 + blk.g0 * 1024, 1024, blk.nrows);
    b_preload(ring, w2, lane);
    __syncthreads();
    zero_acc(acc);
    gemm32<16>(A16, 0, w2, ring, acc, lane);
    dump_acc(Xa, acc, wave, lane);
    if (NEXT) b_preload(ring, wq, lane);
    __syncthreads();
  } else if (NEXT) {
    b_preload(ring, wq, lane);
  }
  {
    f32x4 v[RW];
    if (FFN2) {
      f32x4 vd[RW];
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        v[rr] = ld4(&Xa[(wave * RW + rr) * XS + c0]);
        v[rr] += b2;
        v[rr] += xo.r[rr];
        st16(rx, row_off(blk, wave, rr, 256, c0, 4), v[rr]);
        vd[rr] = v[rr];
      }
      rmem_ln256_rows<RW>(vd, gd, bd, d.eps);
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) st8(rdec, row_off(blk, wave, rr, d.ld_dec, c0, 2), cvt4(vd[rr]));
    } else {
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) v[rr] = xo.r[rr];
    }
    if (NEXT) {
      rmem_ln256_rows<RW>(v, g1, b1, d.eps);
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) put_a(A16, 0, wave * RW + rr, lane, cvt4(v[rr]));
    }
  }
  if (!NEXT) return;
  __syncthreads();
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    float* X = (nb & 1) ? Xb : Xa;
    const Rows8<f32x4> pos = load_rows_f32(d.pos_qk + nb * 256, 768, blk, wave, c0);     // in flight under the GEMM
    zero_acc(acc);
    gemm32<4>(A16, 0, wq, ring, acc, lane);
    dump_acc(X, acc, wave, lane);
    if (nb < 2) { wq = wstream((const e16*)d.w_qkv, nb + 1, wave, 8); b_preload(ring, wq, lane); }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      f32x4 v = ld4(&X[(wave * RW + rr) * XS + c0]);
      v += bq[nb];
      v += pos.r[rr];
      st8(rqkv, row_off(blk, wave, rr, 768, nb * 256 + c0, 2), cvt4(v));
    }
  }
}

#ifndef RMEM_F16
}  // namespace
extern "C" int rmem_lstt_chain_waves(void) { return NW; }
namespace {
#endif
bool al16(const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; }

}  // namespace

extern "C" int RMEM_API(rmem_lstt_chain_a)(const rmem_chain_a_desc* d, void* stream) {
  RMEM_REQUIRE(d && d->L > 0 && d->clips >= 1 && (long)d->L * d->clips < (1L << 20), "rmem_lstt_chain_a: bad geometry (rows must stay below 2^20)");
  RMEM_REQUIRE(d->att && d->x && d->w_proj && d->b_proj && d->ln2_g && d->ln2_b && d->curr_v && d->w_q && d->b_q && d->curr_q && d->short_k &&
               d->short_v && d->ln4_g && d->ln4_b && d->k4 && d->v4, "rmem_lstt_chain_a: null argument");
  RMEM_REQUIRE(al16(d->att) && al16(d->x) && al16(d->w_proj) && al16(d->w_q) && al16(d->curr_v) && al16(d->curr_q) && al16(d->short_k) &&
               al16(d->short_v) && al16(d->k4) && al16(d->v4) && al16(d->b_proj) && al16(d->b_q) && al16(d->ln2_g) && al16(d->ln2_b) &&
               al16(d->ln4_g) && al16(d->ln4_b), "rmem_lstt_chain_a: operands must be 16-byte aligned");
  hipLaunchKernelGGL(k_chain_a, dim3((d->L + BM - 1) / BM, d->clips), dim3(NT), 0, (hipStream_t)stream, *d);
  return rmem_check_launch("rmem_lstt_chain_a");
}

extern "C" int RMEM_API(rmem_lstt_chain_b)(const rmem_chain_b_desc* d, void* stream) {
  RMEM_REQUIRE(d && d->L > 0 && d->clips >= 1 && (long)d->L * d->clips < (1L << 20), "rmem_lstt_chain_b: bad geometry (rows must stay below 2^20)");
  RMEM_REQUIRE(d->att_long && d->att_short && d->x && d->w_long && d->b_long && d->w_short && d->b_short && d->tgt3 && d->ln3_g && d->ln3_b &&
               d->w1 && d->b1 && d->h1, "rmem_lstt_chain_b: null argument");
  RMEM_REQUIRE(al16(d->att_long) && al16(d->att_short) && al16(d->x) && al16(d->w_long) && al16(d->w_short) && al16(d->tgt3) && al16(d->w1) &&
               al16(d->h1) && al16(d->b_long) && al16(d->b_short) && al16(d->b1) && al16(d->ln3_g) && al16(d->ln3_b),
               "rmem_lstt_chain_b: operands must be 16-byte aligned");
  RMEM_REQUIRE(!d->gn_partial || d->gn_splits >= (d->L + BM - 1) / BM, "rmem_lstt_chain_b: gn_splits must cover the row blocks of a clip");
  if (d->gn_partial) hipLaunchKernelGGL(k_chain_b<true>, dim3((d->L + BM - 1) / BM, d->clips), dim3(NT), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(k_chain_b<false>, dim3((d->L + BM - 1) / BM, d->clips), dim3(NT), 0, (hipStream_t)stream, *d);
  return rmem_check_launch("rmem_lstt_chain_b");
}

extern "C" int RMEM_API(rmem_lstt_chain_c)(const rmem_chain_c_desc* d, void* stream) {
  RMEM_REQUIRE(d && d->L > 0 && d->clips >= 1 && (long)d->L * d->clips < (1L << 20) && d->x, "rmem_lstt_chain_c: bad geometry (rows must stay below 2^20)");
  const bool ffn2 = d->h3 != nullptr, next = d->w_qkv != nullptr;
  RMEM_REQUIRE(ffn2 || next, "rmem_lstt_chain_c: nothing to do (neither h3 nor w_qkv)");
  RMEM_REQUIRE(!ffn2 || (d->w2 && d->b2 && d->dec_g && d->dec_b && d->dec_out && d->ld_dec >= 256 && d->ld_dec % 8 == 0),
               "rmem_lstt_chain_c: the linear2 stage needs w2, b2, the decoder norm and dec_out (ld_dec >= 256, multiple of 8)");
  RMEM_REQUIRE(!next || (d->ln1_g && d->ln1_b && d->b_qkv && d->pos_qk && d->qkv), "rmem_lstt_chain_c: the QKV stage needs ln1, b_qkv, pos_qk, qkv");
  RMEM_REQUIRE(al16(d->x) && al16(d->h3) && al16(d->w2) && al16(d->b2) && al16(d->dec_g) && al16(d->dec_b) && al16(d->dec_out) && al16(d->ln1_g) &&
               al16(d->ln1_b) && al16(d->w_qkv) && al16(d->b_qkv) && al16(d->pos_qk) && al16(d->qkv), "rmem_lstt_chain_c: operands must be 16-byte aligned");
  const dim3 grid((d->L + BM - 1) / BM, d->clips);
  hipStream_t s = (hipStream_t)stream;
  if (ffn2 && next) hipLaunchKernelGGL((k_chain_c<true, true>), grid, dim3(NT), 0, s, *d);
  else if (ffn2) hipLaunchKernelGGL((k_chain_c<true, false>), grid, dim3(NT), 0, s, *d);
  else hipLaunchKernelGGL((k_chain_c<false, true>), grid, dim3(NT), 0, s, *d);
  return rmem_check_launch("rmem_lstt_chain_c");
}
