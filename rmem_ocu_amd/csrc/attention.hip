// Space-time memory-read attention for gfx950 (CDNA4), head dim 32.
//
// One kernel serves the three attention call sites of an LSTT block
// (layers/transformer.py:569, 632-635, 657-662 through layers/attention.py:45-74):
//   long-term : Q = curr_Q + cur_pe, keys = the restricted memory bank (T frames x HW
//               tokens, K[t] + mem_pe[slot(t)]), optional per-memory-frame probability
//               mass side output (transformer.py:636-643);
//   short-term and self attention : one key frame, no temporal embedding -- as launches of their own (MEM = false) or, for the
//               short-term attention, as extra grid slices of the memory-read launch (rmem_lstt_attn_pair_clips: same queries,
//               independent keys, second output).
//
// Decomposition: grid = (query tiles of 128) x heads x key GROUPS x clips (+ the second attention's tiles x heads x clips).  The
// keys are described by a device-resident table of ROWS (a row = a contiguous key range of ONE memory frame: slot, range,
// temporal-PE slot), so a captured hipGraph stays valid while the bank's slot table changes.  A workgroup walks a GROUP of
// consecutive rows (several memory frames when enough workgroups exist without splitting further) and keeps ONE running
// (m_ref, l, O) for all of them; at every row end it records that row's (m_ref, l_row) pair, from which k_attn_mass derives the
// per-memory-frame probability mass (transformer.py:636-643) without ever materialising the attention matrix.  With one group
// the normalised e16 output is written by this kernel itself; with several, each group leaves a partial normalised by its own
// sum (stored as e16) + (m, l) in fp32, and k_attn_combine merges them (a convex combination).
//
// Per workgroup: 4 waves x 32 query rows.  K/V tiles of 64 keys go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no
// VGPR staging, no ds_write) into a ring of NB = 4 tiles, three tiles in flight, the DMA cursor running across row boundaries
// ahead of the arithmetic; one counted s_waitcnt vmcnt + one bare s_barrier per tile.  The buffer descriptor is rebuilt per tile
// so that it ends with the row: lanes beyond the row's last key read zeros from the hardware range check (the ragged tail costs
// no per-lane test).  K stays row-major [key][32] with an XOR chunk swizzle applied to the DMA's SOURCE address (conflict-free
// ds_read_b128), V row-major and read TRANSPOSED (ds_read_b64_tr_b16); the loop's LDS reads are inline asm with counted
// lgkmcnt waits (the compiler would drain the DMA ring in front of every read).
//   S^T = K . Q^T   : v_mfma_f32_32x32x16, A = K rows (keys), B = Q^T held in registers for the whole kernel; the query sits
//                     on the lane, so row max / row sum are in-lane plus one exchange with lane ^ 32.  The reference m_ref and
//                     the temporal-PE term q'.pe[slot] are the accumulator's INITIAL value (a persistent C-operand block).
//   O^T += V^T . P^T: the S^T accumulator (exp2'd, packed to e16) IS the B operand: registers 8s..8s+7 of lane half h are
//                     keys 16s+8(j>>2)+4h+(j&3), and the V^T A-fragment is read in that same key order.  P never touches LDS.
//   row sums        : v_dot2c_f32 (packed pair . (1, 1) + acc) on the SAME rounded probabilities the P.V MFMAs consume.
// Logits live in the log2 domain: Q is pre-scaled by log2(e)/sqrt(32).
//
// Softmax reference.  With d = 32 every score costs one v_exp_f32 against 128 MFMA FLOPs, so the loop carries nothing else per
// score.  The FAST pass fixes m_ref at the maximum of the group's FIRST tile and never computes a maximum again: floating point
// keeps the relative precision of P = 2^(S - m_ref), O and l whatever the offset, so a later, larger score is harmless until
// 2^(S - m_ref) leaves the fp32 range.  That case is detected after the walk (l above 2^64, inf or NaN for any query of the
// workgroup) and the whole group is redone by the SAFE pass, the classic online softmax (tile maximum, m_ref moved and O / l
// rescaled whenever a tile exceeds it by 2^8).  Both give the same result to fp32 rounding; tests force the fallback with a key
// 2^100 above the first tile.  The padded keys of a row's ragged last tile get the score -1e30 in THAT tile only (behind a real
// branch): they score q.0 + (bias - m_ref), which can lie far ABOVE every real score.  The IEEE-half build always takes SAFE.
#include "common.h"
#include "../../include/rmem.h"
#include <type_traits>

namespace {

constexpr int D = 32;          // head dim
constexpr int KT = 64;         // keys per LDS tile
constexpr int NB = 4;          // LDS ring depth: NB - 1 tiles in flight by LDS-DMA (one tile ahead does not cover the
                               // L2 -> LDS latency when every CU streams: the kernel then waits on vmcnt at every tile)
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 8.0f;          // SAFE pass, log2 units: P <= 2^8 between rescales
[[maybe_unused]] constexpr float L_LIMIT = 1.8446744e19f;     // 2^64: a fast-pass row sum above it (or inf / NaN) sends the group to the SAFE pass

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct AttnParams {
  const e16* q; int ldq;
  const e16* k; const e16* v; long slot_stride; int ldkv;
  const rmem_attn_chunk* chunks; int nchunks; int lk; int per_chunk;
  const float* pe_cur; const float* pe_mem;
  int Lq, heads, C, nq;
  int ngroups, rpg;               // key groups per (query tile, head); table rows per group
  float* opart; float* mlg;       // per group: partial O normalised by its own l, stored as e16 (the workspace is sized and
                                  // strided in floats; half of each block is used), and (m_ref, l) in fp32     (ngroups > 1)
  float* ml;                      // per table row: (m_ref at the row's end, l_row)     (mass output wanted), or null
  float qscale;
  e16* out; int ldo;             // normalised e16 output (written here when ngroups == 1)
  // several clips in one launch (identical shapes; clip c's operands sit c * stride further, its table rows at c * nchunks)
  int nclips; long q_cs, kv_cs, out_cs, opart_cs, mlg_cs, ml_cs;
  // a SECOND, independent attention of the same queries in the same launch (MEM flavour only): one plain key frame per clip
  // (the short-term attention of an LSTT block, layers/transformer.py:656-662: same curr_Q, keys / values = the norm4 outputs),
  // no temporal embedding, its own output.  n2 = clips of it (0: none); its workgroups are extra grid slices behind the
  // memory read's, so the launch's tail is filled with them instead of a second launch starting from an empty GPU.
  int n2; const e16* k2; const e16* v2; int lk2; long kv2_cs; e16* out2; long out2_cs;
};

struct Row { int slot, kb, kn, pe_slot; };

__device__ __forceinline__ int kswz(int row, int chunk) { return row * D + ((chunk ^ ((row >> 2) & 3)) << 3); }

// max over the two lanes that share a query (lane, lane ^ 32) without touching LDS
__device__ __forceinline__ float pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// buffer-descriptor LDS-DMA (buffer_load_dwordx4 ... offen lds): global -> LDS without VGPR staging; a lane whose offset
// lies beyond the descriptor's byte count gets zeros from the hardware range check (the ragged tail of a row needs no
// per-lane test).  The descriptor type and the builtins exist only in the device pass; the host pass needs the names.
typedef __attribute__((address_space(3))) void* lptr_t;
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

// LDS reads of the tile loop as inline asm with counted waits: the compiler cannot tell that the LDS-DMA in flight (next tile,
// OTHER buffer) does not alias the tile being read and would drain it (s_waitcnt vmcnt(0)) in front of the first ds_read of
// every tile, which serialises the prefetch with the arithmetic.  LDS operations return in order, so lgkmcnt(N) = "all but
// the N youngest have landed"; the fragments pass through the wait so that their consumers stay behind it.
#ifdef RMEM_ATTN_C_LDS
template <int OFF>
__device__ __forceinline__ e16x8 lds_b128(unsigned addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) e16x8*>((size_t)(addr + OFF));
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)(addr + OFF));
}
#else
template <int OFF>
__device__ __forceinline__ e16x8 lds_b128(unsigned addr) {
  e16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16(unsigned addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
#endif
template <int N>
__device__ __forceinline__ void lds_wait4(e16x8& a, e16x8& b, e16x8& c, e16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
struct PvFrag { s16x4 lo[2], hi[2]; };      // one 32-key block: the V^T fragments of its 2 slabs
template <int N>
__device__ __forceinline__ void lds_wait_pv(PvFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.lo[0]), "+v"(f.hi[0]), "+v"(f.lo[1]), "+v"(f.hi[1]) : "n"(N));
}

#ifdef RMEM_ATTN_TIMELINE
// Measurement build only (scripts/build_attn_variants.sh timeline=-DRMEM_ATTN_TIMELINE): wave 0 of every workgroup reads the
// shader clock at four points of every tile and sums the phases; rmem_attn_timeline_read() returns the sums.
//   A: barrier left -> K fragments in registers      B: -> first exponential issued (S^T out of the matrix pipe)
//   C: -> last row-sum instruction issued            D: -> next barrier left (scalar work, DMA issue, vmcnt wait, barrier wait)
__device__ unsigned g_attn_timeline[8192 * 8];
#define RMEM_TL_STAMP(var, tie) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var), "+v"(tie))
#endif

// MEM = true: memory-read flavour (row table, temporal PE); false: one plain key frame cut into nchunks ranges.  Two
// symbols so a kernel trace separates the long-term memory read from the short-term / self attention launches.
// TIMED changes nothing but the symbol: launches bracketed by rmem_profile_* HIP events use the <true, true> instance, so
// a kernel trace of the same run lists exactly the launches bench.py timed under their own name.
#ifndef RMEM_ATTN_WGS_PER_CU
#define RMEM_ATTN_WGS_PER_CU 3          // workgroups (= waves per SIMD) the register budget is set for
#endif
template <bool MEM, bool TIMED = false>
__global__ __launch_bounds__(256, RMEM_ATTN_WGS_PER_CU) void k_attn_partial(AttnParams pin) {
  AttnParams p = pin;
  __shared__ __attribute__((aligned(16))) e16 Ks[NB][KT * D];  // ring of NB tiles: [key][32], 16-byte chunks XOR-swizzled
  __shared__ __attribute__((aligned(16))) e16 Vs[NB][KT * D];  // [key][32] row-major, read transposed (ds_read_b64_tr_b16)
  // A row's ragged last tile (HW = 1674 = 26 * 64 + 10) is padded with K = V = 0 keys (the DMA's range check); their scores
  // are overwritten with -1e30 in that tile only (P = 0 exactly: no part in the maximum, the row sum or O).

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: LDS-DMA destinations stay in SGPRs
  const int lq = lane & 31, lh = lane >> 5;
  // XCD-aware decode of the 1-D grid: hardware deals consecutive block ids round-robin over the 8 XCDs (private 4 MiB
  // L2 each), so ids that are congruent mod 8 are made to walk (head, group) pairs contiguously -- all query tiles that
  // read the same K/V rows run on one XCD and share its L2 (speed only; any placement is correct).
  int qt, head, g;
  bool second = false;                 // this workgroup belongs to the second (plain, one-frame) attention of the launch
  {
    const int nq = p.nq, total1 = nq * p.heads * p.ngroups * p.nclips;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int qd = total1 >> 3, rm = total1 & 7;
    int idx = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + j;
    if (MEM && p.n2 > 0) {
      // every XCD gets its share of BOTH kinds: first its memory-read workgroups (j < n1), then second-attention ones; the
      // launch has cnt(x) = total / 8 (+ 1) blocks on XCD x, of which n1(x) = total1 / 8 (+ 1) are memory-read blocks
      const int total = total1 + nq * p.heads * p.n2, td = total >> 3, tr = total & 7;
      const int n1 = qd + (xcd < rm ? 1 : 0);
      if (j >= n1) {
        second = true;
        int start2 = 0;
        for (int y = 0; y < xcd; ++y) start2 += td + (y < tr ? 1 : 0) - (qd + (y < rm ? 1 : 0));
        idx = start2 + (j - n1);
      }
    }
    const int pair = idx / nq;
    qt = idx - pair * nq;
    head = pair % p.heads;
    const int rest = pair / p.heads;
    int clip;
    if (second) {
      g = 0; clip = rest;
      p.k = p.k2 + clip * p.kv2_cs; p.v = p.v2 + clip * p.kv2_cs;
      p.out = p.out2 + clip * p.out2_cs;
      p.ml = nullptr; p.pe_cur = nullptr; p.pe_mem = nullptr;
      p.nchunks = 1; p.rpg = 1;
    } else {
      g = rest % p.ngroups;
      clip = rest / p.ngroups;
      if (!MEM) { p.k += clip * p.kv_cs; p.v += clip * p.kv_cs; }
      else p.chunks += clip * p.nchunks;          // bank slots in the table are global (clip * slots + slot)
      p.opart += clip * p.opart_cs;
      p.mlg += clip * p.mlg_cs;
      if (p.ml) p.ml += clip * p.ml_cs;
      p.out += clip * p.out_cs;
    }
    p.q += clip * p.q_cs;
  }
  const int ngroups = second ? 1 : p.ngroups;
  const int c0 = g * p.rpg, c1 = min(c0 + p.rpg, p.nchunks);

  auto row_info = [&](int c) -> Row {
    Row r;
    if (MEM && second) {
      r.slot = 0; r.kb = 0; r.kn = p.lk2; r.pe_slot = -1;
    } else if (MEM) {
      // the table row is the same for the whole workgroup; say so (the row's descriptors must live in SGPRs, otherwise every
      // DMA is wrapped in a waterfall loop)
      const rmem_attn_chunk ch = p.chunks[c];
      r.slot = __builtin_amdgcn_readfirstlane(ch.slot); r.kb = __builtin_amdgcn_readfirstlane(ch.key_begin);
      r.kn = __builtin_amdgcn_readfirstlane(ch.key_count); r.pe_slot = __builtin_amdgcn_readfirstlane(ch.pe_slot);
    } else {
      r.slot = 0; r.kb = c * p.per_chunk; r.kn = min(p.per_chunk, p.lk - r.kb); r.pe_slot = -1;
    }
    return r;
  };

  // ---- Q^T fragment (B operand), pre-scaled, with the current-frame temporal PE added ----
  const int qrow = min(qt * 128 + wave * 32 + lq, p.Lq - 1);
  e16x8 qf[2];
  const bool has_cur = MEM && p.pe_cur != nullptr;        // workgroup-uniform
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int d0 = head * D + 16 * s + 8 * lh;
    const e16x8 raw = *reinterpret_cast<const e16x8*>(p.q + (long)qrow * p.ldq + d0);
    f32x4 c0v = {0.f, 0.f, 0.f, 0.f}, c1v = c0v;
    if (has_cur) { c0v = *reinterpret_cast<const f32x4*>(p.pe_cur + d0); c1v = *reinterpret_cast<const f32x4*>(p.pe_cur + d0 + 4); }
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[s][j] = (e16)(((float)raw[j] + (j < 4 ? c0v[j & 3] : c1v[j & 3])) * p.qscale);
  }
  // logit bias of a row: q' . pe_mem[pe_slot] (the reference adds pe_mem to the keys, transformer.py:594-626)
  auto row_bias = [&](int pe_slot) -> float {
    float bias = 0.f;
    if (MEM && pe_slot >= 0 && p.pe_mem != nullptr) {     // workgroup-uniform
      // (the fragments pass through an empty asm so that their 16 fp32 conversions are redone here, once per row, instead of
      // being hoisted into 16 registers that stay live across the whole tile loop)
      asm volatile("" : "+v"(qf[0]), "+v"(qf[1]));
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float* pm = p.pe_mem + pe_slot * p.C + head * D + 16 * s + 8 * lh;
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(pm), m1 = *reinterpret_cast<const f32x4*>(pm + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) bias += (float)qf[s][j] * (j < 4 ? m0[j & 3] : m1[j & 3]);
      }
      bias += __shfl_xor(bias, 32, 64);
    }
    return bias;
  };

  // ---- staging: a wave-instruction moves 16 keys x 64 B (this head's slice of K or V) straight into LDS.  Lane ->
  // (key = 16 * wave + lane / 4, LDS position lane % 4); the K tile's XOR swizzle is applied to the SOURCE chunk. ----
  const int skey = 16 * wave + (lane >> 2), spos = lane & 3;
  const int voff_k = (skey * p.ldkv + ((spos ^ ((skey >> 2) & 3)) << 3)) * 2;
  const int voff_v = (skey * p.ldkv + (spos << 3)) * 2;
  const int tile_elems = KT * p.ldkv;
  // The DMA cursor runs NB - 1 tiles ahead of the arithmetic, across row boundaries: (row dc, tile dt) with the row's base
  // pointers at this head's columns (wave-uniform: SGPRs).
  const e16* row_k = nullptr;
  const e16* row_v = nullptr;
  int row_bytes = 0, dc = 0, dt = 0, dnt = 0;
  auto open_row = [&](int c) {
    const Row r = row_info(c);
    const long off = (long)r.slot * p.slot_stride + (long)r.kb * p.ldkv + head * D;
    row_k = p.k + off; row_v = p.v + off;
    row_bytes = r.kn > 0 ? ((r.kn - 1) * p.ldkv + D) * 2 : 0;  // up to the end of the last key's slice; an EMPTY row (key_count 0:
                                                                // padding of a clip whose bank is shorter than its group's) is one
    dc = c; dt = 0; dnt = max(1, (r.kn + KT - 1) / KT);         // tile of out-of-range reads = zeros, its row sum is forced to 0
  };
  // Next tile of the group -> LDS buffer BUF (asynchronous: counted s_waitcnt vmcnt before use); false once the group is
  // exhausted.  The descriptor is rebuilt per tile (scalar work) so that it starts at the tile and ends with the row: the
  // hardware range check covers the per-lane offset only (not soffset), and this way a lane beyond the row's last key is out
  // of range by its own offset and reads zeros.
  auto dma_next = [&](auto buf_tag) -> bool {
    constexpr int BUF = decltype(buf_tag)::value;
    if (dc >= c1) return false;
    const int left = row_bytes - dt * tile_elems * 2;
#ifndef RMEM_ATTN_ABLATE_DMA        // timing experiments only
    buf_load_lds16(make_rsrc(row_k + (long)dt * tile_elems, left), (lptr_t)&Ks[BUF][wave * 16 * D], voff_k, 0);
    buf_load_lds16(make_rsrc(row_v + (long)dt * tile_elems, left), (lptr_t)&Vs[BUF][wave * 16 * D], voff_v, 0);
#endif
    if (++dt >= dnt) {
      if (dc + 1 < c1) open_row(dc + 1);
      else dc = c1;
    }
    return true;
  };

  // transposed-read addressing of the V tile: 16-lane group g reads a 4-key x 16-d block; lane 4q+p of the group
  // supplies the address of key row q, d columns 4p..4p+3 and receives d column (lane & 15), keys 0..3
  const int tr_off = ((4 * lh + ((lane & 15) >> 2)) * D) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  // LDS byte addresses of this lane's fragments in buffer 0 (buffer and block are instruction offsets)
  const unsigned k_addr[2] = {(unsigned)(size_t)(lptr_t)&Ks[0][kswz(lq, lh)], (unsigned)(size_t)(lptr_t)&Ks[0][kswz(lq, 2 + lh)]};
  const unsigned v_addr = (unsigned)(size_t)(lptr_t)&Vs[0][tr_off];

  // HW = 1674 = 13 * 128 + 10: in the last query tile only wave 0 owns real rows.  The other waves still stage K/V and meet
  // the barriers, but skip the softmax / MFMA work (3 of 56 wave-tiles per (head, group) saved).
  const bool wave_active = qt * 128 + wave * 32 < p.Lq;
  const int qg = qt * 128 + wave * 32 + lq;

  f32x16 oacc;
  float m_ref, ltot;
#ifdef RMEM_ATTN_TIMELINE
  unsigned long long tl_t0 = 0, tl_t1 = 0, tl_t2 = 0, tl_t3 = 0;
  unsigned tl_a = 0, tl_b = 0, tl_c = 0, tl_d = 0, tl_n = 0;
#endif

  // One pass over the group's rows.  SAFE = false: m_ref is the first tile's maximum and stays; true: online softmax.
  auto walk = [&](auto safe_tag) {
    constexpr bool SAFE = decltype(safe_tag)::value;
    f32x16 cinit;
    // row sum of the probabilities this LANE holds (the other 32 keys of every tile sit in lane ^ 32), two chains
    float lrow[2] = {0.f, 0.f};
    const e16x2 one2 = {(e16)1.0f, (e16)1.0f};
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    m_ref = 0.f; ltot = 0.f;
    int c = c0;
    Row cur = row_info(c);
    int ntiles = max(1, (cur.kn + KT - 1) / KT), t = 0;
    {
      // (bias - m_ref) replicated over the 16 accumulator registers: the C operand of the first S^T MFMA of every
      // block, rewritten only at a row boundary or a rescale, so no per-tile register fill is needed (D != C)
      const float bias = wave_active ? row_bias(cur.pe_slot) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[r] = bias;
    }
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>;
    using B3 = std::integral_constant<int, 3>;
    static_assert(NB == 4, "the step sequence below is written out for a ring of 4");
    open_row(c0);
    dma_next(B0{});
    dma_next(B1{});
    bool more = dma_next(B2{});      // false: every tile of the group has been issued

    // One 64-key tile out of LDS buffer BUF.  MAXMODE 0: no maximum at all (FAST pass after its first tile); 1: the group's
    // first tile (m_ref := tile maximum, nothing accumulated yet); 2: SAFE pass (move m_ref and rescale when a tile exceeds
    // it by 2^8).  The FAST steady state must not carry any of the extras: predicated off they are if-converted into ~60
    // subtractions / maxima per tile, which is exactly the VALU work this kernel cannot afford -- hence one copy per
    // MAXMODE.  ``valid`` = keys of this tile that exist (workgroup-uniform); only a row's last tile has fewer than KT.
    auto tile = [&](auto mode_tag, auto buf_tag, int valid) {
      constexpr int MAXMODE = decltype(mode_tag)::value;
      constexpr int BUF = decltype(buf_tag)::value;
      // S'^T = K . Q^T + (bias - m_ref) for the two 32-key blocks of this tile
      constexpr int KB = BUF * KT * D * 2, VB = BUF * KT * D * 2;      // byte offsets of the ring slot
      e16x8 ka[2][2];
#ifndef RMEM_ATTN_ABLATE_KREAD      // timing experiments only
      ka[0][0] = lds_b128<KB>(k_addr[0]);
      ka[0][1] = lds_b128<KB>(k_addr[1]);
      ka[1][0] = lds_b128<KB + 32 * D * 2>(k_addr[0]);
      ka[1][1] = lds_b128<KB + 32 * D * 2>(k_addr[1]);
      lds_wait4<0>(ka[0][0], ka[0][1], ka[1][0], ka[1][1]);
#ifdef RMEM_ATTN_TIMELINE
      RMEM_TL_STAMP(tl_t1, ka[1][1]);
#endif
#else
      ka[0][0] = qf[0]; ka[0][1] = qf[1]; ka[1][0] = qf[1]; ka[1][1] = qf[0];
      asm volatile("" : "+v"(ka[0][0]), "+v"(ka[0][1]), "+v"(ka[1][0]), "+v"(ka[1][1]));
#endif
      f32x16 sacc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // D != C spelled out in the steady state: left to the compiler, one of the two blocks is computed in place on a
        // 16-register COPY of cinit (8 v_mov_b64 per tile).  An MFMA inside inline asm is invisible to the compiler's hazard
        // recogniser, so it is used ONLY where nothing near it can be a hazard: in the steady-state tile cinit was last
        // written before the previous barrier (a VALU write of SrcC directly in front of the MFMA needs wait states -- the
        // first tile, whose cinit is a fresh constant, read stale registers through this path), and the marker below keeps
        // the operands allocated until both asm MFMAs have started (the matrix pipe reads SrcC when the instruction executes).
        if (MAXMODE == 0) asm volatile(RMEM_MFMA_32x32x16_ASM " %0, %1, %2, %3" : "=&v"(sacc[b]) : "v"(ka[b][0]), "v"(qf[0]), "v"(cinit));
        else sacc[b] = RMEM_MFMA_32x32x16(ka[b][0], qf[0], cinit, 0, 0, 0);
        sacc[b] = RMEM_MFMA_32x32x16(ka[b][1], qf[1], sacc[b], 0, 0, 0);
      }
      if (MAXMODE == 0) asm volatile("" : "+v"(sacc[1]) : "v"(cinit), "v"(ka[0][0]), "v"(ka[1][0]), "v"(qf[0]));
      if (valid < KT) {
        // ragged tile (one in 27 at HW = 1674), a real branch: the padded keys scored q . 0 + (bias - m_ref), which may lie far
        // ABOVE every real score (logits of a trained model can all be very negative), so neither "count them and subtract
        // them again" nor "let them meet V = 0" is safe -- their scores become -1e30 here.  Register r of block b and lane half
        // h is key 32 b + 8 (r >> 2) + 4 h + (r & 3).  The empty asm keeps the compiler from if-converting the block into 64
        // selects that every tile would execute.
        asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
        const int lim = valid - 4 * lh;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * b + 8 * (r >> 2) + (r & 3) >= lim) sacc[b][r] = NEG_BIG;
      }
      // the 4 reads of one 32-key block's V^T fragments; issued one block ahead of their use, so that their latency sits
      // under the exponentials (and at most one block's fragments are live)
      PvFrag pv;
      auto read_pv = [&](auto blk_tag) {
        constexpr int B = decltype(blk_tag)::value;
#ifdef RMEM_ATTN_ABLATE_VREAD       // timing experiments only
        asm volatile("" : "=v"(pv.lo[0]), "=v"(pv.hi[0]), "=v"(pv.lo[1]), "=v"(pv.hi[1]));
        return;
#endif
        pv.lo[0] = lds_tr16<VB + (B * 32) * D * 2>(v_addr);
        pv.hi[0] = lds_tr16<VB + (B * 32 + 8) * D * 2>(v_addr);
        pv.lo[1] = lds_tr16<VB + (B * 32 + 16) * D * 2>(v_addr);
        pv.hi[1] = lds_tr16<VB + (B * 32 + 24) * D * 2>(v_addr);
      };
      read_pv(std::integral_constant<int, 0>{});
      if (MAXMODE != 0) {
        // tile maximum per query (the other 16 keys of each block sit in lane ^ 32)
        float tmax = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(fmaxf(tmax, sacc[0][r]), sacc[1][r]);   // v_max3_f32
        tmax = pair_max(tmax);
        if (MAXMODE == 1) {
          if (tmax < 0.5f * NEG_BIG) tmax = 0.f;          // the group starts with an empty row: any reference will do
          m_ref = tmax;
#pragma unroll
          for (int r = 0; r < 16; ++r) cinit[r] -= tmax;
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[b][r] -= tmax;
        } else {
          const bool need = tmax > RESCALE_THR;
          if (__any(need)) {                     // rare: move the reference, rescale O and l
            const float delta = need ? tmax : 0.f;
            m_ref += delta;
            const float sc = __builtin_amdgcn_exp2f(-delta);
            ltot *= sc;
            lrow[0] *= sc; lrow[1] *= sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { oacc[r] *= sc; cinit[r] -= delta; }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int r = 0; r < 16; ++r) sacc[b][r] -= delta;
          }
        }
      }
      // P = exp2(S'), packed to e16: the B operand of O^T += V^T . P^T.  The row sum adds the SAME rounded probabilities: one
      // v_dot2c_f32 (pair . (1, 1) + acc, full-rate VALU) per packed register -- 16 per tile instead of the 4 "ones"-operand
      // MFMAs (a third of the matrix pipe's time for no FLOPs) and their 16 accumulator registers
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        e16x8 pb[2];
#pragma unroll
#ifndef RMEM_ATTN_ABLATE_EXP
        for (int r = 0; r < 16; ++r) {
          float ev = __builtin_amdgcn_exp2f(sacc[b][r]);
#ifdef RMEM_ATTN_TIMELINE
          if (b == 0 && r == 0) RMEM_TL_STAMP(tl_t2, ev);
#endif
          pb[r >> 3][r & 7] = (e16)ev;
        }
#else
        for (int r = 0; r < 16; ++r) pb[r >> 3][r & 7] = (e16)(sacc[b][r]);
#endif
#ifndef RMEM_ATTN_ABLATE_VREAD
        lds_wait_pv<0>(pv);
#endif
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const __attribute__((ext_vector_type(8))) short a16 = {pv.lo[sl][0], pv.lo[sl][1], pv.lo[sl][2], pv.lo[sl][3],
                                                                 pv.hi[sl][0], pv.hi[sl][1], pv.hi[sl][2], pv.hi[sl][3]};
          const e16x8 a = __builtin_bit_cast(e16x8, a16);
#ifndef RMEM_ATTN_ABLATE_PV         // timing experiments only
          oacc = RMEM_MFMA_32x32x16(a, pb[sl], oacc, 0, 0, 0);
#else
          oacc[sl] += (float)pb[sl][1] * (float)a[0];
#endif
#ifndef RMEM_ATTN_ABLATE_L      // timing experiments only (results are then wrong by construction)
#pragma unroll
          for (int j = 0; j < 4; ++j) lrow[sl] = rmem_dot2(e16x2{pb[sl][2 * j], pb[sl][2 * j + 1]}, one2, lrow[sl]);
#else
          lrow[sl] += (float)pb[sl][0];
#endif
        }
        if (b == 0) read_pv(std::integral_constant<int, 1>{});
      }
#ifdef RMEM_ATTN_TIMELINE
      RMEM_TL_STAMP(tl_t3, lrow[1]);
      tl_a += (unsigned)(tl_t1 - tl_t0); tl_b += (unsigned)(tl_t2 - tl_t1); tl_c += (unsigned)(tl_t3 - tl_t2); ++tl_n;
#endif
    };

    // One loop step on ring slot BUF.  The slot's tile was issued three steps ago; with the DMA cursor still running exactly
    // two younger tiles (4 wave-instructions) are in flight behind it, so vmcnt(4) means "my part of this tile has landed";
    // the barrier then says so for the whole workgroup AND that everybody is done with the previous tile, whose slot the
    // next DMA overwrites.  Returns true after the group's last tile.
    auto step = [&](auto mode_tag, auto buf_tag) -> bool {
      constexpr int BUF = decltype(buf_tag)::value;
      if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef RMEM_ATTN_ABLATE_BARRIER
      // a bare s_barrier: __syncthreads() is fence + barrier, and the fence drains EVERY LDS-DMA in flight (s_waitcnt vmcnt(0)
      // in front of the barrier) -- the two younger tiles included, which turns the ring into "load, wait, compute".  What the
      // barrier has to order here is covered by the counted waits: this wave's share of the tile has landed (vmcnt above), and
      // its LDS reads of the previous tile were waited for (lgkmcnt) before the MFMAs that consumed them.
      asm volatile("s_barrier" ::: "memory");
#endif
#ifdef RMEM_ATTN_TIMELINE
      {
        const unsigned long long prev = tl_t3;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_t0));
        if (prev) tl_d += (unsigned)(tl_t0 - prev);
      }
#endif
      if (more) more = dma_next(std::integral_constant<int, (BUF + NB - 1) % NB>{});
      const bool last_in_row = t + 1 >= ntiles;
      const bool last = last_in_row && c + 1 >= c1;
      if (wave_active) tile(mode_tag, buf_tag, cur.kn - t * KT);
      if (last_in_row) {
        if (wave_active) {
          float lr = lrow[0] + lrow[1];
          lr += __shfl_xor(lr, 32, 64);                  // + the other half of the keys
          // the row's own (reference, sum): all the mass output needs; the running total keeps the group's normaliser
          if (p.ml != nullptr && lh == 0 && qg < p.Lq)
            *reinterpret_cast<f32x2*>(p.ml + (((long)c * p.heads + head) * p.Lq + qg) * 2) = f32x2{m_ref, lr};
          ltot += lr;
          lrow[0] = 0.f; lrow[1] = 0.f;
        }
        if (last) return true;
        ++c; cur = row_info(c); t = 0;
        ntiles = max(1, (cur.kn + KT - 1) / KT);
        if (wave_active) {
          const float bias = row_bias(cur.pe_slot) - m_ref;
#pragma unroll
          for (int r = 0; r < 16; ++r) cinit[r] = bias;
        }
      } else {
        ++t;
      }
      return false;
    };
    constexpr int STEADY = SAFE ? 2 : 0;
    using M1 = std::integral_constant<int, 1>;
    using MS = std::integral_constant<int, STEADY>;
    if (step(M1{}, B0{})) return;
    for (;;) {                                   // four steps per trip: the ring slot is a compile-time constant
      if (step(MS{}, B1{})) return;
      if (step(MS{}, B2{})) return;
      if (step(MS{}, B3{})) return;
      if (step(MS{}, B0{})) return;
    }
  };

#ifdef RMEM_F16
  // IEEE half carries P only up to 2^16: the reference must be a running maximum, so this flavour always takes the SAFE pass
  walk(std::true_type{});
#else
  walk(std::false_type{});
  // any query of the workgroup whose sums left the safe range sends the whole group through the online-softmax pass
  // (the vote is also the barrier that frees the LDS tiles for it)
#ifndef RMEM_ATTN_ABLATE_FALLBACK     // timing experiments only (their garbage operands must not send every group through both passes)
  if (__syncthreads_or(wave_active && !(ltot <= L_LIMIT))) walk(std::true_type{});
#endif
#endif
#ifdef RMEM_ATTN_TIMELINE
  if (tid == 0 && blockIdx.x < 8192) {
    unsigned* o = g_attn_timeline + blockIdx.x * 8;
    o[0] = tl_a; o[1] = tl_b; o[2] = tl_c; o[3] = tl_d; o[4] = tl_n;
  }
#endif
  if (!wave_active) return;

  if (ngroups == 1) {
    if (qg < p.Lq) {
      const float inv = 1.f / ltot;
      e16* o = p.out + (long)qg * p.ldo + head * D + 4 * lh;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)    // C/D rows (r&3) + 8(r>>2) + 4h -> d = 8g + 4h + (0..3)
        *reinterpret_cast<e16x4*>(o + 8 * gg) = e16x4{(e16)(oacc[4 * gg] * inv), (e16)(oacc[4 * gg + 1] * inv),
                                                        (e16)(oacc[4 * gg + 2] * inv), (e16)(oacc[4 * gg + 3] * inv)};
    }
    return;
  }
  if (qg < p.Lq) {
    // partial O layout [group][head][G = d / 4][q] x e16x4: a half-wave stores 256 contiguous bytes per instruction.  The
    // partial is normalised by the group's own sum and rounded to e16 -- the merge is a convex combination of the groups, so
    // this costs one more e16 rounding of the output's own size and halves the bytes the partials write and the merge reads
    // (they were a quarter of the launch's HBM traffic); the weights (m_ref, l) stay fp32.
    const long ch = (long)g * p.heads + head;
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    e16x4* o = reinterpret_cast<e16x4*>(p.opart) + ch * 8 * p.Lq + qg;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)  // C/D rows (r&3) + 8(r>>2) + 4h  ->  d = 8g + 4h + (0..3)  ->  G = 2g + h
      o[(long)(2 * gg + lh) * p.Lq] = e16x4{(e16)(oacc[4 * gg] * inv), (e16)(oacc[4 * gg + 1] * inv), (e16)(oacc[4 * gg + 2] * inv),
                                            (e16)(oacc[4 * gg + 3] * inv)};
    if (lh == 0) *reinterpret_cast<f32x2*>(p.mlg + (ch * p.Lq + qg) * 2) = f32x2{m_ref, ltot};
  }
}

struct CombineParams {
  const float* opart; const float* mlg; int ngroups;      // per key group: partial O, (m, l)
  const float* ml; const rmem_attn_chunk* chunks; int nchunks;   // per table row: (m, l_row) -> mass
  int Lq, heads;
  e16* out; int ldo;
  float* mass; int T;
  long out_cs, opart_cs, mlg_cs, ml_cs, mass_cs;     // per-clip strides (blockIdx.z = clip)
};

// merge the key groups.  grid = (query blocks of 64, 16, clips); thread = (query, head, 4 channels): consecutive lanes
// read consecutive queries of the [group][head][G][q] partial layout.
__global__ __launch_bounds__(256) void k_attn_combine(CombineParams pin) {
  CombineParams p = pin;
  p.opart += blockIdx.z * p.opart_cs; p.mlg += blockIdx.z * p.mlg_cs; p.out += blockIdx.z * p.out_cs;
  const int tid = threadIdx.x;
  const int ql = tid & 63;
  const int q = blockIdx.x * 64 + ql;
  const bool live = q < p.Lq;
  const int qc = live ? q : p.Lq - 1;
  const int hg = blockIdx.y * 4 + (tid >> 6);  // 0..63
  const int head = hg >> 3, G = hg & 7;
  if (head < p.heads) {
    const e16x4* op = reinterpret_cast<const e16x4*>(p.opart);
    float m = NEG_BIG;      // (groups / rows that saw no real key have l = 0: their reference is meaningless and must not set the scale)
    for (int c = 0; c < p.ngroups; ++c) {
      const f32x2 mlv = *reinterpret_cast<const f32x2*>(p.mlg + (((long)c * p.heads + head) * p.Lq + qc) * 2);
      if (mlv[1] > 0.f) m = fmaxf(m, mlv[0]);
    }
    f32x4 num = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int c = 0; c < p.ngroups; ++c) {
      const long ch = (long)c * p.heads + head;
      const f32x2 mlv = *reinterpret_cast<const f32x2*>(p.mlg + (ch * p.Lq + qc) * 2);
      if (!(mlv[1] > 0.f)) continue;
      const float w = __builtin_amdgcn_exp2f(mlv[0] - m) * mlv[1];      // the partial is normalised: its weight is w l
      den += w;
      const e16x4 o4 = op[(ch * 8 + G) * p.Lq + qc];
      num += f32x4{(float)o4[0], (float)o4[1], (float)o4[2], (float)o4[3]} * w;
    }
    if (live) {
      const f32x4 o = num * (1.f / den);
      *reinterpret_cast<e16x4*>(p.out + (long)q * p.ldo + head * D + 4 * G) = e16x4{(e16)o[0], (e16)o[1], (e16)o[2], (e16)o[3]};
    }
  }
}

// per-memory-frame probability mass from the per-row (m, l) pairs: mass[q][t] = mean_h sum_{rows of t} w_r l_r / den_h.
// thread = (query, head pair); 64 queries per block
__global__ __launch_bounds__(256) void k_attn_mass(CombineParams pin) {
  CombineParams p = pin;
  p.ml += blockIdx.z * p.ml_cs; p.chunks += blockIdx.z * p.nchunks; p.mass += blockIdx.z * p.mass_cs;
  __shared__ float macc[4][64][33];
  __shared__ int ct[32];
  const int tid = threadIdx.x, ql = tid & 63, hq = tid >> 6;
  const int q = blockIdx.x * 64 + ql;
  const int qc = q < p.Lq ? q : p.Lq - 1;
  if (tid < p.nchunks) ct[tid] = p.chunks[tid].t;
  for (int t = 0; t < 33; ++t) macc[hq][ql][t] = 0.f;
  __syncthreads();
  for (int h = hq; h < p.heads; h += 4) {
    float m = NEG_BIG, den = 0.f;
    for (int c = 0; c < p.nchunks; ++c) {
      const f32x2 mlv = *reinterpret_cast<const f32x2*>(p.ml + (((long)c * p.heads + h) * p.Lq + qc) * 2);
      if (mlv[1] > 0.f) m = fmaxf(m, mlv[0]);
    }
    for (int c = 0; c < p.nchunks; ++c) {
      const f32x2 mlv = *reinterpret_cast<const f32x2*>(p.ml + (((long)c * p.heads + h) * p.Lq + qc) * 2);
      if (mlv[1] > 0.f) den += __builtin_amdgcn_exp2f(mlv[0] - m) * mlv[1];
    }
    const float inv = 1.f / (den * (float)p.heads);
    for (int c = 0; c < p.nchunks; ++c) {
      const f32x2 mlv = *reinterpret_cast<const f32x2*>(p.ml + (((long)c * p.heads + h) * p.Lq + qc) * 2);
      if (mlv[1] > 0.f) macc[hq][ql][ct[c]] += __builtin_amdgcn_exp2f(mlv[0] - m) * mlv[1] * inv;
    }
  }
  __syncthreads();
  for (int i = tid; i < 64 * p.T; i += 256) {
    const int qq = i / p.T, t = i - qq * p.T;
    const int qo = blockIdx.x * 64 + qq;
    if (qo < p.Lq) p.mass[(long)qo * p.T + t] = macc[0][qq][t] + macc[1][qq][t] + macc[2][qq][t] + macc[3][qq][t];
  }
}

}  // namespace

#ifndef RMEM_F16
extern "C" size_t rmem_attn_workspace_bytes(int Lq, int heads, int nchunks) {
  // worst case one group per table row: partial O (D floats) + group (m, l) + row (m, l) per (row, head, query)
  return (size_t)nchunks * heads * Lq * (D + 4) * sizeof(float);
}
#endif

#if defined(RMEM_ATTN_TIMELINE) && !defined(RMEM_F16)
// out[0..4] = mean over the first nwg workgroups of (A, B, C, D) cycles per tile and tiles per workgroup (measurement build only)
extern "C" int rmem_attn_timeline_read(double* out, int nwg) {
  static unsigned host[8192 * 8];
  if (nwg > 8192) nwg = 8192;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_timeline), sizeof(unsigned) * 8 * nwg) != hipSuccess) return 2;
  double s[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < nwg; ++i) for (int j = 0; j < 5; ++j) s[j] += host[i * 8 + j];
  for (int j = 0; j < 4; ++j) out[j] = s[4] > 0 ? s[j] / s[4] : 0;
  out[4] = s[4] / nwg;
  return 0;
}
#endif

// workgroups a launch should have before several table rows are walked by one workgroup (7 per CU; experiments:
// RMEM_ATTN_WGS).  Fewer, longer workgroups write fewer fp32 partials (none at all with one group).
// occupancy experiments only (RMEM_ATTN_LDS_PAD = bytes): extra dynamic LDS per workgroup limits the workgroups per CU
static unsigned attn_lds_pad() {
  const char* e = getenv("RMEM_ATTN_LDS_PAD");
  const int x = e ? atoi(e) : 0;
  return x > 0 ? (unsigned)x : 0u;
}

static int attn_target_wgs() {
  const char* e = getenv("RMEM_ATTN_WGS");        // read per call (not inside graph replays): tests force 1 group / 1 row per group
  const int x = e ? atoi(e) : 0;
  return x > 0 ? x : 1792;
}

static int attn_launch(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride,
                       int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_single,
                       const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out, int ldo,
                       float* attn_mass, int T, int nclips, long long q_clip_stride, long long kv_clip_stride,
                       long long out_clip_stride, void* workspace, void* stream,
                       const void* k2, const void* v2, int lk2, long long kv2_clip_stride, void* out2, long long out2_clip_stride) {
  RMEM_REQUIRE(nclips >= 1 && nclips <= 64, "rmem_mem_read_attn: 1..64 clips");
  RMEM_REQUIRE(q_clip_stride % 8 == 0 && kv_clip_stride % 8 == 0 && out_clip_stride % 8 == 0, "rmem_mem_read_attn: clip strides must be multiples of 8 elements");
  const long long prof_keys = chunks ? (long long)lk_single : 0;   // with a row table lk_single carries the total key count (timing only)
  RMEM_REQUIRE(q && k_bank && v_bank && out && workspace, "rmem_mem_read_attn: null argument");
  RMEM_REQUIRE(heads >= 1 && heads <= 8, "rmem_mem_read_attn: heads must be in 1..8 (head dim is fixed at 32)");
  RMEM_REQUIRE(Lq > 0 && nchunks >= 1 && nchunks <= 32, "rmem_mem_read_attn: need Lq > 0 and 1 <= nchunks <= 32");
  RMEM_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && slot_stride % 8 == 0, "rmem_mem_read_attn: strides must be multiples of 8 elements");
  RMEM_REQUIRE(ldq >= heads * D && ldkv >= heads * D && ldo >= heads * D, "rmem_mem_read_attn: leading dimension < heads*32");
  RMEM_REQUIRE(chunks || lk_single > 0, "rmem_mem_read_attn: lk_single must be > 0 when no chunk table is given");
  RMEM_REQUIRE(lk_single >= 0, "rmem_mem_read_attn: lk_single must be >= 0");
  RMEM_REQUIRE(!attn_mass || (chunks && T >= 1 && T <= 32), "rmem_mem_read_attn: the mass output needs a chunk table and 1 <= T <= 32");
  RMEM_REQUIRE(!pe_mem || chunks, "rmem_mem_read_attn: pe_mem needs a chunk table");
  hipStream_t s = (hipStream_t)stream;
  AttnParams p;
  p.q = (const e16*)q; p.ldq = ldq; p.k = (const e16*)k_bank; p.v = (const e16*)v_bank;
  p.slot_stride = slot_stride; p.ldkv = ldkv; p.chunks = chunks; p.nchunks = nchunks;
  p.lk = lk_single; p.per_chunk = chunks ? 0 : (lk_single + nchunks - 1) / nchunks;
  RMEM_REQUIRE(chunks || (long)p.per_chunk * (nchunks - 1) < lk_single, "rmem_mem_read_attn: too many chunks for lk_single");
  p.pe_cur = pe_cur; p.pe_mem = pe_mem; p.Lq = Lq; p.heads = heads; p.C = heads * D;
  p.nclips = nclips; p.q_cs = q_clip_stride; p.kv_cs = kv_clip_stride; p.out_cs = out_clip_stride;
  p.qscale = 1.4426950408889634f / sqrtf((float)D);
  p.nq = (Lq + 127) / 128;
  // rows per workgroup: as many as keep >= attn_target_wgs() workgroups in the launch
  {
    const int base = p.nq * heads * nclips;
    int groups = (attn_target_wgs() + base - 1) / base;
    groups = groups < 1 ? 1 : (groups > nchunks ? nchunks : groups);
    p.rpg = (nchunks + groups - 1) / groups;
    p.ngroups = (nchunks + p.rpg - 1) / p.rpg;
  }
  // workspace: [clip][group] partial O | [clip][group] (m, l) | [clip][row] (m, l)
  p.opart_cs = (long)p.ngroups * heads * Lq * D; p.mlg_cs = (long)p.ngroups * heads * Lq * 2; p.ml_cs = (long)nchunks * heads * Lq * 2;
  p.opart = (float*)workspace; p.mlg = p.opart + (size_t)nclips * p.opart_cs;
  p.ml = attn_mass ? p.mlg + (size_t)nclips * p.mlg_cs : nullptr;
  p.out = (e16*)out; p.ldo = ldo;
  p.n2 = 0; p.k2 = p.v2 = nullptr; p.lk2 = 0; p.kv2_cs = 0; p.out2 = nullptr; p.out2_cs = 0;
  if (k2) {
    RMEM_REQUIRE(chunks && v2 && out2 && lk2 > 0, "rmem_lstt_attn_pair: the second attention rides on a memory read (chunk table) and needs k2, v2, out2, lk2 > 0");
    RMEM_REQUIRE(kv2_clip_stride % 8 == 0 && out2_clip_stride % 8 == 0 && ((uintptr_t)k2 % 16) == 0 && ((uintptr_t)v2 % 16) == 0,
                 "rmem_lstt_attn_pair: second attention operands must be 16-byte aligned, strides multiples of 8 elements");
    RMEM_REQUIRE(p.nq * heads * nclips >= 8, "rmem_lstt_attn_pair: launch too small for the per-XCD split");
    p.n2 = nclips; p.k2 = (const e16*)k2; p.v2 = (const e16*)v2; p.lk2 = lk2; p.kv2_cs = kv2_clip_stride;
    p.out2 = (e16*)out2; p.out2_cs = out2_clip_stride;
  }
  dim3 grid(p.nq * heads * (p.ngroups * nclips + p.n2));
  if (chunks) {
    // time this launch if asked to (rmem_profile_start; never while the stream is being captured into a graph)
    const int slot_i = prof_keys > 0 ? rmem_prof_begin(RMEM_PROF_MEM_READ, s, 4.0 * (double)Lq * (double)prof_keys * (double)(heads * D) * nclips)
                                     : -1;                                                  // FLOPs: QK^T + PV
    if (slot_i >= 0) hipLaunchKernelGGL((k_attn_partial<true, true>), grid, dim3(256), attn_lds_pad(), s, p);
    else hipLaunchKernelGGL((k_attn_partial<true, false>), grid, dim3(256), attn_lds_pad(), s, p);
    if (slot_i >= 0) rmem_prof_end(RMEM_PROF_MEM_READ, slot_i, s);
  } else {
    hipLaunchKernelGGL((k_attn_partial<false, false>), grid, dim3(256), 0, s, p);
  }
  CombineParams cp;
  cp.opart = p.opart; cp.mlg = p.mlg; cp.ngroups = p.ngroups; cp.ml = p.ml; cp.chunks = chunks; cp.nchunks = nchunks;
  cp.Lq = Lq; cp.heads = heads; cp.out = (e16*)out; cp.ldo = ldo; cp.mass = attn_mass; cp.T = T;
  cp.out_cs = out_clip_stride; cp.opart_cs = p.opart_cs; cp.mlg_cs = p.mlg_cs; cp.ml_cs = p.ml_cs; cp.mass_cs = (long)Lq * T;
  if (p.ngroups > 1) hipLaunchKernelGGL(k_attn_combine, dim3((Lq + 63) / 64, 16, nclips), dim3(256), 0, s, cp);
  if (attn_mass) hipLaunchKernelGGL(k_attn_mass, dim3((Lq + 63) / 64, 1, nclips), dim3(256), 0, s, cp);
  return rmem_check_launch("rmem_mem_read_attn");
}

extern "C" int RMEM_API(rmem_mem_read_attn_clips)(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride,
                                        int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_single,
                                        const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out, int ldo,
                                        float* attn_mass, int T, int nclips, long long q_clip_stride, long long kv_clip_stride,
                                        long long out_clip_stride, void* workspace, void* stream) {
  return attn_launch(q, ldq, k_bank, v_bank, slot_stride, ldkv, chunks, nchunks, lk_single, pe_cur, pe_mem, Lq, heads, out, ldo, attn_mass, T,
                     nclips, q_clip_stride, kv_clip_stride, out_clip_stride, workspace, stream, nullptr, nullptr, 0, 0, nullptr, 0);
}

extern "C" int RMEM_API(rmem_lstt_attn_pair_clips)(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride,
                                         int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_total,
                                         const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out_long, int ldo,
                                         float* attn_mass, int T, int nclips, long long q_clip_stride, long long out_clip_stride,
                                         const void* k_short, const void* v_short, int lk_short, long long kv_short_clip_stride,
                                         void* out_short, long long out_short_clip_stride, void* workspace, void* stream) {
  RMEM_REQUIRE(chunks && k_short && v_short && out_short, "rmem_lstt_attn_pair: null argument");
  return attn_launch(q, ldq, k_bank, v_bank, slot_stride, ldkv, chunks, nchunks, lk_total, pe_cur, pe_mem, Lq, heads, out_long, ldo, attn_mass, T,
                     nclips, q_clip_stride, 0, out_clip_stride, workspace, stream, k_short, v_short, lk_short, kv_short_clip_stride, out_short,
                     out_short_clip_stride);
}

extern "C" int RMEM_API(rmem_mem_read_attn)(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride,
                                  int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_single,
                                  const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out, int ldo,
                                  float* attn_mass, int T, void* workspace, void* stream) {
  return RMEM_API(rmem_mem_read_attn_clips)(q, ldq, k_bank, v_bank, slot_stride, ldkv, chunks, nchunks, lk_single, pe_cur, pe_mem, Lq, heads, out,
                                  ldo, attn_mass, T, 1, 0, 0, 0, workspace, stream);
}
