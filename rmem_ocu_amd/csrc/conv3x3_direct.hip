// 3x3 stride-1 convolution with 64 input and 64 output channels (+ folded BatchNorm, ReLU) read in place from an LDS patch: the second
// conv of the ResNet layer-1 bottlenecks (encoders/resnet.py:52-56), 121 x 213 pixels x 16 frames per launch in the bench.
//
// Same idea as stem.hip.  The implicit-GEMM form (gemm_conv.hip) copies every activation byte through LDS-DMA once per filter tap
// (9 x) plus a weight panel per k-step, and is bound by that copy.  Here a workgroup takes a strip of 64 output columns and walks down it:
//   * each input row of the strip (66 pixels x 64 channels, 8.4 KB) goes to LDS ONCE, into a ring of four row slots, with zero fill
//     outside the image (out-of-range offsets) and the 16-byte pieces permuted at the SOURCE into a column-interleaved layout, so
//     that the B fragment of output pixel p, tap (dy, dx), channel slice s is one conflict-free ds_read_b128 at column p + dx of the
//     slot of row y - 1 + dy, at any alignment;
//   * the whole weight tensor of this wave's 32 output channels (2 x 18 fragments, K = 9 taps x 64) lives in REGISTERS for the life
//     of a workgroup: the weights are the A operand, so a lane's accumulators are 4 consecutive channels of a pixel.
// The k order (tap-major, then input channel) and the epilogue arithmetic are those of rmem_conv2d_nhwc: results are bit-identical
// (tests/test_hip_ops.py::test_conv3x3_c64_direct).  256 threads = 2 (channel halves) x 2 (pixel halves) waves, the next input row in flight
// behind a counted vmcnt, one shared object and raw barriers (see stem.hip for why).
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

constexpr int TP = 64;                          // output pixels a tile computes (one piece of an output row) ...
constexpr int TPV = 62;                         // ... of which it delivers 62: the input row piece is then 64 pixels = 8 KB = exactly 2 LDS-DMA
constexpr int PWP = 64;                         // instructions per thread, and a ring slot has no dead bytes
constexpr int RCHUNKS = PWP * 8;                // 16-byte pieces of one input row of a strip (128 B per pixel)
constexpr int NDMA = RCHUNKS / 256;             // LDS-DMA instructions per thread and row
constexpr int ROW_BYTES = RCHUNKS * 16;         // one slot of the row ring
constexpr int NSLOT = 8, DEPTH = NSLOT - 3;     // three rows in use, five in flight
constexpr int SROW = 144;                       // staging row (see stem.hip)
constexpr int OOB = (int)0x80000000;
static_assert(NDMA == 2 && NSLOT == 8, "the counted waits below assume 2 requests per row and per store pass");

struct C3Params {
  const e16* x; const e16* w; const float* bias; e16* y;
  int images, H, W, tiles_x, nruns, run_len;
  long x_bytes;
};

// A workgroup owns a column strip of 62 output pixels and walks DOWN a run of output rows: output row y reads input rows y - 1 .. y + 1,
// of which two were already read for row y - 1, so the LDS holds a ring of row slots and every input row of the strip is fetched once
// per run.  Walking along the row with one 3-row patch in flight (the first version) the kernel ran at one DMA round trip per tile:
// 47.6 us per 16 frames whatever the arithmetic; the ring keeps FIVE rows in flight per workgroup.
__global__ __launch_bounds__(256) void k_conv3x3_c64(C3Params p) {
  __shared__ __attribute__((aligned(16))) char smem[NSLOT * ROW_BYTES + TP * SROW];     // ONE shared object (see stem.hip)
  char* const stage = smem + NSLOT * ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;      // channel half, pixel half
  const int fr = lane & 15, fc = lane >> 4;
  const int u = blockIdx.x;
  const int xt = u % p.tiles_x, run = (u / p.tiles_x) % p.nruns, img = u / (p.tiles_x * p.nruns);
  const int x0 = xt * TPV, y0 = run * p.run_len, y1 = min(p.H, y0 + p.run_len);
  if (y0 >= y1) return;
  // descriptor base = one pixel before image 0, so that column x0 - 1 has a non-negative offset
  const rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) - 128, p.x_bytes + 128);

  // this thread's pieces of a row.  A slot is laid out [column block of 16][channel chunk of 8][column in block] x 16 B: the 16 lanes
  // of a ds_read_b128 group read 16 CONSECUTIVE columns (whatever the tap shift, whatever the chunk), which then fall on 16 distinct
  // 16-byte bank slots -- no conflicts at any alignment (the GEMM panels' XOR swizzle is conflict-free only for aligned column
  // groups: 41 % of this kernel's LDS cycles were conflicts with it).  LDS-DMA fills a slot linearly, so the permutation is applied
  // to the SOURCE: piece q = tid + 256 i is column (q >> 7) * 16 + (q & 15), chunk (q >> 4) & 7.
  int poff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int q = tid + 256 * i;
    const int col = (q >> 7) * 16 + (q & 15), kc = (q >> 4) & 7;
    poff[i] = (unsigned)(x0 + col - 1) < (unsigned)p.W ? col * 128 + kc * 16 : OOB;
  }
  auto issue_row = [&](int row) {               // input row `row` of the strip -> slot row & 7 (rows outside the image: zero fill)
    const bool ok = (unsigned)row < (unsigned)p.H;
    const int base = ((img * p.H + (ok ? row : 0)) * p.W + x0) * 128;
    char* dst = smem + (row & (NSLOT - 1)) * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) buf_load_lds16(rs, (lptr_t)(dst + (wave * 64 + 256 * i) * 16), ok ? poff[i] : OOB, base);
  };

#pragma unroll
  for (int r = -1; r <= DEPTH; ++r) issue_row(y0 + r);
  __builtin_amdgcn_sched_barrier(0);
  // weights [64][3][3][64]: A operand (rows = output channels): lane -> channel wc * 32 + 16 ct + (lane & 15), k = 64 tap + 32 s + 8 (lane >> 4) ..
  e16x8 wf[2][18];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
      wf[ct][ks] = *reinterpret_cast<const e16x8*>(p.w + (wc * 32 + ct * 16 + fr) * 576 + ks * 32 + fc * 8);
  f32x4 bv[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(p.bias + wc * 32 + ct * 16 + fc * 4);
  __builtin_amdgcn_sched_barrier(0);
  // byte offset of this lane's (pixel, chunk fc) inside a row slot under horizontal tap dx, channel slice 0 (slice 1: + 4 chunk windows)
  int coff[2][3];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int col = wp * 32 + pt * 16 + fr + dx;        // (columns 64, 65 -- outputs 62, 63, never stored -- read the next slot)
      coff[pt][dx] = (((col >> 4) * 8 + fc) * 16 + (col & 15)) * 16;
    }
  const int npix = min(TPV, p.W - x0);

  for (int y = y0; y < y1; ++y) {
    // row y + 1 must have landed.  Requests are counted in issue order: behind row y + 1 there are, in the steady state, the two stores
    // of the iteration that requested it and (2 requests + 2 stores) of each of the DEPTH - 1 iterations since: 4 DEPTH - 2 = 18; while
    // the rows of the prologue are being consumed at least 2 (DEPTH - 1) + 2 (y - y0) >= 10 (8 is used); the first row waits for all.
    if (y == y0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (y - y0 < DEPTH) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    __builtin_amdgcn_s_barrier();               // (raw barriers: see stem.hip)
    issue_row(y + DEPTH + 1);                   // unconditionally; its slot held row y - 2
    const long pix0 = ((long)img * p.H + y) * p.W + x0;

    f32x4 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 18 k-slices (tap, channel half), software-pipelined: the two B fragments of slice k + 1 are requested before the four MFMAs of
    // slice k (at two waves per SIMD an exposed LDS round trip per slice is most of the tile otherwise)
    const char* rb[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) rb[dy] = smem + ((y - 1 + dy) & (NSLOT - 1)) * ROW_BYTES;
    e16x8 bf[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) bf[0][pt] = *reinterpret_cast<const e16x8*>(rb[0] + coff[pt][0]);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      if (k + 1 < 18) {
        const int tap = (k + 1) >> 1, s = (k + 1) & 1, dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) bf[(k + 1) & 1][pt] = *reinterpret_cast<const e16x8*>(rb[dy] + coff[pt][dx] + s * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = RMEM_MFMA_16x16x32(wf[ct][k], bf[k & 1][pt], acc[ct][pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        e16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (e16)fmaxf(acc[ct][pt][r] + bv[ct][r], 0.f);
        *reinterpret_cast<e16x4*>(stage + (wp * 32 + pt * 16 + fr) * SROW + (wc * 32 + ct * 16 + fc * 4) * 2) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {               // both stores always issued; pixels past the row end repeat its last pixel
      const int v = tid + 256 * j, px = min(v >> 3, npix - 1), c16 = v & 7;
      const e16x8 o = *reinterpret_cast<const e16x8*>(stage + px * SROW + c16 * 16);
      *reinterpret_cast<e16x8*>(p.y + (pix0 + px) * 64 + c16 * 8) = o;
    }
  }
}

}  // namespace

extern "C" int RMEM_API(rmem_conv3x3_c64_direct)(const void* x, int images, int H, int W, const void* w, const float* bias, void* y, void* stream) {
  RMEM_REQUIRE(x && w && bias && y && images >= 1 && H >= 1 && W >= 1, "rmem_conv3x3_c64_direct: bad argument");
  RMEM_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)y % 16) == 0,
               "rmem_conv3x3_c64_direct: operands must be 16-byte aligned");
  C3Params p;
  p.x = (const e16*)x; p.w = (const e16*)w; p.bias = bias; p.y = (e16*)y;
  p.images = images; p.H = H; p.W = W;
  p.tiles_x = (W + TPV - 1) / TPV;
  p.x_bytes = (long)images * H * W * 128;
  RMEM_REQUIRE(p.x_bytes + (long)(W + TP + 2) * 128 < (1L << 31) - (1L << 22), "rmem_conv3x3_c64_direct: the input exceeds the 2 GB a buffer descriptor addresses");
  // runs of output rows per column strip: enough workgroups for two per CU, rows per run as long as that allows (every run re-reads 2 rows)
  static const int wgs = getenv("RMEM_CONV3_WGS") ? atoi(getenv("RMEM_CONV3_WGS")) : 512;
  const long strips = (long)images * p.tiles_x;
  long nruns = (wgs + strips - 1) / strips;
  if (nruns < 1) nruns = 1;
  if (nruns > H) nruns = H;
  p.run_len = (int)((H + nruns - 1) / nruns);
  p.nruns = (H + p.run_len - 1) / p.run_len;
  const long grid = strips * p.nruns;
  RMEM_REQUIRE(grid < (1L << 30), "rmem_conv3x3_c64_direct: too many workgroups");
  hipLaunchKernelGGL(k_conv3x3_c64, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_conv3x3_c64_direct");
}
