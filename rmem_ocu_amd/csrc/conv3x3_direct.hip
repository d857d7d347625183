// 3x3 stride-1 convolution with C input and C output channels, C = 64 or 128 (+ bias, optional ReLU), read in place from rows kept in LDS:
// the second conv of the ResNet layer-1 / layer-2 bottlenecks (encoders/resnet.py:52-56; 121 x 213 and 61 x 107 pixels x 16 frames per
// launch in the bench) and the decoder's conv_4x (decoders/fpn.py:54-58, 121 x 213 x 8 clips).
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
// (tests/test_hip_ops.py::test_conv3x3_direct).  256 threads = 2 (channel halves) x 2 (pixel halves) waves, the next input row in flight
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
constexpr int TPV = 62;                         // ... of which it delivers 62: the input row piece is then 64 pixels = exactly 2 LDS-DMA
constexpr int PWP = 64;                         // instructions per thread, and a ring slot has no dead bytes
constexpr int NSLOT = 8, DEPTH = NSLOT - 3;     // three rows in use, five in flight
constexpr int OOB = (int)0x80000000;

struct C3Params {
  const e16* x; const e16* w; const float* bias; e16* y;
  int images, H, W, tiles_x, nruns, run_len, relu;
  long x_bytes;
};

// A workgroup owns a column strip of 62 output pixels and walks DOWN a run of output rows: output row y reads input rows y - 1 .. y + 1,
// of which two were already read for row y - 1, so the LDS holds a ring of row slots and every input row of the strip is fetched once
// per run.  Walking along the row with one 3-row patch in flight (the first version) the kernel ran at one DMA round trip per tile:
// 47.6 us per 16 frames whatever the arithmetic; the ring keeps FIVE rows in flight per workgroup.
// C = 64: 4 waves = 2 (channel halves of 32) x 2 (pixel halves of 32), two workgroups per CU.  C = 128: 8 waves, each 16 output channels
// (36 weight fragments: the same 144 registers) x all 64 pixels, one workgroup per CU (ring 128 KB).
template <int C>
__global__ __launch_bounds__(C * 4) void k_conv3x3_direct(C3Params p) {
  constexpr int NT = C * 4;                     // threads: 256 / 512
  constexpr int CT = C == 64 ? 2 : 1;           // 16-channel tiles per wave
  constexpr int PT = C == 64 ? 2 : 4;           // 16-pixel tiles per wave
  constexpr int KC = C / 8;                     // 16-byte channel chunks per pixel
  constexpr int KS = 9 * C / 32;                // k-slices of 32: tap-major, then input channel (the GEMM's k order)
  constexpr int SPT = C / 32;                   // slices per tap
  constexpr int PIXB = C * 2;                   // bytes per pixel
  constexpr int ROW_BYTES = PWP * PIXB;         // one slot of the row ring
  constexpr int NDMA = ROW_BYTES / 16 / NT;     // LDS-DMA instructions per thread and row
  constexpr int SROW = PIXB + 16;               // staging row (see stem.hip)
  static_assert(NDMA == 2 && TP * KC / NT == 2, "the counted waits below assume 2 requests per row and 2 stores per tile and thread");
  __shared__ __attribute__((aligned(16))) char smem[NSLOT * ROW_BYTES + TP * SROW];     // ONE shared object (see stem.hip)
  char* const stage = smem + NSLOT * ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = C == 64 ? wave >> 1 : wave;    // this wave's channels: wc * CT * 16 ..
  const int wp = C == 64 ? wave & 1 : 0;        // this wave's pixels: wp * PT * 16 ..
  const int fr = lane & 15, fc = lane >> 4;
  const int u = blockIdx.x;
  const int xt = u % p.tiles_x, run = (u / p.tiles_x) % p.nruns, img = u / (p.tiles_x * p.nruns);
  const int x0 = xt * TPV, y0 = run * p.run_len, y1 = min(p.H, y0 + p.run_len);
  if (y0 >= y1) return;
  // descriptor base = one pixel before image 0, so that column x0 - 1 has a non-negative offset
  const rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) - PIXB, p.x_bytes + PIXB);

  // this thread's pieces of a row.  A slot is laid out [column block of 16][channel chunk][column in block] x 16 B: the 16 lanes
  // of a ds_read_b128 group read 16 CONSECUTIVE columns (whatever the tap shift, whatever the chunk), which then fall on 16 distinct
  // 16-byte bank slots -- no conflicts at any alignment (the GEMM panels' XOR swizzle is conflict-free only for aligned column
  // groups: 41 % of this kernel's LDS cycles were conflicts with it).  LDS-DMA fills a slot linearly, so the permutation is applied
  // to the SOURCE: piece q = tid + NT i is column (q / (16 KC)) * 16 + (q & 15), chunk (q >> 4) % KC.
  int poff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int q = tid + NT * i;
    const int col = (q / (16 * KC)) * 16 + (q & 15), kc = (q >> 4) % KC;
    poff[i] = (unsigned)(x0 + col - 1) < (unsigned)p.W ? col * PIXB + kc * 16 : OOB;
  }
  auto issue_row = [&](int row) {               // input row `row` of the strip -> slot row & 7 (rows outside the image: zero fill)
    const bool ok = (unsigned)row < (unsigned)p.H;
    const int base = ((img * p.H + (ok ? row : 0)) * p.W + x0) * PIXB;
    char* dst = smem + (row & (NSLOT - 1)) * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) buf_load_lds16(rs, (lptr_t)(dst + (wave * 64 + NT * i) * 16), ok ? poff[i] : OOB, base);
  };

#pragma unroll
  for (int r = -1; r <= DEPTH; ++r) issue_row(y0 + r);
  __builtin_amdgcn_sched_barrier(0);
  // weights [C][3][3][C]: A operand (rows = output channels): lane -> channel wc * CT * 16 + 16 ct + (lane & 15), k = 32 slice + 8 (lane >> 4) ..
  e16x8 wf[CT][KS];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[ct][ks] = *reinterpret_cast<const e16x8*>(p.w + ((wc * CT + ct) * 16 + fr) * (9 * C) + ks * 32 + fc * 8);
  f32x4 bv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(p.bias + (wc * CT + ct) * 16 + fc * 4);
  __builtin_amdgcn_sched_barrier(0);
  // byte offset of this lane's (pixel, chunk fc) inside a row slot under horizontal tap dx, channel slice 0 (slice s: + 4 s chunk windows)
  int coff[PT][3];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int col = (wp * PT + pt) * 16 + fr + dx;       // (columns 64, 65 -- outputs 62, 63, never stored -- read the next slot)
      coff[pt][dx] = (((col >> 4) * KC + fc) * 16 + (col & 15)) * 16;
    }
  const int npix = min(TPV, p.W - x0);
  const float lo = p.relu ? 0.f : -3.0e38f;     // ReLU as a clamp from below (a select, not a branch)

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

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // k-slices (tap, channel slice), software-pipelined: the B fragments of slice k + 1 are requested before the MFMAs of slice k
    // (at two waves per SIMD an exposed LDS round trip per slice is most of the tile otherwise)
    const char* rb[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) rb[dy] = smem + ((y - 1 + dy) & (NSLOT - 1)) * ROW_BYTES;
    e16x8 bf[2][PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) bf[0][pt] = *reinterpret_cast<const e16x8*>(rb[0] + coff[pt][0]);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k + 1 < KS) {
        const int tap = (k + 1) / SPT, s = (k + 1) % SPT, dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bf[(k + 1) & 1][pt] = *reinterpret_cast<const e16x8*>(rb[dy] + coff[pt][dx] + s * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = RMEM_MFMA_16x16x32(wf[ct][k], bf[k & 1][pt], acc[ct][pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        e16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (e16)fmaxf(acc[ct][pt][r] + bv[ct][r], lo);
        *reinterpret_cast<e16x4*>(stage + ((wp * PT + pt) * 16 + fr) * SROW + ((wc * CT + ct) * 16 + fc * 4) * 2) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {               // both stores always issued; pixels past the row end repeat its last pixel
      const int v = tid + NT * j, px = min(v / KC, npix - 1), c16 = v % KC;
      const e16x8 o = *reinterpret_cast<const e16x8*>(stage + px * SROW + c16 * 16);
      *reinterpret_cast<e16x8*>(p.y + (pix0 + px) * C + c16 * 8) = o;
    }
  }
}

}  // namespace

extern "C" int RMEM_API(rmem_conv3x3_direct)(const void* x, int images, int H, int W, int C, const void* w, const float* bias, int relu, void* y,
                                             void* stream) {
  RMEM_REQUIRE(x && w && bias && y && images >= 1 && H >= 1 && W >= 1 && (C == 64 || C == 128), "rmem_conv3x3_direct: bad argument (C must be 64 or 128)");
  RMEM_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)y % 16) == 0,
               "rmem_conv3x3_direct: operands must be 16-byte aligned");
  C3Params p;
  p.x = (const e16*)x; p.w = (const e16*)w; p.bias = bias; p.y = (e16*)y;
  p.images = images; p.H = H; p.W = W; p.relu = relu ? 1 : 0;
  p.tiles_x = (W + TPV - 1) / TPV;
  p.x_bytes = (long)images * H * W * C * 2;
  RMEM_REQUIRE(p.x_bytes + (long)(W + TP + 2) * C * 2 < (1L << 31) - (1L << 22), "rmem_conv3x3_direct: the input exceeds the 2 GB a buffer descriptor addresses");
  // runs of output rows per column strip: enough workgroups for two (C = 64) / one (C = 128) per CU, rows per run as long as that allows
  // (every run re-reads 2 rows and re-loads the weights)
  static const int wgs64 = getenv("RMEM_CONV3_WGS") ? atoi(getenv("RMEM_CONV3_WGS")) : 512;
  static const int wgs128 = getenv("RMEM_CONV3_WGS128") ? atoi(getenv("RMEM_CONV3_WGS128")) : 256;
  const int wgs = C == 64 ? wgs64 : wgs128;
  const long strips = (long)images * p.tiles_x;
  long nruns = (wgs + strips - 1) / strips;
  if (nruns < 1) nruns = 1;
  if (nruns > H) nruns = H;
  p.run_len = (int)((H + nruns - 1) / nruns);
  p.nruns = (H + p.run_len - 1) / p.run_len;
  const long grid = strips * p.nruns;
  RMEM_REQUIRE(grid < (1L << 30), "rmem_conv3x3_direct: too many workgroups");
  if (C == 64) hipLaunchKernelGGL(k_conv3x3_direct<64>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(k_conv3x3_direct<128>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_conv3x3_direct");
}

extern "C" int RMEM_API(rmem_conv3x3_c64_direct)(const void* x, int images, int H, int W, const void* w, const float* bias, void* y, void* stream) {
  return RMEM_API(rmem_conv3x3_direct)(x, images, H, W, 64, w, bias, 1, y, stream);
}
