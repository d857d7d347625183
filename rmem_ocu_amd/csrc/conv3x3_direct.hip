// 3x3 stride-1 convolution with 64 input and 64 output channels (+ folded BatchNorm, ReLU) read in place from an LDS patch: the second
// conv of the ResNet layer-1 bottlenecks (encoders/resnet.py:52-56), 121 x 213 pixels x 16 frames per launch in the bench.
//
// Same idea as stem.hip.  The implicit-GEMM form (gemm_conv.hip) copies every activation byte through LDS-DMA once per filter tap
// (9 x) plus a weight panel per k-step, and is bound by that copy.  Here a workgroup takes 64 consecutive outputs of an output row:
//   * the 3 input rows x 66 pixels x 64 channels they read (25 KB) go to LDS ONCE, with zero fill outside the image (per-lane
//     out-of-range offsets) and the 16-byte channel chunks XOR-swizzled at the SOURCE ((pixel >> 1) & 7, as the GEMM panels), so
//     that the B fragment of output pixel p, tap (dy, dx), channel slice s is one conflict-free ds_read_b128 at patch pixel
//     dy * 66 + p + dx;
//   * the whole weight tensor of this wave's 32 output channels (2 x 18 fragments, K = 9 taps x 64) lives in REGISTERS for the life
//     of a persistent workgroup: the weights are the A operand, so a lane's accumulators are 4 consecutive channels of a pixel.
// The k order (tap-major, then input channel) and the epilogue arithmetic are those of rmem_conv2d_nhwc: results are bit-identical
// (tests/test_hip_ops.py::test_conv3x3_c64_direct).  256 threads = 2 (channel halves) x 2 (pixel halves) waves, patch double-buffered
// across tiles behind a counted vmcnt, one shared object and raw barriers (see stem.hip for why).
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

constexpr int TP = 64;                          // output pixels per tile
constexpr int PWP = TP + 2;                     // patch width in pixels
constexpr int PPIX = 3 * PWP;                   // patch pixels (128 B each)
constexpr int PCHUNKS = PPIX * 8;               // 16-byte pieces
constexpr int NDMA = (PCHUNKS + 255) / 256;     // LDS-DMA instructions per thread and patch
constexpr int PATCH_BYTES = NDMA * 256 * 16;
constexpr int SROW = 144;                       // staging row (see stem.hip)
constexpr int OOB = (int)0x80000000;

struct C3Params {
  const e16* x; const e16* w; const float* bias; e16* y;
  int images, H, W, tiles_x, ntiles;
  long x_bytes;
};

__global__ __launch_bounds__(256) void k_conv3x3_c64(C3Params p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * PATCH_BYTES + TP * SROW];
  char* const stage = smem + 2 * PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;      // channel half, pixel half
  const int fr = lane & 15, fc = lane >> 4;
  // descriptor base = the (virtual) pixel (-1, -1) of image 0, so that every in-image offset below is non-negative
  const long shift = ((long)p.W + 1) * 128;
  const rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) - shift, p.x_bytes + shift);

  // this thread's pieces of a patch: piece q = tid + 256 i fills LDS bytes 16 q ..: patch pixel q >> 3 = (row, column), physical chunk
  // q & 7, which holds the logical chunk (q & 7) ^ ((pixel >> 1) & 7)
  int prow[NDMA], pcol[NDMA], poff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int q = tid + 256 * i;
    const int pix = q >> 3;
    prow[i] = pix / PWP;
    pcol[i] = pix - prow[i] * PWP;
    const int kc = (q & 7) ^ ((pix >> 1) & 7);
    poff[i] = q < PCHUNKS ? (prow[i] * p.W + pcol[i]) * 128 + kc * 16 : OOB;
    if (q >= PCHUNKS) prow[i] = 1 << 20;        // never inside an image
  }
  auto decode = [&](int t, int& img, int& y, int& x0) {
    const int xt = t % p.tiles_x, r = t / p.tiles_x;
    img = r / p.H; y = r - img * p.H; x0 = xt * TP;
  };
  auto issue = [&](int t, int buf) {
    int img, y, x0;
    decode(t, img, y, x0);
    const int base = ((img * p.H + y) * p.W + x0) * 128;      // soffset: the tile's output pixel (y, x0) relative to the shifted base
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const bool ok = (unsigned)(y + prow[i] - 1) < (unsigned)p.H && (unsigned)(x0 + pcol[i] - 1) < (unsigned)p.W;
      buf_load_lds16(rs, (lptr_t)(smem + buf * PATCH_BYTES + (wave * 64 + 256 * i) * 16), ok ? poff[i] : OOB, base);
    }
  };

  int t = blockIdx.x;
  if (t >= p.ntiles) return;
  issue(t, 0);
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

  int buf = 0;
  bool first = true;
  for (; t < p.ntiles; t += gridDim.x, buf ^= 1) {
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // this tile's patch has landed; the previous tile's 2 stores may be in flight
    first = false;
    __builtin_amdgcn_s_barrier();
    const int tn = t + gridDim.x;
    issue(tn < p.ntiles ? tn : t, buf ^ 1);     // unconditionally (see stem.hip)
    int img, y, x0;
    decode(t, img, y, x0);
    const int npix = min(TP, p.W - x0);
    const long pix0 = ((long)img * p.H + y) * p.W + x0;

    f32x4 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* pb = smem + buf * PATCH_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        e16x8 bf[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const int pp = dy * PWP + wp * 32 + pt * 16 + fr + dx;          // patch pixel of this lane's output pixel under the tap
          bf[pt] = *reinterpret_cast<const e16x8*>(pb + pp * 128 + (((4 * s + fc) ^ ((pp >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = RMEM_MFMA_16x16x32(wf[ct][2 * tap + s], bf[pt], acc[ct][pt], 0, 0, 0);
      }
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
  p.tiles_x = (W + TP - 1) / TP;
  const long nt = (long)images * H * p.tiles_x;
  p.x_bytes = (long)images * H * W * 128;
  RMEM_REQUIRE(p.x_bytes + ((long)W + 1) * 128 + (long)(2 * W + TP + 2) * 128 < (1L << 31) - (1L << 22) && nt < (1L << 30),
               "rmem_conv3x3_c64_direct: the input exceeds the 2 GB a buffer descriptor addresses");
  p.ntiles = (int)nt;
  static const int wgs = getenv("RMEM_CONV3_WGS") ? atoi(getenv("RMEM_CONV3_WGS")) : 512;       // persistent workgroups (2 per CU)
  const unsigned grid = (unsigned)(nt < wgs ? nt : wgs);
  hipLaunchKernelGGL(k_conv3x3_c64, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_conv3x3_c64_direct");
}
