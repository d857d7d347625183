// ResNet stem: 7x7 stride-2 convolution (3 -> 64 channels) + folded BatchNorm + ReLU for gfx950, without im2col traffic
// (encoders/resnet.py:131-135 conv1 / bn1 / relu; models/aot.py:116-134 is the caller).
//
// The generic row-run GEMM form (gemm_conv.hip) pads the 3 input channels to 8 and walks one filter row per 64-wide k-step:
// K = 7 x 64 = 448 for 147 real products per output, and every k-step drags a 64 x 128 B activation panel through the LDS-DMA
// path although neighbouring outputs share 3/4 of their window -- 214 us per 16 frames, 4 % of the frame.  Here
//   * the image is stored NHWC with FOUR channels (r, g, b, 0) inside a zero border of 3 pixels (and an 8th row / column for the window's zero-weight tail), so
//     one filter row of one output is 8 consecutive pixels = 32 elements = ONE MFMA k-slice, whatever the position (no masks),
//     and K = 8 rows x 32 = 256 (the 8th row / 8th pixel / 4th channel meet zero weights);
//   * a workgroup takes 64 consecutive outputs of one output row: their windows are 8 image rows x 134 pixels = 8.5 KB, copied
//     to LDS ONCE by LDS-DMA; the B fragment of output pixel p and filter row ky is the 64 bytes at pixel 2p of patch row ky,
//     read in place with ds_read_b128 (a conflict-free 16-byte stride) -- the im2col matrix is never formed anywhere;
//   * the weights are the A operand (rows = output channels) and stay in REGISTERS for the life of a persistent workgroup
//     (16 fragments per wave), so the accumulators hold 4 consecutive channels of one pixel per lane and the inner loop reads
//     LDS only for the patch: 2 ds_read_b128 per 4 MFMAs.
// 256 threads = 2 (channel halves) x 2 (pixel halves) waves; the patch is double-buffered across tiles so that the next tile's
// copy is in flight behind the MFMAs and the stores of the current one (counted vmcnt: stores count too on CDNA4).
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
constexpr int PW = 2 * TP + 8;                  // patch width in pixels (2 * 63 + 8 = 134, rounded to a multiple of 4 pixels)
constexpr int PROW = PW * 8;                    // bytes per patch row (4 channels x 2 B per pixel)
constexpr int PCHUNKS = 8 * PROW / 16;          // 16-byte pieces per patch (8 image rows)
constexpr int NDMA = (PCHUNKS + 255) / 256;     // LDS-DMA instructions per thread and patch
constexpr int PATCH_BYTES = NDMA * 256 * 16;    // (the last instruction's surplus lanes read out of range: zero fill, never used)
constexpr int SROW = 144;                       // staging row: 64 channels x 2 B + 16 B (conflict-free 8-byte writes, aligned 16-byte reads)
constexpr int OOB = (int)0x80000000;

struct StemParams {
  const e16* x; const e16* w; const float* bias; e16* y;
  int images, Hp, Wp, Ho, Wo, tiles_x, ntiles;
  long x_bytes;
};

__global__ __launch_bounds__(256) void k_stem7x7s2(StemParams p) {
  // ONE shared object (two patch buffers, then the staging rows): with several, the compiler tags LDS accesses with alias scopes and
  // then drains vmcnt before every ds_read that follows an LDS-DMA request it cannot tell apart from it
  __shared__ __attribute__((aligned(16))) char smem[2 * PATCH_BYTES + TP * SROW];
  char* const stage = smem + 2 * PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;      // channel half, pixel half
  const int fr = lane & 15, fc = lane >> 4;
  const rsrc_t rs = make_rsrc(p.x, p.x_bytes);

  // this thread's pieces of a patch: piece q = tid + 256 i is bytes 16 q .. of the linear patch image (row q / (PROW / 16))
  int poff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int q = tid + 256 * i;
    const int row = q / (PROW / 16), col = q - row * (PROW / 16);
    poff[i] = q < PCHUNKS ? row * p.Wp * 8 + col * 16 : OOB;
  }
  auto tile_base = [&](int t, int& pix0, int& npix) -> int {   // byte offset of the patch's first pixel; first output pixel, count
    const int xt = t % p.tiles_x, r = t / p.tiles_x;           // r = image * Ho + oy
    const int img = r / p.Ho, oy = r - img * p.Ho;
    pix0 = r * p.Wo + xt * TP;
    npix = min(TP, p.Wo - xt * TP);
    return ((img * p.Hp + 2 * oy) * p.Wp + 2 * xt * TP) * 8;
  };
  auto issue = [&](int t, int buf) {
    int pix0, npix;
    const int base = tile_base(t, pix0, npix);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) buf_load_lds16(rs, (lptr_t)(smem + buf * PATCH_BYTES + (wave * 64 + 256 * i) * 16), poff[i], base);
  };

  int t = blockIdx.x;
  if (t >= p.ntiles) return;
  issue(t, 0);
  __builtin_amdgcn_sched_barrier(0);
  // weights: A operand of v_mfma_f32_16x16x32 (rows = output channels): lane -> channel wc * 32 + 16 ct + (lane & 15), k = 32 ky + 8 (lane >> 4) ..
  e16x8 wf[2][8];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ky = 0; ky < 8; ++ky)
      wf[ct][ky] = *reinterpret_cast<const e16x8*>(p.w + (wc * 32 + ct * 16 + fr) * 256 + ky * 32 + fc * 8);
  f32x4 bv[2];                                  // D rows = channels wc * 32 + 16 ct + 4 (lane >> 4) + r
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(p.bias + wc * 32 + ct * 16 + fc * 4);
  __builtin_amdgcn_sched_barrier(0);

  int buf = 0;
  bool first = true;
  for (; t < p.ntiles; t += gridDim.x, buf ^= 1) {
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // the patch of this tile has landed; the 2 stores of the previous tile may be in flight
    first = false;
    // (raw barriers: __syncthreads() is a fence that drains vmcnt, i.e. waits for the NEXT tile's copy and the stores as well)
    __builtin_amdgcn_s_barrier();               // every wave's pieces landed; everyone is done with the other buffer and with `stage`
    // the next tile's patch, requested unconditionally (a request behind a branch makes the compiler drain vmcnt at the join): the
    // last tile of a workgroup requests itself once more into the idle buffer
    const int tn = t + gridDim.x;
    issue(tn < p.ntiles ? tn : t, buf ^ 1);
    int pix0, npix;
    tile_base(t, pix0, npix);

    f32x4 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* pb = smem + buf * PATCH_BYTES + (wp * 32 + fr) * 16 + fc * 16;   // pixel 2 (wp * 32 + 16 pt + fr) + 2 fc of patch row ky
#pragma unroll
    for (int ky = 0; ky < 8; ++ky) {
      e16x8 bf[2];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) bf[pt] = *reinterpret_cast<const e16x8*>(pb + ky * PROW + pt * 256);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = RMEM_MFMA_16x16x32(wf[ct][ky], bf[pt], acc[ct][pt], 0, 0, 0);
    }
    // epilogue: + bias, ReLU, round; lane holds channels 4 fc .. + 3 of pixel fr of each (ct, pt) tile -> staging rows of whole pixels
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        e16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (e16)fmaxf(acc[ct][pt][r] + bv[ct][r], 0.f);
        *reinterpret_cast<e16x4*>(stage + (wp * 32 + pt * 16 + fr) * SROW + (wc * 32 + ct * 16 + fc * 4) * 2) = o;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's staging writes are in LDS
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {               // 64 pixels x 128 B, contiguous in y: 512 vectors of 16 B
      // both stores are always issued (the counted wait above relies on it): pixels past the end of the output row repeat the
      // row's last pixel, same bytes to the same address
      const int v = tid + 256 * j, px = min(v >> 3, npix - 1), c16 = v & 7;
      const e16x8 o = *reinterpret_cast<const e16x8*>(stage + px * SROW + c16 * 16);
      const long dst = (long)(pix0 + px) * 64 + c16 * 8;
      *reinterpret_cast<e16x8*>(p.y + dst) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Stem + 3x3 stride-2 max-pool in one pass (encoders/resnet.py:131-136: conv1, bn1, relu, maxpool): the 64-channel map at half
// resolution (210 MB per 16 frames) is never written.  A workgroup owns a strip of 27 pooled columns = 55 conv columns (61 are
// computed: their windows are then 128 input pixels = ONE LDS-DMA instruction per input row) and walks DOWN a run of pooled rows,
// two conv rows per iteration:
//   * input rows live in a ring of 32 one-KB slots; an iteration needs 10 of them, 4 are new: each of the four waves requests
//     exactly one row per iteration, five iterations ahead (counted vmcnt: one request + one store per wave and iteration);
//   * conv rows (+ bias, ReLU, rounded as stem7x7s2 stores them; columns / rows outside the map as zeros, which is what the pool's
//     padding amounts to after a ReLU) go to a ring of four row buffers in LDS;
//   * iteration k computes conv rows 2k - 1 and 2k and emits pooled row k - 1 = max over conv rows 2k - 3 .. 2k - 1 x 3 columns.
// Same conv arithmetic as k_stem7x7s2 (weights in registers, fragments read in place), an exact max: bit-identical to
// rmem_stem7x7s2 followed by rmem_maxpool3x3s2_nhwc (tests/test_hip_ops.py::test_stem_pool_fused).
constexpr int SP_PV = 27;                       // pooled columns per strip
constexpr int SP_NIN = 32;                      // input row slots (1 KB: 128 pixels x 4 channels)
constexpr int SP_D = 5;                         // iterations of look-ahead
constexpr int SP_CROW = TP * SROW;              // one conv row buffer
static_assert(4 * SP_D + 10 <= SP_NIN, "input row ring");

struct StemPoolParams {
  const e16* x; const e16* w; const float* bias; e16* y;
  int images, Hp, Wp, H2, W2, HP, WP, strips, nruns, run_len;
  long x_bytes;
};

__global__ __launch_bounds__(256) void k_stem_pool(StemPoolParams p) {
  __shared__ __attribute__((aligned(16))) char smem[SP_NIN * 1024 + 4 * SP_CROW];       // ONE shared object
  char* const crows = smem + SP_NIN * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;      // channel half, pixel half
  const int fr = lane & 15, fc = lane >> 4;
  const int u = blockIdx.x;
  const int xs = u % p.strips, run = (u / p.strips) % p.nruns, img = u / (p.strips * p.nruns);
  const int px0 = xs * SP_PV;                   // first pooled column
  const int c0 = 2 * px0 - 1;                   // first conv column (computed column j = conv column c0 + j)
  const int k0 = run * p.run_len, k1 = min(p.HP, k0 + p.run_len);      // pooled rows [k0, k1): iterations k0 .. k1
  if (k0 >= k1) return;
  // descriptor base 16 B before the buffer: the first strip's row pieces start at padded column -2
  const rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(p.x) - 16, p.x_bytes + 16);
  // this lane's 16-byte piece of a row: padded columns 2 c0 + 2 lane, + 1
  const int pc = 2 * c0 + 2 * lane;
  const int poff = (pc >= 0 && pc + 1 < p.Wp) ? lane * 16 : OOB;
  auto issue_row = [&](int prow) {              // padded input row prow -> slot prow & 31 (rows outside the buffer: zeros)
    const bool ok = (unsigned)prow < (unsigned)p.Hp;
    const int base = ((img * p.Hp + (ok ? prow : 0)) * p.Wp + 2 * c0) * 8 + 16;
    buf_load_lds16(rs, (lptr_t)(smem + (prow & (SP_NIN - 1)) * 1024), ok ? poff : OOB, base);
  };
  // prologue: the rows of the first SP_D iterations (4 k0 - 2 .. 4 (k0 + SP_D - 1) + 7), row r by wave r & 3
  for (int r = 4 * k0 - 2; r <= 4 * (k0 + SP_D - 1) + 7; ++r)
    if ((r & 3) == wave) issue_row(r);
  __builtin_amdgcn_sched_barrier(0);
  e16x8 wf[2][8];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int ky = 0; ky < 8; ++ky)
      wf[ct][ky] = *reinterpret_cast<const e16x8*>(p.w + (wc * 32 + ct * 16 + fr) * 256 + ky * 32 + fc * 8);
  f32x4 bv[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(p.bias + wc * 32 + ct * 16 + fc * 4);
  __builtin_amdgcn_sched_barrier(0);
  // which of this lane's two output pixels are real conv columns (the others are stored as zeros)
  bool colok[2];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) colok[pt] = (unsigned)(c0 + wp * 32 + pt * 16 + fr) < (unsigned)p.W2;
  // the pooled vector this thread emits: pooled column px0 + pp, channels 8 c16 ..; lanes past the strip repeat its last column
  const int npool = min(SP_PV, p.WP - px0);
  const int pp = min(tid >> 3, npool - 1), c16 = tid & 7;

  for (int k = k0; k <= k1; ++k) {
    if (k == k0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");      // this wave's row of iteration k has landed (requested SP_D iterations ago)
    __builtin_amdgcn_s_barrier();
    issue_row(4 * (k + SP_D) + 4 + wave);       // one row per wave, unconditionally
#pragma unroll
    for (int h = 0; h < 2; ++h) {               // conv rows 2k - 1, 2k
      const int cr = 2 * k - 1 + h;
      f32x4 acc[2][2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 8; ++ky) {
        const char* rb = smem + ((2 * cr + ky) & (SP_NIN - 1)) * 1024 + (wp * 32 + fr) * 16 + fc * 16;
        e16x8 bf[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) bf[pt] = *reinterpret_cast<const e16x8*>(rb + pt * 256);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = RMEM_MFMA_16x16x32(wf[ct][ky], bf[pt], acc[ct][pt], 0, 0, 0);
      }
      const bool rowok = (unsigned)cr < (unsigned)p.H2;
      char* cb = crows + ((cr + 1) & 3) * SP_CROW;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          e16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (rowok && colok[pt]) ? (e16)fmaxf(acc[ct][pt][r] + bv[ct][r], 0.f) : (e16)0.f;
          *reinterpret_cast<e16x4*>(cb + (wp * 32 + pt * 16 + fr) * SROW + (wc * 32 + ct * 16 + fc * 4) * 2) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      // pooled row k - 1 (the first iteration of a run has nothing to emit yet: it stores to the slot of row k0, which the next
      // iteration overwrites -- the store is issued regardless, the counted wait above relies on it)
      const int py = max(k - 1, k0);
      float m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const char* cb = crows + ((2 * py - 1 + r + 1) & 3) * SP_CROW + (2 * pp) * SROW + c16 * 16;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const e16x8 d = *reinterpret_cast<const e16x8*>(cb + dx * SROW);
#pragma unroll
          for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)d[j]);
        }
      }
      e16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (e16)m[j];
      *reinterpret_cast<e16x8*>(p.y + (((long)img * p.HP + py) * p.WP + px0 + pp) * 64 + c16 * 8) = o;
    }
  }
}

// fp32 planar frames named by a device table -> NHWC4 e16 inside the zero border the stem kernel expects (the border is never written)
__global__ __launch_bounds__(256) void k_image_ptrs_to_nhwc4p(const float* const* imgs, e16* out, int H, int W, int Hp, int Wp) {
  const long n = (long)H * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* img = imgs[blockIdx.y];
  const int y = (int)(i / W), x = (int)(i - (long)y * W);
  e16x4 o = {(e16)img[i], (e16)img[n + i], (e16)img[2 * n + i], (e16)0.f};
  *reinterpret_cast<e16x4*>(out + (((long)blockIdx.y * Hp + y + 3) * Wp + x + 3) * 4) = o;
}

}  // namespace

#ifndef RMEM_F16
extern "C" int rmem_stem_padded_size(int H, int W, int* Hp, int* Wp) {
  if (H <= 0 || W <= 0 || !Hp || !Wp) return -1;
  const int Wo = (W + 6 - 7) / 2 + 1;
  (void)Wo;
  *Hp = H + 7;                                  // 3 + 3 border rows and the 8th (zero-weight) filter row of the last output row
  *Wp = (W + 7 + 1) & ~1;                       // 3 + 3 border columns and the 8th (zero-weight) pixel of the last output; even: 16-byte rows
  return 0;
}
#endif

extern "C" int RMEM_API(rmem_image_ptrs_to_nhwc4p)(const float* const* img_ptrs, void* out, int images, int H, int W, void* stream) {
  RMEM_REQUIRE(img_ptrs && out && images >= 1 && H > 0 && W > 0, "rmem_image_ptrs_to_nhwc4p: bad argument");
  int Hp, Wp;
  rmem_stem_padded_size(H, W, &Hp, &Wp);
  hipLaunchKernelGGL(k_image_ptrs_to_nhwc4p, dim3((unsigned)(((long)H * W + 255) / 256), images), dim3(256), 0, (hipStream_t)stream, img_ptrs,
                     (e16*)out, H, W, Hp, Wp);
  return rmem_check_launch("rmem_image_ptrs_to_nhwc4p");
}

extern "C" int RMEM_API(rmem_stem7x7s2_pool)(const void* x_padded, int images, int H, int W, const void* w, const float* bias, void* y_pooled,
                                             void* stream) {
  RMEM_REQUIRE(x_padded && w && bias && y_pooled && images >= 1 && H >= 7 && W >= 7, "rmem_stem7x7s2_pool: bad argument");
  RMEM_REQUIRE(((uintptr_t)x_padded % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)y_pooled % 16) == 0,
               "rmem_stem7x7s2_pool: operands must be 16-byte aligned");
  StemPoolParams p;
  p.x = (const e16*)x_padded; p.w = (const e16*)w; p.bias = bias; p.y = (e16*)y_pooled;
  p.images = images;
  rmem_stem_padded_size(H, W, &p.Hp, &p.Wp);
  p.H2 = (H - 1) / 2 + 1; p.W2 = (W - 1) / 2 + 1;
  p.HP = (p.H2 - 1) / 2 + 1; p.WP = (p.W2 - 1) / 2 + 1;
  p.strips = (p.WP + SP_PV - 1) / SP_PV;
  p.x_bytes = (long)images * p.Hp * p.Wp * 8;
  RMEM_REQUIRE(p.x_bytes < (1L << 31) - (1L << 22), "rmem_stem7x7s2_pool: the padded frames exceed the 2 GB a buffer descriptor addresses");
  static const int wgs = getenv("RMEM_STEM_POOL_WGS") ? atoi(getenv("RMEM_STEM_POOL_WGS")) : 1024;     // 91.5 us at 640, 81 at 1024, 88 at 2048 (16 frames)
  const long cols = (long)images * p.strips;
  long nruns = (wgs + cols - 1) / cols;
  if (nruns < 1) nruns = 1;
  if (nruns > p.HP) nruns = p.HP;
  p.run_len = (int)((p.HP + nruns - 1) / nruns);
  p.nruns = (p.HP + p.run_len - 1) / p.run_len;
  hipLaunchKernelGGL(k_stem_pool, dim3((unsigned)(cols * p.nruns)), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_stem7x7s2_pool");
}

extern "C" int RMEM_API(rmem_stem7x7s2)(const void* x_padded, int images, int H, int W, const void* w, const float* bias, void* y, void* stream) {
  RMEM_REQUIRE(x_padded && w && bias && y && images >= 1 && H >= 7 && W >= 7, "rmem_stem7x7s2: bad argument");
  RMEM_REQUIRE(((uintptr_t)x_padded % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)bias % 16) == 0 && ((uintptr_t)y % 16) == 0,
               "rmem_stem7x7s2: operands must be 16-byte aligned");
  StemParams p;
  p.x = (const e16*)x_padded; p.w = (const e16*)w; p.bias = bias; p.y = (e16*)y;
  p.images = images;
  rmem_stem_padded_size(H, W, &p.Hp, &p.Wp);
  p.Ho = (H + 6 - 7) / 2 + 1; p.Wo = (W + 6 - 7) / 2 + 1;
  p.tiles_x = (p.Wo + TP - 1) / TP;
  const long nt = (long)images * p.Ho * p.tiles_x;
  p.x_bytes = (long)images * p.Hp * p.Wp * 8;
  RMEM_REQUIRE(p.x_bytes < (1L << 31) - (1L << 22) && nt < (1L << 30), "rmem_stem7x7s2: the padded frames exceed the 2 GB a buffer descriptor addresses");
  p.ntiles = (int)nt;
  static const int wgs = getenv("RMEM_STEM_WGS") ? atoi(getenv("RMEM_STEM_WGS")) : 1024;      // persistent workgroups (4 per CU)
  const unsigned grid = (unsigned)(nt < wgs ? nt : wgs);
  hipLaunchKernelGGL(k_stem7x7s2, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_stem7x7s2");
}
