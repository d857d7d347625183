// Layout / resampling kernels around the encoder, the FPN head and the engine
// (all HBM-bound, one pass each):
//   NCHW fp32 image -> NHWC8 e16                      (input of encoders/resnet.py:179)
//   3x3 stride-2 max-pool, NHWC                        (encoders/resnet.py:105, 182)
//   bilinear resize NHWC e16 (align_corners on/off)   (decoders/fpn.py:49-52, 57-60)
//   logits: mask unused ids, bilinear to output size, NCHW fp32 + argmax labels
//                                                      (engines/aot_engine.py:450-463; managers/evaluator.py:430-441)
//   label map -> nearest resize -> one-hot(+ignore) NHWC16 e16
//                                                      (utils/image.py:69-74; aot_engine.py:208-224; evaluator.py:518-522)
//   eviction scores: sum_q mass[q, t] * (1 - softmax(bilinear(logits))[0])
//                                                      (aot_engine.py:355-362; layers/transformer.py:341-351)
#include "common.h"
#include "../../include/rmem.h"

namespace {

__global__ __launch_bounds__(256) void k_image_to_nhwc8(const float* img, e16* out, int H, int W) {
  const long n = (long)H * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  img += (long)blockIdx.y * 3 * n;            // blockIdx.y = image of a batch
  out += (long)blockIdx.y * 8 * n;
  e16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
  o[0] = (e16)img[i];
  o[1] = (e16)img[n + i];
  o[2] = (e16)img[2 * n + i];
  reinterpret_cast<e16x8*>(out)[i] = o;
}

// the same with one source pointer per image (a device table): frames stay where the caller keeps its clips
__global__ __launch_bounds__(256) void k_image_ptrs_to_nhwc8(const float* const* imgs, e16* out, int H, int W) {
  const long n = (long)H * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* img = imgs[blockIdx.y];        // blockIdx.y = image of a batch
  out += (long)blockIdx.y * 8 * n;
  e16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
  o[0] = (e16)img[i];
  o[1] = (e16)img[n + i];
  o[2] = (e16)img[2 * n + i];
  reinterpret_cast<e16x8*>(out)[i] = o;
}

__global__ __launch_bounds__(256) void k_maxpool3s2(const e16* x, e16* y, int H, int W, int C, int Ho, int Wo) {
  const int vpr = C / 8;
  const long total = (long)Ho * Wo * vpr;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c0 = (int)(i % vpr) * 8;
  const int pix = (int)(i / vpr);
  const int oy = pix / Wo, ox = pix - oy * Wo;
  x += (long)blockIdx.y * H * W * C;          // blockIdx.y = image of a batch
  y += (long)blockIdx.y * total * 8;
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -3.0e38f;
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = oy * 2 - 1 + dy;
    if ((unsigned)yy >= (unsigned)H) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = ox * 2 - 1 + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      const e16x8 d = *reinterpret_cast<const e16x8*>(x + ((long)yy * W + xx) * C + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)d[j]);
    }
  }
  e16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (e16)m[j];
  reinterpret_cast<e16x8*>(y)[i] = o;
}

__device__ __forceinline__ void src_coord(int d, int in, int out, int align, int& i0, int& i1, float& w1) {
  rmem_src_coord(d, in, out, align, i0, i1, w1);
}

__global__ __launch_bounds__(256) void k_bilinear_nhwc(const e16* x, e16* y, int Hi, int Wi, int Ho, int Wo, int C, int align) {
  const int vpr = C / 8;
  const long total = (long)Ho * Wo * vpr;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  x += (long)blockIdx.y * Hi * Wi * C;            // blockIdx.y = image of a batch
  y += (long)blockIdx.y * Ho * Wo * C;
  const int c0 = (int)(i % vpr) * 8;
  const int pix = (int)(i / vpr);
  const int oy = pix / Wo, ox = pix - oy * Wo;
  int y0, y1, x0, x1; float wy, wx;
  src_coord(oy, Hi, Ho, align, y0, y1, wy);
  src_coord(ox, Wi, Wo, align, x0, x1, wx);
  const e16x8 a = *reinterpret_cast<const e16x8*>(x + ((long)y0 * Wi + x0) * C + c0);
  const e16x8 b = *reinterpret_cast<const e16x8*>(x + ((long)y0 * Wi + x1) * C + c0);
  const e16x8 c = *reinterpret_cast<const e16x8*>(x + ((long)y1 * Wi + x0) * C + c0);
  const e16x8 d = *reinterpret_cast<const e16x8*>(x + ((long)y1 * Wi + x1) * C + c0);
  e16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (e16)rmem_bilerp((float)a[j], (float)b[j], (float)c[j], (float)d[j], wx, wy);
  reinterpret_cast<e16x8*>(y)[i] = o;
}

// labels only from 16-float logit rows (the per-frame fast path of k_logits_post): a thread makes 4 consecutive output pixels of
// one row.  At the path's ~4 x upsampling they share their two source columns most of the time, so the 4 x 16-float taps are
// reloaded only when the source column moves (a third of the cached loads of one-pixel-per-thread); same arithmetic per pixel.
__global__ __launch_bounds__(256) void k_logits_labels4(const float* lg, int nc, int keep, int Hi, int Wi, int Ho, int Wo, int align,
                                                        uint8_t* label, float* label_f32) {
  const int wq = (Wo + 3) / 4;
  const long item = (long)blockIdx.x * 256 + threadIdx.x;
  if (item >= (long)Ho * wq) return;
  const long total = (long)Ho * Wo;
  lg += (long)blockIdx.y * Hi * Wi * 16;
  if (label) label += (long)blockIdx.y * total;
  if (label_f32) label_f32 += (long)blockIdx.y * total;
  const int oy = (int)(item / wq), oxb = (int)(item - (long)oy * wq) * 4;
  int y0, y1; float wy;
  src_coord(oy, Hi, Ho, align, y0, y1, wy);
  float va[16], vb[16], vc[16], vd[16];
  int cx0 = -1, cx1 = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ox = oxb + k;
    if (ox >= Wo) break;
    int x0, x1; float wx;
    src_coord(ox, Wi, Wo, align, x0, x1, wx);
    if (x0 != cx0 || x1 != cx1) {
      const float* a = lg + ((long)y0 * Wi + x0) * 16;
      const float* b = lg + ((long)y0 * Wi + x1) * 16;
      const float* c = lg + ((long)y1 * Wi + x0) * 16;
      const float* d = lg + ((long)y1 * Wi + x1) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ta = *reinterpret_cast<const f32x4*>(a + 4 * q), tb = *reinterpret_cast<const f32x4*>(b + 4 * q);
        const f32x4 tc = *reinterpret_cast<const f32x4*>(c + 4 * q), td = *reinterpret_cast<const f32x4*>(d + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[4 * q + j] = ta[j]; vb[4 * q + j] = tb[j]; vc[4 * q + j] = tc[j]; vd[4 * q + j] = td[j]; }
      }
      cx0 = x0; cx1 = x1;
    }
    float best = -3.0e38f; int arg = 0;
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      if (ch < nc) {
        float v;
        if (ch > keep) v = -1.0e10f;
        else {
          const float top = va[ch] * (1.f - wx) + vb[ch] * wx;
          const float bot = vc[ch] * (1.f - wx) + vd[ch] * wx;
          v = top * (1.f - wy) + bot * wy;
        }
        if (v > best) { best = v; arg = ch; }
      }
    }
    const long i = (long)oy * Wo + ox;
    if (label) label[i] = (uint8_t)arg;
    if (label_f32) label_f32[i] = (float)arg;
  }
}

// the same for >= 2 x upsampling (the path's 1/4-resolution logits): a workgroup makes a 16 x 64 tile of output pixels; the <= 10 x
// 34 source pixels under it go to LDS once (16 floats each) and the four taps of every output pixel are LDS reads -- per output
// pixel the one-pixel-per-thread form issued 256 bytes of cached global loads.  Same arithmetic per pixel.
constexpr int LT_H = 16, LT_W = 64, LT_SRC = 12 * 40;
__global__ __launch_bounds__(256) void k_logits_labels_tile(const float* lg, int nc, int keep, int Hi, int Wi, int Ho, int Wo, int align,
                                                            uint8_t* label, float* label_f32) {
  __shared__ __attribute__((aligned(16))) float src[LT_SRC * 16];
  const int tiles_x = (Wo + LT_W - 1) / LT_W;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const long total = (long)Ho * Wo;
  lg += (long)blockIdx.y * Hi * Wi * 16;
  if (label) label += (long)blockIdx.y * total;
  if (label_f32) label_f32 += (long)blockIdx.y * total;
  const int oy0 = ty * LT_H, ox0 = tx * LT_W;
  const int oy1 = min(oy0 + LT_H, Ho) - 1, ox1 = min(ox0 + LT_W, Wo) - 1;
  int ylo, yhi, xlo, xhi, t0, t1; float tw;
  src_coord(oy0, Hi, Ho, align, ylo, t1, tw);
  src_coord(oy1, Hi, Ho, align, t0, yhi, tw);
  src_coord(ox0, Wi, Wo, align, xlo, t1, tw);
  src_coord(ox1, Wi, Wo, align, t0, xhi, tw);
  const int ncols = xhi - xlo + 1, nrows = yhi - ylo + 1;      // <= LT_SRC source pixels: lt_tiles_fit() on the host checked every tile
  for (int i = threadIdx.x; i < nrows * ncols * 4; i += 256) {
    const int pix = i >> 2, q = i & 3;
    const int r = pix / ncols, c = pix - r * ncols;
    *reinterpret_cast<f32x4*>(&src[pix * 16 + 4 * q]) = *reinterpret_cast<const f32x4*>(lg + ((long)(ylo + r) * Wi + xlo + c) * 16 + 4 * q);
  }
  __syncthreads();
  const int oy = oy0 + (threadIdx.x >> 4), oxb = ox0 + (threadIdx.x & 15) * 4;
  if (oy >= Ho) return;
  int y0, y1; float wy;
  src_coord(oy, Hi, Ho, align, y0, y1, wy);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ox = oxb + k;
    if (ox >= Wo) break;
    int x0, x1; float wx;
    src_coord(ox, Wi, Wo, align, x0, x1, wx);
    const float* a = &src[((y0 - ylo) * ncols + (x0 - xlo)) * 16];
    const float* b = &src[((y0 - ylo) * ncols + (x1 - xlo)) * 16];
    const float* c = &src[((y1 - ylo) * ncols + (x0 - xlo)) * 16];
    const float* d = &src[((y1 - ylo) * ncols + (x1 - xlo)) * 16];
    float best = -3.0e38f; int arg = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 ta = *reinterpret_cast<const f32x4*>(a + 4 * q), tb = *reinterpret_cast<const f32x4*>(b + 4 * q);
      const f32x4 tc = *reinterpret_cast<const f32x4*>(c + 4 * q), td = *reinterpret_cast<const f32x4*>(d + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = 4 * q + j;
        if (ch < nc) {
          float v;
          if (ch > keep) v = -1.0e10f;
          else {
            const float top = ta[j] * (1.f - wx) + tb[j] * wx;
            const float bot = tc[j] * (1.f - wx) + td[j] * wx;
            v = top * (1.f - wy) + bot * wy;
          }
          if (v > best) { best = v; arg = ch; }
        }
      }
    }
    const long i = (long)oy * Wo + ox;
    if (label) label[i] = (uint8_t)arg;
    if (label_f32) label_f32[i] = (float)arg;
  }
}

// host side of k_logits_labels_tile: the source pixels under EVERY output tile (same rmem_src_coord as the kernel) fit its LDS tile
bool lt_tiles_fit(int Hi, int Wi, int Ho, int Wo, int align) {
  int rows = 0, cols = 0, lo, hi, t; float w;
  for (int o0 = 0; o0 < Ho; o0 += LT_H) {
    rmem_src_coord(o0, Hi, Ho, align, lo, t, w);
    rmem_src_coord((o0 + LT_H < Ho ? o0 + LT_H : Ho) - 1, Hi, Ho, align, t, hi, w);
    rows = hi - lo + 1 > rows ? hi - lo + 1 : rows;
  }
  for (int o0 = 0; o0 < Wo; o0 += LT_W) {
    rmem_src_coord(o0, Wi, Wo, align, lo, t, w);
    rmem_src_coord((o0 + LT_W < Wo ? o0 + LT_W : Wo) - 1, Wi, Wo, align, t, hi, w);
    cols = hi - lo + 1 > cols ? hi - lo + 1 : cols;
  }
  return rows * cols <= LT_SRC;
}

// logits NHWC fp32 [Hi][Wi][ldl] -> NCHW fp32 [nc][Ho][Wo] (optional) + argmax labels (optional)
__global__ __launch_bounds__(256) void k_logits_post(const float* lg, int ldl, int nc, int keep, int Hi, int Wi, int Ho, int Wo,
                                                     int align, float* out, uint8_t* label, float* label_f32) {
  const long total = (long)Ho * Wo;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  lg += (long)blockIdx.y * Hi * Wi * ldl;     // blockIdx.y = image of a batch (labels only: out is per image)
  if (label) label += (long)blockIdx.y * total;
  if (label_f32) label_f32 += (long)blockIdx.y * total;
  const int oy = (int)(i / Wo), ox = (int)(i - (long)oy * Wo);
  int y0, y1, x0, x1; float wy, wx;
  src_coord(oy, Hi, Ho, align, y0, y1, wy);
  src_coord(ox, Wi, Wo, align, x0, x1, wx);
  const float* a = lg + ((long)y0 * Wi + x0) * ldl;
  const float* b = lg + ((long)y0 * Wi + x1) * ldl;
  const float* c = lg + ((long)y1 * Wi + x0) * ldl;
  const float* d = lg + ((long)y1 * Wi + x1) * ldl;
  float best = -3.0e38f; int arg = 0;
  if (ldl == 16 && !out) {
    // labels only (the per-frame fast path): the 4 taps as 16-float rows in registers, fully unrolled; same arithmetic
    float va[16], vb[16], vc[16], vd[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 ta = *reinterpret_cast<const f32x4*>(a + 4 * q), tb = *reinterpret_cast<const f32x4*>(b + 4 * q);
      const f32x4 tc = *reinterpret_cast<const f32x4*>(c + 4 * q), td = *reinterpret_cast<const f32x4*>(d + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) { va[4 * q + j] = ta[j]; vb[4 * q + j] = tb[j]; vc[4 * q + j] = tc[j]; vd[4 * q + j] = td[j]; }
    }
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      if (ch < nc) {
        float v;
        if (ch > keep) v = -1.0e10f;
        else {
          const float top = va[ch] * (1.f - wx) + vb[ch] * wx;
          const float bot = vc[ch] * (1.f - wx) + vd[ch] * wx;
          v = top * (1.f - wy) + bot * wy;
        }
        if (v > best) { best = v; arg = ch; }
      }
    }
    if (label) label[i] = (uint8_t)arg;
    if (label_f32) label_f32[i] = (float)arg;
    return;
  }
  for (int ch = 0; ch < nc; ++ch) {
    float v;
    if (ch > keep) v = -1.0e10f;  // aot_engine.py:451-453
    else {
      const float top = a[ch] * (1.f - wx) + b[ch] * wx;
      const float bot = c[ch] * (1.f - wx) + d[ch] * wx;
      v = top * (1.f - wy) + bot * wy;
    }
    if (out) out[(long)ch * total + i] = v;
    if (v > best) { best = v; arg = ch; }
  }
  if (label) label[i] = (uint8_t)arg;
  if (label_f32) label_f32[i] = (float)arg;
}

// planes [C][Hs][Ws] fp32 -> (optionally mirrored along W first: utils/image.py:109-113 flip_tensor(dim 3)) -> nearest resize
// (F.interpolate(mode='nearest') index rule) to [C][Hd][Wd]: the order of managers/evaluator.py:490-522 (flip_tensor(pred_label, 3),
// then F.interpolate to the engine's input size) -- the flipped-augmentation engines of the evaluator get their frames and label
// maps through this instead of ATen flip / interpolate
__global__ __launch_bounds__(256) void k_resize_nearest_flip(const float* src, int Hs, int Ws, float* dst, int Hd, int Wd, int flip) {
  const long total = (long)Hd * Wd;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  src += (long)blockIdx.y * Hs * Ws;          // blockIdx.y = plane
  dst += (long)blockIdx.y * total;
  const int y = (int)(i / Wd), x = (int)(i - (long)y * Wd);
  const int sy = min((int)floorf((float)y * ((float)Hs / (float)Hd)), Hs - 1);
  const int sx = min((int)floorf((float)x * ((float)Ws / (float)Wd)), Ws - 1);
  dst[i] = src[(long)sy * Ws + (flip ? Ws - 1 - sx : sx)];      // resize(flip(src)): column sx of the mirrored source
}

// label [Hs][Ws] (uint8 or fp32) -> nearest resize to [Hd][Wd] -> one-hot NHWC16 e16
// channels 0..ncls-1 one-hot (channel 0 cleared where label == 255), channel ncls = ignore (label == 255)
__global__ __launch_bounds__(256) void k_label_onehot(const void* lab, int lab_f32, int Hs, int Ws, int Hd, int Wd, int ncls, e16* out) {
  const long total = (long)Hd * Wd;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  lab = reinterpret_cast<const char*>(lab) + (long)blockIdx.y * Hs * Ws * (lab_f32 ? 4 : 1);   // blockIdx.y = image of a batch
  out += (long)blockIdx.y * total * 16;
  const int y = (int)(i / Wd), x = (int)(i - (long)y * Wd);
  const int sy = min((int)floorf((float)y * ((float)Hs / (float)Hd)), Hs - 1);   // F.interpolate(mode='nearest')
  const int sx = min((int)floorf((float)x * ((float)Ws / (float)Wd)), Ws - 1);
  const long si = (long)sy * Ws + sx;
  const int v = lab_f32 ? (int)reinterpret_cast<const float*>(lab)[si] : (int)reinterpret_cast<const uint8_t*>(lab)[si];
  e16x8 lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    float f = 0.f;
    if (c < ncls) f = (v == c) ? 1.f : 0.f;
    else if (c == ncls) f = (v == 255) ? 1.f : 0.f;
    if (c < 8) lo[c] = (e16)f; else hi[c - 8] = (e16)f;
  }
  reinterpret_cast<e16x8*>(out)[2 * i] = lo;
  reinterpret_cast<e16x8*>(out)[2 * i + 1] = hi;
}

// scores[t] = sum_q mass[q][t] * fg[q],  fg = 1 - softmax(bilinear_ac(logits -> enc size))[0]
// pass 1: one block per 256 tokens writes a partial score row; pass 2 (one wave) sums the rows in block order
// (deterministic).  partial is [nblocks][32].
__global__ __launch_bounds__(256) void k_evict_partial(const float* lg, int ldl, int nc, int keep, int Hi, int Wi, int He, int We,
                                                       const float* mass, int T, float* partial) {
  __shared__ float red[4][32];
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int total = He * We;
  float fg = 0.f;
  if (q < total) {
    const int oy = q / We, ox = q - oy * We;
    int y0, y1, x0, x1; float wy, wx;
    src_coord(oy, Hi, He, 1, y0, y1, wy);
    src_coord(ox, Wi, We, 1, x0, x1, wx);
    const float* a = lg + ((long)y0 * Wi + x0) * ldl;
    const float* b = lg + ((long)y0 * Wi + x1) * ldl;
    const float* c = lg + ((long)y1 * Wi + x0) * ldl;
    const float* d = lg + ((long)y1 * Wi + x1) * ldl;
    float v0 = 0.f, mx = -3.0e38f, den = 0.f;
    float v[16];
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) {
      float f = -3.0e38f;
      if (ch < nc) f = ch > keep ? -1.0e10f : (a[ch] * (1.f - wx) + b[ch] * wx) * (1.f - wy) + (c[ch] * (1.f - wx) + d[ch] * wx) * wy;
      v[ch] = f; mx = fmaxf(mx, f);
    }
    v0 = v[0];
#pragma unroll
    for (int ch = 0; ch < 16; ++ch) den += ch < nc ? expf(v[ch] - mx) : 0.f;
    fg = 1.f - expf(v0 - mx) / den;
  }
  for (int t = 0; t < T; ++t) {
    const float s = wave_sum(q < total ? mass[(long)q * T + t] * fg : 0.f);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < T) partial[blockIdx.x * 32 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void k_evict_final(const float* partial, int nblocks, int T, float* scores) {
  if ((int)threadIdx.x < T) {
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += partial[b * 32 + threadIdx.x];
    scores[threadIdx.x] = s;
  }
}

// test-time augmentation merge (managers/evaluator.py:427-441): softmax of every augmentation's logits (horizontally
// flipped back where that augmentation was flipped), mean over augmentations, argmax.  logits: up to 4 NCHW fp32 maps.
struct TtaParams { const float* lg[8]; int flip[8]; int n_aug, nc, H, W; uint8_t* label; float* label_f32; float* prob; };

__global__ __launch_bounds__(256) void k_tta_merge(TtaParams p) {
  const long total = (long)p.H * p.W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int y = (int)(i / p.W), x = (int)(i - (long)y * p.W);
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  for (int a = 0; a < p.n_aug; ++a) {
    const long src = (long)y * p.W + (p.flip[a] ? p.W - 1 - x : x);
    float v[16], mx = -3.0e38f, den = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { v[c] = c < p.nc ? p.lg[a][(long)c * total + src] : -3.0e38f; mx = fmaxf(mx, v[c]); }
#pragma unroll
    for (int c = 0; c < 16; ++c) { v[c] = c < p.nc ? expf(v[c] - mx) : 0.f; den += v[c]; }
    const float inv = 1.f / den;
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] += v[c] * inv;
  }
  const float sc = 1.f / (float)p.n_aug;
  float best = -1.f; int arg = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < p.nc) {
      const float pr = acc[c] * sc;
      if (p.prob) p.prob[(long)c * total + i] = pr;
      if (pr > best) { best = pr; arg = c; }
    }
  }
  if (p.label) p.label[i] = (uint8_t)arg;
  if (p.label_f32) p.label_f32[i] = (float)arg;
}

// > 10 objects: one AOTEngine per 10 objects (engines/aot_engine.py:604-673).
// split: the label map of engine e keeps ids start..end renumbered from 1, everything else 0 (separate_mask, 610-628).
__global__ __launch_bounds__(256) void k_split_label(const float* label, int start_id, int end_id, float* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = label[i];
  out[i] = (v >= (float)start_id && v <= (float)end_id) ? v - (float)start_id + 1.f : 0.f;
}
// aggregate (soft_logit_aggregation, 650-673): per engine softmax over its nc channels; background = product of the engines'
// background probabilities, foreground channels concatenated; clamp to [1e-5, 1 - 1e-5]; logit.
struct AggParams { const float* lg[8]; int n, nc, nobj; long total; float* out; };
__global__ __launch_bounds__(256) void k_soft_aggregate(AggParams p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.total) return;
  float bg = 1.f;
  for (int e = 0; e < p.n; ++e) {
    float v[16], mx = -3.0e38f, den = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { v[c] = c < p.nc ? p.lg[e][(long)c * p.total + i] : -3.0e38f; mx = fmaxf(mx, v[c]); }
#pragma unroll
    for (int c = 0; c < 16; ++c) { v[c] = c < p.nc ? expf(v[c] - mx) : 0.f; den += v[c]; }
    const float inv = 1.f / den;
    bg *= v[0] * inv;
#pragma unroll
    for (int c = 1; c < 16; ++c)
      if (c <= p.nobj) {
        const float pr = fminf(fmaxf(v[c] * inv, 1e-5f), 1.f - 1e-5f);
        p.out[(long)(1 + e * p.nobj + (c - 1)) * p.total + i] = logf(pr / (1.f - pr));
      }
  }
  bg = fminf(fmaxf(bg, 1e-5f), 1.f - 1e-5f);
  p.out[i] = logf(bg / (1.f - bg));
}

// Jaccard counts per object id (evaluation/source/metrics.py:6-37 applied per id): counts[id] = {|pred == id & gt == id|,
// |pred == id | gt == id|}, pixels whose ground truth is the void label are skipped.  Integer atomics: deterministic.
__global__ __launch_bounds__(256) void k_mask_iou(const uint8_t* pred, const uint8_t* gt, long n, int num_ids, int void_label,
                                                  unsigned long long* counts) {
  __shared__ unsigned int sh[32][2];
  for (int i = threadIdx.x; i < 64; i += 256) sh[i >> 1][i & 1] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int a = pred[i], b = gt[i];
    if (b == void_label) continue;
    if (a == b) { if (a > 0 && a < num_ids) { atomicAdd(&sh[a][0], 1u); atomicAdd(&sh[a][1], 1u); } }
    else {
      if (a > 0 && a < num_ids) atomicAdd(&sh[a][1], 1u);
      if (b > 0 && b < num_ids) atomicAdd(&sh[b][1], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * num_ids; i += 256)
    if (sh[i >> 1][i & 1]) atomicAdd(&counts[i], (unsigned long long)sh[i >> 1][i & 1]);
}

// frame ingest: uint8 RGB HWC [Hs][Ws][3] -> bicubic resize (OpenCV INTER_CUBIC: a = -0.75, pixel-centre mapping, clamped taps)
// -> /255, ImageNet mean/std -> fp32 CHW [3][Hd][Wd] and/or NHWC8 e16 (dataloaders/eval_datasets.py:57-64,
// video_transforms.py:648-652, 676-680).  One thread per destination pixel.
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
  const float A = -0.75f;
  w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
}

__global__ __launch_bounds__(256) void k_ingest(const uint8_t* src, int Hs, int Ws, int Hd, int Wd, float* out_chw, e16* out_nhwc8) {
  const long total = (long)Hd * Wd;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int dy = (int)(i / Wd), dx = (int)(i - (long)dy * Wd);
  float rgb[3];
  if (Hs == Hd && Ws == Wd) {                       // the evaluator skips the resize when the size is unchanged (video_transforms.py:624-625)
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = (float)src[i * 3 + c];
  } else {
    const float fy = ((float)dy + 0.5f) * ((float)Hs / (float)Hd) - 0.5f;
    const float fx = ((float)dx + 0.5f) * ((float)Ws / (float)Wd) - 0.5f;
    const int sy = (int)floorf(fy), sx = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - (float)sy, wy);
    cubic_w(fx - (float)sx, wx);
    rgb[0] = rgb[1] = rgb[2] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(sy - 1 + a, 0), Hs - 1);
      float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(sx - 1 + b, 0), Ws - 1);
        const uint8_t* px = src + ((long)yy * Ws + xx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] += wx[b] * (float)px[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] += wy[a] * row[c];
    }
  }
  const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
  e16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (rgb[c] * (1.f / 255.f) - mean[c]) * istd[c];
    if (out_chw) out_chw[(long)c * total + i] = v;
    o[c] = (e16)v;
  }
  if (out_nhwc8) reinterpret_cast<e16x8*>(out_nhwc8)[i] = o;
}

inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

// block c of src ([nclips][block_bytes]) -> dst + slots[c] * slot_bytes; slots is a DEVICE table (a captured graph stays valid
// while the clips' free bank slots diverge after evictions); slots[c] < 0 skips the clip
__global__ __launch_bounds__(256) void k_scatter_blocks(const uint4* src, uint4* dst, const int* slots, long block16, long slot16) {
  const int c = blockIdx.y;
  const int sl = slots[c];
  if (sl < 0) return;
  const uint4* s = src + (long)c * block16;
  uint4* d = dst + (long)sl * slot16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < block16; i += (long)gridDim.x * 256) d[i] = s[i];
}

}  // namespace

#ifndef RMEM_F16
extern "C" int rmem_scatter_blocks(const void* src, void* dst, const int* slots_dev, int nclips, long long block_bytes,
                                   long long slot_bytes, void* stream) {
  RMEM_REQUIRE(src && dst && slots_dev && nclips >= 1 && block_bytes > 0 && block_bytes % 16 == 0 && slot_bytes % 16 == 0,
               "rmem_scatter_blocks: bad argument (sizes must be multiples of 16 bytes)");
  const long b16 = block_bytes / 16;
  const int blocks = (int)min((long)256, (b16 + 255) / 256);
  hipLaunchKernelGGL(k_scatter_blocks, dim3(blocks, nclips), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, slots_dev,
                     b16, slot_bytes / 16);
  return rmem_check_launch("rmem_scatter_blocks");
}
#endif

extern "C" int RMEM_API(rmem_image_to_nhwc8_images)(const float* img_chw, void* out, int images, int H, int W, void* stream) {
  RMEM_REQUIRE(img_chw && out && images >= 1 && H > 0 && W > 0, "rmem_image_to_nhwc8: bad argument");
  hipLaunchKernelGGL(k_image_to_nhwc8, dim3(nblk((long)H * W), images), dim3(256), 0, (hipStream_t)stream, img_chw, (e16*)out, H, W);
  return rmem_check_launch("rmem_image_to_nhwc8");
}

extern "C" int RMEM_API(rmem_image_ptrs_to_nhwc8)(const float* const* img_ptrs, void* out, int images, int H, int W, void* stream) {
  RMEM_REQUIRE(img_ptrs && out && images >= 1 && H > 0 && W > 0, "rmem_image_ptrs_to_nhwc8: bad argument");
  hipLaunchKernelGGL(k_image_ptrs_to_nhwc8, dim3(nblk((long)H * W), images), dim3(256), 0, (hipStream_t)stream, img_ptrs, (e16*)out, H, W);
  return rmem_check_launch("rmem_image_ptrs_to_nhwc8");
}

extern "C" int RMEM_API(rmem_image_to_nhwc8)(const float* img_chw, void* out, int H, int W, void* stream) {
  return RMEM_API(rmem_image_to_nhwc8_images)(img_chw, out, 1, H, W, stream);
}

extern "C" int RMEM_API(rmem_maxpool3x3s2_nhwc_images)(const void* x, void* y, int images, int H, int W, int C, void* stream) {
  RMEM_REQUIRE(x && y && images >= 1 && H > 0 && W > 0 && C % 8 == 0, "rmem_maxpool3x3s2_nhwc: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(k_maxpool3s2, dim3(nblk((long)Ho * Wo * (C / 8)), images), dim3(256), 0, (hipStream_t)stream, (const e16*)x, (e16*)y, H, W, C, Ho, Wo);
  return rmem_check_launch("rmem_maxpool3x3s2_nhwc");
}

extern "C" int RMEM_API(rmem_maxpool3x3s2_nhwc)(const void* x, void* y, int H, int W, int C, void* stream) {
  return RMEM_API(rmem_maxpool3x3s2_nhwc_images)(x, y, 1, H, W, C, stream);
}

extern "C" int RMEM_API(rmem_bilinear_nhwc_images)(const void* x, void* y, int images, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                                         void* stream) {
  RMEM_REQUIRE(x && y && images >= 1 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C % 8 == 0, "rmem_bilinear_nhwc: bad argument");
  hipLaunchKernelGGL(k_bilinear_nhwc, dim3(nblk((long)Ho * Wo * (C / 8)), images), dim3(256), 0, (hipStream_t)stream, (const e16*)x,
                     (e16*)y, Hi, Wi, Ho, Wo, C, align_corners);
  return rmem_check_launch("rmem_bilinear_nhwc");
}

extern "C" int RMEM_API(rmem_bilinear_nhwc)(const void* x, void* y, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream) {
  return RMEM_API(rmem_bilinear_nhwc_images)(x, y, 1, Hi, Wi, Ho, Wo, C, align_corners, stream);
}

#ifndef RMEM_F16
extern "C" int rmem_logits_post_images(const float* logits_nhwc, int images, int ldl, int num_classes, int keep_max_id, int Hi, int Wi,
                                       int Ho, int Wo, int align_corners, float* out_nchw, unsigned char* label_u8, float* label_f32,
                                       void* stream) {
  RMEM_REQUIRE(logits_nhwc && images >= 1 && num_classes >= 1 && num_classes <= 16 && ldl >= num_classes, "rmem_logits_post: bad argument");
  RMEM_REQUIRE(out_nchw || label_u8 || label_f32, "rmem_logits_post: no output requested");
  RMEM_REQUIRE(images == 1 || !out_nchw, "rmem_logits_post: a batch of images produces labels only");
  if (ldl == 16 && !out_nchw && ((uintptr_t)logits_nhwc % 16) == 0 && Ho >= 2 * Hi && Wo >= 2 * Wi && lt_tiles_fit(Hi, Wi, Ho, Wo, align_corners))
    hipLaunchKernelGGL(k_logits_labels_tile, dim3(((Ho + LT_H - 1) / LT_H) * ((Wo + LT_W - 1) / LT_W), images), dim3(256), 0, (hipStream_t)stream,
                       logits_nhwc, num_classes, keep_max_id, Hi, Wi, Ho, Wo, align_corners, label_u8, label_f32);
  else if (ldl == 16 && !out_nchw && ((uintptr_t)logits_nhwc % 16) == 0)
    hipLaunchKernelGGL(k_logits_labels4, dim3(nblk((long)Ho * ((Wo + 3) / 4)), images), dim3(256), 0, (hipStream_t)stream, logits_nhwc,
                       num_classes, keep_max_id, Hi, Wi, Ho, Wo, align_corners, label_u8, label_f32);
  else
    hipLaunchKernelGGL(k_logits_post, dim3(nblk((long)Ho * Wo), images), dim3(256), 0, (hipStream_t)stream, logits_nhwc, ldl, num_classes,
                       keep_max_id, Hi, Wi, Ho, Wo, align_corners, out_nchw, label_u8, label_f32);
  return rmem_check_launch("rmem_logits_post");
}
#endif

#ifndef RMEM_F16
extern "C" int rmem_logits_post(const float* logits_nhwc, int ldl, int num_classes, int keep_max_id, int Hi, int Wi, int Ho, int Wo,
                                int align_corners, float* out_nchw, unsigned char* label_u8, float* label_f32, void* stream) {
  return rmem_logits_post_images(logits_nhwc, 1, ldl, num_classes, keep_max_id, Hi, Wi, Ho, Wo, align_corners, out_nchw, label_u8, label_f32,
                                 stream);
}
#endif

extern "C" int RMEM_API(rmem_label_to_onehot16_images)(const void* label, int label_is_f32, int images, int Hs, int Ws, int Hd, int Wd, int num_classes,
                                             void* out, void* stream) {
  RMEM_REQUIRE(label && out && images >= 1 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && num_classes >= 1 && num_classes <= 15,
               "rmem_label_to_onehot16: bad argument");
  hipLaunchKernelGGL(k_label_onehot, dim3(nblk((long)Hd * Wd), images), dim3(256), 0, (hipStream_t)stream, label, label_is_f32, Hs, Ws, Hd, Wd,
                     num_classes, (e16*)out);
  return rmem_check_launch("rmem_label_to_onehot16");
}

extern "C" int RMEM_API(rmem_label_to_onehot16)(const void* label, int label_is_f32, int Hs, int Ws, int Hd, int Wd, int num_classes, void* out, void* stream) {
  return RMEM_API(rmem_label_to_onehot16_images)(label, label_is_f32, 1, Hs, Ws, Hd, Wd, num_classes, out, stream);
}

#ifndef RMEM_F16
extern "C" int rmem_resize_nearest_flip_f32(const float* src, int planes, int Hs, int Ws, float* dst, int Hd, int Wd, int flip_w, void* stream) {
  RMEM_REQUIRE(src && dst && planes >= 1 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "rmem_resize_nearest_flip_f32: bad argument");
  hipLaunchKernelGGL(k_resize_nearest_flip, dim3(nblk((long)Hd * Wd), planes), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, dst, Hd, Wd, flip_w);
  return rmem_check_launch("rmem_resize_nearest_flip_f32");
}
#endif

#ifndef RMEM_F16
extern "C" int rmem_evict_scores(const float* logits_nhwc, int ldl, int num_classes, int keep_max_id, int Hi, int Wi, int He, int We,
                                 const float* attn_mass, int T, float* scores, void* stream) {
  RMEM_REQUIRE(logits_nhwc && attn_mass && scores && T >= 1 && T <= 32 && num_classes >= 1 && num_classes <= 16, "rmem_evict_scores: bad argument");
  const int nb = (He * We + 255) / 256;
  RMEM_REQUIRE(nb <= 64, "rmem_evict_scores: more than 16384 tokens");
  // partial rows live behind the T scores in the caller's buffer: scores must hold 32 + 64 * 32 floats
  float* partial = scores + 32;
  hipLaunchKernelGGL(k_evict_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits_nhwc, ldl, num_classes, keep_max_id, Hi, Wi, He, We, attn_mass, T, partial);
  hipLaunchKernelGGL(k_evict_final, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, T, scores);
  return rmem_check_launch("rmem_evict_scores");
}
#endif

#ifndef RMEM_F16
extern "C" int rmem_tta_merge(const float* const* logits_nchw, const int* flips, int n_aug, int num_classes, int H, int W,
                              unsigned char* label_u8, float* label_f32, float* prob_nchw, void* stream) {
  RMEM_REQUIRE(logits_nchw && flips && n_aug >= 1 && n_aug <= 8 && num_classes >= 1 && num_classes <= 16, "rmem_tta_merge: 1..8 augmentations, <= 16 classes");
  RMEM_REQUIRE(label_u8 || label_f32 || prob_nchw, "rmem_tta_merge: no output requested");
  TtaParams p;
  for (int a = 0; a < 8; ++a) { p.lg[a] = a < n_aug ? logits_nchw[a] : nullptr; p.flip[a] = a < n_aug ? flips[a] : 0; }
  p.n_aug = n_aug; p.nc = num_classes; p.H = H; p.W = W; p.label = label_u8; p.label_f32 = label_f32; p.prob = prob_nchw;
  hipLaunchKernelGGL(k_tta_merge, dim3(nblk((long)H * W)), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_tta_merge");
}
#endif

#ifndef RMEM_F16
extern "C" int rmem_split_label(const float* label, int start_id, int end_id, float* out, long long n, void* stream) {
  RMEM_REQUIRE(label && out && n > 0 && start_id >= 1 && end_id >= start_id, "rmem_split_label: bad argument");
  hipLaunchKernelGGL(k_split_label, dim3(nblk((long)n)), dim3(256), 0, (hipStream_t)stream, label, start_id, end_id, out, (long)n);
  return rmem_check_launch("rmem_split_label");
}

extern "C" int rmem_soft_logit_aggregate(const float* const* logits_nchw, int n_engines, int num_classes, int objs_per_engine, int H, int W,
                                         float* out_nchw, void* stream) {
  RMEM_REQUIRE(logits_nchw && out_nchw && n_engines >= 1 && n_engines <= 8 && num_classes >= 2 && num_classes <= 16 &&
               objs_per_engine >= 1 && objs_per_engine < num_classes && H > 0 && W > 0, "rmem_soft_logit_aggregate: bad argument");
  AggParams p;
  for (int e = 0; e < 8; ++e) p.lg[e] = e < n_engines ? logits_nchw[e] : nullptr;
  p.n = n_engines; p.nc = num_classes; p.nobj = objs_per_engine; p.total = (long)H * W; p.out = out_nchw;
  hipLaunchKernelGGL(k_soft_aggregate, dim3(nblk(p.total)), dim3(256), 0, (hipStream_t)stream, p);
  return rmem_check_launch("rmem_soft_logit_aggregate");
}

extern "C" int rmem_mask_iou_counts(const unsigned char* pred, const unsigned char* gt, long long n, int num_ids, int void_label,
                                    unsigned long long* counts, void* stream) {
  RMEM_REQUIRE(pred && gt && counts && n > 0 && num_ids >= 2 && num_ids <= 32, "rmem_mask_iou_counts: bad argument");
  const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_mask_iou, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, gt, (long)n, num_ids, void_label, counts);
  return rmem_check_launch("rmem_mask_iou_counts");
}
#endif

extern "C" int RMEM_API(rmem_ingest_rgb8)(const unsigned char* rgb_hwc, int Hs, int Ws, int Hd, int Wd, float* out_chw, void* out_nhwc8, void* stream) {
  RMEM_REQUIRE(rgb_hwc && (out_chw || out_nhwc8) && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "rmem_ingest_rgb8: bad argument");
  hipLaunchKernelGGL(k_ingest, dim3(nblk((long)Hd * Wd)), dim3(256), 0, (hipStream_t)stream, rgb_hwc, Hs, Ws, Hd, Wd, out_chw, (e16*)out_nhwc8);
  return rmem_check_launch("rmem_ingest_rgb8");
}
