/* librmem_hip.so -- C ABI of the MI355X (gfx950) space-time memory-reading engine.
 *
 * The reference (Bardli/RMem_ocu) has no FFI: its hot path is nn.Modules calling
 * ATen/cuDNN kernels.  This header is the boundary one level below
 * the reference's Python inference API; each entry point names the reference call
 * site(s) whose vendor kernels it replaces (paths relative to aot_plus/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work and never
 *     synchronise, so a caller may wrap any sequence of them in a hipGraph capture;
 *   - return value: 0 on success, negative on error (rmem_last_error_string() says why);
 *     no exception crosses the boundary; the only global mutable state is the
 *     thread-local error string and the opt-in launch timers (rmem_profile_* /
 *     rmem_gated_profile_*: process-wide, mutex-protected, off unless started);
 *   - "bf16" = IEEE bfloat16 stored as uint16; feature maps are NHWC with N = 1, which
 *     for the LSTT is the same memory as the reference's [L, B=1, C] token layout;
 *   - leading dimensions (ld*) are in elements.
 */
#ifndef RMEM_H_
#define RMEM_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMEM_ABI_VERSION 8

int rmem_abi_version(void);
const char* rmem_last_error_string(void);

/* ------------------------------------------------------------------ convolution / linear
 * Implicit-GEMM NHWC convolution, bf16 in, fp32 accumulate:
 *   y = act(conv(x, w) + bias (+ residual)),  optional y2 = bf16(conv(x, w) + bias).
 * Replaces: encoders/resnet.py:48-68, 179-195 (conv + FrozenBatchNorm2d folded into w/bias + ReLU
 * + residual), models/aot.py:25-29, 133 (encoder_projector), models/aot.py:68-74, 112 (id bank),
 * decoders/fpn.py:22-32 (ConvGN convs, adapters, conv_out), and every nn.Linear of
 * layers/transformer.py:487-512 and layers/attention.py:19-25 (a 1x1 conv with H = rows, W = 1).
 * w is [Cout][KH][KW][Cin] bf16; Cin must be a multiple of 8. */
typedef struct rmem_conv_desc {
  int H, W, Cin;      /* input feature map */
  int Ho, Wo, Cout;   /* output feature map (checked against the geometry) */
  int KH, KW, stride, pad;
  int ldo;            /* y row stride  (>= Cout) */
  int ldr;            /* residual row stride */
  int ld2;            /* y2 row stride */
  int relu;           /* activation after bias/residual: 0 none, 1 ReLU, 2 exact (erf) GELU, 3 SiLU (layers/attention.py:89) */
  int out_f32;        /* 1: y is fp32, 0: bf16 */
  int res_f32;        /* 1: residual is fp32, 0: bf16 */
  int ldx;            /* 1x1 stride-1 problems only: input row stride in elements (0 = Cin); lets a GEMM read a column
                         range of a wider activation buffer (the torch.split / torch.cat of transformer.py:1104-1124) */
  int batch;          /* images in x / y (0 or 1 = one): x is [batch][H][W][Cin], y [batch][Ho][Wo] rows of ldo; several
                         clips' frames go through the encoder as one launch per layer */
  int act_begin;      /* the activation applies to output channels >= act_begin (multiple of 8; 0 = all): one GEMM for
                         linear_QV, whose Q half is raw and whose V half goes through SiLU (transformer.py:1104-1110) */
  int res_up_h, res_up_w, res_up_align;
                      /* res_up_h > 0: `residual` is a LOWER-resolution bf16 map [batch][res_up_h][res_up_w] rows of ldr and is
                         bilinearly resized to (Ho, Wo) on the fly (rounded to bf16 like rmem_bilinear_nhwc, so both routes
                         are bit-identical): `F.interpolate(x, size) + adapter(shortcut)` of decoders/fpn.py:49-52, 57-60
                         without materialising the resized map.  Needs Cout % 8 == 0 and 16-byte aligned operands. */
} rmem_conv_desc;

/* Problems with few output tiles are cut along K (split-K) when a workspace of at least
 * rmem_conv_workspace_bytes(desc) bytes is supplied (fp32 slabs, summed in slice order: reproducible);
 * workspace may be NULL (no split-K). */
size_t rmem_conv_workspace_bytes(const rmem_conv_desc* desc);
int rmem_conv2d_nhwc(const rmem_conv_desc* desc, const void* x, const void* w, const float* bias,
                     const void* residual, void* y, void* y2, void* workspace, void* stream);

/* n <= 4 GEMMs of IDENTICAL shape (desc: 1x1, stride 1) with different operands as one launch: the per-layer linears of one
 * memory update (linear_QMem / linear_VMem / linear_V of layers/transformer.py:279-285 for the 3 LSTT layers) do not depend on
 * each other, and at M = HW = 1674 a launch costs more than its arithmetic.  bias / residual may be NULL or hold NULLs. */
int rmem_linear_grouped(const rmem_conv_desc* desc, int n, const void* const* x, const void* const* w,
                        const float* const* bias, const void* const* residual, void* const* y, void* stream);

/* y = act([x | x2 sampled at stride2] * w_cat^T + bias): the closing 1x1 convolution of a ResNet bottleneck and its (strided) 1x1
 * shortcut (encoders/resnet.py:48-68: `out = conv3(out); identity = downsample(x); out += identity; relu`) as ONE GEMM over
 * K = desc->Cin + Cin2, so the shortcut tensor is neither written nor re-read.  desc describes the dense 1x1 problem on x
 * (batch, H = Ho, W = Wo, Cin, Cout, relu, ldo; no residual); x2 is NHWC [batch][H2][W2][Cin2] with (H2 - 1) / stride2 + 1 == Ho;
 * w_cat is [Cout][Cin + Cin2] bf16 (both BN-folded weights side by side), bias the sum of both folded biases.
 * Cin and Cin2 must be multiples of 64. */
int rmem_conv1x1_dual_nhwc(const rmem_conv_desc* desc, const void* x, const void* x2, int H2, int W2, int Cin2, int stride2,
                           const void* w_cat, const float* bias, void* y, void* stream);

/* The tail of a ResNet bottleneck chained into the FIRST 1x1 convolution of the following block, one launch
 * (encoders/resnet.py:62-68 of block i, then :48-50 of block i + 1 -- or of the next layer's first block):
 *   y  = relu(b . w3^T + bias3 + res)      b [M][K1] (the 3x3 conv's output), w3 [Cout = 256][K1], res / y [M][256], M = batch * Ho * Wo
 *   a2 = relu(y . w1^T + bias1)            w1 [N2][256], a2 [M][N2], N2 = 64 or 128
 * y is stored (later blocks need it as their shortcut) but never read back: its 64-row tile stays in LDS, rounded as stored, as the
 * A operand of the second GEMM.  Dual form (x2 != NULL, res == NULL; the block with the strided 1x1 shortcut, as
 * rmem_conv1x1_dual_nhwc): y = relu([b | x2 sampled at stride2] . w3^T + bias3) with w3 [256][K1 + Cin2], x2 NHWC [batch][H2][W2][Cin2].
 * Bit-identical to rmem_conv2d_nhwc (or rmem_conv1x1_dual_nhwc) followed by rmem_conv2d_nhwc.  K1, Cin2 multiples of 64. */
typedef struct rmem_bneck_chain_desc { int batch, Ho, Wo, K1, Cout, N2, H2, W2, Cin2, stride2; } rmem_bneck_chain_desc;
int rmem_bneck_chain(const rmem_bneck_chain_desc* desc, const void* b, const void* x2, const void* w3, const float* bias3, const void* res,
                     void* y, const void* w1, const float* bias1, void* a2, void* stream);

/* ------------------------------------------------------------------ memory-read attention
 * out[q, 32h:32h+32] = softmax_k( (Q[q,h]+pe_cur[h]) . (K[k,h]+pe_mem[slot(k),h]) / sqrt(32) ) V[k,h]
 * over the keys named by the chunk table, plus (optionally) the per-memory-frame probability mass
 * mass[q, t] = mean_h sum_{k in frame t} p[h, q, k].
 * Replaces: layers/attention.py:45-64 as called from layers/transformer.py:632-635 (long-term),
 * the K + temporal-PE materialisation of transformer.py:594-626, and the attention-weight recording of
 * transformer.py:636-643.  With chunks == NULL it is plain attention over one key frame of lk_single
 * keys split into nchunks ranges (transformer.py:569 self-attention, 657-662 short-term attention).
 * The output is pre-projection (attention.py:79 is a rmem_conv2d_nhwc call).
 * Head dim is fixed at 32; heads <= 8. */
typedef struct rmem_attn_chunk {
  int slot;       /* bank slot: keys at k_bank + slot * slot_stride */
  int key_begin;  /* first key (row) of the chunk inside the slot */
  int key_count;  /* >= 0; 0 = padding row (no keys, no mass): lets clips with shorter banks share a launch, rmem_mem_read_attn_clips */
  int pe_slot;    /* row of pe_mem added to these keys, or -1 */
  int t;          /* memory-frame index the chunk's probability mass is credited to */
  int reserved[3];
} rmem_attn_chunk;

size_t rmem_attn_workspace_bytes(int Lq, int heads, int nchunks);

int rmem_mem_read_attn(const void* q, int ldq,                     /* bf16 [Lq][ldq] */
                       const void* k_bank, const void* v_bank,     /* bf16, rows [Lk][ldkv] per slot */
                       long long slot_stride, int ldkv,
                       const rmem_attn_chunk* chunks, int nchunks, /* device table, or NULL */
                       int lk_single,                              /* keys when chunks == NULL; with a table: total keys
                                                                      named by it (used by rmem_profile_* only) */
                       const float* pe_cur,                        /* fp32 [C] or NULL */
                       const float* pe_mem,                        /* fp32 [4][C] or NULL */
                       int Lq, int heads,
                       void* out, int ldo,                         /* bf16 [Lq][ldo] */
                       float* attn_mass, int T,                    /* fp32 [Lq][T] or NULL */
                       void* workspace, void* stream);

/* The same for nclips independent clips of identical shape in ONE launch (clips of a group advance in lockstep): clip c's
 * queries / one-frame keys / outputs sit c * {q, kv, out}_clip_stride elements further, its chunk-table rows are
 * chunks[c * nchunks ...] (bank slots in them are global indexes into k_bank / v_bank), its mass is attn_mass + c * Lq * T,
 * the workspace is nclips times rmem_attn_workspace_bytes. */
int rmem_mem_read_attn_clips(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride, int ldkv,
                             const rmem_attn_chunk* chunks, int nchunks, int lk_single, const float* pe_cur, const float* pe_mem,
                             int Lq, int heads, void* out, int ldo, float* attn_mass, int T, int nclips,
                             long long q_clip_stride, long long kv_clip_stride, long long out_clip_stride,
                             void* workspace, void* stream);

/* The long-term memory read AND the short-term attention of an LSTT block (layers/transformer.py:632-635, 656-662) as ONE launch:
 * both use the same queries (curr_Q) and are independent of each other, so the short-term attention's workgroups ride behind the
 * memory read's as extra grid slices (every XCD gets its share of both) instead of a second launch that starts on an empty
 * GPU.  The memory read is exactly rmem_mem_read_attn_clips with a chunk table (out_long, attn_mass); the second attention
 * is softmax(Q K_short^T / sqrt(32)) V_short over lk_short keys of one frame per clip ([lk_short][ldkv] rows, clip c at
 * c * kv_short_clip_stride), no temporal embedding, written to out_short ([Lq][ldo] rows, clip c at c * out_short_clip_stride).
 * Results are bit-identical to the two separate launches. */
int rmem_lstt_attn_pair_clips(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride, int ldkv,
                              const rmem_attn_chunk* chunks, int nchunks, int lk_total, const float* pe_cur, const float* pe_mem,
                              int Lq, int heads, void* out_long, int ldo, float* attn_mass, int T, int nclips,
                              long long q_clip_stride, long long out_clip_stride,
                              const void* k_short, const void* v_short, int lk_short, long long kv_short_clip_stride,
                              void* out_short, long long out_short_clip_stride, void* workspace, void* stream);

/* Optional timing of the memory-read launches (chunks != NULL) with HIP events on the launch stream:
 * between start and stop every such launch outside a graph capture is bracketed by two events;
 * start() also calibrates what an event bracket costs around an empty kernel and stop() subtracts that per launch;
 * stop() synchronises on the events and returns the summed kernel time, the summed algorithmic FLOPs
 * (4 * Lq * keys * C per launch) and the launch count.  Used by bench.py's roofline leg. */
int rmem_profile_start(int max_launches);
int rmem_profile_stop(double* total_ms, double* total_flops, int* launches);

/* ------------------------------------------------------------------ normalisation / activation
 * LayerNorm over 256 channels of (a [+ b]); writes any of: bf16 y, fp32 y, bf16 (y + pos).
 * Replaces: layers/transformer.py:566-568 (norm1 and with_pos_embed), 574 (norm2), 659-660 (norm4 of a
 * sum), 683 (norm3), 250-259 (decoder_norms). */
int rmem_layernorm256(const void* a, int a_is_f32, int lda, const void* b, int b_is_f32, int ldb,
                      const float* gamma, const float* beta, float eps, int M,
                      void* y_bf16, int ldy, const float* pos, void* ypos_bf16, int ldyp,
                      float* y_f32, int ldyf, void* stream);

/* Two LayerNorm(256) of summed bf16 inputs with ONE weight set as one launch: y0 = LN(a0 + b0), y1 = LN(a1 + b1), all
 * [M][256] bf16 contiguous.  Replaces the two norm4 calls of layers/transformer.py:659-660. */
int rmem_layernorm256_pair(const void* a0, const void* b0, void* y0, const void* a1, const void* b1, void* y1,
                           const float* gamma, const float* beta, float eps, int M, void* stream);

/* ------------------------------------------------------------------ LSTT block chains (row-local sequences as ONE launch)
 * Between its three attentions an LSTT block (layers/transformer.py:553-692) is a sequence of row-local operations on the
 * [tokens][256] residual stream.  Each entry point below runs one such sequence for all rows of `clips` clips of L tokens
 * (rows are [clip][token], everything contiguous unless a leading dimension is given), a workgroup owning 32 consecutive
 * tokens of one clip with the rows resident in LDS; results are bit-identical to the same sequence of rmem_conv2d_nhwc /
 * rmem_layernorm256 / rmem_layernorm256_pair launches (same accumulation order, same points of rounding).
 * Weights are 16-bit [N][K] matrices (nn.Linear layout) re-packed once in FRAGMENT ORDER for NW = rmem_lstt_chain_waves()
 * waves per workgroup: [N / 256][NW][K / 32][16 / NW][64][8] = [column block][wave][k chunk][column tile j][lane][8 consecutive k],
 * element (n, k) with n = 256 nb + (256 / NW) wave + 16 j + (lane & 15), k = 32 kc + 8 (lane >> 4) + e
 * (rmem_ocu_amd/pack.py::pack_frag), so that a wave reads one MFMA B fragment as 1 KiB of contiguous memory.  Biases and
 * LayerNorm parameters are fp32; every pointer must be 16-byte aligned.
 *
 * chain A, after the self attention (transformer.py:571-576, 659-660):
 *   x += att . w_proj^T + b_proj;  curr_v = LN2(x);  curr_q = curr_v . w_q^T + b_q;
 *   k4 = LN4(short_k + curr_q);  v4 = LN4(short_v + curr_v) */
int rmem_lstt_chain_waves(void);      /* waves per workgroup the chain kernels were built for (4 or 8): the weight packing depends on it */
typedef struct rmem_chain_a_desc {
  int L, clips; float eps; int reserved;
  const void* att; float* x;
  const void* w_proj; const float* b_proj; const float* ln2_g; const float* ln2_b; void* curr_v;
  const void* w_q; const float* b_q; void* curr_q;
  const void* short_k; const void* short_v; const float* ln4_g; const float* ln4_b; void* k4; void* v4;
} rmem_chain_a_desc;
int rmem_lstt_chain_a(const rmem_chain_a_desc* d, void* stream);

/* chain B, after the long-term and the short-term attention (transformer.py:635, 662, 673-685):
 *   x += att_long . w_long^T + b_long;  tgt3 = att_short . w_short^T + b_short;  x += tgt3;  h1 = LN3(x) . w1^T + b1   ([rows][1024])
 * gn_partial (optional): per-row-block (sum, sum of squares) of h1 over each of the 32 GroupNorm groups of 32 channels,
 * fp32 [clips][32][gn_splits][2], entry `split` = the row block (gn_splits >= ceil(L / 32); the caller keeps the unused
 * entries zero): with gn_splits = 64 the statistics rmem_gn_act_dwconv5x5_prestats_nhwc_images consumes (layers/basic.py:27-35). */
typedef struct rmem_chain_b_desc {
  int L, clips; float eps; int gn_splits;
  const void* att_long; const void* att_short; float* x;
  const void* w_long; const float* b_long; const void* w_short; const float* b_short; void* tgt3;
  const float* ln3_g; const float* ln3_b; const void* w1; const float* b1; void* h1; float* gn_partial;
} rmem_chain_b_desc;
int rmem_lstt_chain_b(const rmem_chain_b_desc* d, void* stream);

/* chain C, after GroupNorm + GELU + depth-wise 5x5 (transformer.py:685-687, 250-259; 565-569 of the NEXT block):
 *   h3 != NULL:     x += h3 . w2^T + b2  (h3 [rows][1024]);  dec_out[row * ld_dec + 0..255] = LN_dec(x)
 *   w_qkv != NULL:  qkv = LN1'(x) . w_qkv^T + b_qkv + pos_qk  ([rows][768]; pos_qk fp32 [rows][768]) of the next block
 * (h3 == NULL: the first block of the stack, x as the projector left it; w_qkv == NULL: the last block). */
typedef struct rmem_chain_c_desc {
  int L, clips; float eps; int ld_dec;
  float* x;
  const void* h3; const void* w2; const float* b2; const float* dec_g; const float* dec_b; void* dec_out;
  const float* ln1_g; const float* ln1_b; const void* w_qkv; const float* b_qkv; const float* pos_qk; void* qkv;
} rmem_chain_c_desc;
int rmem_lstt_chain_c(const rmem_chain_c_desc* d, void* stream);

/* LayerNorm over C in {128, 256, 512, 1024} channels; bf16 and/or fp32 output.  Replaces the nn.LayerNorm call sites of the
 * Swin-B encoder (encoders/swin/swin_transformer.py:266, 318, 354, 538, 704).  Rows move as 8 / 16-byte vectors: lda, ldy, ldyf
 * multiples of 4, a / y_f32 / gamma / beta 16-byte and y_bf16 8-byte aligned. */
int rmem_layernorm(const void* a, int a_is_f32, int lda, const float* gamma, const float* beta, float eps, int M, int C,
                   void* y_bf16, int ldy, float* y_f32, int ldyf, void* stream);

/* Swin patch merging without the linear: gather the 2x2 neighbourhood of every output token of an fp32 [H][W][C] map
 * (zero beyond an odd border), LayerNorm over 4C, bf16 [ceil(H/2)*ceil(W/2)][4C] (swin_transformer.py:336-355; the
 * reduction Linear at 356 is a rmem_conv2d_nhwc call).  C in {128, 256}. */
int rmem_patch_merge_ln(const float* x, int H, int W, int C, const float* gamma, const float* beta, float eps,
                        void* y_bf16, void* stream);
/* the same over a batch: x fp32 [images][H][W][C] -> y [images][ceil(H/2)*ceil(W/2)][4C] (encoder_batch.SwinBatchEncoder) */
int rmem_patch_merge_ln_images(const float* x, int images, int H, int W, int C, const float* gamma, const float* beta, float eps,
                               void* y_bf16, void* stream);

/* Swin (shifted-)window attention over a [H][W] token map, 7x7 windows, head dim 32: padding, cyclic shift, window
 * partition / reverse, relative-position bias and the shifted-window mask are index arithmetic inside the kernel.
 * qkv: bf16 [H*W][3C] (q | k | v); qkv_bias: fp32 [3C] (what a padded zero token projects to);
 * bias_mask_table: fp32 [4][heads][49][49] = (relative_position_bias + mask of window type t) * log2(e), type = 2 * (last
 * window row) + (last window column) (only type 0 is read when shift == 0); out: bf16 [H*W][C], pre-projection.
 * Replaces encoders/swin/swin_transformer.py:156-195 and the token plumbing of 263-305. */
int rmem_window_attn(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int H, int W,
                     int C, int heads, int shift, void* stream);
/* the same over a batch of token maps: qkv [images][H*W][3C], out [images][H*W][C]; windows never cross images */
int rmem_window_attn_images(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int images, int H, int W,
                            int C, int heads, int shift, void* stream);

/* y = a + b (bf16).  Replaces the `curr_v + curr_id_emb` adds of layers/transformer.py:279-285. */
int rmem_add16(const void* a, const void* b, void* y, long long n, void* stream);
/* n <= 8 such adds of equal length as one launch (the per-layer adds of one memory update are independent). */
int rmem_add16_grouped(int n, const void* const* a, const void* const* b, void* const* y, long long count, void* stream);

/* GroupNorm on NHWC bf16 with fused activation (0 none, 1 ReLU, 2 exact GELU).
 * Replaces: layers/basic.py:31-32 (GN(32)+GELU of the conv-FFN) and layers/basic.py:69-70 +
 * decoders/fpn.py:44-64 (GN(8)+ReLU of the FPN head). */
size_t rmem_groupnorm_workspace_bytes(int groups);
int rmem_groupnorm_nhwc(const void* x, int M, int C, int groups, const float* gamma, const float* beta,
                        float eps, int act, void* y, float* workspace, void* stream);

/* Same with an fp32 input: the final GroupNorm1D(512, 2 groups) over the concatenated residual streams of the DeAOT
 * stack (layers/transformer.py:755-758, 806-808; layers/basic.py:6-12). */
int rmem_groupnorm_f32_nhwc(const float* x, int M, int C, int groups, const float* gamma, const float* beta,
                            float eps, int act, void* y, float* workspace, void* stream);

/* Batched forms: x / y hold `images` maps back to back ([images][M][C]); statistics are per image.  workspace: images times
 * rmem_groupnorm_workspace_bytes(groups). */
int rmem_groupnorm_nhwc_images(const void* x, int images, int M, int C, int groups, const float* gamma, const float* beta,
                               float eps, int act, void* y, float* workspace, void* stream);
/* y[m][n] = bias[n] + sum_c w[n][c] * act(GroupNorm(x))[m][c], n < N <= 16, fp32 rows of ldy: the segmentation head's
 * `conv_out(relu(gn(x)))` (decoders/fpn.py:62-66) in one pass over x (the normalised 128-channel map is never written;
 * it is rounded to bf16 on chip exactly as rmem_groupnorm_nhwc would store it).  C must be 128; w is [N][128] bf16. */
int rmem_groupnorm_head_nhwc_images(const void* x, int images, int M, int C, int groups, const float* gamma, const float* beta,
                                    float eps, int act, const void* w, const float* bias, int N, float* y, int ldy,
                                    float* workspace, void* stream);
int rmem_gn_act_dwconv5x5_nhwc_images(const void* x, int images, int H, int W, int C, int groups, const float* gamma,
                                      const float* beta, float eps, int act, const float* w_t, void* y, float* workspace, void* stream);
int rmem_bilinear_nhwc_images(const void* x, void* y, int images, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream);
/* rmem_gn_act_dwconv5x5_nhwc_images without its statistics launch: `stats` already holds, per image and group, the (sum, sum of
 * squares) of up to 64 row ranges -- fp32 [images][groups][64][2], unused entries zero -- written by the producer of x
 * (rmem_lstt_chain_b: gn_partial with gn_splits = 64, i.e. at most 2048 rows per image). */
int rmem_gn_act_dwconv5x5_prestats_nhwc_images(const void* x, int images, int H, int W, int C, int groups, const float* gamma,
                                               const float* beta, float eps, int act, const float* w_t, void* y, const float* stats,
                                               void* stream);

/* Depth-wise 5x5, pad 2, NHWC bf16; w_t is [25][C] fp32.  Replaces layers/basic.py:19-25, 33. */
int rmem_dwconv5x5_nhwc(const void* x, const float* w_t, void* y, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------ DeAOT gated propagation attention
 * Single head, d_att = 128, values DV wide (1024 in R50-DeAOTL):
 *   out[q, :] = ( softmax_k( (Q[q]+pe_cur) . (K[k]+pe_mem[slot(k)]) / sqrt(128) ) V[k, :] ) * U[q, :]
 * Replaces layers/attention.py:138-208 (GatedPropagation.forward through `outputs * U`; use_linear False) as called from
 * layers/transformer.py:1183-1184 (long-term attention over the restricted bank: the K + temporal-embedding
 * materialisation of 1141-1177 and the [V | ID_V] concatenation of 1179 happen by indexing; attn_mass is
 * record_attn_weight of 1185-1192) and from 1229 (self-attention, after the linear_QK / V / U GEMMs of attention.py:151-173).
 * The depth-wise 5x5 and the projection (attention.py:210-211) are rmem_dwconv5x5_nhwc / rmem_conv2d_nhwc calls.
 * chunks: the rmem_attn_chunk table (<= 64 rows, key_begin a multiple of 64) or NULL for ONE key frame of
 * keys_per_frame keys cut into nchunks ranges.  The gate U is given as two column ranges: u_a covers [0, usplit),
 * u_b covers [usplit, DV) or is NULL for all ones (layer 0, transformer.py:1117-1118).
 * out: bf16 [Lq][ldo]; attn_mass: fp32 [Lq][frames] or NULL. */
size_t rmem_gated_attn_workspace_bytes(int Lq, int DV, int frames, int keys_per_frame, int nchunks);
int rmem_gated_attn(const void* q, int ldq,                                        /* bf16 [Lq][ldq], 128 used */
                    const void* k_bank, long long k_slot_stride, int ldk,          /* bf16 rows of 128 */
                    const void* v_bank, long long v_slot_stride, int ldv,          /* bf16 rows of DV */
                    const rmem_attn_chunk* chunks, int nchunks, int frames, int keys_per_frame,
                    const float* pe_cur, const float* pe_mem,                      /* fp32 [128], [4][128] or NULL */
                    int Lq, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit,
                    void* out, int ldo, float* attn_mass,
                    const float* dw_w_t, int H, int W,   /* optional: out = dw_conv5x5(gated) (attention.py:210), weights [25][DV],
                                                            H * W == Lq; NULL: out = gated */
                    void* workspace, void* stream);
/* 15x15 local window flavour (layers/attention.py:281-349, use_linear False, one head): keys/values are the previous
 * frame's [H*W] tokens; a key contributes to a query iff |dy| <= 7 and |dx| <= 7 (the zero padding + 1e8 mask of
 * 299-303, 338); rel: fp32 [H*W][ldrel] = relative_emb_k(q) (attention.py:305, a rmem_conv2d_nhwc call with fp32 output),
 * column (dy+7)*15 + (dx+7).  Workspace: rmem_gated_attn_workspace_bytes(H*W, DV, 1, H*W, 8). */
int rmem_local_gated_attn(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* rel, int ldrel,
                          int H, int W, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit,
                          void* out, int ldo, const float* dw_w_t, void* workspace, void* stream);

/* Both gated attentions for nclips independent clips of identical shape in ONE launch per kernel (clips of a group advance in
 * lockstep): q, the gates, out and (one-frame calls) k / v / rel are [clip][rows][ld] arrays, i.e. clip c sits rows * ld elements
 * further; the bank is addressed through GLOBAL slot indexes in the table, clip c's table rows are chunks[c * nchunks ...], its
 * mass attn_mass + c * Lq * frames; the workspace is nclips times rmem_gated_attn_workspace_bytes.  Results per clip are those
 * of the single-clip entry points (the split of the key stream into partial-sum slabs depends on the clip count: fp32 summation
 * order only). */
int rmem_gated_attn_clips(const void* q, int ldq, const void* k_bank, long long k_slot_stride, int ldk, const void* v_bank,
                          long long v_slot_stride, int ldv, const rmem_attn_chunk* chunks, int nchunks, int frames, int keys_per_frame,
                          const float* pe_cur, const float* pe_mem, int Lq, int DV, const void* u_a, int ldua, const void* u_b, int ldub,
                          int usplit, void* out, int ldo, float* attn_mass, const float* dw_w_t, int H, int W, int nclips,
                          void* workspace, void* stream);
int rmem_local_gated_attn_clips(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* rel, int ldrel,
                                int H, int W, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit, void* out, int ldo,
                                const float* dw_w_t, int nclips, void* workspace, void* stream);
/* HIP-event timing of the k_gp_pv launches of rmem_gated_attn calls that carry a chunk table (bench.py's roofline leg) */
int rmem_gated_profile_start(void);
int rmem_gated_profile_stop(double* total_ms, double* total_flops, int* launches);
/* Memory-bank append for a group of clips: block c of src ([nclips][block_bytes], the new K or V entries of all clips) goes to
 * dst + slots_dev[c] * slot_bytes.  slots_dev is a device table, so one captured launch serves whatever slots the clips'
 * eviction policies have freed (layers/transformer.py:305-322 appends, 432-433 removes). */
int rmem_scatter_blocks(const void* src, void* dst, const int* slots_dev, int nclips, long long block_bytes, long long slot_bytes,
                        void* stream);
/* strided 2-D device copy (rows of row_bytes): the torch.cat / slice bookkeeping of transformer.py:840-872 */
int rmem_copy2d_async(void* dst, long long dst_pitch, const void* src, long long src_pitch, long long row_bytes, int rows, void* stream);

/* GroupNorm + activation + depth-wise 5x5 as two launches (statistics; normalise + activate + convolve through an LDS tile)
 * instead of three, bit-identical to rmem_groupnorm_nhwc followed by rmem_dwconv5x5_nhwc.  Replaces layers/basic.py:27-35
 * (GNActDWConv2d.forward).  C % 64 == 0, channels per group in {8, 16, 32, 64}. */
int rmem_gn_act_dwconv5x5_nhwc(const void* x, int H, int W, int C, int groups, const float* gamma, const float* beta, float eps,
                               int act, const float* w_t, void* y, float* workspace, void* stream);

/* ------------------------------------------------------------------ layout / resampling */
/* fp32 [3][H][W] image -> bf16 [H][W][8] (channels 3..7 zero): input of encoders/resnet.py:179. */
int rmem_image_to_nhwc8(const float* img_chw, void* out, int H, int W, void* stream);
/* the `_images` forms of this section take `images` equally sized problems stored back to back (clip groups, encoder
 * look-ahead) as ONE launch; each image's result equals the single-image call. */
int rmem_image_to_nhwc8_images(const float* img_chw, void* out, int images, int H, int W, void* stream);
/* The same with the images named by a DEVICE table of `images` pointers (fp32 [3][H][W] each): the frames of several clips go into
 * one encoder batch from wherever the caller keeps them, without a staging copy (models/aot.py:116-134: the encoder's input). */
int rmem_image_ptrs_to_nhwc8(const float* const* img_ptrs, void* out, int images, int H, int W, void* stream);
/* ResNet stem without im2col traffic (encoders/resnet.py:131-135: conv1 7x7 stride 2 pad 3, bn1 folded, ReLU).
 * rmem_stem_padded_size: the frame layout the kernel reads -- NHWC with 4 channels (r, g, b, 0) inside a zero border: [images][Hp][Wp][4],
 *   pixel (y, x) at row y + 3, column x + 3; the caller zeroes the buffer ONCE, rmem_image_ptrs_to_nhwc4p writes only the interior.
 * rmem_stem7x7s2: y [images][Ho][Wo][64] = relu(conv + bias), Ho = (H - 1) / 2 + 1; w is [64][8][8][4] (ky, kx, c; zero where ky = 7,
 *   kx = 7 or c = 3), i.e. K = 256.  A persistent workgroup keeps the weights in registers and copies each 64-output window block
 *   (8 rows x 134 pixels) to LDS once; MFMA B fragments are read from it in place.  Same products as rmem_conv2d_nhwc on the 8-channel
 *   layout, summed in another order (fp32). */
int rmem_stem_padded_size(int H, int W, int* Hp, int* Wp);
int rmem_image_ptrs_to_nhwc4p(const float* const* img_ptrs, void* out_padded, int images, int H, int W, void* stream);
int rmem_stem7x7s2(const void* x_padded, int images, int H, int W, const void* w, const float* bias, void* y, void* stream);
/* The same followed by the 3x3 stride-2 pad-1 max-pool (encoders/resnet.py:136) in ONE pass: y_pooled [images][HP][WP][64], HP = (Ho - 1) / 2 + 1;
 * the half-resolution map is never written.  Bit-identical to rmem_stem7x7s2 + rmem_maxpool3x3s2_nhwc_images. */
int rmem_stem7x7s2_pool(const void* x_padded, int images, int H, int W, const void* w, const float* bias, void* y_pooled, void* stream);
/* 3x3 stride-1 pad-1 convolution with C input and C output channels (C = 64 or 128) + bias (+ ReLU), read in place from rows kept in
 * LDS with the weights in registers: the second conv of the ResNet layer-1 / layer-2 bottlenecks (encoders/resnet.py:52-56) and the
 * decoder's conv_4x (decoders/fpn.py:54-58).  x NHWC [images][H][W][C], w [C][3][3][C] (rmem_conv2d_nhwc's layout), y like x.
 * Bit-identical to rmem_conv2d_nhwc (same k order, same epilogue).  rmem_conv3x3_c64_direct = C 64 with ReLU. */
int rmem_conv3x3_direct(const void* x, int images, int H, int W, int C, const void* w, const float* bias, int relu, void* y, void* stream);
int rmem_conv3x3_c64_direct(const void* x, int images, int H, int W, const void* w, const float* bias, void* y, void* stream);
/* Frame ingest: decoded uint8 RGB [Hs][Ws][3] -> bicubic resize to the network size (OpenCV INTER_CUBIC semantics) ->
 * ImageNet normalise -> fp32 [3][Hd][Wd] (the engine API's input) and/or bf16 [Hd][Wd][8] (the encoder's input).
 * Replaces dataloaders/video_transforms.py:648-652 (cv2.resize) + 676-680 (normalise) on the host. */
int rmem_ingest_rgb8(const unsigned char* rgb_hwc, int Hs, int Ws, int Hd, int Wd, float* out_chw, void* out_nhwc8, void* stream);
/* 3x3 stride-2 pad-1 max-pool (encoders/resnet.py:105, 182). */
int rmem_maxpool3x3s2_nhwc(const void* x, void* y, int H, int W, int C, void* stream);
int rmem_maxpool3x3s2_nhwc_images(const void* x, void* y, int images, int H, int W, int C, void* stream);
/* bilinear resize, NHWC bf16 (decoders/fpn.py:49-52, 57-60). */
int rmem_bilinear_nhwc(const void* x, void* y, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream);
/* logits fp32 NHWC [Hi][Wi][ldl]: ids > keep_max_id forced to -1e10, bilinear to (Ho, Wo); writes any of
 * NCHW fp32 logits, uint8 argmax labels, fp32 argmax labels (engines/aot_engine.py:450-463;
 * managers/evaluator.py:430-441). */
int rmem_logits_post(const float* logits_nhwc, int ldl, int num_classes, int keep_max_id, int Hi, int Wi,
                     int Ho, int Wo, int align_corners, float* out_nchw, unsigned char* label_u8,
                     float* label_f32, void* stream);
/* images > 1: labels only (out_nchw must be NULL) */
int rmem_logits_post_images(const float* logits_nhwc, int images, int ldl, int num_classes, int keep_max_id, int Hi, int Wi,
                            int Ho, int Wo, int align_corners, float* out_nchw, unsigned char* label_u8,
                            float* label_f32, void* stream);
/* Identity-bank embedding straight from a label map, without the one-hot tensor (models/aot.py:139-147 patch_wise_id_bank applied to
 * one_hot(mask), engines/aot_engine.py:208-232): label [images][Hs][Ws] (uint8, or fp32 with label_is_f32) is resized to the network
 * size H x W (nearest, as rmem_label_to_onehot16 does) into the INTERIOR of label_scratch_u8 [images][Hpd][Wpd] (rmem_label_id_embed_scratch_size:
 * a border of `pad` pixels all round and one more row below, which the caller fills with 255 ONCE and which is never written), then
 *   out[pos][0..255] = bias + sum over the KH x KW window of pos of w[:, ky, kx, label]      (stride, pad as the conv; labels >= num_classes: 0)
 * with w [256][KH][KW][16] (rmem_conv2d_nhwc's layout of the Cin-padded weight).  The one-hot MFMA operand is built in registers from
 * the label bytes.  Same products as rmem_label_to_onehot16 + rmem_conv2d_nhwc, summed in another order (fp32). */
int rmem_label_id_embed_scratch_size(int H, int W, int pad, int* Hpd, int* Wpd);
int rmem_label_id_embed(const void* label, int label_is_f32, int images, int Hs, int Ws, int H, int W, int KH, int KW, int stride, int pad,
                        int num_classes, const void* w, const float* bias, void* label_scratch_u8, void* out, void* stream);

/* label map (uint8 or fp32) -> nearest resize -> one-hot + ignore channel, bf16 [Hd][Wd][16]
 * (utils/image.py:69-74; engines/aot_engine.py:208-224; managers/evaluator.py:518-522). */
int rmem_label_to_onehot16(const void* label, int label_is_f32, int Hs, int Ws, int Hd, int Wd,
                           int num_classes, void* out, void* stream);
int rmem_label_to_onehot16_images(const void* label, int label_is_f32, int images, int Hs, int Ws, int Hd, int Wd,
                                  int num_classes, void* out, void* stream);
/* scores[t] = sum_q mass[q][t] * (1 - softmax(bilinear_ac(logits -> He x We))[0])
 * (engines/aot_engine.py:355-362 + layers/transformer.py:341-351, the device half of the eviction policy).
 * `scores` must hold 32 + 64 * 32 floats: the first T are the result, the rest is reduction scratch. */
int rmem_evict_scores(const float* logits_nhwc, int ldl, int num_classes, int keep_max_id, int Hi, int Wi,
                      int He, int We, const float* attn_mass, int T, float* scores, void* stream);

/* Asynchronous copy on `stream` (device<->device, or pinned host<->device): frame ingest into the fixed input
 * buffer, bank append of curr_K (layers/transformer.py:319), chunk-table upload, eviction-score readback.
 * Capturable into a hipGraph (memcpy node). */
int rmem_copy_async(void* dst, const void* src, size_t bytes, void* stream);

/* Clips with more than 10 objects run one engine per 10 objects (engines/aot_engine.py:604-673).
 * rmem_split_label: engine e's label map = ids start_id..end_id renumbered from 1, everything else 0 (separate_mask, 610-628).
 * rmem_soft_logit_aggregate: soft_logit_aggregation (650-673) -- per engine softmax over its num_classes channels, background =
 * product of the engines' background probabilities, then the engines' objs_per_engine foreground channels concatenated;
 * clamp to [1e-5, 1 - 1e-5] and logit.  logits_nchw: HOST array of n_engines device pointers to [num_classes][H][W] fp32;
 * out: [1 + n_engines * objs_per_engine][H][W] fp32. */
int rmem_split_label(const float* label, int start_id, int end_id, float* out, long long n, void* stream);
int rmem_soft_logit_aggregate(const float* const* logits_nchw, int n_engines, int num_classes, int objs_per_engine, int H, int W,
                              float* out_nchw, void* stream);

/* fp32 planes [planes][Hs][Ws] -> optionally mirrored along W (utils/image.py:109-113 flip_tensor(dim 3)) -> nearest-neighbour resize
 * (F.interpolate(mode='nearest') index rule: src = floor(dst * in / out)) to [planes][Hd][Wd]: the frames and label maps the
 * evaluator hands to its flipped-augmentation engines, in the reference's order flip, then resize (managers/evaluator.py:342-355,
 * 490-522). */
int rmem_resize_nearest_flip_f32(const float* src, int planes, int Hs, int Ws, float* dst, int Hd, int Wd, int flip_w, void* stream);

/* Test-time-augmentation merge: softmax of each augmentation's NCHW logits (read horizontally flipped where flips[a] != 0),
 * mean over the <= 8 augmentations (scales x flips), argmax; writes any of uint8 labels, fp32 labels, NCHW mean probabilities.
 * logits_nchw / flips are HOST arrays of n_aug entries.  Replaces managers/evaluator.py:427-441 (flip / multi-scale TTA). */
int rmem_tta_merge(const float* const* logits_nchw, const int* flips, int n_aug, int num_classes, int H, int W,
                   unsigned char* label_u8, float* label_f32, float* prob_nchw, void* stream);

/* Region-similarity (Jaccard) counts per object id for one mask pair: counts[2*id] += |pred==id & gt==id|,
 * counts[2*id+1] += |pred==id | gt==id| over the n pixels whose ground truth is not `void_label`; the caller zeroes
 * counts (uint64 [2 * num_ids]).  Replaces evaluation/source/metrics.py:6-37 (db_eval_iou) per object. */
int rmem_mask_iou_counts(const unsigned char* pred, const unsigned char* gt, long long n, int num_ids, int void_label,
                         unsigned long long* counts, void* stream);

/* ------------------------------------------------------------------ stream capture helpers
 * Thin wrappers over hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch so the Python host
 * can replay one frame's launch sequence as a hipGraph. */
int rmem_graph_begin(void* stream);
int rmem_graph_end(void* stream, void** graph_exec_out);
int rmem_graph_launch(void* graph_exec, void* stream);
int rmem_graph_destroy(void* graph_exec);


/* ------------------------------------------------------------------ IEEE-half flavour
 * Every entry point above that takes 16-bit operands ("bf16" pointers: activations, weights, bank entries) exists a second time
 * with the suffix _f16: identical signature, layout and semantics, but the 16-bit element type is IEEE binary16 instead of
 * bfloat16 (accumulation, residual streams and all statistics stay fp32).  This is the operand type of the reference's --amp
 * path (tools/eval.py:45-47: torch.cuda.amp.autocast) and of BASELINE cfg 5; with 11 significant bits instead of 8 it is also the
 * flavour that reaches >= 0.999 mask IoU against the fp32 reference.  Differences: the memory-read kernel
 * always runs its online-softmax pass (P = exp2(S - m) must stay below 2^16), and values beyond +-65504 overflow, as under
 * autocast.  Element-type-agnostic entry points (workspace sizes, fp32 / uint8 post-processing, copies, graphs, timers) have no twin. */
int rmem_conv2d_nhwc_f16(const rmem_conv_desc* desc, const void* x, const void* w, const float* bias, const void* residual, void* y, void* y2, void* workspace, void* stream);
int rmem_mem_read_attn_f16(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride, int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_single, const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out, int ldo, float* attn_mass, int T, void* workspace, void* stream);
int rmem_mem_read_attn_clips_f16(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride, int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_single, const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out, int ldo, float* attn_mass, int T, int nclips, long long q_clip_stride, long long kv_clip_stride, long long out_clip_stride, void* workspace, void* stream);
int rmem_lstt_attn_pair_clips_f16(const void* q, int ldq, const void* k_bank, const void* v_bank, long long slot_stride, int ldkv, const rmem_attn_chunk* chunks, int nchunks, int lk_total, const float* pe_cur, const float* pe_mem, int Lq, int heads, void* out_long, int ldo, float* attn_mass, int T, int nclips, long long q_clip_stride, long long out_clip_stride, const void* k_short, const void* v_short, int lk_short, long long kv_short_clip_stride, void* out_short, long long out_short_clip_stride, void* workspace, void* stream);
int rmem_layernorm256_f16(const void* a, int a_is_f32, int lda, const void* b, int b_is_f32, int ldb, const float* gamma, const float* beta, float eps, int M, void* y_bf16, int ldy, const float* pos, void* ypos_bf16, int ldyp, float* y_f32, int ldyf, void* stream);
int rmem_layernorm_f16(const void* a, int a_is_f32, int lda, const float* gamma, const float* beta, float eps, int M, int C, void* y_bf16, int ldy, float* y_f32, int ldyf, void* stream);
int rmem_patch_merge_ln_f16(const float* x, int H, int W, int C, const float* gamma, const float* beta, float eps, void* y_bf16, void* stream);
int rmem_window_attn_f16(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int H, int W, int C, int heads, int shift, void* stream);
int rmem_window_attn_images_f16(const void* qkv, const float* qkv_bias, const float* bias_mask_table, void* out, int images, int H, int W, int C, int heads, int shift, void* stream);
int rmem_patch_merge_ln_images_f16(const float* x, int images, int H, int W, int C, const float* gamma, const float* beta, float eps, void* y_bf16, void* stream);
int rmem_add16_f16(const void* a, const void* b, void* y, long long n, void* stream);
int rmem_add16_grouped_f16(int n, const void* const* a, const void* const* b, void* const* y, long long count, void* stream);
int rmem_lstt_chain_a_f16(const rmem_chain_a_desc* d, void* stream);
int rmem_lstt_chain_b_f16(const rmem_chain_b_desc* d, void* stream);
int rmem_lstt_chain_c_f16(const rmem_chain_c_desc* d, void* stream);
int rmem_layernorm256_pair_f16(const void* a0, const void* b0, void* y0, const void* a1, const void* b1, void* y1, const float* gamma, const float* beta, float eps, int M, void* stream);
int rmem_bneck_chain_f16(const rmem_bneck_chain_desc* desc, const void* b, const void* x2, const void* w3, const float* bias3, const void* res, void* y, const void* w1, const float* bias1, void* a2, void* stream);
int rmem_conv1x1_dual_nhwc_f16(const rmem_conv_desc* desc, const void* x, const void* x2, int H2, int W2, int Cin2, int stride2, const void* w_cat, const float* bias, void* y, void* stream);
int rmem_linear_grouped_f16(const rmem_conv_desc* desc, int n, const void* const* x, const void* const* w, const float* const* bias, const void* const* residual, void* const* y, void* stream);
int rmem_groupnorm_nhwc_f16(const void* x, int M, int C, int groups, const float* gamma, const float* beta, float eps, int act, void* y, float* workspace, void* stream);
int rmem_groupnorm_f32_nhwc_f16(const float* x, int M, int C, int groups, const float* gamma, const float* beta, float eps, int act, void* y, float* workspace, void* stream);
int rmem_groupnorm_nhwc_images_f16(const void* x, int images, int M, int C, int groups, const float* gamma, const float* beta, float eps, int act, void* y, float* workspace, void* stream);
int rmem_groupnorm_head_nhwc_images_f16(const void* x, int images, int M, int C, int groups, const float* gamma, const float* beta, float eps, int act, const void* w, const float* bias, int N, float* y, int ldy, float* workspace, void* stream);
int rmem_gn_act_dwconv5x5_nhwc_images_f16(const void* x, int images, int H, int W, int C, int groups, const float* gamma, const float* beta, float eps, int act, const float* w_t, void* y, float* workspace, void* stream);
int rmem_gn_act_dwconv5x5_prestats_nhwc_images_f16(const void* x, int images, int H, int W, int C, int groups, const float* gamma, const float* beta, float eps, int act, const float* w_t, void* y, const float* stats, void* stream);
int rmem_gn_act_dwconv5x5_nhwc_f16(const void* x, int H, int W, int C, int groups, const float* gamma, const float* beta, float eps, int act, const float* w_t, void* y, float* workspace, void* stream);
int rmem_dwconv5x5_nhwc_f16(const void* x, const float* w_t, void* y, int H, int W, int C, void* stream);
int rmem_image_ptrs_to_nhwc4p_f16(const float* const* img_ptrs, void* out_padded, int images, int H, int W, void* stream);
int rmem_conv3x3_direct_f16(const void* x, int images, int H, int W, int C, const void* w, const float* bias, int relu, void* y, void* stream);
int rmem_conv3x3_c64_direct_f16(const void* x, int images, int H, int W, const void* w, const float* bias, void* y, void* stream);
int rmem_stem7x7s2_pool_f16(const void* x_padded, int images, int H, int W, const void* w, const float* bias, void* y_pooled, void* stream);
int rmem_stem7x7s2_f16(const void* x_padded, int images, int H, int W, const void* w, const float* bias, void* y, void* stream);
int rmem_image_ptrs_to_nhwc8_f16(const float* const* img_ptrs, void* out, int images, int H, int W, void* stream);
int rmem_image_to_nhwc8_f16(const float* img_chw, void* out, int H, int W, void* stream);
int rmem_image_to_nhwc8_images_f16(const float* img_chw, void* out, int images, int H, int W, void* stream);
int rmem_ingest_rgb8_f16(const unsigned char* rgb_hwc, int Hs, int Ws, int Hd, int Wd, float* out_chw, void* out_nhwc8, void* stream);
int rmem_maxpool3x3s2_nhwc_f16(const void* x, void* y, int H, int W, int C, void* stream);
int rmem_maxpool3x3s2_nhwc_images_f16(const void* x, void* y, int images, int H, int W, int C, void* stream);
int rmem_bilinear_nhwc_f16(const void* x, void* y, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream);
int rmem_bilinear_nhwc_images_f16(const void* x, void* y, int images, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, void* stream);
int rmem_label_id_embed_f16(const void* label, int label_is_f32, int images, int Hs, int Ws, int H, int W, int KH, int KW, int stride, int pad, int num_classes, const void* w, const float* bias, void* label_scratch_u8, void* out, void* stream);
int rmem_label_to_onehot16_f16(const void* label, int label_is_f32, int Hs, int Ws, int Hd, int Wd, int num_classes, void* out, void* stream);
int rmem_label_to_onehot16_images_f16(const void* label, int label_is_f32, int images, int Hs, int Ws, int Hd, int Wd, int num_classes, void* out, void* stream);
int rmem_gated_attn_f16(const void* q, int ldq, const void* k_bank, long long k_slot_stride, int ldk, const void* v_bank, long long v_slot_stride, int ldv, const rmem_attn_chunk* chunks, int nchunks, int frames, int keys_per_frame, const float* pe_cur, const float* pe_mem, int Lq, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit, void* out, int ldo, float* attn_mass, const float* dw_w_t, int H, int W, void* workspace, void* stream);
int rmem_local_gated_attn_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* rel, int ldrel, int H, int W, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit, void* out, int ldo, const float* dw_w_t, void* workspace, void* stream);
int rmem_gated_attn_clips_f16(const void* q, int ldq, const void* k_bank, long long k_slot_stride, int ldk, const void* v_bank, long long v_slot_stride, int ldv, const rmem_attn_chunk* chunks, int nchunks, int frames, int keys_per_frame, const float* pe_cur, const float* pe_mem, int Lq, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit, void* out, int ldo, float* attn_mass, const float* dw_w_t, int H, int W, int nclips, void* workspace, void* stream);
int rmem_local_gated_attn_clips_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* rel, int ldrel, int H, int W, int DV, const void* u_a, int ldua, const void* u_b, int ldub, int usplit, void* out, int ldo, const float* dw_w_t, int nclips, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RMEM_H_ */
