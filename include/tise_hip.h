/*
 * libtise_hip.so -- C ABI of the MI355X (gfx950) image-realism hot path of the TISE toolbox.
 *
 * The reference (VinAIResearch/tise-toolbox) has no native code and no FFI: its statistics
 * layer is numpy/scipy called from Python.  Each entry point below replaces the numpy/scipy
 * (or Pillow) call cited next to it; the Python shims in tise_toolbox_amd/ (same function
 * names as the reference) are the only callers and bind these symbols with ctypes
 * (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer named *_dev is DEVICE memory (hipMalloc or a torch CUDA tensor's
 *     data_ptr()); everything else is host memory.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  All work is enqueued
 *     on it; nothing synchronises unless the comment says so.
 *   - return value: 0 = TISE_OK, negative = error (tise_status_string() names it).  No C++
 *     exception crosses the boundary.  Data-dependent conditions (rank deficiency, non-finite
 *     input) are reported through device-side flag words, not through the return code.
 *   - no hidden allocation of caller-visible memory: scratch lives in handles created and
 *     destroyed explicitly.
 */
#ifndef TISE_HIP_H
#define TISE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TISE_OK 0
#define TISE_ERR_INVALID_ARG (-1)   /* -> AssertionError / ValueError in the Python mirror   */
#define TISE_ERR_HIP (-2)           /* a HIP runtime call failed (tise_last_hip_error())     */
#define TISE_ERR_NO_DEVICE (-3)     /* no gfx950 device visible                              */
#define TISE_ERR_UNSUPPORTED (-4)   /* size outside what the kernels are built for          */

const char* tise_status_string(int status);
int tise_last_hip_error(void);                 /* hipError_t of the last failing HIP call    */
int tise_version(void);                        /* ABI version, currently 1                   */
int tise_device_info(int* cu_count, int* gcn_arch_is_gfx950, size_t* total_mem);

/* ------------------------------------------------------------------------------------------
 * (a2) Host feed.  The reference hands decoded images from 8 DataLoader worker processes to the main process through
 * pickling queues as fp32 tensors (image_realism/FID/fid_score.py:215-217, img_data.py:19-25).  Here the decode workers
 * write uint8 pixels into one shared-memory ring owned by the caller (tise_toolbox_amd/png_ring.py); these three calls
 * page-lock that ring once and enqueue host->device copies straight from it.
 *   tise_host_register    hipHostRegister of caller-owned host memory (the memory stays the caller's)
 *   tise_host_unregister  before the caller unmaps it
 *   tise_memcpy_h2d_async enqueue on `stream`; asynchronous when src_host is page-locked (registered)
 * ------------------------------------------------------------------------------------------ */
int tise_host_register(void* host_ptr, size_t bytes);
int tise_host_unregister(void* host_ptr);
int tise_memcpy_h2d_async(void* dst_dev, const void* src_host, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a2) PNG row filters on the device.  Replaces the second half of ``Image.open(f).convert("RGB")`` of
 * Dataset.__getitem__ (image_realism/FID/img_data.py:19-25; third-party Pillow: zlib inflate, then the five row filters
 * of RFC 2083 section 6).  The feed's decode processes only inflate a file into a ring slot (libtise_png.so,
 * csrc/png_decode.c: tise_png_inflate_slot); the slots are copied to HBM as they are and this call reconstructs the
 * pixels.  Slot = [64-byte header | payload]; header byte 0: 0 = payload is h*w*3 RGB pixels (decoded on the host: copied),
 * 3 / 4 = payload is h rows of (1 filter-type byte + w*3 / w*4 filtered bytes) -- None, Sub, Up, Average, Paeth are
 * reversed and the alpha byte of RGBA is dropped (Pillow's convert("RGB") of an RGBA PNG: no blending).
 *   slots_dev    n slots, slot_stride bytes apart (multiple of 4, 4-byte aligned base)
 *   dst_dev      (n, h, w, 3) uint8, the layout tise_resize_u8 reads
 * Bit-exact against Pillow (tests/test_gpu_png.py).  The filter-type bytes are trusted to be <= 4 (the host checks).
 * ------------------------------------------------------------------------------------------ */
int tise_png_unfilter_rgb8(const uint8_t* slots_dev, int64_t n, int64_t slot_stride, int h, int w, uint8_t* dst_dev,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * (a3) PIL-exact uint8 bilinear resize + ToTensor + input affine, fused.
 * Replaces transforms.Resize((299,299)) + ToTensor()   image_realism/FID/fid_score.py:208-213
 * (Pillow ImagingResample, 8bpc: 22-bit fixed-point coefficients, horizontal pass -> u8 ->
 * vertical pass -> u8) and the per-channel affine of image_realism/FID/inception.py:120-124.
 *
 *   src_dev   n images, HWC uint8, contiguous (n, h, w, 3)
 *   dst_dev   fp32, (n, 3, oh, ow) when nhwc == 0, (n, oh, ow, 3) when nhwc != 0; NULL: only u8_out_dev is produced
 *             (the latter is torch.channels_last storage of an NCHW tensor)
 *   lut       host pointer, 3*256 floats: lut[c*256 + v] = value written for channel c and
 *             resized byte v (the shim fills it with fp32(v)/255 then the inception.py affine,
 *             evaluated with the reference's own op order, so the kernel is exact by table).
 *   u8_out_dev optional (may be NULL): the resized uint8 image (n, oh, ow, 3), for parity tests.
 * ------------------------------------------------------------------------------------------ */
int tise_resize_bilinear_u8(const uint8_t* src_dev, int n, int h, int w,
                            float* dst_dev, int oh, int ow, int nhwc,
                            const float* lut, uint8_t* u8_out_dev, void* stream);
/* The same kernel with Pillow's BICUBIC filter (filter = 1; 0 = BILINEAR): replaces, for the CLIP metrics, the
 * Resize(224, interpolation=BICUBIC) + ToTensor + Normalize of clip.load's preprocess (third-party `clip`,
 * text_relevance/RP_coco.py:31,64; positional_alignment/PA.py:30,34) -- the table carries (v/255 - mean)/std;
 * dst_dev planar (n, 3, oh, ow) with nhwc == 0.  Square inputs need no CenterCrop. */
int tise_resize_u8(const uint8_t* src_dev, int n, int h, int w, float* dst_dev, int oh, int ow, int nhwc,
                   const float* lut, uint8_t* u8_out_dev, int filter, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a7) Streaming activation statistics: n, s = sum_i x_i, S = sum_i x_i x_i^T in fp64 from
 * fp32 feature rows that never leave the device.
 * Replaces pred_arr[start:end] = pred.cpu()...  + np.mean(act, 0) + np.cov(act, rowvar=False)
 *                                           image_realism/FID/fid_score.py:98,113,194-195
 * ------------------------------------------------------------------------------------------ */
typedef struct tise_stats tise_stats_t;

int tise_stats_create(int d, tise_stats_t** h);           /* allocates (d*d + d + 2) doubles, zeroed */
int tise_stats_destroy(tise_stats_t* h);
int tise_stats_reset(tise_stats_t* h, void* stream);
/* accumulate `rows` feature rows: feats_dev[r*ld + c], c < d.  Callable once per batch.  ONE launch when d % 64 == 0
 * and the rows are 16-byte aligned (the covariance kernel's diagonal-tile workgroups also fold the column sums and
 * the row count); otherwise the two halves below, one after the other. */
int tise_stats_update(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld, void* stream);
/* the two halves of tise_stats_update, separately launchable:
 * _cov: S += X^T X (fp64 MFMA)   _sum: s += column sums, n += rows (column tiles x 8 row slices, fixed-order merge) */
int tise_stats_update_cov(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld, void* stream);
int tise_stats_update_sum(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld, void* stream);
/* GROUPED accumulation (per-class O-FID, BASELINE configs[4]): feats_dev holds rows sorted by group; group g = rows
 * [row_offsets[g], row_offsets[g + 1]) (HOST array of n_groups + 1 offsets) is folded into handles[g] -- all groups in ONE
 * launch (grid.y = group), so every class's S is read-modify-written once per call instead of once per class and batch. */
int tise_stats_update_grouped(tise_stats_t* const* handles, int n_groups, const float* feats_dev, const int64_t* row_offsets,
                              int64_t ld, void* stream);
/* the contiguous fp64 buffer [S (d*d, upper triangle by 64x64 tile) | s (d) | n | pad] that a
 * data-parallel caller hands to RCCL / torch.distributed.all_reduce(SUM). */
int tise_stats_buffer(tise_stats_t* h, double** buf_dev, size_t* n_doubles);
/* mu = s/n ; sigma = (S - s s^T / n) / (n - 1)   (np.cov ddof = 1).  mu_dev: d, sigma_dev: d*d */
int tise_stats_finalize(tise_stats_t* h, double* mu_dev, double* sigma_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a6) Frechet distance.
 * Replaces calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps)
 *                                           image_realism/FID/fid_score.py:121-171
 * Tr sqrtm(S1 S2) is evaluated as sum_i sqrt(lambda_i(L^T S2 L)), S1 = L L^T: an unpivoted blocked
 * Cholesky factor when every pivot is safely positive, otherwise the diagonally pivoted one
 * (rank-revealing, runs to the last positive pivot); L^T S2 L reduced to tridiagonal form by
 * Householder reflections, eigenvalues by Sturm bisection; all fp64.
 *
 *   out_dev[0] = fid            out_dev[1] = tr sqrt(S1 S2)     out_dev[2] = |mu1-mu2|^2
 *   out_dev[3] = tr S1          out_dev[4] = tr S2              out_dev[5] = rank(L) as double
 *   out_dev[6] = number of negative eigenvalues clipped         out_dev[7] = flags as double
 *   flags: bit0 = non-finite value met (the reference's "singular product" branch, :156-160,
 *          is taken by the Python mirror on this bit), bit1 = sigma1 numerically rank deficient.
 * ------------------------------------------------------------------------------------------ */
typedef struct tise_frechet tise_frechet_t;

#define TISE_FRECHET_OUT_DOUBLES 8
#define TISE_FLAG_NONFINITE 1
#define TISE_FLAG_RANK_DEFICIENT 2

int tise_frechet_create(int d, tise_frechet_t** h);       /* allocates ~4 d*d doubles of scratch */
int tise_frechet_destroy(tise_frechet_t* h);
/* Synchronises `stream` once (to read the numerical rank back). */
int tise_frechet_distance(tise_frechet_t* h, const double* mu1_dev, const double* sigma1_dev,
                          const double* mu2_dev, const double* sigma2_dev, double diag_offset,
                          double* out_dev, void* stream);
/* Two-step form.  Tr sqrtm(S1 S2) is symmetric in its arguments and the factor of ONE covariance does not depend on
 * the other: when one side's statistics are known early -- the reference .npz of fid_score.py:200-203 -- its pivoted
 * Cholesky (tise_frechet_prefactor, any stream; synchronises THAT stream to read the rank) can overlap the network
 * passes of the other side, and tise_frechet_distance_prefactored (which waits for the factor through an event) leaves
 * only GEMM + tridiagonalisation + bisection after the last batch.  (mu_f, sigma_f) = the factored side, sigma_f the
 * same matrix that was given to tise_frechet_prefactor.  No diag_offset: on TISE_FLAG_NONFINITE the caller falls back
 * to tise_frechet_distance with the reference's eps (fid_score.py:156-160). */
int tise_frechet_prefactor(tise_frechet_t* h, const double* sigma_dev, void* stream);
int tise_frechet_distance_prefactored(tise_frechet_t* h, const double* mu_f_dev, const double* sigma_f_dev,
                                      const double* mu_o_dev, const double* sigma_o_dev, double* out_dev, void* stream);
int tise_frechet_prefactor_ms(tise_frechet_t* h, double* ms_host);   /* HIP-event duration of the last prefactor */
/* Optional phase timing of tise_frechet_distance with HIP events on the caller's stream.
 * ms_host[5] = Cholesky | GEMMs | tridiagonalisation | bisection | final reduction.
 * tise_frechet_phase_ms waits for the last recorded call to finish. */
int tise_frechet_set_profiling(tise_frechet_t* h, int on);
int tise_frechet_phase_ms(tise_frechet_t* h, double* ms_host, int* rank_host);
/* Test hook: eigenvalues (ascending, n of them) of the symmetric n x n matrix a_dev (ld = n),
 * through the same tridiagonalisation + bisection kernels.  a_dev is not modified. */
int tise_eigvalsh(tise_frechet_t* h, const double* a_dev, int n, double* w_dev, void* stream);
/* Test hook: pivoted Cholesky of sigma_dev (d x d).  lt_dev receives L^T (d x d, row k = column k
 * of L, rows >= rank zero); rank_host receives the numerical rank.  Synchronises the stream. */
int tise_pivoted_cholesky(tise_frechet_t* h, const double* sigma_dev, double* lt_dev,
                          int* rank_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a8, a8', a8'') IS* reduction with temperature.
 * Replaces tf.div(logits, T); tf.nn.softmax            IS/coco/inception_score_star_coco.py:107-108
 *      and the split / KL / exp loop                    IS/coco/inception_score_star_coco.py:52-60
 *          (same loop: IS/bird/inception_score_star_bird.py:97-108, slice+T :189-194;
 *           per-row entropy form: O-IS/object_centric_inception_score.py:69-81)
 * One-pass additive form: per split k keep A_k = sum_i sum_c p_ic log p_ic and
 * B_kc = sum_i p_ic; score_k = exp(A_k/n_k - sum_c pbar_c log pbar_c).
 *
 *   split_rule 0: split k = [k*N/splits, (k+1)*N/splits)          (coco, bird)
 *   split_rule 1: split k = [k*(N/splits), (k+1)*(N/splits)), the tail is dropped   (O-IS)
 *   drop_first != 0: class 0 is discarded before the softmax       (bird, :189)
 *   acc_dev: splits * (1 + C_eff) doubles, [A_0..A_{splits-1} | B row-major]; C_eff = C - drop.
 *            Additive across calls and across GPUs (all_reduce SUM).  Caller zeroes it.
 *   ws_dev:  scratch, at least 2*rows doubles.
 *   Rows carry global indices idx_base .. idx_base + rows - 1 of an N_total-image set.
 * ------------------------------------------------------------------------------------------ */
int tise_is_update(const float* logits_dev, int64_t rows, int64_t ld, int C, double temperature,
                   int drop_first, int64_t idx_base, int64_t n_total, int splits, int split_rule,
                   double* acc_dev, double* ws_dev, void* stream);
/* out_dev: 2 + splits doubles = [mean, std (ddof 0), score_0 .. score_{splits-1}] */
int tise_is_finalize(const double* acc_dev, int C_eff, int64_t n_total, int splits, int split_rule,
                     double* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a5, epilogues) Fused channels-last fp32 epilogues around the MIOpen convolutions of the
 * InceptionV3 trunk (image_realism/FID/inception.py:59-95 via torchvision BasicConv2d = conv ->
 * BatchNorm(eval) -> ReLU, InceptionA/C/E branch_pool = avg_pool2d(3,1,1) -> 1x1 conv,
 * InceptionB/D branch_pool and the stem = max_pool2d(3,2), torch.cat of the branches).
 * BatchNorm is folded into the conv weights and the per-channel bias by the caller.
 * All tensors are NHWC fp32; x is addressed as x[pixel*x_ld + x_off + c] and out as
 * out[pixel*out_ld + out_off + c], c < C, so a kernel can read a channel slice of a fused 1x1
 * conv output and write straight into the channel slice of a block's concatenated output.
 * C, offsets and leading dimensions must be multiples of 4.
 * ------------------------------------------------------------------------------------------ */
/* out = max(x + bias, 0); may run in place (out == x, same ld/off). */
int tise_bias_relu_nhwc(const float* x_dev, int64_t x_ld, int x_off, int64_t pixels, int C,
                        const float* bias_dev, float* out_dev, int64_t out_ld, int out_off, void* stream);
/* out = max(avgpool3x3(stride 1, pad 1, count_include_pad)(x) + bias, 0) over an (n,h,w) grid. */
int tise_avgpool3_bias_relu_nhwc(const float* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C,
                                 const float* bias_dev, float* out_dev, int64_t out_ld, int out_off, void* stream);
/* out (n, (h-3)/2+1, (w-3)/2+1) = maxpool3x3(stride 2)(x); with bias_dev != NULL the input is a raw
 * conv output and max(. + bias, 0) is applied (ReLU and max commute). */
int tise_maxpool3s2_nhwc(const float* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C,
                         const float* bias_dev, float* out_dev, int64_t out_ld, int out_off, void* stream);

/* Split-fp16 activations (v ~= hi + lo * 2^-11, the format tise_conv_split_f16 consumes and produces).
 * LAYOUT of a split tensor of C channels (C % 16 == 0): NHWC; inside a pixel the channels come in blocks of 32 with
 * the two halves of a block side by side -- [hi c0..c31 | lo c0..c31][hi c32..c63 | lo c32..c63]... (128 bytes per
 * block) -- followed, when C % 32 == 16, by one 64-byte block [hi x16 | lo x16].  A pixel is 4*C bytes, the same as
 * fp32.  `ld` arguments of split tensors are the tensor's channel count C, `off` the first channel of a slice.
 *
 * out = split(max(avgpool3x3(stride 1, pad 1, count_include_pad)(x) + bias, 0)): x is raw fp32 (n, h, w, x_ld);
 * C % 8 == 0, x_ld / x_off multiples of 4 floats, bias_dev 16-byte aligned. */
int tise_avgpool3_bias_relu_split_nhwc(const float* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C,
                                       const float* bias_dev, void* out_dev, int64_t out_ld, int out_off,
                                       void* stream);
/* 3x3 / stride 2 max pool, split tensor -> split tensor (channel slice [out_off, out_off + C) of out; C % 8 == 0). */
int tise_maxpool3s2_split_nhwc(const void* x_dev, int64_t x_ld, int x_off, int n, int h, int w,
                               int C, void* out_dev, int64_t out_ld, int out_off, void* stream);

/* Stem layer Conv2d_1a_3x3 (3 -> 32, 3x3, stride 2) from the fp32 NHWC input (n, h, w, 3), folded bias +
 * ReLU + fp16 split fused: out = split tensor (n, oh, ow, 32).  w_dev: [kh][kw][cin][cout] fp32 (27 x 32). */
int tise_stem_conv3x3s2_split(const float* x_dev, int n, int h, int w, const float* w_dev, const float* bias_dev,
                              void* out_dev, void* stream);
/* The same layer from the uint8 NHWC result of tise_resize_bilinear_u8 (called with dst_dev = NULL and u8_out_dev set):
 * lut_dev is the 3 x 256 fp32 table (device copy) that ToTensor + the inception.py:120-124 affine tabulate, applied
 * while loading; bit-identical to the fp32 entry point on the table's values. */
int tise_stem_conv3x3s2_split_u8(const uint8_t* x_dev, const float* lut_dev, int n, int h, int w, const float* w_dev,
                                 const float* bias_dev, void* out_dev, void* stream);
/* Round 3: the same layer on the matrix cores (K = 27 padded to 32 = one K-step of the split-precision scheme).
 * wsplit_dev: fp16 [2][32 couts][32 k] (hi plane, lo plane) of the BatchNorm-folded weights, each cout pre-scaled by a
 * power of two, in the kernel's K order: k = 16 (u / 8) + 8 h + u % 8 for half h in {0, 1} and slot u in 0..15, where
 * (h = 0, u < 9) = tap (kh 0, t = u), (0, u >= 9) = (kh 1, t = u - 9), (1, u < 9) = (kh 2, t = u), (1, u = 9, 10) = (kh 1,
 * t = 7, 8), all other slots zero; t = 3 kw + cin.  scale_dev[32] undoes the pre-scaling, bias_dev[32]; w >= 5, x_dev
 * 4-byte aligned.  Same values as the entry point above to split-precision accuracy (not the same bits). */
int tise_stem_conv3x3s2_split_u8_mfma(const uint8_t* x_dev, const float* lut_dev, int n, int h, int w, const void* wsplit_dev,
                                      const float* scale_dev, const float* bias_dev, void* out_dev, void* stream);
/* Global average of a split tensor (n, hw, C), C % 32 == 0 -> fp32 (n, C): AdaptiveAvgPool2d((1,1)) of the last block. */
int tise_split_mean_nhwc(const void* x_dev, int n, int hw, int C, float* out_dev, void* stream);
/* the same mean, also written as a split row (n, 2C) fp16 (layout above): the operand of the classifier layer, which runs
 * as a 1x1 split-precision convolution on it (the IS* logits: pool3 x W, inception_score_star_coco.py:104-105) */
int tise_split_mean_both_nhwc(const void* x_dev, int n, int hw, int C, float* out_dev, void* out_split_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a11, section 8 f3) Top-1 text retrieval for R-precision and the 2-way softmax test of PA.
 * Replaces the tail of the per-item loop of text_relevance/RP_coco.py:72-78 (logits_per_image ->
 * softmax -> argmax == 0) and positional_alignment/PA.py:37-42 (softmax[0] > 0.6) for n items at once.
 * img_emb: (n, d); txt_emb: a table of DISTINCT caption embeddings (rows, d); txt_index: (n, c) int32 rows of
 * that table, candidate 0 = the true caption (NULL: item i's candidates are table rows i*c .. i*c+c-1).
 * dtype 0 = fp32, 1 = fp16 embeddings; normalize != 0 divides by both norms (cosine), 0 takes the dot product
 * of already normalised features as CLIP.forward does.  Every intermediate CLIP.forward stores (normalised and
 * scaled image features, logits) and the softmax output are rounded to the embeddings' dtype, fp32 arithmetic
 * inside, as torch does for the fp16 model clip.load serves on a GPU: top1_out[i] = FIRST maximum of the rounded
 * softmax (np.argmax of the reference: near-ties go to candidate 0), p0_out (nullable) = its entry 0.  c <= 1024.
 * ------------------------------------------------------------------------------------------ */
int tise_cosine_top1(const void* img_emb_dev, const void* txt_emb_dev, const int32_t* txt_index_dev, int64_t n, int c,
                     int d, int dtype, int normalize, float logit_scale, int32_t* top1_out_dev, float* p0_out_dev,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * (a11, section 8 f3) The CLIP ViT-B/32 towers (third-party `clip`, called per item by text_relevance/RP_coco.py:56-80
 * and positional_alignment/PA.py:33-43), batched: fp16 tensors, fp32 accumulation and statistics -- the arithmetic of
 * the fp16 model `clip.load` serves on a GPU.  Row-major matrices with explicit leading dimensions (elements).
 *   tise_gemm_f16        out[m][n] = act(sum_k a[m][k] w[n][k] + bias[n]) + residual[m][n]     (nn.Linear layout of w;
 *                        act 0 = none, 1 = QuickGELU x*sigmoid(1.702x); bias / residual nullable; k % 64 == 0,
 *                        n % 8 == 0, leading dimensions % 8 == 0)
 *   tise_layernorm_f16   per row over C <= 1024 columns (C, ldx, ldo multiples of 8; 16-byte aligned pointers), fp32 mean / variance (clip.model.LayerNorm)
 *   tise_attention_f16   qkv [batch*seq][3*heads*64] (q | k | v) -> out [batch*seq][heads*64], softmax(q k^T / 8) v per
 *                        (sequence, head), optional causal mask (text tower); seq <= 80, head_dim == 64
 *   tise_patchify_f16    image (batch, 3, res, res) NCHW -> [batch*(res/patch)^2][3*patch*patch], columns in the order of
 *                        conv1.weight.flatten(1): the patch embedding becomes one tise_gemm_f16
 *   tise_vit_tokens_f16  x[b][0] = class_emb + pos[0]; x[b][1+p] = patch_out[b*n_patches+p] + pos[1+p]
 *   tise_text_tokens_f16 x[r] = table[tokens[r]] + pos[r % seq]
 *   tise_gather_rows_f16 out[i] = x[index[i]]   (class token of every image / end-of-text token of every caption)
 * ------------------------------------------------------------------------------------------ */
int tise_gemm_f16(const void* a_dev, int64_t lda, const void* w_dev, int64_t ldw, const void* bias_dev, const void* res_dev,
                  int64_t ldr, void* out_dev, int64_t ldo, int m, int n, int k, int act, void* stream);
int tise_layernorm_f16(const void* x_dev, int64_t ldx, const void* gamma_dev, const void* beta_dev, void* out_dev, int64_t ldo,
                       int64_t rows, int C, float eps, void* stream);
int tise_attention_f16(const void* qkv_dev, int batch, int seq, int heads, int head_dim, int causal, void* out_dev, void* stream);
int tise_patchify_f16(const void* img_dev, int batch, int res, int patch, void* out_dev, void* stream);
int tise_vit_tokens_f16(const void* patch_out_dev, const void* class_emb_dev, const void* pos_emb_dev, int batch, int n_patches,
                        int width, void* x_dev, void* stream);
int tise_text_tokens_f16(const int32_t* tokens_dev, const void* table_dev, const void* pos_emb_dev, int64_t rows, int seq, int width,
                         void* x_dev, void* stream);
int tise_gather_rows_f16(const void* x_dev, const int64_t* index_dev, int64_t n, int width, void* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * (a5, convolution) Implicit-GEMM convolution on fp16 MFMA with 3-term split-precision operands.
 * Replaces the Conv2d + BatchNorm(eval) + ReLU of torchvision's BasicConv2d for split tensors (layout above;
 * v ~= hi + lo * 2^-11): D = max(scale * conv(x, w) + bias, 0), re-split and written
 * into up to four destination channel slices (mode 0), or raw fp32 scale*conv (mode 1, pool branch).
 * `args` points to a host-side tise_conv_args (copied into the launch).  `tn` = tile width | kernel variant:
 * the low four bits give the output tile, 128 pixels x 32*tn channels with tn in {1..5}; weights and
 * scale/bias must be padded to a multiple of 32*tn rows and of 32 in K.
 * Variant bits (same arithmetic; 16 and 128 give bit-identical results when Cin % 32 == 0):
 *   16   direct-to-LDS DMA, two stages, addresses recomputed per K-step (generic form: the reference kernel of
 *        the tests, also serves M >= 2^31); weights fp16 [2][Cout_pad][Kpad] (hi plane, lo plane w_plane further)
 *        K = (kh, kw, cin) with cin fastest.
 *   128  DMA with the address arithmetic hoisted out of the K loop (the default of the Python layer; M < 2^31,
 *        H, W < 16128); weights fp16 [Cout_pad][Kpad / 32][hi x32 | lo x32], K order (tap, 32-channel block) for
 *        all taps, then -- Cin % 32 == 16 -- the 16-channel tails two taps per 32-wide step.
 *   64   row-window kernel for stride-1 layers with KW in 2..8 (the kw taps of a filter row share one fetch of the
 *        pixel operand); tn in {2, 3, 4}; weights fp16 [Cout_pad][Kpad / 32][hi x32 | lo x32] in the K order
 *        (kh, 32-channel block, kw), and per kh -- Cin % 32 == 16 -- the 16-channel tails two taps per 32-wide step
 *        last.  Same accuracy as 16 / 128, not the same bits (fp32 summation order).
 *   512  | 34: conv_pipe.hip configuration 34, register-resident-weights sliding-window kernel for Cin = 32, 3x3,
 *        stride 1, Cout = 32 or 64; weights as for 16.
 *   256  (round 3) POOLED INPUT: the convolution (1x1, stride 1, no padding, Cin % 32 == 0, weights as for 128) reads
 *        max_pool2d(x, 3, stride 2): args->H, W describe the UN-pooled tensor x, args->OH, OW the pooled grid
 *        ((H - 3) / 2 + 1), which is also the output grid; M = N * OH * OW; the low bits of tn are ignored (tiles of 64 pixels x 128 couts when
 *        Cout <= 128, else x 256 couts; weights, scale and bias zero-padded to that many rows).  The pool is taken while
 *        the pixel operand is loaded (torchvision's MaxPool2d(3, 2) before Conv2d_3b_1x1 and before Mixed_5b,
 *        image_realism/FID/inception.py:61-71); bit-identical to tise_maxpool3s2_split_nhwc followed by variant 128.
 *   Round 4 modifiers (OR-ed into `tn`):
 *   1024 with 512 | 34 (64 couts, unpadded: Conv2d_2b on its zero-bordered input): max_pool2d(3, stride 2) of the RESULT is
 *        taken in the epilogue (image_realism/FID/inception.py:63-65) and only the pooled split tensor
 *        N x ((OH - 3) / 2 + 1) x ((OW - 3) / 2 + 1) is written; split segments only; 128 <= W, LDS-bounded (~W <= 152).
 *   2048 with 64 (tile width 3): the HORIZONTAL half of that pool in the row-window kernel's epilogue -- destinations
 *        N x OH x ((OW - 3) / 2 + 1) (Conv2d_4a, inception.py:69-70); with 256: the pooled-input kernel takes the three
 *        VERTICAL taps of such an input (H = the producer's rows, W = OW = its pooled columns).  Both halves together are
 *        bit-identical to tise_maxpool3s2_split_nhwc of the stored result.
 *   4096 with 128: K order (32-channel block, tap) instead of (tap, block); unpadded layers with Cin % 32 == 0 and more
 *        than one tap, weights packed in that order (the stride-2 3x3 layers of Mixed_6a / 7a: their tap-shifted re-reads
 *        hit L2).  Same products, another summation order than variant 16 / 128.
 *   (Round 1's variants 0 / 32 and the other pipe configurations tied with 128 and were removed.)
 * Bits 8..13 of args->nseg are measurement switches (tools/conv_ablate.py, tools/conv_stamps.py; 0x2000: the epilogue
 * converts and stages but does not store) and must be zero in product calls.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int c0, c1;             /* output-channel range [c0, c1) of this segment (c0 % 8 == 0)             */
    void* dst;              /* mode 0: split tensor (layout above); mode 1: float*                     */
    long long ld;           /* mode 0: channel count C of the destination tensor (a pixel is 4*C bytes);
                               mode 1: destination floats per pixel                                    */
    int off;                /* first destination channel (mode 0: % 8 == 0, mode 1: % 4 == 0)          */
    int mode;
} tise_conv_seg;

typedef struct {
    const void* x;          /* split tensor [N][H][W][Cin], layout above                               */
    const void* w;          /* fp16 weights, K = (kh, kw, cin) in the order described above            */
    long long w_plane;      /* variants 16 / 512: elements between the hi and lo weight planes         */
    const float* scale;     /* [Cout_pad] un-scaling of the (power-of-two pre-scaled) weights          */
    const float* bias;      /* [Cout_pad]                                                              */
    int N, H, W, Cin, KH, KW, SH, SW, PH, PW, OH, OW;
    int Cout, K, Kpad;
    long long M;            /* N*OH*OW                                                                 */
    int nseg;
    tise_conv_seg seg[4];
    /* Variant 512|34 only (the one configuration tise_conv_pipe_launch accepts; 0 everywhere else): write the (OH, OW) result INTO a larger
     * destination image of out_hp x out_wp pixels per image at offset (out_y0, out_x0) -- Conv2d_2a writes into the
     * interior of a zero-bordered 149 x 149 buffer so that the padded Conv2d_2b runs as a valid convolution with no
     * per-lane tap masks.  out_hp = 0: the destination is the plain (OH, OW) image. */
    int out_hp, out_wp, out_y0, out_x0;
} tise_conv_args;

int tise_conv_split_f16(const tise_conv_args* args, int tn, void* stream);

/* Range guard of the split format: every kernel that writes split tensors (the convolution epilogues, the stem
 * convolution, the average-pool tail) raises a per-device flag when a value it converts exceeds the fp16 range
 * (65504; it would become +inf in the hi half) or is NaN.  Reads the flag into *flag_host (0 / 1), clears it and
 * synchronises `stream`.  The Python mirror calls it once per image set and raises FloatingPointError. */
int tise_split_overflow_check(int* flag_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp64 GEMM building block (MFMA v_mfma_f64_16x16x4_f64), exported for tests/bench only:
 * C[m][n] (ldc) = sum_k A(m,k) * B(k,n) with A(m,k) = a[m*sam + k*sak], B(k,n) = b[k*sbk + n*sbn].
 * ------------------------------------------------------------------------------------------ */
int tise_gemm_f64(const double* a_dev, int64_t sam, int64_t sak, const double* b_dev, int64_t sbk,
                  int64_t sbn, double* c_dev, int64_t ldc, int m, int n, int k, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TISE_HIP_H */
