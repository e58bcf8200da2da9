/*
 * cgs_hip.h -- C ABI of libcgs_hip.so: the MI355X (gfx950) kernels under the
 * collaborative-sampling refinement hot path.
 *
 * The reference (vita-epfl/collaborative-gan-sampling) has no FFI layer: its hot
 * path (sampling/collaborator.py:26-88) is made of stock TensorFlow ops reached
 * through nsgan/ops.py.  Each entry point below replaces one of those TF call
 * sites (cited per function; paths are under the reference root).  The host
 * side that mirrors the reference's Python operator / sampler classes sits on
 * top of this ABI (collaborative-gan-sampling_amd/ops.py, sampling/ *.py) and
 * binds it with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions:
 *   - plain C: raw DEVICE pointers + ints; no torch / hip types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream);
 *   - activations NHWC fp32; conv weights HWIO [kh,kw,Cin,Cout]
 *     (nsgan/ops.py:39); deconv weights [kh,kw,Cout,Cin] (nsgan/ops.py:51);
 *     linear weights [in,out] (nsgan/ops.py:76);
 *   - TF 'SAME' padding everywhere (extra pixel bottom/right);
 *   - caller allocates every output and workspace; nothing is allocated or
 *     freed inside; calls are asynchronous on `stream` and graph-capturable;
 *   - return 0 on success, a negative CGS_E* code otherwise; never throws;
 *     cgs_last_error() gives a thread-local message.
 */
#ifndef CGS_HIP_H
#define CGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGS_OK 0
#define CGS_EINVAL (-1)   /* bad argument / unsupported shape */
#define CGS_EWORKSPACE (-2) /* workspace too small */
#define CGS_ELAUNCH (-3)  /* HIP launch error */

/* epilogues fused into the conv / deconv / linear kernels (applied after +bias) */
#define CGS_EPI_NONE 0
#define CGS_EPI_LRELU 1        /* max(v, leak*v), leak = 0.2        nsgan/ops.py:69-70            */
#define CGS_EPI_AFFINE_RELU 2  /* relu(a[c]*v + b[c]) = inference-mode bn + relu, nsgan/GAN.py:96-98 */
#define CGS_EPI_TANH 3         /* tanh(v)                            nsgan/GAN.py:100              */
/* backward-data epilogues: the produced gradient is w.r.t. the OUTPUT of the layer below; these fold that
 * layer's activation gradient in (aux = the layer-below's saved output, same shape as the result; no bias) */
#define CGS_EPI_RELU_BWD_AFFINE 4  /* aux > 0 ? v*a[c] : 0     through relu(a*x+b)   nsgan/GAN.py:96-98 */
#define CGS_EPI_LRELU_BWD 5        /* v * (aux > 0 ? 1 : 0.2)  through lrelu         nsgan/ops.py:69-70 */
#define CGS_EPI_TANH_BWD 6         /* v * (1 - aux*aux)        through tanh          nsgan/GAN.py:100   */

/* which transposition of the weights a conv-family call contracts over */
#define CGS_CONV_FWD 0          /* tf.nn.conv2d                        nsgan/ops.py:41 */
#define CGS_CONV_BWD_DATA 1     /* its input gradient (tf.gradients)   sampling/collaborator.py:31 */
#define CGS_DECONV_FWD 2        /* tf.nn.conv2d_transpose              nsgan/ops.py:55 */
#define CGS_DECONV_BWD_DATA 3   /* its input gradient                  sampling/collaborator.py:31 */

int cgs_version(void);
/* sha256 (64 hex digits) of the kernel sources this library was built from (every .hip and .h file of csrc/, then include/cgs_hip.h), embedded at
 * build time (csrc/Makefile, csrc/stamp.hip).  The host binding refuses a library whose stamp differs from the sources it ships with
 * (cgs_amd/lib.py::load); bench.py reports this embedded value, so a measured number names the code that produced it. */
const char* cgs_source_sha(void);
const char* cgs_last_error(void);
/* Name of the compute kernel the calling thread's most recent conv-family call launched (for profiling). */
const char* cgs_last_kernel(void);
/* Floating-point operations that call really issued to the matrix cores: 2 x (algorithmic multiply-accumulates minus those
 * of zero-padding taps the kernel skipped).  0 = the kernel skips nothing (executed = algorithmic).  For rooflines. */
double cgs_last_executed_flops(void);
/* Tail split of the calling thread's most recent conv-family call (0 / 0 = none): how many of the launch's last output tiles were
 * contracted by several workgroups over disjoint ranges of the reduction, and by how many each -- a scheduling detail of the
 * implicit GEMM (a launch whose tile count leaves a partial last round of workgroups; csrc/igemm.hip), reported for profiling and
 * tests.  The partial tiles are added in a fixed order: results stay bit-identical from run to run. */
int cgs_last_tail_tiles(void);
int cgs_last_tail_split(void);

/* Contraction arithmetic of the implicit-GEMM layers, per calling THREAD (default CGS_CONTRACTION_F32; no process-wide state).
 * F32: v_mfma_f32_32x32x2_f32 -- exact fp32 products, an fp32 fma chain over K: what tf.nn.conv2d / conv2d_transpose / matmul
 *      (nsgan/ops.py:41,55,81) compute at the reference's precision.  The headline numbers are measured in this mode.
 * BX6: opt-in.  Calls whose output channels are a multiple of 64, that reduce over whole 32-channel chunks (Cred % 32 == 0, at most
 *      16 taps per axis) and whose grid fills the GPU (>= 256 blocks, K >= 512) -- cgs_conv_family says which -- run on the bf16 matrix
 *      cores instead: every fp32 operand is split exactly into three bf16 pieces and six of the nine piece
 *      products are accumulated in fp32 (the dropped ones are below 3 * 2^-24 of the product): the result differs from the F32
 *      mode's by rounding errors of the size of an fp32 chain's own (igemm_bx6.hip; error analysis: DESIGN.md), in 6/16 of
 *      the matrix time.  Everything else (families, epilogues, fused statistics, layouts) is unchanged; the packed-weight
 *      layout differs, which cgs_conv_family reports as CGS_FAMILY_IGEMM_BX6 (key cached workspaces by it).
 * BX6_ALL: BX6 for every call whose geometry allows it, whatever its size (test coverage of the kernel on small shapes).
 * Set it before the thread's calls -- including cgs_conv_family / cgs_conv_signs_ok / cgs_conv_stat_* queries, which answer
 * for the mode in force.  Returns CGS_OK or CGS_EINVAL. */
#define CGS_CONTRACTION_F32 0
#define CGS_CONTRACTION_BX6 1
#define CGS_CONTRACTION_BX6_ALL 2
int cgs_set_contraction(int mode);
int cgs_get_contraction(void);

/* Bytes of workspace a conv-family call needs for its packed copy of the weights
 * (op = one of CGS_CONV_* above; Cin/Cout are those of the LAYER, i.e. of the
 * forward op, for the backward variants too). */
size_t cgs_conv_ws_bytes(int op, int kh, int kw, int sh, int sw, int Cin, int Cout);
/* Same, plus room for the split-K partial slabs the library uses when the launch would otherwise
 * leave most CUs idle (small batch x small spatial size, e.g. the MNIST fc layers at batch 64).
 * B, H, W as passed to the entry point.  Optional: with only cgs_conv_ws_bytes() bytes the call runs un-split.
 * Layout: [packed weights | slabs]; ws_prepacked refers to the first part only. */
size_t cgs_conv_ws_bytes_for(int op, int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw);

/* Kernel family a conv-family call with these arguments will run, given 16-byte aligned pointers and a workspace of
 * ws_bytes (B, H, W, Cin, Cout as passed to the entry point; Ho, Wo = the deconv ops' output size, ignored for the conv
 * ops).  Each family keeps its OWN packed-weight layout in the workspace, so a caller
 * that sets ws_prepacked must key its cached workspaces by (weights, op, geometry, family): the epilogue takes part in
 * the choice.  CGS_FAMILY_SMALLN_T packs nothing.  Negative = error. */
#define CGS_FAMILY_IGEMM 0      /* implicit GEMM on the fp32 matrix cores (igemm.hip)                     */
#define CGS_FAMILY_QUAD 1       /* <= 4 output channels, transposed stride 2 (convt_quad.hip)             */
#define CGS_FAMILY_SMALLN_T 2   /* same, VALU form (convt_smalln.hip)                                     */
#define CGS_FAMILY_SMALLN_F 3   /* stride-1 forward conv with <= 4 output channels (conv_smalln_f.hip)    */
#define CGS_FAMILY_PATCH 4      /* 3-channel strided / stem convs from an LDS patch (conv_patch.hip)      */
#define CGS_FAMILY_TAPS 5       /* 4x4 stride-2 conv from ONE channel, K = 16 taps (convt_taps.hip); reads the weights unpacked */
#define CGS_FAMILY_DOT 6        /* forward conv to <= 4 channels over a deep reduction (conv_dot.hip); weights unpacked         */
#define CGS_FAMILY_IGEMM_BX6 7  /* implicit GEMM through split-bf16 MFMA (igemm_bx6.hip): only after cgs_set_contraction       */
int cgs_conv_family(int op, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                    int epilogue, size_t ws_bytes);

/* conv2d: y[B,Ho,Wo,Cout] = conv(x[B,H,W,Cin], w[kh,kw,Cin,Cout], stride, 'SAME') + bias, then epilogue.
 * Replaces tf.nn.conv2d + tf.nn.bias_add at nsgan/ops.py:41-44 (Ho = ceil(H/sh)).
 * ws/ws_bytes: see cgs_conv_ws_bytes; ws_prepacked != 0 means ws already holds the packed
 * weights from an earlier call with the same w (frozen weights: pack once). bias may be NULL. */
int cgs_conv2d_nhwc_fwd(const float* x, const float* w, const float* bias, float* y,
                        int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw,
                        int epilogue, const float* ep_a, const float* ep_b,
                        void* ws, size_t ws_bytes, int ws_prepacked, void* stream);

/* The same conv (no fused epilogue) that also leaves per-block column sums of its OUTPUT -- sum and sum of squares per
 * channel -- in stat_part[G][2][Cout], so the batch-statistics batch norm that follows it in D (nsgan/GAN.py:65 -> nsgan/ops.py:19-26)
 * does not re-read the tensor for its statistics pass.  G = cgs_conv_stat_partials(...) (same B, H, W, ..., ws_bytes as the call);
 * 0 = the fused form is not available for this call (another kernel family, Cout % 4 != 0, a batch the library would split):
 * use cgs_conv2d_nhwc_fwd + cgs_bn_train_lrelu_fwd.  Deterministic (fixed reduction trees). */
int cgs_conv_stat_partials(int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, size_t ws_bytes);
int cgs_conv2d_nhwc_fwd_stats(const float* x, const float* w, const float* bias, float* y,
                              int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw,
                              void* ws, size_t ws_bytes, int ws_prepacked,
                              float* stat_part, size_t stat_part_bytes, void* stream);

/* The same statistics per GROUP of group_images consecutive images, and for the transposed convolution too: what an instance norm
 * behind a conv / deconv needs (a group = one sample; CycleGAN generator and PatchGAN discriminator of BASELINE config 5 -- the
 * reference ships no code for them), and D's batch norm (nsgan/GAN.py:65,67 -> nsgan/ops.py:19-26) when several logical batches
 * share one launch (a group = one logical batch: every batch keeps the statistics the reference computes for it alone).
 * cgs_conv_stat_layout says where a group's partial rows lie in the [rows][2][Cout] buffer that cgs_conv2d_nhwc_fwd_stats
 * (op = CGS_CONV_FWD; Ho, Wo ignored) / cgs_deconv2d_nhwc_fwd_stats (op = CGS_DECONV_FWD) fill (and, with the *_BWD_DATA ops, the
 * [rows][2][Cin] buffer of the *_bwd_data_nstats entry points below): group g owns, for every segment
 * s < *nseg, the *rows_per_seg rows starting at row s * *seg_stride + g * *rows_per_seg.  Returns the buffer's row count; 0 = not
 * available (another kernel family, Cout % 4 != 0, a split batch, parity classes of unequal size, a group that does not end on
 * a 64-row boundary of the launch's row order).  cgs_groupnorm_lrelu_fwd_from_partials (below) consumes it. */
int cgs_conv_stat_layout(int op, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                         int group_images, size_t ws_bytes, int* rows_per_seg, int* nseg, int* seg_stride);
int cgs_deconv2d_nhwc_fwd_stats(const float* x, const float* w, const float* bias, float* y,
                                int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                                void* ws, size_t ws_bytes, int ws_prepacked,
                                float* stat_part, size_t stat_part_bytes, void* stream);

/* The BACKWARD half of the same fusion (round 6).  tf.gradients through D's batch norm (sampling/collaborator.py:31 over
 * nsgan/ops.py:19-26, nsgan/GAN.py:65,67) needs two column sums over the gradient dy that arrives at the norm's output:
 * sum d and sum d * xhat, d = dy * lrelu'(gamma * xhat + beta), xhat = (x - mean) * invstd.  dy is the RESULT of the backward-data
 * pass of the convolution above the norm, so that launch can leave them: cgs_conv2d_nhwc_bwd_data_nstats /
 * cgs_deconv2d_nhwc_bwd_data_nstats are cgs_conv2d_nhwc_bwd_data / cgs_deconv2d_nhwc_bwd_data (no epilogue) that also read
 * x_norm -- the norm's input, same shape as dx -- and write the sums as partial rows [rows][2][Cin] (one per 64 GEMM rows):
 * cgs_conv_stat_layout with op = CGS_CONV_BWD_DATA / CGS_DECONV_BWD_DATA (same argument meaning as the entry point's) says where
 * the rows of every group of group_images consecutive images lie (group_images = B: batch norm over the whole batch; 1: instance
 * norm; b: fused logical batches), 0 = not available (then: the plain backward-data + cgs_bn_train_lrelu_bwd_data /
 * cgs_instnorm_lrelu_bwd_data).  mean / invstd are [B / group_images][Cin], gamma / beta [Cin].
 * cgs_norm_lrelu_bwd_from_partials finishes the norm's backward-data from them: finalize + ONE pass over dy and x (the sums pass of
 * cgs_bn_train_lrelu_bwd_data -- 2 of its 5 tensor passes -- is gone); dy / x are [groups][M_group][C], dx may be dy.
 * ws: cgs_bn_ws_bytes(M, C) / cgs_instnorm_ws_bytes(groups, M_group, C) cover it.  Deterministic (fixed reduction trees). */
int cgs_conv2d_nhwc_bwd_data_nstats(const float* dy, const float* w, float* dx,
                                    int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw,
                                    const float* x_norm, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    float leak, int group_images, void* ws, size_t ws_bytes, int ws_prepacked,
                                    float* stat_part, size_t stat_part_bytes, void* stream);
int cgs_deconv2d_nhwc_bwd_data_nstats(const float* dy, const float* w, float* dx,
                                      int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                                      const float* x_norm, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      float leak, int group_images, void* ws, size_t ws_bytes, int ws_prepacked,
                                      float* stat_part, size_t stat_part_bytes, void* stream);
int cgs_norm_lrelu_bwd_from_partials(const float* dy, const float* x, const float* part, int groups, int rows_per_seg, int nseg,
                                     int seg_stride, const float* gamma, const float* beta, const float* mean, const float* invstd,
                                     float leak, float* dx, int M_group, int C, void* ws, size_t ws_bytes, void* stream);

/* conv2d backward-data: dx[B,H,W,Cin] = d/dx of the conv above applied to dy[B,Ho,Wo,Cout].
 * Replaces the Conv2DBackpropInput node tf.gradients emits (sampling/collaborator.py:31).
 * epilogue: CGS_EPI_NONE or one of the *_BWD modes (ep_aux [B,H,W,Cin], ep_a [Cin]). */
int cgs_conv2d_nhwc_bwd_data(const float* dy, const float* w, float* dx,
                             int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw,
                             int epilogue, const float* ep_a, const float* ep_aux,
                             void* ws, size_t ws_bytes, int ws_prepacked, void* stream);

/* deconv2d: y[B,Ho,Wo,Cout] = conv2d_transpose(x[B,H,W,Cin], w[kh,kw,Cout,Cin], output_shape, stride) + bias,
 * then epilogue.  Replaces tf.nn.conv2d_transpose + bias_add at nsgan/ops.py:55,61-62
 * (requires H == ceil(Ho/sh), W == ceil(Wo/sw)). */
int cgs_deconv2d_nhwc_fwd(const float* x, const float* w, const float* bias, float* y,
                          int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                          int epilogue, const float* ep_a, const float* ep_b,
                          void* ws, size_t ws_bytes, int ws_prepacked, void* stream);

/* deconv2d backward-data: dx[B,H,W,Cin] from dy[B,Ho,Wo,Cout] (a strided 'SAME' conv with the deconv weights). */
int cgs_deconv2d_nhwc_bwd_data(const float* dy, const float* w, float* dx,
                               int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                               int epilogue, const float* ep_a, const float* ep_aux,
                               void* ws, size_t ws_bytes, int ws_prepacked, void* stream);

/* ---- sign masks -------------------------------------------------------------------------------------------------------
 * d relu / d lrelu need ONE BIT of the saved activation (is it > 0), not the fp32 tensor: where the consumer of that
 * gradient is an HBM-bound kernel -- the backward-data of the generator's last, 3-channel deconv (a strided conv of the
 * image gradient whose epilogue applies relu'(y) * a of nsgan/GAN.py:98-99's bn + relu; tf.gradients at
 * sampling/collaborator.py:31) -- the producing forward launch leaves a bitmask next to its output and the backward launch
 * reads that instead (8 MB instead of 268 MB for the 32x32x64 map at batch 1024).
 * Layout, for a tensor [P pixels][C channels], C % 32 == 0: one plane of P words per 32-channel group,
 * uint32 word[(c / 32) * P + p], bit 8 * (c % 4) + (c % 32) / 4 is set iff x[p][c] > 0 (the bit order of the producing
 * epilogue's lane layout; the consumer un-shuffles it).  Caller-allocated, P * C / 8 bytes, 4-byte aligned.
 * cgs_conv_signs_ok: 1 if the call with these arguments can LEAVE a mask (op a forward with CGS_EPI_LRELU / CGS_EPI_AFFINE_RELU)
 * or TAKE one (op a backward-data with CGS_EPI_LRELU_BWD / CGS_EPI_RELU_BWD_AFFINE); 0 = use the fp32 aux tensor. */
int cgs_conv_signs_ok(int op, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                      int epilogue, size_t ws_bytes);
/* cgs_deconv2d_nhwc_fwd that also writes the sign mask of y (after the epilogue) to signs. */
int cgs_deconv2d_nhwc_fwd_signs(const float* x, const float* w, const float* bias, float* y,
                                int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                                int epilogue, const float* ep_a, const float* ep_b, unsigned* signs,
                                void* ws, size_t ws_bytes, int ws_prepacked, void* stream);
/* cgs_deconv2d_nhwc_bwd_data whose relu' / lrelu' epilogue reads the sign mask of the saved activation (shape of dx). */
int cgs_deconv2d_nhwc_bwd_data_signs(const float* dy, const float* w, float* dx,
                                     int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw,
                                     int epilogue, const float* ep_a, const unsigned* aux_signs,
                                     void* ws, size_t ws_bytes, int ws_prepacked, void* stream);

/* linear: y[B,out] = x[B,in] @ w[in,out] + bias, then epilogue (NONE or LRELU).
 * Replaces tf.matmul + bias at nsgan/ops.py:81-83.  ws as for conv (op CGS_CONV_FWD, kh=kw=1). */
int cgs_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in, int out,
                   int epilogue, void* ws, size_t ws_bytes, int ws_prepacked, void* stream);
/* The single-logit head of D (nsgan/GAN.py:68: linear(net, 1)) and the loss seed of the refinement loop in one launch:
 * logits[b] = x[b,:] . w + bias; dlogits[b] = sigmoid(logits[b]) - 1 (nsgan/GAN.py:176-177 through tf.gradients,
 * sampling/collaborator.py:31); logit_mean[b] = logits[b] (collaborator.py:34-37 with one logit per sample).
 * = cgs_linear_fwd(out = 1) followed by cgs_bce_ones_grad_rowmean(P = 1), bit for bit. */
int cgs_linear_out1_bce(const float* x, const float* w, const float* bias, float* logits, float* dlogits, float* logit_mean, int B, int in,
                        void* stream);
/* linear backward-data: dx[B,in] = dy[B,out] @ w^T   (ws: op CGS_CONV_BWD_DATA, kh=kw=1). */
int cgs_linear_bwd_data(const float* dy, const float* w, float* dx, int B, int in, int out,
                        void* ws, size_t ws_bytes, int ws_prepacked, void* stream);

/* Batch-statistics batch norm (tf.contrib.layers.batch_norm(is_training=True), nsgan/ops.py:19-26)
 * fused with the lrelu that follows it in D (nsgan/GAN.py:65,67).  x is [M,C] (M = B*H*W or B).
 *   fwd : mean[c], invstd[c] = 1/sqrt(biased_var+eps) are written (saved for backward);
 *         y = lrelu(gamma*(x-mean)*invstd + beta, leak)   (leak = 1 gives plain bn).
 *   ws  : cgs_bn_ws_bytes(M, C) bytes of scratch for the two-stage deterministic reduction. */
size_t cgs_bn_ws_bytes(int M, int C);
int cgs_bn_train_lrelu_fwd(const float* x, const float* gamma, const float* beta, float eps, float leak,
                           float* y, float* mean, float* invstd, int M, int C,
                           void* ws, size_t ws_bytes, void* stream);
/*   fwd from the partial sums a cgs_conv2d_nhwc_fwd_stats call left (part[G][2][C]): statistics pass skipped. */
int cgs_bn_train_lrelu_fwd_from_partials(const float* x, const float* part, int G, const float* gamma, const float* beta,
                                         float eps, float leak, float* y, float* mean, float* invstd, int M, int C,
                                         void* ws, size_t ws_bytes, void* stream);
/*   bwd : dx = gamma*invstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)), dy' = dy*lrelu'(bn(x)),
 *         the input gradient tf.gradients builds for the two ops (sampling/collaborator.py:31). */
int cgs_bn_train_lrelu_bwd_data(const float* dy, const float* x, const float* gamma, const float* beta,
                                const float* mean, const float* invstd, float leak, float* dx, int M, int C,
                                void* ws, size_t ws_bytes, void* stream);

/* Synchronised batch statistics: ONE logical batch whose rows are split over several GPUs (the reference computes D's
 * batch norm over the whole batch, nsgan/GAN.py:175 -> nsgan/ops.py:19-26).  Each pass is cut at the per-channel sums:
 *   cgs_bn_sync_fwd_sums : sums[0..C) = sum_m x, sums[C..2C) = sum_m x^2 over the LOCAL rows, in double
 *   -- caller: all-reduce(SUM) of the 2*C doubles over the ranks (RCCL) --
 *   cgs_bn_sync_fwd_apply: statistics from the global sums and M_total = total rows; y, mean, invstd as cgs_bn_train_lrelu_fwd
 *   cgs_bn_sync_bwd_sums : sums = {sum dy', sum dy'*xhat} over the local rows (mean / invstd = the saved global statistics)
 *   -- all-reduce --
 *   cgs_bn_sync_bwd_apply: dx as cgs_bn_train_lrelu_bwd_data with the global means.
 * With one rank (M_total = M) the results equal the unsplit entry points'.  ws: cgs_bn_ws_bytes(M, C). */
int cgs_bn_sync_fwd_sums(const float* x, double* sums, int M, int C, void* ws, size_t ws_bytes, void* stream);
int cgs_bn_sync_fwd_apply(const float* x, const float* gamma, const float* beta, float eps, float leak, const double* sums,
                          long long M_total, float* y, float* mean, float* invstd, int M, int C,
                          void* ws, size_t ws_bytes, void* stream);
int cgs_bn_sync_bwd_sums(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                         const float* invstd, float leak, double* sums, int M, int C, void* ws, size_t ws_bytes, void* stream);
int cgs_bn_sync_bwd_apply(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* invstd, float leak, const double* sums, long long M_total, float* dx, int M, int C,
                          void* ws, size_t ws_bytes, void* stream);

/* Instance norm (+ lrelu; leak = 0 gives relu, 1 plain): statistics over the HW pixels of every (sample, channel);
 * x is [B,HW,C].  CycleGAN-style generators / PatchGAN discriminators (BASELINE config 5; the reference ships no
 * code for them).  mean / invstd are [B,C].  ws: cgs_instnorm_ws_bytes(B, HW, C). */
size_t cgs_instnorm_ws_bytes(int B, int HW, int C);
int cgs_instnorm_lrelu_fwd(const float* x, const float* scale, const float* offset, float eps, float leak, float* y,
                           float* mean, float* invstd, int B, int HW, int C, void* ws, size_t ws_bytes, void* stream);
int cgs_instnorm_lrelu_bwd_data(const float* dy, const float* x, const float* scale, const float* offset,
                                const float* mean, const float* invstd, float leak, float* dx, int B, int HW, int C,
                                void* ws, size_t ws_bytes, void* stream);
/*   fwd of a norm over groups of rows from the partial sums the producing convolution left (cgs_conv_stat_layout): x is
 *   [groups][M_group][C], mean / invstd [groups][C]; ws >= groups * 4 * C floats.  Instance norm: groups = B, M_group = HW. */
int cgs_groupnorm_lrelu_fwd_from_partials(const float* x, const float* part, int groups, int rows_per_seg, int nseg, int seg_stride,
                                          const float* gamma, const float* beta, float eps, float leak, float* y, float* mean,
                                          float* invstd, int M_group, int C, void* ws, size_t ws_bytes, void* stream);
/* out = a + b (residual connections). */
int cgs_add(const float* a, const float* b, float* out, size_t n, void* stream);

/* Inference-mode bn folded to a per-channel affine (nsgan/GAN.py:87,94): a = gamma/sqrt(mv+eps), b = beta - a*mm. */
int cgs_bn_fold(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                float eps, float* a, float* b, int C, void* stream);
/* y = relu(a[c]*x + b[c]) and its input gradient dx = dy*(y>0)*a[c]  (G-tail bn+relu, nsgan/GAN.py:96-98). */
int cgs_affine_relu_fwd(const float* x, const float* a, const float* b, float* y, int M, int C, void* stream);
int cgs_affine_relu_bwd(const float* dy, const float* y, const float* a, float* dx, int M, int C, void* stream);
/* y = a[c]*x + b[c] (inference-mode bn with no activation after it) and dx = dy*a[c]. */
int cgs_affine_fwd(const float* x, const float* a, const float* b, float* y, int M, int C, void* stream);
int cgs_affine_bwd(const float* dy, const float* a, float* dx, int M, int C, void* stream);
/* y = max(x, leak*x) and dx = dy*(y>0 ? 1 : leak)   (nsgan/ops.py:69-70). */
int cgs_lrelu_fwd(const float* x, float leak, float* y, size_t n, void* stream);
int cgs_lrelu_bwd(const float* dy, const float* y, float leak, float* dx, size_t n, void* stream);
/* y = tanh(x), dx = dy*(1-y*y)   (nsgan/GAN.py:100). */
int cgs_tanh_fwd(const float* x, float* y, size_t n, void* stream);
int cgs_tanh_bwd(const float* dy, const float* y, float* dx, size_t n, void* stream);

/* Loss seed: dlogit = sigmoid(logit) - 1 = d softplus(-logit)/d logit, the gradient of
 * tf.nn.sigmoid_cross_entropy_with_logits(labels=1) summed over the batch (nsgan/GAN.py:176-177,
 * sampling/collaborator.py:31), and the per-sample mean logit over P patch entries
 * (sampling/collaborator.py:34-37).  logits is [B,P]. */
int cgs_bce_ones_grad_rowmean(const float* logits, float* dlogits, float* logit_mean, int B, int P, void* stream);

/* loss[i] = softplus(-logits[i]) = tf.nn.sigmoid_cross_entropy_with_logits(labels=1), unreduced (nsgan/GAN.py:176-177),
 * and its input gradient dlogits[i] = dloss[i] * (sigmoid(logits[i]) - 1)  (what tf.gradients emits, collaborator.py:31). */
int cgs_bce_ones_fwd(const float* logits, float* loss, size_t n, void* stream);
int cgs_bce_ones_bwd(const float* dloss, const float* logits, float* dlogits, size_t n, void* stream);
/* sigmoids[b] = mean over the P logits of sample b of sigmoid(logit): self.fake_sigmoids = tf.nn.sigmoid(self.fake_logits)
 * (nsgan/GAN.py:154-155; P = 1 there), the discriminator score the accept / reject step reads (nsgan/GAN.py:409,422). */
int cgs_sigmoid_rowmean(const float* logits, float* sigmoids, int B, int P, void* stream);
/* y = min(max(x, vmin), vmax): tf.clip_by_value on the refined map (sampling/collaborator.py:69-70). */
int cgs_clip(const float* x, float vmin, float vmax, float* y, size_t n, void* stream);

/* One refinement step's state update, fused (sampling/policy.py:31-37 momentum / :27-29 sgd,
 * sampling/collaborator.py:66-70 clip):
 *   m = first ? rate*g : alpha*m + rate*g ; theta -= m ; theta = clip(theta, vmin, vmax) if use_clip.
 * alpha = 0 and first = 1 give sgd.  n = B*F elements. */
int cgs_refine_update(float* theta, float* m, const float* g, float rate, float alpha, int first,
                      int use_clip, float vmin, float vmax, size_t n, void* stream);
/* Best-sample selection (sampling/collaborator.py:76-83): for each sample b
 *   upd = forced ? forced[b] == step_index : logit[b] > best_logit[b]      (strict >)
 *   best_logit[b], best_theta[b,:], best_step[b] = upd ? (logit[b], theta[b,:], step_index+1) : unchanged
 * forced = the probabilistic-mode index vector (int32, device) or NULL for deterministic mode. */
int cgs_refine_select(const float* theta, const float* logit, const int32_t* forced, int step_index,
                      float* best_theta, float* best_logit, float* best_step, int B, int F, void* stream);
/* cgs_refine_select_rows(rows -> best_rows) followed by cgs_refine_select(theta -> best_theta, scalars) with the two row copies in one
 * launch: the per-step bookkeeping of the engine (the rendered image follows the selection of the refined map).
 * tickets: NULL, or B ints of device memory that are ZERO at the call and are left zero by it (the caller clears them once and hands the
 * same buffer to every call on one stream): the scalars are then updated in the same launch, by the last block of each selected sample. */
int cgs_refine_select2(const float* rows, float* best_rows, int Frows, const float* theta, float* best_theta, int F, const float* logit,
                       const int32_t* forced, int step_index, float* best_logit, float* best_step, int* tickets, int B, void* stream);
/* The row copy of cgs_refine_select alone (same predicate, best_logit is only read): lets a second per-sample
 * tensor -- e.g. the rendered image of the step -- follow the same selection.  Call it BEFORE cgs_refine_select. */
int cgs_refine_select_rows(const float* src, const float* logit, const int32_t* forced, int step_index, float* dst,
                           const float* best_logit, int B, int F, void* stream);

/* ---- the 2-D path (BASELINE config 1: synthetic/ MLP GAN) ----------------------------------------------------------
 * ReLU MLP discriminator on 2-D points, 2 -> nhidden x (nlayers-1) -> 1 (synthetic/GAN.py:28-37); w[l] is layer l's
 * [din,dout] kernel (tf.layers.dense), b[l] its bias; w / b are HOST arrays of nlayers DEVICE pointers; nhidden <= 64.
 *   sigmoid[B]    = sigmoid(D(x))                                              synthetic/GAN.py:108
 *   saliency[B,2] = inv_batch * d sum_b softplus(-logit_b) / dx  (inv_batch = 1/B keeps the reduce_mean factor of :109-111)
 * saliency may be NULL. */
int cgs_mlp2d_sigmoid_saliency(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x,
                               float* sigmoid, float* saliency, int B, float inv_batch, void* stream);
/* The whole host loop of sampling/refiner_cpu.py:26-66 in one launch (one wave per sample): K steps of
 * sgd (method 0) / momentum (1) / ladam (2) (sampling/policy.py:26-61) on x[B,2] with loss = real_sigmoid_mean - sigmoid,
 * best-loss tracking (best_x[B,2], best_step[B]) and, if traj != NULL, the trajectory traj[B,K+1,2]. */
int cgs_refine2d(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x,
                 float real_sigmoid_mean, float inv_batch, int steps, float rate, int method,
                 float* best_x, float* best_step, float* traj, int B, void* stream);

/* The same with the baseline read from DEVICE memory (one float): the real batch's mean sigmoid never visits the host, so
 * consecutive batches queue without a synchronisation. */
int cgs_refine2d_devbase(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x,
                         const float* real_sigmoid_mean_dev, float inv_batch, int steps, float rate, int method,
                         float* best_x, float* best_step, float* traj, int B, void* stream);

/* The 2-D net's D shaping step (synthetic/main.py:366-370): gradients of
 *   d_loss = mean_b BCE(D(real_b), 1) + mean_b BCE(D(fake_b), 0)                     synthetic/GAN.py:69-74
 * w.r.t. every D variable and, if lr != 0, tf.train.GradientDescentOptimizer(lr)'s step w -= lr*g IN PLACE (synthetic/GAN.py:98-99).
 * w / b (and the optional gw / gb gradient outputs, same shapes; NULL or NULL entries = not returned) are HOST arrays of nlayers
 * DEVICE pointers; loss (device, 2 floats, may be NULL) <- (d_loss_real, d_loss_fake) BEFORE the update.  Deterministic
 * (fixed summation order).  ws: cgs_mlp2d_train_ws_bytes(B_real + B_fake, nlayers). */
size_t cgs_mlp2d_train_ws_bytes(int B_total, int nlayers);
int cgs_mlp2d_d_step(float* const* w, float* const* b, int nlayers, int nhidden, const float* real, int B_real,
                     const float* fake, int B_fake, float lr, float* const* gw, float* const* gb, float* loss,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- discriminator shaping step (the caller after the refinement path: nsgan/GAN.py:270-272, 126-146) ----------
 * Weight gradients of D's layers, the BCE seed with 0/1 targets, and the Adam update.  NOT part of the frozen-weight
 * refinement loop; provided so the method's only training step (shape D on refined samples) runs on the same ABI. */
/* dw[kh,kw,Cin,Cout] (+)= d/dw of conv2d_nhwc_fwd(x, w) contracted with dy[B,Ho,Wo,Cout] (Conv2DBackpropFilter). */
size_t cgs_conv_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw);
int cgs_conv2d_nhwc_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                               int kh, int kw, int sh, int sw, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* dw[in,out] (+)= x[B,in]^T dy[B,out]   (ws: cgs_conv_wgrad_ws_bytes(B,1,1,in,out,1,1,1,1)). */
int cgs_linear_bwd_weight(const float* x, const float* dy, float* dw, int B, int in, int out, int accumulate,
                          void* ws, size_t ws_bytes, void* stream);
/* db[C] (+)= column sums of dy[M,C]   (ws: cgs_bn_ws_bytes(M, C)). */
int cgs_bias_grad(const float* dy, float* db, int M, int C, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* dgamma, dbeta (+)= from the statistics the immediately preceding cgs_bn_train_lrelu_bwd_data call left in ITS
 * workspace `bwd_ws` (same M, C). */
int cgs_bn_train_param_grads(const void* bwd_ws, int M, int C, float* dgamma, float* dbeta, int accumulate, void* stream);
/* dlogits[i] = scale*(sigmoid(logits[i]) - target); loss_sum[0] = scale * sum_i BCE(logits[i], target) (may be NULL).
 * tf.nn.sigmoid_cross_entropy_with_logits + reduce_mean (nsgan/GAN.py:126-131) with scale = 1/n. */
int cgs_bce_logits_grad(const float* logits, float target, float scale, float* dlogits, float* loss_sum, int n, void* stream);
/* Adam (tf.train.AdamOptimizer semantics, nsgan/GAN.py:141-146): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * w -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) supplied by the caller. */
int cgs_adam_step(float* w, const float* g, float* m, float* v, float lr_t, float beta1, float beta2, float eps,
                  size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CGS_HIP_H */
