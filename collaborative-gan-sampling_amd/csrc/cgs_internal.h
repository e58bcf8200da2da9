// Internal declarations shared by the kernels of libcgs_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cgs_hip.h"

int cgs_set_error(int code, const char* fmt, ...);
void cgs_note_kernel(const char* name);   // remembered per thread, read back by cgs_last_kernel()
void cgs_note_flops(double executed);     // ... by cgs_last_executed_flops()
void cgs_add_flops(double executed);      // (accumulating form: one entry point may launch several batch chunks)
void cgs_note_tail(int tiles, int split); // ... by cgs_last_tail_tiles() / cgs_last_tail_split()

#define CGS_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) return cgs_set_error(CGS_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int cgs_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int cgs_round_up(int a, int b) { return cgs_ceil_div(a, b) * b; }

// TF 'SAME' padding before the first pixel.
static inline int cgs_same_pad_before(int size, int k, int s) {
    int out = cgs_ceil_div(size, s);
    int tot = (out - 1) * s + k - size;
    if (tot < 0) tot = 0;
    return tot / 2;
}

// ---------------------------------------------------------------------------------------------
// One "conv layer relation": a BIG tensor [B,Hb,Wb,Cb] and a SMALL tensor [B,Hs,Ws,Cs] with
// Hs = ceil(Hb/s), tied by weights w[kh][kw][Cb][Cs] (conv HWIO: Cb=Cin,Cs=Cout; deconv
// [kh,kw,Cout,Cin]: Cb=Cout,Cs=Cin -- the same layout, because conv2d_transpose is defined as the
// adjoint of the conv with that filter).  Two contractions exist:
//   F (big -> small): conv fwd, deconv bwd-data.   reduce over (tap, Cb), N = Cs
//   T (small -> big): deconv fwd, conv bwd-data.   reduce over (tap, Cs), N = Cb; s*s parity classes
// ---------------------------------------------------------------------------------------------
struct CgsLayer {
    int kh, kw, sh, sw;
    int Hb, Wb, Cb, Hs, Ws, Cs;
};

#define CGS_BN_MAX_BLOCKS 512   // stage-1 partial blocks of the per-channel reductions (bn.hip workspace layout)
#define CGS_BK 32           // K-tile of the implicit GEMM
#define CGS_MAX_CLASSES 4   // stride <= 2 for the T direction

struct IgemmClass {
    int R, C;        // base-pixel grid of this class per image (rows of the GEMM: M = B*R*C)
    int py, px;      // output pixel = (r*So + py, c*So + px)
    int nty, ntx;    // taps of this class: t = ta*ntx + tb
    int dy0, dx0;    // input pixel = (r*S + dy0 + ta*dstep, c*S + dx0 + tb*dstep)
    int ky0, kx0;    // weight tap of (ta,tb) = (ky0 + ta*kstep, kx0 + tb*kstep)
    int K;           // nty*ntx*Cred
    int w_off;       // float offset of this class' packed weights
    int slab_off;    // float offset of this class' split-K slabs
};

struct IgemmParams {
    const float* in;
    const float* wp;     // packed weights [class][Kpad/4][Np][4]
    const float* bias;
    const float* ep_a;
    const float* ep_b;
    const float* ep_aux; // [same shape as out] for the *_BWD epilogues
    float* out;
    int B, Hin, Win, Cred;      // input tensor (reduction channels per tap)
    int Hout, Wout, N, Np;      // output tensor, N channels (Np = N rounded up to 64)
    int S, So, dstep, kstep;
    int epilogue;
    int pix_major;       // GEMM rows ordered (pixel, image) instead of (image, pixel): enables zero-tap skipping
    int splitk;          // > 1: blockIdx.z splits the K tiles; raw partial tiles go to slab, reduced by a second kernel
    float* slab;         // [class][split][M_c][Np]
    int tap_parity;      // VEC K order visits the taps even offsets first, then odd, per axis (stride-2 F direction)
    int xcd_map;         // block id -> (XCD, local index) decode: an m-tile's n-tiles run on one XCD (see igemm_kernel)
    int cls_flip;        // two-round launches of several parity classes: in the odd round every group of cls_flip consecutive classes runs in reverse order (0 = off; igemm_kernel)
    int uni;             // uniform-tile K loop allowed (host policy, see cgs_igemm_launch)
    int lpt;             // pixel-major only: m-tiles visit the pixels in perm[] order (most valid taps first)
    unsigned* sign_out;  // != null: sign bitmask of the stored output (layout: cgs_hip.h, "sign masks"); wide epilogue, N % 32 == 0, no split-K
    long sign_plane;     // words per 32-channel plane of the mask = pixels of the WHOLE tensor (a launch may be one batch chunk of it)
    float* stat_part;    // != null: per-(m-tile, wave row) column sums / sums of squares of the output, [class][2 * m-tiles][2][N] (fused norm statistics)
    int stat_cls_rows;   // partial rows per parity class = 2 * m-tiles of a class (the launcher sets it)
    // Norm-BACKWARD statistics (round 6; stat_part != null and ns_mean != null, CGS_EPI_NONE, ep_aux = x): the launch is the backward-data
    // pass that produces the gradient dy w.r.t. the OUTPUT of a norm (+ lrelu) whose input x has the shape of this launch's result; the
    // partial rows then hold [sum d | sum d * xhat] with d = dy * lrelu'(scale * x + shift), xhat = (x - mean) * invstd -- the two column
    // sums the norm's backward pass otherwise takes in a pass of its own over dy and x (bn.hip, bn_partial_kernel<1>).
    // ns_mean / ns_inv are [groups][N], ns_gamma / ns_beta [N]; ns_gimg = images per statistics group (0: the whole batch is one group).
    const float* ns_mean = nullptr;
    const float* ns_inv = nullptr;
    const float* ns_gamma = nullptr;
    const float* ns_beta = nullptr;
    float ns_leak = 1.f;
    int ns_gimg = 0;
    int vec;             // the 32-channel-chunk K order / 16-byte row gathers apply: Cred % 32 == 0 and at most 16 taps per axis
    int prio_t[3];       // progress thresholds (1/256 of the block's K tiles) at which a block steps its wave priority down; 0 = off
    // tail split (igemm.hip, "tail split"): the LAST tail_n tiles of the launch (the end of the last class in dispatch order) are each
    // contracted by tail_s blocks over disjoint K ranges into compact partial tiles slab[tail tile][split][BM][BN], which
    // tail_reduce_kernel sums in a fixed order and finishes (bias, epilogue, store).  tail_s <= 1: off.
    int tail_from, tail_n, tail_s;
    int nclasses;
    IgemmClass cls[CGS_MAX_CLASSES];
    unsigned char perm[CGS_MAX_CLASSES][256];   // per class: base pixels sorted by descending valid-tap count
};

// i-th tap visited along one axis of an n-tap kernel: natural order, or evens then odds
__host__ __device__ inline int cgs_tap_order(int i, int n, int parity_first) {
    if (!parity_first) return i;
    const int ne = (n + 1) >> 1;
    return i < ne ? 2 * i : 2 * (i - ne) + 1;
}

// geometry builders (igemm.hip)
void cgs_geom_F(const CgsLayer& L, IgemmParams& p);
void cgs_geom_T(const CgsLayer& L, IgemmParams& p);
size_t cgs_packed_floats(const IgemmParams& p);

// launchers
int cgs_pack_weights(const IgemmParams& p, const CgsLayer& L, bool dirT, const float* w, float* packed, hipStream_t s);
int cgs_igemm_launch(const IgemmParams& p, void* slab, size_t slab_bytes, hipStream_t s);
size_t cgs_igemm_splitk_bytes(const IgemmParams& p);   // slab bytes the launch would like for split-K (0 = no split-K)
size_t cgs_igemm_slab_bytes(const IgemmParams& p);     // ... for split-K or the tail split: what a workspace should offer behind the packed weights
void cgs_igemm_row_policy(IgemmParams& p, int BM);     // sets pix_major, lpt, perm (BM = rows of the launch's block tile)
void cgs_igemm_count_flops(const IgemmParams& p, int BM);
// split-bf16 implicit GEMM (igemm_bx6.hip): fp32 operands as three bf16 pieces each, six bf16 MFMA products, fp32 accumulate
int cgs_igemm_bx6_ok(const CgsLayer& L, bool dirT, int B, bool any_size);
size_t cgs_igemm_bx6_packed_bytes(const IgemmParams& p);
int cgs_pack_weights_bx6(const IgemmParams& p, const CgsLayer& L, bool dirT, const float* w, void* packed, hipStream_t s);
int cgs_igemm_bx6_launch(const IgemmParams& p, hipStream_t s, void* dbg = nullptr, size_t dbg_bytes = 0);   // (dbg: stamps of a diagnostic build)
int cgs_contraction_mode();                             // thread-local: CGS_CONTRACTION_* (api.hip)
int cgs_igemm_row_order(const IgemmParams& p);         // GEMM row order the launcher will pick: 0 (image, pixel); 1 (pixel, image); 2 (pixel, image) in whole 128-image tiles
size_t cgs_convt_quad_ws_floats_bound(int kh, int kw, int Cs);
int cgs_convt_taps_ok(const CgsLayer& L);          // 4x4 stride-2 transposed conv to one channel: all 16 taps as MFMA columns (convt_taps.hip)
int cgs_convt_taps_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out, int epilogue,
                          const float* ep_a, const float* ep_aux, hipStream_t s);
int cgs_conv_dot_ok(const CgsLayer& L, int epilogue);       // forward conv to <= 4 channels over a deep reduction: a wave per output pixel (conv_dot.hip)
int cgs_conv_dot_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out, int epilogue,
                        const float* ep_a, const float* ep_b, hipStream_t s);
int cgs_conv_taps_ok(const CgsLayer& L, int epilogue);      // its forward twin: 4x4 stride-2 conv FROM one channel (K = 16)
int cgs_conv_taps_signs_ok(const CgsLayer& L, int epilogue);
int cgs_conv_taps_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out, int epilogue,
                         const float* ep_a, const float* ep_b, const float* ep_aux, const unsigned* aux_signs, hipStream_t s);
int cgs_convt_quad_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out,
                          int epilogue, const float* ep_a, const float* ep_aux, float* ws, size_t ws_bytes, int prepacked,
                          hipStream_t s);
int cgs_convt_smalln_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias,
                            float* out, int epilogue, hipStream_t s);

// fused epilogue math shared by the conv-family kernels (device code only)
#ifdef __HIPCC__
__device__ __forceinline__ float epilogue_apply(float v, int mode, float a, float b, float aux) {
    switch (mode) {
        case CGS_EPI_LRELU: return fmaxf(v, 0.2f * v);
        case CGS_EPI_AFFINE_RELU: return fmaxf(fmaf(a, v, b), 0.f);
        case CGS_EPI_TANH: return tanhf(v);
        case CGS_EPI_RELU_BWD_AFFINE: return aux > 0.f ? v * a : 0.f;
        case CGS_EPI_LRELU_BWD: return aux > 0.f ? v : 0.2f * v;
        case CGS_EPI_TANH_BWD: return v * (1.f - aux * aux);
        default: return v;
    }
}
#endif


// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the DEVICE's code object: set it once per device (a process may drive
// several GPUs) and report a failure instead of letting the launch that follows fail with an unrelated message.
// CGS_SMEM_ATTR(bytes, "who", kernel<template, args>) -- the kernel last, so its commas need no parentheses.
#define CGS_SMEM_ATTR(bytes, who, ...)                                                                                  \
    do {                                                                                                                \
        static bool smem_done_[64] = {};                                                                                \
        int smem_dev_ = 0;                                                                                              \
        (void)hipGetDevice(&smem_dev_);                                                                                 \
        smem_dev_ &= 63;                                                                                                \
        if (!smem_done_[smem_dev_]) {                                                                                   \
            hipError_t smem_e_ = hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (smem_e_ != hipSuccess) return cgs_set_error(CGS_ELAUNCH, "%s: MaxDynamicSharedMemorySize: %s", who, hipGetErrorString(smem_e_)); \
            smem_done_[smem_dev_] = true;                                                                               \
        }                                                                                                               \
    } while (0)

bool cgs_convt_quad_fits(const CgsLayer& L);
int cgs_conv_smalln_f_ok(const CgsLayer& L, int B, int epilogue);
size_t cgs_conv_smalln_f_ws_floats(const CgsLayer& L);
int cgs_conv_smalln_f_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out,
                             int epilogue, float* ws, size_t ws_bytes, int prepacked, hipStream_t s);
int cgs_conv_patch_ok(const CgsLayer& L, int epilogue);
int cgs_conv_patch_T_ok(const CgsLayer& L);          // backward-data of a stride-1 conv with <= 4 output channels
size_t cgs_conv_patch_ws_floats(const CgsLayer& L, bool dirT);
int cgs_conv_patch_launch(const CgsLayer& L, bool dirT, int B, const float* in, const float* w, const float* bias, float* out,
                          int epilogue, const float* ep_a, const float* ep_b, const float* ep_aux, float* ws, size_t ws_bytes,
                          int prepacked, hipStream_t s, const unsigned* aux_signs = nullptr);
int cgs_conv_patch_signs_ok(const CgsLayer& L, bool dirT, int epilogue);     // can the *_BWD epilogue take a sign mask instead of the fp32 aux tensor?
int cgs_igemm_signs_ok(const IgemmParams& p);                                 // can this launch leave the sign mask of its output? (call after the geometry is set)
