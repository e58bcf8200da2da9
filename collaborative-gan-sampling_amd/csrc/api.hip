// extern "C" entry points of libcgs_hip.so (declared in include/cgs_hip.h): argument checking,
// geometry, weight packing and kernel dispatch.  No allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>

#include "cgs_internal.h"

static thread_local char g_err[512] = "";
static thread_local const char* g_last_kernel = "";

static thread_local double g_last_flops = 0.0;
static thread_local int g_last_tail[2] = {0, 0};
void cgs_note_tail(int tiles, int split) { g_last_tail[0] = tiles; g_last_tail[1] = split; }
static thread_local int g_contraction = CGS_CONTRACTION_F32;
int cgs_contraction_mode() { return g_contraction; }

void cgs_note_kernel(const char* name) { g_last_kernel = name; }
void cgs_note_flops(double f) { g_last_flops = f; }
void cgs_add_flops(double f) { g_last_flops += f; }      // per launch: a call that splits its batch (run_dir) sums its chunks

int cgs_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

int cgs_version(void) { return 106; }
const char* cgs_last_error(void) { return g_err; }
const char* cgs_last_kernel(void) { return g_last_kernel; }
double cgs_last_executed_flops(void) { return g_last_flops; }
int cgs_last_tail_tiles(void) { return g_last_tail[0]; }
int cgs_last_tail_split(void) { return g_last_tail[1]; }
int cgs_set_contraction(int mode) {
    if (mode < CGS_CONTRACTION_F32 || mode > CGS_CONTRACTION_BX6_ALL) return cgs_set_error(CGS_EINVAL, "set_contraction: mode %d", mode);
    g_contraction = mode;
    return CGS_OK;
}
int cgs_get_contraction(void) { return g_contraction; }

}  // extern "C"

static bool smalln_ok(const CgsLayer& L, int epilogue) {
    return L.Cb <= 4 && L.sh == 2 && L.sw == 2 && L.Hb == 2 * L.Hs && L.Wb == 2 * L.Ws && (L.Cs % 4) == 0 &&
           epilogue != CGS_EPI_AFFINE_RELU;
}

static int make_layer(CgsLayer& L, int kh, int kw, int sh, int sw, int Hb, int Wb, int Cb, int Hs, int Ws, int Cs,
                      const char* who) {
    if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || Hb <= 0 || Wb <= 0 || Cb <= 0 || Cs <= 0)
        return cgs_set_error(CGS_EINVAL, "%s: non-positive dimension", who);
    if (Hs != cgs_ceil_div(Hb, sh) || Ws != cgs_ceil_div(Wb, sw))
        return cgs_set_error(CGS_EINVAL, "%s: 'SAME' geometry mismatch: big %dx%d stride %dx%d needs small %dx%d, got %dx%d", who,
                             Hb, Wb, sh, sw, cgs_ceil_div(Hb, sh), cgs_ceil_div(Wb, sw), Hs, Ws);
    L.kh = kh; L.kw = kw; L.sh = sh; L.sw = sw; L.Hb = Hb; L.Wb = Wb; L.Cb = Cb; L.Hs = Hs; L.Ws = Ws; L.Cs = Cs;
    return CGS_OK;
}

// Which kernel family (and therefore which packed-weight layout in the workspace) serves a call.  ONE decision function:
// run_dir() dispatches on it and cgs_conv_family() reports it, so a caller that caches packed workspaces can key them by
// the family and never hand one family's packed image to another (the epilogue and the pointer alignment take
// part in the choice, not only the geometry).
static int choose_family(const CgsLayer& L, bool dirT, int B, int epilogue, bool have_ws, size_t ws_bytes, bool in_al,
                         bool ws_al, bool rest_al) {
    if (dirT && smalln_ok(L, epilogue)) {
        if ((L.Cs % 16) == 0 && have_ws && cgs_convt_quad_fits(L)) return CGS_FAMILY_QUAD;
        if (epilogue < CGS_EPI_RELU_BWD_AFFINE) return CGS_FAMILY_SMALLN_T;             // VALU form (any Cs % 4 == 0), packs nothing
    }
    if (!dirT && cgs_conv_smalln_f_ok(L, B, epilogue) && have_ws && ws_bytes >= cgs_conv_smalln_f_ws_floats(L) * sizeof(float) &&
        in_al && ws_al)
        return CGS_FAMILY_SMALLN_F;
    if (!dirT && cgs_conv_taps_ok(L, epilogue)) return CGS_FAMILY_TAPS;                     // K = 16 taps: no packed weights, no alignment needs
    if (!dirT && cgs_conv_dot_ok(L, epilogue) && in_al) return CGS_FAMILY_DOT;             // (after the stride-1 VALU head kernel, which serves the large grids)
    const bool patch_f = !dirT && cgs_conv_patch_ok(L, epilogue), patch_t = dirT && cgs_conv_patch_T_ok(L);
    if ((patch_f || patch_t) && have_ws && ws_bytes >= cgs_conv_patch_ws_floats(L, dirT) * sizeof(float) && ws_al && rest_al)
        return CGS_FAMILY_PATCH;
    // the calling thread opted into the split-bf16 contraction (cgs_set_contraction): the calls it serves leave the fp32 kernel
    // (... if the workspace can hold its three bf16 weight planes, 6 bytes per weight: an undersized one keeps the fp32 kernel, whose
    // packed image it was sized for, instead of failing with EWORKSPACE -- the patch family above falls back the same way)
    if (g_contraction != CGS_CONTRACTION_F32 && cgs_igemm_bx6_ok(L, dirT, B, g_contraction == CGS_CONTRACTION_BX6_ALL)) {
        IgemmParams q;
        q.B = B; q.stat_part = nullptr; q.sign_out = nullptr; q.sign_plane = 0; q.epilogue = epilogue;
        if (dirT) cgs_geom_T(L, q); else cgs_geom_F(L, q);
        if (have_ws && ws_bytes >= cgs_igemm_bx6_packed_bytes(q)) return CGS_FAMILY_IGEMM_BX6;
    }
    return CGS_FAMILY_IGEMM;
}

// norm-backward statistics of a backward-data call (IgemmParams::ns_*; x rides in ep_aux)
struct NsArgs { const float *mean, *invstd, *gamma, *beta; float leak; int group_images; };

static int run_dir(const CgsLayer& L, bool dirT, int B, const float* in, const float* w, const float* bias, float* out,
                   int epilogue, const float* ep_a, const float* ep_b, const float* ep_aux, void* ws, size_t ws_bytes,
                   int prepacked, hipStream_t s, const char* who, float* stat_part = nullptr, unsigned* sign_out = nullptr,
                   const unsigned* aux_signs = nullptr, const NsArgs* ns = nullptr) {
    cgs_note_flops(0.0);
    cgs_note_tail(0, 0);
    if (B <= 0) return cgs_set_error(CGS_EINVAL, "%s: B=%d", who, B);
    if (!in || !w || !out) return cgs_set_error(CGS_EINVAL, "%s: null tensor", who);
    if (epilogue < CGS_EPI_NONE || epilogue > CGS_EPI_TANH_BWD) return cgs_set_error(CGS_EINVAL, "%s: epilogue %d", who, epilogue);
    if (epilogue == CGS_EPI_AFFINE_RELU && (!ep_a || !ep_b)) return cgs_set_error(CGS_EINVAL, "%s: affine epilogue needs a,b", who);
    if (epilogue >= CGS_EPI_RELU_BWD_AFFINE && ((!ep_aux && !aux_signs) || (epilogue == CGS_EPI_RELU_BWD_AFFINE && !ep_a)))
        return cgs_set_error(CGS_EINVAL, "%s: backward epilogue %d needs aux%s", who, epilogue, epilogue == CGS_EPI_RELU_BWD_AFFINE ? " and a" : "");
    if (dirT && (L.sh > 2 || L.sw > 2)) return cgs_set_error(CGS_EINVAL, "%s: transposed direction supports stride <= 2", who);
    const bool rest_al = !(((uintptr_t)out & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)ep_a & 15) || ((uintptr_t)ep_b & 15) ||
                           ((uintptr_t)ep_aux & 15));
    const int fam = choose_family(L, dirT, B, epilogue, ws != nullptr, ws_bytes, !((uintptr_t)in & 15), !((uintptr_t)ws & 15), rest_al);
    if (stat_part && fam != CGS_FAMILY_IGEMM && fam != CGS_FAMILY_IGEMM_BX6) return cgs_set_error(CGS_EINVAL, "%s: fused statistics are an implicit-GEMM feature (see cgs_conv_stat_partials)", who);
    if (ns && (!stat_part || fam != CGS_FAMILY_IGEMM)) return cgs_set_error(CGS_EINVAL, "%s: norm-backward statistics are a feature of the exact-fp32 implicit GEMM (see cgs_conv_stat_layout)", who);
    if (sign_out && fam != CGS_FAMILY_IGEMM && fam != CGS_FAMILY_IGEMM_BX6) return cgs_set_error(CGS_EINVAL, "%s: this call cannot leave a sign mask (see cgs_conv_signs_ok)", who);
    if (aux_signs && fam != CGS_FAMILY_PATCH && fam != CGS_FAMILY_TAPS) return cgs_set_error(CGS_EINVAL, "%s: this call cannot take a sign mask (see cgs_conv_signs_ok)", who);
    if (((uintptr_t)sign_out & 3) || ((uintptr_t)aux_signs & 3)) return cgs_set_error(CGS_EINVAL, "%s: sign mask must be 4-byte aligned", who);
    switch (fam) {
        case CGS_FAMILY_QUAD:
            return cgs_convt_quad_launch(L, B, in, w, bias, out, epilogue, ep_a, ep_aux, (float*)ws, ws_bytes, prepacked, s);
        case CGS_FAMILY_SMALLN_T:
            return cgs_convt_smalln_launch(L, B, in, w, bias, out, epilogue, s);
        case CGS_FAMILY_SMALLN_F:
            return cgs_conv_smalln_f_launch(L, B, in, w, bias, out, epilogue, (float*)ws, ws_bytes, prepacked, s);
        case CGS_FAMILY_DOT:
            return cgs_conv_dot_launch(L, B, in, w, bias, out, epilogue, ep_a, ep_b, s);
        case CGS_FAMILY_TAPS:
            return cgs_conv_taps_launch(L, B, in, w, bias, out, epilogue, ep_a, ep_b, ep_aux, aux_signs, s);
        case CGS_FAMILY_PATCH:
            return cgs_conv_patch_launch(L, dirT, B, in, w, bias, out, epilogue, ep_a, ep_b, ep_aux, (float*)ws, ws_bytes, prepacked, s, aux_signs);
        default: break;
    }
    IgemmParams p;
    p.in = in; p.bias = bias; p.ep_a = ep_a; p.ep_b = ep_b; p.ep_aux = ep_aux; p.out = out; p.B = B; p.epilogue = epilogue;
    p.stat_part = stat_part;
    if (ns) { p.ns_mean = ns->mean; p.ns_inv = ns->invstd; p.ns_gamma = ns->gamma; p.ns_beta = ns->beta; p.ns_leak = ns->leak; p.ns_gimg = ns->group_images >= B ? 0 : ns->group_images; }
    p.sign_out = sign_out;
    p.sign_plane = (long)B * (dirT ? L.Hb * L.Wb : L.Hs * L.Ws);
    if (dirT) cgs_geom_T(L, p); else cgs_geom_F(L, p);
    const bool bx6 = fam == CGS_FAMILY_IGEMM_BX6;
    const size_t need = bx6 ? cgs_igemm_bx6_packed_bytes(p) : cgs_packed_floats(p) * sizeof(float);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    if (((uintptr_t)ws & 15) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) ||
        ((uintptr_t)ep_a & 15) || ((uintptr_t)ep_b & 15) || ((uintptr_t)ep_aux & 15))
        return cgs_set_error(CGS_EINVAL, "%s: pointers must be 16-byte aligned", who);
    p.wp = (const float*)ws;
    if (!prepacked) {
        int rc = bx6 ? cgs_pack_weights_bx6(p, L, dirT, w, ws, s) : cgs_pack_weights(p, L, dirT, w, (float*)ws, s);
        if (rc) return rc;
    }
    // one launch addresses its tensors with 32-bit byte offsets: split the batch so each stays < 2 GiB
    const size_t in_img = (size_t)p.Hin * p.Win * p.Cred * 4, out_img = (size_t)p.Hout * p.Wout * p.N * 4;
    const size_t per = in_img > out_img ? in_img : out_img;
    long chunk = (long)(0x7fffffffUL / per);
    if (chunk < 1) return cgs_set_error(CGS_EINVAL, "%s: one image exceeds 2 GiB", who);
    if (stat_part && chunk < B) return cgs_set_error(CGS_EINVAL, "%s: fused statistics need the batch in one launch", who);
    for (long b0 = 0; b0 < B; b0 += chunk) {
        p.B = (int)(B - b0 < chunk ? B - b0 : chunk);
        p.in = in + (size_t)b0 * (in_img / 4);
        p.out = out + (size_t)b0 * (out_img / 4);
        p.ep_aux = ep_aux ? ep_aux + (size_t)b0 * (out_img / 4) : nullptr;
        p.sign_out = sign_out ? sign_out + (size_t)b0 * p.Hout * p.Wout : nullptr;      // (plane-major: the chunk's pixels inside every plane)
        int rc = bx6 ? cgs_igemm_bx6_launch(p, s, (char*)ws + need, ws_bytes - need) : cgs_igemm_launch(p, (char*)ws + need, ws_bytes - need, s);
        if (rc) return rc;
    }
    return CGS_OK;
}

extern "C" {

static size_t conv_packed_bytes(int op, int kh, int kw, int sh, int sw, int Cin, int Cout);

size_t cgs_conv_ws_bytes(int op, int kh, int kw, int sh, int sw, int Cin, int Cout) {
    return conv_packed_bytes(op, kh, kw, sh, sw, Cin, Cout);
}

size_t cgs_conv_ws_bytes_for(int op, int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw) {
    const size_t packed = conv_packed_bytes(op, kh, kw, sh, sw, Cin, Cout);
    if (B <= 0 || H <= 0 || W <= 0 || packed == 0) return packed;
    // H, W are the op's INPUT spatial size for conv fwd / deconv fwd and for their backward-datas the size of
    // the tensor the gradient is taken w.r.t. (the arguments of the entry points)
    const bool deconv = (op == CGS_DECONV_FWD || op == CGS_DECONV_BWD_DATA);
    const bool dirT = (op == CGS_CONV_BWD_DATA || op == CGS_DECONV_FWD);
    CgsLayer L;
    L.kh = kh; L.kw = kw; L.sh = sh; L.sw = sw;
    if (!deconv) { L.Hb = H; L.Wb = W; L.Cb = Cin; L.Hs = cgs_ceil_div(H, sh); L.Ws = cgs_ceil_div(W, sw); L.Cs = Cout; }
    else { L.Hs = H; L.Ws = W; L.Cs = Cin; L.Hb = H * sh; L.Wb = W * sw; L.Cb = Cout; }   // bound: output = stride * input
    if (dirT && (sh > 2 || sw > 2)) return packed;
    IgemmParams p;
    p.B = B; p.stat_part = nullptr; p.sign_out = nullptr; p.sign_plane = 0;
    if (dirT) cgs_geom_T(L, p); else cgs_geom_F(L, p);
    return packed + cgs_igemm_slab_bytes(p);
}

int cgs_conv_family(int op, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int epilogue,
                    size_t ws_bytes) {
    if (op < CGS_CONV_FWD || op > CGS_DECONV_BWD_DATA || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 ||
        sh <= 0 || sw <= 0)
        return cgs_set_error(CGS_EINVAL, "conv_family: bad argument");
    const bool deconv = (op == CGS_DECONV_FWD || op == CGS_DECONV_BWD_DATA);
    const bool dirT = (op == CGS_CONV_BWD_DATA || op == CGS_DECONV_FWD);
    CgsLayer L;
    L.kh = kh; L.kw = kw; L.sh = sh; L.sw = sw;
    if (!deconv) { L.Hb = H; L.Wb = W; L.Cb = Cin; L.Hs = cgs_ceil_div(H, sh); L.Ws = cgs_ceil_div(W, sw); L.Cs = Cout; }
    else { L.Hs = H; L.Ws = W; L.Cs = Cin; L.Hb = Ho; L.Wb = Wo; L.Cb = Cout; }
    return choose_family(L, dirT, B, epilogue, ws_bytes > 0, ws_bytes, true, true, true);
}

static size_t conv_packed_bytes(int op, int kh, int kw, int sh, int sw, int Cin, int Cout) {
    if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || Cin <= 0 || Cout <= 0) return 0;
    // weights are [kh][kw][Cb][Cs]: conv Cb=Cin,Cs=Cout; deconv Cb=Cout,Cs=Cin
    const bool deconv = (op == CGS_DECONV_FWD || op == CGS_DECONV_BWD_DATA);
    const bool dirT = (op == CGS_CONV_BWD_DATA || op == CGS_DECONV_FWD);
    const int Cb = deconv ? Cout : Cin, Cs = deconv ? Cin : Cout;
    // (the split-bf16 form keeps three bf16 planes = 6 bytes per weight: sized for it wherever its geometry applies, whatever the
    // calling thread's contraction mode, so a workspace serves both)
    const int Cred = dirT ? Cs : Cb, Nn = dirT ? Cb : Cs;
    const size_t per_w = ((Cred % 32) == 0 && (Nn % 64) == 0 && kh <= 16 && kw <= 16) ? 6 : sizeof(float);
    if (!dirT) return (size_t)cgs_round_up(kh * kw * Cb, CGS_BK) * cgs_round_up(Cs, 64) * per_w;
    // T: per parity class, taps of that class (independent of the spatial size: pads only permute classes)
    size_t n = 0;
    for (int a = 0; a < sh; ++a)
        for (int b = 0; b < sw; ++b) {
            const int nty = a < kh ? (kh - a + sh - 1) / sh : 0, ntx = b < kw ? (kw - b + sw - 1) / sw : 0;
            n += (size_t)cgs_round_up(nty * ntx * Cs, CGS_BK) * cgs_round_up(Cb, 64);
        }
    if (Cb <= 4 && sh == 2 && sw == 2) {      // quad-form small-N kernel keeps its own packed copy
        const size_t q = cgs_convt_quad_ws_floats_bound(kh, kw, Cs);
        if (q > n) n = q;
    }
    return n * (per_w > sizeof(float) && !(Cb <= 4 && sh == 2 && sw == 2) ? per_w : sizeof(float));
}

int cgs_conv2d_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                        int Cout, int kh, int kw, int sh, int sw, int epilogue, const float* ep_a, const float* ep_b,
                        void* ws, size_t ws_bytes, int ws_prepacked, void* stream) {
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, H, W, Cin, cgs_ceil_div(H, sh > 0 ? sh : 1), cgs_ceil_div(W, sw > 0 ? sw : 1), Cout, "conv2d_nhwc_fwd");
    if (rc) return rc;
    return run_dir(L, false, B, x, w, bias, y, epilogue, ep_a, ep_b, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream, "conv2d_nhwc_fwd");
}

int cgs_conv_stat_partials(int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw, size_t ws_bytes) {
    CgsLayer L;
    if (make_layer(L, kh, kw, sh, sw, H, W, Cin, cgs_ceil_div(H, sh > 0 ? sh : 1), cgs_ceil_div(W, sw > 0 ? sw : 1), Cout, "conv_stat_partials")) return 0;
    const int fam = choose_family(L, false, B, CGS_EPI_NONE, ws_bytes > 0, ws_bytes, true, true, true);
    if (B <= 0 || (Cout & 3) || (fam != CGS_FAMILY_IGEMM && fam != CGS_FAMILY_IGEMM_BX6)) return 0;       // (both leave the same partial rows)
    const size_t per = (size_t)(L.Hb * L.Wb * L.Cb > L.Hs * L.Ws * L.Cs ? L.Hb * L.Wb * L.Cb : L.Hs * L.Ws * L.Cs) * 4;
    if ((size_t)B * per > 0x7fffffffUL) return 0;                    // the entry point would split the batch
    const long M = (long)B * L.Hs * L.Ws;
    return (int)(2 * ((M + 127) / 128));                             // one partial row per (128-row tile, wave row)
}

// Where the partial rows of one GROUP of group_images consecutive images lie in the [rows][2][Cout] buffer a *_fwd_stats call fills
// (one row per 64 GEMM rows; the parity classes of a transposed convolution back to back): group g owns, for every segment
// s < nseg, the rows_per_seg rows from s * seg_stride + g * rows_per_seg.  Image-major launches: a segment per parity class;
// pixel-major launches (whole 128-image tiles): a segment per (class, base pixel).  Returns the total row count, 0 = unavailable.
int cgs_conv_stat_layout(int op, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int group_images,
                         size_t ws_bytes, int* rows_per_seg, int* nseg, int* seg_stride) {
    if (op < CGS_CONV_FWD || op > CGS_DECONV_BWD_DATA || B <= 0 || group_images <= 0 || (B % group_images) || !rows_per_seg || !nseg || !seg_stride)
        return 0;
    // the launch's direction and the channel count of the tensor it WRITES (the statistics' columns): the forward ops write Cout channels,
    // the backward-data ops (norm-backward statistics, cgs_*_bwd_data_nstats) the gradient w.r.t. their input: Cin channels
    const bool deconv = (op == CGS_DECONV_FWD || op == CGS_DECONV_BWD_DATA), bwd = (op == CGS_CONV_BWD_DATA || op == CGS_DECONV_BWD_DATA);
    const bool dirT = (op == CGS_CONV_BWD_DATA || op == CGS_DECONV_FWD);
    if ((bwd ? Cin : Cout) & 3) return 0;
    CgsLayer L;
    if (!deconv) { if (make_layer(L, kh, kw, sh, sw, H, W, Cin, cgs_ceil_div(H, sh > 0 ? sh : 1), cgs_ceil_div(W, sw > 0 ? sw : 1), Cout, "conv_stat_layout")) return 0; }
    else if (make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "conv_stat_layout")) return 0;
    if (dirT && (sh > 2 || sw > 2)) return 0;
    const int fam = choose_family(L, dirT, B, CGS_EPI_NONE, ws_bytes > 0, ws_bytes, true, true, true);
    if (fam != CGS_FAMILY_IGEMM && (bwd || fam != CGS_FAMILY_IGEMM_BX6)) return 0;           // (the backward sums: the exact-fp32 kernel only)
    const size_t per = (size_t)(L.Hb * L.Wb * L.Cb > L.Hs * L.Ws * L.Cs ? L.Hb * L.Wb * L.Cb : L.Hs * L.Ws * L.Cs) * 4;
    if ((size_t)B * per > 0x7fffffffUL) return 0;                    // the entry point would split the batch
    IgemmParams p;
    p.B = B; p.stat_part = nullptr; p.sign_out = nullptr; p.sign_plane = 0; p.epilogue = CGS_EPI_NONE;
    if (dirT) cgs_geom_T(L, p); else cgs_geom_F(L, p);
    const int RC = p.cls[0].R * p.cls[0].C;
    for (int i = 1; i < p.nclasses; ++i)
        if (p.cls[i].R * p.cls[i].C != RC) return 0;                 // (odd output sizes: parity classes of different size)
    if (RC <= 0) return 0;
    const long M = (long)B * RC;
    const long cls_rows = 2 * ((M + 127) / 128);
    if (cls_rows * p.nclasses > 0x7fffffffL) return 0;
    if (group_images == B) {                                         // one group: every partial row belongs to it, whatever the row order
        *rows_per_seg = (int)(cls_rows * p.nclasses); *nseg = 1; *seg_stride = 0;
        return (int)(cls_rows * p.nclasses);
    }
    const int order = cgs_igemm_row_order(p);
    if (order == 0) {                                                // rows (image, pixel): a group's rows are contiguous inside every class
        if (((long)group_images * RC) % 64) return 0;
        *rows_per_seg = (int)((long)group_images * RC / 64); *nseg = p.nclasses; *seg_stride = (int)cls_rows;
    } else if (order == 2) {                                         // rows (pixel, image), B % 128 == 0: 64 consecutive images of one pixel per row
        if (group_images % 64) return 0;
        *rows_per_seg = group_images / 64; *nseg = p.nclasses * RC; *seg_stride = B / 64;
    } else {
        return 0;
    }
    return (int)(cls_rows * p.nclasses);
}

// Backward-data of a conv / deconv whose result is the gradient w.r.t. the OUTPUT of a norm (+ lrelu) over x_norm (same shape as the
// result): the launch also leaves the norm backward's two column sums as partial rows (cgs_conv_stat_layout with the *_BWD_DATA op says
// where a group's rows lie; cgs_norm_lrelu_bwd_from_partials consumes them).  The result itself is the plain backward-data (no epilogue).
static int nstats_check(const char* who, int G, int C, const float* x_norm, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        float* stat_part, size_t stat_part_bytes) {
    if (G == 0) return cgs_set_error(CGS_EINVAL, "%s: not available for this call (cgs_conv_stat_layout == 0)", who);
    if (!x_norm || !mean || !invstd || !gamma || !beta) return cgs_set_error(CGS_EINVAL, "%s: null statistics argument", who);
    if (((uintptr_t)x_norm | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)stat_part) & 15)
        return cgs_set_error(CGS_EINVAL, "%s: pointers must be 16-byte aligned", who);
    if (!stat_part || stat_part_bytes < (size_t)G * 2 * C * sizeof(float))
        return cgs_set_error(CGS_EWORKSPACE, "%s: partials buffer %zu < %zu bytes", who, stat_part_bytes, (size_t)G * 2 * C * sizeof(float));
    return CGS_OK;
}

int cgs_conv2d_nhwc_bwd_data_nstats(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh,
                                    int sw, const float* x_norm, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    float leak, int group_images, void* ws, size_t ws_bytes, int ws_prepacked, float* stat_part,
                                    size_t stat_part_bytes, void* stream) {
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, H, W, Cin, cgs_ceil_div(H, sh > 0 ? sh : 1), cgs_ceil_div(W, sw > 0 ? sw : 1), Cout, "conv2d_nhwc_bwd_data_nstats");
    if (rc) return rc;
    int a, b, c;
    const int G = cgs_conv_stat_layout(CGS_CONV_BWD_DATA, B, H, W, Cin, 0, 0, Cout, kh, kw, sh, sw, group_images, ws_bytes, &a, &b, &c);
    if ((rc = nstats_check("conv2d_nhwc_bwd_data_nstats", G, Cin, x_norm, mean, invstd, gamma, beta, stat_part, stat_part_bytes))) return rc;
    const NsArgs ns{mean, invstd, gamma, beta, leak, group_images};
    return run_dir(L, true, B, dy, w, nullptr, dx, CGS_EPI_NONE, nullptr, nullptr, x_norm, ws, ws_bytes, ws_prepacked, (hipStream_t)stream,
                   "conv2d_nhwc_bwd_data_nstats", stat_part, nullptr, nullptr, &ns);
}

int cgs_deconv2d_nhwc_bwd_data_nstats(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh,
                                      int kw, int sh, int sw, const float* x_norm, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, float leak, int group_images, void* ws, size_t ws_bytes, int ws_prepacked,
                                      float* stat_part, size_t stat_part_bytes, void* stream) {
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "deconv2d_nhwc_bwd_data_nstats");
    if (rc) return rc;
    int a, b, c;
    const int G = cgs_conv_stat_layout(CGS_DECONV_BWD_DATA, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, group_images, ws_bytes, &a, &b, &c);
    if ((rc = nstats_check("deconv2d_nhwc_bwd_data_nstats", G, Cin, x_norm, mean, invstd, gamma, beta, stat_part, stat_part_bytes))) return rc;
    const NsArgs ns{mean, invstd, gamma, beta, leak, group_images};
    return run_dir(L, false, B, dy, w, nullptr, dx, CGS_EPI_NONE, nullptr, nullptr, x_norm, ws, ws_bytes, ws_prepacked, (hipStream_t)stream,
                   "deconv2d_nhwc_bwd_data_nstats", stat_part, nullptr, nullptr, &ns);
}

int cgs_deconv2d_nhwc_fwd_stats(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin, int Ho, int Wo,
                                int Cout, int kh, int kw, int sh, int sw, void* ws, size_t ws_bytes, int ws_prepacked, float* stat_part,
                                size_t stat_part_bytes, void* stream) {
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "deconv2d_nhwc_fwd_stats");
    if (rc) return rc;
    int a, b, c;
    const int G = cgs_conv_stat_layout(CGS_DECONV_FWD, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, B, ws_bytes, &a, &b, &c);
    if (G == 0) return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_fwd_stats: not available for this call (cgs_conv_stat_layout == 0)");
    if (!stat_part || stat_part_bytes < (size_t)G * 2 * Cout * sizeof(float))
        return cgs_set_error(CGS_EWORKSPACE, "deconv2d_nhwc_fwd_stats: partials buffer %zu < %zu bytes", stat_part_bytes, (size_t)G * 2 * Cout * sizeof(float));
    if ((uintptr_t)stat_part & 15) return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_fwd_stats: partials buffer must be 16-byte aligned");
    return run_dir(L, true, B, x, w, bias, y, CGS_EPI_NONE, nullptr, nullptr, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream,
                   "deconv2d_nhwc_fwd_stats", stat_part);
}

int cgs_conv2d_nhwc_fwd_stats(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                              int Cout, int kh, int kw, int sh, int sw, void* ws, size_t ws_bytes, int ws_prepacked,
                              float* stat_part, size_t stat_part_bytes, void* stream) {
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, H, W, Cin, cgs_ceil_div(H, sh > 0 ? sh : 1), cgs_ceil_div(W, sw > 0 ? sw : 1), Cout, "conv2d_nhwc_fwd_stats");
    if (rc) return rc;
    const int G = cgs_conv_stat_partials(B, H, W, Cin, Cout, kh, kw, sh, sw, ws_bytes);
    if (G == 0) return cgs_set_error(CGS_EINVAL, "conv2d_nhwc_fwd_stats: not available for this call (cgs_conv_stat_partials == 0)");
    if (!stat_part || stat_part_bytes < (size_t)G * 2 * Cout * sizeof(float))
        return cgs_set_error(CGS_EWORKSPACE, "conv2d_nhwc_fwd_stats: partials buffer %zu < %zu bytes", stat_part_bytes, (size_t)G * 2 * Cout * sizeof(float));
    if ((uintptr_t)stat_part & 15) return cgs_set_error(CGS_EINVAL, "conv2d_nhwc_fwd_stats: partials buffer must be 16-byte aligned");
    return run_dir(L, false, B, x, w, bias, y, CGS_EPI_NONE, nullptr, nullptr, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream,
                   "conv2d_nhwc_fwd_stats", stat_part);
}

int cgs_conv2d_nhwc_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout, int kh,
                             int kw, int sh, int sw, int epilogue, const float* ep_a, const float* ep_aux, void* ws,
                             size_t ws_bytes, int ws_prepacked, void* stream) {
    if (epilogue != CGS_EPI_NONE && epilogue < CGS_EPI_RELU_BWD_AFFINE) return cgs_set_error(CGS_EINVAL, "conv2d_nhwc_bwd_data: epilogue %d", epilogue);
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, H, W, Cin, cgs_ceil_div(H, sh > 0 ? sh : 1), cgs_ceil_div(W, sw > 0 ? sw : 1), Cout, "conv2d_nhwc_bwd_data");
    if (rc) return rc;
    return run_dir(L, true, B, dy, w, nullptr, dx, epilogue, ep_a, nullptr, ep_aux, ws, ws_bytes, ws_prepacked, (hipStream_t)stream, "conv2d_nhwc_bwd_data");
}

int cgs_deconv2d_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                          int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int epilogue, const float* ep_a,
                          const float* ep_b, void* ws, size_t ws_bytes, int ws_prepacked, void* stream) {
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "deconv2d_nhwc_fwd");
    if (rc) return rc;
    return run_dir(L, true, B, x, w, bias, y, epilogue, ep_a, ep_b, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream, "deconv2d_nhwc_fwd");
}

int cgs_deconv2d_nhwc_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo,
                               int Cout, int kh, int kw, int sh, int sw, int epilogue, const float* ep_a,
                               const float* ep_aux, void* ws, size_t ws_bytes, int ws_prepacked, void* stream) {
    if (epilogue != CGS_EPI_NONE && epilogue < CGS_EPI_RELU_BWD_AFFINE) return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_bwd_data: epilogue %d", epilogue);
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "deconv2d_nhwc_bwd_data");
    if (rc) return rc;
    return run_dir(L, false, B, dy, w, nullptr, dx, epilogue, ep_a, nullptr, ep_aux, ws, ws_bytes, ws_prepacked, (hipStream_t)stream, "deconv2d_nhwc_bwd_data");
}

// ---- sign masks (cgs_hip.h): the activation gradient of relu / lrelu needs one bit per element of the saved activation ----
int cgs_conv_signs_ok(int op, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int epilogue,
                      size_t ws_bytes) {
    const int fam = cgs_conv_family(op, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, epilogue, ws_bytes);
    if (fam < 0) return 0;
    const bool deconv = (op == CGS_DECONV_FWD || op == CGS_DECONV_BWD_DATA);
    const bool dirT = (op == CGS_CONV_BWD_DATA || op == CGS_DECONV_FWD);
    CgsLayer L;
    L.kh = kh; L.kw = kw; L.sh = sh; L.sw = sw;
    if (!deconv) { L.Hb = H; L.Wb = W; L.Cb = Cin; L.Hs = cgs_ceil_div(H, sh); L.Ws = cgs_ceil_div(W, sw); L.Cs = Cout; }
    else { L.Hs = H; L.Ws = W; L.Cs = Cin; L.Hb = Ho; L.Wb = Wo; L.Cb = Cout; }
    if (epilogue >= CGS_EPI_RELU_BWD_AFFINE)          // consumer: a backward-data whose epilogue applies relu' / lrelu'
        return (fam == CGS_FAMILY_PATCH && cgs_conv_patch_signs_ok(L, dirT, epilogue)) || (fam == CGS_FAMILY_TAPS && cgs_conv_taps_signs_ok(L, epilogue));
    if (fam == CGS_FAMILY_IGEMM_BX6)            // producer on the split-bf16 kernel: whole 64-column wave tiles, never split over K
        return (epilogue == CGS_EPI_AFFINE_RELU || epilogue == CGS_EPI_LRELU) &&
               (size_t)B * ((size_t)L.Hb * L.Wb * L.Cb > (size_t)L.Hs * L.Ws * L.Cs ? (size_t)L.Hb * L.Wb * L.Cb : (size_t)L.Hs * L.Ws * L.Cs) * 4 <= 0x7fffffffUL;
    if (fam != CGS_FAMILY_IGEMM || (dirT && (sh > 2 || sw > 2))) return 0;        // producer: a forward with the relu / lrelu epilogue
    IgemmParams p;
    p.B = B; p.stat_part = nullptr; p.sign_out = nullptr; p.sign_plane = 0; p.epilogue = epilogue;
    if (dirT) cgs_geom_T(L, p); else cgs_geom_F(L, p);
    const size_t in_img = (size_t)p.Hin * p.Win * p.Cred * 4, out_img = (size_t)p.Hout * p.Wout * p.N * 4;
    if ((size_t)B * (in_img > out_img ? in_img : out_img) > 0x7fffffffUL) return 0;      // (a split batch may fall under the split-K threshold)
    return cgs_igemm_signs_ok(p);
}

int cgs_deconv2d_nhwc_fwd_signs(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                                int Ho, int Wo, int Cout, int kh, int kw, int sh, int sw, int epilogue, const float* ep_a,
                                const float* ep_b, unsigned* signs, void* ws, size_t ws_bytes, int ws_prepacked, void* stream) {
    if (!signs) return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_fwd_signs: null sign mask");
    if (!cgs_conv_signs_ok(CGS_DECONV_FWD, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, epilogue, ws_bytes))
        return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_fwd_signs: not available for this call (cgs_conv_signs_ok == 0)");
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "deconv2d_nhwc_fwd_signs");
    if (rc) return rc;
    return run_dir(L, true, B, x, w, bias, y, epilogue, ep_a, ep_b, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream,
                   "deconv2d_nhwc_fwd_signs", nullptr, signs);
}

int cgs_deconv2d_nhwc_bwd_data_signs(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo,
                                     int Cout, int kh, int kw, int sh, int sw, int epilogue, const float* ep_a,
                                     const unsigned* aux_signs, void* ws, size_t ws_bytes, int ws_prepacked, void* stream) {
    if (!aux_signs) return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_bwd_data_signs: null sign mask");
    if (!cgs_conv_signs_ok(CGS_DECONV_BWD_DATA, B, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, epilogue, ws_bytes))
        return cgs_set_error(CGS_EINVAL, "deconv2d_nhwc_bwd_data_signs: not available for this call (cgs_conv_signs_ok == 0)");
    CgsLayer L;
    int rc = make_layer(L, kh, kw, sh, sw, Ho, Wo, Cout, H, W, Cin, "deconv2d_nhwc_bwd_data_signs");
    if (rc) return rc;
    return run_dir(L, false, B, dy, w, nullptr, dx, epilogue, ep_a, nullptr, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream,
                   "deconv2d_nhwc_bwd_data_signs", nullptr, nullptr, aux_signs);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// linear with a single output column (the D logit head, nsgan/GAN.py:68): wave-reduced dot product
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_out1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int B, int K, int epilogue) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* row = x + (size_t)b * K;
    float s = 0.f;
    if ((K & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const float4 a = *(const float4*)(row + k), ww = *(const float4*)(w + k);
            s = fmaf(a.x, ww.x, s); s = fmaf(a.y, ww.y, s); s = fmaf(a.z, ww.z, s); s = fmaf(a.w, ww.w, s);
        }
    } else {
        for (int k = lane; k < K; k += 64) s = fmaf(row[k], w[k], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        float v = s + (bias ? bias[0] : 0.f);
        if (epilogue == CGS_EPI_LRELU) v = fmaxf(v, 0.2f * v);
        y[b] = v;
    }
}

// the logit head and the loss seed in one launch (the last launches of every D forward of the refinement loop, both a few microseconds of
// latency): y[b] = x[b] . w + bias; dl[b] = sigmoid(y[b]) - 1 (d softplus(-y) / dy, the stable form of bce_rowmean_kernel); lm[b] = y[b]
__global__ __launch_bounds__(256) void linear_out1_bce_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, float* __restrict__ dl, float* __restrict__ lm, int B, int K) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* row = x + (size_t)b * K;
    float s = 0.f;
    if ((K & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const float4 a = *(const float4*)(row + k), ww = *(const float4*)(w + k);
            s = fmaf(a.x, ww.x, s); s = fmaf(a.y, ww.y, s); s = fmaf(a.z, ww.z, s); s = fmaf(a.w, ww.w, s);
        }
    } else {
        for (int k = lane; k < K; k += 64) s = fmaf(row[k], w[k], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        const float v = s + (bias ? bias[0] : 0.f);
        y[b] = v;
        dl[b] = v >= 0.f ? -expf(-v) / (1.f + expf(-v)) : -1.f / (1.f + expf(v));
        lm[b] = v;
    }
}

__global__ __launch_bounds__(256) void linear_out1_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int B, int K) {
    const size_t n = (size_t)B * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / K), k = (int)(i - (size_t)b * K);
        dx[i] = dy[b] * w[k];
    }
}

extern "C" {

int cgs_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in, int out, int epilogue,
                   void* ws, size_t ws_bytes, int ws_prepacked, void* stream) {
    if (B <= 0 || in <= 0 || out <= 0) return cgs_set_error(CGS_EINVAL, "linear_fwd: B=%d in=%d out=%d", B, in, out);
    if (epilogue != CGS_EPI_NONE && epilogue != CGS_EPI_LRELU) return cgs_set_error(CGS_EINVAL, "linear_fwd: epilogue %d", epilogue);
    if (out == 1) {
        hipLaunchKernelGGL(linear_out1_fwd_kernel, dim3(cgs_ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, in, epilogue);
        CGS_CHECK_LAUNCH("linear_out1_fwd");
        return CGS_OK;
    }
    CgsLayer L;
    int rc = make_layer(L, 1, 1, 1, 1, 1, 1, in, 1, 1, out, "linear_fwd");
    if (rc) return rc;
    return run_dir(L, false, B, x, w, bias, y, epilogue, nullptr, nullptr, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream, "linear_fwd");
}

int cgs_linear_out1_bce(const float* x, const float* w, const float* bias, float* logits, float* dlogits, float* logit_mean, int B, int in,
                        void* stream) {
    if (B <= 0 || in <= 0 || !x || !w || !logits || !dlogits || !logit_mean) return cgs_set_error(CGS_EINVAL, "linear_out1_bce: bad argument");
    hipLaunchKernelGGL(linear_out1_bce_kernel, dim3(cgs_ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, logits, dlogits, logit_mean, B, in);
    CGS_CHECK_LAUNCH("linear_out1_bce");
    return CGS_OK;
}

int cgs_linear_bwd_data(const float* dy, const float* w, float* dx, int B, int in, int out, void* ws, size_t ws_bytes,
                        int ws_prepacked, void* stream) {
    if (B <= 0 || in <= 0 || out <= 0) return cgs_set_error(CGS_EINVAL, "linear_bwd_data: B=%d in=%d out=%d", B, in, out);
    if (out == 1) {
        size_t n = (size_t)B * in;
        unsigned blocks = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
        hipLaunchKernelGGL(linear_out1_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, w, dx, B, in);
        CGS_CHECK_LAUNCH("linear_out1_bwd");
        return CGS_OK;
    }
    CgsLayer L;
    int rc = make_layer(L, 1, 1, 1, 1, 1, 1, in, 1, 1, out, "linear_bwd_data");
    if (rc) return rc;
    return run_dir(L, true, B, dy, w, nullptr, dx, CGS_EPI_NONE, nullptr, nullptr, nullptr, ws, ws_bytes, ws_prepacked, (hipStream_t)stream, "linear_bwd_data");
}

}  // extern "C"
