// Host-side sanitizer run of libcgs_hip ("ASAN build of the C-ABI host side").  Built with
// -fsanitize=address (host code only) and run WITHOUT a GPU: every entry point does its argument checking, geometry, tile
// planning (tap tables, LPT / balanced tile dealing, split-K sizing, workspace arithmetic) on the host and then fails at the
// first HIP call with CGS_ELAUNCH ("no ROCm-capable device") -- which is exactly the part AddressSanitizer can watch here.
// Device pointers are fake non-null values that the host never dereferences.
#include <stdio.h>
#include <stdlib.h>

#include "cgs_hip.h"

static int bad = 0;
static void expect(int rc, const char* what) {
    if (rc != CGS_ELAUNCH && rc != CGS_EINVAL && rc != CGS_EWORKSPACE && rc != CGS_OK) { printf("unexpected rc %d from %s\n", rc, what); ++bad; }
}

int main() {
    float* P = (float*)0x10000;                 // 16-byte aligned fake device pointer
    const size_t WS = (size_t)1 << 30;
    struct { int B, H, Cin, Cout, k, s; } convs[] = {
        {1024, 64, 3, 64, 5, 2}, {1024, 32, 64, 128, 5, 2}, {1024, 16, 128, 256, 5, 2}, {1024, 8, 256, 512, 5, 2},   // dcgan64 D
        {256, 32, 3, 64, 5, 2}, {256, 4, 256, 512, 5, 2}, {64, 28, 1, 64, 4, 2}, {64, 14, 64, 128, 4, 2},              // dcgan32, mnist
        {8, 256, 3, 64, 7, 1}, {8, 64, 256, 256, 3, 1}, {8, 32, 512, 1, 4, 1}, {8, 256, 64, 3, 7, 1},                  // cyclegan256
        {2048, 28, 1, 64, 4, 2}, {3, 27, 1, 96, 4, 2}, {8, 32, 512, 4, 4, 1}, {2, 8, 64, 3, 5, 2},                      // taps / deep-dot forms
        {1, 1, 32, 64, 1, 1}, {130, 9, 96, 40, 3, 2}, {5, 6, 32, 64, 3, 2}, {2048, 8, 64, 64, 5, 2}, {3, 17, 5, 7, 5, 2}};
    for (auto& c : convs) {
        const int Ho = (c.H + c.s - 1) / c.s;
        for (int epi = CGS_EPI_NONE; epi <= CGS_EPI_TANH; ++epi)
            expect(cgs_conv2d_nhwc_fwd(P, P, P, P, c.B, c.H, c.H, c.Cin, c.Cout, c.k, c.k, c.s, c.s, epi, P, P, P, WS, 0, nullptr), "conv fwd");
        for (int epi : {CGS_EPI_NONE, CGS_EPI_RELU_BWD_AFFINE, CGS_EPI_LRELU_BWD, CGS_EPI_TANH_BWD})
            expect(cgs_conv2d_nhwc_bwd_data(P, P, P, c.B, c.H, c.H, c.Cin, c.Cout, c.k, c.k, c.s, c.s, epi, P, P, P, WS, 0, nullptr), "conv bwd");
        // the same relation as a deconv (small -> big)
        expect(cgs_deconv2d_nhwc_fwd(P, P, P, P, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, CGS_EPI_TANH, nullptr, nullptr, P, WS, 0, nullptr), "deconv fwd");
        expect(cgs_deconv2d_nhwc_bwd_data(P, P, P, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, CGS_EPI_NONE, nullptr, nullptr, P, WS, 0, nullptr), "deconv bwd");
        (void)cgs_conv_ws_bytes(CGS_CONV_FWD, c.k, c.k, c.s, c.s, c.Cin, c.Cout);
        for (int op = CGS_CONV_FWD; op <= CGS_DECONV_BWD_DATA; ++op) {
            (void)cgs_conv_ws_bytes_for(op, c.B, op >= CGS_DECONV_FWD ? Ho : c.H, op >= CGS_DECONV_FWD ? Ho : c.H, op >= CGS_DECONV_FWD ? c.Cout : c.Cin,
                                        op >= CGS_DECONV_FWD ? c.Cin : c.Cout, c.k, c.k, c.s, c.s);
            (void)cgs_conv_family(op, c.B, op >= CGS_DECONV_FWD ? Ho : c.H, op >= CGS_DECONV_FWD ? Ho : c.H, op >= CGS_DECONV_FWD ? c.Cout : c.Cin, c.H, c.H,
                                  op >= CGS_DECONV_FWD ? c.Cin : c.Cout, c.k, c.k, c.s, c.s, CGS_EPI_NONE, WS);
        }
        // sign masks: the query for both roles, and the two entry points (refused with CGS_EINVAL where the query says no)
        for (int epi : {CGS_EPI_LRELU, CGS_EPI_AFFINE_RELU})
            if (cgs_conv_signs_ok(CGS_DECONV_FWD, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, epi, WS) || c.B == 130)
                expect(cgs_deconv2d_nhwc_fwd_signs(P, P, P, P, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, epi, P, P, (unsigned*)P, P, WS, 0, nullptr), "deconv fwd signs");
        for (int epi : {CGS_EPI_RELU_BWD_AFFINE, CGS_EPI_LRELU_BWD, CGS_EPI_TANH_BWD}) {
            (void)cgs_conv_signs_ok(CGS_DECONV_BWD_DATA, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, epi, WS);
            expect(cgs_deconv2d_nhwc_bwd_data_signs(P, P, P, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, epi, P, (const unsigned*)P, P, WS, 0, nullptr), "deconv bwd signs");
        }
        const int G = cgs_conv_stat_partials(c.B, c.H, c.H, c.Cin, c.Cout, c.k, c.k, c.s, c.s, WS);
        expect(cgs_conv2d_nhwc_fwd_stats(P, P, P, P, c.B, c.H, c.H, c.Cin, c.Cout, c.k, c.k, c.s, c.s, P, WS, 0, P, (size_t)(G > 0 ? G : 1) * 2 * c.Cout * 4, nullptr), "conv fwd stats");
        // statistics per group of images (instance norm: 1; a logical batch of 64), forward and transposed direction
        for (int grp : {1, 64, c.B}) {
            int a = 0, b = 0, d = 0;
            const int rows = cgs_conv_stat_layout(CGS_CONV_FWD, c.B, c.H, c.H, c.Cin, 0, 0, c.Cout, c.k, c.k, c.s, c.s, grp, WS, &a, &b, &d);
            const int rowsT = cgs_conv_stat_layout(CGS_DECONV_FWD, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, grp, WS, &a, &b, &d);
            if (rowsT > 0 || c.B == 130)
                expect(cgs_deconv2d_nhwc_fwd_stats(P, P, P, P, c.B, Ho, Ho, c.Cout, c.H, c.H, c.Cin, c.k, c.k, c.s, c.s, P, WS, 0, P,
                                                   (size_t)(rowsT > 0 ? rowsT : 1) * 2 * c.Cin * 4, nullptr), "deconv fwd stats");
            if (rows > 0)
                expect(cgs_groupnorm_lrelu_fwd_from_partials(P, P, c.B / grp, a > 0 ? a : 1, b > 0 ? b : 1, d, P, P, 1e-5f, 0.2f, P, P, P, grp * Ho * Ho, c.Cout, P,
                                                             cgs_instnorm_ws_bytes(c.B / grp, grp * Ho * Ho, c.Cout), nullptr), "groupnorm from partials");
        }
        expect(cgs_conv2d_nhwc_bwd_weight(P, P, P, c.B, c.H, c.H, c.Cin, c.Cout, c.k, c.k, c.s, c.s, 0, P, WS, nullptr), "conv wgrad");
    }
    expect(cgs_linear_fwd(P, P, P, P, 64, 6272, 1024, CGS_EPI_LRELU, P, WS, 0, nullptr), "linear fwd");
    expect(cgs_linear_bwd_data(P, P, P, 64, 6272, 1024, P, WS, 0, nullptr), "linear bwd");
    expect(cgs_linear_fwd(P, P, P, P, 1024, 8192, 1, CGS_EPI_NONE, nullptr, 0, 0, nullptr), "linear out1");
    expect(cgs_bn_train_lrelu_fwd(P, P, P, 1e-5f, 0.2f, P, P, P, 262144, 128, P, cgs_bn_ws_bytes(262144, 128), nullptr), "bn fwd");
    expect(cgs_instnorm_lrelu_fwd(P, P, P, 1e-5f, 0.f, P, P, P, 8, 4096, 256, P, cgs_instnorm_ws_bytes(8, 4096, 256), nullptr), "instnorm fwd");
    float* wl[6] = {P, P, P, P, P, P};
    expect(cgs_mlp2d_d_step(wl, wl, 6, 64, P, 1000, P, 1000, 8e-3f, nullptr, nullptr, P, P, cgs_mlp2d_train_ws_bytes(2000, 6), nullptr), "mlp d step");
    expect(cgs_mlp2d_d_step(wl, wl, 7, 64, P, 10, P, 10, 0.f, nullptr, nullptr, nullptr, P, 1 << 20, nullptr), "mlp d step (too many layers)");
    // argument errors must come back as codes, not crashes
    expect(cgs_conv2d_nhwc_fwd(nullptr, P, P, P, 1, 8, 8, 32, 64, 5, 5, 2, 2, 0, nullptr, nullptr, P, WS, 0, nullptr), "null x");
    expect(cgs_conv2d_nhwc_fwd(P, P, P, P, 0, 8, 8, 32, 64, 5, 5, 2, 2, 0, nullptr, nullptr, P, WS, 0, nullptr), "B = 0");
    expect(cgs_conv2d_nhwc_fwd(P, P, P, P, 1, 8, 8, 32, 64, 5, 5, 2, 2, 99, nullptr, nullptr, P, WS, 0, nullptr), "bad epilogue");
    expect(cgs_conv2d_nhwc_fwd(P, P, P, P, 1, 8, 8, 32, 64, 5, 5, 2, 2, 0, nullptr, nullptr, P, 16, 0, nullptr), "tiny workspace");
    printf(bad ? "ASAN_HOST_FAIL %d\n" : "ASAN_HOST_OK\n", bad);
    return bad ? 1 : 0;
}
