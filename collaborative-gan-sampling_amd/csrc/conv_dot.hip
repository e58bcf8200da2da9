// Forward convolution to <= 4 output channels over a DEEP reduction (K = kh * kw * Cin >= 1024), e.g. the PatchGAN logit head
// 4x4 x 512 -> 1 of BASELINE config 5 (the reference's generic hook for it: the per-sample mean over a logit MAP,
// sampling/collaborator.py:34-37).  On the implicit GEMM one output channel is padded to a 64-column tile: 1.6 % of the issued
// MFMA work is useful (81 us for 0.13 GFLOP at batch 8).  It is a dot product per output pixel: one wave per pixel, the K
// elements spread over the lanes as float4 of consecutive channels (coalesced 1 KB rows per tap), the weights read unpacked
// (w[kh][kw][Cin][N]: for N = 1 the same float4 pattern, L2-resident), a fixed xor-shuffle tree at the end (deterministic), bias
// and the forward epilogues.  The input is re-read once per tap from L1 / L2 (neighbouring pixels are neighbouring waves).
#include <stdio.h>

#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DotParams {
    const float* in;     // [B,Hin,Win,C]
    const float* w;      // [kh][kw][C][N]
    const float* bias;   // [N] or null
    const float* ep_a;   // [N] for AFFINE_RELU
    const float* ep_b;
    float* out;          // [B,Hout,Wout,N]
    int B, Hin, Win, C, Hout, Wout;
    int kh, kw, S, pt, pl;
    int epilogue;
};

template <int N>
__global__ __launch_bounds__(256) void conv_dot_kernel(DotParams p, long pixels) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= pixels) return;
    const int RC = p.Hout * p.Wout;
    const int b = (int)(m / RC), rem = (int)(m - (long)b * RC);
    const int r = rem / p.Wout, c = rem - r * p.Wout;
    const int cq = p.C >> 2;                                 // float4 per pixel
    const bool wal = (((uintptr_t)p.w) & 15) == 0;            // (the float4 weight path needs an aligned weight tensor; else element-wise)
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
        const int iy = r * p.S + ky - p.pt;
        if ((unsigned)iy >= (unsigned)p.Hin) continue;       // (wave-uniform: a padding tap adds nothing)
        for (int kx = 0; kx < p.kw; ++kx) {
            const int ix = c * p.S + kx - p.pl;
            if ((unsigned)ix >= (unsigned)p.Win) continue;
            const f32x4* src = (const f32x4*)(p.in + ((size_t)(b * p.Hin + iy) * p.Win + ix) * p.C);
            const float* wt = p.w + (size_t)(ky * p.kw + kx) * p.C * N;
            for (int q = lane; q < cq; q += 64) {
                const f32x4 x = src[q];
                if (N == 1 && wal) {
                    const f32x4 ww = *(const f32x4*)(wt + 4 * q);
                    acc[0] = fmaf(x.x, ww.x, acc[0]); acc[0] = fmaf(x.y, ww.y, acc[0]);
                    acc[0] = fmaf(x.z, ww.z, acc[0]); acc[0] = fmaf(x.w, ww.w, acc[0]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int n = 0; n < N; ++n) acc[n] = fmaf(x[e], wt[(size_t)(4 * q + e) * N + n], acc[n]);
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[n] += __shfl_xor(acc[n], off, 64);
    if (lane == 0) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float a = p.epilogue == CGS_EPI_AFFINE_RELU ? p.ep_a[n] : 1.f, sh = p.epilogue == CGS_EPI_AFFINE_RELU ? p.ep_b[n] : 0.f;
            p.out[(size_t)m * N + n] = epilogue_apply(acc[n] + (p.bias ? p.bias[n] : 0.f), p.epilogue, a, sh, 0.f);
        }
    }
}

// forward direction, <= 4 output channels, whole float4 of input channels, a reduction deep enough that a wave per pixel pays
int cgs_conv_dot_ok(const CgsLayer& L, int epilogue) {
    return L.Cs <= 4 && (L.Cb % 4) == 0 && (long)L.kh * L.kw * L.Cb >= 1024 && epilogue < CGS_EPI_RELU_BWD_AFFINE
        && L.sh == L.sw;                                              // (the kernel carries one stride for both axes)
}

int cgs_conv_dot_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out, int epilogue,
                        const float* ep_a, const float* ep_b, hipStream_t s) {
    if ((uintptr_t)in & 15) return cgs_set_error(CGS_EINVAL, "conv_dot: input must be 16-byte aligned");
    DotParams p;
    p.in = in; p.w = w; p.bias = bias; p.ep_a = ep_a; p.ep_b = ep_b; p.out = out;
    p.B = B; p.Hin = L.Hb; p.Win = L.Wb; p.C = L.Cb; p.Hout = L.Hs; p.Wout = L.Ws;
    p.kh = L.kh; p.kw = L.kw; p.S = L.sh;
    p.pt = cgs_same_pad_before(L.Hb, L.kh, L.sh); p.pl = cgs_same_pad_before(L.Wb, L.kw, L.sw);
    p.epilogue = epilogue;
    const long pixels = (long)B * L.Hs * L.Ws;
    if (pixels == 0) return CGS_OK;
    const long blocks = (pixels + 3) / 4;
    if (blocks > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_dot: grid too large");
    switch (L.Cs) {
        case 1: hipLaunchKernelGGL((conv_dot_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, p, pixels); break;
        case 2: hipLaunchKernelGGL((conv_dot_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, p, pixels); break;
        case 3: hipLaunchKernelGGL((conv_dot_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, s, p, pixels); break;
        default: hipLaunchKernelGGL((conv_dot_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, p, pixels); break;
    }
    CGS_CHECK_LAUNCH("conv_dot");
    static thread_local char name[32];
    snprintf(name, sizeof(name), "conv_dot_kernel<%d>", L.Cs);
    cgs_note_kernel(name);
    return CGS_OK;
}
