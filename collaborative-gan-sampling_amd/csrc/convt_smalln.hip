// Transposed stride-2 convolution with a tiny channel count on the OUTPUT side (N <= 4):
// the last generator deconv (64 -> 3 / 64 -> 1 channels, nsgan/GAN.py:99) and the backward-data
// of the first discriminator conv (nsgan/GAN.py:64 through sampling/collaborator.py:31).
//
// N = 3 would waste >90 % of a 32-wide MFMA tile, and the f32 matrix rate equals the f32 vector
// rate on gfx950, so this runs on the VALU: one thread owns one 2x2 output quad (all four
// parity classes of one input-resolution pixel), walks the <=3x3 input neighbourhood once
// (each input float4 feeds every class whose tap hits it), with the weights as wave-uniform
// scalar loads.  Output rows of a quad are 2*N contiguous floats, adjacent lanes adjacent quads.
#include "cgs_internal.h"

struct SmallNParams {
    const float* in;     // [B,Hs,Ws,Cs]
    const float* w;      // [kh][kw][N][Cs]
    const float* bias;   // [N] or null
    float* out;          // [B,2Hs,2Ws,N]
    int B, Hs, Ws, Cs, kh, kw, pt, pl, dmin_y, dmax_y, dmin_x, dmax_x, epilogue;
};

template <int N>
__global__ __launch_bounds__(256) void convt_smalln_kernel(SmallNParams p) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)p.B * p.Hs * p.Ws;
    if (q >= total) return;
    const int c = (int)(q % p.Ws);
    const int r = (int)((q / p.Ws) % p.Hs);
    const int b = (int)(q / ((long)p.Ws * p.Hs));

    float acc[2][2][N];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int n = 0; n < N; ++n) acc[py][px][n] = 0.f;

    for (int dy = p.dmin_y; dy <= p.dmax_y; ++dy) {
        const int iy = r + dy;
        for (int dx = p.dmin_x; dx <= p.dmax_x; ++dx) {
            const int ix = c + dx;
            const bool ok = (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
            const float* src = p.in + ((size_t)(b * p.Hs + (ok ? iy : 0)) * p.Ws + (ok ? ix : 0)) * p.Cs;
            // taps (wave-uniform): ky = py + pt - 2*dy, kx = px + pl - 2*dx
            int ky[2], kx[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                ky[s] = s + p.pt - 2 * dy;
                kx[s] = s + p.pl - 2 * dx;
            }
            for (int ci = 0; ci < p.Cs; ci += 4) {
                float4 v = ok ? *(const float4*)(src + ci) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    if ((unsigned)ky[py] >= (unsigned)p.kh) continue;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        if ((unsigned)kx[px] >= (unsigned)p.kw) continue;
                        const float* wt = p.w + ((size_t)(ky[py] * p.kw + kx[px]) * N) * p.Cs + ci;
#pragma unroll
                        for (int n = 0; n < N; ++n) {
                            const float4 wv = *(const float4*)(wt + (size_t)n * p.Cs);   // uniform -> s_load
                            acc[py][px][n] = fmaf(v.x, wv.x, acc[py][px][n]);
                            acc[py][px][n] = fmaf(v.y, wv.y, acc[py][px][n]);
                            acc[py][px][n] = fmaf(v.z, wv.z, acc[py][px][n]);
                            acc[py][px][n] = fmaf(v.w, wv.w, acc[py][px][n]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        float* o = p.out + ((size_t)(b * 2 * p.Hs + 2 * r + py) * (2 * p.Ws) + 2 * c) * N;
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float v = acc[py][px][n] + (p.bias ? p.bias[n] : 0.f);
                if (p.epilogue == CGS_EPI_TANH) v = tanhf(v);
                else if (p.epilogue == CGS_EPI_LRELU) v = fmaxf(v, 0.2f * v);
                o[px * N + n] = v;
            }
    }
}

int cgs_convt_smalln_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias,
                            float* out, int epilogue, hipStream_t s) {
    SmallNParams p;
    p.in = in; p.w = w; p.bias = bias; p.out = out;
    p.B = B; p.Hs = L.Hs; p.Ws = L.Ws; p.Cs = L.Cs; p.kh = L.kh; p.kw = L.kw;
    p.pt = cgs_same_pad_before(L.Hb, L.kh, 2); p.pl = cgs_same_pad_before(L.Wb, L.kw, 2);
    p.epilogue = epilogue;
    // neighbourhood: dy = (py + pt - ky)/2 over the (py, ky) pairs of matching parity
    auto range = [](int k, int pad, int& lo, int& hi) {
        lo = 1 << 20; hi = -(1 << 20);
        for (int par = 0; par < 2; ++par)
            for (int kk = 0; kk < k; ++kk)
                if (((par + pad - kk) & 1) == 0) {
                    const int d = (par + pad - kk) / 2;   // exact
                    if (d < lo) lo = d;
                    if (d > hi) hi = d;
                }
    };
    range(L.kh, p.pt, p.dmin_y, p.dmax_y);
    range(L.kw, p.pl, p.dmin_x, p.dmax_x);
    const long total = (long)B * L.Hs * L.Ws;
    if (total == 0) return CGS_OK;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    switch (L.Cb) {
        case 1: hipLaunchKernelGGL(convt_smalln_kernel<1>, dim3(blocks), dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL(convt_smalln_kernel<2>, dim3(blocks), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(convt_smalln_kernel<3>, dim3(blocks), dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL(convt_smalln_kernel<4>, dim3(blocks), dim3(256), 0, s, p); break;
        default: return cgs_set_error(CGS_EINVAL, "convt_smalln: N=%d unsupported", L.Cb);
    }
    CGS_CHECK_LAUNCH("convt_smalln");
    cgs_note_kernel("convt_smalln_kernel");
    return CGS_OK;
}
