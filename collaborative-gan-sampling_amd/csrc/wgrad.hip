// Weight gradients and the optimizer step of the discriminator "shaping" update: the caller on the other side of
// the refinement path (nsgan/GAN.py:270-272: refine a batch probabilistically, then ONE Adam step of D on
// BCE(D(real),1) + BCE(D(refined),0)).  Not on the refinement hot path itself (frozen weights there).
//
//   dW[tap][cb][cs] = sum_{b,r,c} big[b, r*s+ky-pt, c*s+kx-pl, cb] * small[b, r, c, cs]        (conv: big = x, small = dy)
//
// as a GEMM on v_mfma_f32_32x32x2_f32: rows i = (tap, cb), columns j = cs, reduction over the M = B*Hs*Ws small
// pixels.  Tile 128 x 128, 32 pixels per step; both operands are staged pixel-major in LDS ([m][i], [m][j]) so the
// fragment reads are stride-1 ds_read_b32.  The reduction is split over blockIdx.z into partial slabs that a second
// kernel adds in a fixed order (deterministic), optionally accumulating into dW (two D passes: real + refined).
#include "cgs_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const float* big;     // [B,Hb,Wb,Cb]
    const float* small;   // [B,Hs,Ws,Cs]
    float* slab;          // [splits][Kc][Csp]
    int B, Hb, Wb, Cb, Hs, Ws, Cs, Csp;
    int kh, kw, sh, sw, pt, pl;
    int Kc;               // kh*kw*Cb
    int splits, m_per_split;
};

static constexpr int WT = 128;       // tile rows (i) and columns (j)
static constexpr int WK = 32;        // pixels per step
static constexpr int WLD = WT + 4;   // LDS row pitch (floats)

template <bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][WK][WLD]  big, gathered:  As[m][i]
    float* Bs = smem + 2 * WK * WLD;       // [2][WK][WLD]  small:          Bs[m][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, j = lane & 31;
    const int i0 = blockIdx.x * WT, j0 = blockIdx.y * WT;
    const int M = p.B * p.Hs * p.Ws;
    const int mbeg = blockIdx.z * p.m_per_split;
    const int mend = mbeg + p.m_per_split < M ? mbeg + p.m_per_split : M;

    // staging role: thread -> float4 column q of pixel rows mr + 8*u
    const int q = tid & 31, mr = tid >> 5;
    // A: this thread's 4 consecutive i's are (tap, cb..cb+3) -- fixed for the block
    int a_dy[4], a_dx[4], a_cb[4];
    bool a_in[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = i0 + 4 * q + e;
        a_in[e] = i < p.Kc;
        const int tap = a_in[e] ? i / p.Cb : 0;
        a_cb[e] = i - tap * p.Cb;
        a_dy[e] = tap / p.kw - p.pt;
        a_dx[e] = tap % p.kw - p.pl;
    }
    const __amdgpu_buffer_rsrc_t big_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.big, 0, (int)((unsigned)p.B * (unsigned)p.Hb * (unsigned)p.Wb * (unsigned)p.Cb * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t small_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.small, 0, (int)((unsigned)M * (unsigned)p.Cs * 4u), 0x00020000);

    f32x4 ra[4], rb[4];
#define WLOAD(ms_)                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
        const int m = (ms_) + mr + 8 * u;                                                                      \
        const bool mv = m < mend;                                                                              \
        const int b = m / (p.Hs * p.Ws), rem = m - b * (p.Hs * p.Ws);                                          \
        const int r = rem / p.Ws, c = rem - r * p.Ws;                                                          \
        if (VEC) {                                                                                             \
            const int iy = r * p.sh + a_dy[0], ix = c * p.sw + a_dx[0];                                        \
            const bool ok = mv && a_in[0] && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb;  \
            const unsigned off = ok ? (unsigned)(((b * p.Hb + iy) * p.Wb + ix) * p.Cb + a_cb[0]) * 4u : 0xFFFFFFF0u; \
            ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(big_rsrc, off, 0, 0));     \
        } else {                                                                                               \
            float v[4];                                                                                        \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                const int iy = r * p.sh + a_dy[e], ix = c * p.sw + a_dx[e];                                    \
                const bool ok = mv && a_in[e] && (unsigned)iy < (unsigned)p.Hb && (unsigned)ix < (unsigned)p.Wb; \
                const unsigned off = ok ? (unsigned)(((b * p.Hb + iy) * p.Wb + ix) * p.Cb + a_cb[e]) * 4u : 0xFFFFFFF0u; \
                v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(big_rsrc, off, 0, 0));   \
            }                                                                                                  \
            ra[u] = f32x4{v[0], v[1], v[2], v[3]};                                                             \
        }                                                                                                      \
        {                                                                                                      \
            const int jj = j0 + 4 * q;                                                                         \
            if ((p.Cs & 3) == 0) {                                                                             \
                const unsigned off = (mv && jj < p.Cs) ? (unsigned)(m * p.Cs + jj) * 4u : 0xFFFFFFF0u;         \
                rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(small_rsrc, off, 0, 0)); \
            } else {                                                                                           \
                float v[4];                                                                                    \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
                    const unsigned off = (mv && jj + e < p.Cs) ? (unsigned)(m * p.Cs + jj + e) * 4u : 0xFFFFFFF0u; \
                    v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(small_rsrc, off, 0, 0)); \
                }                                                                                              \
                rb[u] = f32x4{v[0], v[1], v[2], v[3]};                                                         \
            }                                                                                                  \
        }                                                                                                      \
    }
#define WSTORE(buf_)                                                                                            \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
        *(f32x4*)(As + ((buf_) * WK + mr + 8 * u) * WLD + 4 * q) = ra[u];                                      \
        *(f32x4*)(Bs + ((buf_) * WK + mr + 8 * u) * WLD + 4 * q) = rb[u];                                      \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (mbeg < mend) {
        WLOAD(mbeg);
        WSTORE(0);
    }
    __syncthreads();
    int buf = 0;
    for (int ms = mbeg; ms < mend; ms += WK, buf ^= 1) {
        const bool more = ms + WK < mend;
        if (more) { WLOAD(ms + WK); }
        const float* a = As + buf * WK * WLD + wm * 64 + j;
        const float* b = Bs + buf * WK * WLD + wn * 64 + j;
#pragma unroll
        for (int ks = 0; ks < WK / 2; ++ks) {
            const int m = 2 * ks + h;
            const float a0 = a[m * WLD], a1 = a[m * WLD + 32];
            const float b0 = b[m * WLD], b1 = b[m * WLD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) { WSTORE(buf ^ 1); }
        __syncthreads();
    }
#undef WLOAD
#undef WSTORE
    // partial tile -> slab[z][i][j]
    float* slab = p.slab + (size_t)blockIdx.z * p.Kc * p.Csp;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int jj = j0 + wn * 64 + tn * 32 + j;
        if (jj >= p.Csp) continue;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i < p.Kc) slab[(size_t)i * p.Csp + jj] = acc[tm][tn][r];
            }
    }
}

// dW[i][j] (+)= sum_z slab[z][i][j], fixed order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Kc,
                                                           int Cs, int Csp, int splits, int accumulate) {
    const long total = (long)Kc * Cs;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t / Cs), jj = (int)(t - (long)i * Cs);
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += slab[((size_t)z * Kc + i) * Csp + jj];
        dw[t] = accumulate ? dw[t] + s : s;
    }
}

static void wgrad_geom(WgradParams& p, int B, int Hb, int Wb, int Cb, int Hs, int Ws, int Cs, int kh, int kw, int sh, int sw) {
    p.B = B; p.Hb = Hb; p.Wb = Wb; p.Cb = Cb; p.Hs = Hs; p.Ws = Ws; p.Cs = Cs; p.Csp = cgs_round_up(Cs, 4);
    p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw;
    p.pt = cgs_same_pad_before(Hb, kh, sh); p.pl = cgs_same_pad_before(Wb, kw, sw);
    p.Kc = kh * kw * Cb;
    const long M = (long)B * Hs * Ws;
    const long tiles = (long)cgs_ceil_div(p.Kc, WT) * cgs_ceil_div(Cs, WT);
    long splits = (1024 + tiles - 1) / tiles;                   // aim at >= 2 rounds of blocks over 512 slots
    const long max_splits = (M + 4 * WK - 1) / (4 * WK);         // at least 4 steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits > 256) splits = 256;
    if (splits < 1) splits = 1;
    long mps = (M + splits - 1) / splits;
    mps = (mps + WK - 1) / WK * WK;
    p.m_per_split = (int)mps;
    p.splits = (int)((M + mps - 1) / mps);
}

static size_t wgrad_ws(const WgradParams& p) { return (size_t)p.splits * p.Kc * p.Csp * sizeof(float); }

static int wgrad_run(WgradParams& p, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s, const char* who) {
    if ((long)p.B * p.Hb * p.Wb * p.Cb * 4 > 0x7fffffffL || (long)p.B * p.Hs * p.Ws * p.Cs * 4 > 0x7fffffffL)
        return cgs_set_error(CGS_EINVAL, "%s: a tensor exceeds 2 GiB", who);
    const size_t need = wgrad_ws(p);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    p.slab = (float*)ws;
    constexpr size_t smem = (size_t)4 * WK * WLD * sizeof(float);
    CGS_SMEM_ATTR(smem, who, wgrad_kernel<true>);
    CGS_SMEM_ATTR(smem, who, wgrad_kernel<false>);
    const dim3 grid(cgs_ceil_div(p.Kc, WT), cgs_ceil_div(p.Cs, WT), p.splits);
    // VEC: a thread's 4 consecutive rows (tap, cb..cb+3) never straddle a tap when Cb % 4 == 0 -> one 16-byte load
    if ((p.Cb & 3) == 0)
        hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), smem, s, p);
    else
        hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), smem, s, p);
    CGS_CHECK_LAUNCH(who);
    const long total = (long)p.Kc * p.Cs;
    const unsigned rb = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, s, p.slab, dw, p.Kc, p.Cs, p.Csp, p.splits, accumulate);
    CGS_CHECK_LAUNCH(who);
    return CGS_OK;
}

// d/dlogit of scale * sum BCE(logit, target): scale * (sigmoid(l) - target); also the loss terms' sum (optional)
__global__ void bce_grad_kernel(const float* __restrict__ l, float target, float scale, float* __restrict__ dl,
                                float* __restrict__ loss_sum, int n) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v = l[i];
        const float sg = v >= 0.f ? 1.f / (1.f + expf(-v)) : expf(v) / (1.f + expf(v));
        dl[i] = scale * (sg - target);
        // max(v,0) - v*target + log1p(exp(-|v|))
        acc += fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v)));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (loss_sum && threadIdx.x == 0) loss_sum[blockIdx.x] = red[0] * scale;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float lr_t, float b1, float b2, float eps, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// gamma / beta gradients from the statistics cgs_bn_train_lrelu_bwd_data left in its workspace
__global__ void bn_param_grad_kernel(const float* __restrict__ stat2, float Mf, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float db = stat2[c] * Mf, dg = stat2[C + c] * Mf;
    dbeta[c] = accumulate ? dbeta[c] + db : db;
    dgamma[c] = accumulate ? dgamma[c] + dg : dg;
}

extern "C" {

size_t cgs_conv_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw, int sh, int sw) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0) return 0;
    WgradParams p;
    wgrad_geom(p, B, H, W, Cin, cgs_ceil_div(H, sh), cgs_ceil_div(W, sw), Cout, kh, kw, sh, sw);
    return wgrad_ws(p);
}

int cgs_conv2d_nhwc_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int kh,
                               int kw, int sh, int sw, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || !x || !dy || !dw)
        return cgs_set_error(CGS_EINVAL, "conv2d_nhwc_bwd_weight: bad argument");
    WgradParams p;
    wgrad_geom(p, B, H, W, Cin, cgs_ceil_div(H, sh), cgs_ceil_div(W, sw), Cout, kh, kw, sh, sw);
    p.big = x; p.small = dy;
    return wgrad_run(p, dw, accumulate, ws, ws_bytes, (hipStream_t)stream, "conv2d_nhwc_bwd_weight");
}

int cgs_linear_bwd_weight(const float* x, const float* dy, float* dw, int B, int in, int out, int accumulate, void* ws,
                          size_t ws_bytes, void* stream) {
    if (B <= 0 || in <= 0 || out <= 0 || !x || !dy || !dw) return cgs_set_error(CGS_EINVAL, "linear_bwd_weight: bad argument");
    WgradParams p;
    wgrad_geom(p, B, 1, 1, in, 1, 1, out, 1, 1, 1, 1);
    p.big = x; p.small = dy;
    return wgrad_run(p, dw, accumulate, ws, ws_bytes, (hipStream_t)stream, "linear_bwd_weight");
}

int cgs_bce_logits_grad(const float* logits, float target, float scale, float* dlogits, float* loss_sum, int n, void* stream) {
    if (n <= 0) return cgs_set_error(CGS_EINVAL, "bce_logits_grad: n=%d", n);
    hipLaunchKernelGGL(bce_grad_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, scale, dlogits, loss_sum, n);
    CGS_CHECK_LAUNCH("bce_logits_grad");
    return CGS_OK;
}

int cgs_adam_step(float* w, const float* g, float* m, float* v, float lr_t, float beta1, float beta2, float eps, size_t n,
                  void* stream) {
    if (n == 0) return CGS_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, g, m, v, lr_t, beta1, beta2, eps, n);
    CGS_CHECK_LAUNCH("adam_step");
    return CGS_OK;
}

int cgs_bn_train_param_grads(const void* bwd_ws, int M, int C, float* dgamma, float* dbeta, int accumulate, void* stream) {
    if (M <= 0 || C <= 0 || !bwd_ws) return cgs_set_error(CGS_EINVAL, "bn_train_param_grads: bad argument");
    // workspace layout of bn.hip: partials [CGS_BN_MAX_BLOCKS][2][C] | stat [4][C] | stat2 [2][C] = {mean(dy'), mean(dy'*xhat)}
    const float* stat2 = (const float*)bwd_ws + (size_t)CGS_BN_MAX_BLOCKS * 2 * C + 4 * (size_t)C;
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cgs_ceil_div(C, 128)), dim3(128), 0, (hipStream_t)stream, stat2, (float)M, dgamma, dbeta, C, accumulate);
    CGS_CHECK_LAUNCH("bn_train_param_grads");
    return CGS_OK;
}

}  // extern "C"
