// Element-wise and per-sample kernels of the refinement loop: activations and their input
// gradients (nsgan/ops.py:69-70, nsgan/GAN.py:96-100), the loss seed and per-sample mean logit
// (nsgan/GAN.py:176-177, sampling/collaborator.py:31-37), the momentum update
// (sampling/policy.py:27-37, sampling/collaborator.py:66-70) and the best-sample selection
// (sampling/collaborator.py:76-83).  All HBM-bound: 16-byte accesses, grid-stride.
#include "cgs_internal.h"

static unsigned ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b == 0) b = 1;
    return (unsigned)b;
}

#define GRID_STRIDE(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

__global__ void bn_fold_kernel(const float* g, const float* b, const float* mm, const float* mv, float eps, float* a,
                               float* bo, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = g[c] / sqrtf(mv[c] + eps);
    a[c] = s; bo[c] = b[c] - s * mm[c];
}

int cgs_bn_fold(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, float* a, float* b,
                int C, void* stream) {
    if (C <= 0) return cgs_set_error(CGS_EINVAL, "bn_fold: C=%d", C);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cgs_ceil_div(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, mm, mv, eps, a, b, C);
    CGS_CHECK_LAUNCH("bn_fold");
    return CGS_OK;
}

// MODE 0: y = relu(a*x+b).  MODE 1: dx = dy*(y>0)*a
template <int MODE>
__global__ __launch_bounds__(256) void affine_relu_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                          const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ o, size_t n, int C) {
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        if (MODE == 0) o[i] = fmaxf(fmaf(a[c], p0[i], b[c]), 0.f);
        else o[i] = p1[i] > 0.f ? p0[i] * a[c] : 0.f;
    }
}

int cgs_affine_relu_fwd(const float* x, const float* a, const float* b, float* y, int M, int C, void* stream) {
    if (M < 0 || C <= 0) return cgs_set_error(CGS_EINVAL, "affine_relu_fwd: M=%d C=%d", M, C);
    const size_t n = (size_t)M * C;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(affine_relu_kernel<0>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, nullptr, a, b, y, n, C);
    CGS_CHECK_LAUNCH("affine_relu_fwd");
    return CGS_OK;
}

int cgs_affine_relu_bwd(const float* dy, const float* y, const float* a, float* dx, int M, int C, void* stream) {
    if (M < 0 || C <= 0) return cgs_set_error(CGS_EINVAL, "affine_relu_bwd: M=%d C=%d", M, C);
    const size_t n = (size_t)M * C;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(affine_relu_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, a, nullptr, dx, n, C);
    CGS_CHECK_LAUNCH("affine_relu_bwd");
    return CGS_OK;
}

// plain per-channel affine (inference-mode bn without an activation) and its input gradient
template <int MODE>
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ p0, const float* __restrict__ a,
                                                     const float* __restrict__ b, float* __restrict__ o, size_t n, int C) {
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        o[i] = MODE == 0 ? fmaf(a[c], p0[i], b[c]) : p0[i] * a[c];
    }
}

int cgs_affine_fwd(const float* x, const float* a, const float* b, float* y, int M, int C, void* stream) {
    if (M < 0 || C <= 0) return cgs_set_error(CGS_EINVAL, "affine_fwd: M=%d C=%d", M, C);
    const size_t n = (size_t)M * C;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(affine_kernel<0>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, a, b, y, n, C);
    CGS_CHECK_LAUNCH("affine_fwd");
    return CGS_OK;
}

int cgs_affine_bwd(const float* dy, const float* a, float* dx, int M, int C, void* stream) {
    if (M < 0 || C <= 0) return cgs_set_error(CGS_EINVAL, "affine_bwd: M=%d C=%d", M, C);
    const size_t n = (size_t)M * C;
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(affine_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, a, nullptr, dx, n, C);
    CGS_CHECK_LAUNCH("affine_bwd");
    return CGS_OK;
}

// OP 0 lrelu fwd, 1 lrelu bwd, 2 tanh fwd, 3 tanh bwd
template <int OP>
__global__ __launch_bounds__(256) void unary_kernel(const float* __restrict__ p0, const float* __restrict__ p1, float leak,
                                                    float* __restrict__ o, size_t n) {
    GRID_STRIDE(i, n) {
        float r;
        if (OP == 0) { const float v = p0[i]; r = fmaxf(v, leak * v); }
        else if (OP == 1) r = p0[i] * (p1[i] > 0.f ? 1.f : leak);
        else if (OP == 2) r = tanhf(p0[i]);
        else { const float y = p1[i]; r = p0[i] * (1.f - y * y); }
        o[i] = r;
    }
}

#define UNARY_ENTRY(fn, OP, a0, a1, lk)                                                                  \
    if (n == 0) return CGS_OK;                                                                           \
    hipLaunchKernelGGL(unary_kernel<OP>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a0, a1, lk, out, n); \
    CGS_CHECK_LAUNCH(fn);                                                                                \
    return CGS_OK;

int cgs_lrelu_fwd(const float* x, float leak, float* out, size_t n, void* stream) { UNARY_ENTRY("lrelu_fwd", 0, x, nullptr, leak) }
int cgs_lrelu_bwd(const float* dy, const float* y, float leak, float* out, size_t n, void* stream) { UNARY_ENTRY("lrelu_bwd", 1, dy, y, leak) }
int cgs_tanh_fwd(const float* x, float* out, size_t n, void* stream) { UNARY_ENTRY("tanh_fwd", 2, x, nullptr, 0.f) }
int cgs_tanh_bwd(const float* dy, const float* y, float* out, size_t n, void* stream) { UNARY_ENTRY("tanh_bwd", 3, dy, y, 0.f) }

__global__ void bce_rowmean_kernel(const float* __restrict__ l, float* __restrict__ dl, float* __restrict__ lm, int B, int P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < P; ++i) {
        const float v = l[(size_t)b * P + i];
        s += v;
        // sigmoid(v) - 1 = -1/(1+exp(v)), written to stay accurate for large |v|
        dl[(size_t)b * P + i] = v >= 0.f ? -expf(-v) / (1.f + expf(-v)) : -1.f / (1.f + expf(v));
    }
    lm[b] = s / (float)P;
}

// PatchGAN logit maps (P = hundreds of logits per sample): one wave per sample, lanes stride over the row, fixed shuffle tree
__global__ __launch_bounds__(256) void bce_rowmean_wave_kernel(const float* __restrict__ l, float* __restrict__ dl, float* __restrict__ lm,
                                                               int B, int P) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float s = 0.f;
    for (int i = lane; i < P; i += 64) {
        const float v = l[(size_t)b * P + i];
        s += v;
        dl[(size_t)b * P + i] = v >= 0.f ? -expf(-v) / (1.f + expf(-v)) : -1.f / (1.f + expf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) lm[b] = s / (float)P;
}

int cgs_bce_ones_grad_rowmean(const float* logits, float* dlogits, float* logit_mean, int B, int P, void* stream) {
    if (B <= 0 || P <= 0) return cgs_set_error(CGS_EINVAL, "bce_ones_grad_rowmean: B=%d P=%d", B, P);
    if (P >= 64)
        hipLaunchKernelGGL(bce_rowmean_wave_kernel, dim3(cgs_ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, dlogits, logit_mean, B, P);
    else
        hipLaunchKernelGGL(bce_rowmean_kernel, dim3(cgs_ceil_div(B, 128)), dim3(128), 0, (hipStream_t)stream, logits, dlogits, logit_mean, B, P);
    CGS_CHECK_LAUNCH("bce_ones_grad_rowmean");
    return CGS_OK;
}

// sig[b] = mean_p sigmoid(l[b, p]): fake_sigmoids = tf.nn.sigmoid(fake_logits) (nsgan/GAN.py:154-155; P = 1 there), the score the
// accept / reject step reads (nsgan/GAN.py:409, sampling/idpsampler.py:18).  One thread per sample (P < 64) or one wave (P >= 64).
__device__ __forceinline__ float sigmoid_stable(float v) { return v >= 0.f ? 1.f / (1.f + expf(-v)) : expf(v) / (1.f + expf(v)); }

__global__ void sigmoid_rowmean_kernel(const float* __restrict__ l, float* __restrict__ sig, int B, int P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < P; ++i) s += sigmoid_stable(l[(size_t)b * P + i]);
    sig[b] = P == 1 ? s : s / (float)P;
}

__global__ __launch_bounds__(256) void sigmoid_rowmean_wave_kernel(const float* __restrict__ l, float* __restrict__ sig, int B, int P) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float s = 0.f;
    for (int i = lane; i < P; i += 64) s += sigmoid_stable(l[(size_t)b * P + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) sig[b] = s / (float)P;
}

int cgs_sigmoid_rowmean(const float* logits, float* sigmoids, int B, int P, void* stream) {
    if (B <= 0 || P <= 0) return cgs_set_error(CGS_EINVAL, "sigmoid_rowmean: B=%d P=%d", B, P);
    if (P >= 64)
        hipLaunchKernelGGL(sigmoid_rowmean_wave_kernel, dim3(cgs_ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, sigmoids, B, P);
    else
        hipLaunchKernelGGL(sigmoid_rowmean_kernel, dim3(cgs_ceil_div(B, 128)), dim3(128), 0, (hipStream_t)stream, logits, sigmoids, B, P);
    CGS_CHECK_LAUNCH("sigmoid_rowmean");
    return CGS_OK;
}

// OP 0: softplus(-l) (stable form);  1: dy * (sigmoid(l) - 1);  2: clip
template <int OP>
__global__ __launch_bounds__(256) void loss_unary_kernel(const float* __restrict__ p0, const float* __restrict__ p1, float lo, float hi,
                                                         float* __restrict__ o, size_t n) {
    GRID_STRIDE(i, n) {
        float r;
        if (OP == 0) { const float v = p0[i]; r = fmaxf(-v, 0.f) + log1pf(expf(-fabsf(v))); }
        else if (OP == 1) { const float v = p1[i]; r = p0[i] * (v >= 0.f ? -expf(-v) / (1.f + expf(-v)) : -1.f / (1.f + expf(v))); }
        else r = fminf(fmaxf(p0[i], lo), hi);
        o[i] = r;
    }
}

int cgs_bce_ones_fwd(const float* logits, float* loss, size_t n, void* stream) {
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(loss_unary_kernel<0>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, logits, nullptr, 0.f, 0.f, loss, n);
    CGS_CHECK_LAUNCH("bce_ones_fwd");
    return CGS_OK;
}

int cgs_bce_ones_bwd(const float* dloss, const float* logits, float* dlogits, size_t n, void* stream) {
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(loss_unary_kernel<1>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dloss, logits, 0.f, 0.f, dlogits, n);
    CGS_CHECK_LAUNCH("bce_ones_bwd");
    return CGS_OK;
}

int cgs_clip(const float* x, float vmin, float vmax, float* y, size_t n, void* stream) {
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(loss_unary_kernel<2>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, nullptr, vmin, vmax, y, n);
    CGS_CHECK_LAUNCH("clip");
    return CGS_OK;
}

__global__ __launch_bounds__(256) void refine_update_kernel(float* __restrict__ theta, float* __restrict__ m,
                                                            const float* __restrict__ g, float rate, float alpha,
                                                            int first, int use_clip, float vmin, float vmax, size_t n) {
#pragma clang fp contract(off)   // separately rounded mul / add, the reference's op sequence (policy.py:33-36)
    GRID_STRIDE(i, n) {
        const float lg = rate * g[i];
        const float am = alpha * m[i];
        const float mv = first ? lg : am + lg;
        m[i] = mv;
        float t = theta[i] - mv;
        if (use_clip) t = fminf(fmaxf(t, vmin), vmax);          // collaborator.py:69-70
        theta[i] = t;
    }
}

int cgs_refine_update(float* theta, float* m, const float* g, float rate, float alpha, int first, int use_clip,
                      float vmin, float vmax, size_t n, void* stream) {
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(refine_update_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, theta, m, g, rate, alpha, first, use_clip, vmin, vmax, n);
    CGS_CHECK_LAUNCH("refine_update");
    return CGS_OK;
}

// one sample's row: 16-byte copies where the row length and both rows' addresses allow it (F % 4 == 0 and 16-byte aligned bases: every row then
// starts on a 16-byte boundary), single floats otherwise.  The scalar loop alone moved config 5's 33.5 MB of refined maps + 6.3 MB of images at 1.8 TB/s.
__device__ __forceinline__ void copy_row(const float* __restrict__ src, float* __restrict__ dst, int F, bool vec4) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (vec4) {
        const float4* s4 = (const float4*)src;
        float4* d4 = (float4*)dst;
        for (int i = t; i < (F >> 2); i += nt) d4[i] = s4[i];
    } else {
        for (int i = t; i < F; i += nt) dst[i] = src[i];
    }
}

// pass 1: copy theta rows whose sample improves (reads best_logit, never writes it)
__global__ __launch_bounds__(256) void refine_select_copy_kernel(const float* __restrict__ theta, const float* __restrict__ logit,
                                                                 const int32_t* __restrict__ forced, int step,
                                                                 float* __restrict__ best_theta, const float* __restrict__ best_logit,
                                                                 int B, int F) {
    const int b = blockIdx.y;
    const bool upd = forced ? forced[b] == step : logit[b] > best_logit[b];
    if (!upd) return;
    copy_row(theta + (size_t)b * F, best_theta + (size_t)b * F, F, (F & 3) == 0 && (((uintptr_t)theta | (uintptr_t)best_theta) & 15) == 0);
}
// pass 2: per-sample scalars
__global__ void refine_select_scalar_kernel(const float* __restrict__ logit, const int32_t* __restrict__ forced, int step,
                                            float* __restrict__ best_logit, float* __restrict__ best_step, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool upd = forced ? forced[b] == step : logit[b] > best_logit[b];
    if (upd) { best_logit[b] = logit[b]; best_step[b] = (float)(step + 1); }
}

int cgs_refine_select_rows(const float* src, const float* logit, const int32_t* forced, int step_index, float* dst,
                           const float* best_logit, int B, int F, void* stream) {
    if (B <= 0 || F <= 0 || B > 65535) return cgs_set_error(CGS_EINVAL, "refine_select_rows: B=%d F=%d", B, F);
    int bx = cgs_ceil_div(F, 256 * 4);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(refine_select_copy_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, src, logit, forced, step_index, dst, best_logit, B, F);
    CGS_CHECK_LAUNCH("refine_select_rows");
    return CGS_OK;
}

// both row copies of one step in ONE launch: blockIdx.z = 0 copies the rendered image rows, 1 the refined-map rows (same predicate).
// With ``tickets`` (B ints, zero between launches) the per-sample scalars are updated in the same launch: every block of a selected
// sample takes a ticket after its copy, and the one that takes the last (all others have read best_logit[b] by then) writes
// best_logit / best_step and puts the ticket counter back to zero -- one dependent 4.5-us launch less per refinement step.
__global__ __launch_bounds__(256) void refine_select_copy2_kernel(const float* __restrict__ src0, float* __restrict__ dst0, int F0,
                                                                  const float* __restrict__ src1, float* __restrict__ dst1, int F1,
                                                                  const float* __restrict__ logit, const int32_t* __restrict__ forced, int step,
                                                                  float* best_logit, float* __restrict__ best_step, int* __restrict__ tickets) {
    const int b = blockIdx.y;
    const float lg = logit[b];
    const bool upd = forced ? forced[b] == step : lg > best_logit[b];
    if (!upd) return;                              // (nothing of this sample changes: no ticket needed)
    const int F = blockIdx.z ? F1 : F0;
    const float* src = blockIdx.z ? src1 : src0;
    float* dst = blockIdx.z ? dst1 : dst0;
    copy_row(src + (size_t)b * F, dst + (size_t)b * F, F, (F & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    if (!tickets) return;
    __syncthreads();                               // every thread of the block has read the predicate (it is the first thing a thread does)
    if (threadIdx.x == 0) {
        const int last = (int)(gridDim.x * gridDim.z) - 1;
        if (atomicAdd(tickets + b, 1) == last) {
            best_logit[b] = lg; best_step[b] = (float)(step + 1);
            tickets[b] = 0;
        }
    }
}

int cgs_refine_select2(const float* rows, float* best_rows, int Frows, const float* theta, float* best_theta, int F, const float* logit,
                       const int32_t* forced, int step_index, float* best_logit, float* best_step, int* tickets, int B, void* stream) {
    if (B <= 0 || F <= 0 || Frows <= 0 || B > 65535) return cgs_set_error(CGS_EINVAL, "refine_select2: B=%d F=%d Frows=%d", B, F, Frows);
    int bx = cgs_ceil_div(F > Frows ? F : Frows, 256 * 4);
    if (bx < 1) bx = 1;
    // (every block of a selected sample takes a ticket at ONE address: config 5's 64x64x256 maps made 2048 blocks per sample, and the serialised
    // atomics cost 145 us per step -- 5 % of its call; the copy loop strides over the grid, so 64 blocks per sample and tensor copy the same rows)
    if (tickets && bx > 64) bx = 64;
    hipLaunchKernelGGL(refine_select_copy2_kernel, dim3(bx, B, 2), dim3(256), 0, (hipStream_t)stream, rows, best_rows, Frows, theta, best_theta, F, logit, forced,
                       step_index, best_logit, best_step, tickets);
    if (!tickets)
        hipLaunchKernelGGL(refine_select_scalar_kernel, dim3(cgs_ceil_div(B, 128)), dim3(128), 0, (hipStream_t)stream, logit, forced, step_index, best_logit, best_step, B);
    CGS_CHECK_LAUNCH("refine_select2");
    return CGS_OK;
}

int cgs_refine_select(const float* theta, const float* logit, const int32_t* forced, int step_index, float* best_theta,
                      float* best_logit, float* best_step, int B, int F, void* stream) {
    if (B <= 0 || F <= 0 || B > 65535) return cgs_set_error(CGS_EINVAL, "refine_select: B=%d F=%d", B, F);
    int bx = cgs_ceil_div(F, 256 * 4);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(refine_select_copy_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, theta, logit, forced, step_index, best_theta, best_logit, B, F);
    hipLaunchKernelGGL(refine_select_scalar_kernel, dim3(cgs_ceil_div(B, 128)), dim3(128), 0, (hipStream_t)stream, logit, forced, step_index, best_logit, best_step, B);
    CGS_CHECK_LAUNCH("refine_select");
    return CGS_OK;
}


// out = a + b  (residual connection and the sum of its two gradient branches)
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, size_t n) {
    GRID_STRIDE(i, n) o[i] = a[i] + b[i];
}

int cgs_add(const float* a, const float* b, float* out, size_t n, void* stream) {
    if (n == 0) return CGS_OK;
    hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    CGS_CHECK_LAUNCH("add");
    return CGS_OK;
}
