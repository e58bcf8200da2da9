// Batch-statistics batch norm fused with LeakyReLU, forward and backward-data
// (tf.contrib.layers.batch_norm(is_training=True) + lrelu, nsgan/ops.py:19-26,69-70 as used by D at
// nsgan/GAN.py:65,67 inside the differentiated path, sampling/collaborator.py:31).
//
// HBM-bound: every pass streams [M,C] fp32 once with 16-byte accesses, channels across lanes.
// Per-channel reductions are two-stage and deterministic (no float atomics): stage 1 writes one
// (sum_a, sum_b) partial per block and channel, stage 2 adds the partials in a fixed order in
// double and finishes the statistics.  Run-to-run results are bit-identical.
#include <stdlib.h>

#include "cgs_internal.h"

#define BN_MAX_BLOCKS CGS_BN_MAX_BLOCKS

struct BnGeom {
    int G;               // stage-1 blocks
    int rows_per_block;
};

static BnGeom bn_geom(int M, int C) {
    // a block covers whole rows; aim at >= 8 rows per block pass and <= BN_MAX_BLOCKS blocks
    BnGeom g;
    int rpb = cgs_ceil_div(M, BN_MAX_BLOCKS);
    if (rpb < 16) rpb = 16;
    g.rows_per_block = rpb;
    g.G = cgs_ceil_div(M, rpb);
    (void)C;
    return g;
}

size_t cgs_bn_ws_bytes(int M, int C) {
    (void)M;
    return ((size_t)BN_MAX_BLOCKS * 2 * C + 6 * (size_t)C) * sizeof(float);   // partials | stat[4][C] | stat2[2][C]
}

// MODE 0: a = x, b = x*x.   MODE 1: a = dy', b = dy'*xhat  (dy' = dy * lrelu'(scale*x+shift))
// 16-byte loads: a thread owns 4 consecutive channels and strides over the block's rows; the row groups of a
// block are combined through LDS in a fixed order (deterministic).  C % 4 == 0.
// scale = gamma * invstd, shift = beta - mean * scale of 4 channels, formed where they are used (a separate kernel that wrote them
// to memory was one more dependent 4-us launch in front of every backward pass); the same expression in both backward kernels
__device__ __forceinline__ void bn_affine4(const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, int c, float4& mu, float4& inv, float4& sc, float4& sh) {
    mu = *(const float4*)(mean + c); inv = *(const float4*)(invstd + c);
    const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
    sc.x = g.x * inv.x; sc.y = g.y * inv.y; sc.z = g.z * inv.z; sc.w = g.w * inv.w;
    sh.x = fmaf(-mu.x, sc.x, b.x); sh.y = fmaf(-mu.y, sc.y, b.y); sh.z = fmaf(-mu.z, sc.z, b.z); sh.w = fmaf(-mu.w, sc.w, b.w);
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ mean_p, const float* __restrict__ invstd_p /* [groups][C] */,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float leak, float* __restrict__ part, int M, int C,
                                                         int rows_per_block) {
    __shared__ float4 red[2][256];
    // blockIdx.y = independent group (instance norm: one per sample; batch norm: a single group)
    x += (size_t)blockIdx.y * M * C;
    if (MODE == 1) { dy += (size_t)blockIdx.y * M * C; mean_p += (size_t)blockIdx.y * C; invstd_p += (size_t)blockIdx.y * C; }
    part += (size_t)blockIdx.y * gridDim.x * 2 * C;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const int CQ = C >> 2;
    // threads-per-row: largest power of two <= min(CQ, 256)
    int tpr = 1;
    while (tpr * 2 <= CQ && tpr * 2 <= 256) tpr *= 2;
    const int rg = tid / tpr, RG = 256 / tpr, tc = tid - rg * tpr;
    for (int q0 = 0; q0 < CQ; q0 += tpr) {
        const int cq = q0 + tc;
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
        if (cq < CQ) {
            float4 mean = sa, invstd = sa, scale = sa, shift = sa;
            if (MODE == 1) bn_affine4(mean_p, invstd_p, gamma, beta, 4 * cq, mean, invstd, scale, shift);
            for (int r = r0 + rg; r < r1; r += RG) {
                const float4 xv = ((const float4*)(x + (size_t)r * C))[cq];
                if (MODE == 0) {
                    sa.x += xv.x; sa.y += xv.y; sa.z += xv.z; sa.w += xv.w;
                    sb.x = fmaf(xv.x, xv.x, sb.x); sb.y = fmaf(xv.y, xv.y, sb.y); sb.z = fmaf(xv.z, xv.z, sb.z); sb.w = fmaf(xv.w, xv.w, sb.w);
                } else {
                    const float4 dv = ((const float4*)(dy + (size_t)r * C))[cq];
#define P1(f)                                                                   \
                    {                                                           \
                        const float u = fmaf(xv.f, scale.f, shift.f);           \
                        const float d = dv.f * (u > 0.f ? 1.f : leak);          \
                        sa.f += d; sb.f = fmaf(d, (xv.f - mean.f) * invstd.f, sb.f); \
                    }
                    P1(x) P1(y) P1(z) P1(w)
#undef P1
                }
            }
        }
        red[0][tid] = sa; red[1][tid] = sb;
        __syncthreads();
        if (rg == 0 && cq < CQ) {
            float4 ta = make_float4(0.f, 0.f, 0.f, 0.f), tb = ta;
            for (int gq = 0; gq < RG; ++gq) {
                const float4 a = red[0][gq * tpr + tc], b = red[1][gq * tpr + tc];
                ta.x += a.x; ta.y += a.y; ta.z += a.z; ta.w += a.w;
                tb.x += b.x; tb.y += b.y; tb.z += b.z; tb.w += b.w;
            }
            ((float4*)(part + ((size_t)blockIdx.x * 2 + 0) * C))[cq] = ta;
            ((float4*)(part + ((size_t)blockIdx.x * 2 + 1) * C))[cq] = tb;
        }
        __syncthreads();
    }
}

// MODE 0: -> stat = {mean, invstd, scale = gamma*invstd, shift = beta - mean*scale}; also mean/invstd outputs
// MODE 1: -> stat2 = {mean(dy'), mean(dy'*xhat)}
// Block = 8 channels x 32 slices of the G partials; each slice is summed in double in a fixed order and the 32 slice sums are
// added in a fixed order -> deterministic, G/32 loads deep, C/8 blocks (the fused conv statistics bring G = 4096 partial rows for
// the 16x16x128 layer: with 16 channels x 16 slices and C/16 = 8 blocks this kernel took 42 us there).
#define BNF_CH 8
#define BNF_SL 32
// nseg > 1 (statistics left by a convolution, cgs_conv_stat_layout): a group's G rows repeat in nseg segments seg_stride rows apart.
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int G, int M, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ stat, float* __restrict__ mean_out,
                                                          float* __restrict__ invstd_out, int nseg = 1, int seg_stride = 0) {
    __shared__ double red[2][BNF_SL][BNF_CH + 1];
    part += (size_t)blockIdx.y * G * 2 * C;
    stat += (size_t)blockIdx.y * (MODE == 0 ? 4 : 2) * C;
    if (MODE == 0) { mean_out += (size_t)blockIdx.y * C; invstd_out += (size_t)blockIdx.y * C; }
    const int cl = threadIdx.x % BNF_CH, sl = threadIdx.x / BNF_CH;
    const int c = blockIdx.x * BNF_CH + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int q = sl; q < nseg * G; q += BNF_SL) {                   // (fixed order per slice: deterministic)
            const int sg = q / G, g = q - sg * G;
            const float* row = part + ((size_t)sg * seg_stride + g) * 2 * C;
            a += (double)row[c]; b += (double)row[C + c];
        }
    red[0][sl][cl] = a; red[1][sl][cl] = b;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    a = 0.0; b = 0.0;
    for (int i = 0; i < BNF_SL; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
    if (MODE == 0) {
        const double mean = a / M;
        double var = b / M - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * invstd;
        stat[c] = (float)mean; stat[C + c] = invstd; stat[2 * C + c] = scale; stat[3 * C + c] = beta[c] - (float)mean * scale;
        mean_out[c] = (float)mean; invstd_out[c] = invstd;
    } else {
        stat[c] = (float)(a / M); stat[C + c] = (float)(b / M);
    }
}

// Two-stage form of bn_finalize_kernel<0> for the many partial rows a fused convolution leaves (two per 128-row m-tile: G = 4096
// for the 16x16x128 layer at batch 1024; the one-stage kernel has C/8 = 16 blocks walk them in 128 dependent steps: 23 us).
// Stage A: block (channel group, slice) sums its slice of the rows in double; stage B: one thread per channel adds the slices
// in a fixed order and finishes the statistics.  No atomics: deterministic.
#define BNF_SLICES 16
__global__ __launch_bounds__(256) void bn_slice_sums_kernel(const float* __restrict__ part, int G, int C, double* __restrict__ slices) {
    __shared__ double red[2][BNF_SL][BNF_CH + 1];
    const int cl = threadIdx.x % BNF_CH, sl = threadIdx.x / BNF_CH;
    const int c = blockIdx.x * BNF_CH + cl;
    const int per = (G + BNF_SLICES - 1) / BNF_SLICES;
    const int g0 = blockIdx.y * per, g1 = g0 + per < G ? g0 + per : G;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int g = g0 + sl; g < g1; g += BNF_SL) { a += (double)part[((size_t)g * 2 + 0) * C + c]; b += (double)part[((size_t)g * 2 + 1) * C + c]; }
    red[0][sl][cl] = a; red[1][sl][cl] = b;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    a = 0.0; b = 0.0;
    for (int i = 0; i < BNF_SL; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
    slices[((size_t)blockIdx.y * 2 + 0) * C + c] = a; slices[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
}

__global__ void bn_finalize_slices_kernel(const double* __restrict__ slices, int M, int C, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float eps, float* __restrict__ stat,
                                          float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < BNF_SLICES; ++i) { a += slices[((size_t)i * 2 + 0) * C + c]; b += slices[((size_t)i * 2 + 1) * C + c]; }
    const double mean = a / M;
    double var = b / M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * invstd;
    stat[c] = (float)mean; stat[C + c] = invstd; stat[2 * C + c] = scale; stat[3 * C + c] = beta[c] - (float)mean * scale;
    mean_out[c] = (float)mean; invstd_out[c] = invstd;
}

// ... and for the two backward column sums (stat2 = {mean(dy'), mean(dy' * xhat)}) when the producing backward-data launch left them
// as partial rows (cgs_norm_lrelu_bwd_from_partials).
__global__ void bn_finalize2_slices_kernel(const double* __restrict__ slices, int M, int C, float* __restrict__ stat2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < BNF_SLICES; ++i) { a += slices[((size_t)i * 2 + 0) * C + c]; b += slices[((size_t)i * 2 + 1) * C + c]; }
    stat2[c] = (float)(a / M); stat2[C + c] = (float)(b / M);
}

// The two streaming passes of a norm.  A thread walks the tensor with a stride of gridDim.x * 256 float4; whenever that stride is a
// whole number of channel rows (C | 1024 * gridDim.x: every power-of-two channel count), its four channels never change, so the
// per-channel parameters are loaded ONCE per thread (and group) instead of with every element -- the element loop then issues one
// 16-byte load per 16 bytes streamed (round 3's form: 3 loads forward, 8 backward, all through the same vector-memory path: 4.5-4.8 TB/s).
// Same arithmetic, same order: bit-identical results.

// y = lrelu(scale*x + shift)
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ stat0,
                                                           float leak, float* __restrict__ y, size_t n4, int C, size_t group_n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;                                       // (before the parameter loads: a thread past the end has no group)
    if (((stride * 4) % (size_t)C) == 0) {                     // (wave-uniform)
        const int c = (int)((i * 4) % C);
        size_t grp = i / group_n4, grp_end = (grp + 1) * group_n4;
        float4 sc = *(const float4*)(stat0 + grp * 4 * C + 2 * C + c), sh = *(const float4*)(stat0 + grp * 4 * C + 3 * C + c);
        for (; i < n4; i += stride) {
            if (i >= grp_end) {                                // next group of images (instance norm / fused logical batches)
                grp = i / group_n4; grp_end = (grp + 1) * group_n4;
                sc = *(const float4*)(stat0 + grp * 4 * C + 2 * C + c); sh = *(const float4*)(stat0 + grp * 4 * C + 3 * C + c);
            }
            const float4 v = ((const float4*)x)[i];
            float4 o;
            o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
            o.x = o.x > 0.f ? o.x : leak * o.x; o.y = o.y > 0.f ? o.y : leak * o.y;
            o.z = o.z > 0.f ? o.z : leak * o.z; o.w = o.w > 0.f ? o.w : leak * o.w;
            ((float4*)y)[i] = o;
        }
        return;
    }
    for (; i < n4; i += stride) {
        const float* stat = stat0 + (i / group_n4) * 4 * C;
        const float* scale = stat + 2 * C;
        const float* shift = stat + 3 * C;
        const int c = (int)((i * 4) % C);
        const float4 v = ((const float4*)x)[i];
        const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
        float4 o;
        o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
        o.x = o.x > 0.f ? o.x : leak * o.x; o.y = o.y > 0.f ? o.y : leak * o.y;
        o.z = o.z > 0.f ? o.z : leak * o.z; o.w = o.w > 0.f ? o.w : leak * o.w;
        ((float4*)y)[i] = o;
    }
}

// dx = gamma*invstd*(dy' - m1 - xhat*m2)
#define BWD1(f)                                                       \
        {                                                             \
            const float u = fmaf(xv.f, sc.f, sh.f);                   \
            const float d = dv.f * (u > 0.f ? 1.f : leak);            \
            const float xh = (xv.f - mean.f) * inv.f;                 \
            o.f = sc.f * (d - m1.f - xh * m2.f);                      \
        }
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean0, const float* __restrict__ invstd0,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ stat20,
                                                           float leak, float* __restrict__ dx, size_t n4, int C, size_t group_n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    if (((stride * 4) % (size_t)C) == 0) {                     // (wave-uniform) the thread's four channels are fixed: parameters once per group
        const int c = (int)((i * 4) % C);
        size_t grp = i / group_n4, grp_end = (grp + 1) * group_n4;
        float4 mean, inv, sc, sh, m1, m2;
        bn_affine4(mean0 + grp * C, invstd0 + grp * C, gamma, beta, c, mean, inv, sc, sh);
        m1 = *(const float4*)(stat20 + grp * 2 * C + c); m2 = *(const float4*)(stat20 + grp * 2 * C + C + c);
        for (; i < n4; i += stride) {
            if (i >= grp_end) {
                grp = i / group_n4; grp_end = (grp + 1) * group_n4;
                bn_affine4(mean0 + grp * C, invstd0 + grp * C, gamma, beta, c, mean, inv, sc, sh);
                m1 = *(const float4*)(stat20 + grp * 2 * C + c); m2 = *(const float4*)(stat20 + grp * 2 * C + C + c);
            }
            const float4 xv = ((const float4*)x)[i], dv = ((const float4*)dy)[i];
            float4 o;
            BWD1(x) BWD1(y) BWD1(z) BWD1(w)
            ((float4*)dx)[i] = o;
        }
        return;
    }
    for (; i < n4; i += stride) {
        const size_t grp = i / group_n4;
        const float* stat2 = stat20 + grp * 2 * C;
        const int c = (int)((i * 4) % C);
        const float4 xv = ((const float4*)x)[i], dv = ((const float4*)dy)[i];
        float4 mean, inv, sc, sh;
        bn_affine4(mean0 + grp * C, invstd0 + grp * C, gamma, beta, c, mean, inv, sc, sh);
        const float4 m1 = *(const float4*)(stat2 + c), m2 = *(const float4*)(stat2 + C + c);
        float4 o;
        BWD1(x) BWD1(y) BWD1(z) BWD1(w)
        ((float4*)dx)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Small groups (<= 128 rows per group: the batch norm behind the fully connected layer of the reference's MNIST D, nsgan/GAN.py:66-67, at
// its own batch size 64 -- one group per logical batch when several share a launch): a block = (64 channels, one group) keeps its rows in
// registers, so statistics AND apply are ONE launch instead of three latency-bound ones (partial sums, finalize, apply: 20 us forward /
// 37 us backward for an 8 MB tensor, round 5).  Same arithmetic as the three-kernel form: float partial sums per row lane, the 16 lanes
// added in index order in double, var = E[x^2] - mean^2 in double; deterministic.
// ------------------------------------------------------------------------------------------------
#define NS_ROWS 8            // rows per thread: 16 row lanes x 8 = 128 rows
template <bool BWD>
__global__ __launch_bounds__(256) void norm_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float leak, float* __restrict__ out,
                                                         float* __restrict__ mean_io, float* __restrict__ invstd_io, float* __restrict__ stat2,
                                                         int M, int C) {
    __shared__ float4 red[2][16][17];
    __shared__ float4 bc[2][16];
    const int tid = threadIdx.x, q = tid & 15, rl = tid >> 4;
    const int c = blockIdx.x * 64 + q * 4;
    const int grp = blockIdx.y;
    const bool live = c < C;
    const size_t base = (size_t)grp * M * C;
    float4 xv[NS_ROWS], dv[NS_ROWS];
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    float4 mu = sa, inv = sa, sc = sa, sh = sa;
    if (BWD && live) bn_affine4(mean_io + (size_t)grp * C, invstd_io + (size_t)grp * C, gamma, beta, c, mu, inv, sc, sh);
#pragma unroll
    for (int i = 0; i < NS_ROWS; ++i) {
        const int r = rl + 16 * i;
        xv[i] = make_float4(0.f, 0.f, 0.f, 0.f); dv[i] = xv[i];
        if (live && r < M) {
            xv[i] = *(const float4*)(x + base + (size_t)r * C + c);
            if (BWD) dv[i] = *(const float4*)(dy + base + (size_t)r * C + c);
            if (!BWD) {
                sa.x += xv[i].x; sa.y += xv[i].y; sa.z += xv[i].z; sa.w += xv[i].w;
                sb.x = fmaf(xv[i].x, xv[i].x, sb.x); sb.y = fmaf(xv[i].y, xv[i].y, sb.y); sb.z = fmaf(xv[i].z, xv[i].z, sb.z); sb.w = fmaf(xv[i].w, xv[i].w, sb.w);
            } else {
#define NS1(f)                                                                          \
                {                                                                       \
                    const float u = fmaf(xv[i].f, sc.f, sh.f);                          \
                    const float d = dv[i].f * (u > 0.f ? 1.f : leak);                   \
                    dv[i].f = d;                                                        \
                    sa.f += d; sb.f = fmaf(d, (xv[i].f - mu.f) * inv.f, sb.f);          \
                }
                NS1(x) NS1(y) NS1(z) NS1(w)
#undef NS1
            }
        }
    }
    red[0][rl][q] = sa; red[1][rl][q] = sb;
    __syncthreads();
    if (rl == 0 && live) {
        double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        for (int g = 0; g < 16; ++g) {
            const float4 u = red[0][g][q], v = red[1][g][q];
            a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
            b[0] += v.x; b[1] += v.y; b[2] += v.z; b[3] += v.w;
        }
        float o0[4], o1[4];
        if (!BWD) {
            for (int e = 0; e < 4; ++e) {
                const double mean = a[e] / M;
                double var = b[e] / M - mean * mean;
                if (var < 0.0) var = 0.0;
                o0[e] = (float)mean; o1[e] = (float)(1.0 / sqrt(var + (double)eps));
            }
            *(float4*)(mean_io + (size_t)grp * C + c) = make_float4(o0[0], o0[1], o0[2], o0[3]);
            *(float4*)(invstd_io + (size_t)grp * C + c) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        } else {
            for (int e = 0; e < 4; ++e) { o0[e] = (float)(a[e] / M); o1[e] = (float)(b[e] / M); }
            if (stat2) {
                *(float4*)(stat2 + (size_t)grp * 2 * C + c) = make_float4(o0[0], o0[1], o0[2], o0[3]);
                *(float4*)(stat2 + (size_t)grp * 2 * C + C + c) = make_float4(o1[0], o1[1], o1[2], o1[3]);
            }
        }
        bc[0][q] = make_float4(o0[0], o0[1], o0[2], o0[3]); bc[1][q] = make_float4(o1[0], o1[1], o1[2], o1[3]);
    }
    __syncthreads();
    if (!live) return;
    const float4 s0 = bc[0][q], s1 = bc[1][q];
    if (!BWD) {      // y = lrelu(scale * x + shift): the affine exactly as bn_finalize_kernel<0> forms it
        const float4 g = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
        float4 scl, shf;
        scl.x = g.x * s1.x; scl.y = g.y * s1.y; scl.z = g.z * s1.z; scl.w = g.w * s1.w;
        shf.x = bt.x - s0.x * scl.x; shf.y = bt.y - s0.y * scl.y; shf.z = bt.z - s0.z * scl.z; shf.w = bt.w - s0.w * scl.w;
#pragma unroll
        for (int i = 0; i < NS_ROWS; ++i) {
            const int r = rl + 16 * i;
            if (r >= M) break;
            float4 o;
            o.x = fmaf(xv[i].x, scl.x, shf.x); o.y = fmaf(xv[i].y, scl.y, shf.y); o.z = fmaf(xv[i].z, scl.z, shf.z); o.w = fmaf(xv[i].w, scl.w, shf.w);
            o.x = o.x > 0.f ? o.x : leak * o.x; o.y = o.y > 0.f ? o.y : leak * o.y;
            o.z = o.z > 0.f ? o.z : leak * o.z; o.w = o.w > 0.f ? o.w : leak * o.w;
            *(float4*)(out + base + (size_t)r * C + c) = o;
        }
    } else {         // dx = gamma * invstd * (dy' - m1 - xhat * m2)
#pragma unroll
        for (int i = 0; i < NS_ROWS; ++i) {
            const int r = rl + 16 * i;
            if (r >= M) break;
            float4 o;
            o.x = sc.x * (dv[i].x - s0.x - (xv[i].x - mu.x) * inv.x * s1.x);
            o.y = sc.y * (dv[i].y - s0.y - (xv[i].y - mu.y) * inv.y * s1.y);
            o.z = sc.z * (dv[i].z - s0.z - (xv[i].z - mu.z) * inv.z * s1.z);
            o.w = sc.w * (dv[i].w - s0.w - (xv[i].w - mu.w) * inv.w * s1.w);
            *(float4*)(out + base + (size_t)r * C + c) = o;
        }
    }
}
#define NORM_SMALL_MAX_ROWS (16 * NS_ROWS)
static bool norm_small_ok(int rows) {
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_NORM_SMALL") && atoi(getenv("CGS_NORM_SMALL")) == 0) return false;      // (A/B switch of experiment builds)
#endif
    return rows <= NORM_SMALL_MAX_ROWS;
}

static unsigned ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b == 0) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// Finalize + apply in ONE launch for norms whose statistics arrive as FEW partial rows (round 6): at the reference's batch size (64 images a
// call, nsgan/main.py:32) a norm's tensor is 0.5-8 MB and its group owns 4-256 partial rows (one per 64 GEMM rows of the producing launch), so
// the finalize kernel and the apply kernel are two dependent launches of ~5 us each whatever they do -- 6 of the ~40 launches of a refinement
// step on the DCGAN nets.  Here every block re-derives the statistics of its 64 channels from the partial rows (<= 256 rows x 2 x 64 floats out
// of L2: 16 row lanes sum their rows in double in index order, the 16 lane sums are added in index order -- deterministic) and applies them to its
// chunk of rows; the block of the first chunk also writes the saved mean / invstd (forward).  Same formulas as bn_finalize_kernel + bn_apply_*.
// Taken for groups of <= 16 partial rows and tensors of <= 2 MB (norm_fa_rows_per_block): beyond, the per-block re-derivation costs what the launch saved.
// Layout of the partial rows as in bn_finalize_kernel: group g owns, for every segment sg < nseg, the rows sg * seg_stride + g * R ... + R - 1.
// ------------------------------------------------------------------------------------------------
#define NFA_MAX_ROWS 256
template <bool BWD>
__global__ __launch_bounds__(256) void norm_fa_kernel(const float* __restrict__ x, const float* dy, const float* __restrict__ part, int R, int nseg,
                                                      int seg_stride, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      float leak, float* out, float* __restrict__ mean_io, float* __restrict__ invstd_io, int M, int C,
                                                      int rows_per_block) {
    __shared__ double red[2][16][16][4];
    __shared__ float4 bc[2][16];
    const int tid = threadIdx.x, q = tid & 15, rl = tid >> 4;
    const int c = blockIdx.x * 64 + q * 4;
    const int grp = blockIdx.z;
    const bool live = c < C;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (live) {
        const float* pg = part + (size_t)grp * R * 2 * C;
        for (int r = rl; r < nseg * R; r += 16) {          // (fixed order per row lane: deterministic)
            const int sg = r / R, g = r - sg * R;
            const float* row = pg + ((size_t)sg * seg_stride + g) * 2 * C;
            const float4 u = *(const float4*)(row + c), v = *(const float4*)(row + C + c);
            a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
            b[0] += v.x; b[1] += v.y; b[2] += v.z; b[3] += v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][rl][q][e] = a[e]; red[1][rl][q][e] = b[e]; }
    __syncthreads();
    if (rl == 0 && live) {
        float o0[4], o1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double sa = 0.0, sb = 0.0;
            for (int i = 0; i < 16; ++i) { sa += red[0][i][q][e]; sb += red[1][i][q][e]; }
            if (!BWD) {
                const double mean = sa / M;
                double var = sb / M - mean * mean;
                if (var < 0.0) var = 0.0;
                o0[e] = (float)mean; o1[e] = (float)(1.0 / sqrt(var + (double)eps));
            } else {
                o0[e] = (float)(sa / M); o1[e] = (float)(sb / M);
            }
        }
        bc[0][q] = make_float4(o0[0], o0[1], o0[2], o0[3]); bc[1][q] = make_float4(o1[0], o1[1], o1[2], o1[3]);
        if (!BWD && blockIdx.y == 0) {
            *(float4*)(mean_io + (size_t)grp * C + c) = bc[0][q];
            *(float4*)(invstd_io + (size_t)grp * C + c) = bc[1][q];
        }
    }
    __syncthreads();
    if (!live) return;
    const float4 s0 = bc[0][q], s1 = bc[1][q];
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const size_t base = (size_t)grp * M * C + c;
    if (!BWD) {      // y = lrelu(scale * x + shift): the affine exactly as bn_finalize_kernel<0> forms it
        const float4 g = *(const float4*)(gamma + c), bt = *(const float4*)(beta + c);
        float4 sc, sh;
        sc.x = g.x * s1.x; sc.y = g.y * s1.y; sc.z = g.z * s1.z; sc.w = g.w * s1.w;
        sh.x = bt.x - s0.x * sc.x; sh.y = bt.y - s0.y * sc.y; sh.z = bt.z - s0.z * sc.z; sh.w = bt.w - s0.w * sc.w;
        for (int r = r0 + rl; r < r1; r += 16) {
            const float4 v = *(const float4*)(x + base + (size_t)r * C);
            float4 o;
            o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
            o.x = o.x > 0.f ? o.x : leak * o.x; o.y = o.y > 0.f ? o.y : leak * o.y;
            o.z = o.z > 0.f ? o.z : leak * o.z; o.w = o.w > 0.f ? o.w : leak * o.w;
            *(float4*)(out + base + (size_t)r * C) = o;
        }
    } else {         // dx = gamma * invstd * (dy' - m1 - xhat * m2)   (dy / dx may be the same buffer: every thread reads its element before it writes it)
        float4 mean, inv, sc, sh;
        bn_affine4(mean_io + (size_t)grp * C, invstd_io + (size_t)grp * C, gamma, beta, c, mean, inv, sc, sh);
        const float4 m1 = s0, m2 = s1;
        for (int r = r0 + rl; r < r1; r += 16) {
            const float4 xv = *(const float4*)(x + base + (size_t)r * C), dv = *(const float4*)(dy + base + (size_t)r * C);
            float4 o;
            BWD1(x) BWD1(y) BWD1(z) BWD1(w)
            *(float4*)(out + base + (size_t)r * C) = o;
        }
    }
}

#undef BWD1

// rows per block of norm_fa_kernel for a call, 0 = the two-launch form serves it (too many partial rows, or a tensor large enough that the
// statistics' re-derivation per block and the plain row loop would cost more than the launch they save)
static int norm_fa_rows_per_block(int groups, int rows_per_seg, int nseg, int M_group, int C) {
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_NORM_FA") && atoi(getenv("CGS_NORM_FA")) == 0) return 0;
#endif
    // (limits from a same-process A/B over (rows, MB), profiles/r06_za_norm_fa_limits_ab.txt: every block re-derives the statistics, so the launch it saves
    // is only a gain while that is a handful of loads -- <= 16 rows, <= 2 MB: dcgan32 at batch 64 -2.2 % per call, neutral elsewhere; with <= 64 rows / 4 MB
    // -0.8 % there and +0.2 ... +0.4 % on dcgan64 at batch 64 and dcgan32 at batch 256)
    long max_rows = 16;
    size_t max_bytes = (size_t)2 << 20;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_NORM_FA_ROWS")) max_rows = atol(getenv("CGS_NORM_FA_ROWS"));
    if (getenv("CGS_NORM_FA_MB")) max_bytes = (size_t)atol(getenv("CGS_NORM_FA_MB")) << 20;
#endif
    if ((long)rows_per_seg * nseg > max_rows || (long)rows_per_seg * nseg > NFA_MAX_ROWS || groups > 65535) return 0;
    if ((size_t)groups * M_group * C * sizeof(float) > max_bytes) return 0;
    const long cg = cgs_ceil_div(C, 64);
    long chunks = 1024 / ((long)groups * cg);            // ~1024 blocks in all
    if (chunks < 1) chunks = 1;
    long rpb = cgs_ceil_div(M_group, (int)chunks);
    if (rpb < 16) rpb = 16;
    rpb = (rpb + 15) / 16 * 16;
    if (cgs_ceil_div(M_group, (int)rpb) > 65535) return 0;
    return (int)rpb;
}

int cgs_bn_train_lrelu_fwd(const float* x, const float* gamma, const float* beta, float eps, float leak, float* y,
                           float* mean, float* invstd, int M, int C, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || C <= 0 || (C & 3)) return cgs_set_error(CGS_EINVAL, "bn fwd: M=%d C=%d (C must be a multiple of 4)", M, C);
    if (ws_bytes < cgs_bn_ws_bytes(M, C)) return cgs_set_error(CGS_EWORKSPACE, "bn fwd: workspace %zu < %zu", ws_bytes, cgs_bn_ws_bytes(M, C));
    hipStream_t s = (hipStream_t)stream;
    if (norm_small_ok(M)) {               // one launch: statistics + apply from registers
        hipLaunchKernelGGL(norm_small_kernel<false>, dim3(cgs_ceil_div(C, 64), 1), dim3(256), 0, s, x, nullptr, gamma, beta, eps, leak, y, mean, invstd, nullptr, M, C);
        CGS_CHECK_LAUNCH("bn_train_lrelu_fwd");
        return CGS_OK;
    }
    const BnGeom g = bn_geom(M, C);
    float* part = (float*)ws;
    float* stat = part + (size_t)BN_MAX_BLOCKS * 2 * C;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(g.G), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, leak, part, M, C, g.rows_per_block);
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(cgs_ceil_div(C, BNF_CH)), dim3(256), 0, s, part, g.G, M, C, gamma, beta, eps, stat, mean, invstd);
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, stat, leak, y, n4, C, n4);
    CGS_CHECK_LAUNCH("bn_train_lrelu_fwd");
    return CGS_OK;
}

int cgs_bn_train_lrelu_fwd_from_partials(const float* x, const float* part, int G, const float* gamma, const float* beta, float eps,
                                         float leak, float* y, float* mean, float* invstd, int M, int C, void* ws, size_t ws_bytes,
                                         void* stream) {
    if (M <= 0 || C <= 0 || (C & 3) || G <= 0 || !part) return cgs_set_error(CGS_EINVAL, "bn fwd from partials: M=%d C=%d G=%d", M, C, G);
    if (ws_bytes < cgs_bn_ws_bytes(M, C)) return cgs_set_error(CGS_EWORKSPACE, "bn fwd: workspace %zu < %zu", ws_bytes, cgs_bn_ws_bytes(M, C));
    hipStream_t s = (hipStream_t)stream;
    if (const int rpb = norm_fa_rows_per_block(1, G, 1, M, C)) {          // few partial rows, small tensor: finalize + apply in one launch
        hipLaunchKernelGGL(norm_fa_kernel<false>, dim3(cgs_ceil_div(C, 64), cgs_ceil_div(M, rpb), 1), dim3(256), 0, s, x, nullptr, part, G, 1, 0, gamma, beta, eps,
                           leak, y, mean, invstd, M, C, rpb);
        CGS_CHECK_LAUNCH("bn_train_lrelu_fwd_from_partials");
        return CGS_OK;
    }
    float* stat = (float*)ws + (size_t)BN_MAX_BLOCKS * 2 * C;
    if (G >= 1024 && !((uintptr_t)ws & 7)) {      // (the workspace's own partial area is unused on this path: BN_MAX_BLOCKS * 2 * C floats >= 16 * 2 * C doubles)
        double* slices = (double*)ws;
        hipLaunchKernelGGL(bn_slice_sums_kernel, dim3(cgs_ceil_div(C, BNF_CH), BNF_SLICES), dim3(256), 0, s, part, G, C, slices);
        hipLaunchKernelGGL(bn_finalize_slices_kernel, dim3(cgs_ceil_div(C, 64)), dim3(64), 0, s, slices, M, C, gamma, beta, eps, stat, mean, invstd);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(cgs_ceil_div(C, BNF_CH)), dim3(256), 0, s, part, G, M, C, gamma, beta, eps, stat, mean, invstd);
    }
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, stat, leak, y, n4, C, n4);
    CGS_CHECK_LAUNCH("bn_train_lrelu_fwd_from_partials");
    return CGS_OK;
}

int cgs_bn_train_lrelu_bwd_data(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, float leak, float* dx, int M, int C, void* ws, size_t ws_bytes,
                                void* stream) {
    if (M <= 0 || C <= 0 || (C & 3)) return cgs_set_error(CGS_EINVAL, "bn bwd: M=%d C=%d (C must be a multiple of 4)", M, C);
    if (ws_bytes < cgs_bn_ws_bytes(M, C)) return cgs_set_error(CGS_EWORKSPACE, "bn bwd: workspace %zu < %zu", ws_bytes, cgs_bn_ws_bytes(M, C));
    hipStream_t s = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    float* part = (float*)ws;
    float* stat = part + (size_t)BN_MAX_BLOCKS * 2 * C;           // [4][C]
    float* stat2 = stat + 4 * (size_t)C;                          // [2][C]
    if (norm_small_ok(M)) {               // one launch (stat2 is left where cgs_bn_train_param_grads reads it)
        hipLaunchKernelGGL(norm_small_kernel<true>, dim3(cgs_ceil_div(C, 64), 1), dim3(256), 0, s, x, dy, gamma, beta, 0.f, leak, dx, (float*)mean, (float*)invstd, stat2, M, C);
        CGS_CHECK_LAUNCH("bn_train_lrelu_bwd_data");
        return CGS_OK;
    }
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(g.G), dim3(256), 0, s, x, dy, mean, invstd, gamma, beta, leak, part, M, C, g.rows_per_block);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(cgs_ceil_div(C, BNF_CH)), dim3(256), 0, s, part, g.G, M, C, nullptr, nullptr, 0.f, stat2, nullptr, nullptr);
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, mean, invstd, gamma, beta, stat2, leak, dx, n4, C, n4);
    CGS_CHECK_LAUNCH("bn_train_lrelu_bwd_data");
    return CGS_OK;
}



// ------------------------------------------------------------------------------------------------
// Synchronised batch statistics (one logical batch split over several GPUs; SURVEY.md 8e caveat).
// The local reduction stops at per-channel SUMS in double; the caller all-reduces the 2*C doubles over the ranks
// (RCCL) and the second half finishes the statistics from the global sums and the global row count.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_sums_kernel(const float* __restrict__ part, int G, int C, double* __restrict__ sums) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int g = sl; g < G; g += 16) { a += (double)part[((size_t)g * 2 + 0) * C + c]; b += (double)part[((size_t)g * 2 + 1) * C + c]; }
    red[0][sl][cl] = a; red[1][sl][cl] = b;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    a = 0.0; b = 0.0;
    for (int i = 0; i < 16; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
    sums[c] = a; sums[C + c] = b;
}

__global__ void bn_stats_from_sums_kernel(const double* __restrict__ sums, double Mtot, int C, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float eps, float* __restrict__ stat,
                                          float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = sums[c] / Mtot;
    double var = sums[C + c] / Mtot - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float scale = gamma[c] * invstd;
    stat[c] = (float)mean; stat[C + c] = invstd; stat[2 * C + c] = scale; stat[3 * C + c] = beta[c] - (float)mean * scale;
    mean_out[c] = (float)mean; invstd_out[c] = invstd;
}

__global__ void bn_stat2_from_sums_kernel(const double* __restrict__ sums, double Mtot, int C, float* __restrict__ stat2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stat2[c] = (float)(sums[c] / Mtot); stat2[C + c] = (float)(sums[C + c] / Mtot);
}

static int bn_sync_check(const char* who, int M, int C, size_t ws_bytes, const void* sums) {
    if (M <= 0 || C <= 0 || (C & 3)) return cgs_set_error(CGS_EINVAL, "%s: M=%d C=%d (C must be a multiple of 4)", who, M, C);
    if (ws_bytes < cgs_bn_ws_bytes(M, C)) return cgs_set_error(CGS_EWORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, cgs_bn_ws_bytes(M, C));
    if (!sums || ((uintptr_t)sums & 7)) return cgs_set_error(CGS_EINVAL, "%s: sums must be an 8-byte aligned [2][C] double buffer", who);
    return CGS_OK;
}

int cgs_bn_sync_fwd_sums(const float* x, double* sums, int M, int C, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = bn_sync_check("bn sync fwd sums", M, C, ws_bytes, sums)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    float* part = (float*)ws;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(g.G), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, part, M, C, g.rows_per_block);
    hipLaunchKernelGGL(bn_sums_kernel, dim3(cgs_ceil_div(C, 16)), dim3(256), 0, s, part, g.G, C, sums);
    CGS_CHECK_LAUNCH("bn_sync_fwd_sums");
    return CGS_OK;
}

int cgs_bn_sync_fwd_apply(const float* x, const float* gamma, const float* beta, float eps, float leak, const double* sums,
                          long long M_total, float* y, float* mean, float* invstd, int M, int C, void* ws, size_t ws_bytes,
                          void* stream) {
    if (int rc = bn_sync_check("bn sync fwd apply", M, C, ws_bytes, sums)) return rc;
    if (M_total < M) return cgs_set_error(CGS_EINVAL, "bn sync fwd apply: M_total %lld < local M %d", M_total, M);
    hipStream_t s = (hipStream_t)stream;
    float* stat = (float*)ws + (size_t)BN_MAX_BLOCKS * 2 * C;
    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3(cgs_ceil_div(C, 128)), dim3(128), 0, s, sums, (double)M_total, C, gamma, beta, eps, stat, mean, invstd);
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, stat, leak, y, n4, C, n4);
    CGS_CHECK_LAUNCH("bn_sync_fwd_apply");
    return CGS_OK;
}

int cgs_bn_sync_bwd_sums(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                         const float* invstd, float leak, double* sums, int M, int C, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = bn_sync_check("bn sync bwd sums", M, C, ws_bytes, sums)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    float* part = (float*)ws;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(g.G), dim3(256), 0, s, x, dy, mean, invstd, gamma, beta, leak, part, M, C, g.rows_per_block);
    hipLaunchKernelGGL(bn_sums_kernel, dim3(cgs_ceil_div(C, 16)), dim3(256), 0, s, part, g.G, C, sums);
    CGS_CHECK_LAUNCH("bn_sync_bwd_sums");
    return CGS_OK;
}

int cgs_bn_sync_bwd_apply(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* invstd, float leak, const double* sums, long long M_total, float* dx, int M, int C,
                          void* ws, size_t ws_bytes, void* stream) {
    if (int rc = bn_sync_check("bn sync bwd apply", M, C, ws_bytes, sums)) return rc;
    if (M_total < M) return cgs_set_error(CGS_EINVAL, "bn sync bwd apply: M_total %lld < local M %d", M_total, M);
    hipStream_t s = (hipStream_t)stream;
    float* stat = (float*)ws + (size_t)BN_MAX_BLOCKS * 2 * C;
    float* stat2 = stat + 4 * (size_t)C;
    hipLaunchKernelGGL(bn_stat2_from_sums_kernel, dim3(cgs_ceil_div(C, 128)), dim3(128), 0, s, sums, (double)M_total, C, stat2);
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, mean, invstd, gamma, beta, stat2, leak, dx, n4, C, n4);
    CGS_CHECK_LAUNCH("bn_sync_bwd_apply");
    return CGS_OK;
}


// bias gradient db[c] (+)= sum_m dy[m][c]: the same two-stage deterministic column reduction.  Stage 2 in bn_finalize_kernel's form
// (8 channels x 32 slices of the G partial rows per block, each slice and then the 32 slice sums added in a fixed order in double): one
// thread per channel walking all G rows took 63 us on average in the D-shaping step of dcgan64 (profiles/r05_i_shaping_kernel_stats.csv).
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int G, int C, float* __restrict__ db, int accumulate) {
    __shared__ double red[BNF_SL][BNF_CH + 1];
    const int cl = threadIdx.x % BNF_CH, sl = threadIdx.x / BNF_CH;
    const int c = blockIdx.x * BNF_CH + cl;
    double a = 0.0;
    if (c < C)
        for (int g = sl; g < G; g += BNF_SL) a += (double)part[((size_t)g * 2 + 0) * C + c];
    red[sl][cl] = a;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    a = 0.0;
    for (int i = 0; i < BNF_SL; ++i) a += red[i][cl];
    db[c] = accumulate ? db[c] + (float)a : (float)a;
}

int cgs_bias_grad(const float* dy, float* db, int M, int C, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (M <= 0 || C <= 0 || (C & 3)) return cgs_set_error(CGS_EINVAL, "bias_grad: M=%d C=%d (C must be a multiple of 4)", M, C);
    if (ws_bytes < cgs_bn_ws_bytes(M, C)) return cgs_set_error(CGS_EWORKSPACE, "bias_grad: workspace %zu < %zu", ws_bytes, cgs_bn_ws_bytes(M, C));
    hipStream_t s = (hipStream_t)stream;
    const BnGeom g = bn_geom(M, C);
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(g.G), dim3(256), 0, s, dy, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, (float*)ws, M, C, g.rows_per_block);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cgs_ceil_div(C, BNF_CH)), dim3(256), 0, s, (const float*)ws, g.G, C, db, accumulate);
    CGS_CHECK_LAUNCH("bias_grad");
    return CGS_OK;
}


// ------------------------------------------------------------------------------------------------
// Instance norm (+ lrelu / relu): the same kernels with one independent group per sample (blockIdx.y) -- statistics
// over the H*W pixels of each (sample, channel).  Used by CycleGAN-style generators / PatchGAN discriminators
// (BASELINE config 5; no reference code exists for them).
// ------------------------------------------------------------------------------------------------
static BnGeom in_geom(int B, int HW) {
    BnGeom g;
    int maxb = 2048 / (B > 0 ? B : 1);
    if (maxb < 1) maxb = 1;
    if (maxb > BN_MAX_BLOCKS) maxb = BN_MAX_BLOCKS;
    int rpb = cgs_ceil_div(HW, maxb);
    if (rpb < 16) rpb = 16;
    g.rows_per_block = rpb;
    g.G = cgs_ceil_div(HW, rpb);
    return g;
}

size_t cgs_instnorm_ws_bytes(int B, int HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    const BnGeom g = in_geom(B, HW);
    return ((size_t)B * g.G * 2 * C + (size_t)B * 6 * C) * sizeof(float);   // partials | stat[B][4][C] | stat2[B][2][C]
}

int cgs_instnorm_lrelu_fwd(const float* x, const float* scale, const float* offset, float eps, float leak, float* y,
                           float* mean, float* invstd, int B, int HW, int C, void* ws, size_t ws_bytes, void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || (C & 3) || B > 65535) return cgs_set_error(CGS_EINVAL, "instnorm fwd: B=%d HW=%d C=%d", B, HW, C);
    if (ws_bytes < cgs_instnorm_ws_bytes(B, HW, C)) return cgs_set_error(CGS_EWORKSPACE, "instnorm fwd: workspace %zu < %zu", ws_bytes, cgs_instnorm_ws_bytes(B, HW, C));
    hipStream_t s = (hipStream_t)stream;
    if (norm_small_ok(HW)) {              // small groups: one launch (see norm_small_kernel)
        hipLaunchKernelGGL(norm_small_kernel<false>, dim3(cgs_ceil_div(C, 64), B), dim3(256), 0, s, x, nullptr, scale, offset, eps, leak, y, mean, invstd, nullptr, HW, C);
        CGS_CHECK_LAUNCH("instnorm_lrelu_fwd");
        return CGS_OK;
    }
    const BnGeom g = in_geom(B, HW);
    float* part = (float*)ws;
    float* stat = part + (size_t)B * g.G * 2 * C;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(g.G, B), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, leak, part, HW, C, g.rows_per_block);
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(cgs_ceil_div(C, BNF_CH), B), dim3(256), 0, s, part, g.G, HW, C, scale, offset, eps, stat, mean, invstd);
    const size_t n4 = (size_t)B * HW * C / 4, gn4 = (size_t)HW * C / 4;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, stat, leak, y, n4, C, gn4);
    CGS_CHECK_LAUNCH("instnorm_lrelu_fwd");
    return CGS_OK;
}

// Norm over GROUPS of rows whose statistics the producing convolution left as partial rows (cgs_conv_stat_layout): instance norm
// (a group = one sample) and the batch norm of several logical batches fused into one launch (a group = one logical batch).
// x is [groups][M_group][C]; mean / invstd [groups][C].  ws: >= groups * 4 * C floats (cgs_instnorm_ws_bytes(groups, M_group, C) covers it).
int cgs_groupnorm_lrelu_fwd_from_partials(const float* x, const float* part, int groups, int rows_per_seg, int nseg, int seg_stride,
                                          const float* gamma, const float* beta, float eps, float leak, float* y, float* mean, float* invstd,
                                          int M_group, int C, void* ws, size_t ws_bytes, void* stream) {
    if (groups <= 0 || groups > 65535 || M_group <= 0 || C <= 0 || (C & 3) || rows_per_seg <= 0 || nseg <= 0 || seg_stride < 0 || !part)
        return cgs_set_error(CGS_EINVAL, "groupnorm fwd from partials: groups=%d M=%d C=%d rows=%d x %d", groups, M_group, C, rows_per_seg, nseg);
    if (ws_bytes < (size_t)groups * 4 * C * sizeof(float))
        return cgs_set_error(CGS_EWORKSPACE, "groupnorm fwd from partials: workspace %zu < %zu", ws_bytes, (size_t)groups * 4 * C * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    if (const int rpb = norm_fa_rows_per_block(groups, rows_per_seg, nseg, M_group, C)) {
        hipLaunchKernelGGL(norm_fa_kernel<false>, dim3(cgs_ceil_div(C, 64), cgs_ceil_div(M_group, rpb), groups), dim3(256), 0, s, x, nullptr, part, rows_per_seg, nseg,
                           seg_stride, gamma, beta, eps, leak, y, mean, invstd, M_group, C, rpb);
        CGS_CHECK_LAUNCH("groupnorm_lrelu_fwd_from_partials");
        return CGS_OK;
    }
    float* stat = (float*)ws;                                          // [groups][4][C]
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(cgs_ceil_div(C, BNF_CH), groups), dim3(256), 0, s, part, rows_per_seg, M_group, C, gamma, beta, eps, stat,
                       mean, invstd, nseg, seg_stride);
    const size_t n4 = (size_t)groups * M_group * C / 4, gn4 = (size_t)M_group * C / 4;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, stat, leak, y, n4, C, gn4);
    CGS_CHECK_LAUNCH("groupnorm_lrelu_fwd_from_partials");
    return CGS_OK;
}

// Backward-data of a norm (+ lrelu) over groups of rows whose two column sums the PRODUCING backward-data launch left as partial rows
// (cgs_conv2d_nhwc_bwd_data_nstats / cgs_deconv2d_nhwc_bwd_data_nstats, layout = cgs_conv_stat_layout of that call): finalize + apply --
// the sums pass over dy and x (2 of the 5 tensor passes of cgs_bn_train_lrelu_bwd_data / cgs_instnorm_lrelu_bwd_data) is gone.
// ws: >= groups * 2 * C floats (+ 16 * 2 * C doubles for groups == 1 with >= 1024 partial rows); cgs_bn_ws_bytes / cgs_instnorm_ws_bytes cover it.
int cgs_norm_lrelu_bwd_from_partials(const float* dy, const float* x, const float* part, int groups, int rows_per_seg, int nseg, int seg_stride,
                                     const float* gamma, const float* beta, const float* mean, const float* invstd, float leak, float* dx,
                                     int M_group, int C, void* ws, size_t ws_bytes, void* stream) {
    if (groups <= 0 || groups > 65535 || M_group <= 0 || C <= 0 || (C & 3) || rows_per_seg <= 0 || nseg <= 0 || seg_stride < 0 || !part || !dy || !x || !dx)
        return cgs_set_error(CGS_EINVAL, "norm bwd from partials: groups=%d M=%d C=%d rows=%d x %d", groups, M_group, C, rows_per_seg, nseg);
    const bool sliced = groups == 1 && nseg == 1 && rows_per_seg >= 1024 && !((uintptr_t)ws & 7);
    const size_t need = (size_t)groups * 2 * C * sizeof(float) + (sliced ? (size_t)BNF_SLICES * 2 * C * sizeof(double) : 0);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "norm bwd from partials: workspace %zu < %zu", ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    if (const int rpb = norm_fa_rows_per_block(groups, rows_per_seg, nseg, M_group, C)) {
        hipLaunchKernelGGL(norm_fa_kernel<true>, dim3(cgs_ceil_div(C, 64), cgs_ceil_div(M_group, rpb), groups), dim3(256), 0, s, x, dy, part, rows_per_seg, nseg,
                           seg_stride, gamma, beta, 0.f, leak, dx, (float*)mean, (float*)invstd, M_group, C, rpb);
        CGS_CHECK_LAUNCH("norm_lrelu_bwd_from_partials");
        return CGS_OK;
    }
    float* stat2;
    if (sliced) {
        double* slices = (double*)ws;
        stat2 = (float*)(slices + (size_t)BNF_SLICES * 2 * C);
        hipLaunchKernelGGL(bn_slice_sums_kernel, dim3(cgs_ceil_div(C, BNF_CH), BNF_SLICES), dim3(256), 0, s, part, rows_per_seg, C, slices);
        hipLaunchKernelGGL(bn_finalize2_slices_kernel, dim3(cgs_ceil_div(C, 64)), dim3(64), 0, s, slices, M_group, C, stat2);
    } else {
        stat2 = (float*)ws;                                                   // [groups][2][C]
        hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(cgs_ceil_div(C, BNF_CH), groups), dim3(256), 0, s, part, rows_per_seg, M_group, C, nullptr, nullptr, 0.f,
                           stat2, nullptr, nullptr, nseg, seg_stride);
    }
    const size_t n4 = (size_t)groups * M_group * C / 4, gn4 = (size_t)M_group * C / 4;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, mean, invstd, gamma, beta, stat2, leak, dx, n4, C, gn4);
    CGS_CHECK_LAUNCH("norm_lrelu_bwd_from_partials");
    return CGS_OK;
}

int cgs_instnorm_lrelu_bwd_data(const float* dy, const float* x, const float* scale, const float* offset, const float* mean,
                                const float* invstd, float leak, float* dx, int B, int HW, int C, void* ws, size_t ws_bytes,
                                void* stream) {
    if (B <= 0 || HW <= 0 || C <= 0 || (C & 3) || B > 65535) return cgs_set_error(CGS_EINVAL, "instnorm bwd: B=%d HW=%d C=%d", B, HW, C);
    if (ws_bytes < cgs_instnorm_ws_bytes(B, HW, C)) return cgs_set_error(CGS_EWORKSPACE, "instnorm bwd: workspace %zu < %zu", ws_bytes, cgs_instnorm_ws_bytes(B, HW, C));
    hipStream_t s = (hipStream_t)stream;
    const BnGeom g = in_geom(B, HW);
    float* part = (float*)ws;
    float* stat = part + (size_t)B * g.G * 2 * C;
    float* stat2 = stat + (size_t)B * 4 * C;
    if (norm_small_ok(HW)) {
        hipLaunchKernelGGL(norm_small_kernel<true>, dim3(cgs_ceil_div(C, 64), B), dim3(256), 0, s, x, dy, scale, offset, 0.f, leak, dx, (float*)mean, (float*)invstd, stat2, HW, C);
        CGS_CHECK_LAUNCH("instnorm_lrelu_bwd_data");
        return CGS_OK;
    }
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(g.G, B), dim3(256), 0, s, x, dy, mean, invstd, scale, offset, leak, part, HW, C, g.rows_per_block);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(cgs_ceil_div(C, BNF_CH), B), dim3(256), 0, s, part, g.G, HW, C, nullptr, nullptr, 0.f, stat2, nullptr, nullptr);
    const size_t n4 = (size_t)B * HW * C / 4, gn4 = (size_t)HW * C / 4;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, mean, invstd, scale, offset, stat2, leak, dx, n4, C, gn4);
    CGS_CHECK_LAUNCH("instnorm_lrelu_bwd_data");
    return CGS_OK;
}
