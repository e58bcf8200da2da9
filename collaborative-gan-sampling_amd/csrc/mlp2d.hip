// The 2-D path of the reference on the GPU: synthetic/GAN.py:28-37,105-111 (ReLU MLP discriminator on 2-D points,
// its sigmoid and the "saliency" d mean_b softplus(-logit_b) / dx) and the whole host loop of
// sampling/refiner_cpu.py:19-81 (K ladam / momentum / sgd steps with best-loss tracking and trajectory recording)
// as ONE launch: BASELINE config 1 (Imbal-8Gaussians, batch 512, K = 10) without the K+2 host<->framework round trips.
//
// One wave per sample (samples are independent given the real-batch baseline), lane j = hidden unit j (nhidden <= 64):
//   forward  h_out[j] = relu(b[j] + sum_k h_in[k] * W[k][j])   : h_in[k] broadcast by readlane, W row k from LDS
//   backward g_in[k]  = sum_j g_out[j] * W[k][j]               : lane k walks row k of the TRANSPOSED copy W^T[j][k]
// Both LDS walks are stride-1 across lanes (conflict free).  All layers' weights (and transposes) live in LDS.
#include "cgs_internal.h"

#define MLP_MAX_LAYERS 6      // all layers (and their transposes) LDS-resident: 6 layers = 133 KB of 160 KB

struct MlpParams {
    const float* w[MLP_MAX_LAYERS];   // layer l: [din_l][dout_l] row-major (tf.layers.dense kernel)
    const float* b[MLP_MAX_LAYERS];
    int nlayers, nh;                   // dims: 2 -> nh -> ... -> nh -> 1
};

__device__ __forceinline__ float bcast(float v, int k) { return __shfl(v, k, 64); }

// LDS layout: per hidden->hidden layer l (1 .. nlayers-2): W [64][64] then W^T [64][64] (zero padded to 64);
// first layer W1 [2][64], b of every layer [64], last layer w [64].
struct MlpLds {
    float* w1;      // [2][64]
    float* wl;      // [64]   last layer column
    float* bias;    // [nlayers][64]
    float* wh;      // [(nlayers-2)][2][64][64]
};

__device__ __forceinline__ MlpLds mlp_lds(float* smem, int nlayers) {
    MlpLds L;
    L.w1 = smem; L.wl = smem + 128; L.bias = smem + 192; L.wh = smem + 192 + nlayers * 64;
    return L;
}

__device__ void mlp_load(const MlpParams& p, const MlpLds& L) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < 128; i += nt) { const int k = i >> 6, j = i & 63; L.w1[i] = j < p.nh ? p.w[0][k * p.nh + j] : 0.f; }
    for (int i = tid; i < 64; i += nt) L.wl[i] = i < p.nh ? p.w[p.nlayers - 1][i] : 0.f;
    for (int i = tid; i < p.nlayers * 64; i += nt) {
        const int l = i >> 6, j = i & 63;
        const int dout = l == p.nlayers - 1 ? 1 : p.nh;
        L.bias[i] = j < dout ? p.b[l][j] : 0.f;
    }
    for (int l = 1; l < p.nlayers - 1; ++l)
        for (int i = tid; i < 4096; i += nt) {
            const int k = i >> 6, j = i & 63;
            const float v = (k < p.nh && j < p.nh) ? p.w[l][k * p.nh + j] : 0.f;
            L.wh[(size_t)(l - 1) * 8192 + i] = v;                       // W[k][j]
            L.wh[(size_t)(l - 1) * 8192 + 4096 + j * 64 + k] = v;       // W^T[j][k]
        }
    __syncthreads();
}

// one evaluation for the wave's sample at (x0, x1): logit and d logit / d x.  Lane j = hidden unit j.
__device__ __forceinline__ void mlp_eval(const MlpParams& p, const MlpLds& L, float x0, float x1, int lane, float& logit,
                                         float& dldx0, float& dldx1) {
    unsigned long long masks[MLP_MAX_LAYERS];
    float h = fmaf(x1, L.w1[64 + lane], fmaf(x0, L.w1[lane], L.bias[lane]));
    masks[0] = __ballot(h > 0.f);
    h = fmaxf(h, 0.f);
    for (int l = 1; l < p.nlayers - 1; ++l) {
        const float* W = L.wh + (size_t)(l - 1) * 8192;
        float a = L.bias[l * 64 + lane];
#pragma unroll 8
        for (int k = 0; k < 64; ++k) a = fmaf(bcast(h, k), W[k * 64 + lane], a);
        masks[l] = __ballot(a > 0.f);
        h = fmaxf(a, 0.f);
    }
    float part = h * L.wl[lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    logit = part + L.bias[(p.nlayers - 1) * 64];
    // backward: g = d logit / d h (lane = unit)
    float g = L.wl[lane];
    for (int l = p.nlayers - 2; l >= 1; --l) {
        g = ((masks[l] >> lane) & 1ull) ? g : 0.f;
        const float* WT = L.wh + (size_t)(l - 1) * 8192 + 4096;
        float a = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < 64; ++jj) a = fmaf(bcast(g, jj), WT[jj * 64 + lane], a);
        g = a;
    }
    g = ((masks[0] >> lane) & 1ull) ? g : 0.f;
    float d0 = g * L.w1[lane], d1 = g * L.w1[64 + lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d0 += __shfl_xor(d0, off, 64); d1 += __shfl_xor(d1, off, 64); }
    dldx0 = d0; dldx1 = d1;
}

__device__ __forceinline__ float sigmoidf_(float v) { return v >= 0.f ? 1.f / (1.f + expf(-v)) : expf(v) / (1.f + expf(v)); }

// sigmoid [B] and saliency [B,2] = inv_batch * (sigmoid - 1) * d logit / dx   (synthetic/GAN.py:108-111)
__global__ __launch_bounds__(1024) void mlp_saliency_kernel(MlpParams p, const float* __restrict__ x, float* __restrict__ sig,
                                                            float* __restrict__ sal, int B, float inv_batch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpLds L = mlp_lds(smem, p.nlayers);
    mlp_load(p, L);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int s = blockIdx.x * nw + wave; s < B; s += gridDim.x * nw) {
        float logit, d0, d1;
        mlp_eval(p, L, x[2 * s], x[2 * s + 1], lane, logit, d0, d1);
        if (lane == 0) {
            const float sg = sigmoidf_(logit);
            sig[s] = sg;
            if (sal) { sal[2 * s] = inv_batch * (sg - 1.f) * d0; sal[2 * s + 1] = inv_batch * (sg - 1.f) * d1; }
        }
    }
}

// the whole refiner_cpu loop for one sample per wave.  method: 0 sgd, 1 momentum, 2 ladam (policy.py:26-61, numpy branch)
__global__ __launch_bounds__(1024) void refine2d_kernel(MlpParams p, const float* __restrict__ x_in, float real_mean_host,
                                                        const float* __restrict__ real_mean_dev,
                                                        float inv_batch, int steps, float rate, int method,
                                                        float* __restrict__ best_x, float* __restrict__ best_step,
                                                        float* __restrict__ traj /* [B][steps+1][2] or null */, int B) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpLds L = mlp_lds(smem, p.nlayers);
    mlp_load(p, L);
    const float real_mean = real_mean_dev ? real_mean_dev[0] : real_mean_host;      // device scalar: no host round trip per batch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int s = blockIdx.x * nw + wave; s < B; s += gridDim.x * nw) {
        float x0 = x_in[2 * s], x1 = x_in[2 * s + 1];
        float logit, d0, d1;
        mlp_eval(p, L, x0, x1, lane, logit, d0, d1);
        float sg = sigmoidf_(logit);
        float g0 = inv_batch * (sg - 1.f) * d0, g1 = inv_batch * (sg - 1.f) * d1;
        float loss = real_mean - sg;                                   // refiner_cpu.py:28
        float bx0 = x0, bx1 = x1, bl = loss, bs = 0.f;
        float m0 = 0.f, m1 = 0.f, v0 = 0.f, v1 = 0.f, ll = 0.f;
        if (traj && lane == 0) { traj[((size_t)s * (steps + 1)) * 2] = x0; traj[((size_t)s * (steps + 1)) * 2 + 1] = x1; }
        for (int i = 0; i < steps; ++i) {
#pragma clang fp contract(off)
            if (method == 0) {
                x0 -= rate * g0; x1 -= rate * g1;
            } else if (method == 1) {
                const float a0 = rate * g0, a1 = rate * g1;
                m0 = i == 0 ? a0 : 0.9f * m0 + a0; m1 = i == 0 ? a1 : 0.9f * m1 + a1;
                x0 -= m0; x1 -= m1;
            } else {                                                   // ladam, policy.py:39-61
                if (i == 0) { m0 = g0; m1 = g1; v0 = g0 * g0; v1 = g1 * g1; ll = loss; }
                else {
                    m0 = 0.9f * m0 + 0.1f * g0; m1 = 0.9f * m1 + 0.1f * g1;        // (1. - 0.9) rounds to 0.1f in float32
                    v0 = 0.5f * v0 + 0.5f * (g0 * g0); v1 = 0.5f * v1 + 0.5f * (g1 * g1);
                    ll = 0.5f * ll + 0.5f * loss;
                }
                const float c = fmaxf(ll + 0.5f, 0.f);
                const float rs = c * c;
                x0 -= rate * m0 / (sqrtf(v0) + 1e-8f) * rs; x1 -= rate * m1 / (sqrtf(v1) + 1e-8f) * rs;
            }
            mlp_eval(p, L, x0, x1, lane, logit, d0, d1);
            sg = sigmoidf_(logit);
            g0 = inv_batch * (sg - 1.f) * d0; g1 = inv_batch * (sg - 1.f) * d1;
            loss = real_mean - sg;
            if (bl - loss > 0.f) { bl = loss; bx0 = x0; bx1 = x1; bs = (float)(i + 1); }          // refiner_cpu.py:58-61
            if (traj && lane == 0) { traj[((size_t)s * (steps + 1) + i + 1) * 2] = x0; traj[((size_t)s * (steps + 1) + i + 1) * 2 + 1] = x1; }
        }
        if (lane == 0) { best_x[2 * s] = bx0; best_x[2 * s + 1] = bx1; best_step[s] = bs; }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// D shaping step of the 2-D net (synthetic/main.py:366-370 -> synthetic/GAN.py:69-74,98-99):
//   d_loss = mean_b BCE(D(real_b), 1) + mean_b BCE(D(refined_b), 0);  GradientDescentOptimizer(lrd).minimize(d_loss, d_vars)
// Pass A (one wave per sample, the same LDS-resident walk as above): forward keeping every hidden activation, backward
// from the loss seed keeping every pre-activation gradient.  Pass B (one block per layer): dW_l = A_{l-1}^T Delta_l and
// db_l = column sums of Delta_l, summed over the samples in a FIXED order (deterministic), then w -= lr * g in place.
// ------------------------------------------------------------------------------------------------------------------
// acts / deltas: [B_total][nlayers-1][64]; dlast / bce: [B_total]
__global__ __launch_bounds__(1024) void mlp_train_fwdbwd_kernel(MlpParams p, const float* __restrict__ x, int B, int row0, float target,
                                                                float scale, float* __restrict__ acts, float* __restrict__ deltas,
                                                                float* __restrict__ dlast, float* __restrict__ bce) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const MlpLds L = mlp_lds(smem, p.nlayers);
    mlp_load(p, L);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nhid = p.nlayers - 1;
    for (int s = blockIdx.x * nw + wave; s < B; s += gridDim.x * nw) {
        float* A = acts + (size_t)(row0 + s) * nhid * 64;
        float* D = deltas + (size_t)(row0 + s) * nhid * 64;
        unsigned long long masks[MLP_MAX_LAYERS];
        const float x0 = x[2 * s], x1 = x[2 * s + 1];
        float h = fmaf(x1, L.w1[64 + lane], fmaf(x0, L.w1[lane], L.bias[lane]));
        masks[0] = __ballot(h > 0.f);
        h = fmaxf(h, 0.f);
        A[lane] = h;
        for (int l = 1; l < nhid; ++l) {
            const float* W = L.wh + (size_t)(l - 1) * 8192;
            float a = L.bias[l * 64 + lane];
#pragma unroll 8
            for (int k = 0; k < 64; ++k) a = fmaf(bcast(h, k), W[k * 64 + lane], a);
            masks[l] = __ballot(a > 0.f);
            h = fmaxf(a, 0.f);
            A[l * 64 + lane] = h;
        }
        float part = h * L.wl[lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        const float logit = part + L.bias[(p.nlayers - 1) * 64];
        const float seed = scale * (sigmoidf_(logit) - target);            // d (scale * BCE(logit, target)) / d logit
        if (lane == 0) {
            dlast[row0 + s] = seed;
            bce[row0 + s] = scale * (fmaxf(logit, 0.f) - logit * target + log1pf(expf(-fabsf(logit))));
        }
        float g = seed * L.wl[lane];
        for (int l = nhid - 1; l >= 1; --l) {
            g = ((masks[l] >> lane) & 1ull) ? g : 0.f;
            D[l * 64 + lane] = g;
            const float* WT = L.wh + (size_t)(l - 1) * 8192 + 4096;
            float a = 0.f;
#pragma unroll 8
            for (int jj = 0; jj < 64; ++jj) a = fmaf(bcast(g, jj), WT[jj * 64 + lane], a);
            g = a;
        }
        g = ((masks[0] >> lane) & 1ull) ? g : 0.f;
        D[lane] = g;
    }
}

struct MlpTrainPtrs {
    float* w[MLP_MAX_LAYERS];
    float* b[MLP_MAX_LAYERS];
    float* gw[MLP_MAX_LAYERS];      // may be null
    float* gb[MLP_MAX_LAYERS];
};

// block l < nlayers: gradient (+ SGD step) of layer l; block nlayers: the two loss sums.  1024 threads: thread -> row i = t / 16
// of the [din][dout] kernel and 4 columns j4 .. j4+3; the sample dimension is walked in tiles of 64 staged through LDS.
__global__ __launch_bounds__(1024) void mlp_train_grad_kernel(MlpTrainPtrs q, int nlayers, int nh, const float* __restrict__ xr, int Br,
                                                              const float* __restrict__ xf, int Bf, const float* __restrict__ acts,
                                                              const float* __restrict__ deltas, const float* __restrict__ dlast,
                                                              const float* __restrict__ bce, float lr, float* __restrict__ loss) {
    __shared__ float As[64][65];
    __shared__ __attribute__((aligned(16))) float Ds[64][64];
    const int t = threadIdx.x, Bt = Br + Bf, nhid = nlayers - 1;
    const int l = blockIdx.x;
    if (l == nlayers) {            // d_loss_real, d_loss_fake: fixed-order strided sums + a fixed tree
        float* red = &As[0][0];
        for (int part = 0; part < 2; ++part) {
            const int lo = part ? Br : 0, hi = part ? Bt : Br;
            float s = 0.f;
            for (int i = lo + t; i < hi; i += 1024) s += bce[i];
            red[t] = s;
            __syncthreads();
            for (int w = 512; w > 0; w >>= 1) { if (t < w) red[t] += red[t + w]; __syncthreads(); }
            if (t == 0 && loss) loss[part] = red[0];
            __syncthreads();
        }
        return;
    }
    const int din = l == 0 ? 2 : nh, dout = l == nlayers - 1 ? 1 : nh;
    const int i = t >> 4, j4 = (t & 15) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < Bt; b0 += 64) {
        for (int e = t; e < 4096; e += 1024) {             // stage 64 samples: inputs of the layer and its output gradients
            const int bb = e >> 6, c = e & 63, b = b0 + bb;
            float a = 0.f, d = 0.f;
            if (b < Bt) {
                if (l == 0) { if (c < 2) a = b < Br ? xr[2 * b + c] : xf[2 * (b - Br) + c]; }
                else a = acts[((size_t)b * nhid + (l - 1)) * 64 + c];
                if (l == nlayers - 1) { if (c == 0) d = dlast[b]; }
                else d = deltas[((size_t)b * nhid + l) * 64 + c];
            }
            As[bb][c] = a; Ds[bb][c] = d;
        }
        __syncthreads();
#pragma unroll 8
        for (int bb = 0; bb < 64; ++bb) {
            const float a = As[bb][i];
            const float4 d = *(const float4*)&Ds[bb][j4];
            acc[0] = fmaf(a, d.x, acc[0]); acc[1] = fmaf(a, d.y, acc[1]); acc[2] = fmaf(a, d.z, acc[2]); acc[3] = fmaf(a, d.w, acc[3]);
            accb[0] += d.x; accb[1] += d.y; accb[2] += d.z; accb[3] += d.w;
        }
        __syncthreads();
    }
    for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)      // var -= lr * grad as two roundings (GradientDescentOptimizer's ApplyGradientDescent)
        const int j = j4 + e;
        if (i < din && j < dout) {
            const size_t o = (size_t)i * dout + j;
            if (q.gw[l]) q.gw[l][o] = acc[e];
            if (lr != 0.f) q.w[l][o] = q.w[l][o] - lr * acc[e];
        }
        if (i == 0 && j < dout) {
            if (q.gb[l]) q.gb[l][j] = accb[e];
            if (lr != 0.f) q.b[l][j] = q.b[l][j] - lr * accb[e];
        }
    }
}

static int mlp_fill(MlpParams& p, const float* const* w, const float* const* b, int nlayers, int nh, const char* who) {
    if (nlayers < 2 || nlayers > MLP_MAX_LAYERS || nh < 1 || nh > 64) return cgs_set_error(CGS_EINVAL, "%s: nlayers=%d nhidden=%d (need 2..6, 1..64)", who, nlayers, nh);
    for (int l = 0; l < nlayers; ++l) {
        if (!w[l] || !b[l]) return cgs_set_error(CGS_EINVAL, "%s: null weight", who);
        p.w[l] = w[l]; p.b[l] = b[l];
    }
    p.nlayers = nlayers; p.nh = nh;
    return CGS_OK;
}

static size_t mlp_smem(int nlayers) { return (size_t)(192 + nlayers * 64 + (nlayers - 2) * 8192) * sizeof(float); }

extern "C" {

int cgs_mlp2d_sigmoid_saliency(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x,
                               float* sigmoid, float* saliency, int B, float inv_batch, void* stream) {
    MlpParams p;
    int rc = mlp_fill(p, w, b, nlayers, nhidden, "mlp2d_sigmoid_saliency");
    if (rc) return rc;
    if (B <= 0 || !x || !sigmoid) return cgs_set_error(CGS_EINVAL, "mlp2d_sigmoid_saliency: bad argument");
    const size_t smem = mlp_smem(nlayers);
    CGS_SMEM_ATTR(160 * 1024, "mlp2d_sigmoid_saliency", mlp_saliency_kernel);
    int blocks = (B + 15) / 16; if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(mlp_saliency_kernel, dim3(blocks), dim3(1024), smem, (hipStream_t)stream, p, x, sigmoid, saliency, B, inv_batch);
    CGS_CHECK_LAUNCH("mlp2d_sigmoid_saliency");
    return CGS_OK;
}

static int refine2d_launch(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x, float mean_host,
                           const float* mean_dev, float inv_batch, int steps, float rate, int method, float* best_x, float* best_step,
                           float* traj, int B, void* stream) {
    MlpParams p;
    int rc = mlp_fill(p, w, b, nlayers, nhidden, "refine2d");
    if (rc) return rc;
    if (B <= 0 || steps < 0 || method < 0 || method > 2 || !x || !best_x || !best_step) return cgs_set_error(CGS_EINVAL, "refine2d: bad argument");
    const size_t smem = mlp_smem(nlayers);
    CGS_SMEM_ATTR(160 * 1024, "refine2d", refine2d_kernel);
    int blocks = (B + 15) / 16; if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(refine2d_kernel, dim3(blocks), dim3(1024), smem, (hipStream_t)stream, p, x, mean_host, mean_dev, inv_batch, steps, rate,
                       method, best_x, best_step, traj, B);
    CGS_CHECK_LAUNCH("refine2d");
    return CGS_OK;
}

int cgs_refine2d(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x, float real_sigmoid_mean,
                 float inv_batch, int steps, float rate, int method, float* best_x, float* best_step, float* traj, int B,
                 void* stream) {
    return refine2d_launch(w, b, nlayers, nhidden, x, real_sigmoid_mean, nullptr, inv_batch, steps, rate, method, best_x, best_step, traj, B, stream);
}

int cgs_refine2d_devbase(const float* const* w, const float* const* b, int nlayers, int nhidden, const float* x,
                         const float* real_sigmoid_mean_dev, float inv_batch, int steps, float rate, int method, float* best_x,
                         float* best_step, float* traj, int B, void* stream) {
    if (!real_sigmoid_mean_dev) return cgs_set_error(CGS_EINVAL, "refine2d_devbase: null baseline pointer");
    return refine2d_launch(w, b, nlayers, nhidden, x, 0.f, real_sigmoid_mean_dev, inv_batch, steps, rate, method, best_x, best_step, traj, B, stream);
}

size_t cgs_mlp2d_train_ws_bytes(int B_total, int nlayers) {
    if (B_total <= 0 || nlayers < 2 || nlayers > MLP_MAX_LAYERS) return 0;
    return ((size_t)B_total * (nlayers - 1) * 64 * 2 + (size_t)B_total * 2) * sizeof(float);
}

int cgs_mlp2d_d_step(float* const* w, float* const* b, int nlayers, int nhidden, const float* real, int B_real, const float* fake,
                     int B_fake, float lr, float* const* gw, float* const* gb, float* loss, void* ws, size_t ws_bytes, void* stream) {
    MlpParams p;
    int rc = mlp_fill(p, (const float* const*)w, (const float* const*)b, nlayers, nhidden, "mlp2d_d_step");
    if (rc) return rc;
    if (B_real <= 0 || B_fake <= 0 || !real || !fake) return cgs_set_error(CGS_EINVAL, "mlp2d_d_step: bad argument");
    const int Bt = B_real + B_fake;
    const size_t need = cgs_mlp2d_train_ws_bytes(Bt, nlayers);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "mlp2d_d_step: workspace %zu < %zu bytes", ws_bytes, need);
    float* acts = (float*)ws;
    float* deltas = acts + (size_t)Bt * (nlayers - 1) * 64;
    float* dlast = deltas + (size_t)Bt * (nlayers - 1) * 64;
    float* bce = dlast + Bt;
    const size_t smem = mlp_smem(nlayers);
    CGS_SMEM_ATTR(160 * 1024, "mlp2d_d_step", mlp_train_fwdbwd_kernel);
    for (int part = 0; part < 2; ++part) {      // real rows (target 1) then refined rows (target 0); each loss term is a MEAN
        const int B = part ? B_fake : B_real;
        int blocks = (B + 15) / 16; if (blocks > 256) blocks = 256;
        hipLaunchKernelGGL(mlp_train_fwdbwd_kernel, dim3(blocks), dim3(1024), smem, (hipStream_t)stream, p, part ? fake : real, B,
                           part ? B_real : 0, part ? 0.f : 1.f, 1.f / (float)B, acts, deltas, dlast, bce);
        CGS_CHECK_LAUNCH("mlp_train_fwdbwd");
    }
    MlpTrainPtrs q;
    for (int l = 0; l < MLP_MAX_LAYERS; ++l) {
        q.w[l] = l < nlayers ? w[l] : nullptr; q.b[l] = l < nlayers ? b[l] : nullptr;
        q.gw[l] = (gw && l < nlayers) ? gw[l] : nullptr; q.gb[l] = (gb && l < nlayers) ? gb[l] : nullptr;
    }
    hipLaunchKernelGGL(mlp_train_grad_kernel, dim3(nlayers + 1), dim3(1024), 0, (hipStream_t)stream, q, nlayers, nhidden, real, B_real, fake,
                       B_fake, acts, deltas, dlast, bce, lr, loss);
    CGS_CHECK_LAUNCH("mlp_train_grad");
    return CGS_OK;
}

}  // extern "C"
