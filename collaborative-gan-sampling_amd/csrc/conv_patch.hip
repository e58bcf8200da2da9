// Strided 'SAME' convolution whose INPUT has a handful of channels (<= 4): the first discriminator conv
// (3 -> 64, nsgan/GAN.py:64 / DCGAN d_h0_conv) and the backward-data of the last generator deconv (3 -> 64 too:
// a strided conv of the image gradient with the deconv weights, sampling/collaborator.py:31).
//
// K = kh*kw*Cin is tiny (75) and a K chunk spans several taps, so the generic implicit GEMM has to gather A element
// by element from global memory (16 dword loads per thread and tile).  Here a block owns an 8 x 16 tile of output
// pixels of one image: the (2*8+kh-2) x (2*16+kw-2) x Cin input patch (8 KB) and the whole [K][BN] weight slab are
// staged once in LDS, and every A fragment of v_mfma_f32_32x32x2_f32 is one ds_read_b32 at
//   patch[(2*oy+ky)*pitch + (2*ox+kx)*Cin + ci] = rowbase(pixel) + koff(k),  koff(k) = (k / (kw*Cin))*pitch + k % (kw*Cin)
// (for a fixed ky the (kx,ci) run is contiguous in NHWC).  Output through the same LDS-transposed 16-byte epilogue.
#include "cgs_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PatchParams {
    const float* in;      // [B,Hin,Win,Cred]
    const float* wk;      // [Kp][Np]  (k-major rows; rows >= K are zero)
    const float* bias;
    const float* ep_a;
    const float* ep_b;
    const float* ep_aux;
    float* out;           // [B,Hout,Wout,N]
    int B, Hin, Win, Cred, Hout, Wout, N, Np;
    int kh, kw, pt, pl, K, Kp;
    int S;                // stride (1 or 2)
    int PH, PW, pitch;    // patch rows, pixels per row, floats per LDS patch row
    int epilogue;
};

static constexpr int TR = 8, TC = 16;      // output tile (rows x cols) = 128 GEMM rows
static constexpr int PBN = 64;             // output channels per block

__global__ void pack_patch_weights_kernel(const float* __restrict__ w, float* __restrict__ wk, int K, int Kp, int N, int Np) {
    const int total = Kp * Np;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i / Np, n = i - k * Np;
        wk[i] = (k < K && n < N) ? w[(size_t)k * N + n] : 0.f;     // w[kh][kw][Cb][Cs] is already [K][N] row-major
    }
}

template <int EPI>
__device__ __forceinline__ void patch_rows(const PatchParams& p, const float* E, int b, int oy0, int ox0, int wm, int n, int rsub,
                                           int c4, f32x4 bias, f32x4 ea, f32x4 eb) {
    constexpr int LDE = 32 + 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {                   // 64 rows per wave, 8 rows per pass (8 lanes x float4 per row)
        const int lrow = it * 8 + rsub;
        const int m = wm * 64 + lrow;
        const int oy = oy0 + m / TC, ox = ox0 + m % TC;
        const size_t o = ((size_t)(b * p.Hout + oy) * p.Wout + ox) * p.N + n;
        const f32x4 v = *(const f32x4*)(E + lrow * LDE + c4);
        f32x4 aux = {0.f, 0.f, 0.f, 0.f};
        if (EPI >= CGS_EPI_RELU_BWD_AFFINE) aux = *(const f32x4*)(p.ep_aux + o);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = epilogue_apply(v[e] + bias[e], EPI, ea[e], eb[e], aux[e]);
        *(f32x4*)(p.out + o) = y;
    }
}

__global__ __launch_bounds__(256, 2) void conv_patch_kernel(PatchParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                                   // [Kp][PBN]
    float* Ps = smem + (size_t)p.Kp * PBN;               // [PH][pitch]
    int* koff_t = (int*)(Ps + (size_t)p.PH * p.pitch);    // [Kp] patch offset of reduction index k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, j = lane & 31;

    const int tiles_x = p.Wout / TC, tiles_y = p.Hout / TR;
    const int nblk_n = p.Np / PBN;
    int t = blockIdx.x;
    const int nb = t % nblk_n; t /= nblk_n;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * TR, ox0 = tx * TC, n0 = nb * PBN;

    // weights: [Kp][PBN] slab of this n-tile
    for (int q = tid; q < p.Kp * (PBN / 4); q += 256) {
        const int k = q / (PBN / 4), c = q - k * (PBN / 4);
        ((f32x4*)Bs)[q] = *(const f32x4*)(p.wk + (size_t)k * p.Np + n0 + c * 4);
    }
    // input patch, zero padded (hardware bounds check on an out-of-range offset)
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * (unsigned)p.Cred * 4u), 0x00020000);
    const int rowf = p.PW * p.Cred;                      // valid floats per patch row
    const int iy0 = p.S * oy0 - p.pt, ix0 = p.S * ox0 - p.pl;
    {   // a patch row is one contiguous run of the NHWC image row: range-check the flattened column; all of a
        // thread's loads are issued before the first LDS store so they are in flight together
        const int rowlen = p.Win * p.Cred, col0 = ix0 * p.Cred;
        const float inv_rowf = 1.0f / (float)rowf;
        const int total = p.PH * rowf;
        constexpr int PLD = 10;                           // ceil(max patch floats (21 x 37 x 4 = 3108) / 256) -> checked on the host
        float v[PLD];
        int dst[PLD];
#pragma unroll
        for (int u = 0; u < PLD; ++u) {
            const int q = tid + 256 * u;
            int pr = (int)((float)q * inv_rowf);          // q / rowf for q < 2^20 up to one off ...
            if (pr * rowf > q) --pr;                       // ... fixed here
            if ((pr + 1) * rowf <= q) ++pr;
            const int e = q - pr * rowf;
            const int iy = iy0 + pr, col = col0 + e;
            const bool ok = q < total && (unsigned)iy < (unsigned)p.Hin && (unsigned)col < (unsigned)rowlen;
            const unsigned off = ok ? (unsigned)((b * p.Hin + iy) * rowlen + col) * 4u : 0xFFFFFFF0u;
            v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off, 0, 0));
            dst[u] = q < total ? pr * p.pitch + e : -1;
        }
#pragma unroll
        for (int u = 0; u < PLD; ++u)
            if (dst[u] >= 0) Ps[dst[u]] = v[u];
    }
    {
        const int run = p.kw * p.Cred;                   // contiguous (kx, ci) run of one ky
        for (int k = tid; k < p.Kp; k += 256) {
            const int kk = k < p.K ? k : p.K - 1;        // padded k: its weight row is zero, any valid patch element will do
            const int ky = kk / run;
            koff_t[k] = ky * p.pitch + (kk - ky * run);
        }
    }
    __syncthreads();

    // rowbase of this lane's two GEMM rows (pixels) inside the patch
    int rowbase[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = wm * 64 + tm * 32 + j;
        rowbase[tm] = (p.S * (m / TC)) * p.pitch + (p.S * (m % TC)) * p.Cred;
    }
    f32x16 acc[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
    const float* bcol = Bs + wn * 32 + j;
#pragma unroll 4
    for (int s = 0; s < p.Kp / 2; ++s) {
        const int k = 2 * s + h;
        const float bv = bcol[k * PBN];
        const int koff = koff_t[k];
        const float a0 = Ps[rowbase[0] + koff], a1 = Ps[rowbase[1] + koff];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
    }
    __syncthreads();                                     // patch / weights dead: reuse the LDS for the epilogue staging

    constexpr int LDE = 32 + 4;
    float* E = smem + wave * 64 * LDE;                   // [64 rows][32 cols (+4)]
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) E[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + j] = acc[tm][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const int n = n0 + wn * 32 + c4;
    if (n >= p.N) return;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, ea = {1.f, 1.f, 1.f, 1.f}, eb = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias = *(const f32x4*)(p.bias + n);
    if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = *(const f32x4*)(p.ep_a + n); eb = *(const f32x4*)(p.ep_b + n); }
    if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = *(const f32x4*)(p.ep_a + n);
    switch (p.epilogue) {
        case CGS_EPI_NONE: patch_rows<CGS_EPI_NONE>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
        case CGS_EPI_LRELU: patch_rows<CGS_EPI_LRELU>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
        case CGS_EPI_AFFINE_RELU: patch_rows<CGS_EPI_AFFINE_RELU>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
        case CGS_EPI_TANH: patch_rows<CGS_EPI_TANH>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
        case CGS_EPI_RELU_BWD_AFFINE: patch_rows<CGS_EPI_RELU_BWD_AFFINE>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
        case CGS_EPI_LRELU_BWD: patch_rows<CGS_EPI_LRELU_BWD>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
        default: patch_rows<CGS_EPI_TANH_BWD>(p, E, b, oy0, ox0, wm, n, rsub, c4, bias, ea, eb); break;
    }
}

// dirT = false: the conv itself (big -> small, Cb <= 4 input channels, stride 1 or 2).
// dirT = true : the backward-data of a STRIDE-1 conv whose output has <= 4 channels (small -> big at the same resolution):
//               dx[i] = sum_ky dy[i + pt - ky] w[ky] = sum_ky' dy[i - (kh-1-pt) + ky'] w[kh-1-ky'] -- the same stride-1
//               correlation with the taps flipped, the padding mirrored and the weight matrix transposed (pack kernel).
static void patch_geom(const CgsLayer& L, bool dirT, PatchParams& p) {
    const int S = dirT ? 1 : L.sh;
    const int pt = cgs_same_pad_before(L.Hb, L.kh, L.sh), pl = cgs_same_pad_before(L.Wb, L.kw, L.sw);
    if (!dirT) { p.Hin = L.Hb; p.Win = L.Wb; p.Cred = L.Cb; p.Hout = L.Hs; p.Wout = L.Ws; p.N = L.Cs; p.pt = pt; p.pl = pl; }
    else { p.Hin = L.Hs; p.Win = L.Ws; p.Cred = L.Cs; p.Hout = L.Hb; p.Wout = L.Wb; p.N = L.Cb; p.pt = L.kh - 1 - pt; p.pl = L.kw - 1 - pl; }
    p.Np = cgs_round_up(p.N, PBN);
    p.kh = L.kh; p.kw = L.kw; p.S = S;
    p.K = L.kh * L.kw * p.Cred; p.Kp = cgs_round_up(p.K, 2);
    p.PH = S * (TR - 1) + L.kh; p.PW = S * (TC - 1) + L.kw;
    p.pitch = p.PW * p.Cred + 1;
}

// <= 4 reduction channels, output tiles of 8 x 16 pixels, N % 4 == 0
int cgs_conv_patch_ok(const CgsLayer& L, int epilogue) {
    (void)epilogue;
    return L.Cb <= 4 && L.sh == L.sw && (L.sh == 1 || L.sh == 2) && (L.Hs % TR) == 0 && (L.Ws % TC) == 0 && (L.Cs % 4) == 0 &&
           L.kh * L.kw * L.Cb <= 160 && L.kh <= 7 && L.kw <= 7 &&
           (L.sh * (TR - 1) + L.kh) * (L.sh * (TC - 1) + L.kw) * L.Cb <= 10 * 256;
}

int cgs_conv_patch_T_ok(const CgsLayer& L) {
    return L.Cs <= 4 && L.sh == 1 && L.sw == 1 && (L.Hb % TR) == 0 && (L.Wb % TC) == 0 && (L.Cb % 4) == 0 &&
           L.kh * L.kw * L.Cs <= 160 && L.kh <= 7 && L.kw <= 7 && ((TR - 1) + L.kh) * ((TC - 1) + L.kw) * L.Cs <= 10 * 256;
}

size_t cgs_conv_patch_ws_floats(const CgsLayer& L, bool dirT) {
    return dirT ? (size_t)cgs_round_up(L.kh * L.kw * L.Cs, 2) * cgs_round_up(L.Cb, PBN)
                : (size_t)cgs_round_up(L.kh * L.kw * L.Cb, 2) * cgs_round_up(L.Cs, PBN);
}

// dirT: wk[(ky', kx', c)][n] = w[kh-1-ky'][kw-1-kx'][n][c]   (w is [kh][kw][Cb = n][Cs = c])
__global__ void pack_patch_weights_T_kernel(const float* __restrict__ w, float* __restrict__ wk, int kh, int kw, int Cb, int Cs,
                                            int K, int Kp, int Np) {
    const int total = Kp * Np;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i / Np, n = i - k * Np;
        float v = 0.f;
        if (k < K && n < Cb) {
            const int c = k % Cs, t = k / Cs, kx = t % kw, ky = t / kw;
            v = w[(((size_t)(kh - 1 - ky) * kw + (kw - 1 - kx)) * Cb + n) * Cs + c];
        }
        wk[i] = v;
    }
}

int cgs_conv_patch_launch(const CgsLayer& L, bool dirT, int B, const float* in, const float* w, const float* bias, float* out,
                          int epilogue, const float* ep_a, const float* ep_b, const float* ep_aux, float* ws, size_t ws_bytes,
                          int prepacked, hipStream_t s) {
    PatchParams p;
    patch_geom(L, dirT, p);
    p.in = in; p.wk = ws; p.bias = bias; p.ep_a = ep_a; p.ep_b = ep_b; p.ep_aux = ep_aux; p.out = out; p.B = B; p.epilogue = epilogue;
    const size_t need = cgs_conv_patch_ws_floats(L, dirT) * sizeof(float);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "conv_patch: workspace %zu < %zu bytes", ws_bytes, need);
    if ((long)B * p.Hin * p.Win * p.Cred * 4 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_patch: input exceeds 2 GiB");
    if (!prepacked) {
        if (dirT)
            hipLaunchKernelGGL(pack_patch_weights_T_kernel, dim3(cgs_ceil_div(p.Kp * p.Np, 256)), dim3(256), 0, s, w, ws, L.kh, L.kw, L.Cb,
                               L.Cs, p.K, p.Kp, p.Np);
        else
            hipLaunchKernelGGL(pack_patch_weights_kernel, dim3(cgs_ceil_div(p.Kp * p.Np, 256)), dim3(256), 0, s, w, ws, p.K, p.Kp, p.N, p.Np);
        CGS_CHECK_LAUNCH("pack_patch_weights");
    }
    size_t kloop = ((size_t)p.Kp * PBN + (size_t)p.PH * p.pitch + p.Kp) * sizeof(float);
    const size_t stage = (size_t)4 * 64 * 36 * sizeof(float);
    const size_t smem = kloop > stage ? kloop : stage;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)conv_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        done = true;
    }
    if (smem > 96 * 1024 || p.PH * p.PW * p.Cred > 10 * 256) return cgs_set_error(CGS_EINVAL, "conv_patch: patch too large");
    const long blocks = (long)B * (p.Hout / TR) * (p.Wout / TC) * (p.Np / PBN);
    if (blocks == 0) return CGS_OK;
    if (blocks > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_patch: grid too large");
    hipLaunchKernelGGL(conv_patch_kernel, dim3((unsigned)blocks), dim3(256), smem, s, p);
    CGS_CHECK_LAUNCH("conv_patch");
    cgs_note_kernel("conv_patch_kernel");
    return CGS_OK;
}
