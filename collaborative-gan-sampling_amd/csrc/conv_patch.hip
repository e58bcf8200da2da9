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

// Persistent form: a block loads the weight slab and the k -> patch-offset table ONCE and then walks a contiguous run of
// tiles (image-major, so consecutive tiles share their halo rows / columns in L1/L2); the next tile's patch is fetched
// into registers before the MFMA loop of the current one and written to the other LDS buffer after it, and the epilogue
// staging has its own LDS area, so one barrier per tile is all the synchronisation there is.  (The one-tile-per-block
// form spent half of every block's life re-loading the 19 KB weight slab and waiting for its own patch: 142 us for the
// 64x64x3 -> 32x32x64 layer at batch 1024 against an MFMA floor of 65 us.)
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(PatchParams p, int tiles_total, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int patch_f = p.PH * p.pitch;
    float* Bs = smem;                                   // [Kp][PBN]
    float* Ps = smem + (size_t)p.Kp * PBN;               // [2][PH][pitch]
    int* koff_t = (int*)(Ps + (size_t)2 * patch_f);       // [Kp] patch offset of reduction index k
    constexpr int LDE = 32 + 4;
    float* E = (float*)(koff_t + p.Kp) + (threadIdx.x >> 6) * 64 * LDE;     // this wave's [64 rows][32 cols (+4)] staging tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, j = lane & 31;

    const int tiles_x = p.Wout / TC, tiles_y = p.Hout / TR;
    const int nblk_n = p.Np / PBN;
    const int t_begin = blockIdx.x * tiles_per_block;
    const int t_end = t_begin + tiles_per_block < tiles_total ? t_begin + tiles_per_block : tiles_total;
    if (t_begin >= t_end) return;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * (unsigned)p.Cred * 4u), 0x00020000);
    const int rowf = p.PW * p.Cred;                      // valid floats per patch row
    const int rowlen = p.Win * p.Cred;
    // a patch row is one contiguous run of the NHWC image row.  The element -> (patch row, column) split of this thread's
    // loads is the same for every tile: done once
    constexpr int PLD = 10;                               // ceil(max patch floats (21 x 37 x 4 = 3108) / 256) -> checked on the host
    int l_pr[PLD], l_e[PLD];
    {
        const float inv_rowf = 1.0f / (float)rowf;
        const int total = p.PH * rowf;
#pragma unroll
        for (int u = 0; u < PLD; ++u) {
            const int q = tid + 256 * u;
            int pr = (int)((float)q * inv_rowf);          // q / rowf for q < 2^20 up to one off ...
            if (pr * rowf > q) --pr;                       // ... fixed here
            if ((pr + 1) * rowf <= q) ++pr;
            l_pr[u] = q < total ? pr : -1;
            l_e[u] = q - pr * rowf;
        }
    }
    float pv[PLD];
    int cur_n0 = -1;
    // tile id -> (image, tile row, tile col, n-tile); n-tile innermost
#define TILE_DECODE(t_, b_, oy0_, ox0_, n0_)                                                       \
    do {                                                                                           \
        int r_ = (t_);                                                                             \
        n0_ = (r_ % nblk_n) * PBN; r_ /= nblk_n;                                                   \
        ox0_ = (r_ % tiles_x) * TC; r_ /= tiles_x;                                                 \
        oy0_ = (r_ % tiles_y) * TR; b_ = r_ / tiles_y;                                             \
    } while (0)
#define LOAD_PATCH(b_, oy0_, ox0_)                                                                 \
    do {                                                                                           \
        const int iy0_ = p.S * (oy0_) - p.pt, col0_ = (p.S * (ox0_) - p.pl) * p.Cred;              \
        _Pragma("unroll") for (int u = 0; u < PLD; ++u) {                                          \
            const int iy = iy0_ + l_pr[u], col = col0_ + l_e[u];                                   \
            const bool ok = l_pr[u] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)col < (unsigned)rowlen; \
            const unsigned off = ok ? (unsigned)(((b_) * p.Hin + iy) * rowlen + col) * 4u : 0xFFFFFFF0u; \
            pv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off, 0, 0)); \
        }                                                                                          \
    } while (0)
#define STORE_PATCH(buf_)                                                                          \
    do {                                                                                           \
        float* P_ = Ps + (size_t)(buf_) * patch_f;                                                 \
        _Pragma("unroll") for (int u = 0; u < PLD; ++u)                                            \
            if (l_pr[u] >= 0) P_[l_pr[u] * p.pitch + l_e[u]] = pv[u];                              \
    } while (0)

    int b, oy0, ox0, n0;
    TILE_DECODE(t_begin, b, oy0, ox0, n0);
    LOAD_PATCH(b, oy0, ox0);
    {
        const int run = p.kw * p.Cred;                   // contiguous (kx, ci) run of one ky
        for (int k = tid; k < p.Kp; k += 256) {
            const int kk = k < p.K ? k : p.K - 1;        // padded k: its weight row is zero, any valid patch element will do
            const int ky = kk / run;
            koff_t[k] = ky * p.pitch + (kk - ky * run);
        }
    }
    STORE_PATCH(0);

    // rowbase of this lane's two GEMM rows (pixels) inside the patch
    int rowbase[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = wm * 64 + tm * 32 + j;
        rowbase[tm] = (p.S * (m / TC)) * p.pitch + (p.S * (m % TC)) * p.Cred;
    }
    const float* bcol = Bs + wn * 32 + j;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;

    for (int t = t_begin, it = 0; t < t_end; ++t, ++it) {
        if (n0 != cur_n0) {                              // (re)load the [Kp][PBN] weight slab of this n-tile: once per block when N <= 64
            if (it > 0) __syncthreads();                 // every wave is done reading the old slab
            for (int q = tid; q < p.Kp * (PBN / 4); q += 256) {
                const int k = q / (PBN / 4), c = q - k * (PBN / 4);
                ((f32x4*)Bs)[q] = *(const f32x4*)(p.wk + (size_t)k * p.Np + n0 + c * 4);
            }
            cur_n0 = n0;
        }
        __syncthreads();                                 // patch[it & 1] (and the weights) are in LDS; patch[(it+1) & 1] is free
        int nb_, noy0, nox0, nn0;
        const bool more = t + 1 < t_end;
        if (more) {
            TILE_DECODE(t + 1, nb_, noy0, nox0, nn0);
            LOAD_PATCH(nb_, noy0, nox0);
        }
        const float* P = Ps + (size_t)(it & 1) * patch_f;
        f32x16 acc[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][r] = 0.f;
#pragma unroll 4
        for (int s = 0; s < p.Kp / 2; ++s) {
            const int k = 2 * s + h;
            const float bv = bcol[k * PBN];
            const int koff = koff_t[k];
            const float a0 = P[rowbase[0] + koff], a1 = P[rowbase[1] + koff];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
        }
        if (more) STORE_PATCH((it + 1) & 1);

        // epilogue through this wave's own staging tile (the same wave wrote it last tile and has finished reading it)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) E[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + j] = acc[tm][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = n0 + wn * 32 + c4;
        if (n < p.N) {
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next tile's staging writes must not pass these reads
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (more) { b = nb_; oy0 = noy0; ox0 = nox0; n0 = nn0; }
    }
#undef TILE_DECODE
#undef LOAD_PATCH
#undef STORE_PATCH
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
    const size_t smem = ((size_t)p.Kp * PBN + (size_t)2 * p.PH * p.pitch + p.Kp + (size_t)4 * 64 * 36) * sizeof(float);
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)conv_patch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        done = true;
    }
    if (smem > 96 * 1024 || p.PH * p.PW * p.Cred > 10 * 256) return cgs_set_error(CGS_EINVAL, "conv_patch: patch too large");
    const long tiles = (long)B * (p.Hout / TR) * (p.Wout / TC) * (p.Np / PBN);
    if (tiles == 0) return CGS_OK;
    if (tiles > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_patch: grid too large");
    // persistent blocks: two per CU, each a contiguous run of tiles (>= 4, so the weight slab load is amortised);
    // small launches keep one tile per block and simply fill the GPU
    long per = (tiles + 511) / 512;
    if (per < 4) per = tiles >= 4 * 256 ? 4 : 1;
    const long blocks = (tiles + per - 1) / per;
    hipLaunchKernelGGL(conv_patch_kernel, dim3((unsigned)blocks), dim3(256), smem, s, p, (int)tiles, (int)per);
    CGS_CHECK_LAUNCH("conv_patch");
    cgs_note_kernel("conv_patch_kernel");
    return CGS_OK;
}
