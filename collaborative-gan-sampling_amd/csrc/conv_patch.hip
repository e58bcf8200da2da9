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
#include <stdlib.h>

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
    const unsigned* aux_signs;   // sign mask of the aux tensor (cgs_hip.h, "sign masks") instead of ep_aux: 1 bit per element
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

// one 32-row half (row tile tm) of the wave's 64 rows: the staging tile holds 32 rows, so a 7x7x3 layer's 38 KB weight slab leaves room
// for two blocks per CU (the 64-row staging made it 83 KB per block: one block, four waves per CU)
template <int EPI>
__device__ __forceinline__ void patch_rows(const PatchParams& p, const float* E, int b, int oy0, int ox0, int wm, int tm, int n, int rsub,
                                           int c4, f32x4 bias, f32x4 ea, f32x4 eb) {
    constexpr int LDE = 32 + 4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {                   // 32 rows, 8 rows per pass (8 lanes x float4 per row)
        const int lrow = it * 8 + rsub;
        const int m = wm * 64 + tm * 32 + lrow;
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
// FK > 0: the geometry of the reduction is fixed at compile time (K = FK = kh * kw * Cred, a tap row's run FRUN = kw * Cred, the
// LDS patch pitch FPITCH): the lane's weight column lives in registers for the whole persistent block and every patch address is
// one of two lane bases + an instruction immediate -- one ds_read_b32 per MFMA and nothing else in the loop, where the general
// form (FK = 0) issues a weight read, a table look-up, two patch reads and an address add per MFMA pair (config 5's 7x7x3 stride-1
// layers and its 4x4x3 stride-2 PatchGAN stem).
template <int FK, int FRUN, int FPITCH>
__global__ __launch_bounds__(256, 2) void conv_patch_kernel(PatchParams p, int tiles_total, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int patch_f = p.PH * p.pitch;
    float* Bs = smem;                                   // [Kp][PBN]
    float* Ps = smem + (size_t)p.Kp * PBN;               // [2][PH][pitch]
    int* koff_t = (int*)(Ps + (size_t)2 * patch_f);       // [Kp] patch offset of reduction index k
    constexpr int LDE = 32 + 4;
    float* E = (float*)(koff_t + p.Kp) + (threadIdx.x >> 6) * 32 * LDE;     // this wave's [32 rows][32 cols (+4)] staging tile (half its rows at a time)
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
    constexpr int FKP = (FK + 1) & ~1, FNS = FKP / 2;
    constexpr bool WREG = FK > 0 && FNS <= 32;           // few steps: the lane's weight column in registers; else from the LDS slab at lane base + immediate
    float bw[WREG ? FNS : 1];                            // this lane's weight column (k = 2 s + h), all steps
    const float* bcol_h = bcol + h * PBN;
    // FK > 0: k = 2 s + h sits at patch offset koff(2 s) + h, except where 2 s + 1 opens the next tap row: a second lane base
    int rb1[2], rb2[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) { rb1[tm] = rowbase[tm] + h; rb2[tm] = rowbase[tm] + h * (1 + FPITCH - FRUN); }

    for (int t = t_begin, it = 0; t < t_end; ++t, ++it) {
        if (n0 != cur_n0) {                              // (re)load the [Kp][PBN] weight slab of this n-tile: once per block when N <= 64
            if constexpr (WREG) {
#pragma unroll
                for (int q = 0; q < FNS; ++q) bw[q] = p.wk[(size_t)(2 * q + h) * p.Np + n0 + wn * 32 + j];
            } else {
                if (it > 0) __syncthreads();             // every wave is done reading the old slab
                for (int q = tid; q < p.Kp * (PBN / 4); q += 256) {
                    const int k = q / (PBN / 4), c = q - k * (PBN / 4);
                    ((f32x4*)Bs)[q] = *(const f32x4*)(p.wk + (size_t)k * p.Np + n0 + c * 4);
                }
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
        if constexpr (FK > 0) {
#pragma unroll
            for (int s = 0; s < FNS; ++s) {
                const int k0 = 2 * s < FK ? 2 * s : FK - 1;                       // (a padded k: zero weight, any valid patch element)
                const int k1 = 2 * s + 1 < FK ? 2 * s + 1 : FK - 1;
                const int ko0 = (k0 / FRUN) * FPITCH + k0 % FRUN;               // compile-time after unrolling
                const int ko1 = (k1 / FRUN) * FPITCH + k1 % FRUN;
                // lanes h = 0 read ko0; lanes h = 1 read ko1 = ko0 + 1 (same tap row), ko0 + 1 + FPITCH - FRUN (next row), or ko0 (padded)
                const bool same = ko1 == ko0 + 1, pad = ko1 == ko0;
                const float a0 = pad ? P[rowbase[0] + ko0] : same ? P[rb1[0] + ko0] : P[rb2[0] + ko0];
                const float a1 = pad ? P[rowbase[1] + ko0] : same ? P[rb1[1] + ko0] : P[rb2[1] + ko0];
                const float bv = WREG ? bw[WREG ? s : 0] : bcol_h[2 * s * PBN];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
                if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);             // (bound the scheduler's look-ahead: it hoists every patch read otherwise)
            }
        } else {
#pragma unroll 4
        for (int s = 0; s < p.Kp / 2; ++s) {
            const int k = 2 * s + h;
            const float bv = bcol[k * PBN];
            const int koff = koff_t[k];
            const float a0 = P[rowbase[0] + koff], a1 = P[rowbase[1] + koff];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
        }
        }
        if (more) STORE_PATCH((it + 1) & 1);

        // epilogue through this wave's own staging tile, one 32-row half at a time (the same wave wrote it last and has finished reading it)
        const int n = n0 + wn * 32 + c4;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, ea = {1.f, 1.f, 1.f, 1.f}, eb = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) {
            if (p.bias) bias = *(const f32x4*)(p.bias + n);
            if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = *(const f32x4*)(p.ep_a + n); eb = *(const f32x4*)(p.ep_b + n); }
            if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = *(const f32x4*)(p.ep_a + n);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) E[((r & 3) + 8 * (r >> 2) + 4 * h) * LDE + j] = acc[tm][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (n < p.N) {
                switch (p.epilogue) {
                    case CGS_EPI_NONE: patch_rows<CGS_EPI_NONE>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                    case CGS_EPI_LRELU: patch_rows<CGS_EPI_LRELU>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                    case CGS_EPI_AFFINE_RELU: patch_rows<CGS_EPI_AFFINE_RELU>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                    case CGS_EPI_TANH: patch_rows<CGS_EPI_TANH>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                    case CGS_EPI_RELU_BWD_AFFINE: patch_rows<CGS_EPI_RELU_BWD_AFFINE>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                    case CGS_EPI_LRELU_BWD: patch_rows<CGS_EPI_LRELU_BWD>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                    default: patch_rows<CGS_EPI_TANH_BWD>(p, E, b, oy0, ox0, wm, tm, n, rsub, c4, bias, ea, eb); break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next staging writes must not pass these reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (more) { b = nb_; oy0 = noy0; ox0 = nox0; n0 = nn0; }
    }
#undef TILE_DECODE
#undef LOAD_PATCH
#undef STORE_PATCH
}

// ------------------------------------------------------------------------------------------------------------------
// Second form for the hot shapes (stride 2, kh = KH, a (kx, ci) run of R <= RP floats per tap row, e.g. 5x5x3: KH = 5, R = 15,
// RP = 16).  The kernel above issues two ds_read_b32 and one address add per MFMA: it is bound by LDS-instruction issue and by
// the vector ALU, which shares the SIMD's issue with the matrix pipe (tools/probe/mfma_probe.hip), not by the matrix pipe.
// Here the reduction is re-ordered so that a lane reads CONSECUTIVE patch floats: every tap row is padded to RP elements and
// split in two halves, lane-half h (the k index of v_mfma_f32_32x32x2_f32) takes half h, so the RP/2 MFMA steps of a tap row
// consume RP/2 consecutive floats per lane = RP/4 ds_read_b64 (8-byte aligned: stride-2 pixels of 3 channels are 24 bytes
// apart, the row pitch is even), and the lane's weight column sits in REGISTERS for the whole (persistent) block:
// 0.5 LDS reads per MFMA instead of 2, no table look-ups.  Padding elements (run index >= R) carry zero weights AND are
// zeroed on the A side, so a non-finite neighbour outside the receptive field cannot leak in.
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void pack_patch2_weights_kernel(const float* __restrict__ w, float* __restrict__ wk, int KH, int RP, int R, int N, int Np) {
    const int HALF = RP / 2, total = KH * RP * Np;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i % Np, th = i / Np, h = th & 1, t = th >> 1;      // wk[(t * 2 + h) * Np + n]
        const int ky = t / HALF, sidx = t - ky * HALF, e = h * HALF + sidx;
        wk[i] = (e < R && n < N) ? w[((size_t)ky * R + e) * N + n] : 0.f;
    }
}

// epilogue of conv_patch2_kernel for one tile: acc[tm][r] is output row (r & 3) + 8 * (r >> 2) + 4 * h of row tile tm (32 GEMM rows =
// 2 tile rows of 16 pixels), channel j.  base = byte offset of (tile origin, this lane's channel, + 4 * h pixels); 32-bit offsets through a
// buffer descriptor, the per-register part folds into the store's immediate offset.
template <int EPI>
__device__ __forceinline__ void patch2_store(const PatchParams& p, __amdgpu_buffer_rsrc_t out_rsrc, __amdgpu_buffer_rsrc_t aux_rsrc,
                                             const f32x16 (&acc)[2], unsigned base, unsigned rowstride, float ea, float eb) {
    const unsigned pixb = (unsigned)p.N * 4u;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);                       // + 4 * h is in the lane base; row / 16 picks the tile row
            const unsigned o = base + (unsigned)(2 * tm + row / TC) * rowstride + (unsigned)(row % TC) * pixb;
            float aux = 0.f;
            if (EPI >= CGS_EPI_RELU_BWD_AFFINE) aux = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(aux_rsrc, o, 0, 0));
            const float y = epilogue_apply(acc[tm][r], EPI, ea, eb, aux);       // (the bias is already in the accumulator)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), out_rsrc, o, 0, 0);
        }
}

// the same store with relu' / lrelu' taken from the sign mask: nw = lane L's un-shuffled word of pixel L of the wave's 4 x 16
// pixels.  Register r of row tile tm is pixel 32 tm + (r & 3) + 8 (r >> 2) in lanes 0-31 and that + 4 in lanes 32-63, and a
// lane's channel is its index in the half: {readlane(nw, pixel), readlane(nw, pixel + 4)} IS the 64-bit lane mask of "aux > 0".
template <int EPI>
__device__ __forceinline__ void patch2_store_signs(const PatchParams& p, __amdgpu_buffer_rsrc_t out_rsrc, const f32x16 (&acc)[2],
                                                   unsigned base, unsigned rowstride, float ea, unsigned nw) {
    const unsigned pixb = (unsigned)p.N * 4u;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2);
            const unsigned o = base + (unsigned)(2 * tm + row / TC) * rowstride + (unsigned)(row % TC) * pixb;
            const unsigned lo = __builtin_amdgcn_readlane(nw, 32 * tm + row), hi = __builtin_amdgcn_readlane(nw, 32 * tm + row + 4);
            const bool pos = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
            const float v = acc[tm][r];
            const float y = EPI == CGS_EPI_RELU_BWD_AFFINE ? (pos ? v * ea : 0.f) : (pos ? v : 0.2f * v);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), out_rsrc, o, 0, 0);
        }
}

// AUXM = 1: the *_BWD epilogues read an aux tensor of the output's shape (the saved activation of the layer below).  Loading it in the
// epilogue put a chain of global-load latencies into every tile (221 -> 262 us with one dword per register); here its 8 float4
// per lane are PREFETCHED right after the tile's barrier, in the 16-byte-per-lane layout, and the accumulators are transposed to
// that layout through a small LDS tile when the MFMAs are done.
// AUXM = 2: relu' / lrelu' need the SIGN of the saved activation only: the producing forward epilogue left a bitmask
// (p.aux_signs, one 32-bit word per pixel and 32 channels, a plane per channel group: 8 MB instead of the 268 MB fp32 tensor of
// the 32x32x64 layer at batch 1024).  A wave's 4 tile rows x 16 pixels are 64 words = ONE coalesced dword load per tile, lane L
// holding pixel L's word, prefetched after the tile's barrier; after the MFMAs each lane un-shuffles its word to "bit c = channel
// c" (31 VALU ops per tile for the whole wave), and for accumulator register r the two pixels the wave's lane halves hold come
// out with two v_readlane straight into a 64-bit lane mask: the epilogue stays in the ACCUMULATOR layout of the plain kernel
// (no LDS transpose, one v_cndmask per element) and three blocks fit a CU again.
template <int KH, int RP, int R, int PLD, int AUXM>     // R = kw * Cin floats per tap row (<= RP), PLD = patch floats per thread (ceil(PH * PW * Cin / 256))
__global__ __launch_bounds__(256, AUXM == 1 ? 2 : 4) void conv_patch2_kernel(PatchParams p, int tiles_total, int tiles_per_block) {
    constexpr bool AUXP = AUXM != 0;
    constexpr int HALF = RP / 2, NQ = RP / 4, NS = KH * HALF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int patch_f = p.PH * p.pitch;
    float* Ps = smem;                                    // [2][PH][pitch]
    constexpr int LDE = 32 + 4;
    float* E = Ps + (size_t)2 * patch_f + (threadIdx.x >> 6) * 32 * LDE;     // AUXP: this wave's [32 rows][32 cols (+4)] staging tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, j = lane & 31;
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;     // AUXP epilogue layout: 8 lanes x float4 per row, 8 rows per pass

    const int tiles_x = p.Wout / TC, tiles_y = p.Hout / TR;
    const int tiles_per_n = p.B * tiles_x * tiles_y;     // tile id = n-tile * tiles_per_n + (image, tile row, tile col): a block's run
    const int t_begin = blockIdx.x * tiles_per_block;    // of tiles stays inside one n-tile except at a boundary (weights reloaded there)
    const int t_end = t_begin + tiles_per_block < tiles_total ? t_begin + tiles_per_block : tiles_total;
    if (t_begin >= t_end) return;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * (unsigned)p.Cred * 4u), 0x00020000);
    const unsigned out_bytes = (unsigned)p.B * (unsigned)p.Hout * (unsigned)p.Wout * (unsigned)p.N * 4u;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t aux_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ep_aux ? p.ep_aux : p.out), 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sg_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.aux_signs ? (const void*)p.aux_signs : (const void*)p.out), 0,
                                                                             (int)(out_bytes >> 5), 0x00020000);     // one word per 32 floats
    const unsigned sg_plane_b = (unsigned)p.B * (unsigned)p.Hout * (unsigned)p.Wout * 4u;                            // bytes of one 32-channel plane
    const int rowf = p.PW * p.Cred;                      // valid floats per patch row
    const int rowlen = p.Win * p.Cred;
    // this thread's patch elements: (patch row, column) -> LDS offset and image offset relative to the tile's patch origin (loop invariant)
    // Which of them fall outside the image depends on the tile only through "first / last tile row / column": four static flags per
    // element, eight elements in one register (the rows and columns as separate registers were 16 VGPRs of the 21 that kept a fourth
    // block off the CU).  An element past the patch (the last thread rows) carries all four flags AND no LDS slot.
    int l_dst[PLD], l_src[PLD];
    unsigned l_flags = 0;
    {
        const float inv_rowf = 1.0f / (float)rowf;
        const int total = p.PH * rowf;
        const int row_lo = p.pt, row_hi = p.Hin - (p.S * (p.Hout - TR) - p.pt);            // patch rows < row_lo lie above the image in a first tile row, >= row_hi below it in a last one
        const int col_lo = p.pl * p.Cred, col_hi = rowlen - (p.S * (p.Wout - TC) - p.pl) * p.Cred;
#pragma unroll
        for (int u = 0; u < PLD; ++u) {
            const int q = tid + 256 * u;
            int pr = (int)((float)q * inv_rowf);
            if (pr * rowf > q) --pr;
            if ((pr + 1) * rowf <= q) ++pr;
            const int e = q - pr * rowf;
            l_dst[u] = q < total ? pr * p.pitch + e : -1;
            l_src[u] = (pr * rowlen + e) * 4;
            const unsigned f = q < total ? (unsigned)(pr < row_lo) | ((unsigned)(pr >= row_hi) << 1) | ((unsigned)(e < col_lo) << 2) | ((unsigned)(e >= col_hi) << 3) : 15u;
            l_flags |= f << (4 * u);
        }
    }
    // the pad floats behind a patch row's rowf valid ones are read by the widened runs: keep them finite (zero) in both buffers
    for (int q = tid; q < 2 * p.PH * (p.pitch - rowf); q += 256) {
        const int r = q / (p.pitch - rowf), cidx = q - r * (p.pitch - rowf);
        Ps[(size_t)r * p.pitch + rowf + cidx] = 0.f;
    }
    float pv[PLD];
#define TILE_DECODE(t_, b_, oy0_, ox0_, n0_)                                                       \
    do {                                                                                           \
        int r_ = (t_);                                                                             \
        n0_ = (r_ / tiles_per_n) * PBN; r_ -= (r_ / tiles_per_n) * tiles_per_n;                    \
        ox0_ = (r_ % tiles_x) * TC; r_ /= tiles_x;                                                 \
        oy0_ = (r_ % tiles_y) * TR; b_ = r_ / tiles_y;                                             \
    } while (0)
#define LOAD_PATCH(b_, oy0_, ox0_)                                                                 \
    do {                                                                                           \
        const int iy0_ = p.S * (oy0_) - p.pt, col0_ = (p.S * (ox0_) - p.pl) * p.Cred;              \
        const int org_ = (((b_) * p.Hin + iy0_) * rowlen + col0_) * 4;        /* scalar */          \
        const unsigned edge_ = ((oy0_) == 0 ? 0x11111111u : 0u) | ((oy0_) == p.Hout - TR ? 0x22222222u : 0u) |       /* scalar */ \
                               ((ox0_) == 0 ? 0x44444444u : 0u) | ((ox0_) == p.Wout - TC ? 0x88888888u : 0u) | 0u;   \
        const unsigned bad_ = (l_flags & edge_) | (l_flags & (l_flags >> 1) & (l_flags >> 2) & (l_flags >> 3) & 0x11111111u);   /* (all four flags: no element) */ \
        _Pragma("unroll") for (int u = 0; u < PLD; ++u) {                                          \
            const unsigned off = ((bad_ >> (4 * u)) & 15u) == 0u ? (unsigned)(org_ + l_src[u]) : 0xFFFFFFF0u; \
            pv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off, 0, 0)); \
        }                                                                                          \
    } while (0)
#define STORE_PATCH(buf_)                                                                          \
    do {                                                                                           \
        float* P_ = Ps + (size_t)(buf_) * patch_f;                                                 \
        _Pragma("unroll") for (int u = 0; u < PLD; ++u)                                            \
            if (l_dst[u] >= 0) P_[l_dst[u]] = pv[u];                                               \
    } while (0)

    int b, oy0, ox0, n0;
    TILE_DECODE(t_begin, b, oy0, ox0, n0);
    LOAD_PATCH(b, oy0, ox0);
    STORE_PATCH(0);

    int rowbase[2];                                      // this lane's two GEMM rows (pixels) inside the patch, + its half of the run
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int m = wm * 64 + tm * 32 + j;
        rowbase[tm] = (p.S * (m / TC)) * p.pitch + (p.S * (m % TC)) * p.Cred + h * HALF;
    }
    int t = t_begin, it = 0;
    while (t < t_end) {                                  // one pass per n-tile the block's run touches (normally one)
        float bw[NS];                                    // this lane's weight column, all K steps: registers for the whole pass
        {
            const float* wcol = p.wk + (size_t)h * p.Np + n0 + wn * 32 + j;
#pragma unroll
            for (int q = 0; q < NS; ++q) bw[q] = wcol[(size_t)q * 2 * p.Np];
        }
        const int nj = n0 + wn * 32 + j;                 // this lane's output channel
        float bias1 = 0.f, ea1 = 1.f, eb1 = 0.f;
        if (nj < p.N) {
            if (p.bias) bias1 = p.bias[nj];
            if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea1 = p.ep_a[nj]; eb1 = p.ep_b[nj]; }
            if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea1 = p.ep_a[nj];
        }
        // AUXM != 1: the bias is one more k-step instead of the accumulators' start value (a 16-register splat that lived through the whole
        // block): the first tap row's padding slot (run index R of lane half 1, zero on both operands otherwise) carries A = 1, B = bias
        if constexpr (AUXM != 1 && R < RP) { if (h) bw[R - HALF] = bias1; }
        const unsigned rowstride = (unsigned)p.Wout * p.N * 4u;
        // lane part of an output offset: channel, the + 4 * h pixel shift of the accumulator layout, the wave's 4 tile rows
        const unsigned obase_l = (unsigned)(nj + 4 * h * p.N) * 4u + (unsigned)(wm * 4) * rowstride;
        const int seg_n0 = n0;
        for (; t < t_end && n0 == seg_n0; ++t, ++it) {
#ifdef CGS_PATCH_STAMPS
            unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                             // patch[it & 1] is in LDS; patch[(it+1) & 1] is free
#ifdef CGS_PATCH_STAMPS
            unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
            int nb_ = b, noy0 = oy0, nox0 = ox0, nn0 = n0;
            const bool more = t + 1 < t_end;
            if (more) {
                // the next tile by increment-and-carry (scalar adds / compares): decoding t + 1 from scratch is six integer divisions
                // by run-time divisors, which the compiler emulates on the VECTOR ALU (~270 VALU ops per tile next to 80 MFMAs)
                nox0 += TC;
                if (nox0 == p.Wout) {
                    nox0 = 0; noy0 += TR;
                    if (noy0 == p.Hout) {
                        noy0 = 0; ++nb_;
                        if (nb_ == p.B) { nb_ = 0; nn0 += PBN; }
                    }
                }
                LOAD_PATCH(nb_, noy0, nox0);
            }
            const float* P = Ps + (size_t)(it & 1) * patch_f;
            const unsigned tile_off = (unsigned)(((b * p.Hout + oy0) * p.Wout + ox0) * p.N) * 4u;      // scalar
            f32x4 auxv[2][4];
            unsigned sgw = 0;
            if constexpr (AUXM == 2) {
                // lane L <-> pixel (tile row wm * 4 + L / 16, column L % 16) of the wave's 64 pixels, plane of its 32-channel group
                const unsigned pixw = (unsigned)((b * p.Hout + oy0 + wm * 4 + (lane >> 4)) * p.Wout + ox0 + (lane & 15));
                // (the plane index rides in the s_offset, which the descriptor's range check does not cover: a wave whose 32-channel
                // group lies in the padding of N -- N = 32, 96 -- must not load, its word would come from past the mask)
                if ((n0 >> 5) + wn < (p.N >> 5))
                    sgw = __builtin_amdgcn_raw_buffer_load_b32(sg_rsrc, pixw * 4u, (unsigned)((n0 >> 5) + wn) * sg_plane_b, 0);
            } else if constexpr (AUXM == 1) {
                const unsigned ab = tile_off + (unsigned)(n0 + wn * 32 + c4) * 4u + (unsigned)(wm * 4) * rowstride;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int lrow = q * 8 + rsub;                       // row of the 32-row tile: pixel (lrow / 16, lrow % 16) of tile rows 2 tm, 2 tm + 1
                        auxv[tm][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            aux_rsrc, ab + (unsigned)(2 * tm + lrow / TC) * rowstride + (unsigned)(lrow % TC) * (unsigned)p.N * 4u, 0, 0));
                    }
            }
#ifdef CGS_PATCH_STAMPS
            unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
            // the accumulators start at the bias: every plain VALU op of the epilogue costs matrix time on this SIMD (32 v_add per
            // tile were worth 16 % of the kernel), the MFMA adds onto whatever is there for free
            f32x16 acc[2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][r] = (AUXM != 1 && R < RP) ? 0.f : bias1;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                f32x2 av[2][NQ];
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) av[tm][q] = *(const f32x2*)(P + rowbase[tm] + ky * p.pitch + 2 * q);
#pragma unroll
                for (int sidx = 0; sidx < HALF; ++sidx) {
                    float a0 = av[0][sidx >> 1][sidx & 1], a1 = av[1][sidx >> 1][sidx & 1];
                    if (sidx >= R - HALF) {      // padding element of the run (compile-time sidx: one step per tap row); the first one is the bias step
                        const float padv = (AUXM != 1 && ky == 0 && sidx == R - HALF) ? 1.f : 0.f;
                        a0 = h ? padv : a0; a1 = h ? padv : a1;
                    }
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bw[ky * HALF + sidx], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bw[ky * HALF + sidx], acc[1], 0, 0, 0);
                }
            }
#ifdef CGS_PATCH_STAMPS
            unsigned long long st3 = __builtin_amdgcn_s_memtime();
#endif
            // the next patch goes to LDS BEFORE this tile's output stores are issued: its wait (vmcnt counts loads and stores in
            // issue order) then covers only the patch loads and the previous tile's stores, both a whole MFMA phase old
            if (more) STORE_PATCH((it + 1) & 1);
#ifdef CGS_PATCH_STAMPS
            unsigned long long st4 = __builtin_amdgcn_s_memtime();
#endif
            // epilogue straight from the accumulator registers: lane (h, j) holds channel j of 16 rows per row tile, so one
            // buffer_store_dword writes two whole 128-byte channel runs (rows R and R + 4).  No LDS round trip, and every
            // address is one of two per-lane bases (tile row 0 / 1 of the row tile) plus an instruction immediate.
            // (The LDS-transposed 16-byte form took 7k cycles per tile here -- a chain of LDS latencies -- against 5k of MFMAs.)
            if constexpr (AUXM == 2) {
                // un-shuffle this lane's word: bit 8e + q holds channel 4q + e  ->  bit c holds channel c
                unsigned nw = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned t8 = (sgw >> (8 * e)) & 0xffu;
                    t8 = (t8 | (t8 << 12)) & 0x000F000Fu;
                    t8 = (t8 | (t8 << 6)) & 0x03030303u;
                    t8 = (t8 | (t8 << 3)) & 0x11111111u;
                    nw |= t8 << e;
                }
                if (nj < p.N) {
                    if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) patch2_store_signs<CGS_EPI_RELU_BWD_AFFINE>(p, out_rsrc, acc, obase_l + tile_off, rowstride, ea1, nw);
                    else patch2_store_signs<CGS_EPI_LRELU_BWD>(p, out_rsrc, acc, obase_l + tile_off, rowstride, ea1, nw);
                }
            } else if constexpr (AUXM == 1) {
                const unsigned ob = tile_off + (unsigned)(n0 + wn * 32 + c4) * 4u + (unsigned)(wm * 4) * rowstride;
                const bool live = n0 + wn * 32 + c4 < p.N;
                f32x4 ea4 = {1.f, 1.f, 1.f, 1.f};
                if (live && p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea4 = *(const f32x4*)(p.ep_a + n0 + wn * 32 + c4);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) E[((r & 3) + 8 * (r >> 2) + 4 * h) * LDE + j] = acc[tm][r];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int lrow = q * 8 + rsub;
                        const f32x4 v = *(const f32x4*)(E + lrow * LDE + c4);
                        f32x4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ax = auxv[tm][q][e];
                            y[e] = p.epilogue == CGS_EPI_RELU_BWD_AFFINE ? (ax > 0.f ? v[e] * ea4[e] : 0.f)
                                 : p.epilogue == CGS_EPI_LRELU_BWD ? (ax > 0.f ? v[e] : 0.2f * v[e]) : v[e] * (1.f - ax * ax);
                        }
                        if (live)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y), out_rsrc,
                                                                   ob + (unsigned)(2 * tm + lrow / TC) * rowstride + (unsigned)(lrow % TC) * (unsigned)p.N * 4u, 0, 0);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            } else if (nj < p.N) {
                switch (p.epilogue) {
                    case CGS_EPI_NONE: patch2_store<CGS_EPI_NONE>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                    case CGS_EPI_LRELU: patch2_store<CGS_EPI_LRELU>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                    case CGS_EPI_AFFINE_RELU: patch2_store<CGS_EPI_AFFINE_RELU>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                    case CGS_EPI_TANH: patch2_store<CGS_EPI_TANH>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                    case CGS_EPI_RELU_BWD_AFFINE: patch2_store<CGS_EPI_RELU_BWD_AFFINE>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                    case CGS_EPI_LRELU_BWD: patch2_store<CGS_EPI_LRELU_BWD>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                    default: patch2_store<CGS_EPI_TANH_BWD>(p, out_rsrc, aux_rsrc, acc, obase_l + tile_off, rowstride, ea1, eb1); break;
                }
            }
#ifdef CGS_PATCH_STAMPS
            if (p.ep_b && lane == 0 && blockIdx.x < 64 && it < 16) {
                unsigned long long* dbg = (unsigned long long*)p.ep_b + (((size_t)blockIdx.x * 4 + wave) * 16 + it) * 8;
                dbg[0] = st0; dbg[1] = st1; dbg[2] = st2; dbg[3] = st3; dbg[4] = st4; dbg[5] = __builtin_amdgcn_s_memtime();
            }
#endif
            b = nb_; oy0 = noy0; ox0 = nox0; n0 = nn0;
        }
    }
#undef TILE_DECODE
#undef LOAD_PATCH
#undef STORE_PATCH
}

// which (KH, RP) instantiation of conv_patch2_kernel serves a layer (0 = none: the general kernel above)
static int patch2_rp(const CgsLayer& L, bool dirT) {
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_PATCH_V1")) return 0;
#endif
    if (dirT || L.sh != 2 || L.sw != 2) return 0;
    if (((L.sw * L.Cb) & 1) != 0) return 0;                 // 8-byte aligned pixel starts
    // the instantiated shape: 5 tap rows of 5 x 3 floats, a 19 x 35 x 3 patch (8 floats per thread)
    if (L.kh == 5 && L.kw == 5 && L.Cb == 3 && (L.Hs % TR) == 0 && (L.Ws % TC) == 0) return 16;
    return 0;
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
    if (patch2_rp(L, dirT)) p.pitch = cgs_round_up(p.PW * p.Cred + 1, 16);      // even rows for ds_read_b64; 112 floats is conflict-free for 5x5x3
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
    if (const int rp = patch2_rp(L, dirT)) return (size_t)L.kh * rp * cgs_round_up(L.Cs, PBN);
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

int cgs_conv_patch_signs_ok(const CgsLayer& L, bool dirT, int epilogue) {
    return patch2_rp(L, dirT) != 0 && (L.Cs % 32) == 0 && (epilogue == CGS_EPI_RELU_BWD_AFFINE || epilogue == CGS_EPI_LRELU_BWD);
}

int cgs_conv_patch_launch(const CgsLayer& L, bool dirT, int B, const float* in, const float* w, const float* bias, float* out,
                          int epilogue, const float* ep_a, const float* ep_b, const float* ep_aux, float* ws, size_t ws_bytes,
                          int prepacked, hipStream_t s, const unsigned* aux_signs) {
    if (aux_signs && !cgs_conv_patch_signs_ok(L, dirT, epilogue))
        return cgs_set_error(CGS_EINVAL, "conv_patch: a sign mask serves the relu' / lrelu' epilogues of the 5x5x3 stride-2 kernel with N %% 32 == 0");
    PatchParams p;
    patch_geom(L, dirT, p);
    p.in = in; p.wk = ws; p.bias = bias; p.ep_a = ep_a; p.ep_b = ep_b; p.ep_aux = ep_aux; p.aux_signs = aux_signs; p.out = out; p.B = B; p.epilogue = epilogue;
    const size_t need = cgs_conv_patch_ws_floats(L, dirT) * sizeof(float);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "conv_patch: workspace %zu < %zu bytes", ws_bytes, need);
    if ((long)B * p.Hin * p.Win * p.Cred * 4 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_patch: input exceeds 2 GiB");
    const int rp2 = patch2_rp(L, dirT);
    if (rp2) {
        if (!prepacked) {
            hipLaunchKernelGGL(pack_patch2_weights_kernel, dim3(cgs_ceil_div(L.kh * rp2 * p.Np, 256)), dim3(256), 0, s, w, ws, L.kh, rp2,
                               L.kw * p.Cred, p.N, p.Np);
            CGS_CHECK_LAUNCH("pack_patch2_weights");
        }
        const bool auxp = epilogue >= CGS_EPI_RELU_BWD_AFFINE;
        const bool sgn = auxp && aux_signs != nullptr;
        const size_t smem2 = ((size_t)2 * p.PH * p.pitch + ((auxp && !sgn) ? (size_t)4 * 32 * 36 : 0)) * sizeof(float);
        CGS_SMEM_ATTR(96 * 1024, "conv_patch2", conv_patch2_kernel<5, 16, 15, 8, 0>);
        CGS_SMEM_ATTR(96 * 1024, "conv_patch2 (aux)", conv_patch2_kernel<5, 16, 15, 8, 1>);
        CGS_SMEM_ATTR(96 * 1024, "conv_patch2 (signs)", conv_patch2_kernel<5, 16, 15, 8, 2>);
        if (smem2 > 96 * 1024 || p.PH * p.PW * p.Cred > 8 * 256) return cgs_set_error(CGS_EINVAL, "conv_patch: patch too large");
        const long tiles2 = (long)B * (p.Hout / TR) * (p.Wout / TC) * (p.Np / PBN);
        if (tiles2 == 0) return CGS_OK;
        if (tiles2 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_patch: grid too large");
        const long slots = (auxp && !sgn) ? 512 : 1024;         // persistent blocks: two (fp32 aux-prefetch form) or four per CU
        long per2 = (tiles2 + slots - 1) / slots;
        if (per2 < 4) per2 = tiles2 >= 4 * 256 ? 4 : 1;
        const unsigned nblk2 = (unsigned)((tiles2 + per2 - 1) / per2);
        if (sgn) hipLaunchKernelGGL((conv_patch2_kernel<5, 16, 15, 8, 2>), dim3(nblk2), dim3(256), smem2, s, p, (int)tiles2, (int)per2);
        else if (auxp) hipLaunchKernelGGL((conv_patch2_kernel<5, 16, 15, 8, 1>), dim3(nblk2), dim3(256), smem2, s, p, (int)tiles2, (int)per2);
        else hipLaunchKernelGGL((conv_patch2_kernel<5, 16, 15, 8, 0>), dim3(nblk2), dim3(256), smem2, s, p, (int)tiles2, (int)per2);
        CGS_CHECK_LAUNCH("conv_patch2");
        cgs_note_kernel(sgn ? "conv_patch2_kernel<5, 16, 15, 8, 2>" : auxp ? "conv_patch2_kernel<5, 16, 15, 8, 1>" : "conv_patch2_kernel<5, 16, 15, 8, 0>");
        return CGS_OK;
    }
    if (!prepacked) {
        if (dirT)
            hipLaunchKernelGGL(pack_patch_weights_T_kernel, dim3(cgs_ceil_div(p.Kp * p.Np, 256)), dim3(256), 0, s, w, ws, L.kh, L.kw, L.Cb,
                               L.Cs, p.K, p.Kp, p.Np);
        else
            hipLaunchKernelGGL(pack_patch_weights_kernel, dim3(cgs_ceil_div(p.Kp * p.Np, 256)), dim3(256), 0, s, w, ws, p.K, p.Kp, p.N, p.Np);
        CGS_CHECK_LAUNCH("pack_patch_weights");
    }
    const size_t smem = ((size_t)p.Kp * PBN + (size_t)2 * p.PH * p.pitch + p.Kp + (size_t)4 * 32 * 36) * sizeof(float);
    CGS_SMEM_ATTR(96 * 1024, "conv_patch", conv_patch_kernel<0, 1, 1>);
    if (smem > 96 * 1024 || p.PH * p.PW * p.Cred > 10 * 256) return cgs_set_error(CGS_EINVAL, "conv_patch: patch too large");
    const long tiles = (long)B * (p.Hout / TR) * (p.Wout / TC) * (p.Np / PBN);
    if (tiles == 0) return CGS_OK;
    if (tiles > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_patch: grid too large");
    // persistent blocks: two per CU, each a contiguous run of tiles (>= 4, so the weight slab load is amortised);
    // small launches keep one tile per block and simply fill the GPU
    long per = (tiles + 511) / 512;
    if (per < 4) per = tiles >= 4 * 256 ? 4 : 1;
    const long blocks = (tiles + per - 1) / per;
    // the two fixed geometries of config 5: 7x7x3 stride 1 (RGB stem, and the backward-data of the RGB head), 4x4x3 stride 2 (PatchGAN stem)
    if (p.K == 147 && p.kw * p.Cred == 21 && p.pitch == 67) {
        CGS_SMEM_ATTR(96 * 1024, "conv_patch <147>", conv_patch_kernel<147, 21, 67>);
        hipLaunchKernelGGL((conv_patch_kernel<147, 21, 67>), dim3((unsigned)blocks), dim3(256), smem, s, p, (int)tiles, (int)per);
        cgs_note_kernel("conv_patch_kernel<147, 21, 67>");
    } else if (p.K == 48 && p.kw * p.Cred == 12 && p.pitch == 103) {
        CGS_SMEM_ATTR(96 * 1024, "conv_patch <48>", conv_patch_kernel<48, 12, 103>);
        hipLaunchKernelGGL((conv_patch_kernel<48, 12, 103>), dim3((unsigned)blocks), dim3(256), smem, s, p, (int)tiles, (int)per);
        cgs_note_kernel("conv_patch_kernel<48, 12, 103>");
    } else {
        hipLaunchKernelGGL((conv_patch_kernel<0, 1, 1>), dim3((unsigned)blocks), dim3(256), smem, s, p, (int)tiles, (int)per);
        cgs_note_kernel("conv_patch_kernel");
    }
    CGS_CHECK_LAUNCH("conv_patch");
    return CGS_OK;
}
