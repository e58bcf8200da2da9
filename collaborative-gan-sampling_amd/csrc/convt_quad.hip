// Transposed stride-2 convolution with <= 4 output channels on the fp32 matrix cores, "quad" form:
// the last generator deconv (64 -> 3, nsgan/GAN.py:99 / DCGAN g_h4) and the backward-data of the first
// discriminator conv (3 <- 64, nsgan/GAN.py:64 via sampling/collaborator.py:31).
//
// With N = 3 output channels a per-parity-class GEMM fills 3 of 32 MFMA columns.  Instead one GEMM row
// is an input-resolution pixel ("quad") and its columns are all 2x2 output pixels x N channels of that
// quad (4N <= 16 columns = one v_mfma_f32_16x16x4_f32 tile); K runs over the quad's input neighbourhood
// (<= 3x3 pixels) x Cs channels, with zero weights where a (neighbour, parity) pair has no tap:
//
//   out[b, 2r+py, 2c+px, n] = epi(bias[n] + sum_{dy,dx,ci} in[b, r+dy, c+dx, ci] * Wq[(dy,dx,ci)][(py,px,n)])
//
// 12 of 16 columns and 25 of 36 (neighbour,parity) pairs carry work: ~52 % useful MFMA work instead of ~9 %.
// A fragments are read straight from global memory (each lane one float4 = 4 consecutive ci of its
// row; the K order inside a 16-channel chunk is permuted identically for A and B so no shuffle is
// needed); the packed weights (37 KB for 9 x 64 x 16) sit in LDS for the whole block.
#include <stdio.h>

#include <stdlib.h>

#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct QuadParams {
    const float* in;     // [B,Hs,Ws,Cs]
    const float* wq;     // packed [K/4][16][4], K = ny*nx*Cs
    const float* bias;   // [N] or null
    const float* ep_a;   // [N] for RELU_BWD_AFFINE
    const float* ep_aux; // [B,2Hs,2Ws,N] for the *_BWD epilogues
    float* out;          // [B,2Hs,2Ws,N]
    int B, Hs, Ws, Cs, N;
    int dmin_y, ny, dmin_x, nx;
    int epilogue;
};

// w[kh][kw][N][Cs] -> Wq[(a,bq,ci)][(py,px,n)], a/bq index the neighbourhood rows/cols (dy = dmin_y + a)
__global__ void pack_quad_weights_kernel(const float* __restrict__ w, float* __restrict__ wq, int kh, int kw, int N, int Cs,
                                         int pt, int pl, int dmin_y, int ny, int dmin_x, int nx) {
    const int K = ny * nx * Cs;
    const int total = K * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 3, col = (i >> 2) & 15, k = (i >> 6) * 4 + e;
        const int nb = k / Cs, ci = k - nb * Cs;
        const int a = nb / nx, bq = nb - a * nx;
        const int cls = col / N, n = col - cls * N;
        float v = 0.f;
        if (cls < 4) {
            const int py = cls >> 1, px = cls & 1;
            const int ky = py + pt - 2 * (dmin_y + a), kx = px + pl - 2 * (dmin_x + bq);
            if (ky >= 0 && ky < kh && kx >= 0 && kx < kw) v = w[((size_t)(ky * kw + kx) * N + n) * Cs + ci];
        }
        wq[i] = v;
    }
}

template <int MT>
__global__ __launch_bounds__(256) void convt_quad_mfma_kernel(QuadParams p) {
    extern __shared__ __attribute__((aligned(16))) float Bs[];
    const int tid = threadIdx.x;
    const int K = p.ny * p.nx * p.Cs;
    for (int i = tid; i < K * 4; i += 256) ((f32x4*)Bs)[i] = ((const f32x4*)p.wq)[i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const long HW = (long)p.Hs * p.Ws;
    const long total = (long)p.B * HW;
    const long q0 = ((long)blockIdx.x * 4 + wave) * (16 * MT);
    if (q0 >= total) return;

    int pbase[MT], pr[MT], pc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const long q = q0 + t * 16 + i;
        if (q < total) {
            const int b = (int)(q / HW);
            const int rem = (int)(q - (long)b * HW);
            pr[t] = rem / p.Ws; pc[t] = rem - pr[t] * p.Ws; pbase[t] = b * p.Hs;
        } else {
            pr[t] = -(1 << 20); pc[t] = 0; pbase[t] = 0;
        }
    }
    const int nchunk = p.Cs >> 4;
    const int nit = p.ny * p.nx * nchunk;

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 a_cur[MT], a_nxt[MT];
#define LOAD_A(dst, it_)                                                                              \
    do {                                                                                              \
        const int nb_ = (it_) / nchunk, ch_ = (it_) - nb_ * nchunk;                                   \
        const int a_ = nb_ / p.nx, b_ = nb_ - a_ * p.nx;                                              \
        const int dy_ = p.dmin_y + a_, dx_ = p.dmin_x + b_;                                           \
        _Pragma("unroll") for (int t = 0; t < MT; ++t) {                                              \
            const int iy = pr[t] + dy_, ix = pc[t] + dx_;                                             \
            f32x4 v = {0.f, 0.f, 0.f, 0.f};                                                           \
            if ((unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws)                       \
                v = *(const f32x4*)(p.in + ((size_t)(pbase[t] + iy) * p.Ws + ix) * p.Cs + ch_ * 16 + g * 4); \
            dst[t] = v;                                                                               \
        }                                                                                             \
    } while (0)

    LOAD_A(a_cur, 0);
    for (int it = 0; it < nit; ++it) {
        if (it + 1 < nit) LOAD_A(a_nxt, it + 1);
        // B fragment: k-quad (it*4 + g), column i
        const f32x4 fb = *(const f32x4*)(Bs + ((size_t)(it * 4 + g) * 16 + i) * 4);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t].x, fb.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t].y, fb.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t].z, fb.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t].w, fb.w, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) a_cur[t] = a_nxt[t];
    }
#undef LOAD_A

    // epilogue.  C/D of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int col = i;
    const int cls = col / p.N, n = col - cls * p.N;
    if (cls >= 4) return;
    const int py = cls >> 1, px = cls & 1;
    const float bias = p.bias ? p.bias[n] : 0.f;
    const float ea = p.epilogue == CGS_EPI_RELU_BWD_AFFINE ? p.ep_a[n] : 1.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long q = q0 + t * 16 + g * 4 + r;
            if (q >= total) continue;
            const int b = (int)(q / HW);
            const int rem = (int)(q - (long)b * HW);
            const int rr = rem / p.Ws, cc = rem - rr * p.Ws;
            float v = acc[t][r] + bias;
            const size_t o = ((size_t)(b * 2 * p.Hs + 2 * rr + py) * (2 * p.Ws) + 2 * cc + px) * p.N + n;
            if (p.epilogue == CGS_EPI_TANH) v = tanhf(v);
            else if (p.epilogue == CGS_EPI_LRELU) v = fmaxf(v, 0.2f * v);
            else if (p.epilogue == CGS_EPI_TANH_BWD) { const float y = p.ep_aux[o]; v *= (1.f - y * y); }
            else if (p.epilogue == CGS_EPI_LRELU_BWD) v = p.ep_aux[o] > 0.f ? v : 0.2f * v;
            else if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) v = p.ep_aux[o] > 0.f ? v * ea : 0.f;
            p.out[o] = v;
        }
}

// ------------------------------------------------------------------------------------------------
// LDS-patch variant: a block owns a TH x TW tile of quads of one image (8 x 32, or 16 x 16 for narrow / ragged grids: tiles may
// hang over the right / bottom edge, their loads read zeros and their stores are clipped).  Per
// 16-channel chunk the (TH+ny-1) x (TW+nx-1) input patch is staged ONCE in LDS (double buffered, the next chunk's
// global loads in flight under the MFMAs) and every neighbour's A fragment is a conflict-free ds_read_b128 from it:
// each input element leaves L2 ~1.3x instead of 9x.  The 4 image rows x 64 pixels x N channels a wave produces
// are transposed through LDS and stored (and the aux tensor loaded) as 16-byte accesses of whole 768-byte rows.
// ------------------------------------------------------------------------------------------------
// Persistent: a block copies the packed weights to LDS ONCE and walks a contiguous run of tiles (image-major: the tiles of an
// image follow each other, so their halo rows are re-read from this CU's L1 / this XCD's L2 instead of from HBM); the patch
// pipeline runs across tile boundaries (chunk 0 of the next tile is fetched under the last chunk's MFMAs of this one).
// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp (5 VALU ops; ocml's tanhf is ~25, and every VALU op of this epilogue
// costs matrix time on its SIMD): absolute error <= 2.4e-7 over the whole range (1 ulp of exp2 and rcp at results <= 1), +-1 at +-inf
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // exp(2x); inf for large x -> rcp 0 -> 1, 0 for very negative x -> -1
    return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}

// NYX = 3: the 3 x 3 neighbourhood of a 5 x 5 (or 4 x 4) stride-2 layer at compile time -- every LDS fragment address is then the
// lane's base plus an instruction immediate; NYX = 0: neighbourhood from the parameters.
template <int NPAD, int TW, int NYX>   // NPAD = N (<= 4); TW = tile width in quads (32 or 16), tile height 256 / TW
__global__ __launch_bounds__(256, 2) void convt_quad_lds_kernel(QuadParams p, int tiles_total, int tiles_per_block) {
    constexpr int TH = 256 / TW;
    constexpr int QR = TH / 4;                              // quad rows per wave (2 or 4)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i = lane & 15;
    const int ny = NYX ? NYX : p.ny, nx = NYX ? NYX : p.nx;
    const int K = ny * nx * p.Cs;
    const int PH = TH + ny - 1, PW = TW + nx - 1;          // patch rows / cols (pixels)
    const int patch_f4 = PH * PW * 4;                      // float4 per 16-channel patch
    float* Bs = smem;                                      // [K/4][16][4]
    float* Ps = smem + (size_t)K * 16;                     // [2][PH][PW][16]
    const int t_begin = blockIdx.x * tiles_per_block;
    const int t_end = t_begin + tiles_per_block < tiles_total ? t_begin + tiles_per_block : tiles_total;
    if (t_begin >= t_end) return;
    for (int q = tid; q < K * 4; q += 256) ((f32x4*)Bs)[q] = ((const f32x4*)p.wq)[q];

    const int tiles_x = (p.Ws + TW - 1) / TW, tiles_y = (p.Hs + TH - 1) / TH;
    const int tpi = tiles_x * tiles_y;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hs * (unsigned)p.Ws * (unsigned)p.Cs * 4u), 0x00020000);
    constexpr int PL = 6;                                  // float4 staged per thread: ceil(10*34*4 / 256), ceil(18*18*4 / 256)
    f32x4 st[PL];
    int l_pr[PL], l_pc[PL];                                // this thread's patch elements (pixel row / col inside the patch), -2^20 = none
    unsigned l_src[PL];                                    // ... and their byte offset relative to the patch origin pixel, chunk 0
#pragma unroll
    for (int u = 0; u < PL; ++u) {
        const int q = tid + 256 * u;
        const int pixel = q >> 2;
        const int pr = pixel / PW, pc = pixel - pr * PW;
        l_pr[u] = q < patch_f4 ? pr : -(1 << 20); l_pc[u] = pc;
        l_src[u] = (unsigned)(((pr * p.Ws + pc) * p.Cs + (q & 3) * 4) * 4);
    }
    const int nchunk = p.Cs >> 4;
    unsigned okmask = 0;                                   // bit u: element u of the CURRENTLY LOADING tile lies inside the image
#define TILE_DECODE(t_, b_, r0_, c0_)                                                                       \
    do {                                                                                                    \
        b_ = (t_) / tpi;                                                                                    \
        const int trem_ = (t_) - b_ * tpi;                                                                  \
        r0_ = (trem_ / tiles_x) * TH; c0_ = (trem_ - (trem_ / tiles_x) * tiles_x) * TW;                     \
    } while (0)
    // validity of this thread's elements for the tile at (r0_, c0_): once per tile, not per chunk
#define PATCH_VALID(r0_, c0_)                                                                              \
    do {                                                                                                    \
        okmask = 0;                                                                                         \
        _Pragma("unroll") for (int u = 0; u < PL; ++u)                                                      \
            if ((unsigned)((r0_) + l_pr[u] + p.dmin_y) < (unsigned)p.Hs && (unsigned)((c0_) + l_pc[u] + p.dmin_x) < (unsigned)p.Ws) okmask |= 1u << u; \
    } while (0)
#define LOAD_PATCH(b_, r0_, c0_, ch_)                                                                      \
    do {                                                                                                    \
        const unsigned org_ = (unsigned)(((((b_) * p.Hs + (r0_) + p.dmin_y) * p.Ws + (c0_) + p.dmin_x) * p.Cs + (ch_) * 16) * 4);   /* scalar */ \
        _Pragma("unroll") for (int u = 0; u < PL; ++u) {                                                    \
            const unsigned off = (okmask >> u) & 1u ? org_ + l_src[u] : 0xFFFFFFF0u;                        \
            st[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));   \
        }                                                                                                   \
    } while (0)
#define STORE_PATCH(buf_)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < PL; ++u) {                                                       \
        const int q = tid + 256 * u;                                                                       \
        if (q < patch_f4) ((f32x4*)(Ps + (size_t)(buf_) * patch_f4 * 4))[q] = st[u];                       \
    }

    int b, r0, c0;
    TILE_DECODE(t_begin, b, r0, c0);
    PATCH_VALID(r0, c0);
    LOAD_PATCH(b, r0, c0, 0);
    STORE_PATCH(0);
    __syncthreads();
    constexpr int N = NPAD;
    const int rowf = 2 * TW * N;                           // floats per output image row of the tile
    const int cls = i / N, n = i - cls * N;
    const float bias = (cls < 4 && p.bias) ? p.bias[n] : 0.f;
    int buf = 0;
    for (int t = t_begin; t < t_end; ++t) {
        int nb = b, nr0 = r0, nc0 = c0;
        const bool more = t + 1 < t_end;
        if (more) TILE_DECODE(t + 1, nb, nr0, nc0);
        f32x4 acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nchunk; ++ch) {
            const bool last = ch + 1 == nchunk;
            const bool fetch = !last || more;                               // next chunk of this tile, or chunk 0 of the next tile
            if (fetch) {
                if (!last) { LOAD_PATCH(b, r0, c0, ch + 1); }
                else { PATCH_VALID(nr0, nc0); LOAD_PATCH(nb, nr0, nc0, 0); }
            }
            const float* P = Ps + (size_t)buf * patch_f4 * 4;
#pragma unroll
            for (int a = 0; a < ny; ++a)
#pragma unroll
                for (int bq = 0; bq < nx; ++bq) {
                    const int it = (a * nx + bq) * nchunk + ch;            // k-quad group index of (neighbour, chunk)
                    const f32x4 fb = *(const f32x4*)(Bs + ((size_t)(it * 4 + g) * 16 + i) * 4);
                    f32x4 fa[4];
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int pr = (TW == 32 ? 2 * wave + (tt >> 1) : 4 * wave + tt) + a, pc = (TW == 32 ? 16 * (tt & 1) : 0) + i + bq;
                        fa[tt] = *(const f32x4*)(P + ((size_t)(pr * PW + pc) * 4 + g) * 4);
                    }
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tt].x, fb.x, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tt].y, fb.y, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tt].z, fb.z, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tt].w, fb.w, acc[tt], 0, 0, 0);
                    }
                }
            if (fetch) { STORE_PATCH(buf ^ 1); }
            __syncthreads();
            buf ^= 1;
        }
        // ---- epilogue: transpose the wave's image rows x (2 TW px x N) through LDS, then 16-byte row accesses.  The staging
        // area is the patch buffer the last chunk was read from (buf ^ 1 now: free since the barrier above; the other buffer
        // already holds chunk 0 of the next tile) ----
        float* E = Ps + (size_t)(buf ^ 1) * patch_f4 * 4 + (size_t)wave * 2 * QR * rowf;      // [2*QR image rows][2*TW*N]
        if (cls < 4) {
            const int py = cls >> 1, px = cls & 1;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = (TW == 32 ? 16 * (tt & 1) : 0) + g * 4 + r;    // quad column inside the tile
                    E[(2 * (TW == 32 ? (tt >> 1) : tt) + py) * rowf + (2 * c + px) * N + n] = acc[tt][r] + bias;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int f4_per_row = rowf / 4;                       // 48 for N = 3, TW = 32
        const int live_f = 2 * (p.Ws - c0 < TW ? p.Ws - c0 : TW) * N;      // floats of a tile row that lie inside the image
        for (int q = lane; q < 2 * QR * f4_per_row; q += 64) {
            const int yr = q / f4_per_row, xq = q - yr * f4_per_row;
            const int y = 2 * (r0 + QR * wave) + yr;
            if (y >= 2 * p.Hs || xq * 4 >= live_f) continue;
            const size_t o = ((size_t)(b * 2 * p.Hs + y) * (2 * p.Ws) + 2 * c0) * N + xq * 4;
            f32x4 v = *(const f32x4*)(E + yr * rowf + xq * 4);
            if (p.epilogue == CGS_EPI_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fast_tanh(v[e]);
            } else if (p.epilogue == CGS_EPI_LRELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.2f * v[e]);
            } else if (p.epilogue >= CGS_EPI_RELU_BWD_AFFINE) {
                const f32x4 y4 = *(const f32x4*)(p.ep_aux + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.epilogue == CGS_EPI_TANH_BWD) v[e] *= (1.f - y4[e] * y4[e]);
                    else if (p.epilogue == CGS_EPI_LRELU_BWD) v[e] = y4[e] > 0.f ? v[e] : 0.2f * v[e];
                    else v[e] = y4[e] > 0.f ? v[e] * p.ep_a[(xq * 4 + e) % N] : 0.f;
                }
            }
            *(f32x4*)(p.out + o) = v;
        }
        if (more) __syncthreads();          // every wave is done with its staging rows before the next tile's chunk 1 lands there
        b = nb; r0 = nr0; c0 = nc0;
    }
#undef TILE_DECODE
#undef PATCH_VALID
#undef LOAD_PATCH
#undef STORE_PATCH
}

// ------------------------------------------------------------------------------------------------
// "Rows" form (the fast path for kw * N <= 16, e.g. the 5x5 64 -> 3 layers of every DCGAN here): the quad form above spends
// 48 % of its MFMA work on structural zeros (12 of 16 columns, 25 of 36 (neighbour, parity) pairs).  Here the vertical taps are
// folded into K EXACTLY (an output row of parity py sums over the ky of that parity: 2 + 3 = kh row taps for a row PAIR, no
// zero rows) and the horizontal taps become the GEMM's columns, (kx, n) = kw * N of 16:
//
//   Q[Y][c][(kx, n)] = sum_{ky = Y + pt - 2r, ci} in[b, r, c, ci] * w[ky][kx][n][ci]              (MFMA, 94 % useful for 5x5x3)
//   out[b, Y, X, n]  = epi(bias[n] + sum_{kx = X + pl - 2c} Q[Y][c][(kx, n)])                     (<= 3 terms, from LDS)
//
// One wave owns TWO consecutive output row pairs (rows 2R0 .. 2R0+3) of one image: M = the Ws input pixels of a row (MT tiles of
// 16), its A fragments come straight from global memory (each lane one float4 = 4 channels of its pixel; the ny + 1 = 4 input rows
// are read once for both pairs, and again by the neighbouring task that shares two of them -> L1/L2), the packed weights
// (kh * Cs * 16 floats, 20 KB) sit in LDS for the whole persistent block, and the waves never synchronise with each other.  The
// horizontal gather runs through a per-wave LDS staging tile with a per-lane plan kept in an LDS table; the summation order is
// fixed (deterministic).  Reference semantics: tf.nn.conv2d_transpose 'SAME' (nsgan/ops.py:48-67) and the conv's
// Conv2DBackpropInput (sampling/collaborator.py:31).
// ------------------------------------------------------------------------------------------------
struct RowsParams {
    const float* in;     // [B,Hs,Ws,Cs]
    const float* wp;     // packed [kh*Cs/4][16][4]: k = ky*Cs + ci, column = kx*N + n
    const float* bias;   // [N] or null
    const float* ep_a;   // [N] for RELU_BWD_AFFINE
    const float* ep_aux; // [B,2Hs,2Ws,N] for the *_BWD epilogues
    float* out;          // [B,2Hs,2Ws,N]
    int B, Hs, Ws, Cs;
    int kh, kw, pt, pl;
    int dmin_y, ny;      // input rows R + dmin_y .. R + dmin_y + ny - 1 feed the output row pair (2R, 2R+1)
    int epilogue;
};

// w[kh][kw][N][Cs] -> Wp[(ky*Cs + ci)/4][col = kx*N + n][(ky*Cs + ci)%4]
__global__ void pack_rows_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int kh, int kw, int N, int Cs) {
    const int total = kh * Cs * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i & 3, col = (i >> 2) & 15, k = (i >> 6) * 4 + e;
        const int ky = k / Cs, ci = k - ky * Cs;
        const int kx = col / N, n = col - kx * N;
        wp[i] = kx < kw ? w[((size_t)(ky * kw + kx) * N + n) * Cs + ci] : 0.f;
    }
}

// S5: the geometry of a 5x5 'SAME' layer with 64 input channels at compile time (kh = 5, pt = 1, rows R0-1 .. R0+2, 4 chunks): the 16
// (row, chunk) steps of a task become straight-line code -- which row taps apply to a step is known, the loads of step s + 2
// are in flight under the MFMAs of step s with exact vmcnt waits (with the generic form's uniform branches the compiler waits
// for vmcnt(0) before every MFMA group, which serialises loads and matrix work: 133 us instead of 9x for dcgan64's g_h4)
// NP: output row pairs per task.  2 (round 2): 4 input rows per 2 pairs, every input row read by two tasks (HBM bytes 475 MB for
// 319 algorithmic on dcgan64's g_h4).  4: 6 input rows per 4 pairs -- a quarter fewer loads, 1.5 reads per row instead of 2 -- for
// twice the accumulators (64 VGPRs; the launch runs two blocks = 4 waves per SIMD anyway, so 128 VGPRs are there).
// AUXE: the *_BWD epilogues (an aux tensor of the output's shape); compiled apart so that the forward form carries no aux registers.
template <int N, int MT, bool S5, int NP, bool AUXE>   // N output channels, MT 16-pixel tiles per input row (Ws <= 16 * MT)
__global__ __launch_bounds__(512, 4) void convt_rows_kernel(RowsParams p, int tasks_total, int tasks_per_block) {
    const int kh = S5 ? 5 : p.kh, pt = S5 ? 1 : p.pt, dmin_y = S5 ? -1 : p.dmin_y, ny = S5 ? 3 : p.ny;
    constexpr int NYO = 2 * NP;                             // output rows per task
    constexpr int NWV = 8;                                  // waves per block
    // staging of one output row pair: [2 rows][pixel -2 .. ROWS][SL floats]; the pixels < 0 and >= Ws and the columns >= 16 stay
    // zero, so the gather needs no bounds: its three terms are the lane's base address + instruction immediates
    constexpr int ROWS = MT * 16, SL = 20, SP = ROWS + 3;
    constexpr int SW = 2 * SP * SL;
    constexpr int NU = (2 * (2 * ROWS * N / 4) + 63) / 64;  // float4 of a row pair per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (tells the compiler it is wave-uniform: task indices, row offsets and descriptors stay in SGPRs)
    const int g = lane >> 4, i = lane & 15;
    float* Wl = smem;
    float* S = smem + (size_t)kh * p.Cs * 16 + (size_t)wave * SW;
    const int t_begin = blockIdx.x * tasks_per_block;
    const int t_end = t_begin + tasks_per_block < tasks_total ? t_begin + tasks_per_block : tasks_total;
    if (t_begin >= t_end) return;
    for (int q = tid; q < kh * p.Cs * 4; q += 64 * NWV) ((f32x4*)Wl)[q] = ((const f32x4*)p.wp)[q];
    for (int q = lane; q < SW; q += 64) S[q] = 0.f;
    __syncthreads();

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hs * (unsigned)p.Ws * (unsigned)p.Cs * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
    // loop-invariant per-lane part of the A addresses: pixel (16m + i), channels 4g.. (past the row -> out of range -> zeros)
    unsigned coff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) coff[m] = 16 * m + i < p.Ws ? (unsigned)(((16 * m + i) * p.Cs + 4 * g) * 4) : 0xFFFFFFF0u;
    // gather plan of this lane: float4 q = lane + 64u of a row pair -> output row yr, floats 4xq..4xq+3 of that row; each float
    // (X, n) sums Q[yr][c][(kx, n)] over kx = kx0, kx0 + 2, kx0 + 4 with kx0 = (X + pl) & 1 and 2c = X + pl - kx: one step in kx
    // is one pixel down and 2N columns up, a constant stride, and every out-of-range term lands on a zero of the staging tile
    // (c < 0 or >= Ws: zero pixels; kx >= kw: zero-weight column 15 or the never-written columns 16..19).  goff is the address
    // of the LAST term (lowest address), the others are immediates.
    const int f4_per_row = 2 * p.Ws * N / 4;
    constexpr int GSTEP = SL - 2 * N;                       // floats from term kx + 2 to term kx
    // (the plan depends on the lane only: it lives in LDS, one copy per block, and is read back per row pair -- its 18 values per
    // lane would otherwise sit in VGPRs through the K loop)
    int* Tg = (int*)(smem + (size_t)kh * p.Cs * 16 + (size_t)NWV * SW);       // [NU][64][4] staging offsets of the last term
    float* Tb = (float*)(Tg + NU * 64 * 4);                                        // [NU][64][4] bias of each float
    int* Tq = (int*)(Tb + NU * 64 * 4);                                            // [NU][64] float offset of the float4 inside the row pair's output, -1 = none
    if (wave == 0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int q = lane + 64 * u;
            const int yr = q / f4_per_row, xq = q - yr * f4_per_row;
            Tq[u * 64 + lane] = q < 2 * f4_per_row ? yr * (2 * p.Ws * N) + xq * 4 : -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = xq * 4 + e;
                const int X = f / N, n = f - X * N;
                Tb[(u * 64 + lane) * 4 + e] = p.bias ? p.bias[n] : 0.f;
                const int kx = ((X + p.pl) & 1) + 4;            // the last of the three terms
                const int c = (X + p.pl - kx) >> 1;             // exact (even numerator); -2 <= c <= Ws for kw <= 6 (pl <= 2)
                Tg[(u * 64 + lane) * 4 + e] = q < 2 * f4_per_row ? ((yr * SP + c + 2) * SL + kx * N + n) : 0;
            }
        }
    }
    __syncthreads();
    const int nchunk = S5 ? 4 : p.Cs >> 4;
    const int nrow = ny + NP - 1;                         // input rows feeding NP consecutive output row pairs
    constexpr int NIT5 = (3 + NP - 1) * 4;                // S5: (row, chunk) steps of a task
    const int nit = S5 ? NIT5 : nrow * nchunk;
    const size_t out_row = (size_t)2 * p.Ws * N;

    // a task = NP consecutive output row pairs (R0 .. R0 + NP - 1) of one image: the ny + NP - 1 input rows R0 + dmin_y .. are read
    // once for all of them (a row pair alone needs 3), the packed weights of a row tap serve both M tiles
    const int pairs_y = (p.Hs + NP - 1) / NP;
    for (int t = t_begin + wave; t < t_end; t += NWV) {
        const int b = t / pairs_y, R0 = (t - b * pairs_y) * NP;
        f32x4 acc[NYO][MT];
#pragma unroll
        for (int y = 0; y < NYO; ++y)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[y][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 a0[MT], a1[MT], a2[MT];
#define ROWS_LOAD_A(dst, it_)                                                                              \
    do {                                                                                                    \
        const int d_ = (it_) / nchunk, j_ = (it_) - d_ * nchunk;                                            \
        const int rr_ = R0 + dmin_y + d_;                                                                 \
        /* a row outside the image reads through the EMPTY descriptor (num_records 0: every lane gets zeros): no branch */ \
        const bool rok_ = (unsigned)rr_ < (unsigned)p.Hs;                                                   \
        const unsigned sb_ = rok_ ? (unsigned)((((b * p.Hs + rr_) * p.Ws) * p.Cs + 16 * j_) * 4) : 0u;   /* scalar */ \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                      \
            dst[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rok_ ? in_rsrc : null_rsrc, coff[m], sb_, 0)); \
    } while (0)
#define ROWS_MFMA(a_, acc_, ky_, j_)                                                                        \
    do {                                                                                                    \
        const f32x4 fb = *(const f32x4*)(Wl + ((size_t)((((ky_) * p.Cs) >> 2) + 4 * (j_) + g) * 16 + i) * 4); \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                    \
            acc_[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[m].x, fb.x, acc_[m], 0, 0, 0);                \
            acc_[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[m].y, fb.y, acc_[m], 0, 0, 0);                \
            acc_[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[m].z, fb.z, acc_[m], 0, 0, 0);                \
            acc_[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[m].w, fb.w, acc_[m], 0, 0, 0);                \
        }                                                                                                   \
    } while (0)
    // input row R0 + dmin_y + d feeds output row 2 R0 + y through the row tap ky = pt - 2 (dmin_y + d) + y (if that is a tap)
#define ROWS_STEP(a_, it_)                                                                                  \
    do {                                                                                                    \
        const int d_ = (it_) / nchunk, j_ = (it_) - d_ * nchunk;                                            \
        const int ky_ = pt - 2 * (dmin_y + d_);                                                         \
        _Pragma("unroll") for (int y = 0; y < NYO; ++y)                                                     \
            if ((unsigned)(ky_ + y) < (unsigned)kh) ROWS_MFMA(a_, acc[y], ky_ + y, j_);                   \
    } while (0)
        // the aux values of the backward epilogues (NP = 2: the first row pair's are requested before the K loop, the next pair's
        // before the current pair's gather; the tall form has no registers for a second set and requests each pair's before its
        // accumulators go through the staging tile)
        constexpr bool AUX2 = AUXE && NP == 2;
        f32x4 aux[AUXE ? NU : 1], aux_n[AUX2 ? NU : 1];
        if constexpr (AUX2) {
            const size_t o0 = ((size_t)(b * 2 * p.Hs + 2 * R0)) * out_row;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int gq0 = Tq[u * 64 + lane];
                if (gq0 >= 0) aux[u] = *(const f32x4*)(p.ep_aux + o0 + gq0);
            }
        }
        ROWS_LOAD_A(a0, 0);
        if (1 < nit) ROWS_LOAD_A(a1, 1);
#pragma unroll
        for (int it = 0; it < (S5 ? NIT5 : nit); it += 3) {   // two steps of loads in flight under each step's MFMAs
            // (sched_barrier: left alone, the scheduler hoists the straight-line form's fragment reads and loads far ahead and spills)
            if (it + 2 < nit) ROWS_LOAD_A(a2, it + 2);
            if (S5) __builtin_amdgcn_sched_barrier(0);
            ROWS_STEP(a0, it);
            if (S5) __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < nit) {
                if (it + 3 < nit) ROWS_LOAD_A(a0, it + 3);
                if (S5) __builtin_amdgcn_sched_barrier(0);
                ROWS_STEP(a1, it + 1);
                if (S5) __builtin_amdgcn_sched_barrier(0);
            }
            if (it + 2 < nit) {
                if (it + 4 < nit) ROWS_LOAD_A(a1, it + 4);
                if (S5) __builtin_amdgcn_sched_barrier(0);
                ROWS_STEP(a2, it + 2);
                if (S5) __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef ROWS_LOAD_A
#undef ROWS_MFMA
#undef ROWS_STEP
#pragma unroll
        for (int hp = 0; hp < NP; ++hp) {                   // the row pairs, one after the other through the staging tile
            if (R0 + hp >= p.Hs) break;
            const size_t o_pair = ((size_t)(b * 2 * p.Hs + 2 * (R0 + hp))) * out_row;
            int gq[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) gq[u] = Tq[u * 64 + lane];
            if constexpr (AUX2) {     // this pair's aux values were requested one pair ago; request the next pair's now
                if (hp > 0) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) aux[u] = aux_n[u];
                }
                if (hp + 1 < NP && R0 + hp + 1 < p.Hs) {
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        if (gq[u] >= 0) aux_n[u] = *(const f32x4*)(p.ep_aux + o_pair + 2 * out_row + gq[u]);
                }
            } else if constexpr (AUXE) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (gq[u] >= 0) aux[u] = *(const f32x4*)(p.ep_aux + o_pair + gq[u]);
            }
            // Q -> staging (C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + reg); the previous reads of the tile
            // are complete (LDS operations of one wave finish in order; the fences only pin the compiler)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    S[(2 + 16 * m + 4 * g + r) * SL + i] = acc[2 * hp][m][r];
                    S[(SP + 2 + 16 * m + 4 * g + r) * SL + i] = acc[2 * hp + 1][m][r];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (gq[u] < 0) continue;
                f32x4 v;
                const int4 go = *(const int4*)(Tg + (u * 64 + lane) * 4);
                const f32x4 gb = *(const f32x4*)(Tb + (u * 64 + lane) * 4);
                const int goff[4] = {go.x, go.y, go.z, go.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* q_ = S + goff[e];
                    v[e] = ((gb[e] + q_[2 * GSTEP]) + q_[GSTEP]) + q_[0];             // kx0, kx0 + 2, kx0 + 4: a fixed order
                }
                if (p.epilogue == CGS_EPI_TANH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_tanh(v[e]);
                } else if (p.epilogue == CGS_EPI_LRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.2f * v[e]);
                } else if (AUXE && p.epilogue >= CGS_EPI_RELU_BWD_AFFINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = aux[AUXE ? u : 0][e];
                        if (p.epilogue == CGS_EPI_TANH_BWD) v[e] *= (1.f - y * y);
                        else if (p.epilogue == CGS_EPI_LRELU_BWD) v[e] = y > 0.f ? v[e] : 0.2f * v[e];
                        else v[e] = y > 0.f ? v[e] * p.ep_a[(gq[u] + e) % N] : 0.f;
                    }
                }
                *(f32x4*)(p.out + o_pair + gq[u]) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// geometry-only eligibility (the packed layout differs from the quad form's, and a cached packed workspace is keyed by the
// family and the geometry): kw * N columns fit one MFMA tile, whole float4 output rows, weights + staging fit the LDS
static int rows_mt(const CgsLayer& L) {
    if (!(L.Cb == 1 || L.Cb == 3) || L.kw * L.Cb > 16 || L.kw > 6 || (L.Cs % 16) != 0 || ((2 * L.Ws * L.Cb) % 4) != 0) return 0;
    if ((size_t)L.kh * L.Cs * 16 * sizeof(float) > 64 * 1024) return 0;
    if (L.Ws > 32) return 0;                   // (four pixel tiles x four output rows of accumulators do not fit 128 VGPRs)
    return L.Ws <= 16 ? 1 : 2;
}

static void quad_range(int k, int pad, int& lo, int& hi) {
    lo = 1 << 20; hi = -(1 << 20);
    for (int par = 0; par < 2; ++par)
        for (int kk = 0; kk < k; ++kk)
            if (((par + pad - kk) & 1) == 0) {
                const int d = (par + pad - kk) / 2;
                if (d < lo) lo = d;
                if (d > hi) hi = d;
            }
}

size_t cgs_convt_quad_ws_floats_bound(int kh, int kw, int Cs) {
    return (size_t)((kh + 1) / 2 + 1) * ((kw + 1) / 2 + 1) * Cs * 16;
}

// the quad form keeps the whole packed weight image [ny*nx*Cs][16] in LDS: large kernels x many channels do not fit
bool cgs_convt_quad_fits(const CgsLayer& L) {
    const int pt = cgs_same_pad_before(L.Hb, L.kh, 2), pl = cgs_same_pad_before(L.Wb, L.kw, 2);
    int dy0, hy, dx0, hx;
    quad_range(L.kh, pt, dy0, hy);
    quad_range(L.kw, pl, dx0, hx);
    return (size_t)(hy - dy0 + 1) * (hx - dx0 + 1) * L.Cs * 16 * sizeof(float) <= 150 * 1024;
}

int cgs_convt_quad_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out,
                          int epilogue, const float* ep_a, const float* ep_aux, float* ws, size_t ws_bytes, int prepacked,
                          hipStream_t s) {
    QuadParams p;
    p.in = in; p.wq = ws; p.bias = bias; p.out = out; p.ep_a = ep_a; p.ep_aux = ep_aux;
    p.B = B; p.Hs = L.Hs; p.Ws = L.Ws; p.Cs = L.Cs; p.N = L.Cb; p.epilogue = epilogue;
    const int pt = cgs_same_pad_before(L.Hb, L.kh, 2), pl = cgs_same_pad_before(L.Wb, L.kw, 2);
    int hy, hx;
    quad_range(L.kh, pt, p.dmin_y, hy);
    quad_range(L.kw, pl, p.dmin_x, hx);
    p.ny = hy - p.dmin_y + 1; p.nx = hx - p.dmin_x + 1;
    if ((long)B * L.Hs * L.Ws * L.Cs * 4 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "convt_quad: input exceeds 2 GiB (split the batch)");
    // the taps kernel reads the raw weights and the input as float4: a view at an unaligned offset keeps the packed rows / quad forms below
    const bool taps_aligned = !(((uintptr_t)w | (uintptr_t)in) & 15);
#ifdef CGS_EXPERIMENT
    if (cgs_convt_taps_ok(L) && taps_aligned && !getenv("CGS_NO_TAPS")) return cgs_convt_taps_launch(L, B, in, w, bias, out, epilogue, ep_a, ep_aux, s);
#else
    if (cgs_convt_taps_ok(L) && taps_aligned) return cgs_convt_taps_launch(L, B, in, w, bias, out, epilogue, ep_a, ep_aux, s);      // (reads the unpacked weights)
#endif
    if (const int mt = rows_mt(L)) {
        RowsParams r;
        r.in = in; r.wp = ws; r.bias = bias; r.out = out; r.ep_a = ep_a; r.ep_aux = ep_aux;
        r.B = B; r.Hs = L.Hs; r.Ws = L.Ws; r.Cs = L.Cs; r.kh = L.kh; r.kw = L.kw; r.pt = pt; r.pl = pl;
        r.dmin_y = p.dmin_y; r.ny = p.ny; r.epilogue = epilogue;
        const size_t wfl = (size_t)L.kh * L.Cs * 16;
        if (!ws || ws_bytes < wfl * sizeof(float)) return cgs_set_error(CGS_EWORKSPACE, "convt_rows: workspace %zu < %zu bytes", ws_bytes, wfl * sizeof(float));
        if (!prepacked) {
            hipLaunchKernelGGL(pack_rows_weights_kernel, dim3((unsigned)((wfl + 255) / 256)), dim3(256), 0, s, w, ws, L.kh, L.kw, L.Cb, L.Cs);
            CGS_CHECK_LAUNCH("pack_rows_weights");
        }
        const bool s5 = L.kh == 5 && L.Cs == 64 && pt == 1 && p.dmin_y == -1 && p.ny == 3;
        // output row pairs per task: the forward 5x5 64 -> 3 layers run the tall form (with an aux epilogue its 64 accumulators + the
        // aux values do not fit 128 VGPRs: 35 spilled registers)
        const bool tall_ok = s5 && L.Cb == 3 && L.Hs >= 8 && epilogue < CGS_EPI_RELU_BWD_AFFINE;
        int np = tall_ok ? 4 : 2;
#ifdef CGS_EXPERIMENT
        if (getenv("CGS_ROWS_NP")) np = atoi(getenv("CGS_ROWS_NP")) == 4 && tall_ok ? 4 : 2;
#endif
        const long tasks = (long)B * ((L.Hs + np - 1) / np);
        if (tasks == 0) return CGS_OK;
        // persistent blocks of 8 independent waves, two per CU; a block's waves walk neighbouring row pairs of a contiguous run
        long blocks = (tasks + 7) / 8;
        if (blocks > 512) blocks = 512;
        long per = (tasks + blocks - 1) / blocks;
        per = (per + 7) / 8 * 8;
        blocks = (tasks + per - 1) / per;
        const int nu = (2 * (2 * mt * 16 * L.Cb / 4) + 63) / 64;
        const size_t smem = (wfl + (size_t)8 * (2 * (mt * 16 + 3) * 20) + (size_t)nu * 64 * 9) * sizeof(float);
#define ROWS_LAUNCH(NN, MM, SS, PP)                                                                                  \
    if (epilogue >= CGS_EPI_RELU_BWD_AFFINE) ROWS_LAUNCH_(NN, MM, SS, PP, true) else ROWS_LAUNCH_(NN, MM, SS, PP, false)
#define ROWS_LAUNCH_(NN, MM, SS, PP, AA)                                                                             \
    {                                                                                                              \
        static bool done_[64] = {};                                                                                \
        int dv_ = 0;                                                                                               \
        (void)hipGetDevice(&dv_);                                                                                  \
        dv_ &= 63;                                                                                                 \
        if (!done_[dv_]) {                                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)convt_rows_kernel<NN, MM, SS, PP, AA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return cgs_set_error(CGS_ELAUNCH, "convt_rows smem attr: %s", hipGetErrorString(e)); \
            done_[dv_] = true;                                                                                     \
        }                                                                                                          \
        hipLaunchKernelGGL((convt_rows_kernel<NN, MM, SS, PP, AA>), dim3((unsigned)blocks), dim3(512), smem, s, r, (int)tasks, (int)per); \
    }
        if (L.Cb == 3) {
            if (mt == 1) { if (s5 && np == 4) ROWS_LAUNCH_(3, 1, true, 4, false) else if (s5) ROWS_LAUNCH(3, 1, true, 2) else ROWS_LAUNCH(3, 1, false, 2) }
            else { if (s5 && np == 4) ROWS_LAUNCH_(3, 2, true, 4, false) else if (s5) ROWS_LAUNCH(3, 2, true, 2) else ROWS_LAUNCH(3, 2, false, 2) }
        } else {
            if (mt == 1) ROWS_LAUNCH(1, 1, false, 2) else ROWS_LAUNCH(1, 2, false, 2)
        }
#undef ROWS_LAUNCH
#undef ROWS_LAUNCH_
        CGS_CHECK_LAUNCH("convt_rows");
        static thread_local char name[48];
        snprintf(name, sizeof(name), "convt_rows_kernel<%d, %d, %s, %d, %s>", L.Cb, mt, (s5 && L.Cb == 3) ? "true" : "false", np,
                 epilogue >= CGS_EPI_RELU_BWD_AFFINE ? "true" : "false");       // as rocprofv3 prints it
        cgs_note_kernel(name);
        return CGS_OK;
    }
    const size_t K = (size_t)p.ny * p.nx * L.Cs;
    const size_t need = K * 16 * sizeof(float);
    if (!ws || ws_bytes < need) return cgs_set_error(CGS_EWORKSPACE, "convt_quad: workspace %zu < %zu bytes", ws_bytes, need);
    if (need > 150 * 1024) return cgs_set_error(CGS_EINVAL, "convt_quad: packed weights (%zu B) exceed LDS", need);
    if (!prepacked) {
        hipLaunchKernelGGL(pack_quad_weights_kernel, dim3((unsigned)((K * 16 + 255) / 256)), dim3(256), 0, s, w, ws, L.kh, L.kw,
                           L.Cb, L.Cs, pt, pl, p.dmin_y, p.ny, p.dmin_x, p.nx);
        CGS_CHECK_LAUNCH("pack_quad_weights");
    }
    const long total_q = (long)B * L.Hs * L.Ws;
    if (total_q == 0) return CGS_OK;
    // LDS-patch variant: 8 x 32 quad tiles, or 16 x 16 (clipped at the edges) for grids that are not whole 8 x 32 tiles;
    // needs 16-byte aligned output rows
    if (p.ny <= 3 && p.nx <= 3 && ((2 * L.Ws * L.Cb) % 4) == 0 && ((32 * L.Cb) % 4) == 0) {
        const bool wide = (L.Ws % 32) == 0 && (L.Hs % 8) == 0;
        const bool nyx3 = p.ny == 3 && p.nx == 3;
        const int TWr = wide ? 32 : 16, THr = 256 / TWr;
        const int PH = THr + p.ny - 1, PW = TWr + p.nx - 1;
        const size_t smem = need + (size_t)2 * PH * PW * 16 * sizeof(float);
        const long tiles = (long)B * cgs_ceil_div(L.Hs, THr) * cgs_ceil_div(L.Ws, TWr);
        // persistent blocks: two per CU, each a contiguous run of tiles (the 37 KB weight image is copied to LDS once per block);
        // small launches keep one tile per block
        long per = (tiles + 511) / 512;
        if (per < 4) per = tiles >= 4 * 256 ? 4 : 1;
        const long blocks = (tiles + per - 1) / per;
#define QUAD_LDS_LAUNCH(NN, TT, YX)                                                                                \
    {                                                                                                              \
        CGS_SMEM_ATTR(160 * 1024, "convt_quad_lds", convt_quad_lds_kernel<NN, TT, YX>);                            \
        hipLaunchKernelGGL((convt_quad_lds_kernel<NN, TT, YX>), dim3((unsigned)blocks), dim3(256), smem, s, p, (int)tiles, (int)per); \
    }
#define QUAD_LDS_CASE(NN)                                                                                          \
    case NN:                                                                                                       \
        if (wide && nyx3) QUAD_LDS_LAUNCH(NN, 32, 3) else if (wide) QUAD_LDS_LAUNCH(NN, 32, 0)                     \
        else if (nyx3) QUAD_LDS_LAUNCH(NN, 16, 3) else QUAD_LDS_LAUNCH(NN, 16, 0)                                  \
        break;
        if (smem <= 160 * 1024 && (size_t)4 * 256 * L.Cb <= (size_t)2 * PH * PW * 16) {
            switch (L.Cb) {
                QUAD_LDS_CASE(1) QUAD_LDS_CASE(2) QUAD_LDS_CASE(3) QUAD_LDS_CASE(4)
                default: return cgs_set_error(CGS_EINVAL, "convt_quad: N=%d", L.Cb);
            }
#undef QUAD_LDS_CASE
#undef QUAD_LDS_LAUNCH
            CGS_CHECK_LAUNCH("convt_quad_lds");
            static thread_local char name[48];
            snprintf(name, sizeof(name), "convt_quad_lds_kernel<%d, %d, %d>", L.Cb, TWr, nyx3 ? 3 : 0);       // as rocprofv3 prints it
            cgs_note_kernel(name);
            return CGS_OK;
        }
    }
    constexpr int MT = 4;
    static bool attr_done[64] = {};       // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_done[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)convt_quad_mfma_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return cgs_set_error(CGS_ELAUNCH, "convt_quad smem attr: %s", hipGetErrorString(e));
        attr_done[dev_] = true;
    }
    const long total = (long)B * L.Hs * L.Ws;
    if (total == 0) return CGS_OK;
    const long blocks = (total + 64 * MT - 1) / (64 * MT);
    hipLaunchKernelGGL(convt_quad_mfma_kernel<MT>, dim3((unsigned)blocks), dim3(256), need, s, p);
    CGS_CHECK_LAUNCH("convt_quad_mfma");
    cgs_note_kernel("convt_quad_mfma_kernel<4>");
    return CGS_OK;
}
