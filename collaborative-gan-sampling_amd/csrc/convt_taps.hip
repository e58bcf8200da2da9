// Transposed stride-2 convolution to ONE output channel with a 4x4 kernel, "taps" form: the last generator deconv of the
// reference's own (MNIST) net, 64 -> 1 (nsgan/GAN.py:99-100: deconv2d 4x4 s2 + sigmoid/tanh head), and the backward-data of its
// first discriminator conv, 1 <- 64 (nsgan/GAN.py:64 through tf.gradients, sampling/collaborator.py:31).
//
// The rows form (convt_quad.hip) makes the horizontal taps the GEMM's columns: kw * N = 4 of 16 -> a quarter of the MFMA work is
// useful, and it reads every input row twice.  With kh * kw * N = 16 ALL taps fit one v_mfma_f32_16x16x4_f32 tile:
//
//   P[b, r, c][(ky, kx)] = sum_ci in[b, r, c, ci] * w[ky][kx][0][ci]                          (one GEMM: M = pixels, N = 16, K = Cs;
//                                                                                             every MFMA column carries work)
//   out[b, Y, X]         = epi(bias + sum over the (r, ky) with 2r + ky - pt = Y and the (c, kx) with 2c + kx - pl = X of P[b, r, c][(ky, kx)])
//                                                                                             (2 x 2 terms, gathered from LDS)
//
// One input row of <= 16 pixels is one MFMA row tile, so P of an input row is ONE accumulator register quad per lane.  A wave
// walks down an image (a task = a block of input rows of one image): per input row 4 coalesced float4 loads per lane (prefetched a
// row ahead), Cs / 4 MFMAs, the P tile into a 3-slot ring of LDS staging tiles, and for the row above -- whose two neighbours are
// then both staged -- the output row pair (2 x 2Ws <= 64 floats: one per lane) as four LDS reads in a fixed order (deterministic).
// The weights need no packing: lane (g, i) reads w[(ky, kx) = i][ci = 16 j + 4 g ..] as one float4 per 16-channel chunk and keeps them
// in registers.  Every input element is read from HBM once (plus a one-row halo per task), every output written once.
#include <stdio.h>

#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TapsParams {
    const float* in;     // [B,Hs,Ws,Cs]
    const float* w;      // [4][4][1][Cs]  (deconv [kh,kw,Cout,Cin] / conv [kh,kw,Cin,Cout] with the 1-channel side in the middle)
    const float* bias;   // [1] or null
    const float* ep_a;   // [1] for RELU_BWD_AFFINE
    const float* ep_aux; // [B,2Hs,2Ws,1] for the *_BWD epilogues
    float* out;          // [B,2Hs,2Ws,1]
    int B, Hs, Ws, Cs;
    int rb;              // input rows per task
    int epilogue;
};

__device__ __forceinline__ float taps_tanh(float x) {      // tanh on exp2 / rcp (5 VALU ops, |error| <= 2.4e-7; see convt_quad.hip)
    const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}

template <int NCH>      // 16-channel chunks of the input: Cs = 16 * NCH
__global__ __launch_bounds__(256) void convt_taps_kernel(TapsParams p, int tasks_total) {
    constexpr int SLP = 17, SPX = 18;                       // staged floats per pixel (16 taps + 1: conflict-free column reads), pixels c = -1 .. 16
    constexpr int SW = 3 * SPX * SLP;                       // one wave's ring of three P tiles
    __shared__ float smem[4 * SW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    float* S = smem + wave * SW;
    for (int q = lane; q < SW; q += 64) S[q] = 0.f;         // the pixels c = -1 and c = 16 (and the pad float) stay zero for good
    // B operand: column i = tap (ky, kx) = (i >> 2, i & 3), rows 16 j + 4 g .. + 3 of chunk j
    f32x4 fb[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) fb[j] = *(const f32x4*)(p.w + (size_t)i * p.Cs + 16 * j + 4 * g);
    const float bias = p.bias ? p.bias[0] : 0.f;
    const float ea = (p.epilogue == CGS_EPI_RELU_BWD_AFFINE && p.ep_a) ? p.ep_a[0] : 1.f;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hs * (unsigned)p.Ws * (unsigned)p.Cs * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
    const unsigned coff = i < p.Ws ? (unsigned)((i * p.Cs + 4 * g) * 4) : 0xFFFFFFF0u;      // pixel i of a row, channels 4 g .. of a chunk
    // gather plan of this lane: output (row yr of the pair, column X): the two row taps and the two column taps
    const int yr = lane >> 5, X = lane & 31;
    const int W2 = 2 * p.Ws;
    //  Y = 2 r' + yr:  yr = 0 -> (r', ky = 1), (r' - 1, ky = 3);  yr = 1 -> (r', ky = 2), (r' + 1, ky = 0)
    //  X = 2 c' + xo:  xo = 0 -> (c', kx = 1), (c' - 1, kx = 3);  xo = 1 -> (c', kx = 2), (c' + 1, kx = 0)
    const int cp = X >> 1, xo = X & 1;
    const int kxA = xo ? 2 : 1, kxB = xo ? 0 : 3, cB = xo ? cp + 1 : cp - 1;
    const int kyA = yr ? 2 : 1, kyB = yr ? 0 : 3, drB = yr ? 1 : -1;
    const int offAA = (cp + 1) * SLP + kyA * 4 + kxA, offAB = (cB + 1) * SLP + kyA * 4 + kxB;      // inside the slot of row r'
    const int offBA = (cp + 1) * SLP + kyB * 4 + kxA, offBB = (cB + 1) * SLP + kyB * 4 + kxB;      // inside the slot of row r' + drB
    const int tasks_per_img = (p.Hs + p.rb - 1) / p.rb;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int t = blockIdx.x * 4 + wave; t < tasks_total; t += gridDim.x * 4) {
        const int b = t / tasks_per_img, ra = (t - b * tasks_per_img) * p.rb;
        const int re = ra + p.rb < p.Hs ? ra + p.rb : p.Hs;             // this task emits the output row pairs of r' in [ra, re)
        f32x4 a_cur[NCH], a_nxt[NCH];
#define TAPS_LOAD(dst, r_)                                                                                  \
    do {                                                                                                    \
        const bool rok_ = (unsigned)(r_) < (unsigned)p.Hs;      /* a row outside the image reads through the EMPTY descriptor: zeros */ \
        const unsigned sb_ = rok_ ? (unsigned)(((b * p.Hs + (r_)) * p.Ws) * p.Cs * 4) : 0u;                 \
        _Pragma("unroll") for (int j = 0; j < NCH; ++j)                                                     \
            dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rok_ ? in_rsrc : null_rsrc, coff, sb_ + 64u * j, 0)); \
    } while (0)
        TAPS_LOAD(a_cur, ra - 1);
        for (int r = ra - 1; r <= re; ++r) {                              // P of rows ra - 1 .. re; row r' = r - 1 is emitted once row r is staged
            if (r < re) TAPS_LOAD(a_nxt, r + 1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].x, fb[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].y, fb[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].z, fb[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].w, fb[j].w, acc, 0, 0, 0);
            }
            // C/D layout of 16x16x4: column (tap) = lane & 15, row (pixel c) = 4 g + reg
            const int slot = (r + 3) % 3;
            float* Sr = S + slot * (SPX * SLP);
#pragma unroll
            for (int e = 0; e < 4; ++e) Sr[(4 * g + e + 1) * SLP + i] = acc[e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int rp = r - 1;                                           // rows rp - 1, rp, rp + 1 = r are staged now
            if (rp >= ra && X < W2) {
                const float* SA = S + ((rp + 3) % 3) * (SPX * SLP);
                const float* SB = S + ((rp + drB + 3) % 3) * (SPX * SLP);
                float v = (((bias + SA[offAA]) + SA[offAB]) + SB[offBA]) + SB[offBB];      // a fixed order
                const size_t o = ((size_t)(b * 2 * p.Hs + 2 * rp + yr)) * W2 + X;
                if (p.epilogue == CGS_EPI_TANH) v = taps_tanh(v);
                else if (p.epilogue == CGS_EPI_LRELU) v = fmaxf(v, 0.2f * v);
                else if (p.epilogue >= CGS_EPI_RELU_BWD_AFFINE) {
                    const float y = p.ep_aux[o];
                    if (p.epilogue == CGS_EPI_TANH_BWD) v *= (1.f - y * y);
                    else if (p.epilogue == CGS_EPI_LRELU_BWD) v = y > 0.f ? v : 0.2f * v;
                    else v = y > 0.f ? v * ea : 0.f;
                }
                p.out[o] = v;
            }
            // (the slot written next, (r + 1) % 3, holds row r - 2: its last reads are the ones just issued by this same wave, and
            // LDS operations of one wave complete in order)
#pragma unroll
            for (int j = 0; j < NCH; ++j) a_cur[j] = a_nxt[j];
        }
#undef TAPS_LOAD
    }
}

// 4x4 kernel, stride 2, ONE output channel, rows of <= 16 input pixels, 16 / 32 / 64 / 128 input channels
int cgs_convt_taps_ok(const CgsLayer& L) {
    return L.Cb == 1 && L.kh == 4 && L.kw == 4 && L.sh == 2 && L.sw == 2 && L.Hb == 2 * L.Hs && L.Wb == 2 * L.Ws && L.Ws <= 16 &&
           (L.Cs == 16 || L.Cs == 32 || L.Cs == 64 || L.Cs == 128);
}

int cgs_convt_taps_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out, int epilogue,
                          const float* ep_a, const float* ep_aux, hipStream_t s) {
    if (((uintptr_t)in & 15) || ((uintptr_t)w & 15)) return cgs_set_error(CGS_EINVAL, "convt_taps: input and weights must be 16-byte aligned");
    if ((long)B * L.Hs * L.Ws * L.Cs * 4 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "convt_taps: input exceeds 2 GiB (split the batch)");
    TapsParams p;
    p.in = in; p.w = w; p.bias = bias; p.ep_a = ep_a; p.ep_aux = ep_aux; p.out = out;
    p.B = B; p.Hs = L.Hs; p.Ws = L.Ws; p.Cs = L.Cs; p.epilogue = epilogue;
    // input rows per task: whole images while that gives the GPU enough waves (one task per wave, ~4 waves per SIMD), else halves
    // / quarters of an image (each task re-reads one halo row above and below its block)
    int rb = L.Hs;
    while ((long)B * cgs_ceil_div(L.Hs, rb) < 4096 && rb > 4) rb = (rb + 1) / 2;
    p.rb = rb;
    const long tasks = (long)B * cgs_ceil_div(L.Hs, rb);
    if (tasks == 0) return CGS_OK;
    if (tasks > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "convt_taps: grid too large");
    long blocks = (tasks + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    switch (L.Cs / 16) {
        case 1: hipLaunchKernelGGL((convt_taps_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, p, (int)tasks); break;
        case 2: hipLaunchKernelGGL((convt_taps_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, p, (int)tasks); break;
        case 4: hipLaunchKernelGGL((convt_taps_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, p, (int)tasks); break;
        default: hipLaunchKernelGGL((convt_taps_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, s, p, (int)tasks); break;
    }
    CGS_CHECK_LAUNCH("convt_taps");
    static thread_local char name[32];
    snprintf(name, sizeof(name), "convt_taps_kernel<%d>", L.Cs / 16);
    cgs_note_kernel(name);
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The forward twin: a 4x4 stride-2 convolution FROM one channel (the reference net's first discriminator conv, 1 -> 64,
// nsgan/GAN.py:64, and the backward-data of its last generator deconv, 64 <- 1): K = 16 taps is half of one 32-deep K tile of the
// implicit GEMM, whose generic-K form gathers it element-wise and pays a 128-row block's prologue and epilogue for 16 k's (38 / 68 us
// for 103 MB of output at batch 2048).  Here a wave owns 32 consecutive output pixels: the 16 taps are the 8 k-steps of
// v_mfma_f32_32x32x2_f32 (lane-half h, step s -> tap 2s + h = (ky, kx) = (s >> 1, 2 (s & 1) + h): one bounds-checked dword load per step
// and lane, prefetched a tile ahead), the weights [16][N] sit in registers, N / 32 accumulator tiles, and the epilogue stores
// straight from the accumulator layout (one buffer_store_dword = two whole 128-byte channel runs).  relu' / lrelu' epilogues read
// either the fp32 aux tensor or its sign mask (cgs_hip.h "sign masks": one word per pixel and 32 channels, fetched as one dword
// per lane and turned into 64-bit lane masks with v_readlane, exactly as in conv_patch2_kernel).  HBM-bound: the output is
// written once, the 1-channel input is read through L1/L2.
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TapsFParams {
    const float* in;         // [B,Hb,Wb,1]
    const float* w;          // [4][4][1][N]
    const float* bias;       // [N] or null
    const float* ep_a;       // [N] for AFFINE_RELU / RELU_BWD_AFFINE
    const float* ep_b;       // [N] for AFFINE_RELU
    const float* ep_aux;     // [B,Hs,Ws,N] for the *_BWD epilogues ...
    const unsigned* aux_signs;   // ... or its sign mask
    float* out;              // [B,Hs,Ws,N]
    int B, Hb, Wb, Hs, Ws, N;
    int pt, pl;
    int epilogue;
};

template <int NT, int AUXM>      // NT = N / 32 accumulator tiles; AUXM: 0 no aux, 1 fp32 aux tensor, 2 sign mask
__global__ __launch_bounds__(256) void conv_taps_kernel(TapsFParams p, int tiles_total) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, j = lane & 31;
    const long M = (long)p.B * p.Hs * p.Ws;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hb * (unsigned)p.Wb * 4u), 0x00020000);
    const unsigned out_bytes = (unsigned)(M * p.N * 4);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t aux_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ep_aux ? p.ep_aux : p.out), 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.aux_signs ? (const void*)p.aux_signs : (const void*)p.out), 0, (int)(out_bytes >> 5), 0x00020000);
    // B operand: step s, k index h -> tap 2 s + h; column j of tile tn
    float fbw[8][NT];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) fbw[s][tn] = p.w[(size_t)(2 * s + h) * p.N + 32 * tn + j];
    float bias[NT], ea[NT], eb[NT];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
        bias[tn] = p.bias ? p.bias[32 * tn + j] : 0.f;
        ea[tn] = (p.epilogue == CGS_EPI_AFFINE_RELU || p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ? p.ep_a[32 * tn + j] : 1.f;
        eb[tn] = p.epilogue == CGS_EPI_AFFINE_RELU ? p.ep_b[32 * tn + j] : 0.f;
    }
    const int RC = p.Hs * p.Ws;
    float av[8], an[8];
    // this lane's 8 input values of tile t_: pixel m = 32 t_ + j, taps 2 s + h
#define TAPSF_LOAD(dst, t_)                                                                               \
    do {                                                                                                  \
        const long m_ = (long)(t_) * 32 + j;                                                              \
        const bool live_ = m_ < M;                                                                        \
        const int b_ = (int)(m_ / RC), rem_ = (int)(m_ - (long)b_ * RC);                                  \
        const int r_ = rem_ / p.Ws, c_ = rem_ - r_ * p.Ws;                                                \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                   \
            const int iy_ = 2 * r_ + (s >> 1) - p.pt, ix_ = 2 * c_ + 2 * (s & 1) + h - p.pl;               \
            const bool ok_ = live_ && (unsigned)iy_ < (unsigned)p.Hb && (unsigned)ix_ < (unsigned)p.Wb;   \
            const unsigned off_ = ok_ ? (unsigned)(((b_ * p.Hb + iy_) * p.Wb + ix_) * 4) : 0xFFFFFFF0u;   \
            dst[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off_, 0, 0)); \
        }                                                                                                 \
    } while (0)
    int t = blockIdx.x * 4 + wave;
    const int tstep = gridDim.x * 4;
    if (t < tiles_total) TAPSF_LOAD(av, t);
    for (; t < tiles_total; t += tstep) {
        if (t + tstep < tiles_total) TAPSF_LOAD(an, t + tstep);
        const unsigned m0 = (unsigned)t * 32u;
        unsigned sgw = 0;
        if constexpr (AUXM == 2) {       // lane L <-> (tile tn = L >> 5, pixel m0 + (L & 31)): word of that pixel in plane tn
            const unsigned pm = m0 + (unsigned)j;
            // (the plane offset rides in the s_offset, outside the descriptor's range check: predicate on the plane AND on the pixel
            // of a ragged last tile, or the last plane's load runs past the mask)
            if (h < NT && pm < (unsigned)M) sgw = __builtin_amdgcn_raw_buffer_load_b32(sg_rsrc, pm * 4u, (unsigned)h * (unsigned)(M * 4), 0);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = bias[tn];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], fbw[s][tn], acc[tn], 0, 0, 0);
        // epilogue from the accumulator layout: register r is row (r & 3) + 8 (r >> 2) + 4 h of the tile, column j
        unsigned nw[2] = {0u, 0u};       // un-shuffled sign words (bit c <-> channel c): [0] tiles 0 / 1 (lane halves), [1] tiles 2 / 3
        if constexpr (AUXM == 2) {
            unsigned w2 = 0;
            if (h + 2 < NT && m0 + (unsigned)j < (unsigned)M)
                w2 = __builtin_amdgcn_raw_buffer_load_b32(sg_rsrc, (m0 + (unsigned)j) * 4u, (unsigned)(h + 2) * (unsigned)(M * 4), 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned t8 = (sgw >> (8 * e)) & 0xffu, u8 = (w2 >> (8 * e)) & 0xffu;
                t8 = (t8 | (t8 << 12)) & 0x000F000Fu; u8 = (u8 | (u8 << 12)) & 0x000F000Fu;
                t8 = (t8 | (t8 << 6)) & 0x03030303u;  u8 = (u8 | (u8 << 6)) & 0x03030303u;
                t8 = (t8 | (t8 << 3)) & 0x11111111u;  u8 = (u8 | (u8 << 3)) & 0x11111111u;
                nw[0] |= t8 << e; nw[1] |= u8 << e;
            }
        }
        const unsigned obase = (m0 + 4u * (unsigned)h) * (unsigned)p.N * 4u + (unsigned)j * 4u;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                const unsigned o = obase + (unsigned)row * (unsigned)p.N * 4u + 128u * tn;
                float v = acc[tn][r];
                if constexpr (AUXM == 2) {
                    // rows `row` (lanes 0-31) and row + 4 (lanes 32-63) of tile tn: lane (tn & 1) * 32 + pixel holds that pixel's word
                    const unsigned src = nw[tn >> 1];
                    const unsigned lo = __builtin_amdgcn_readlane(src, 32 * (tn & 1) + row), hi = __builtin_amdgcn_readlane(src, 32 * (tn & 1) + row + 4);
                    const bool pos = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
                    v = p.epilogue == CGS_EPI_RELU_BWD_AFFINE ? (pos ? v * ea[tn] : 0.f) : (pos ? v : 0.2f * v);
                } else if constexpr (AUXM == 1) {
                    const float y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(aux_rsrc, o, 0, 0));
                    v = p.epilogue == CGS_EPI_RELU_BWD_AFFINE ? (y > 0.f ? v * ea[tn] : 0.f)
                      : p.epilogue == CGS_EPI_LRELU_BWD ? (y > 0.f ? v : 0.2f * v) : v * (1.f - y * y);
                } else {
                    v = p.epilogue == CGS_EPI_LRELU ? fmaxf(v, 0.2f * v)
                      : p.epilogue == CGS_EPI_AFFINE_RELU ? fmaxf(fmaf(ea[tn], v, eb[tn]), 0.f)
                      : p.epilogue == CGS_EPI_TANH ? taps_tanh(v) : v;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, o, 0, 0);      // (rows past M: past num_records, dropped)
            }
#pragma unroll
        for (int s = 0; s < 8; ++s) av[s] = an[s];
    }
#undef TAPSF_LOAD
}

// 4x4 stride-2 'SAME' conv from ONE input channel to 32 / 64 / 96 / 128 channels
int cgs_conv_taps_ok(const CgsLayer& L, int epilogue) {
    (void)epilogue;
    return L.Cb == 1 && L.kh == 4 && L.kw == 4 && L.sh == 2 && L.sw == 2 && (L.Cs % 32) == 0 && L.Cs <= 128;
}

int cgs_conv_taps_signs_ok(const CgsLayer& L, int epilogue) {
    return cgs_conv_taps_ok(L, epilogue) && (epilogue == CGS_EPI_RELU_BWD_AFFINE || epilogue == CGS_EPI_LRELU_BWD);
}

int cgs_conv_taps_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out, int epilogue,
                         const float* ep_a, const float* ep_b, const float* ep_aux, const unsigned* aux_signs, hipStream_t s) {
    const long M = (long)B * L.Hs * L.Ws;
    if (M * L.Cs * 4 > 0x7fffffffL || (long)B * L.Hb * L.Wb * 4 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_taps: a tensor exceeds 2 GiB (split the batch)");
    if (aux_signs && !cgs_conv_taps_signs_ok(L, epilogue)) return cgs_set_error(CGS_EINVAL, "conv_taps: a sign mask serves the relu' / lrelu' epilogues");
    TapsFParams p;
    p.in = in; p.w = w; p.bias = bias; p.ep_a = ep_a; p.ep_b = ep_b; p.ep_aux = ep_aux; p.aux_signs = aux_signs; p.out = out;
    p.B = B; p.Hb = L.Hb; p.Wb = L.Wb; p.Hs = L.Hs; p.Ws = L.Ws; p.N = L.Cs;
    p.pt = cgs_same_pad_before(L.Hb, 4, 2); p.pl = cgs_same_pad_before(L.Wb, 4, 2);
    p.epilogue = epilogue;
    const long tiles = (M + 31) / 32;
    if (tiles == 0) return CGS_OK;
    long blocks = (tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;               // persistent: 8 blocks of 4 independent waves per CU
    const int nt = L.Cs / 32;
    const int auxm = epilogue >= CGS_EPI_RELU_BWD_AFFINE ? (aux_signs ? 2 : 1) : 0;
#define TAPSF_LAUNCH(NT_, AM_) hipLaunchKernelGGL((conv_taps_kernel<NT_, AM_>), dim3((unsigned)blocks), dim3(256), 0, s, p, (int)tiles)
#define TAPSF_CASE(NT_) case NT_: if (auxm == 2) TAPSF_LAUNCH(NT_, 2); else if (auxm == 1) TAPSF_LAUNCH(NT_, 1); else TAPSF_LAUNCH(NT_, 0); break;
    switch (nt) {
        TAPSF_CASE(1) TAPSF_CASE(2) TAPSF_CASE(3)
        default: if (auxm == 2) TAPSF_LAUNCH(4, 2); else if (auxm == 1) TAPSF_LAUNCH(4, 1); else TAPSF_LAUNCH(4, 0); break;
    }
#undef TAPSF_CASE
#undef TAPSF_LAUNCH
    CGS_CHECK_LAUNCH("conv_taps");
    static thread_local char name[32];
    snprintf(name, sizeof(name), "conv_taps_kernel<%d, %d>", nt, auxm);
    cgs_note_kernel(name);
    return CGS_OK;
}
