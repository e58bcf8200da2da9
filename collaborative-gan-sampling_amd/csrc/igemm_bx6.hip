// Implicit-GEMM NHWC convolution through SPLIT-bf16 MFMA ("bx6"): the opt-in contraction of the layers with >= 128 output
// channels (cgs_set_contraction, include/cgs_hip.h).
//
// gfx950 multiplies bf16 on its matrix cores at 16x the fp32 rate.  An fp32 operand splits EXACTLY into three bf16 pieces by
// truncation, x = x0 + x1 + x2 (8 significant bits each; the residuals are exact in fp32), and
//
//     a * b  =  sum_{i+j<=2} a_i b_j  +  (a_1 b_2 + a_2 b_1 + a_2 b_2)        the dropped terms are <= 3 * 2^-24 |a b|
//
// so six bf16 products (each exact in the fp32 accumulator: 8 x 8 significant bits) reproduce the fp32 product to fp32's own
// rounding error: the same contraction as igemm.hip's v_mfma_f32_32x32x2_f32 chain (same geometry, same K order, same fused
// epilogues and statistics -- the reference's tf.nn.conv2d / conv2d_transpose call sites, nsgan/ops.py:41,55), with an error
// of the size of an fp32 fma chain's own (measured: tools/probe/bf16x6_256.hip), at 6/16 of the matrix time.
//
// Tiling: block tile BM x BN = 128 x 256 (or 256 x 128 when N is not a multiple of 256), four waves, each a 128 x 64 tile of
// 4 x 2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers), two blocks per CU; 16-deep K stages (one half of a
// 32-channel chunk of one tap, as igemm.hip's 16-deep form), double-buffered LDS.  The weights are split ONCE at pack time
// into three bf16 planes [k tile][plane][n][16]; the activations are split in the kernel between their global load and the LDS
// store (5.5 vector-ALU operations per element, once per block and K stage).  LDS rows are 32 bytes (one 16-deep K stage of
// one plane), so a fragment read of the 32 x 32 x 16 MFMA is a linear 1 KB per wave.
#include <stdio.h>
#include <stdlib.h>

#include "cgs_internal.h"
#include "igemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// two floats -> their three bf16 pieces, packed pairwise (low half = x0's piece): hi, mid, lo
__device__ __forceinline__ void bx6_split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    p0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302);
}

// buffer_load_dwordx4 ... lds: lane l's 16 bytes at (voffset_l + soffset) land at lds_base + 16 l (wave-uniform base in M0; an
// out-of-range lane writes zeros).  The host pass of a kernel TEMPLATE cannot take this builtin (the instantiation is dropped
// without a diagnostic and the launch stub stays undefined), hence the device-only body.
__device__ __forceinline__ void bx6_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
#endif
}

// ------------------------------------------------------------------------------------------------
// weight packing: w[kh][kw][Cb][Cs] -> per class [K/16][plane][Np][16] bf16 (K in igemm.hip's VEC order: 32-channel chunk, tap,
// channel; a 16-deep tile is one half of a (chunk, tap))
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_bx6_kernel(IgemmParams p, const float* __restrict__ w, unsigned short* __restrict__ packed,
                                        int kw, int Cb, int Cs, int dirT) {
    const IgemmClass& c = p.cls[blockIdx.y];
    const int Kpad = (c.K + CGS_BK - 1) / CGS_BK * CGS_BK;
    const size_t total = (size_t)Kpad * p.Np;
    unsigned short* dst = packed + (size_t)c.w_off * 3;          // (w_off counts Kpad * Np elements of the classes before)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 15);
        const int n = (int)((i >> 4) % p.Np);
        const int kt16 = (int)((i >> 4) / p.Np);
        const int k = (kt16 >> 1) * 32 + (kt16 & 1) * 16 + kk;
        float v = 0.f;
        if (k < c.K && n < p.N) {
            const int ntaps = c.nty * c.ntx;
            const int kt = k / CGS_BK, chunk = kt / ntaps;
            const int t = kt - chunk * ntaps, ci = chunk * CGS_BK + (k - kt * CGS_BK);
            int ta = t / c.ntx, tb = t - ta * c.ntx;
            ta = cgs_tap_order(ta, c.nty, p.tap_parity); tb = cgs_tap_order(tb, c.ntx, p.tap_parity);
            const int tap = (c.ky0 + ta * p.kstep) * kw + (c.kx0 + tb * p.kstep);
            const size_t src = dirT ? ((size_t)tap * Cb + n) * Cs + ci : ((size_t)tap * Cb + ci) * Cs + n;
            v = w[src];
        }
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const float r = v - __builtin_bit_cast(float, u & 0xffff0000u);
        const unsigned x = __builtin_bit_cast(unsigned, r);
        const float q = r - __builtin_bit_cast(float, x & 0xffff0000u);
        const size_t o = ((size_t)kt16 * 3 * p.Np + n) * 16 + kk;
        dst[o] = (unsigned short)(u >> 16);
        dst[o + (size_t)p.Np * 16] = (unsigned short)(x >> 16);
        dst[o + (size_t)p.Np * 32] = (unsigned short)(__builtin_bit_cast(unsigned, q) >> 16);
    }
}

size_t cgs_igemm_bx6_packed_bytes(const IgemmParams& p) { return cgs_packed_floats(p) * 6; }

int cgs_pack_weights_bx6(const IgemmParams& p, const CgsLayer& L, bool dirT, const float* w, void* packed, hipStream_t s) {
    size_t mx = 0;
    for (int i = 0; i < p.nclasses; ++i) {
        const size_t n = (size_t)cgs_round_up(p.cls[i].K, CGS_BK) * p.Np;
        if (n > mx) mx = n;
    }
    if (mx == 0) return CGS_OK;
    int blocks = (int)((mx + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_bx6_kernel, dim3(blocks, p.nclasses), dim3(256), 0, s, p, w, (unsigned short*)packed, L.kw, L.Cb, L.Cs,
                       dirT ? 1 : 0);
    CGS_CHECK_LAUNCH("pack_weights_bx6");
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
#ifndef BX6_V
#define BX6_V 1          // bit 0: the weight tile goes global -> LDS directly (buffer_load ... lds), no register staging
#endif
// waves per SIMD the register budget is sized for: the four-wave blocks run two per CU (256 registers per lane); the two-wave
// 256 x 64 block's LDS lets two blocks = four waves share a CU, one per SIMD (512 registers: its eight staged float4 fit)
// The one-wave 128 x 64 block ("solo", for layers with 64 output channels per tile): no other wave shares its tile, so it keeps ONE LDS
// image (the fragments of a stage are all in registers before the image is overwritten), 18.5 KB: eight blocks = two waves per SIMD
// per CU, no barriers, no lockstep between them.
template <int BM, int BN, bool PAR>
__global__ __launch_bounds__(BM* BN / 128, BM* BN == 128 * 256 || BM * BN == 128 * 64 ? 2 : 1) void igemm_bx6_kernel(IgemmParams p) {
    constexpr int WNN = BN / 64;                       // waves along N; wave tile 128 x 64
    constexpr int NW = (BM / 128) * WNN, NT = 64 * NW;
    constexpr bool SOLO = NW == 1;
    constexpr int NBUF = SOLO ? 1 : 2;                 // LDS images of a stage
    constexpr int PITCH = 32;                          // bytes of an LDS row: 16 bf16 = one K stage of one plane
    constexpr int PLA = BM * PITCH, PLB = BN * PITCH;  // bytes of an A / B plane
    constexpr int BUF = 3 * (PLA + PLB);               // one stage: A planes [3][BM][16], then B planes [3][BN][16]
    constexpr int AI = BM * 4 / NT, AR = NT / 4;       // float4 of A per thread and stage; rows per staging pass
    [[maybe_unused]] constexpr int NB = BN * 6 / NT;   // 16-byte pieces of B per thread and stage when staged through registers (3 planes x BN columns x 2 halves)
    constexpr int ER = 32, LDE = 64 + 4;               // epilogue staging: 32 rows of the wave tile at a time
    constexpr int STAGE_B = NW * ER * LDE * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int* rowpix = (int*)(lds + (NBUF * BUF > STAGE_B ? NBUF * BUF : STAGE_B));      // [BM] output pixel of each tile row, -1 = out of range

    // block id -> (m-tile, n-tile, class): as igemm_kernel (per-XCD decode, heaviest pixels first)
    const int nblk_n = p.Np / BN;
    unsigned wi = blockIdx.x;
    int cls_i = blockIdx.y;
    if (p.xcd_map == 2) {        // the parity classes of a tile back to back on ONE XCD (as igemm_kernel)
        const unsigned xcd = wi & 7u;
        unsigned q = wi >> 3;
        cls_i = (int)(q % (unsigned)p.nclasses);
        q /= (unsigned)p.nclasses;
        wi = (q << 3) | xcd;
    }
    int nb, mb;
    if (p.xcd_map) {
        const unsigned xcd = wi & 7u, q = wi >> 3;
        nb = (int)(q % (unsigned)nblk_n);
        mb = (int)(q / (unsigned)nblk_n) * 8 + (int)xcd;
    } else {
        nb = (int)(wi % (unsigned)nblk_n);
        mb = (int)(wi / (unsigned)nblk_n);
    }
    const IgemmClass& c = p.cls[cls_i];
    const int RC = c.R * c.C;
    const int M = p.B * RC;
    if (mb * BM >= M) return;
    if (p.lpt) {
        const int gpp = p.B / BM;
        const int rank = mb / gpp, grp = mb - rank * gpp;
        mb = (int)p.perm[cls_i][rank] * gpp + grp;
    }
    const int m0 = mb * BM, n0 = nb * BN;

    const int tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNN, wn = wave - wm * WNN;
    const int fh = lane >> 5, fr = lane & 31;

    const int mlast_ = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
    const bool one_pix = p.pix_major && (m0 / p.B) == (mlast_ / p.B);
    const int t_pix = one_pix ? m0 / p.B : 0;
    const int t_b0 = m0 - t_pix * p.B, t_r = t_pix / c.C, t_cc = t_pix - (t_pix / c.C) * c.C;
#define DECODE_ROW(m_, b_, r_, cc_)                                        \
    do {                                                                   \
        if (one_pix) { b_ = t_b0 + ((m_) - m0); r_ = t_r; cc_ = t_cc; }    \
        else {                                                             \
            int rem_;                                                      \
            if (p.pix_major) { rem_ = (m_) / p.B; b_ = (m_) - rem_ * p.B; } \
            else { b_ = (m_) / RC; rem_ = (m_) - b_ * RC; }                \
            r_ = rem_ / c.C; cc_ = rem_ - r_ * c.C;                        \
        }                                                                  \
    } while (0)

    for (int row = tid; row < BM; row += NT) {
        const int m = m0 + row;
        int pix = -1;
        if (m < M) {
            int b, r, cc;
            DECODE_ROW(m, b, r, cc);
            pix = (b * p.Hout + r * p.So + c.py) * p.Wout + cc * p.So + c.px;
        }
        rowpix[row] = pix;
    }

    // A staging: thread -> float4 number aq of the 16-deep chunk of rows ar + AR * i
    const int aq = tid & 3, ar = tid >> 2;
    int a_base[AI], a_iy[AI], a_ix[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + ar + AR * i;
        if (m < M) {
            int b, r, cc;
            DECODE_ROW(m, b, r, cc);
            a_base[i] = b * p.Hin * p.Win; a_iy[i] = r * p.S + c.dy0; a_ix[i] = cc * p.S + c.dx0;
        } else {
            a_base[i] = 0; a_iy[i] = -(1 << 20); a_ix[i] = 0;
        }
    }
#undef DECODE_ROW
    const int nk = (c.K + 15) / 16;                     // 16-deep K tiles (K % 32 == 0)
    // zero-tap skipping: every row of a one-pixel tile sees the same taps outside the image -> their K tiles are never executed
    const bool skip_ok = one_pix;
    const int u_iy = t_r * p.S + c.dy0, u_ix = t_cc * p.S + c.dx0;
    struct KIt { int kt, sub, ia, ib, chunk; };          // K tile and its decode: 32-channel chunk, tap (ia, ib) in visiting order, 16-deep half
    auto kit_valid = [&](const KIt& s) -> bool {
        if (!skip_ok) return true;
        const int ta = cgs_tap_order(s.ia, c.nty, PAR), tb = cgs_tap_order(s.ib, c.ntx, PAR);
        const int iy = u_iy + ta * p.dstep, ix = u_ix + tb * p.dstep;
        return (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    };
    auto kit_next_tap = [&](KIt& s) {
        for (;;) {
            if (++s.ib == c.ntx) { s.ib = 0; if (++s.ia == c.nty) { s.ia = 0; ++s.chunk; } }
            if (s.kt >= nk || kit_valid(s)) break;
            s.kt += 2;
        }
        if (s.kt > nk) s.kt = nk;
    };
    auto kit_first = [&]() -> KIt {
        KIt s;
        s.kt = 0; s.sub = 0; s.chunk = 0; s.ia = 0; s.ib = 0;
        if (s.kt < nk && !kit_valid(s)) { s.kt += 2; kit_next_tap(s); }
        if (s.kt > nk) s.kt = nk;
        return s;
    };
    auto kit_next = [&](KIt s) -> KIt {
        if (s.sub == 0) { s.sub = 1; ++s.kt; if (s.kt > nk) s.kt = nk; return s; }
        s.sub = 0; ++s.kt;
        kit_next_tap(s);
        return s;
    };

    // loop-invariant address parts
    unsigned rowoff[AI], tapmask[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i)
        rowoff[i] = ((unsigned)(a_base[i] + a_iy[i] * p.Win + a_ix[i]) * (unsigned)p.Cred + (unsigned)aq * 4u) * 4u;
    if (one_pix) {
        unsigned mk = 0;
        for (int ia = 0; ia < c.nty; ++ia)
            if ((unsigned)(u_iy + cgs_tap_order(ia, c.nty, PAR) * p.dstep) < (unsigned)p.Hin) mk |= 1u << ia;
        for (int ib = 0; ib < c.ntx; ++ib)
            if ((unsigned)(u_ix + cgs_tap_order(ib, c.ntx, PAR) * p.dstep) < (unsigned)p.Win) mk |= 1u << (16 + ib);
#pragma unroll
        for (int i = 0; i < AI; ++i) tapmask[i] = m0 + ar + AR * i < M ? mk : 0u;
    } else {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            unsigned mk = 0;
            for (int ia = 0; ia < c.nty; ++ia)
                if ((unsigned)(a_iy[i] + cgs_tap_order(ia, c.nty, PAR) * p.dstep) < (unsigned)p.Hin) mk |= 1u << ia;
            for (int ib = 0; ib < c.ntx; ++ib)
                if ((unsigned)(a_ix[i] + cgs_tap_order(ib, c.ntx, PAR) * p.dstep) < (unsigned)p.Win) mk |= 1u << (16 + ib);
            tapmask[i] = mk;
        }
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * (unsigned)p.Cred * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const unsigned short*)p.wp + (size_t)c.w_off * 3), 0, 0x7ffffff0, 0x00020000);
    const int b_tile_bytes = 3 * p.Np * PITCH;
#if BX6_V & 1
    // LDS-DMA: a wave-instruction copies 1 KB = 32 columns of one plane (lane l -> bytes [16 l, 16 l + 16) of the piece, in global
    // memory and in LDS alike: the packed tile and its LDS image are the same linear array); piece = wave + NW * u
    constexpr int PPP = BN / 32;                       // pieces per plane
    constexpr int PB = 3 * PPP / NW;                   // pieces per wave and stage
    const unsigned b_lane = (unsigned)lane * 16u;      // (the only per-lane part; the piece's place in the tile rides in the scalar offset)
    int b_soff[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int piece = wave + NW * u, plane = piece / PPP, sub = piece - plane * PPP;
        b_soff[u] = __builtin_amdgcn_readfirstlane(plane * p.Np * PITCH + n0 * PITCH + sub * 1024);
    }
#else
    unsigned b_voff[NB], b_lds[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int idx = tid + NT * u;
        const int plane = idx / (BN * 2), rem = idx - plane * (BN * 2);
        b_voff[u] = (unsigned)(plane * p.Np * PITCH + n0 * PITCH + rem * 16);
        b_lds[u] = (unsigned)(3 * PLA + plane * PLB + rem * 16);
    }
    u32x4 rb[NB];
#endif
    unsigned a_off[AI];
    f32x4 ra[AI];
#define ADDR_TILE(s_)                                                                                           \
    do {                                                                                                        \
        const int ta_ = cgs_tap_order((s_).ia, c.nty, PAR), tb_ = cgs_tap_order((s_).ib, c.ntx, PAR);           \
        const int dy_ = ta_ * p.dstep, dx_ = tb_ * p.dstep;                                                     \
        const unsigned soff_ = (unsigned)(((dy_ * p.Win + dx_) * p.Cred + (s_).chunk * 32 + (s_).sub * 16) * 4); \
        const unsigned need_ = (1u << (s_).ia) | (1u << (16 + (s_).ib));                                        \
        _Pragma("unroll") for (int i = 0; i < AI; ++i)                                                          \
            a_off[i] = (tapmask[i] & need_) == need_ ? rowoff[i] + soff_ : 0xFFFFFFF0u;                         \
    } while (0)
#if BX6_V & 1
    // (the weight tile of stage s_ goes straight into LDS buffer DST_: free since the barrier that ended the stage before, complete
    // before the one that ends this stage -- the barrier's fence waits for the DMA)
#define ISSUE_A(RA_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < AI; ++i)                                                              \
        RA_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, a_off[i], 0, 0));
#define ISSUE_TILE(s_, DST_)                                                                                    \
    do {                                                                                                        \
        ISSUE_A(ra);                                                                                            \
        ISSUE_B(s_, DST_);                                                                                      \
    } while (0)
#define ISSUE_B(s_, DST_)                                                                                       \
    do {                                                                                                        \
        const int b_soff_ = (s_).kt * b_tile_bytes;                                                             \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) {                                                        \
            const int piece_ = wave + NW * u, plane_ = piece_ / PPP, sub_ = piece_ - plane_ * PPP;              \
            bx6_dma16(w_rsrc, lds + (DST_) * BUF + 3 * PLA + plane_ * PLB + sub_ * 1024, b_lane, b_soff_ + b_soff[u]); \
        }                                                                                                       \
    } while (0)
#else
#define ISSUE_A(RA_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < AI; ++i)                                                              \
        RA_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, a_off[i], 0, 0));
#define ISSUE_B(s_, DST_)                                                                                       \
    do {                                                                                                        \
        const int b_soff_ = (s_).kt * b_tile_bytes;                                                             \
        _Pragma("unroll") for (int u = 0; u < NB; ++u)                                                          \
            rb[u] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff[u], b_soff_, 0);                       \
    } while (0)
#define ISSUE_TILE(s_, DST_)                                                                                    \
    do {                                                                                                        \
        ISSUE_A(ra);                                                                                            \
        ISSUE_B(s_, DST_);                                                                                      \
    } while (0)
#endif
    const unsigned a_lds = (unsigned)(ar * PITCH + aq * 8);
#define STORE_TILE(BUF_) STORE_TILE_R(BUF_, ra)
#define STORE_TILE_R(BUF_, RA_)                                                                                 \
    do {                                                                                                        \
        unsigned char* base_ = lds + (BUF_) * BUF;                                                              \
        _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                        \
            unsigned q0_[2], q1_[2], q2_[2];                                                                    \
            bx6_split2(RA_[i][0], RA_[i][1], q0_[0], q1_[0], q2_[0]);                                           \
            bx6_split2(RA_[i][2], RA_[i][3], q0_[1], q1_[1], q2_[1]);                                           \
            unsigned char* d_ = base_ + a_lds + i * AR * PITCH;                                                 \
            *(u32x2*)(d_) = u32x2{q0_[0], q0_[1]};                                                              \
            *(u32x2*)(d_ + PLA) = u32x2{q1_[0], q1_[1]};                                                        \
            *(u32x2*)(d_ + 2 * PLA) = u32x2{q2_[0], q2_[1]};                                                    \
        }                                                                                                       \
        STORE_B(base_);                                                                                         \
    } while (0)
#if BX6_V & 1
#define STORE_B(base_)
#else
#define STORE_B(base_) _Pragma("unroll") for (int u = 0; u < NB; ++u) *(u32x4*)((base_) + b_lds[u]) = rb[u];
#endif

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const unsigned char* fa = lds + (wm * 128 + fr) * PITCH + fh * 16;
    const unsigned char* fb = lds + 3 * PLA + (wn * 64 + fr) * PITCH + fh * 16;
#define MM(pa, pb)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[i][pa], b_[j][pb], acc[i][j], 0, 0, 0);
    // (the six products plane by plane, hi x hi first: the other planes' fragments land behind its eight MFMAs)
#define COMPUTE(BUF_)                                                                                           \
    {                                                                                                           \
        bf16x8 a_[4][3], b_[2][3];                                                                              \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) a_[i][pl] = *(const bf16x8*)(fa + (BUF_) * BUF + pl * PLA + i * 32 * PITCH); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) b_[j][pl] = *(const bf16x8*)(fb + (BUF_) * BUF + pl * PLB + j * 32 * PITCH); \
        }                                                                                                       \
        MM(0, 0) MM(0, 1) MM(1, 0) MM(1, 1) MM(0, 2) MM(2, 0)                                                   \
    }
#define KIT_SEL(d_, c_, a_, b_)                                                                                 \
    do {                                                                                                        \
        const bool c__ = (c_);                                                                                  \
        d_.kt = c__ ? a_.kt : b_.kt; d_.sub = c__ ? a_.sub : b_.sub; d_.ia = c__ ? a_.ia : b_.ia;               \
        d_.ib = c__ ? a_.ib : b_.ib; d_.chunk = c__ ? a_.chunk : b_.chunk;                                      \
        /* wave-uniform by construction; say so, or the scalar offsets derived from them get waterfall loops */ \
        d_.kt = __builtin_amdgcn_readfirstlane(d_.kt); d_.sub = __builtin_amdgcn_readfirstlane(d_.sub);         \
        d_.ia = __builtin_amdgcn_readfirstlane(d_.ia); d_.ib = __builtin_amdgcn_readfirstlane(d_.ib);           \
        d_.chunk = __builtin_amdgcn_readfirstlane(d_.chunk);                                                    \
    } while (0)
    // one K stage: issue the next stage's loads, contract this one from LDS buffer BUF_, split + stage the next one into the
    // other buffer, one barrier.  After the last stage the "next" one is a harmless reload of it into the dead buffer.
#ifdef BX6_STAMPS      // diagnostic build only (tools/bx6_probe.py): cycles per phase of a stage, summed over the block's K loop
#define STAMP(i_) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i_] += t_ - st_last; st_last = t_; __builtin_amdgcn_sched_barrier(0); }
    unsigned long long st_sum[4] = {0, 0, 0, 0}, st_last = 0, st_rt0 = 0;
#else
#define STAMP(i_)
#endif
#define TILE_BODY(BUF_, NXT_)                                                                                   \
    {                                                                                                           \
        KIt ld;                                                                                                 \
        KIT_SEL(ld, (NXT_).kt < nk, NXT_, cur);                                                                 \
        ADDR_TILE(ld);                                                                                          \
        ISSUE_TILE(ld, (BUF_) ^ 1);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        STAMP(0);                                                                                               \
        COMPUTE(BUF_);                                                                                          \
        STAMP(1);                                                                                               \
        STORE_TILE((BUF_) ^ 1);                                                                                 \
        STAMP(2);                                                                                               \
        __syncthreads();                                                                                        \
        STAMP(3);                                                                                               \
        cur = NXT_;                                                                                             \
    }

    // Measured and not kept (tools/sessions/r04_i.sh, the ten 107-GFLOP layers of the headline, sum of the launches): the activations two
    // stages ahead in a second register set with their split + LDS stores inside the MFMA stream 5734 us against 5762 us for this
    // form; that plus the stage's loads one per two MFMAs inside the stream (sched_group_barrier) 5691 us, with spills.  The kernel
    // does not hang on its per-wave schedule: the matrix pipe is 43-48 % busy in every form ("bx6").
    KIt cur = kit_first();
    if (cur.kt < nk) {
        ADDR_TILE(cur);
        ISSUE_TILE(cur, 0);
        STORE_TILE(0);
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
#ifdef BX6_STAMPS
    st_rt0 = __builtin_amdgcn_s_memrealtime();
    st_last = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (SOLO) {
        // one wave, one LDS image: read ALL fragments of the stage, then -- the image is dead -- start the DMA of the next stage's weights
        // into it and the loads of its activations, contract, split + store the activations, wait for the DMA.  (__syncthreads of a
        // one-wave block is its fence: vmcnt(0) lgkmcnt(0).)
        while (cur.kt < nk) {
            const KIt nxt = kit_next(cur);
            KIt ld;
            KIT_SEL(ld, nxt.kt < nk, nxt, cur);
            bf16x8 a_[4][3], b_[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a_[i][pl] = *(const bf16x8*)(fa + pl * PLA + i * 32 * PITCH);
#pragma unroll
                for (int j = 0; j < 2; ++j) b_[j][pl] = *(const bf16x8*)(fb + pl * PLB + j * 32 * PITCH);
            }
            ADDR_TILE(ld);
            ISSUE_A(ra);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment has left LDS: the image may be overwritten
            ISSUE_B(ld, 0);
            STAMP(0);
            MM(0, 0) MM(0, 1) MM(1, 0) MM(1, 1) MM(0, 2) MM(2, 0)
            STAMP(1);
            STORE_TILE(0);
            STAMP(2);
            __syncthreads();
            STAMP(3);
            cur = nxt;
        }
    } else {
        KIt n1 = cur;
        if (cur.kt < nk) n1 = kit_next(cur);
        while (n1.kt < nk) {                             // at least two stages left: cur (in buffer 0) and n1
            const KIt n2 = kit_next(n1);
            KIt n3 = n2;
            if (n2.kt < nk) n3 = kit_next(n2);
            TILE_BODY(0, n1);
            TILE_BODY(1, n2);
            n1 = n3;
        }
        if (cur.kt < nk) TILE_BODY(0, n1);               // an odd last stage
    }

#ifdef BX6_STAMPS
    if (p.slab && tid == 0) {
        unsigned long long* dbg = (unsigned long long*)p.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        dbg[0] = st_rt0; dbg[1] = __builtin_amdgcn_s_memrealtime();
        dbg[2] = st_sum[0]; dbg[3] = st_sum[1]; dbg[4] = st_sum[2]; dbg[5] = st_sum[3];
        dbg[6] = (unsigned long long)nk;
        dbg[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    }
#endif
#undef STAMP
#undef TILE_BODY
#undef KIT_SEL
#undef COMPUTE
#undef MM
#undef STORE_TILE
#undef STORE_B
#undef ISSUE_TILE
#undef ISSUE_A
#undef ISSUE_B
#undef STORE_TILE_R
#undef ADDR_TILE

    // ---- epilogue: the wave tile 32 rows at a time through LDS (the K-loop buffers are dead: every wave is past the last
    // barrier), each lane then owns 4 consecutive channels of a row: 16-byte aux loads / stores.  C layout of the 32x32 MFMA:
    // column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    __builtin_amdgcn_s_setprio(3);
    float* E = (float*)lds + wave * ER * LDE;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;      // 16 lanes per 64-column row, 4 rows per pass
    const int n = n0 + wn * 64 + c4;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, ea = {1.f, 1.f, 1.f, 1.f}, eb = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
        if (p.bias) bias = *(const f32x4*)(p.bias + n);
        if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = *(const f32x4*)(p.ep_a + n); eb = *(const f32x4*)(p.ep_b + n); }
        if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = *(const f32x4*)(p.ep_a + n);
    }
    f32x4 st_a = {0.f, 0.f, 0.f, 0.f}, st_b = {0.f, 0.f, 0.f, 0.f};
    // (the four passes are written out with LITERAL accumulator indices: left to the unroller, the pass loop -- once its body grew by
    // the sign-mask forms -- stayed a loop, the accumulators were indexed at run time and moved to scratch memory: 4x slower kernels)
    auto finish_pass = [&](const int tm) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (n < p.N) {
            const int* rp = rowpix + wm * 128 + tm * ER;
            if (p.stat_part) {
                epilogue_rows<CGS_EPI_NONE, ER, 4, LDE, 1>(p, E, rp, rsub, c4, n, bias, ea, eb, &st_a, &st_b);
            } else if (p.sign_out) {      // (N % 64 == 0: every lane of the wave is inside N, the ballots see whole rows; same mask layout as igemm.hip)
                if (p.epilogue == CGS_EPI_AFFINE_RELU) epilogue_rows_signs<CGS_EPI_AFFINE_RELU, ER, 4, LDE, true>(p, E, rp, rsub, c4, n, bias, ea, eb);
                else epilogue_rows_signs<CGS_EPI_LRELU, ER, 4, LDE, true>(p, E, rp, rsub, c4, n, bias, ea, eb);
            } else
            switch (p.epilogue) {
                case CGS_EPI_NONE: epilogue_rows<CGS_EPI_NONE, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                case CGS_EPI_LRELU: epilogue_rows<CGS_EPI_LRELU, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                case CGS_EPI_AFFINE_RELU: epilogue_rows<CGS_EPI_AFFINE_RELU, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                case CGS_EPI_TANH: epilogue_rows<CGS_EPI_TANH, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                case CGS_EPI_RELU_BWD_AFFINE: epilogue_rows<CGS_EPI_RELU_BWD_AFFINE, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                case CGS_EPI_LRELU_BWD: epilogue_rows<CGS_EPI_LRELU_BWD, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                default: epilogue_rows<CGS_EPI_TANH_BWD, ER, 4, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (p.stat_part && (tm & 1)) {
            // fused norm statistics, igemm.hip's layout: one partial row per 64 GEMM rows, [2 * (m / 128) + half][sum | sum of squares][N];
            // xor-shuffle tree over the lanes that share a column group (fixed order: deterministic)
#pragma unroll
            for (int off = 16; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) { st_a[e] += __shfl_xor(st_a[e], off); st_b[e] += __shfl_xor(st_b[e], off); }
            if (rsub == 0 && n < p.N && m0 + wm * 128 < M) {      // (a 128-row tile wholly past M has no partial rows; a half past M writes its zeros)
                float* dst = p.stat_part + ((size_t)(cls_i * p.stat_cls_rows + (m0 / 128 + wm) * 2 + (tm >> 1)) * 2) * p.N + n;
                *(f32x4*)dst = st_a;
                *(f32x4*)(dst + p.N) = st_b;
            }
            st_a = f32x4{0.f, 0.f, 0.f, 0.f}; st_b = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
#define STAGE_PASS(TM_)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int tn = 0; tn < 2; ++tn)                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                      \
                E[((r & 3) + 8 * (r >> 2) + 4 * fh) * LDE + tn * 32 + fr] = acc[TM_][tn][r];                    \
        finish_pass(TM_);                                                                                       \
    }
    STAGE_PASS(0) STAGE_PASS(1) STAGE_PASS(2) STAGE_PASS(3)
#undef STAGE_PASS
}

// which calls the split-bf16 form serves: the 32-channel-chunk K order (Cred % 32 == 0, <= 16 taps per axis), whole 64-column
// wave tiles; unless any_size, grids that fill the GPU (>= 256 blocks) over a reduction deep enough to amortise the 128-register
// epilogue -- smaller calls keep the exact-fp32 kernel with its split-K forms
int cgs_igemm_bx6_ok(const CgsLayer& L, bool dirT, int B, bool any_size) {
    const int Cred = dirT ? L.Cs : L.Cb, N = dirT ? L.Cb : L.Cs;
    if ((Cred % 32) || (N % 64) || L.kh > 16 || L.kw > 16) return 0;
    if (dirT && (L.sh > 2 || L.sw > 2)) return 0;
    if ((long)L.kh * L.kw * Cred * N * 6 > 0x7fffffffL) return 0;      // (the packed planes of a class are addressed with 32-bit byte offsets)
    if (any_size) return 1;
    const long M = (long)B * (dirT ? (long)L.Hb * L.Wb : (long)L.Hs * L.Ws);
    const long blocks = (N % 256) == 0 ? (M / 128) * (N / 256) : (N % 128) == 0 ? (M / 256) * (N / 128) : (M / 128) * (N / 64) / 2;    // (the one-wave blocks run eight per CU)
    const long K = (long)L.kh * L.kw * Cred / (dirT ? L.sh * L.sw : 1);
    return blocks >= 256 && K >= 512;
}

template <int BM, int BN, bool PAR>
static int launch_bx6(const IgemmParams& p, hipStream_t s) {
    constexpr int NT = BM * BN / 128;
    constexpr size_t buf = (NT == 64 ? 1 : 2) * 3 * (size_t)(BM + BN) * 32, stage = (size_t)(NT / 64) * 32 * 68 * 4;
    constexpr size_t smem = (buf > stage ? buf : stage) + BM * sizeof(int);
    CGS_SMEM_ATTR(smem, "igemm_bx6", igemm_bx6_kernel<BM, BN, PAR>);
    long maxM = 0;
    for (int i = 0; i < p.nclasses; ++i) {
        const long m = (long)p.B * p.cls[i].R * p.cls[i].C;
        if (m > maxM) maxM = m;
    }
    if (maxM == 0) return CGS_OK;
    const long mtiles = (maxM + BM - 1) / BM;
    IgemmParams q = p;
    q.xcd_map = (mtiles % 8 == 0 || mtiles >= 64) ? 1 : 0;
    long gx = (q.xcd_map ? (mtiles + 7) / 8 * 8 : mtiles) * (p.Np / BN);
    unsigned gy = (unsigned)p.nclasses;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_CLS_INTER") && atoi(getenv("CGS_CLS_INTER")) && q.xcd_map == 1 && (mtiles % 8) == 0 && p.nclasses > 1 && !p.pix_major && gx >= atol(getenv("CGS_CLS_INTER"))) {
        bool eq = true;
        for (int i = 1; i < p.nclasses; ++i) eq = eq && p.cls[i].R * p.cls[i].C == p.cls[0].R * p.cls[0].C;
        if (eq) { q.xcd_map = 2; gx *= p.nclasses; gy = 1; }
    }
#endif
    if (gx > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "igemm_bx6: grid too large");
    hipLaunchKernelGGL((igemm_bx6_kernel<BM, BN, PAR>), dim3((unsigned)gx, gy, 1), dim3(NT), smem, s, q);
    CGS_CHECK_LAUNCH("igemm_bx6");
    static thread_local char name[64];
    snprintf(name, sizeof(name), "igemm_bx6_kernel<%d, %d, %s>", BM, BN, PAR ? "true" : "false");
    cgs_note_kernel(name);
    return CGS_OK;
}

int cgs_igemm_bx6_launch(const IgemmParams& p_in, hipStream_t s, void* dbg, size_t dbg_bytes) {
    IgemmParams p = p_in;
    p.splitk = 1; p.slab = nullptr;
#ifdef BX6_STAMPS
    if (getenv("CGS_STAMP") && dbg && dbg_bytes >= (1u << 20)) p.slab = (float*)((char*)dbg + dbg_bytes - (1u << 20) - ((uintptr_t)((char*)dbg + dbg_bytes) & 15));   // (the last MiB of the workspace)
#endif
    p.prio_t[0] = p.prio_t[1] = p.prio_t[2] = 0;
    p.uni = 0;
    if (!p.vec || (p.N % 64) || p.Np != p.N) return cgs_set_error(CGS_EINVAL, "igemm_bx6: needs Cred %% 32 == 0 and N %% 64 == 0");
    if (p.sign_out && p.epilogue != CGS_EPI_AFFINE_RELU && p.epilogue != CGS_EPI_LRELU)
        return cgs_set_error(CGS_EINVAL, "igemm_bx6: a sign mask needs the relu / lrelu forward epilogues");
#ifndef BX6_SOLO
#define BX6_SOLO 1       // N % 128 != 0: one-wave 128 x 64 blocks (1) or the two-wave 256 x 64 block (0)
#endif
    const bool n256 = (p.N % 256) == 0;
    const int BM = (n256 || (BX6_SOLO && (p.N % 128) != 0)) ? 128 : 256;
    cgs_igemm_row_policy(p, BM);
    if ((long)p.B * p.Hout * p.Wout * p.N * 4 > 0x7fffffffL || (long)p.B * p.Hin * p.Win * p.Cred * 4 > 0x7fffffffL)
        return cgs_set_error(CGS_EINVAL, "igemm_bx6: a tensor of one launch exceeds 2 GiB (the caller splits the batch)");
    if (p.stat_part && p.epilogue != CGS_EPI_NONE) return cgs_set_error(CGS_EINVAL, "igemm_bx6: fused statistics need no epilogue");
    p.stat_cls_rows = 0;
    if (p.stat_part) {
        const long M0 = (long)p.B * p.cls[0].R * p.cls[0].C;
        for (int i = 1; i < p.nclasses; ++i)
            if ((long)p.B * p.cls[i].R * p.cls[i].C != M0) return cgs_set_error(CGS_EINVAL, "igemm_bx6: fused statistics need parity classes of equal size");
        p.stat_cls_rows = (int)(2 * ((M0 + 127) / 128));
    }
    cgs_igemm_count_flops(p, BM);
    if (n256) return p.tap_parity ? launch_bx6<128, 256, true>(p, s) : launch_bx6<128, 256, false>(p, s);
    if ((p.N % 128) == 0) return p.tap_parity ? launch_bx6<256, 128, true>(p, s) : launch_bx6<256, 128, false>(p, s);
#if BX6_SOLO
    return p.tap_parity ? launch_bx6<128, 64, true>(p, s) : launch_bx6<128, 64, false>(p, s);
#else
    return p.tap_parity ? launch_bx6<256, 64, true>(p, s) : launch_bx6<256, 64, false>(p, s);
#endif
}
