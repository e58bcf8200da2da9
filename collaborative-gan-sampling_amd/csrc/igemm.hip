// Implicit-GEMM NHWC convolution family on the gfx950 fp32 matrix cores.
//
// One kernel serves the four contractions of the refinement hot path (conv fwd, conv bwd-data,
// deconv fwd, deconv bwd-data -- sampling/collaborator.py:26-39 through nsgan/ops.py:41,55) and
// the fully connected layers (nsgan/ops.py:81), all with TF 'SAME' geometry:
//
//   out[b, r*So+py, c*So+px, n] = epi( bias[n] + sum_{ta,tb,ci} in[b, r*S+dy0+ta*dstep, c*S+dx0+tb*dstep, ci]
//                                                               * Wp[(ta*ntx+tb)*Cred + ci][n] )
//
// im2col is never materialised: A-tile rows are gathered straight from the NHWC activation (each
// 32-deep K chunk lies inside one tap, so a row chunk is 128 contiguous bytes), staged through LDS
// together with a pre-packed weight tile, and contracted with v_mfma_f32_32x32x2_f32 (exact fp32,
// k-ordered fma chain).  The transposed direction runs as s*s parity classes (blockIdx.y), each a
// dense GEMM over only the taps that hit that output parity (no zero-stuffing).
//
// Tiling: 256 threads = 4 waves (2x2), block tile BM x BN x {32,16}, each wave (BM/2)x(BN/2) as 32x32
// MFMA tiles; double-buffered LDS with the next tile's global loads issued before the MFMA block
// and written after it (register staging), one barrier per K tile; 2 blocks per CU with 32-deep K
// tiles, 4 with 16-deep ones (chosen per launch from the grid size, see cgs_igemm_launch).
//
// K ordering trick: inside a 32-deep chunk lane-half h of MFMA step (jj,e) contracts
// k = 4*(2*jj+h)+e for BOTH operands, so every lane fetches its 4 consecutive k of a row / column
// with one ds_read_b128 (A rows padded to 36 floats and weights packed [k/4][n][4]: both
// conflict-free).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "cgs_internal.h"
#include "igemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = CGS_BK;     // K padding granule of the packed weights (the kernels tile K by 32 or 16)

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------
void cgs_geom_F(const CgsLayer& L, IgemmParams& p) {
    p.Hin = L.Hb; p.Win = L.Wb; p.Cred = L.Cb;
    p.Hout = L.Hs; p.Wout = L.Ws; p.N = L.Cs; p.Np = cgs_round_up(L.Cs, 64);
    p.S = L.sh; p.So = 1; p.dstep = 1; p.kstep = 1; p.nclasses = 1;
    // parity-first tap order (see igemm_kernel) where tiles are runs of neighbouring pixels; the small grids that run
    // pixel-major (one pixel of 128 images per tile, <= 64 base pixels) gain nothing from it and measured 1-4 % slower.
    // Decided from the layer geometry alone, so a packed-weight workspace stays valid for every batch size.
    p.tap_parity = (L.sh == 2 && L.sw == 2 && L.Hs * L.Ws > 64) ? 1 : 0;
    IgemmClass& c = p.cls[0];
    c.R = L.Hs; c.C = L.Ws; c.py = 0; c.px = 0; c.nty = L.kh; c.ntx = L.kw;
    c.dy0 = -cgs_same_pad_before(L.Hb, L.kh, L.sh); c.dx0 = -cgs_same_pad_before(L.Wb, L.kw, L.sw);
    c.ky0 = 0; c.kx0 = 0; c.K = L.kh * L.kw * L.Cb; c.w_off = 0;
    p.vec = (p.Cred % BK) == 0 && c.nty <= 16 && c.ntx <= 16;
}

void cgs_geom_T(const CgsLayer& L, IgemmParams& p) {
    p.Hin = L.Hs; p.Win = L.Ws; p.Cred = L.Cs;
    p.Hout = L.Hb; p.Wout = L.Wb; p.N = L.Cb; p.Np = cgs_round_up(L.Cb, 64);
    p.S = 1; p.So = L.sh; p.dstep = -1; p.kstep = L.sh; p.nclasses = L.sh * L.sw;
    p.tap_parity = 0;      // consecutive taps of a parity class already read neighbouring input pixels
    const int pt = cgs_same_pad_before(L.Hb, L.kh, L.sh), pl = cgs_same_pad_before(L.Wb, L.kw, L.sw);
    int off = 0;
    for (int py = 0; py < L.sh; ++py)
        for (int px = 0; px < L.sw; ++px) {
            IgemmClass& c = p.cls[py * L.sw + px];
            c.py = py; c.px = px;
            c.R = py < L.Hb ? (L.Hb - py + L.sh - 1) / L.sh : 0;
            c.C = px < L.Wb ? (L.Wb - px + L.sw - 1) / L.sw : 0;
            c.ky0 = (py + pt) % L.sh; c.kx0 = (px + pl) % L.sw;
            c.nty = c.ky0 < L.kh ? (L.kh - c.ky0 + L.sh - 1) / L.sh : 0;
            c.ntx = c.kx0 < L.kw ? (L.kw - c.kx0 + L.sw - 1) / L.sw : 0;
            c.dy0 = (py + pt - c.ky0) / L.sh; c.dx0 = (px + pl - c.kx0) / L.sw;
            c.K = c.nty * c.ntx * L.Cs; c.w_off = off;
            off += cgs_round_up(c.K, BK) * p.Np;
        }
    p.vec = (p.Cred % BK) == 0;
    for (int i = 0; i < p.nclasses; ++i) p.vec = p.vec && p.cls[i].nty <= 16 && p.cls[i].ntx <= 16;
    // heaviest class first: blocks are dispatched in blockIdx.y-major order, so the 9-tap class must not be the tail
    for (int i = 1; i < p.nclasses; ++i)
        for (int j = i; j > 0 && p.cls[j].K > p.cls[j - 1].K; --j) { IgemmClass t = p.cls[j]; p.cls[j] = p.cls[j - 1]; p.cls[j - 1] = t; }
}

size_t cgs_packed_floats(const IgemmParams& p) {
    size_t n = 0;
    for (int i = 0; i < p.nclasses; ++i) n += (size_t)cgs_round_up(p.cls[i].K, BK) * p.Np;
    return n;
}

// ------------------------------------------------------------------------------------------------
// weight packing: w[kh][kw][Cb][Cs] -> per class [Kpad/4][Np][4], zero padded
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(IgemmParams p, const float* __restrict__ w, float* __restrict__ packed,
                                    int kw, int Cb, int Cs, int dirT) {
    const IgemmClass& c = p.cls[blockIdx.y];
    const int Kpad = (c.K + BK - 1) / BK * BK;
    const size_t total = (size_t)Kpad * p.Np;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3);
        const int n = (int)((i >> 2) % p.Np);
        const int k = (int)((i >> 2) / p.Np) * 4 + e;
        float v = 0.f;
        if (k < c.K && n < p.N) {
            int t, ci, ta, tb;
            if (p.vec) {      // K order (32-channel chunk, tap, channel in chunk): see igemm_kernel
                const int ntaps = c.nty * c.ntx;
                const int kt = k / BK, chunk = kt / ntaps;
                t = kt - chunk * ntaps; ci = chunk * BK + (k - kt * BK);
                ta = t / c.ntx; tb = t - ta * c.ntx;
                ta = cgs_tap_order(ta, c.nty, p.tap_parity); tb = cgs_tap_order(tb, c.ntx, p.tap_parity);
            } else {
                t = k / p.Cred; ci = k - t * p.Cred;
                ta = t / c.ntx; tb = t - ta * c.ntx;
            }
            const int tap = (c.ky0 + ta * p.kstep) * kw + (c.kx0 + tb * p.kstep);
            const size_t src = dirT ? ((size_t)tap * Cb + n) * Cs + ci      // reduce over Cs, n indexes Cb
                                    : ((size_t)tap * Cb + ci) * Cs + n;     // reduce over Cb, n indexes Cs
            v = w[src];
        }
        packed[c.w_off + i] = v;
    }
}

int cgs_pack_weights(const IgemmParams& p, const CgsLayer& L, bool dirT, const float* w, float* packed, hipStream_t s) {
    size_t mx = 0;
    for (int i = 0; i < p.nclasses; ++i) {
        size_t n = (size_t)cgs_round_up(p.cls[i].K, BK) * p.Np;
        if (n > mx) mx = n;
    }
    if (mx == 0) return CGS_OK;
    int blocks = (int)((mx + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks, p.nclasses), dim3(256), 0, s, p, w, packed, L.kw, L.Cb, L.Cs,
                       dirT ? 1 : 0);
    CGS_CHECK_LAUNCH("pack_weights");
    return CGS_OK;
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
// NW waves per block, arranged 2 (M) x NW/2 (N); wave tile (BM/2) x (BN/(NW/2)).
// PAR: parity-first tap order (a compile-time variant: the extra scalar decode slowed the natural-order layers by ~0.8 %
// when it was a run-time flag).
// NS: the epilogue can also leave the norm-BACKWARD column sums (IgemmParams::ns_*).  A compile-time twin, not a run-time branch of the one
// kernel: the branch's mere presence cost the launches that never take it 6 % (the dominant kernel of the headline 680 -> 724 us per launch in
// alternating processes on one box, profiles/r06_l_nstat_branch_presence_ab.txt) -- so igemm_kernel is compiled without it, exactly as
// before round 6, and only the launches that want the sums run igemm_ns_kernel.
template <int BM, int BN, int NW, bool VEC, int TBK, bool PAR, bool NS>
__device__ __forceinline__ void igemm_body(const IgemmParams& p) {
    constexpr int BK = TBK;                       // K tile (shadows the packing granule; TBK divides it)
    constexpr int LDA = BK + 4;                   // padded A row (floats): keeps the b128 fragment reads conflict-free
    constexpr int NT = 64 * NW;                   // threads per block
    constexpr int WM = BM / 64;                   // waves along M: 2 (128-row tiles) or 4 (the tall 256 x 64 tile: four 64 x 64 wave tiles stacked)
    constexpr int WN = NW / WM;                   // waves along N
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);   // 32x32 MFMA tiles per wave
    constexpr int QPR = BK / 4;                   // k-quads per A row chunk
    constexpr int AI = BM * QPR / NT, BI = BN * QPR / NT;  // float4 staged per thread (A rows / B columns)
    constexpr int AR = NT / QPR;                  // A rows covered per staging pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [2][BM][LDA]
    float* Bs = smem + 2 * BM * LDA;               // [2][BK/4][BN][4]
    constexpr int KLOOP_F = 2 * BM * LDA + 2 * BK * BN;              // floats of the K-loop double buffers
    constexpr int EH = (VEC && TBK == 16) ? 2 : 1;                    // epilogue passes (the 16-deep variant stages half a wave tile at a time)
    constexpr int STAGE_F = NW * (BM / WM / EH) * (BN / WN + 4);      // floats of the epilogue staging tiles
    int* rowpix = (int*)(smem + (KLOOP_F > STAGE_F ? KLOOP_F : STAGE_F));   // [BM] output pixel of each tile row, -1 = out of range

    // block id -> (m-tile, n-tile); class = blockIdx.y (heaviest class first).  Measured and rejected: computing all
    // parity classes of a tile back to back in one block (FETCH_SIZE only -10 %: the patch does not survive in the
    // 4 MiB L2 across a class pass; 2-4 % slower); running the s*s classes of an m-tile as consecutive workgroups of one
    // XCD (FETCH_SIZE of the 32x32 transposed layers 670 -> 270 MB, but +30 % on the small-grid layers whose four weight
    // sets then compete for the L2, and 13 % SLOWER end to end: the kernel is MFMA-bound, its fetch latency is hidden);
    // sweeping runs of 16-64 m-tiles class after class per XCD (670 -> 515 MB on those layers, +-1 % in time: not kept).
    // Workgroups go to the 8 XCDs round-robin by id, each XCD with its own L2.  With id = mb * nblk_n + nb the n-tiles of
    // one m-tile (same A rows) land on DIFFERENT XCDs and every XCD streams the whole input: measured 26x the input bytes
    // from HBM for the 8x8 256->512 layer.  Decode per XCD instead: XCD x = id % 8 owns the m-tiles = x (mod 8) and runs
    // their n-tiles back to back; in the LPT order (m-tile = pixel rank * groups + image group) that also gives an XCD
    // the same image groups for every pixel, so the tap overlap between neighbouring pixels hits in its L2 too.
    const int nblk_n = p.Np / BN;
    unsigned wi = blockIdx.x;
    int cls_i = blockIdx.y;
    if (p.xcd_map == 2) {        // the parity classes of a tile back to back on ONE XCD (ids 8 * (tile * classes + class) + xcd): they read the same input patch
        const unsigned xcd = wi & 7u;
        unsigned q = wi >> 3;
        cls_i = (int)(q % (unsigned)p.nclasses);
        q /= (unsigned)p.nclasses;
        wi = (q << 3) | xcd;
    }
    // Split launches of several parity classes (9 / 6 / 6 / 4 taps: their K slices differ 2.25x in length): the dispatcher deals workgroups to the
    // 256 CUs round-robin in launch order, so with class = blockIdx.y the two (or more) blocks of a CU are the SAME class of K slices z and
    // z + 256 / (blocks per slice) -- heavy CUs and light CUs.  Every other round of 256 blocks takes the classes in reverse order (per K slice, so
    // the (class, slice) pairs stay a permutation): a CU gets a long and a short block.
    // (unsplit launches of two rounds, 128 blocks per class: round 0 holds classes 0, 1 and round 1 classes 2, 3 -- a CU's pair was (0, 2) or (1, 3),
    // 15 against 10 taps; with the second round's classes swapped it is (0, 3) or (1, 2).  cls_flip = classes per group that is reversed.)
    if (p.cls_flip && (((((unsigned)blockIdx.z * gridDim.y + (unsigned)cls_i) * gridDim.x) >> 8) & 1u)) {
        const int grp = cls_i / p.cls_flip;
        cls_i = grp * p.cls_flip + (p.cls_flip - 1 - (cls_i - grp * p.cls_flip));
    }
    // tail split: the block ids from tail_from on (last class) are (tail tile, K slice) pairs -- tail_s consecutive ids per tile
    int tsplit = 1, tz = 0, ttile = 0;
    if (p.tail_s > 1 && cls_i == p.nclasses - 1 && wi >= (unsigned)p.tail_from) {
        const unsigned t = wi - (unsigned)p.tail_from;
        ttile = (int)(t / (unsigned)p.tail_s); tz = (int)(t - (unsigned)ttile * (unsigned)p.tail_s); tsplit = p.tail_s;
        if (ttile >= p.tail_n) return;
        wi = (unsigned)p.tail_from + (unsigned)ttile;
    }
    int nb, mb;
    if (p.xcd_map) {
        const unsigned xcd = wi & 7u, q = wi >> 3;
        nb = (int)(q % (unsigned)nblk_n);
        mb = (int)(q / (unsigned)nblk_n) * 8 + (int)xcd;
    } else {          // few m-tiles (fc layers, small batches): spread the n-tiles over the XCDs instead of idling most of them
        nb = (int)(wi % (unsigned)nblk_n);
        mb = (int)(wi / (unsigned)nblk_n);
    }
    const IgemmClass& c = p.cls[cls_i];
    const int RC = c.R * c.C;
    const int M = p.B * RC;
    if (mb * BM >= M) return;
    if (p.lpt) {     // B % BM == 0: m-tile = (pixel rank, image group); visit pixels heaviest-first (LPT schedule)
        const int gpp = p.B / BM;                       // image groups (tiles) per pixel
        const int rank = mb / gpp, grp = mb - rank * gpp;
        mb = (int)p.perm[cls_i][rank] * gpp + grp;
    }
    const int m0 = mb * BM, n0 = nb * BN;

    const int tid = threadIdx.x;
    // Plain VALU / memory instructions of a wave issue only in the gaps of the MFMA stream of the OTHER waves on its SIMD (the
    // hardware arbitrates by priority, then age; in-kernel stamps: ~100 epilogue VALU ops take thousands of cycles next to a
    // resident block in its K loop).  Prologue and epilogue therefore run at priority 3 and the K loop below them: the short
    // phases finish at once and cost the matrix stream a few cycles per instruction.
    __builtin_amdgcn_s_setprio(3);
#ifdef CGS_DIAG_STAMPS      // diagnostic build only (tools/clock_probe.py): per-block timeline stamps into the workspace tail
    const bool stamp = p.slab != nullptr && p.splitk == 1 && tid == 0;
    unsigned long long sr_in = 0;
    if (stamp) sr_in = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int h = lane >> 5, j = lane & 31;

    // GEMM row m -> (image b, base pixel r, cc).  Image-major: m = b*RC + pixel (rows of a tile are neighbouring
    // pixels: their taps overlap, L1/L2 reuse).  Pixel-major: m = pixel*B + b (rows of a tile are the SAME pixel of
    // 128 images: a tap that falls in the zero padding does so for the whole tile and is skipped, below).
    // One-pixel tiles (pixel-major and every row of the tile inside one base pixel -- all tiles of an lpt launch): the pixel is a
    // property of the BLOCK, so its decode is two scalar divisions once instead of two vector divisions per row and thread (the
    // prologue ran ~500 VALU instructions per thread, half of them these; it runs at priority 3 in front of the other blocks' MFMAs)
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

    if (tid < BM) {
        const int m = m0 + tid;
        int pix = -1;
        if (m < M) {
            int b, r, cc;
            DECODE_ROW(m, b, r, cc);
            pix = (b * p.Hout + r * p.So + c.py) * p.Wout + cc * p.So + c.px;
        }
        rowpix[tid] = pix;
    }

    // A staging: thread -> k-quad aq of rows ar + 32*i
    const int aq = tid % QPR, ar = tid / QPR;
    int a_base[AI], a_iy[AI], a_ix[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + ar + AR * i;
        if (m < M) {
            int b, r, cc;
            DECODE_ROW(m, b, r, cc);
            a_base[i] = b * p.Hin * p.Win; a_iy[i] = r * p.S + c.dy0; a_ix[i] = cc * p.S + c.dx0;
        } else {
            a_base[i] = 0; a_iy[i] = -(1 << 20); a_ix[i] = 0;    // always out of range -> zeros
        }
    }
    const float* wsrc = p.wp + c.w_off;
    const int nk_all = (c.K + BK - 1) / BK;
    // split-K: this block owns K tiles [kbeg, nk)
    int kbeg = 0, nk = nk_all;
    if (p.splitk > 1) {
        const int cps = (nk_all + p.splitk - 1) / p.splitk;
        kbeg = blockIdx.z * cps;
        nk = kbeg + cps < nk_all ? kbeg + cps : nk_all;
    } else if (tsplit > 1) {
        const int cps = (nk_all + tsplit - 1) / tsplit;
        kbeg = tz * cps;
        if (kbeg > nk_all) kbeg = nk_all;
        nk = kbeg + cps < nk_all ? kbeg + cps : nk_all;
    }

    // zero-tap skipping (pixel-major, VEC): if every row of this tile is the same base pixel, a tap outside the
    // image contributes exact zeros for all rows -> its Cred/32 K-chunks are not loaded or multiplied at all.
    const bool skip_ok = VEC && one_pix;
    const int u_iy = t_r * p.S + c.dy0, u_ix = t_cc * p.S + c.dx0;      // (used only with skip_ok)
    // VEC K order: K tile kt = chunk * ntaps + tap, i.e. the taps are the INNER loop of each 32-channel chunk.
    // The 25 (9/6/6/4) taps of a tile re-read one input patch; with the taps inner only a 32-channel slice of the
    // patch is live at a time, so the working set of an XCD's resident blocks fits its 4 MiB L2 and the tap
    // re-reads hit there instead of going out to the Infinity Cache / HBM.
    // Stride-2 forward direction: the taps are visited even offsets first, then odd, per axis (cgs_tap_order).  Taps of one
    // (row, column) parity read the SAME input pixels shifted by whole output pixels, taps of different parity read
    // disjoint pixels, so a block's live set is one parity quarter of its patch (23 KB instead of 78 KB per 32 channels)
    // and the quarter is used up before the next one is touched.
    const int ntaps = c.nty * c.ntx;
    // first K tile >= kt whose tap is inside the image for this tile (nk if none)
    auto next_chunk = [&](int kt) -> int {
        if (!skip_ok) return kt;
        while (kt < nk) {
            const int t = (VEC && TBK == 16 ? kt >> 1 : kt) % ntaps;     // 16-deep tiles: two per (32-channel chunk, tap)
            const int ia = t / c.ntx, ib = t - ia * c.ntx;
            const int ta = cgs_tap_order(ia, c.nty, PAR), tb = cgs_tap_order(ib, c.ntx, PAR);
            const int iy = u_iy + ta * p.dstep, ix = u_ix + tb * p.dstep;
            if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) break;
            ++kt;
        }
        return kt < nk ? kt : nk;
    };

    f32x4 ra[AI], rb[BI];
    // buffer descriptor over the input tensor (wave-uniform: built from kernel arguments only)
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * (unsigned)p.Cred * 4u), 0x00020000);
    // global -> registers for K tile kt_ (issued early; consumed by STORE_TILE after the MFMA block)
#define LOAD_TILE(kt_)                                                                                          \
    do {                                                                                                        \
        if constexpr (VEC) {                                                                                    \
            const int kt32_ = TBK == 16 ? (kt_) >> 1 : (kt_);        /* K order is packed in 32-channel granules */     \
            const int chunk_ = kt32_ / ntaps, t = kt32_ - chunk_ * ntaps;                                       \
            const int ci = chunk_ * 32 + (TBK == 16 ? ((kt_) & 1) * 16 : 0) + aq * 4;                           \
            const int ia_ = t / c.ntx, ib_ = t - ia_ * c.ntx;                                                   \
            const int ta = cgs_tap_order(ia_, c.nty, PAR), tb = cgs_tap_order(ib_, c.ntx, PAR);                 \
            const int dy = ta * p.dstep, dx = tb * p.dstep;                                                     \
            _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                    \
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;                                                 \
                const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;               \
                /* branch-free zero padding: out-of-range rows use an offset past the buffer's num_records, */  \
                /* for which the hardware bounds check of buffer_load returns 0 */                              \
                const unsigned off = ok ? (unsigned)((a_base[i] + iy * p.Win + ix) * p.Cred + ci) * 4u : 0xFFFFFFF0u; \
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));   \
            }                                                                                                   \
        } else {                                                                                                \
            /* generic K (Cred not a multiple of 32, e.g. 3 image channels): decode each of this thread's */   \
            /* 4 k's once per tile (independent of the row), then gather element-wise */                        \
            float va[AI][4];                                                                                    \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                     \
                const int k = (kt_) * BK + aq * 4 + e;                                                          \
                const bool kin = k < c.K;                                                                       \
                const int t = kin ? k / p.Cred : 0, ci = k - t * p.Cred;                                        \
                const int ta = t / c.ntx, tb = t - ta * c.ntx;                                                  \
                const int dy = ta * p.dstep, dx = tb * p.dstep;                                                 \
                _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                \
                    const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;                                             \
                    const bool ok = kin && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;    \
                    const unsigned off = ok ? (unsigned)((a_base[i] + iy * p.Win + ix) * p.Cred + ci) * 4u : 0xFFFFFFF0u; \
                    va[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off, 0, 0)); \
                }                                                                                               \
            }                                                                                                   \
            _Pragma("unroll") for (int i = 0; i < AI; ++i) ra[i] = f32x4{va[i][0], va[i][1], va[i][2], va[i][3]}; \
        }                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < BI; ++i) {                                                        \
            const int idx = tid + NT * i;                                                                       \
            const int kq = idx / BN, n = idx - kq * BN;                                                         \
            rb[i] = *(const f32x4*)(wsrc + ((size_t)((kt_) * (BK / 4) + kq) * p.Np + n0 + n) * 4);            \
        }                                                                                                       \
    } while (0)
#define STORE_TILE(buf_)                                                                                        \
    do {                                                                                                        \
        float* a_ = As + (buf_) * BM * LDA;                                                                     \
        float* b_ = Bs + (buf_) * BK * BN;                                                                      \
        _Pragma("unroll") for (int i = 0; i < AI; ++i) *(f32x4*)(a_ + (ar + AR * i) * LDA + aq * 4) = ra[i];    \
        _Pragma("unroll") for (int i = 0; i < BI; ++i) *(f32x4*)(b_ + (tid + NT * i) * 4) = rb[i];              \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

#ifdef CGS_DIAG_STAMPS
    unsigned long long sr_loop0 = 0;
    if (stamp) sr_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ------------------------------------------------------------------------------------------------------------------
    // K loop.  Per-tile vector-ALU work is kept to a minimum: VALU instructions and MFMAs of a SIMD share its vector issue, and
    // every VALU op in the loop costs matrix time even from another wave (tools/probe/mfma_probe.hip: 32 MFMAs + 8 ds_read_b128
    // per wave and K tile run at 151 TFLOP/s; 16 v_mul_lo_u32 more per tile and wave 140; 128 plain VALU ops 130).  So for the
    // VEC kernels the position in K is a wave-uniform ITERATOR (chunk, tap row, tap column, half) advanced with scalar counters
    // instead of being re-derived from the tile index by divisions, and a tile's addresses are
    //   A row i :  rowoff[i] (VGPR, loop invariant)  +  soff (SGPR: tap offset + channel offset)      -> 1 v_add + bounds select
    //   B       :  b_voff[i] (VGPR, loop invariant)  +  kt * tile bytes as the buffer load's SCALAR offset -> no VALU at all
    // (33 VALU ops per tile incl. 4 v_mul_lo_u32, 2 v_mad_i64_i32 and 120 SALU ops before; 1-round layers +4..6 %).
    // ------------------------------------------------------------------------------------------------------------------
    struct KIt { int kt, sub, ia, ib, chunk; };          // K tile index and its decode (VEC): 32-channel chunk, tap (ia, ib) in visiting order, 16-deep half
    auto kit_valid = [&](const KIt& s) -> bool {         // is the tap inside the image for this (one-pixel) tile?
        if (!skip_ok) return true;
        const int ta = cgs_tap_order(s.ia, c.nty, PAR), tb = cgs_tap_order(s.ib, c.ntx, PAR);
        const int iy = u_iy + ta * p.dstep, ix = u_ix + tb * p.dstep;
        return (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
    };
    auto kit_next_tap = [&](KIt& s) {                    // s sits on a tap start: move to the next executed tap start (kt = nk when none)
        for (;;) {
            if (++s.ib == c.ntx) { s.ib = 0; if (++s.ia == c.nty) { s.ia = 0; ++s.chunk; } }
            if (s.kt >= nk || kit_valid(s)) break;
            s.kt += (TBK == 16 ? 2 : 1);
        }
        if (s.kt > nk) s.kt = nk;
    };
    auto kit_first = [&](int kt0) -> KIt {               // first executed tile >= kt0 (the only place that divides)
        KIt s;
        const int kt32 = (VEC && TBK == 16) ? kt0 >> 1 : kt0;
        s.kt = kt0; s.sub = (VEC && TBK == 16) ? (kt0 & 1) : 0;
        s.chunk = kt32 / ntaps;
        const int t = kt32 - s.chunk * ntaps;
        s.ia = t / c.ntx; s.ib = t - s.ia * c.ntx;
        if (s.kt < nk && !kit_valid(s)) {
            s.kt += (TBK == 16 ? 2 - s.sub : 1); s.sub = 0;
            kit_next_tap(s);
        }
        if (s.kt > nk) s.kt = nk;
        return s;
    };
    auto kit_next = [&](KIt s) -> KIt {                  // the tile executed after s
        if (TBK == 16 && s.sub == 0) { s.sub = 1; ++s.kt; if (s.kt > nk) s.kt = nk; return s; }
        s.sub = 0; ++s.kt;
        kit_next_tap(s);
        return s;
    };
    // loop-invariant per-thread address parts (VEC)
    unsigned rowoff[AI], b_voff[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i)
        rowoff[i] = ((unsigned)(a_base[i] + a_iy[i] * p.Win + a_ix[i]) * (unsigned)p.Cred + (unsigned)aq * 4u) * 4u;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int idx = tid + NT * i;
        const int kq = idx / BN, n = idx - kq * BN;
        b_voff[i] = (unsigned)(kq * p.Np + n0 + n) * 16u;
    }
    // which taps of row i lie inside the image: bit ia (tap row, visiting order) and bit 16 + ib (tap column) -- the per-tile
    // bounds check is then one v_and + one v_cmp against a scalar (VEC kernels have at most 16 taps per axis, see cgs_geom_*)
    unsigned tapmask[AI];
    if (one_pix) {       // the same mask for every row of the tile: built once on the scalar unit (rows past M: no tap)
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
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 0x7ffffff0, 0x00020000);
    const int b_tile_bytes = (BK / 4) * p.Np * 16;
    unsigned a_off[AI];
    // VGPR byte offsets of tile s_'s A rows (out-of-range rows -> past num_records: the hardware bounds check returns zeros)
#define ADDR_TILE_V(s_)                                                                                         \
    do {                                                                                                        \
        const int ta_ = cgs_tap_order((s_).ia, c.nty, PAR), tb_ = cgs_tap_order((s_).ib, c.ntx, PAR);           \
        const int dy_ = ta_ * p.dstep, dx_ = tb_ * p.dstep;                                                     \
        const unsigned soff_ = (unsigned)(((dy_ * p.Win + dx_) * p.Cred + (s_).chunk * 32 + (TBK == 16 ? (s_).sub * 16 : 0)) * 4); \
        const unsigned need_ = (1u << (s_).ia) | (1u << (16 + (s_).ib));                                        \
        _Pragma("unroll") for (int i = 0; i < AI; ++i)                                                          \
            a_off[i] = (tapmask[i] & need_) == need_ ? rowoff[i] + soff_ : 0xFFFFFFF0u;                         \
    } while (0)
#define ISSUE_TILE_V(s_)                                                                                        \
    do {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < AI; ++i)                                                          \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, a_off[i], 0, 0));  \
        const int b_soff_ = (s_).kt * b_tile_bytes;                                                             \
        _Pragma("unroll") for (int i = 0; i < BI; ++i)                                                          \
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff[i], b_soff_, 0)); \
    } while (0)

    // Uniform tiles (pixel-major, all BM rows are the same base pixel of BM different images, no ragged tail): the taps the
    // iterator visits are inside the image for EVERY row, so no per-row bounds select is needed, and the tap / channel offset is
    // the same for all rows -> it rides in the buffer load's SCALAR offset and the row part is a loop-invariant VGPR: the A
    // addresses of a tile cost no vector-ALU instruction at all (9 VALU per tile in the general form above).  The row part is
    // the image origin (>= 0: the hardware range check sees the VGPR offset), the scalar part the whole pixel offset.
    const bool uni = VEC && p.uni && skip_ok && m0 + BM <= M;
    unsigned rowoff_u[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) rowoff_u[i] = ((unsigned)a_base[i] * (unsigned)p.Cred + (unsigned)aq * 4u) * 4u;
    unsigned a_soff = 0;
#define ADDR_TILE_U(s_)                                                                                         \
    do {                                                                                                        \
        const int ta_ = cgs_tap_order((s_).ia, c.nty, PAR), tb_ = cgs_tap_order((s_).ib, c.ntx, PAR);           \
        const int iy_ = u_iy + ta_ * p.dstep, ix_ = u_ix + tb_ * p.dstep;                                       \
        a_soff = (unsigned)(((iy_ * p.Win + ix_) * p.Cred + (s_).chunk * 32 + (TBK == 16 ? (s_).sub * 16 : 0)) * 4); \
    } while (0)
#define ISSUE_TILE_U(s_)                                                                                        \
    do {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < AI; ++i)                                                          \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, rowoff_u[i], a_soff, 0)); \
        const int b_soff_ = (s_).kt * b_tile_bytes;                                                             \
        _Pragma("unroll") for (int i = 0; i < BI; ++i)                                                          \
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff[i], b_soff_, 0)); \
    } while (0)

    // prologue: first tile -> LDS buffer 0
    KIt cur;
    int kt;                                            // (generic-K path: plain tile index)
    if constexpr (VEC) {
        cur = kit_first(kbeg);
        kt = cur.kt;
        if (cur.kt < nk) {
            ADDR_TILE_V(cur);
            ISSUE_TILE_V(cur);
            STORE_TILE(0);
        }
    } else {
        kt = next_chunk(kbeg);
        cur.kt = kt;
        if (kt < nk) {
            LOAD_TILE(kt);
            STORE_TILE(0);
        }
    }
    __syncthreads();

    // One basic block per K tile (no branches inside): the next tile's global loads are issued after the first MFMA
    // group and staged to LDS before the last one, so the scheduler can slot them into the shadow of the 64-cycle MFMAs
    // instead of running them as a separate phase (the two blocks resident on a CU run in lockstep, so a separate
    // non-MFMA phase of one coincides with the other's and the matrix pipe idles).  After the last tile the "next"
    // tile is a harmless reload of that last tile into the dead buffer.
#define MFMA_GROUP(jj_)                                                                                         \
    do {                                                                                                        \
        const int kq = 2 * (jj_) + h;                                                                           \
        f32x4 fa[TM], fb[TN];                                                                                   \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) fa[tm] = *(const f32x4*)(a + tm * 32 * LDA + kq * 4); \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) fb[tn] = *(const f32x4*)(b + (kq * BN + tn * 32) * 4); \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                       \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                 \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm].x, fb[tn].x, acc[tm][tn], 0, 0, 0);   \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm].y, fb[tn].y, acc[tm][tn], 0, 0, 0);   \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm].z, fb[tn].z, acc[tm][tn], 0, 0, 0);   \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[tm].w, fb[tn].w, acc[tm][tn], 0, 0, 0);   \
            }                                                                                                   \
    } while (0)
    constexpr int NG = BK / 8;                       // MFMA groups per tile
#ifdef CGS_DIAG_STAMPS
    int diag_tiles = 0;
#endif
    // Fair progress among the blocks of a CU: the hardware arbitrates MFMA issue by priority, then AGE, so of the (equally old
    // or not) resident blocks the oldest runs ahead and finishes first, and the launch ends with one block per CU that
    // has most of its work left and nobody to hide its latencies.  Every block therefore starts at priority 3 and steps down as
    // it passes fixed fractions of its own K loop: leaders wait for the laggards at each step and all finish together.
    int pt1 = 1 << 30, pt2 = 1 << 30;
    if (p.prio_t[0] > 0) {
        const int span = nk - kbeg;
        pt1 = kbeg + ((span * p.prio_t[0]) >> 8); pt2 = kbeg + ((span * p.prio_t[1]) >> 8);
        __builtin_amdgcn_s_setprio(2);               // steps 2 -> 1 -> 0 -> 0 (3 is kept for prologues / epilogues)
    } else {
        __builtin_amdgcn_s_setprio(0);
    }
#define PRIO_STEP(kt_)                                                                                          \
    if ((kt_) >= pt1) {        /* (scalar compares; one s_setprio when a threshold is crossed) */               \
        if ((kt_) >= pt2) { __builtin_amdgcn_s_setprio(0); pt1 = 1 << 30; }                                     \
        else { __builtin_amdgcn_s_setprio(1); pt1 = pt2; }                                                      \
    }

    // Software-pipelined across the barrier (the 32-deep VEC kernels: two blocks per CU, i.e. two waves per SIMD, which cannot
    // cover each other's LDS / global latencies the way the four of the 16-deep variant do; measured: 8x8 256<-512 backward-data
    // 717 -> 642 us.  On the 16-deep kernels the second fragment set costs the fourth resident block: -6 %, not used there):
    // the fragments of MFMA group g+1 are read from LDS BEFORE the MFMAs of group g are issued; the LAST group of a tile is
    // issued AFTER the barrier, behind the first fragment read of the next tile; the next tile's global loads go out at the top
    // of the iteration (their addresses were computed under the previous tile's last MFMA group) and are staged to LDS just
    // before the barrier.  sched_barrier pins the phase order: left alone, the compiler sinks the loads to their first use and
    // re-uses the fragment registers, which serialises everything again.
#define FRAG_READ(buf_, jj_, fa_, fb_)                                                                          \
    do {                                                                                                        \
        const float* a_ = As + (buf_) * BM * LDA + (wm * (BM / WM) + j) * LDA;                                   \
        const float* b_ = Bs + (buf_) * BK * BN + (wn * (BN / WN) + j) * 4;                                     \
        const int kq = 2 * (jj_) + h;                                                                           \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) fa_[tm] = *(const f32x4*)(a_ + tm * 32 * LDA + kq * 4); \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) fb_[tn] = *(const f32x4*)(b_ + (kq * BN + tn * 32) * 4); \
    } while (0)
#define MFMA_EXEC(fa_, fb_)                                                                                     \
    do {                                                                                                        \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                       \
            _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                 \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[tm].x, fb_[tn].x, acc[tm][tn], 0, 0, 0); \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[tm].y, fb_[tn].y, acc[tm][tn], 0, 0, 0); \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[tm].z, fb_[tn].z, acc[tm][tn], 0, 0, 0); \
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[tm].w, fb_[tn].w, acc[tm][tn], 0, 0, 0); \
            }                                                                                                   \
    } while (0)
    if constexpr (VEC && TBK == 32) {
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        // (field-wise selects: a ternary over the whole struct sends it through scratch memory)
#define KIT_SEL(d_, c_, a_, b_)                                                                                 \
    do {                                                                                                        \
        const bool c__ = (c_);                                                                                  \
        d_.kt = c__ ? a_.kt : b_.kt; d_.sub = c__ ? a_.sub : b_.sub; d_.ia = c__ ? a_.ia : b_.ia;               \
        d_.ib = c__ ? a_.ib : b_.ib; d_.chunk = c__ ? a_.chunk : b_.chunk;                                      \
    } while (0)
        // (the loop body as a macro over the addressing form: the uniform-tile form -- all rows of the tile one base pixel, the
        // whole tap / channel offset in the buffer loads' scalar operand, no per-row bounds select -- serves the pixel-major
        // launches here exactly as in the 16-deep kernels below)
#define K_LOOP32(ADDR_, ISSUE_)                                                                                 \
    {                                                                                                           \
        KIt nxt = cur, ld = cur;                                                                                \
        if (cur.kt < nk) nxt = kit_next(cur);                                                                   \
        KIT_SEL(ld, nxt.kt < nk, nxt, cur);          /* tile to prefetch (the current one again after the last) */ \
        if (cur.kt < nk) {                                                                                      \
            FRAG_READ(0, 0, fa0, fb0);                                                                          \
            ADDR_(ld);                                                                                          \
        }                                                                                                       \
        for (int buf = 0; cur.kt < nk; buf ^= 1) {                                                              \
            DIAG_TILES32;                                                                                       \
            ISSUE_(ld);                                                                                         \
            _Pragma("unroll") for (int jj = 0; jj + 1 < NG; jj += 2) {     /* NG is even: fragments ping-pong between two register sets */ \
                FRAG_READ(buf, jj + 1, fa1, fb1);                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                MFMA_EXEC(fa0, fb0);                                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                if (jj + 2 < NG) {                                                                              \
                    FRAG_READ(buf, jj + 2, fa0, fb0);                                                           \
                    __builtin_amdgcn_sched_barrier(0);                                                          \
                    MFMA_EXEC(fa1, fb1);                                                                        \
                    __builtin_amdgcn_sched_barrier(0);                                                          \
                }                                                                                               \
            }                                                                                                   \
            STORE_TILE(buf ^ 1);                                                                                \
            __syncthreads();                                                                                    \
            FRAG_READ(buf ^ 1, 0, fa0, fb0);         /* first fragments of the next tile (a harmless re-read after the last one) */ \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            /* last MFMA group of this tile + the address arithmetic of the tile after next */                  \
            cur = nxt;                                                                                          \
            if (cur.kt < nk) {                                                                                  \
                nxt = kit_next(cur);                                                                            \
                KIT_SEL(ld, nxt.kt < nk, nxt, cur);                                                             \
            }                                                                                                   \
            ADDR_(ld);                                                                                          \
            MFMA_EXEC(fa1, fb1);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
    }
#ifdef CGS_DIAG_STAMPS
#define DIAG_TILES32 ++diag_tiles
#else
#define DIAG_TILES32
#endif
        if (uni) K_LOOP32(ADDR_TILE_U, ISSUE_TILE_U)
        else K_LOOP32(ADDR_TILE_V, ISSUE_TILE_V)
#undef K_LOOP32
#undef DIAG_TILES32
    } else if constexpr (VEC) {
        // Two copies of the body, one per LDS buffer: the buffer offsets are then instruction immediates instead of a VALU add
        // per address and tile.  The pair loop has ONE exit (at its top, on a two-tile look-ahead), an odd last tile runs in a
        // third copy after it: with an exit between the two copies the compiler moves all 64 accumulator registers at it.
#define TILE_BODY(BUF_, NXT_, ADDR_, ISSUE_)                                                                   \
    {                                                                                                           \
        PRIO_STEP(cur.kt);                                                                                      \
        KIt ld = NXT_;                                  /* tile to prefetch (the current one again after the last) */ \
        if ((NXT_).kt >= nk) ld = cur;                                                                          \
        const float* a = As + (BUF_) * BM * LDA + (wm * (BM / WM) + j) * LDA;                                    \
        const float* b = Bs + (BUF_) * BK * BN + (wn * (BN / WN) + j) * 4;                                      \
        MFMA_GROUP(0);                                                                                          \
        ADDR_(ld);                                                                                              \
        ISSUE_(ld);                                                                                             \
        _Pragma("unroll") for (int jj = 1; jj < NG - 1; ++jj) MFMA_GROUP(jj);                                   \
        STORE_TILE((BUF_) ^ 1);                                                                                 \
        if (NG > 1) MFMA_GROUP(NG - 1);                                                                         \
        __syncthreads();                                                                                        \
        cur = NXT_;                                                                                             \
    }
#ifdef CGS_DIAG_STAMPS
#define DIAG_TILES(n_) diag_tiles += (n_)
#else
#define DIAG_TILES(n_)
#endif
#define K_LOOP16(ADDR_, ISSUE_)                                                                                 \
    {                                                                                                           \
        KIt n1 = cur;                                                                                           \
        if (cur.kt < nk) n1 = kit_next(cur);                                                                    \
        while (n1.kt < nk) {                             /* at least two tiles left: cur (in buffer 0) and n1 */ \
            const KIt n2 = kit_next(n1);                                                                        \
            KIt n3 = n2;                                                                                        \
            if (n2.kt < nk) n3 = kit_next(n2);                                                                  \
            DIAG_TILES(2);                                                                                      \
            TILE_BODY(0, n1, ADDR_, ISSUE_);                                                                    \
            TILE_BODY(1, n2, ADDR_, ISSUE_);                                                                    \
            n1 = n3;                                                                                            \
        }                                                                                                       \
        if (cur.kt < nk) {                               /* an odd last tile */                                 \
            DIAG_TILES(1);                                                                                      \
            TILE_BODY(0, n1, ADDR_, ISSUE_);                                                                    \
        }                                                                                                       \
    }
        if (uni) K_LOOP16(ADDR_TILE_U, ISSUE_TILE_U)
        else K_LOOP16(ADDR_TILE_V, ISSUE_TILE_V)
#undef K_LOOP16
#undef DIAG_TILES
#undef TILE_BODY
    } else {
        for (int buf = 0; kt < nk; buf ^= 1) {
#ifdef CGS_DIAG_STAMPS
            ++diag_tiles;
#endif
            const int kn = next_chunk(kt + 1);
            const int kl = kn < nk ? kn : kt;            // tile to prefetch (the current one again after the last)
            const float* a = As + buf * BM * LDA + (wm * (BM / WM) + j) * LDA;
            const float* b = Bs + buf * BK * BN + (wn * (BN / WN) + j) * 4;
            MFMA_GROUP(0);
            LOAD_TILE(kl);
#pragma unroll
            for (int jj = 1; jj < NG - 1; ++jj) MFMA_GROUP(jj);
            STORE_TILE(buf ^ 1);
            if (NG > 1) MFMA_GROUP(NG - 1);
            __syncthreads();
            kt = kn;
        }
    }
#undef KIT_SEL
#undef FRAG_READ
#undef MFMA_EXEC
#undef ADDR_TILE_V
#undef ISSUE_TILE_V
#undef ADDR_TILE_U
#undef ISSUE_TILE_U
#undef PRIO_STEP
#undef MFMA_GROUP

#ifdef CGS_DIAG_STAMPS
    unsigned long long sr_loop1 = 0;
    if (stamp) sr_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
#undef LOAD_TILE
#undef STORE_TILE
#undef DECODE_ROW
    __builtin_amdgcn_s_setprio(3);      // epilogue: retire quickly, the slot is what the next block (or kernel) is waiting for
    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (tsplit > 1) {        // raw partial tile -> slab[tail tile][K slice][BM][BN]; tail_reduce_kernel finishes the tile
        float* slab = p.slab + ((size_t)ttile * tsplit + tz) * (BM * BN);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int nl = wn * (BN / WN) + tn * 32 + j;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * (BM / WM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    slab[ml * BN + nl] = acc[tm][tn][r];
                }
        }
        return;
    }
    if (p.splitk > 1) {      // raw partial tile -> slab[class][split][m][Np]; bias / epilogue happen in the reduce kernel
        float* slab = p.slab + c.slab_off + (size_t)blockIdx.z * M * p.Np;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * (BN / WN) + tn * 32 + j;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (BM / WM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < M) slab[(size_t)m * p.Np + n] = acc[tm][tn][r];
                }
        }
        return;
    }
    constexpr int WTN = BN / WN;                 // columns of a wave tile
    if ((p.N & 3) == 0) {
        // Wide epilogue: the accumulators hold column strips (one column per lane); stage the wave tile through
        // LDS (the K-loop buffers are free now) so that each lane owns 4 consecutive channels of a row and
        // the aux loads / output stores are 16 bytes per lane, 256 contiguous bytes per 16 lanes: 4x fewer
        // store (and aux load) instructions than the per-register scalar form below.
        constexpr int LDE = WTN + 4;
        constexpr int ER = BM / WM / EH;          // rows staged per pass
        float* E = smem + wave * ER * LDE;        // this wave's [ER][LDE] staging tile (launch_cfg sizes the LDS for it)
        constexpr int LPR = WTN / 4;              // lanes per row (16 for 64 columns, 8 for 32)
        constexpr int RPP = 64 / LPR;             // rows per pass
        const int c4 = (lane % LPR) * 4, rsub = lane / LPR;
        const int n = n0 + wn * WTN + c4;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, ea = {1.f, 1.f, 1.f, 1.f}, eb = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) {
            if (p.bias) bias = *(const f32x4*)(p.bias + n);
            if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = *(const f32x4*)(p.ep_a + n); eb = *(const f32x4*)(p.ep_b + n); }
            if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = *(const f32x4*)(p.ep_a + n);
        }
        f32x4 st_a = {0.f, 0.f, 0.f, 0.f}, st_b = {0.f, 0.f, 0.f, 0.f};
        // all waves are past the loop's final barrier: the K-loop buffers are dead (rowpix lives behind the staging area)
#pragma unroll
        for (int ph = 0; ph < EH; ++ph) {
#pragma unroll
            for (int tm = ph * (TM / EH); tm < (ph + 1) * (TM / EH); ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        E[((tm - ph * (TM / EH)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + tn * 32 + j] = acc[tm][tn][r];
            // the same wave reads what it wrote: LDS ops of one wave complete in order; only the compiler must keep it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (n < p.N) {
                const int* rp = rowpix + wm * (BM / WM) + ph * ER;
                if constexpr (NS) {               // norm-BACKWARD statistics of the gradient this launch produces (IgemmParams::ns_*; the launcher picks this twin for them only)
                    // a wave's 64 rows lie in ONE statistics group (cgs_conv_stat_layout admits only such launches): its parameters once per wave and pass
                    // (a wave whose 64 rows all lie past M -- the second half of a last tile with M % 128 == 64 -- stores nothing, but its parameter
                    // loads must still hit a real group: row M - 1's.  Found by the topology fuzz: batch 3 x 64 pixels read the statistics of "sample 3")
                    const int mw = min(m0 + wm * (BM / WM), M - 1);
                    const int grp = p.ns_gimg <= 0 ? 0 : (p.pix_major ? (mw % p.B) : (mw / RC)) / p.ns_gimg;
                    const NsLane ns = ns_lane_params(p, grp, n);
                    // (x loads of four rows in flight only in the 32-deep kernels -- the few-block launches, where a row's memory latency is exposed and the
                    // registers are free (167 either way); in the 16-deep kernels the group's registers cost the fourth resident block: 140 against 122)
                    epilogue_rows<CGS_EPI_NONE, ER, RPP, LDE, 2, ((VEC && TBK == 32) ? 4 : 1)>(p, E, rp, rsub, c4, n, bias, ea, eb, &st_a, &st_b, &ns);
                } else if (p.stat_part) {        // (only with CGS_EPI_NONE: the statistics are those of the stored tensor)
                    epilogue_rows<CGS_EPI_NONE, ER, RPP, LDE, 1>(p, E, rp, rsub, c4, n, bias, ea, eb, &st_a, &st_b);
                } else if (p.sign_out) {  // (N % 32 == 0: every lane of the wave is inside N, the ballots see whole rows)
                    if (p.epilogue == CGS_EPI_AFFINE_RELU) epilogue_rows_signs<CGS_EPI_AFFINE_RELU, ER, RPP, LDE, true>(p, E, rp, rsub, c4, n, bias, ea, eb);
                    else epilogue_rows_signs<CGS_EPI_LRELU, ER, RPP, LDE, true>(p, E, rp, rsub, c4, n, bias, ea, eb);
                } else
                switch (p.epilogue) {     // wave-uniform; each case is one compact branch-free row loop
                    case CGS_EPI_NONE: epilogue_rows<CGS_EPI_NONE, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                    case CGS_EPI_LRELU: epilogue_rows<CGS_EPI_LRELU, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                    case CGS_EPI_AFFINE_RELU: epilogue_rows<CGS_EPI_AFFINE_RELU, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                    case CGS_EPI_TANH: epilogue_rows<CGS_EPI_TANH, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                    case CGS_EPI_RELU_BWD_AFFINE: epilogue_rows<CGS_EPI_RELU_BWD_AFFINE, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                    case CGS_EPI_LRELU_BWD: epilogue_rows<CGS_EPI_LRELU_BWD, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                    default: epilogue_rows<CGS_EPI_TANH_BWD, ER, RPP, LDE>(p, E, rp, rsub, c4, n, bias, ea, eb); break;
                }
            }
            if (ph + 1 < EH) {            // the next pass overwrites the staging tile this wave has just read
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        if (p.stat_part) {
            // Fused batch-norm statistics (the producing conv leaves per-block column sums; bn_finalize adds them in a fixed order
            // in double): the lanes that share a column group differ in their row sub-group -> xor-shuffle tree over it (fixed
            // order: deterministic), then one partial row per (m-tile, wave row): [2 * tile + wm][sum | sum of squares][N]
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) { st_a[e] += __shfl_xor(st_a[e], off); st_b[e] += __shfl_xor(st_b[e], off); }
            if (rsub == 0 && n < p.N) {
                float* dst = p.stat_part + ((size_t)(cls_i * p.stat_cls_rows + m0 / 64 + wm) * 2) * p.N + n;      // one partial row per 64 GEMM rows (= a wave's rows)
                *(f32x4*)dst = st_a;
                *(f32x4*)(dst + p.N) = st_b;
            }
        }
#ifdef CGS_DIAG_STAMPS
        if (stamp) {
            unsigned long long* dbg = (unsigned long long*)p.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            dbg[0] = sr_in; dbg[1] = sr_loop0; dbg[2] = sr_loop1; dbg[3] = __builtin_amdgcn_s_memrealtime();
            dbg[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                 // HW_ID: wave / simd / cu / sh / se
                     ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);        // XCC_ID
            dbg[5] = (unsigned long long)diag_tiles;
            dbg[6] = (unsigned long long)mb | ((unsigned long long)nb << 32);
        }
#endif
        return;
    }
#pragma unroll     // (must stay fully unrolled: a run-time tn would index the accumulator array and push it to scratch)
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * WTN + tn * 32 + j;
        if (n >= p.N) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        float ea = 1.f, eb = 0.f;
        if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = p.ep_a[n]; eb = p.ep_b[n]; }
        if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = p.ep_a[n];
        const bool use_aux = p.epilogue >= CGS_EPI_RELU_BWD_AFFINE;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (BM / WM) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int pix = rowpix[row];
                if (pix >= 0) {
                    const size_t o = (size_t)pix * p.N + n;
                    const float aux = use_aux ? p.ep_aux[o] : 0.f;
                    p.out[o] = epilogue_apply(acc[tm][tn][r] + bias, p.epilogue, ea, eb, aux);
                }
            }
    }
}

template <int BM, int BN, int NW, bool VEC, int TBK, bool PAR>
__global__ __launch_bounds__(64 * NW, 2) void igemm_kernel(IgemmParams p) { igemm_body<BM, BN, NW, VEC, TBK, PAR, false>(p); }
template <int BM, int BN, int NW, bool VEC, int TBK, bool PAR>
__global__ __launch_bounds__(64 * NW, 2) void igemm_ns_kernel(IgemmParams p) { igemm_body<BM, BN, NW, VEC, TBK, PAR, true>(p); }

// out[pix(m)][n] = epi(bias[n] + sum_s slab[s][m][n]): fixed summation order -> deterministic
__global__ __launch_bounds__(256) void splitk_reduce_kernel(IgemmParams p) {
    const IgemmClass& c = p.cls[blockIdx.y];
    const int RC = c.R * c.C;
    const int M = p.B * RC;
    const int nq = p.Np / 4;
    const long total = (long)M * nq;
    const float* slab = p.slab + c.slab_off;
    // four adjacent lanes share one output quad: lane g adds the slabs g, g+4, ... in order and the four partial sums are
    // combined as (s0 + s1) + (s2 + s3) -- a fixed tree, so the result does not depend on the launch geometry
    const int g = threadIdx.x & 3;
    const long stride = (long)gridDim.x * (blockDim.x >> 2);
    const long total_pad = (total + stride - 1) / stride * stride;          // every lane takes part in the shuffles
    for (long i = (long)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2); i < total_pad; i += stride) {
        const bool live = i < total;
        const int m = live ? (int)(i / nq) : 0, n = live ? (int)(i - (long)m * nq) * 4 : 0;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (live)
            for (int sidx = g; sidx < p.splitk; sidx += 4) a += *(const f32x4*)(slab + ((size_t)sidx * M + m) * p.Np + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] += __shfl_xor(a[e], 1);
            a[e] += __shfl_xor(a[e], 2);
        }
        if (!live || g != 0) continue;
        int b, r, cc, rem;
        if (p.pix_major) { rem = m / p.B; b = m - rem * p.B; } else { b = m / RC; rem = m - b * RC; }
        r = rem / c.C; cc = rem - r * c.C;
        const size_t o = (size_t)((b * p.Hout + r * p.So + c.py) * p.Wout + cc * p.So + c.px) * p.N;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) break;
            const float bias = p.bias ? p.bias[n + e] : 0.f;
            float ea = 1.f, eb = 0.f;
            if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = p.ep_a[n + e]; eb = p.ep_b[n + e]; }
            if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = p.ep_a[n + e];
            const float aux = p.epilogue >= CGS_EPI_RELU_BWD_AFFINE ? p.ep_aux[o + n + e] : 0.f;
            p.out[o + n + e] = epilogue_apply(a[e] + bias, p.epilogue, ea, eb, aux);
        }
    }
}

// The same reduction for a launch that leaves norm statistics (CGS_EPI_NONE, N % 4 == 0; round 5): block = (one partial row = 64 GEMM rows,
// 32 columns) -- 8 column quads x 32 row lanes of 2 rows each; the K slices are added in index order, the stored values' column sums /
// sums of squares go through LDS and are added over the row lanes in index order (deterministic), into the row the one-pass epilogue
// would have written: [class][m / 64][sum | sum of squares][N].  Why: a launch with statistics could not be split over K before, and at the
// reference's batch size (64 images a call, nsgan/main.py:32) D's statistics-leaving convolutions are 16-128 tiles on 256 CUs (dcgan32's
// 4x4 256->512 forward: 16 tiles, 266 us = 0.04 of peak against 38 us for its backward-data, which was split).
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(IgemmParams p) {
    const int cls_i = blockIdx.z;
    const IgemmClass& c = p.cls[cls_i];
    const int RC = c.R * c.C;
    const int M = p.B * RC;
    const float* slab = p.slab + c.slab_off;
    const int cq = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int n = blockIdx.y * 32 + cq * 4;
    __shared__ f32x4 red[2][32][8];
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
        f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias = *(const f32x4*)(p.bias + n);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = blockIdx.x * 64 + i * 32 + rl;
            if (m >= M) continue;
            f32x4 a = *(const f32x4*)(slab + (size_t)m * p.Np + n);
#pragma unroll 4
            for (int z = 1; z < p.splitk; ++z) a += *(const f32x4*)(slab + ((size_t)z * M + m) * p.Np + n);
            int b, rem;
            if (p.pix_major) { rem = m / p.B; b = m - rem * p.B; } else { b = m / RC; rem = m - b * RC; }
            const int r = rem / c.C, cc = rem - r * c.C;
            const size_t o = (size_t)((b * p.Hout + r * p.So + c.py) * p.Wout + cc * p.So + c.px) * p.N + n;
            const f32x4 y = a + bias;
            *(f32x4*)(p.out + o) = y;
            if (p.ns_mean) {             // norm-backward statistics (the block's 64 rows lie in one statistics group; same arithmetic as the one-pass epilogue)
                const int grp = p.ns_gimg <= 0 ? 0 : b / p.ns_gimg;
                const NsLane ns = ns_lane_params(p, grp, n);
                const f32x4 xv = *(const f32x4*)(p.ep_aux + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = fmaf(xv[e], ns.sc[e], ns.sh[e]);
                    const float d = y[e] * (u > 0.f ? 1.f : ns.leak);
                    sa[e] += d; sb[e] = fmaf(d, (xv[e] - ns.mu[e]) * ns.inv[e], sb[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sa[e] += y[e]; sb[e] = fmaf(y[e], y[e], sb[e]); }
            }
        }
    }
    red[0][rl][cq] = sa; red[1][rl][cq] = sb;
    __syncthreads();
    if (rl < 2 && n < p.N) {             // row lane 0 adds the sums, row lane 1 the sums of squares
        f32x4 t = red[rl][0][cq];
#pragma unroll 8
        for (int r = 1; r < 32; ++r) t += red[rl][r][cq];
        *(f32x4*)(p.stat_part + ((size_t)(cls_i * p.stat_cls_rows + blockIdx.x) * 2 + rl) * p.N + n) = t;
    }
}

// Tail split, second half: block = one tail tile.  out[pix(m)][n] = epi(bias[n] + sum_z slab[tile][z][m - m0][n - n0]), the K slices
// added in index order (deterministic); the tile decode is the main kernel's (same wi -> (m-tile, n-tile), same row -> pixel map).
#define TAIL_RSPLIT 16
__global__ __launch_bounds__(256) void tail_reduce_kernel(IgemmParams p, int BM, int BN) {
    const int cls_i = p.nclasses - 1;
    const IgemmClass& c = p.cls[cls_i];
    const int RC = c.R * c.C, M = p.B * RC;
    const int nblk_n = p.Np / BN;
    const unsigned wi = (unsigned)p.tail_from + blockIdx.x;
    int nb, mb;
    if (p.xcd_map) {
        const unsigned xcd = wi & 7u, q = wi >> 3;
        nb = (int)(q % (unsigned)nblk_n);
        mb = (int)(q / (unsigned)nblk_n) * 8 + (int)xcd;
    } else {
        nb = (int)(wi % (unsigned)nblk_n);
        mb = (int)(wi / (unsigned)nblk_n);
    }
    if (mb * BM >= M) return;
    if (p.lpt) {
        const int gpp = p.B / BM;
        const int rank = mb / gpp, grp = mb - rank * gpp;
        mb = (int)p.perm[cls_i][rank] * gpp + grp;
    }
    const int m0 = mb * BM, n0 = nb * BN;
    const float* slab = p.slab + (size_t)blockIdx.x * p.tail_s * (BM * BN);
    const int qpr = BN / 4;                                   // column quads per row
    // blockIdx.y = one of TAIL_RSPLIT row chunks of the tile (a tile per block left the pass on tail_n CUs: 16 blocks adding 1 MB each)
    const int cq = threadIdx.x % qpr, rstep = 256 / qpr;
    const int rbeg = blockIdx.y * (BM / TAIL_RSPLIT), rend = rbeg + BM / TAIL_RSPLIT;
    const int r0 = rbeg + threadIdx.x / qpr;
    const int n = n0 + cq * 4;
    if (n >= p.N) return;                                     // (N % 4 == 0 on this path: the launcher checks)
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, ea = {1.f, 1.f, 1.f, 1.f}, eb = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias = *(const f32x4*)(p.bias + n);
    if (p.epilogue == CGS_EPI_AFFINE_RELU) { ea = *(const f32x4*)(p.ep_a + n); eb = *(const f32x4*)(p.ep_b + n); }
    if (p.epilogue == CGS_EPI_RELU_BWD_AFFINE) ea = *(const f32x4*)(p.ep_a + n);
    for (int rl = r0; rl < rend; rl += rstep) {
        const int m = m0 + rl;
        if (m >= M) break;
        f32x4 a = *(const f32x4*)(slab + rl * BN + cq * 4);
        for (int z = 1; z < p.tail_s; ++z) a += *(const f32x4*)(slab + (size_t)z * (BM * BN) + rl * BN + cq * 4);
        int b, rem;
        if (p.pix_major) { rem = m / p.B; b = m - rem * p.B; } else { b = m / RC; rem = m - b * RC; }
        const int r = rem / c.C, cc = rem - r * c.C;
        const size_t o = (size_t)((b * p.Hout + r * p.So + c.py) * p.Wout + cc * p.So + c.px) * p.N + n;
        f32x4 aux = {0.f, 0.f, 0.f, 0.f};
        if (p.epilogue >= CGS_EPI_RELU_BWD_AFFINE) aux = *(const f32x4*)(p.ep_aux + o);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = epilogue_apply(a[e] + bias[e], p.epilogue, ea[e], eb[e], aux[e]);
        *(f32x4*)(p.out + o) = y;
    }
}

static long igemm_blocks(const IgemmParams& p, int BN, int BM = 128) {
    long blocks = 0;
    for (int i = 0; i < p.nclasses; ++i) blocks += (((long)p.B * p.cls[i].R * p.cls[i].C + BM - 1) / BM) * (p.Np / BN);
    return blocks;
}

// split factor for grids that leave most of the 256 CUs (x2 block slots) idle
static int choose_splitk(const IgemmParams& p) {
    const int BN = (p.Np % 128) == 0 ? 128 : 64;
    const long blocks = igemm_blocks(p, BN);
    int nk_min = 1 << 30;
    for (int i = 0; i < p.nclasses; ++i) {
        const int nk = cgs_ceil_div(p.cls[i].K, BK);
        if (p.cls[i].R * p.cls[i].C > 0 && nk < nk_min) nk_min = nk;
    }
    // (splitting grids of 256-1023 blocks as well was measured on dcgan32, B = 256: -10 % with two batches in flight)
    long target = 512, maxb = 256;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_SPLITK_TARGET")) target = atol(getenv("CGS_SPLITK_TARGET"));
    if (getenv("CGS_SPLITK_MAXBLOCKS")) maxb = atol(getenv("CGS_SPLITK_MAXBLOCKS"));
#endif
    if (blocks == 0 || blocks >= maxb || nk_min < 8) return 1;
    long cap = nk_min / 4;                       // at least four 32-deep K tiles per slice (in the class with the shortest K)
    if (cap > 64) cap = 64;
    if (cap < 2) return 1;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_SPLITK_MODEL") && atoi(getenv("CGS_SPLITK_MODEL")) == 0) {          // (A/B: the rule before round 5: aim at ``target`` blocks)
        long s = (target + blocks - 1) / blocks;
        if (s > cap) s = cap;
        return s < 2 ? 1 : (int)s;
    }
#endif
    // The factor with the least modelled time (round 5; calibrated on the batch-64 launches of dcgan32 / dcgan64 / mnist, tools/sessions/r05_q.sh:
    // these launches are matrix-bound per block, every block on its own SIMDs).  In units of one 128x128x32 tile on an otherwise idle CU:
    //   K loop  = (K tiles per slice) x (blocks on the fullest CU), x 1.41 if that is ONE block (nobody hides its load -> LDS -> MFMA latencies:
    //             a lone block's tile takes 2.4 us against 1.7 us per block for two that share the SIMDs);
    //   slices  = 0.3 per MB of partial sums (written by the epilogue, read back by the reduce pass at ~4 TB/s).
    // The old rule (aim at 512 blocks) left 400 blocks on 256 CUs for dcgan32's 4x4 256->512 layer at batch 64 -- 144 CUs with two blocks of four K tiles,
    // 50 slices of partial sums: 41.6 us -- where 29 slices of seven tiles, one block per CU, take 36.7 us; dcgan64's layers keep their two even rounds.
    int nk_max = 0;
    double slice_mb = 0.0;
    long maxM = 0;
    for (int i = 0; i < p.nclasses; ++i) {
        const long m = (long)p.B * p.cls[i].R * p.cls[i].C;
        if (m <= 0) continue;
        const int nk = cgs_ceil_div(p.cls[i].K, BK);
        if (nk > nk_max) nk_max = nk;
        if (m > maxM) maxM = m;
        slice_mb += (double)m * p.Np * 4e-6;
    }
    const double tile = (BN == 128 ? 1.0 : 0.5) * (maxM <= 64 ? 0.5 : 1.0);
    int best = 1;
    double best_cost = 1e30;
    for (long s = 2; s <= cap; ++s) {
        const long cps = (nk_max + s - 1) / s;
        if ((nk_max + cps - 1) / cps != s) continue;            // (the same slices with fewer empty ones exist as a smaller s)
        const long load = (blocks * s + 255) / 256;
        const double cost = (load == 1 ? 1.41 : (double)load) * (double)cps * tile + 0.3 * slice_mb * (double)s;
        if (cost < best_cost) { best_cost = cost; best = (int)s; }
    }
    (void)target;
    return best;
}

// Block tile of a launch (the row policy and the split-K decision are made): 128 x 128 or 128 x 64 (``wide``), 32- or 16-deep K tiles
// (``deep``); ``mid`` = a mid-size grid that took the narrow tile to fill the block slots.  One function: the launcher and the
// workspace sizing (tail split) must agree on it.
struct IgemmTiles { bool wide, mid, deep, tall, wide_for_tail; };
struct IgemmTail { int from, n, s; };
static IgemmTail igemm_choose_tail(const IgemmParams& p, const IgemmTiles& t);

// The launch-shape rules below that reason about "rounds" of workgroups (the tail split, the class flip, the one-round balancing) are
// written for the 256 compute units of an MI355X: the kernel's flip decode shifts by 8 and the planners deal blocks modulo 256.  The
// count is asked of the device once; on any other part those rules are switched off (results do not depend on them, only speed).
#define IGEMM_CUS 256
static bool igemm_round_rules_apply() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    return cus == IGEMM_CUS || cus == 0;         // (0: no device visible -- host-only planning queries, e.g. workspace sizing in the CPU suite)
}

// ``tail_wide``: may the rule "wide tiles + tail split" be taken (the launcher asks again without it when the caller's slab is too small)
static IgemmTiles igemm_choose_tiles(const IgemmParams& p, bool tail_wide = true) {
    const bool vec = p.vec != 0;
    int maxRC = 0;
    for (int i = 0; i < p.nclasses; ++i) maxRC = p.cls[i].R * p.cls[i].C > maxRC ? p.cls[i].R * p.cls[i].C : maxRC;
    bool wide = (p.Np % 128) == 0, wide_for_tail = false;
    if (wide && p.lpt && maxRC <= 64) {   // uneven tiles (9..25 valid taps on grids <= 8x8) need >= 2 rounds of blocks over the 512 block slots to balance: halve BN if short
        long blocks = 0;
        for (int i = 0; i < p.nclasses; ++i) blocks += ((long)p.B * p.cls[i].R * p.cls[i].C / 128) * (p.Np / 128);
        if (blocks < 1024) wide = false;
        // ... unless the wide tiles make ONE partial round that the tail split evens out (igemm_choose_tail: 512 < T <= 1024 tiles at four
        // per CU, a remainder of at most 176 over whole CUs, no statistics / sign mask to leave): mnist's 7x7 128<-64 backward-data, 784
        // tiles, 200.2 us as 1568 narrow tiles -> 183.4 us (0.72 -> 0.79 of peak; profiles/r05_e_wide_tail_ab.txt)
        // -- and only if the tail planner, asked about exactly that tiling, does split it (it can still say no: an m-tile count that the per-XCD
        // decode pads, fewer than eight K tiles): otherwise the launch would run the partial round of wide tiles the rule above avoids
        if (!wide && tail_wide && vec && p.splitk == 1 && !p.stat_part && !p.sign_out && (p.N & 3) == 0 && p.nclasses == 1 && blocks >= 512 && blocks <= 1024 &&
            (blocks % 256) != 0 && (blocks % 256) <= 176 && igemm_choose_tail(p, IgemmTiles{true, false, false, false, true}).s > 1) {
            wide = true;
            wide_for_tail = true;
        }
#ifdef CGS_EXPERIMENT
        if (getenv("CGS_FORCE_WIDE")) wide = atoi(getenv("CGS_FORCE_WIDE")) != 0;      // (A/B: 0 = the narrow tiles, 1 = the wide ones, whatever the rules above said)
#endif
    }
    // K tile: 32 when every tile lies inside one tap (VEC); 16 for the generic-K gather (small K: less padding waste)
    // (8-wave 128x128 blocks, 4 waves per SIMD: +1.3 % with one batch in flight, +-0 with two -- not kept)
    // 16-deep K tiles (LDS 37 KB, 110 VGPRs -> FOUR blocks per CU instead of two: more independent waves to fill the matrix
    // pipe's gaps; the epilogue stages half a wave tile at a time to fit) win 2.5-9 % when every parity class brings at
    // least two blocks per CU, and lose up to 11 % on smaller grids, where a CU holds one block and the doubled barrier
    // count per FLOP is all that is left of the change (measured per layer, 40 launches each, dcgan64 / dcgan32 / config 5).
    // Mid-size grids (round 3, measured per layer on config 5's PatchGAN / up-sampling layers at batch 8, a same-session A/B of tools/layer_bench.py): a launch
    // of 256..511 blocks of 128x128 leaves half of the 512 block slots of the 32-deep form empty (one block per CU: nobody hides
    // its barrier and load latencies) -- as 128x64 blocks it fills them: 64x64 128<-256 85 -> 75 us, 128x128 64->128 88 -> 77 us,
    // 32x32 256->512 302 -> 265 us (not for grids < 256 blocks, which are split over K instead: 32x32 256<-512 272 -> 295 us)
    // (with the 32-deep K tiles: the same grids as 16-deep 128x64 blocks measured 93 / 329 us for the last two)
    bool mid = false;
    if (wide && vec && p.splitk == 1) {
        const long wb = igemm_blocks(p, 128);
        if (wb >= 256 && wb < 512) { wide = false; mid = true; }
        // a launch that leaves a sign mask (one-pass epilogue only), or statistics when the workspace has no room for the split-K slabs
        // (with room it is split: splitk_reduce_stats_kernel): under 256 blocks of 128x128 it would leave most CUs idle (config 5's
        // PatchGAN 4x4 128->256 layer: 128 blocks, 54 TFLOP/s) -- as 128x64 blocks at least every CU gets one
#ifndef CGS_NO_STAT_NARROW
        else if (wb < 256 && (p.stat_part || p.sign_out)) { wide = false; mid = true; }
#endif
    }
    bool deep = true;
    if (vec && p.splitk == 1 && !mid) {
        long min_blocks = 1L << 40, tot_blocks = 0;
        for (int i = 0; i < p.nclasses; ++i) {
            const long m = (long)p.B * p.cls[i].R * p.cls[i].C;
            if (m > 0) { const long bl = (m + 127) / 128 * (p.Np / (wide ? 128 : 64)); if (bl < min_blocks) min_blocks = bl; tot_blocks += bl; }
        }
        deep = min_blocks < 512;        // (thresholds 256 / 512 / 1024 measured: 512 is best on dcgan64 and neutral elsewhere)
        // ... except image-major launches whose classes TOGETHER fill the 1024 slots of the 16-deep form with even tiles (the four
        // parity classes of config 5's transposed layers, 256 blocks each: 128x128 64<-128 102 -> 86 us, 64x64 256->128 164 -> 152 us);
        // the pixel-major 8x8 256<-512 layer of dcgan64 (1024 very uneven blocks) stays 32-deep: 0.61 vs 0.74 ms
        if (deep && !p.pix_major && p.nclasses > 1 && tot_blocks >= 1024) deep = false;
    }
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_FORCE_DEEP")) deep = atoi(getenv("CGS_FORCE_DEEP")) != 0;
    if (getenv("CGS_FORCE_NARROW")) wide = false;
#endif
    // Tall tiles, 256 x 64 (four 64 x 64 wave tiles stacked: the per-wave shape of the 128 x 128 block), for launches to 64 output channels
    // whose K loops are SHORT (<= 512 per class: 32 tiles of 16) and whose grid is many rounds of blocks: half the blocks, so half the
    // prologues / epilogues per output, and every A fragment meets two B fragments.  Measured per stage (round 5, A/B interleaved in one
    // process, profiles/r05_j_tall_tiles_ab.txt): mnist's 7x7 128->64 transposed forward 233.7 -> 214.7 us, its 14x14 64<-128
    // backward-data 218.7 -> 206.3 us; with LONGER K loops the three resident blocks (48 KB of LDS each) hide latencies worse than five
    // 128 x 64 ones: dcgan64's 16x16 128->64 forward 754.6 -> 783.3 us, its 32x32 64<-128 backward-data 817.2 -> 862.6 us (K = 1152): not there.
    bool tall = false;
    {
        int kmax = 0;
        for (int i = 0; i < p.nclasses; ++i) kmax = p.cls[i].K > kmax ? p.cls[i].K : kmax;
        // (pixel-major launches only: config 5's image-major 128x128 128->64 transposed layer, K = 512, measured 166.8 -> 168.8 us)
        tall = vec && !wide && !mid && !deep && p.splitk == 1 && p.Np == 64 && p.pix_major && (p.B % 256) == 0 && kmax <= 512 &&
               igemm_blocks(p, 64, 256) >= 1536;
#ifdef CGS_EXPERIMENT
        if (getenv("CGS_TALL")) {        // 0: never; 1: wherever the tile can run (from CGS_TALL_MIN blocks on: parity runs force it onto small launches)
            tall = atoi(getenv("CGS_TALL")) != 0 && vec && !wide && !mid && !deep && p.splitk == 1 && p.Np == 64 && (!p.pix_major || (p.B % 256) == 0) &&
                   igemm_blocks(p, 64, 256) >= (getenv("CGS_TALL_MIN") ? atol(getenv("CGS_TALL_MIN")) : 1536);
        }
#endif
    }
    return IgemmTiles{wide, mid, deep, tall, wide_for_tail && wide};
}

// blocks of this tile shape a CU holds at once (LDS: 2 * 128 * (TBK + 4) + 2 * TBK * BN floats + the row map; registers allow as many)
static int igemm_blocks_per_cu(bool wide, bool deep) { return deep ? (wide ? 2 : 3) : (wide ? 4 : 5); }

// Tail split.  The dispatcher deals the workgroups of a launch to the CUs round-robin (tools/probe/place_probe.hip: workgroups b, b + 256,
// b + 512, ... share a CU), so a launch of T = a * 256 + r tiles that fits one round of block slots leaves r CUs with a + 1 tiles and the
// others with a: it lasts a + 1 tile times although the chip holds a + r / 256 (mnist's 2048 x 6272 x 1024 linear backward: 784 tiles of
// 128 x 128 = 3.06 per CU ran at 0.66 of peak where its tile runs at 0.85).  The last r tiles are therefore cut into S ~ 256 / r K-slices
// each -- one short block per CU behind its a whole tiles; the slices leave raw partial tiles, tail_reduce_kernel adds them in index order
// (deterministic) and runs the epilogue.  Measured per stage, A/B interleaved in one process (profiles/r05_d_tail_split_ab.txt): that launch
// 253.4 -> 209.1 us (0.66 -> 0.80 of peak).
// Launches of SEVERAL rounds (T = q * L + r on L = 256 * blocks-per-CU slots, the short last round cut the same way) were measured too and
// are NOT split: +2 % / +-0 / +4.5 % on three of them, -3.4 % and -1.3 % on two (dcgan32 16x16 64<-128, dcgan64 32x32 64<-128: their
// last round already overlaps the tail of the one before, and the slices pay a prologue, a raw 32 KB store and the reduce pass each);
// experiment builds keep that plan behind CGS_TAIL_MULTI=1.
static IgemmTail igemm_choose_tail(const IgemmParams& p, const IgemmTiles& t) {
    IgemmTail none{0, 0, 0};
    if (!igemm_round_rules_apply()) return none;
    if (!p.vec || p.splitk > 1 || p.stat_part || p.sign_out || (p.N & 3) || t.tall) return none;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_TAIL") && atoi(getenv("CGS_TAIL")) == 0) return none;
#endif
    const int bn = t.wide ? 128 : 64, tbk = t.deep ? 32 : 16;
    const IgemmClass& cl = p.cls[p.nclasses - 1];
    const long Ml = (long)p.B * cl.R * cl.C;
    if (Ml <= 0) return none;
    const long mtl = (Ml + 127) / 128, Tl = mtl * (p.Np / bn);          // tiles of the last class
    const long T = igemm_blocks(p, bn);
    const long L = 256L * igemm_blocks_per_cu(t.wide, t.deep);
    const int nk = cgs_ceil_div(cl.K, tbk);
    long r; int S;
    if (T < 256) return none;                                            // (under-filled grids are split over K as a whole)
    if (T <= L) {
        r = T % 256;
        if (r == 0 || r > 176) return none;                              // (a remainder above ~2/3 of the CUs: the last tile time is mostly used)
        S = (int)((256 + r / 2) / r);
        if (S > 16) S = 16;
    } else {
#ifdef CGS_EXPERIMENT
        if (!getenv("CGS_TAIL_MULTI") || atoi(getenv("CGS_TAIL_MULTI")) == 0) return none;
        r = T % L;
        if (r == 0 || r * 5 > L * 4) return none;
        S = (int)((L + r / 2) / r);
        if (S > 8) S = 8;
#else
        return none;
#endif
    }
    if (S > nk / 4) S = nk / 4;                                          // at least 4 K tiles per slice
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_TAIL_S")) S = atoi(getenv("CGS_TAIL_S"));
#endif
    if (S < 2) return none;
    if (r > Tl) r = Tl;
    for (int i = 0; i < p.nclasses; ++i)                                 // (one id space for all classes: equal sizes only)
        if ((long)p.B * p.cls[i].R * p.cls[i].C != Ml) return none;
    // the tail tiles are the LAST ids of the last class in the launch's own decode (per-XCD decode: ids run over the padded m-tile count)
    const bool xcd_map = (mtl % 8 == 0 || mtl >= 64);
    const long ids = (xcd_map ? (mtl + 7) / 8 * 8 : mtl) * (p.Np / bn);
    if (xcd_map && (mtl % 8) != 0) return none;                          // (padded ids decode to no tile: keep the tail dense)
    return IgemmTail{(int)(ids - r), (int)r, S};
}

static size_t igemm_tail_bytes(const IgemmTail& tl, bool wide) { return (size_t)tl.n * tl.s * 128 * (wide ? 128 : 64) * sizeof(float); }

// slab bytes behind the packed weights a launch of this geometry can use: split-K slabs (under-filled grids) or tail-split partial tiles
size_t cgs_igemm_slab_bytes(const IgemmParams& p_in) {
    const size_t sk = cgs_igemm_splitk_bytes(p_in);
    if (sk) return sk;
    IgemmParams p = p_in;
    p.splitk = 1; p.stat_part = nullptr; p.sign_out = nullptr;
    cgs_igemm_row_policy(p, 128);
    const IgemmTiles t = igemm_choose_tiles(p);
    const IgemmTail tl = igemm_choose_tail(p, t);
    return tl.s > 1 ? igemm_tail_bytes(tl, t.wide) : 0;
}

int cgs_igemm_signs_ok(const IgemmParams& p) {
    return (p.N % 32) == 0 && (p.epilogue == CGS_EPI_AFFINE_RELU || p.epilogue == CGS_EPI_LRELU) && choose_splitk(p) <= 1;
}

size_t cgs_igemm_splitk_bytes(const IgemmParams& p) {
    const int s = choose_splitk(p);
    if (s <= 1) return 0;
    size_t n = 0;
    for (int i = 0; i < p.nclasses; ++i) n += (size_t)s * p.B * p.cls[i].R * p.cls[i].C * p.Np;
    return n * sizeof(float);
}

template <int BM, int BN, int NW, bool VEC, int TBK, bool PAR, bool NS>
static int launch_cfg(const IgemmParams& p, hipStream_t s) {
    void (*const kern)(IgemmParams) = NS ? igemm_ns_kernel<BM, BN, NW, VEC, TBK, PAR> : igemm_kernel<BM, BN, NW, VEC, TBK, PAR>;
    constexpr int EH = (VEC && TBK == 16) ? 2 : 1;
    constexpr int WM = BM / 64;
    constexpr size_t kloop_f = (size_t)2 * BM * (TBK + 4) + 2 * TBK * BN, stage_f = (size_t)NW * (BM / WM / EH) * (BN / (NW / WM) + 4);
    constexpr size_t smem = (kloop_f > stage_f ? kloop_f : stage_f) * sizeof(float) + BM * sizeof(int);
    static bool attr_done[64] = {};       // per device: the attribute belongs to the device's code object
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    dev_ &= 63;
    if (!attr_done[dev_]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return cgs_set_error(CGS_ELAUNCH, "igemm smem attr: %s", hipGetErrorString(e));
        attr_done[dev_] = true;
    }
    long maxM = 0;
    for (int i = 0; i < p.nclasses; ++i) {
        long m = (long)p.B * p.cls[i].R * p.cls[i].C;
        if (m > maxM) maxM = m;
    }
    if (maxM == 0) return CGS_OK;
    const long mtiles = (maxM + BM - 1) / BM;
    IgemmParams q = p;
    q.xcd_map = (mtiles % 8 == 0 || mtiles >= 64) ? 1 : 0;        // per-XCD decode only where it keeps the 8 XCDs evenly loaded
    long gx = (q.xcd_map ? (mtiles + 7) / 8 * 8 : mtiles) * (p.Np / BN);
    if (p.tail_s > 1) {          // the tail tiles' ids are replaced by tail_s ids each (the planner and this decode agree: igemm_choose_tail)
        if ((long)p.tail_from + p.tail_n != gx) return cgs_set_error(CGS_EINVAL, "igemm: tail split planned for another grid (%d + %d != %ld)", p.tail_from, p.tail_n, gx);
        gx = (long)p.tail_from + (long)p.tail_n * p.tail_s;
    }
    unsigned gy = (unsigned)p.nclasses;
    // (image-major launches of two rounds only: measured per stage at 64 / 128 / 256 images, profiles/r05_ai_class_flip_ab.txt -- dcgan64 at batch 64: 102.5 -> 89.0,
    // 95.2 -> 82.4, 89.9 -> 77.5 us; the pixel-major launches of 128- and 256-image batches, whose tiles differ by pixel as well, lose up to 14 % with it)
    {
        // (the flip decides per CLASS range of gx blocks which round it lies in: those ranges must not straddle a round -- gx divides 256 --
        // and a group of classes that is reversed must lie inside one round)
        const long total = gx * (long)p.nclasses * (p.splitk > 1 ? p.splitk : 1);
        q.cls_flip = 0;
        if (igemm_round_rules_apply() && p.nclasses > 1 && !p.pix_major && total > 256 && total <= 512 && gx > 0 && (256 % gx) == 0 && p.tail_s <= 1) {
            const long per_round = 256 / gx;                          // class ranges per round of 256 blocks
            if (p.splitk > 1 && (per_round % p.nclasses) == 0) q.cls_flip = p.nclasses;                 // whole K slices per round: reverse all classes
            else if (p.splitk == 1 && per_round < p.nclasses && (p.nclasses % per_round) == 0) q.cls_flip = (int)per_round;      // several rounds per slice
        }
    }
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_CLS_FLIP") && atoi(getenv("CGS_CLS_FLIP")) == 0) q.cls_flip = 0;
    if (getenv("CGS_CLS_FLIP") && atoi(getenv("CGS_CLS_FLIP")) == 2 && p.splitk == 1) q.cls_flip = 0;      // (2 = split launches only, as first adopted)
    if (getenv("CGS_CLS_INTER") && atoi(getenv("CGS_CLS_INTER")) && q.xcd_map == 1 && (mtiles % 8) == 0 && p.nclasses > 1 && p.tail_s <= 1 && p.splitk == 1 && !p.pix_major &&
        gx >= atol(getenv("CGS_CLS_INTER"))) {
        bool eq = true;
        for (int i = 1; i < p.nclasses; ++i) eq = eq && p.cls[i].R * p.cls[i].C == p.cls[0].R * p.cls[0].C;
        if (eq) { q.xcd_map = 2; gx *= p.nclasses; gy = 1; }
    }
#endif
    if (gx > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "igemm: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy, p.splitk > 1 ? p.splitk : 1), dim3(64 * NW), smem, s, q);
    CGS_CHECK_LAUNCH("igemm");
    if (p.tail_s > 1) {
        hipLaunchKernelGGL(tail_reduce_kernel, dim3((unsigned)p.tail_n, TAIL_RSPLIT), dim3(256), 0, s, q, BM, BN);
        CGS_CHECK_LAUNCH("tail_reduce");
    }
    if (p.splitk > 1) {
        long tot = 0;
        for (int i = 0; i < p.nclasses; ++i) { long t = (long)p.B * p.cls[i].R * p.cls[i].C * (p.Np / 4); if (t > tot) tot = t; }
        unsigned rb = (unsigned)((tot + 63) / 64 > 4096 ? 4096 : (tot + 63) / 64);       // 64 output quads per 256-thread block
        if (p.stat_part) hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)p.stat_cls_rows, (unsigned)((p.N + 31) / 32), p.nclasses), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rb, p.nclasses), dim3(256), 0, s, p);
        CGS_CHECK_LAUNCH("splitk_reduce");
    }
    // the name rocprofv3 prints for this instantiation
    static char name[64];
    snprintf(name, sizeof(name), "%s<%d, %d, %d, %s, %d, %s>", NS ? "igemm_ns_kernel" : "igemm_kernel", BM, BN, NW, VEC ? "true" : "false", TBK, PAR ? "true" : "false");
    cgs_note_kernel(name);
    return CGS_OK;
}

// pixel-major row order (a tile = ONE base pixel of 128 images, so a tap that falls into the zero padding does so for the
// whole tile and is skipped) pays when the batch fills whole tiles and the pixel grid is small enough that the padding is a
// visible share of the taps: 28 % of the MACs at 4x4, 14 % at 8x8, 7 % at 16x16 (measured at batch 1024: 16x16 grids
// -3..-5.6 % per layer, also for the 268 MB input of the 32x32x64 forward layer, which no longer fits the Infinity Cache:
// the ~128 blocks an XCD runs at once are 128 pixels of the SAME image group, so the re-reads meet in its L2)
static bool igemm_pix_major(const IgemmParams& p) {
    int maxRC = 0;
    for (int i = 0; i < p.nclasses; ++i) maxRC = p.cls[i].R * p.cls[i].C > maxRC ? p.cls[i].R * p.cls[i].C : maxRC;
    const size_t in_bytes = (size_t)p.B * p.Hin * p.Win * p.Cred * 4;
    int pix_max = 256;
    size_t pix_bytes = (size_t)384 << 20;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_PIXMAX")) pix_max = atoi(getenv("CGS_PIXMAX"));
    if (getenv("CGS_PIXBYTES")) pix_bytes = (size_t)atoi(getenv("CGS_PIXBYTES")) << 20;
#endif
    return p.vec && p.B >= 128 && maxRC <= pix_max && maxRC > 1 && in_bytes <= pix_bytes;
}

// GEMM row order of a launch and, for whole-tile pixel-major launches, the heaviest-first pixel order (BM = rows of a block tile)
void cgs_igemm_row_policy(IgemmParams& p, int BM) {
    p.pix_major = igemm_pix_major(p);
    p.lpt = p.pix_major && (p.B % BM) == 0;
    if (p.lpt)
        for (int ci = 0; ci < p.nclasses; ++ci) {
            const IgemmClass& c = p.cls[ci];
            int cnt[256];
            for (int pix = 0; pix < c.R * c.C; ++pix) {
                const int r = pix / c.C, cc = pix - r * c.C;
                int ny = 0, nx = 0;
                for (int ta = 0; ta < c.nty; ++ta) { const int iy = r * p.S + c.dy0 + ta * p.dstep; ny += (iy >= 0 && iy < p.Hin); }
                for (int tb = 0; tb < c.ntx; ++tb) { const int ix = cc * p.S + c.dx0 + tb * p.dstep; nx += (ix >= 0 && ix < p.Win); }
                cnt[pix] = ny * nx;
                p.perm[ci][pix] = (unsigned char)pix;
            }
            for (int i = 1; i < c.R * c.C; ++i)       // insertion sort, descending tap count, stable
                for (int j = i; j > 0 && cnt[p.perm[ci][j]] > cnt[p.perm[ci][j - 1]]; --j) {
                    unsigned char t = p.perm[ci][j]; p.perm[ci][j] = p.perm[ci][j - 1]; p.perm[ci][j - 1] = t;
                }
        }
}

// multiply-accumulates a launch really issues -> cgs_last_executed_flops: the algorithmic count minus the zero-padding taps whose
// K tiles the kernel skips (pixel-major tiles whose BM rows are one base pixel; exact, see igemm_kernel) -- for honest rooflines
void cgs_igemm_count_flops(const IgemmParams& p, int BM) {
    double macs = 0.0;
    for (int ci = 0; ci < p.nclasses; ++ci) {
        const IgemmClass& c = p.cls[ci];
        const long M = (long)p.B * c.R * c.C;
        const int ntaps = c.nty * c.ntx;
        if (!(p.vec && p.pix_major)) { macs += (double)M * p.N * p.Cred * ntaps; continue; }
        for (long m0 = 0; m0 < M; m0 += BM) {
            const long ml = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
            const int pf = (int)(m0 / p.B), pl = (int)(ml / p.B);
            int taps = ntaps;
            if (pf == pl) {
                const int r = pf / c.C, cc = pf - r * c.C;
                int ny = 0, nx = 0;
                for (int ta = 0; ta < c.nty; ++ta) { const int iy = r * p.S + c.dy0 + ta * p.dstep; ny += (iy >= 0 && iy < p.Hin); }
                for (int tb = 0; tb < c.ntx; ++tb) { const int ix = cc * p.S + c.dx0 + tb * p.dstep; nx += (ix >= 0 && ix < p.Win); }
                taps = ny * nx;
            }
            macs += (double)(ml - m0 + 1) * p.N * p.Cred * taps;
        }
    }
    cgs_add_flops(2.0 * macs);
}

int cgs_igemm_row_order(const IgemmParams& p) {
    if (!igemm_pix_major(p)) return 0;
    return (p.B % 128) == 0 ? 2 : 1;
}

int cgs_igemm_launch(const IgemmParams& p_in, void* slab, size_t slab_bytes, hipStream_t s) {
    IgemmParams p = p_in;
    p.splitk = 1; p.slab = nullptr;
    p.tail_from = p.tail_n = p.tail_s = 0;
    p.prio_t[0] = p.prio_t[1] = p.prio_t[2] = 0;
#ifdef CGS_DIAG_STAMPS
    // (the stamps go to the LAST MiB of the caller's workspace: 16384 blocks x 64 bytes; tools/clock_probe.py sizes it so)
    if (getenv("CGS_STAMP") && slab && slab_bytes >= (1u << 20)) p.slab = (float*)((char*)slab + slab_bytes - (1u << 20) - ((uintptr_t)((char*)slab + slab_bytes) & 15));
#endif
    int maxRC = 0;
    for (int i = 0; i < p.nclasses; ++i) maxRC = p.cls[i].R * p.cls[i].C > maxRC ? p.cls[i].R * p.cls[i].C : maxRC;
    cgs_igemm_row_policy(p, 128);
    if ((long)p.B * p.Hout * p.Wout * p.N * 4 > 0x7fffffffL || (long)p.B * p.Hin * p.Win * p.Cred * 4 > 0x7fffffffL)
        return cgs_set_error(CGS_EINVAL, "igemm: a tensor of one launch exceeds 2 GiB (the caller splits the batch)");
    if (p.stat_part && ((p.N & 3) || p.epilogue != CGS_EPI_NONE))
        return cgs_set_error(CGS_EINVAL, "igemm: fused statistics need N %% 4 == 0 and no epilogue");
    if (p.stat_part && p.ns_mean && (!p.ns_inv || !p.ns_gamma || !p.ns_beta || !p.ep_aux || p.ns_gimg < 0))
        return cgs_set_error(CGS_EINVAL, "igemm: norm-backward statistics need x (ep_aux), mean, invstd, gamma and beta");
    p.stat_cls_rows = 0;
    if (p.stat_part) {        // one partial row per (128-row tile, wave row), the parity classes back to back (equal M: cgs_conv_stat_layout)
        const long M0 = (long)p.B * p.cls[0].R * p.cls[0].C;
        for (int i = 1; i < p.nclasses; ++i)
            if ((long)p.B * p.cls[i].R * p.cls[i].C != M0) return cgs_set_error(CGS_EINVAL, "igemm: fused statistics need parity classes of equal size");
        p.stat_cls_rows = (int)(2 * ((M0 + 127) / 128));
    }
    if (p.sign_out && !cgs_igemm_signs_ok(p))
        return cgs_set_error(CGS_EINVAL, "igemm: a sign mask needs N %% 32 == 0, the relu / lrelu forward epilogues and a grid that is not split over K");
    {   // split-K for under-filled grids, if the caller's workspace has room for the partial slabs
        const size_t need = p.sign_out ? 0 : cgs_igemm_splitk_bytes(p);      // (a sign mask comes out of the one-pass epilogue; statistics: splitk_reduce_stats_kernel)
#ifdef CGS_EXPERIMENT
        const bool stat_split = !(p.stat_part && getenv("CGS_STAT_SPLIT") && atoi(getenv("CGS_STAT_SPLIT")) == 0);      // (A/B: 0 = statistics launches unsplit, as before round 5)
#else
        const bool stat_split = true;
#endif
        if (need && stat_split && slab && slab_bytes >= need) {
            p.splitk = choose_splitk(p); p.slab = (float*)slab;
            size_t off = 0;
            for (int i = 0; i < p.nclasses; ++i) {
                p.cls[i].slab_off = (int)off;
                off += (size_t)p.splitk * p.B * p.cls[i].R * p.cls[i].C * p.Np;
            }
        }
    }
    const bool vec = p.vec != 0;
    IgemmTiles tiles = igemm_choose_tiles(p);
    IgemmTail tl = igemm_choose_tail(p, tiles);
    bool tail_fits = tl.s > 1 && slab && slab_bytes >= igemm_tail_bytes(tl, tiles.wide);
    if (tiles.wide_for_tail && !tail_fits) {         // wide tiles were chosen FOR the tail split and the caller's slab has no room for it: the narrow plan
        tiles = igemm_choose_tiles(p, false);
        tl = igemm_choose_tail(p, tiles);
        tail_fits = tl.s > 1 && slab && slab_bytes >= igemm_tail_bytes(tl, tiles.wide);
    }
    bool wide = tiles.wide, deep = tiles.deep;
    const bool tall = tiles.tall;
    if (tall) cgs_igemm_row_policy(p, 256);          // (whole 256-image tiles of one pixel: lpt needs B % 256 == 0)
    cgs_igemm_count_flops(p, tall ? 256 : 128);
    if (tail_fits) {   // tail split (see igemm_choose_tail), if the caller's workspace has room for the partial tiles
        p.tail_from = tl.from; p.tail_n = tl.n; p.tail_s = tl.s; p.slab = (float*)slab;
        cgs_note_tail(tl.n, tl.s);
    }
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_PLAN_PRINT")) {
        const long T_ = igemm_blocks(p, wide ? 128 : 64), L_ = 256L * igemm_blocks_per_cu(wide, deep);
        fprintf(stderr, "[igemm plan] B=%d in %dx%dx%d -> out %dx%dx%d classes %d K0=%d: tile 128x%dx%d, T=%ld L=%ld (T%%256=%ld, T%%L=%ld) splitk=%d stats=%d signs=%d pix_major=%d lpt=%d tail=%dx%d\n",
                p.B, p.Hin, p.Win, p.Cred, p.Hout, p.Wout, p.N, p.nclasses, p.cls[0].K, wide ? 128 : 64, deep ? 32 : 16, T_, L_, T_ % 256, T_ % L_, p.splitk,
                p.stat_part != nullptr, p.sign_out != nullptr, p.pix_major, p.lpt, p.tail_n, p.tail_s);
    }
#endif
    {
        // Launches whose blocks are all resident at once (one "round": <= 4 blocks per CU with the 16-deep tiles, 2 with the
        // 32-deep ones) end when their slowest CU ends.  Two things even that out (measured per layer with in-kernel stamps,
        // tools/clock_probe.py: CU finish times 633-761 us before, span 761 -> 700 us for the 16x16 128->256 layer):
        //  (a) fair progress of the blocks sharing a CU (priority steps, see the kernel);
        //  (b) the pixel-major tiles have 9..25 valid taps each and the dispatcher hands workgroups to the 32 CUs of an XCD
        //      round-robin, so the heaviest-first order alone gives the first CUs several 25-tap tiles; instead the tiles are
        //      dealt to the CUs by greedy bin packing (heaviest tile to the least loaded CU with a free slot).
        // (block slots per CU: the one model the tail planner uses too, igemm_blocks_per_cu)
        const int bn_sel = wide ? 128 : 64, per_cu = vec ? igemm_blocks_per_cu(wide, deep) : 2;
        const long total_blocks = igemm_blocks(p, bn_sel);
        const bool one_round = igemm_round_rules_apply() && vec && p.splitk == 1 && !tall && total_blocks <= (long)IGEMM_CUS * per_cu && total_blocks >= IGEMM_CUS;
        // (a) measured per layer at batch 1024: +2..4 % on every pixel-major layer (one or two rounds of 128x128 / 128x64 blocks),
        // -3 % on the transposed 128x64 layers with their 8192 short blocks (a fresh block at priority 3 starves the ones about
        // to finish), neutral elsewhere -> pixel-major launches only
        if (vec && !deep && p.splitk == 1 && p.pix_major && total_blocks >= 256) { p.prio_t[0] = 85; p.prio_t[1] = 170; p.prio_t[2] = 0; }
        const int nblk_n = p.Np / bn_sel;
        if (one_round && p.lpt && p.nclasses == 1 && p.B / 128 == 8 && (32 % nblk_n) == 0) {
            const IgemmClass& c = p.cls[0];
            const int RC = c.R * c.C, nbins = 32 / nblk_n, cap = (RC + nbins - 1) / nbins;
            int cnt[256], load[32] = {}, used[32] = {};
            unsigned char bin_items[32][256];
            for (int pix = 0; pix < RC; ++pix) {
                const int r = pix / c.C, cc = pix - r * c.C;
                int ny = 0, nx = 0;
                for (int ta = 0; ta < c.nty; ++ta) { const int iy = r * p.S + c.dy0 + ta * p.dstep; ny += (iy >= 0 && iy < p.Hin); }
                for (int tb = 0; tb < c.ntx; ++tb) { const int ix = cc * p.S + c.dx0 + tb * p.dstep; nx += (ix >= 0 && ix < p.Win); }
                cnt[pix] = ny * nx;
            }
            for (int i = 0; i < RC; ++i) {                       // p.perm[0] is sorted by descending tap count
                const int pix = p.perm[0][i];
                int best = -1;
                for (int b = 0; b < nbins; ++b) {
                    const int slots = (RC - b + nbins - 1) / nbins;       // positions b, b + nbins, ... < RC
                    if (used[b] < slots && used[b] < cap && (best < 0 || load[b] < load[best])) best = b;
                }
                bin_items[best][used[best]++] = (unsigned char)pix;
                load[best] += cnt[pix];
            }
            for (int b = 0; b < nbins; ++b)
                for (int k = 0; k < used[b]; ++k) p.perm[0][b + k * nbins] = bin_items[b][k];
        }
    }
    p.uni = !(p.nclasses > 1 && wide && maxRC <= 64);
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_UNI")) p.uni = atoi(getenv("CGS_UNI")) == 2 ? p.uni : atoi(getenv("CGS_UNI"));
#endif
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_PRIO")) sscanf(getenv("CGS_PRIO"), "%d,%d,%d", &p.prio_t[0], &p.prio_t[1], &p.prio_t[2]);
    if (getenv("CGS_NOBALANCE")) { /* diagnostic: handled by CGS_PRIO=0,0,0 for (a); (b) has no switch */ }
#endif
    // Launches of at most 64 GEMM rows that are split over K (the fully connected layers at the reference's batch size, nsgan/main.py:32: 64 x 6272 x 1024):
    // 64-row tiles -- a 128-row tile is half padding there and the blocks are matrix-bound (4-8 K tiles each on their own SIMDs).  Same-process
    // A/B (round 5, profiles/r05_t_bm64_ab.txt): mnist's 6272->1024 forward 34.6 -> 25.9 us, backward 32.5 -> 24.0 us, the batch-64 call 12.14 -> 11.38 ms.
    bool half = vec && wide && deep && p.splitk > 1 && !p.tap_parity && (long)p.B * maxRC <= 64;
#ifdef CGS_EXPERIMENT
    if (getenv("CGS_BM64")) half = half && atoi(getenv("CGS_BM64")) != 0;
#endif
    // (a launch that leaves the norm-backward sums runs the igemm_ns_kernel twin of its tile; every other launch the kernel of the rounds before)
    const bool ns = p.stat_part && p.ns_mean;
#define LC(...) (ns ? launch_cfg<__VA_ARGS__, true>(p, s) : launch_cfg<__VA_ARGS__, false>(p, s))
    if (half) return LC(64, 128, 4, true, 32, false);
    if (tall) return p.tap_parity ? LC(256, 64, 4, true, 16, true) : LC(256, 64, 4, true, 16, false);
    if (vec && !deep && !p.tap_parity) return wide ? LC(128, 128, 4, true, 16, false) : LC(128, 64, 4, true, 16, false);
    if (vec && !deep && p.tap_parity) return wide ? LC(128, 128, 4, true, 16, true) : LC(128, 64, 4, true, 16, true);
    if (vec && p.tap_parity) return wide ? LC(128, 128, 4, true, 32, true) : LC(128, 64, 4, true, 32, true);
    if (vec) return wide ? LC(128, 128, 4, true, 32, false) : LC(128, 64, 4, true, 32, false);
    return wide ? LC(128, 128, 4, false, 16, false) : LC(128, 64, 4, false, 16, false);
#undef LC
}
