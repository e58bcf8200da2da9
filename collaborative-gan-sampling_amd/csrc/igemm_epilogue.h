// Row passes of the wide epilogue shared by the implicit-GEMM kernels (igemm.hip: exact fp32 MFMA; igemm_bx6.hip: split-bf16
// MFMA): a wave has staged rows of its accumulator tile in LDS so that each lane owns 4 consecutive channels of a row.
#pragma once
#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row pass of the wide epilogue for ONE epilogue mode (compile-time): branch-free inner loops and a compact
// instruction footprint.  (With the mode as a run-time switch inside the unrolled loops the epilogue was ~21k lines
// of ISA with 1.6k branches and took 20 us per block, 10 % of the block's life.)
// SG: also leave the SIGN MASK of the stored values (p.sign_out; N % 32 == 0): the activation gradient of the layer above needs
// one bit per element, not the fp32 tensor.  A row's 4-channel lanes sit in consecutive lanes, so v_cmp of element e over the
// wave (ballot) holds, in the 8 bits starting at (lane & ~7), channels 4q + e (q = 0..7) of this lane's 32-channel group:
// word = sum_e byte_e << 8e, i.e. bit 8 * (c % 4) + (c % 32) / 4 <-> channel c; the lane with (lane & 7) == 0 stores it into
// the group's plane (one word per pixel, pixels contiguous: the consumer reads a tile row's words as one coalesced run).
template <int EPI, int ROWS, int RPP, int LDE, bool SG>
__device__ __forceinline__ void epilogue_rows_signs(const IgemmParams& p, const float* E, const int* rowpix_tile, int rsub, int c4,
                                                    int n, f32x4 bias, f32x4 ea, f32x4 eb) {
    const int lane = threadIdx.x & 63;
    unsigned* plane = p.sign_out + (size_t)(n >> 5) * p.sign_plane;
#pragma unroll
    for (int it = 0; it < ROWS / RPP; ++it) {
        const int lrow = it * RPP + rsub;
        const int pix = rowpix_tile[lrow];
        const bool live = pix >= 0;
        const f32x4 v = *(const f32x4*)(E + lrow * LDE + c4);
        const size_t o = (size_t)(live ? pix : 0) * p.N + n;
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = epilogue_apply(v[e] + bias[e], EPI, ea[e], eb[e], 0.f);
        if (live) *(f32x4*)(p.out + o) = y;
        unsigned word = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(live && y[e] > 0.f);
            word |= ((unsigned)(bal >> (lane & ~7)) & 0xffu) << (8 * e);
        }
        if (live && (lane & 7) == 0) plane[pix] = word;
    }
}

// per-lane parameters of the norm-backward statistics (IgemmParams::ns_*): this lane's four channels of its wave tile's statistics group
struct NsLane { f32x4 mu, inv, sc, sh; float leak; };
__device__ __forceinline__ NsLane ns_lane_params(const IgemmParams& p, int group, int n) {
    NsLane L;
    L.mu = *(const f32x4*)(p.ns_mean + (size_t)group * p.N + n);
    L.inv = *(const f32x4*)(p.ns_inv + (size_t)group * p.N + n);
    const f32x4 g = *(const f32x4*)(p.ns_gamma + n), b = *(const f32x4*)(p.ns_beta + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) { L.sc[e] = g[e] * L.inv[e]; L.sh[e] = fmaf(-L.mu[e], L.sc[e], b[e]); }      // (the affine as bn.hip's bn_affine4 forms it)
    L.leak = p.ns_leak;
    return L;
}

// ST: 0 = no statistics; 1 = fused norm statistics of the stored values (sum, sum of squares); 2 = norm-BACKWARD statistics: the stored
// value y is a gradient w.r.t. the output of norm + lrelu over x = ep_aux (same shape): sums of d = y * lrelu'(sc * x + sh) and d * xhat.
template <int EPI, int ROWS, int RPP, int LDE, int ST = 0, int NSU = 1>
__device__ __forceinline__ void epilogue_rows(const IgemmParams& p, const float* E, const int* rowpix_tile, int rsub, int c4,
                                              int n, f32x4 bias, f32x4 ea, f32x4 eb, f32x4* sa = nullptr, f32x4* sb = nullptr,
                                              const NsLane* ns = nullptr) {
    if constexpr (ST == 2 && NSU > 1) {
        // Norm-backward statistics: x is needed for the SUMS only, not for the value stored, and the compiler may not move a load of
        // ep_aux above an earlier store to out (it cannot know they do not overlap) -- so the rows go in groups of NSU: all x loads of a
        // group are issued first, then its rows are stored and summed (load -> use -> store per row held the block's slot for a memory
        // latency per row).  Same arithmetic, same order of additions per lane.  (NSU = 1, the generic-K kernels: row by row -- the group's
        // registers would be those kernels' peak and cost them a resident block)
        constexpr int ITERS = ROWS / RPP;
        static_assert(ITERS % NSU == 0, "row groups");
#pragma unroll
        for (int it0 = 0; it0 < ITERS; it0 += NSU) {
            f32x4 xs[NSU];
            int pixs[NSU];
#pragma unroll
            for (int u = 0; u < NSU; ++u) {
                pixs[u] = rowpix_tile[(it0 + u) * RPP + rsub];
                xs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (pixs[u] >= 0) xs[u] = *(const f32x4*)(p.ep_aux + (size_t)pixs[u] * p.N + n);
            }
#pragma unroll
            for (int u = 0; u < NSU; ++u) {
                if (pixs[u] < 0) continue;
                const f32x4 v = *(const f32x4*)(E + ((it0 + u) * RPP + rsub) * LDE + c4);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = v[e] + bias[e];
                *(f32x4*)(p.out + (size_t)pixs[u] * p.N + n) = y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // the arithmetic of bn_partial_kernel<1>, element by element
                    const float uu = fmaf(xs[u][e], ns->sc[e], ns->sh[e]);
                    const float d = y[e] * (uu > 0.f ? 1.f : ns->leak);
                    (*sa)[e] += d; (*sb)[e] = fmaf(d, (xs[u][e] - ns->mu[e]) * ns->inv[e], (*sb)[e]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < ROWS / RPP; ++it) {
        const int lrow = it * RPP + rsub;
        const int pix = rowpix_tile[lrow];
        if (pix < 0) continue;
        const f32x4 v = *(const f32x4*)(E + lrow * LDE + c4);
        const size_t o = (size_t)pix * p.N + n;
        f32x4 aux = {0.f, 0.f, 0.f, 0.f};
        if (EPI >= CGS_EPI_RELU_BWD_AFFINE || ST == 2) aux = *(const f32x4*)(p.ep_aux + o);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = epilogue_apply(v[e] + bias[e], EPI, ea[e], eb[e], aux[e]);
        *(f32x4*)(p.out + o) = y;
        if (ST == 1) {   // fused batch-norm statistics: this lane's 4 channels, summed over its rows
#pragma unroll
            for (int e = 0; e < 4; ++e) { (*sa)[e] += y[e]; (*sb)[e] = fmaf(y[e], y[e], (*sb)[e]); }
        }
        if (ST == 2) {   // (NSU == 1: row by row) the arithmetic of bn_partial_kernel<1>, element by element
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = fmaf(aux[e], ns->sc[e], ns->sh[e]);
                const float d = y[e] * (u > 0.f ? 1.f : ns->leak);
                (*sa)[e] += d; (*sb)[e] = fmaf(d, (aux[e] - ns->mu[e]) * ns->inv[e], (*sb)[e]);
            }
        }
    }
}
