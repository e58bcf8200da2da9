// Variant of bf16x6_256.hip (round 4): the 128 x 256 block with the six products issued as three v_mfma_f32_16x16x32_bf16 on CONCATENATED plane pairs
// ([a0|a1] x [b0|b1] = a0 b0 + a1 b1, ...) against the 32x32x16 form.  Measured: 144 vs 193 TFLOP/s fp32-equivalent (more fragment reads, 33 spilled registers): not adopted.
// Feasibility probe: fp32 GEMM through 6 bf16 MFMA products of 3-way split operands (x = x0+x1+x2 exactly,
// products with i+j <= 2 kept), against fp64 and against an fp32 fma chain.   C[M][N] = A[M][K] * B[K][N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
#ifdef X_NOSPLIT
    p0 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302); p1 = p0; p2 = p0; return;
#endif
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    p0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302);
}

// B pre-split: Bp[plane][K/BK][N][BK] bf16
template <int BK>
__global__ void pack_b(const float* __restrict__ B, unsigned short* __restrict__ Bp, int K, int N) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * N) return;
    const int kk = i % BK; const long t = i / BK; const int n = t % N; const int kc = t / N;
    const float x = B[(long)(kc * BK + kk) * N + n];
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);
    const unsigned v = __builtin_bit_cast(unsigned, r);
    const float q = r - __builtin_bit_cast(float, v & 0xffff0000u);
    const long plane = (long)K * N;
    Bp[i] = u >> 16; Bp[plane + i] = v >> 16; Bp[2 * plane + i] = __builtin_bit_cast(unsigned, q) >> 16;
}



// BM x 256 block tile (BM = 256: 8 waves, one block per CU; BM = 128: 4 waves, TWO independent blocks per CU, which do not run in
// lockstep), wave tile 128 x 64 (4 x 2 MFMA tiles), 16-deep stages, double-buffered LDS with 32-byte rows (a fragment read is a
// linear 1 KB: conflict-free without padding), A split in-kernel, B pre-split, global loads two stages ahead.
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <int BM, bool S16 = false>
__global__ __launch_bounds__(BM * 2, BM == 128 ? 2 : 1) void gemm_bf16x6_t(const float* __restrict__ A, const unsigned short* __restrict__ Bp,
                                                         float* __restrict__ C, int M, int K, int N, int Kuniq) {
    constexpr int BK = 16, BN = 256, NT = BM * 2;
    constexpr int PITCH = 32;
    constexpr int PLA = BM * PITCH, PLB = BN * PITCH;   // bytes of an A / B plane
    constexpr int BUF = 3 * PLA + 3 * PLB;
    constexpr int NB = BN * 2 / NT;                     // B (column, half) pairs per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int arow = tid >> 2, aq = tid & 3;          // A loader role: rows arow and arow + BM / 2, float4 number aq of the 16-deep chunk
    const float* ap = A + (long)(m0 + arow) * Kuniq + aq * 4;
    const int nku = Kuniq / BK;
    const long planeB = (long)K * N;
    f32x4 ra[2][2];
    u32x4 rb[NB][3];                                   // (B: one stage ahead; A: two)
    f32x16 acc[S16 ? 1 : 4][S16 ? 1 : 2];
    f32x4acc acc16[S16 ? 8 : 1][S16 ? 4 : 1];
    if constexpr (!S16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
    }
    const int nk = K / BK;
#define LOADT(kt, S)                                                                                 \
    ra[S][0] = *(const f32x4*)(ap + (long)((kt) % nku) * BK); ra[S][1] = *(const f32x4*)(ap + (long)(BM / 2) * Kuniq + (long)((kt) % nku) * BK);
#define LOADB(kt)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < NB; ++u) {                                               \
        const int idx = tid + NT * u;                                                              \
        const unsigned short* bp = Bp + ((long)(n0 + (idx >> 1))) * BK + (idx & 1) * 8 + (long)(kt) * N * BK; \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) rb[u][p] = *(const u32x4*)(bp + p * planeB); \
    }
#define STORET(buf, S)                                                                               \
    {                                                                                              \
        unsigned char* base = lds + (buf) * BUF;                                                   \
        unsigned q0[4], q1[4], q2[4];                                                              \
        split2(ra[S][0][0], ra[S][0][1], q0[0], q1[0], q2[0]);                                     \
        split2(ra[S][0][2], ra[S][0][3], q0[1], q1[1], q2[1]);                                     \
        split2(ra[S][1][0], ra[S][1][1], q0[2], q1[2], q2[2]);                                     \
        split2(ra[S][1][2], ra[S][1][3], q0[3], q1[3], q2[3]);                                     \
        const int offa = arow * PITCH + aq * 8;                                                    \
        *(u32x2*)(base + 0 * PLA + offa) = u32x2{q0[0], q0[1]}; *(u32x2*)(base + 0 * PLA + offa + (BM / 2) * PITCH) = u32x2{q0[2], q0[3]}; \
        *(u32x2*)(base + 1 * PLA + offa) = u32x2{q1[0], q1[1]}; *(u32x2*)(base + 1 * PLA + offa + (BM / 2) * PITCH) = u32x2{q1[2], q1[3]}; \
        *(u32x2*)(base + 2 * PLA + offa) = u32x2{q2[0], q2[1]}; *(u32x2*)(base + 2 * PLA + offa + (BM / 2) * PITCH) = u32x2{q2[2], q2[3]}; \
        _Pragma("unroll") for (int u = 0; u < NB; ++u) {                                           \
            const int idx = tid + NT * u;                                                          \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) *(u32x4*)(base + 3 * PLA + p * PLB + idx * 16) = rb[u][p]; \
        }                                                                                          \
    }
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* fa = lds + (wm * 128 + fr) * PITCH + fh * 16;
    const unsigned char* fb = lds + 3 * PLA + (wn * 64 + fr) * PITCH + fh * 16;
#define MM(pa, pb)                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
    // 16x16x32 form: one MFMA contracts TWO plane products over the 16-deep stage: A operand [p | p'] = lanes 0-31 read plane p (k 0-7 | 8-15),
    // lanes 32-63 plane p'; B likewise: (a0|a1)x(b0|b1) = a0b0 + a1b1, (a0|a1)x(b1|b0) = a0b1 + a1b0, (a0|a2)x(b2|b0) = a0b2 + a2b0
    const int r16 = lane & 15, kh16 = (lane >> 4) & 1, up = lane >> 5;
    const unsigned char* fa16 = lds + (wm * 128 + r16) * PITCH + kh16 * 16;
    const unsigned char* fb16 = lds + 3 * PLA + (wn * 64 + r16) * PITCH + kh16 * 16;
    const int a01 = up * PLA, a02 = up * 2 * PLA;                 // [a0|a1], [a0|a2]
    const int b01 = up * PLB, b10 = (1 - up) * PLB, b20 = (1 - up) * 2 * PLB;     // [b0|b1], [b1|b0], [b2|b0]
#define COMPUTE16(cur)                                                                               \
    {                                                                                              \
        _Pragma("unroll") for (int ih = 0; ih < 2; ++ih) {                                         \
            bf16x8 xa[4][2];                                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                        \
                xa[i][0] = *(const bf16x8*)(fa16 + (cur) * BUF + a01 + (ih * 4 + i) * 16 * PITCH); \
                xa[i][1] = *(const bf16x8*)(fa16 + (cur) * BUF + a02 + (ih * 4 + i) * 16 * PITCH); \
            }                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
                const bf16x8 y0 = *(const bf16x8*)(fb16 + (cur) * BUF + b01 + j * 16 * PITCH);     \
                const bf16x8 y1 = *(const bf16x8*)(fb16 + (cur) * BUF + b10 + j * 16 * PITCH);     \
                const bf16x8 y2 = *(const bf16x8*)(fb16 + (cur) * BUF + b20 + j * 16 * PITCH);     \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
                    acc16[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], y0, acc16[ih * 4 + i][j], 0, 0, 0); \
                    acc16[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][0], y1, acc16[ih * 4 + i][j], 0, 0, 0); \
                    acc16[ih * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i][1], y2, acc16[ih * 4 + i][j], 0, 0, 0); \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
    }
#define COMPUTE(cur)                                                                                 \
    {                                                                                              \
        bf16x8 a[4][3], b[2][3];                                                                   \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) a[i][p] = *(const bf16x8*)(fa + (cur) * BUF + p * PLA + i * 32 * PITCH); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) b[j][p] = *(const bf16x8*)(fb + (cur) * BUF + p * PLB + j * 32 * PITCH); \
        }                                                                                          \
        MM(0, 0) MM(0, 1) MM(1, 0) MM(1, 1) MM(0, 2) MM(2, 0)                                      \
    }
    LOADT(0, 0); LOADB(0);
    STORET(0, 0);
    LOADT(nk > 1 ? 1 : 0, 1);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {                   // (nk even)
        { const int k2 = kt + 2 < nk ? kt + 2 : nk - 1; LOADT(k2, 0); LOADB(kt + 1); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S16) COMPUTE16(0) else COMPUTE(0);
        STORET(1, 1);
        __syncthreads();
        { const int k3 = kt + 3 < nk ? kt + 3 : nk - 1; LOADT(k3, 1); LOADB(kt + 2 < nk ? kt + 2 : nk - 1); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S16) COMPUTE16(1) else COMPUTE(1);
        STORET(0, 0);
        __syncthreads();
    }
    if constexpr (!S16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 128 + i * 32 + (e >> 2) * 8 + fh * 4 + (e & 3);
                const int col = n0 + wn * 64 + j * 32 + fr;
                C[(long)row * N + col] = acc[i][j][e];
            }
    } else {          // C layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + register
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wm * 128 + i * 16 + 4 * (lane >> 4) + e;
                const int col = n0 + wn * 64 + j * 16 + (lane & 15);
                C[(long)row * N + col] = acc16[i][j][e];
            }
    }
#undef LOADT
#undef LOADB
#undef STORET
#undef MM
#undef COMPUTE
#undef COMPUTE16
}

static void check(const float* dC, int N, int K, int Ku, const std::vector<float>& hA, const std::vector<float>& hB) {
    std::vector<float> hC((size_t)256 * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double e6 = 0, e32 = 0, s6 = 0, s32 = 0; long cnt = 0;
    for (int r = 0; r < 256; r += 5)
        for (int c = 0; c < N; c += 3) {
            double ref = 0, absum = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) {
                ref += (double)hA[(size_t)r * Ku + k % Ku] * hB[(size_t)k * N + c];
                absum += fabs((double)hA[(size_t)r * Ku + k % Ku] * hB[(size_t)k * N + c]);
                f = fmaf(hA[(size_t)r * Ku + k % Ku], hB[(size_t)k * N + c], f);
            }
            const double d6 = fabs(hC[(size_t)r * N + c] - ref) / absum, d32 = fabs(f - ref) / absum;
            e6 = fmax(e6, d6); e32 = fmax(e32, d32); s6 += d6; s32 += d32; ++cnt;
        }
    printf("   error / sum|a||b|:  bf16x6 max %.3e mean %.3e   |  fp32 fma chain max %.3e mean %.3e   (2^-24 = 5.96e-08)\n", e6, s6 / cnt, e32, s32 / cnt);
}

template <int BM, bool S16 = false>
static void run(const float* dA, const unsigned short* dBp, float* dC, int M, int K, int N, int Ku, const std::vector<float>& hA, const std::vector<float>& hB) {
    const size_t smem = 2 * (3 * BM * 32 + 3 * 256 * 32);
    CK(hipFuncSetAttribute((const void*)gemm_bf16x6_t<BM, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(M / BM, N / 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_bf16x6_t<BM, S16>), grid, dim3(BM * 2), smem, 0, dA, dBp, dC, M, K, N, Ku);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_bf16x6_t<BM, S16>), grid, dim3(BM * 2), smem, 0, dA, dBp, dC, M, K, N, Ku);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = 2.0 * M * K * N;
    printf("%s %dx256x16, %d waves: %d x %d (unique %d) x %d: %.1f us  %.1f TFLOP/s fp32-equivalent (%.0f TFLOP/s bf16 executed), smem %zu\n", S16 ? "[16x16x32 pairs]" : "[32x32x16]", BM, BM / 32, M, K, Ku, N,
           ms * 1e3, fl / ms / 1e9, 6 * fl / ms / 1e9, smem);
    check(dC, N, K, Ku, hA, hB);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, K = argc > 2 ? atoi(argv[2]) : 3200, N = argc > 3 ? atoi(argv[3]) : 256, Ku = argc > 4 ? atoi(argv[4]) : K;
    std::vector<float> hA((size_t)M * Ku), hB((size_t)K * N);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto& v : hA) v = rnd() * (1.f + 3.f * fabsf(rnd()));
    for (auto& v : hB) v = rnd() * 0.05f;
    float *dA, *dB, *dC; unsigned short* dBp;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dBp, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_b<16>, dim3((unsigned)(((long)K * N + 255) / 256)), dim3(256), 0, 0, dB, dBp, K, N);
    run<128>(dA, dBp, dC, M, K, N, Ku, hA, hB);
    run<128, true>(dA, dBp, dC, M, K, N, Ku, hA, hB);
    run<128>(dA, dBp, dC, M, K, N, Ku, hA, hB);
    run<128, true>(dA, dBp, dC, M, K, N, Ku, hA, hB);
    return 0;
}
