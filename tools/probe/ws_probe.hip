// Probe v2: wave-specialised bf16x6 GEMM.  Block = 8 waves: waves 0-3 only read fragments and issue MFMAs (64x64 each of a
// 128x128 tile); waves 4-7 load fp32 A + pre-split B, split A into 3 bf16 planes and fill the 2-stage LDS ring two K-steps ahead.
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
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int BK = 16;
constexpr int PITCH = BK * 2 + 16;     // 48 B per row per plane
constexpr int BM = 256, NMW = 8;        // rows per block, MFMA waves (64x64 each, 4 x 2)
constexpr int PLANEA = BM * PITCH, PLANEB = 128 * PITCH;
constexpr int STAGE = 3 * PLANEA + 3 * PLANEB;       // A planes, then B planes

__device__ __forceinline__ void split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    p0 = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float q0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302);
}

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

struct Tile { f32x4 a[BM / 64]; u32x4 b[3]; };
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__((NMW + 4) * 64, 1) void gemm_ws(const float* __restrict__ A, const unsigned short* __restrict__ Bp,
                                                  float* __restrict__ C, int M, int K, int N, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SWZ
    // XCD-aware: id%8 = XCD; inside an XCD consecutive blocks walk a (8 m) x (nN n) patch so co-resident blocks share A and B tiles
    const int nN = N / 128, id = blockIdx.x + gridDim.x * blockIdx.y, xcd = id & 7, loc = id >> 3;
    const int per = 8 * nN, grp = loc / per, rem = loc % per;
    const int mt = (grp * 8 + xcd) * 8 + rem % 8, nt = rem / 8;
    const int m0 = mt * BM, n0 = nt * 128;
    if (m0 >= M) return;
#else
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * 128;
#endif
    const int nk = K / BK;
    if (wave >= NMW) {
        // ---------------- loader role ----------------
#ifdef LPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        const int t = tid - NMW * 64, lrow = t >> 1, lhalf = t & 1;
        const int arow = t >> 2, aq = t & 3;                      // A: 4 lanes cover a row's 64-byte chunk (coalesced)
        const float* ap = A + (long)(m0 + arow) * K + aq * 4;
        unsigned char* awbase = lds + arow * PITCH + aq * 8;
        const long planeB = (long)K * N;
        const unsigned short* bp = Bp + ((long)(n0 + lrow)) * BK + lhalf * 8;
        unsigned char* wbase = lds + lrow * PITCH + lhalf * 16;
        Tile T[4];
        // hand-issued loads: the compiler's own vmcnt bookkeeping would wait for (almost) everything at the loop header,
        // which collapses the prefetch depth to one K-step; these are waited for explicitly with a counted vmcnt
#ifdef SAMEK
#define KCSEL(kt) ((kt) & 1)
#else
#define KCSEL(kt) ((kt) < nk ? (kt) : nk - 1)
#endif
#define GLD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define LOADT(kt, S)                                                                              \
        {                                                                                          \
            const int kc = KCSEL(kt);                                              \
            _Pragma("unroll") for (int r = 0; r < BM / 64; ++r) GLD(T[S].a[r], ap + (long)r * 64 * K + (long)kc * BK); \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) GLD(T[S].b[p], bp + p * planeB + (long)kc * N * BK); \
        }
        // wait until at most NOUT of this wave's loads are outstanding; the tile's registers pass through the asm so that
        // nothing reading them can be scheduled above the wait
#define WAITT(S, NOUT)                                                                             \
        asm volatile("s_waitcnt vmcnt(%7)" : "+v"(T[S].a[0]), "+v"(T[S].a[1]), "+v"(T[S].a[2]), "+v"(T[S].a[3]), \
                     "+v"(T[S].b[0]), "+v"(T[S].b[1]), "+v"(T[S].b[2]) : "n"(NOUT) : "memory")
#define STORET(stage, S)                                                                          \
        {                                                                                          \
            unsigned char* base = wbase + (stage) * STAGE;                                         \
            _Pragma("unroll") for (int r = 0; r < BM / 64; ++r) {                                   \
                unsigned q0[2], q1[2], q2[2];                                                      \
                split2(T[S].a[r][0], T[S].a[r][1], q0[0], q1[0], q2[0]);                           \
                split2(T[S].a[r][2], T[S].a[r][3], q0[1], q1[1], q2[1]);                           \
                unsigned char* ab = awbase + (stage) * STAGE + r * 64 * PITCH;                     \
                *(u32x2*)(ab + 0 * PLANEA) = (u32x2){q0[0], q0[1]};                                \
                *(u32x2*)(ab + 1 * PLANEA) = (u32x2){q1[0], q1[1]};                                \
                *(u32x2*)(ab + 2 * PLANEA) = (u32x2){q2[0], q2[1]};                                \
            }                                                                                      \
            *(u32x4*)(base + 3 * PLANEA + 0 * PLANEB) = T[S].b[0];                                 \
            *(u32x4*)(base + 3 * PLANEA + 1 * PLANEB) = T[S].b[1];                                 \
            *(u32x4*)(base + 3 * PLANEA + 2 * PLANEB) = T[S].b[2];                                 \
        }
        LOADT(0, 0); LOADT(1, 1); LOADT(2, 2); LOADT(3, 3);
        WAITT(0, 21); STORET(0, 0);
        WAITT(1, 14); STORET(1, 1);
        LOADT(4, 0);                                             // in flight now: tiles 2,3,4 (sets 2,3,0)
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): LDS writes landed (vmcnt left alone)
        __builtin_amdgcn_s_barrier();                            // P: tiles 0,1 in LDS
        // step k: issue tile k+5 into set (k+5)%4 = (k+1)%4, then wait for tile k+2 (3 newer tiles = 21 loads may stay
        // outstanding), split it and write it into stage k%2
        unsigned long long ph[5] = {0, 0, 0, 0, 0};
        for (int kt = 0; kt < nk; kt += 4) {
#define LSTEP(k, S2, S5)                                                                           \
            if ((k) < nk) {                                                                        \
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();                        \
                __builtin_amdgcn_s_barrier();                                                      \
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();                        \
                LOADT((k) + 5, S5);                                                                \
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();                        \
                WAITT(S2, 21);                                                                     \
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();                        \
                STORET((k) & 1, S2);                                                               \
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();                        \
                __builtin_amdgcn_s_waitcnt(0xc07f);                                                \
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();                        \
                ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; \
            }
            LSTEP(kt, 2, 1);
            LSTEP(kt + 1, 3, 2);
            LSTEP(kt + 2, 0, 3);
            LSTEP(kt + 3, 1, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamps && t == 0 && blockIdx.y == 0 && blockIdx.x < 8) for (int i = 0; i < 5; ++i) stamps[2 * (M / BM) + blockIdx.x * 5 + i] = ph[i];
        return;
    }
    // ---------------- MFMA role ----------------
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* fa = lds + (wm * 64 + fr) * PITCH + fh * 16;
    const unsigned char* fb = lds + 3 * PLANEA + (wn * 64 + fr) * PITCH + fh * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 Fa[2][2][3], Fb[2][2][3];
#define READF(stage, S)                                                                            \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                             \
            Fa[S][i][p] = *(const bf16x8*)(fa + (stage) * STAGE + p * PLANEA + i * 32 * PITCH);       \
            Fb[S][i][p] = *(const bf16x8*)(fb + (stage) * STAGE + p * PLANEB + i * 32 * PITCH);       \
        }
#define MFMAS(S)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fa[S][i][0], Fb[S][j][2], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fa[S][i][2], Fb[S][j][0], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fa[S][i][1], Fb[S][j][1], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fa[S][i][0], Fb[S][j][1], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fa[S][i][1], Fb[S][j][0], acc[i][j], 0, 0, 0); \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fa[S][i][0], Fb[S][j][0], acc[i][j], 0, 0, 0); \
        }
    __builtin_amdgcn_s_barrier();                                // P
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    READF(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
#define MSTEP(k, S)                                                                                \
        if ((k) < nk) {                                                                            \
            __builtin_amdgcn_s_waitcnt(0xc07f);                  /* fragments of tile k are in registers */ \
            __builtin_amdgcn_s_barrier();                        /* tile k+1 is in LDS, stage k%2 may be overwritten */ \
            READF(((k) + 1) & 1, (S) ^ 1);                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            MFMAS(S);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
        MSTEP(kt, 0);
        MSTEP(kt + 1, 1);
    }
    if (stamps && tid == 0 && n0 == 0) { stamps[2 * (m0 / BM)] = __builtin_amdgcn_s_memtime() - c0; stamps[2 * (m0 / BM) + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e >> 2) * 8 + fh * 4 + (e & 3);
                const int col = n0 + wn * 64 + j * 32 + fr;
                C[(long)row * N + col] = acc[i][j][e];
            }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, K = argc > 2 ? atoi(argv[2]) : 2048, N = argc > 3 ? atoi(argv[3]) : 256;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto& v : hA) v = rnd() * (1.f + 3.f * fabsf(rnd()));
    for (auto& v : hB) v = rnd() * 0.05f;
    float *dA, *dB, *dC; unsigned short* dBp;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dBp, hB.size() * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_b, dim3((unsigned)(((long)K * N + 255) / 256)), dim3(256), 0, 0, dB, dBp, K, N);
    const size_t smem = 2 * STAGE;
    CK(hipFuncSetAttribute((const void*)gemm_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(M / BM, N / 128);
    unsigned long long* dS; CK(hipMalloc(&dS, (size_t)(M / BM) * 16 + 8 * 5 * 8)); CK(hipMemset(dS, 0, (size_t)(M / BM) * 16 + 320));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_ws, grid, dim3((NMW + 4) * 64), smem, 0, dA, dBp, dC, M, K, N, dS);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_ws, grid, dim3((NMW + 4) * 64), smem, 0, dA, dBp, dC, M, K, N, dS);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = 2.0 * M * K * N;
    printf("ws BM=256 BK=%d  %d x %d x %d: %.1f us  %.1f TFLOP/s fp32-equivalent (%.0f TFLOP/s bf16 executed), smem %zu\n", BK, M, K, N, ms * 1e3,
           fl / ms / 1e9, 6 * fl / ms / 1e9, smem);
    { std::vector<unsigned long long> hS((size_t)(M / BM) * 2); CK(hipMemcpy(hS.data(), dS, hS.size() * 8, hipMemcpyDeviceToHost));
      double cyc = 0, rt = 0; for (size_t i = 0; i < hS.size(); i += 2) { cyc += hS[i]; rt += hS[i + 1]; }
      printf("   in-kernel: %.0f shader cycles per block K loop, clock %.2f GHz, MFMA issue share %.2f\n", cyc / (hS.size() / 2), cyc / rt * 0.1, (double)(K / BK) * 24 * 32 * (hS.size() / 2) / cyc); }
    { unsigned long long ph[40]; CK(hipMemcpy(ph, dS + 2 * (M / BM), 320, hipMemcpyDeviceToHost));
      printf("   loader cycles per step (block 0): barrier-wait %.0f | issue loads %.0f | vmcnt wait %.0f | split+issue stores %.0f | lgkm wait %.0f\n", ph[0] / (double)(K / BK), ph[1] / (double)(K / BK), ph[2] / (double)(K / BK), ph[3] / (double)(K / BK), ph[4] / (double)(K / BK)); }
    std::vector<float> hC((size_t)256 * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double e6 = 0, s6 = 0; long cnt = 0;
    for (int r = 0; r < 256; r += 5)
        for (int c = 0; c < N; c += 3) {
            double ref = 0, absum = 0;
            for (int k = 0; k < K; ++k) { const double pr = (double)hA[(size_t)r * K + k] * hB[(size_t)k * N + c]; ref += pr; absum += fabs(pr); }
            const double d6 = fabs(hC[(size_t)r * N + c] - ref) / absum;
            e6 = fmax(e6, d6); s6 += d6; ++cnt;
        }
    printf("   error / sum|a||b|:  max %.3e mean %.3e\n", e6, s6 / cnt);
    return 0;
}
