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

template <int BK>
__global__ __launch_bounds__(256) void gemm_bf16x6(const float* __restrict__ A, const unsigned short* __restrict__ Bp,
                                                   float* __restrict__ C, int M, int K, int N) {
    constexpr int PITCH = BK * 2 + 16;                 // bytes per row per plane (80 or 48): conflict-free b128 reads
    constexpr int PLANE = 128 * PITCH;
    constexpr int BUF = 6 * PLANE;                     // A planes 0..2, B planes 3..5
    constexpr int NV = BK / 8;                         // f32x4 per thread for A (half a row), = b128 per plane /2 ...
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int lrow = tid >> 1, lhalf = tid & 1;        // loader role: row (A) / column (B), which half of the BK chunk
    const float* ap = A + (long)(m0 + lrow) * K + lhalf * (BK / 2);
    const long planeB = (long)K * N;
    const unsigned short* bp = Bp + ((long)(n0 + lrow)) * BK + lhalf * (BK / 2);
    f32x4 raS[2][NV];
    u32x4 rbS[2][3][BK / 16];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / BK;
#define LOADT(kt, S)                                                                                  \
    _Pragma("unroll") for (int v = 0; v < NV; ++v) raS[S][v] = *(const f32x4*)(ap + (long)(kt) * BK + v * 4); \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                   \
        _Pragma("unroll") for (int v = 0; v < BK / 16; ++v)                                         \
            rbS[S][p][v] = *(const u32x4*)(bp + p * planeB + (long)(kt) * N * BK + v * 8);
#define STORET(buf, S)                                                                                \
    {                                                                                              \
        unsigned char* base = lds + (buf) * BUF;                                                   \
        _Pragma("unroll") for (int v = 0; v < BK / 16; ++v) {                                       \
            unsigned q0[4], q1[4], q2[4];                                                          \
            split2(raS[S][2 * v][0], raS[S][2 * v][1], q0[0], q1[0], q2[0]);                               \
            split2(raS[S][2 * v][2], raS[S][2 * v][3], q0[1], q1[1], q2[1]);                               \
            split2(raS[S][2 * v + 1][0], raS[S][2 * v + 1][1], q0[2], q1[2], q2[2]);                       \
            split2(raS[S][2 * v + 1][2], raS[S][2 * v + 1][3], q0[3], q1[3], q2[3]);                       \
            const u32x4 w0 = {q0[0], q0[1], q0[2], q0[3]}, w1 = {q1[0], q1[1], q1[2], q1[3]}, w2 = {q2[0], q2[1], q2[2], q2[3]}; \
            const int off = lrow * PITCH + lhalf * BK + v * 16;                                    \
            *(u32x4*)(base + 0 * PLANE + off) = w0;                                                \
            *(u32x4*)(base + 1 * PLANE + off) = w1;                                                \
            *(u32x4*)(base + 2 * PLANE + off) = w2;                                                \
            *(u32x4*)(base + 3 * PLANE + off) = rbS[S][0][v];                                          \
            *(u32x4*)(base + 4 * PLANE + off) = rbS[S][1][v];                                          \
            *(u32x4*)(base + 5 * PLANE + off) = rbS[S][2][v];                                          \
        }                                                                                          \
    }
    LOADT(0, 0);
    STORET(0, 0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned char* fa = lds + (wm * 64 + fr) * PITCH + fh * 16;
    const unsigned char* fb = lds + 3 * PLANE + (wn * 64 + fr) * PITCH + fh * 16;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int ktn = kt + 1 < nk ? kt + 1 : kt;            // branch-free: the last step re-loads its own tile
        LOADT(ktn, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *(const bf16x8*)(fa + cur * BUF + p * PLANE + i * 32 * PITCH + s * 32);
                    b[i][p] = *(const bf16x8*)(fb + cur * BUF + p * PLANE + i * 32 * PITCH + s * 32);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
        STORET(cur ^ 1, 0);
        __syncthreads();
    }
    // C layout of 32x32 MFMA: lane -> col = lane%32, rows 8*(e/4) + 4*? ... : row = (e/4)*8 + (lane/32)*4 + e%4
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

template <int BK>
static void run(const float* dA, const float* dB, float* dC, unsigned short* dBp, int M, int K, int N, const std::vector<float>& hA,
                const std::vector<float>& hB) {
    hipLaunchKernelGGL(pack_b<BK>, dim3((unsigned)(((long)K * N + 255) / 256)), dim3(256), 0, 0, dB, dBp, K, N);
    constexpr int PITCH = BK * 2 + 16;
    const size_t smem = 2 * 6 * 128 * PITCH;
    CK(hipFuncSetAttribute((const void*)gemm_bf16x6<BK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 grid(M / 128, N / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_bf16x6<BK>, grid, dim3(256), smem, 0, dA, dBp, dC, M, K, N);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_bf16x6<BK>, grid, dim3(256), smem, 0, dA, dBp, dC, M, K, N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double fl = 2.0 * M * K * N;
    printf("BK=%d  %d x %d x %d: %.1f us  %.1f TFLOP/s fp32-equivalent (%.0f TFLOP/s bf16 executed), smem %zu\n", BK, M, K, N, ms * 1e3,
           fl / ms / 1e9, 6 * fl / ms / 1e9, smem);
    // accuracy on a sample of rows
    std::vector<float> hC((size_t)256 * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double e6 = 0, e32 = 0, s6 = 0, s32 = 0, nrm = 0; long cnt = 0;
    for (int r = 0; r < 256; r += 5)
        for (int c = 0; c < N; c += 3) {
            double ref = 0, absum = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) {
                ref += (double)hA[(size_t)r * K + k] * hB[(size_t)k * N + c];
                absum += fabs((double)hA[(size_t)r * K + k] * hB[(size_t)k * N + c]);
                f = fmaf(hA[(size_t)r * K + k], hB[(size_t)k * N + c], f);
            }
            const double d6 = fabs(hC[(size_t)r * N + c] - ref) / absum, d32 = fabs(f - ref) / absum;
            e6 = fmax(e6, d6); e32 = fmax(e32, d32); s6 += d6; s32 += d32; nrm += absum; ++cnt;
        }
    printf("   error / sum|a||b|:  bf16x6 max %.3e mean %.3e   |  fp32 fma chain max %.3e mean %.3e   (2^-24 = 5.96e-08)\n", e6, s6 / cnt, e32, s32 / cnt);
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
    run<32>(dA, dB, dC, dBp, M, K, N, hA, hB);
    run<16>(dA, dB, dC, dBp, M, K, N, hA, hB);
    return 0;
}
