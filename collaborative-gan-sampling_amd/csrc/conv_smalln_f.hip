// Stride-1 'SAME' convolution with <= 4 OUTPUT channels (the c7s1-3 RGB head of a CycleGAN-style generator,
// BASELINE config 5).  N = 3 fills 3 of 32 MFMA columns and the f32 matrix rate equals the f32 vector rate on gfx950,
// so this runs on the VALU: a block owns a 16 x 16 tile of output pixels of one image, stages the (16+k-1)^2 input
// patch in LDS one 16-channel chunk at a time (next chunk's global loads in flight under the FMAs; pixel pitch of
// 20 floats keeps the 16-byte patch reads bank-conflict free), each thread accumulates its pixel's N outputs, and the
// weights are wave-uniform scalar loads.
#include "cgs_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SmallNFParams {
    const float* in;      // [B,H,W,Cb]
    const float* wp;      // packed [kh*kw][Cb/16][N][16]: a channel pair's two weights sit in adjacent (scalar) registers
    const float* bias;
    float* out;           // [B,H,W,N]
    int B, H, W, Cb, kh, kw, pt, pl, epilogue;
};

static constexpr int ST = 16;          // tile side (pixels)
static constexpr int SPP = 20;         // LDS floats per patch pixel (16 channels + 4 pad)

template <int N>
__global__ __launch_bounds__(256, 2) void conv_smalln_f_kernel(SmallNFParams p) {
    extern __shared__ __attribute__((aligned(16))) float patch[];      // [PH][PW][SPP]
    const int tid = threadIdx.x;
    const int PH = ST + p.kh - 1, PW = ST + p.kw - 1;
    const int tiles_x = p.W / ST, tiles_y = p.H / ST;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * ST, ox0 = tx * ST;
    const int py = tid >> 4, px = tid & 15;

    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.B * (unsigned)p.H * (unsigned)p.W * (unsigned)p.Cb * 4u), 0x00020000);
    constexpr int PL = 8;                                    // float4 per thread per chunk: ceil(22*22*4 / 256)
    const int npf4 = PH * PW * 4;
    f32x4 st[PL];
#define SN_LOAD(ch_)                                                                                      \
    _Pragma("unroll") for (int u = 0; u < PL; ++u) {                                                       \
        const int q = tid + 256 * u;                                                                       \
        unsigned off = 0xFFFFFFF0u;                                                                        \
        if (q < npf4) {                                                                                    \
            const int pixel = q >> 2, c4 = q & 3;                                                          \
            const int pr = pixel / PW, pc = pixel - pr * PW;                                               \
            const int iy = oy0 - p.pt + pr, ix = ox0 - p.pl + pc;                                          \
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)                              \
                off = (unsigned)(((b * p.H + iy) * p.W + ix) * p.Cb + (ch_) * 16 + c4 * 4) * 4u;            \
        }                                                                                                  \
        st[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));      \
    }
#define SN_STORE()                                                                                         \
    _Pragma("unroll") for (int u = 0; u < PL; ++u) {                                                       \
        const int q = tid + 256 * u;                                                                       \
        if (q < npf4) *(f32x4*)(patch + (q >> 2) * SPP + (q & 3) * 4) = st[u];                             \
    }

    f32x2 acc[N];                                            // even / odd channel partial sums (v_pk_fma_f32)
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = (f32x2){0.f, 0.f};
    const int nchunk = p.Cb >> 4;
    SN_LOAD(0);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();                                     // previous chunk's reads are done
        SN_STORE();
        __syncthreads();
        if (ch + 1 < nchunk) { SN_LOAD(ch + 1); }
        for (int ky = 0; ky < p.kh; ++ky)
            for (int kx = 0; kx < p.kw; ++kx) {
                const float* src = patch + ((py + ky) * PW + (px + kx)) * SPP;
                const float* wt = p.wp + ((size_t)(ky * p.kw + kx) * nchunk + ch) * (N * 16);   // uniform -> scalar loads
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x4 v = *(const f32x4*)(src + c4 * 4);
#pragma unroll
                    for (int n = 0; n < N; ++n) {
                        const f32x2 w01 = *(const f32x2*)(wt + n * 16 + c4 * 4), w23 = *(const f32x2*)(wt + n * 16 + c4 * 4 + 2);
                        acc[n] = __builtin_elementwise_fma(v.xy, w01, acc[n]);
                        acc[n] = __builtin_elementwise_fma(v.zw, w23, acc[n]);
                    }
                }
            }
    }
#undef SN_LOAD
#undef SN_STORE
    float* o = p.out + ((size_t)(b * p.H + oy0 + py) * p.W + ox0 + px) * N;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float v = (acc[n].x + acc[n].y) + (p.bias ? p.bias[n] : 0.f);
        if (p.epilogue == CGS_EPI_TANH) v = tanhf(v);
        else if (p.epilogue == CGS_EPI_LRELU) v = fmaxf(v, 0.2f * v);
        o[n] = v;
    }
}

__global__ void pack_smalln_f_kernel(const float* __restrict__ w, float* __restrict__ wp, int taps, int Cb, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // over the packed layout
    if (i >= taps * Cb * N) return;
    const int c = i & 15, n = (i >> 4) % N, tc = (i >> 4) / N, ch = tc % (Cb >> 4), tap = tc / (Cb >> 4);
    wp[i] = w[((size_t)tap * Cb + ch * 16 + c) * N + n];
}

size_t cgs_conv_smalln_f_ws_floats(const CgsLayer& L) { return (size_t)L.kh * L.kw * L.Cb * L.Cs; }

int cgs_conv_smalln_f_ok(const CgsLayer& L, int B, int epilogue) {
    // a 16 x 16 tile per block: below ~2 blocks per CU the MFMA path (N padded) is the faster one
    return (long)B * (L.Hb / ST) * (L.Wb / ST) >= 512 && L.Cs <= 4 && L.sh == 1 && L.sw == 1 && (L.Cb % 16) == 0 && (L.Hb % ST) == 0 && (L.Wb % ST) == 0 && L.kh <= 7 &&
           L.kw <= 7 && (epilogue == CGS_EPI_NONE || epilogue == CGS_EPI_TANH || epilogue == CGS_EPI_LRELU);
}

int cgs_conv_smalln_f_launch(const CgsLayer& L, int B, const float* in, const float* w, const float* bias, float* out,
                             int epilogue, float* ws, size_t ws_bytes, int prepacked, hipStream_t s) {
    const size_t nw = cgs_conv_smalln_f_ws_floats(L);
    if (ws_bytes < nw * sizeof(float)) return cgs_set_error(CGS_EWORKSPACE, "conv_smalln_f: workspace %zu < %zu bytes", ws_bytes, nw * sizeof(float));
    if (!prepacked) {
        hipLaunchKernelGGL(pack_smalln_f_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, w, ws, L.kh * L.kw, L.Cb, L.Cs);
        CGS_CHECK_LAUNCH("pack_smalln_f");
    }
    SmallNFParams p;
    p.in = in; p.wp = ws; p.bias = bias; p.out = out; p.B = B; p.H = L.Hb; p.W = L.Wb; p.Cb = L.Cb; p.kh = L.kh; p.kw = L.kw;
    p.pt = cgs_same_pad_before(L.Hb, L.kh, 1); p.pl = cgs_same_pad_before(L.Wb, L.kw, 1); p.epilogue = epilogue;
    if ((long)B * L.Hb * L.Wb * L.Cb * 4 > 0x7fffffffL) return cgs_set_error(CGS_EINVAL, "conv_smalln_f: input exceeds 2 GiB (split the batch)");
    const size_t smem = (size_t)(ST + L.kh - 1) * (ST + L.kw - 1) * SPP * sizeof(float);
    const long blocks = (long)B * (L.Hb / ST) * (L.Wb / ST);
    if (blocks == 0) return CGS_OK;
#define SNF_CASE(NN)                                                                                              \
    case NN: hipLaunchKernelGGL(conv_smalln_f_kernel<NN>, dim3((unsigned)blocks), dim3(256), smem, s, p); break;
    switch (L.Cs) {
        SNF_CASE(1) SNF_CASE(2) SNF_CASE(3) SNF_CASE(4)
        default: return cgs_set_error(CGS_EINVAL, "conv_smalln_f: N=%d", L.Cs);
    }
#undef SNF_CASE
    CGS_CHECK_LAUNCH("conv_smalln_f");
    cgs_note_kernel("conv_smalln_f_kernel");
    return CGS_OK;
}
