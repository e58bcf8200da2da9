// Does an out-of-range lane of a buffer LDS-DMA load (buffer_load_dwordx4 ... lds) write ZEROS to its LDS slot, or leave it alone?
// And is the LDS slot lane-linear (M0 base + lane * 16)?    hipcc -O3 --offload-arch=gfx950 dma_oob.hip -o dma_oob ; ./dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) smem[i] = 7.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // lanes 8..15 and 40.. out of range; soffset = 64 bytes
    const unsigned voff = (lane >= 8 && lane < 16) || lane >= 40 ? 0xFFFFFFF0u : (unsigned)(lane * 32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 512), 16, voff, 64, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = smem[i];
}
int main() {
    float *in, *out; const int n = 4096;
    hipMalloc(&in, n * 4); hipMalloc(&out, 1024 * 4);
    float h[4096]; for (int i = 0; i < n; ++i) h[i] = (float)i;
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    k<<<1, 128, 8192>>>(in, out, n);
    float o[1024]; hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("wave %d:", w);
        for (int l = 0; l < 64; ++l) printf(" [%d]%g,%g,%g,%g", l, o[w * 512 + l * 4], o[w * 512 + l * 4 + 1], o[w * 512 + l * 4 + 2], o[w * 512 + l * 4 + 3]);
        printf("\n   behind: %g %g\n", o[w * 512 + 256], o[w * 512 + 257]);
    }
    return 0;
}
