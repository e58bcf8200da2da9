// Where does the dispatcher put the workgroups of a launch that does not fill whole rounds?  (development probe, GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/place_probe.hip -o /tmp/place_probe && /tmp/place_probe <grid> <lds_kb> <spin_us>
// Each 256-thread workgroup stamps (XCC_ID, HW_ID, start, end) and spins ~spin_us on a dependent FMA chain; lds_kb of dynamic LDS
// bound the workgroups a CU holds (160 KB per CU).  Prints: workgroups per CU (histogram), CUs used, and the launch's span.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int spin_iters) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < spin_iters; ++i) a = fmaf(a, 1.0001f, 1e-7f);
    lds[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* o = out + (size_t)blockIdx.x * 4;
        o[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o[1] = t0; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = (unsigned long long)(lds[17] != 12345.f);
    }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 784;
    const int lds_kb = argc > 2 ? atoi(argv[2]) : 37;
    const int spin_us = argc > 3 ? atoi(argv[3]) : 50;
    unsigned long long* d;
    hipMalloc(&d, (size_t)grid * 32);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    const int iters = spin_us * 2400 / 5;          // ~4-5 cycles per dependent fma at 2.4 GHz
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds_kb * 1024, 0, d, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> per_cu;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < grid; ++b) {
        const unsigned long long id = h[b * 4], hw = id & 0xffffffffull, xcc = (id >> 32) & 0xf;
        const unsigned long long cu = (xcc << 16) | ((hw >> 8) & 0xff);        // cu_id[11:8], sh_id[12], se_id[15:13]
        per_cu[cu].push_back(b);
        tmin = std::min(tmin, h[b * 4 + 1]); tmax = std::max(tmax, h[b * 4 + 2]);
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[(int)kv.second.size()]++;
    printf("grid %d, %d KB LDS per workgroup, spin %d us: %zu CUs used; span %.1f us (100 MHz ticks)\n", grid, lds_kb, spin_us, per_cu.size(), (tmax - tmin) / 100.0);
    for (auto& kv : hist) printf("   %3d CUs hold %d workgroups\n", kv.second, kv.first);
    // concurrency on the fullest CU: did its workgroups overlap in time?
    size_t best = 0; unsigned long long bk = 0;
    for (auto& kv : per_cu) if (kv.second.size() > best) { best = kv.second.size(); bk = kv.first; }
    printf("   fullest CU %llx:", bk);
    for (int b : per_cu[bk]) printf(" [wg %d: %.1f..%.1f us]", b, (h[b * 4 + 1] - tmin) / 100.0, (h[b * 4 + 2] - tmin) / 100.0);
    printf("\n   first 24 workgroups -> (xcc, cu key):");
    for (int b = 0; b < 24 && b < grid; ++b) printf(" %llu:%02llx", (h[b * 4] >> 32) & 0xf, (h[b * 4] >> 8) & 0xff);
    printf("\n");
    return 0;
}
