// Where do the wavefronts of co-resident workgroups land?  Every wave of a grid of 256-thread workgroups (40 KB of LDS each:
// four per CU, the shape of chol_band_lds_kernel) records HW_REG_HW_ID (gfx9: wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13),
// HW_REG_XCC_ID and HW_REG_LDS_ALLOC; the host prints, per wave index inside the workgroup, the histogram of SIMD ids,
// and how often the four workgroups resident on one CU have their wave 0 on the SAME SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/simd_map.hip -o tools/simd_map && tools/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
    extern __shared__ double lds[];
    const int w = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
    unsigned la = __builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11));
    // keep the workgroup resident for a while so that the CU fills up
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0000001 + 0.5;
    lds[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + ((size_t)blockIdx.x * 4 + w) * 4;
        o[0] = hw; o[1] = xcc; o[2] = la; o[3] = (unsigned)(lds[(threadIdx.x + 1) & 255] != 0.0);
    }
}
int main() {
    const int grid = 4096;
    unsigned *d;
    hipMalloc(&d, (size_t)grid * 16 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 40000, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)grid * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 4; ++w) hist[w][(h[((size_t)b * 4 + w) * 4] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d of a workgroup: SIMD histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // first 1024 workgroups (the initial fill): group by (xcc, se, sh, cu) and list wave-0 SIMD + LDS base
    std::map<unsigned, std::vector<std::pair<int, unsigned>>> cu;
    for (int b = 0; b < 1024; ++b) {
        const unsigned hw = h[(size_t)b * 16], xcc = h[(size_t)b * 16 + 1] & 15, la = h[(size_t)b * 16 + 2];
        const unsigned key = (xcc << 16) | (hw & 0xff00);
        cu[key].push_back({b, ((hw >> 4) & 3) | ((la & 0xffff) << 8)});
    }
    int same = 0, tot = 0, shown = 0;
    for (auto &kv : cu) {
        ++tot;
        bool all = true;
        for (auto &p : kv.second) all = all && ((p.second & 3) == (kv.second[0].second & 3));
        same += all && kv.second.size() > 1;
        if (shown < 6) {
            printf("CU key %06x:", kv.first);
            for (auto &p : kv.second) printf("  wg %d: wave0 on SIMD %u, lds_alloc %04x", p.first, p.second & 3, p.second >> 8);
            printf("\n");
            ++shown;
        }
    }
    printf("%d CUs seen among the first 1024 workgroups; on %d of them EVERY resident workgroup has its wave 0 on the same SIMD\n", tot, same);
    return 0;
}
