// Checks that the permlane16/32 swap formulation of "v + v(lane ^ 16)" and "v + v(lane ^ 32)"
// matches __shfl_xor on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline double xsum16(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ inline double xsum32(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__global__ void k(const double *in, double *o1, double *o2, double *r1, double *r2) {
    const int t = threadIdx.x;
    const double v = in[t];
    o1[t] = xsum16(v);
    o2[t] = xsum32(v);
    r1[t] = v + __shfl_xor(v, 16, 64);
    r2[t] = v + __shfl_xor(v, 32, 64);
}
int main() {
    double h[64], *d, *o;
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + i * 0.37 + (i % 7) * 1e-3;
    hipMalloc(&d, 64 * 8); hipMalloc(&o, 4 * 64 * 8);
    hipMemcpy(d, h, 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o + 64, o + 128, o + 192);
    double r[256];
    hipMemcpy(r, o, 256 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += (r[i] != r[128 + i]) + (r[64 + i] != r[192 + i]);
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
