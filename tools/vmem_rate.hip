// Microbenchmark: cycles per vector-memory wave-instruction per CU by access width and alignment (data L2/L1 resident).
// Each wave issues ITER x 8 independent loads of width W bytes per lane, lanes contiguous (64 W bytes per instruction),
// base address per instruction = slot * stride (+ misalign bytes); 1024 workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W, bool GATHER>
__global__ __launch_bounds__(256) void k(const char *__restrict__ buf, double *__restrict__ out, int iters, int mis, int span) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const char *base = buf + (size_t)(wave % span) * 4096 + mis + lane * W;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char *p = base + ((it * 8 + u) % 64) * (64 * W);
            if (W == 4) acc += (double)*(const int *)p;
            if (W == 8) acc += *(const double *)p;
            if (W == 16) { const double2 v = *(const double2 *)p; acc += v.x + v.y; }
        }
    }
    if (acc == 1.2345) out[0] = acc;
}
int main() {
    char *buf; double *out;
    const size_t bytes = (size_t)64 << 20;
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 64, grid = 2048;
    auto run = [&](const char *name, auto kern, int W, int mis, int span) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, out, iters, mis, span);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, out, iters, mis, span);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double instr_per_cu = (double)grid * 4 * iters * 8 / 256.0;
        const double cyc = ms * 1e-3 * 2.4e9 / instr_per_cu;
        printf("%-28s W=%2d misalign=%3d span=%5d: %7.1f us  %5.1f cycles per wave-instruction per CU, %6.2f TB/s L1-side\n", name, W, mis, span,
               ms * 1e3, cyc, (double)grid * 4 * iters * 8 * 64 * W / (ms * 1e-3) * 1e-12);
    };
    for (int span : {64, 8192}) {
        run("dword", k<4, false>, 4, 0, span);
        run("dwordx2", k<8, false>, 8, 0, span);
        run("dwordx2 misaligned", k<8, false>, 8, 72, span);
        run("dwordx4", k<16, false>, 16, 0, span);
        run("dwordx4 misaligned(8)", k<16, false>, 16, 72, span);
    }
    return 0;
}
