// Micro-benchmark: sustained v_fma_f64 rate on gfx950 (VGPR operands, SGPR operand, dependent chain).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, const double *__restrict__ sc, int iters) {
    double a[16];
    const double x = out[threadIdx.x] , y = x * 0.5;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 16 independent chains, VGPR operands
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(a[i], x, y);
        } else if (MODE == 1) {   // 16 independent chains, one SGPR operand
            const double *s = sc + (it & 7) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = fma(a[i], s[i], y);
        } else if (MODE == 2) {   // one dependent chain
#pragma unroll
            for (int i = 0; i < 16; ++i) a[0] = fma(a[0], x, y);
        } else {                  // two dependent chains
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[0] = fma(a[0], x, y); a[1] = fma(a[1], x, y); }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef double v4d __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void kmfma(double *out, int iters) {
    v4d acc[CH];
    const double x = out[threadIdx.x] + 1.0, y = x * 0.5;
    for (int c = 0; c < CH; ++c) acc[c] = v4d{x, y, x, y};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run_mfma(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 2048;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMemset(out, 0, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kmfma<CH>, dim3(blocks), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kmfma<CH>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)CH * iters * blocks * 4;   // wave-level MFMAs
    printf("%-28s waves/SIMD %d : %8.3f ms  %7.2f TFLOP/s  %6.1f clk/MFMA/SIMD @2.4GHz\n", name, blocks_per_cu, ms,
           nm * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / (nm / (256.0 * 4)));
    hipFree(out);
}

template <int MODE>
void run(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 4096;
    double *out, *sc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&sc, sizeof(double) * 128);
    hipMemset(out, 0, sizeof(double) * blocks * 256);
    hipMemset(sc, 0, sizeof(double) * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, sc, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, sc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * iters * (double)blocks * 256;
    printf("%-28s waves/SIMD %d : %8.3f ms  %7.2f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out); hipFree(sc);
}

int main() {
    for (int bpc : {1, 2, 4}) {
        run_mfma<1>("mfma f64 16x16x4, 1 chain", bpc);
        run_mfma<4>("mfma f64 16x16x4, 4 chains", bpc);
    }
    for (int bpc : {1, 2, 4, 8}) {
        run<0>("independent, VGPR", bpc);
        run<1>("independent, SGPR operand", bpc);
        run<2>("1 dependent chain", bpc);
        run<3>("2 dependent chains", bpc);
    }
    return 0;
}
