// Laboratory for the SELL SpMV / smoother kernels on the headline's fine-level operator shape (27-point stencil on
// n^3 nodes, essential boundary eliminated): times the library's kernels in controlled sequences and a set of
// experimental variants, to find what bounds them.  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Isaamge_amd/csrc -Iinclude tools/spmv_lab.hip -Lsaamge_amd -lsaamge_amd \
//         -Wl,-rpath,'$ORIGIN/../saamge_amd' -o tools/spmv_lab ;  tools/spmv_lab [n=257]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sparse.h"
using namespace saamge_amd;

__global__ void count_kernel(int n, int *cnt) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= (long)n * n * n) return;
    const int i = r % n, j = (r / n) % n, k = r / ((long)n * n);
    auto span = [&](int c) { return (c > 0) + 1 + (c < n - 1); };
    cnt[r] = span(i) * span(j) * span(k);
}
__global__ void fill_kernel(int n, const roff_t *rowptr, int *col, double *val) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= (long)n * n * n) return;
    const int i = r % n, j = (r / n) % n, k = r / ((long)n * n);
    auto bnd = [&](int a, int b, int c) { return a == 0 || a == n - 1 || b == 0 || b == n - 1 || c == 0 || c == n - 1; };
    roff_t p = rowptr[r];
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int a = i + dx, b = j + dy, c = k + dz;
                if (a < 0 || a >= n || b < 0 || b >= n || c < 0 || c >= n) continue;
                const int m = abs(dx) + abs(dy) + abs(dz);
                double v = m == 0 ? 8.0 / 3.0 : (m == 1 ? 0.0 : (m == 2 ? -1.0 / 6.0 : -1.0 / 12.0));
                if (m && (bnd(i, j, k) || bnd(a, b, c))) v = 0.0;
                col[p] = (int)(((long)c * n + b) * n + a);
                val[p++] = v;
            }
}
__global__ void init_kernel(long n, double *x, double s) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = s * (1.0 + (double)(i % 1000) * 1e-3);
}
// pure streaming of the smoother's byte mix: code words + 3 vectors in, 1 vector out
__global__ __launch_bounds__(256) void stream_mix_kernel(long n, const unsigned *__restrict__ codes, const double *__restrict__ a,
                                                         const double *__restrict__ b, const double *__restrict__ c,
                                                         double *__restrict__ y, int words_per_row) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned acc = 0;
    const long slice = i >> 6;
    const int lane = (int)(i & 63);
    for (int q = 0; q < words_per_row; ++q) acc += __builtin_nontemporal_load(codes + (slice * words_per_row + q) * 64 + lane);
    y[i] = a[i] + b[i] * c[i] + (double)(acc & 1u);
}

// constructive experiment: start from the stream and add the pieces of the SpMV kernel one by one
struct alignas(16) LabEntry { int off, pad; double val; };
enum { L_SCALAR = 1, L_TABLE = 2, L_FMA = 4, L_GATHER = 8, L_XCD = 16, L_NOCODES = 32, L_LDSX = 64 };
template <int F>
__global__ __launch_bounds__(256) void lab_kernel(int nrows, int nblocks, int per_xcd, const roff_t *__restrict__ sptr,
                                                  const int *__restrict__ ntab, const int *__restrict__ tab,
                                                  const double *__restrict__ vtab, const unsigned *__restrict__ codes,
                                                  const double *__restrict__ x, const double *__restrict__ b,
                                                  const double *__restrict__ dinv, double *__restrict__ y) {
    __shared__ LabEntry ltab[4][64];
    __shared__ double lx[4][64 + 8];
    const int blk = (F & L_XCD) ? (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (blk >= nblocks) return;
    const long row = (long)blk * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane((int)(row >> 6));
    if ((long)slice * 64 >= nrows) return;
    roff_t beg;
    int w, np;
    if (F & L_SCALAR) {
        beg = sptr[slice];
        w = (int)((sptr[slice + 1] - beg) >> 6);
        np = ntab[slice] - 256;
        if (np < 0 || w > 27) { w = 27; np = 27; }
    } else {
        beg = (roff_t)slice * 26 * 64;      // (close to the real offsets and inside the allocation: boundary slices are narrower)
        w = 27;
        np = 27;
    }
    const unsigned *wp = codes + ((size_t)(beg >> 2) + (size_t)slice * 64 + lane);
    unsigned cws[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) cws[q] = (F & L_NOCODES) ? (unsigned)(q * 0x01010101u) : __builtin_nontemporal_load(wp + 64 * q);
    const bool live = row < nrows;
    LabEntry *lt = ltab[threadIdx.x >> 6];
    if (F & L_TABLE) {
        const int mytab = (lane < np) ? tab[(size_t)slice * 64 + lane] : 0;
        const double myval = (lane < np) ? vtab[(size_t)slice * 64 + lane] : 0.0;
        lt[lane] = LabEntry{mytab, 0, myval};
    }
    const double e_b = live ? b[row] : 0.0, e_d = live ? dinv[row] : 0.0, e_x = live ? x[row] : 0.0;
    if (F & L_LDSX) lx[threadIdx.x >> 6][lane + 4] = e_x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double s0 = 0.0, s1 = 0.0;
    const char *xb = (const char *)x;
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        const int idx = (int)((cws[j >> 2] >> (8 * (j & 3))) & 255u);
        int off = 0;
        double v = 1.0 + idx;
        if (F & L_TABLE) { const LabEntry e = lt[idx & 63]; off = e.off; v = e.val; }
        double xv = e_x;
        if (F & L_GATHER) {
            long c = row + off;
            if (c < 0 || c >= nrows) c = live ? row : 0;
            xv = *(const double *)(xb + ((unsigned)c << 3));
        } else if (F & L_LDSX) {
            xv = lx[threadIdx.x >> 6][lane + 4 + (off % 3)];
        }
        if (F & L_FMA) { if (j & 1) s1 = fma(v, xv, s1); else s0 = fma(v, xv, s0); }
        else s0 += (j == 13 ? v * xv : 0.0) + (double)off;
    }
    if (live) y[row] = e_x + 0.7 * (e_d * (s0 + s1 - e_b));
}

template <class F>
static double time_us(hipStream_t s, int reps, F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f(0);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) f(r + 1);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 257;
    const long N = (long)n * n * n;
    hipStream_t s;
    hipStreamCreate(&s);
    set_thread_stream(s);
    DCsr A;
    A.nrows = A.ncols = (int)N;
    DBuf<int> cnt((size_t)N);
    const int grid = (int)((N + 255) / 256);
    hipLaunchKernelGGL(count_kernel, dim3(grid), dim3(256), 0, s, n, cnt.p);
    A.rowptr.alloc((size_t)N + 1);
    exclusive_scan_off(s, (int)N, cnt.p, A.rowptr.p);
    roff_t nnz = 0;
    hipMemcpyAsync(&nnz, A.rowptr.p + N, sizeof(roff_t), hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    A.nnz = nnz;
    A.col.alloc((size_t)nnz);
    A.val.alloc((size_t)nnz);
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, s, n, A.rowptr.p, A.col.p, A.val.p);
    build_sell(s, A);
    hipStreamSynchronize(s);
    printf("n=%d rows=%ld nnz=%lld slices pair/offset/plain %lld/%lld/%lld stream bytes %.3f GB\n", n, N, (long long)nnz,
           (long long)A.sell_class_slices[0], (long long)A.sell_class_slices[1], (long long)A.sell_class_slices[2],
           A.sell_stream_bytes * 1e-9);
    DBuf<double> b((size_t)N), dinv((size_t)N), x0((size_t)N), x1((size_t)N), x2((size_t)N);
    hipLaunchKernelGGL(init_kernel, dim3(grid), dim3(256), 0, s, N, b.p, 1.0);
    hipLaunchKernelGGL(init_kernel, dim3(grid), dim3(256), 0, s, N, dinv.p, -0.1);
    hipLaunchKernelGGL(init_kernel, dim3(grid), dim3(256), 0, s, N, x0.p, 0.5);
    hipLaunchKernelGGL(init_kernel, dim3(grid), dim3(256), 0, s, N, x1.p, 0.25);
    const int reps = 20;
    const double fmt = A.sell_stream_bytes;
    auto report = [&](const char *name, double us, double bytes) {
        printf("%-58s %8.1f us  %6.2f TB/s of %.3f GB\n", name, us, bytes / us * 1e-6, bytes * 1e-9);
        fflush(stdout);
    };
    double us;
    us = time_us(s, reps, [&](int r) { (r & 1) ? smooth_step(s, A, dinv.p, b.p, x1.p, x0.p, 0.7) : smooth_step(s, A, dinv.p, b.p, x0.p, x1.p, 0.7); });
    report("smooth_step ping-pong x0<->x1 (production order)", us, fmt + 32.0 * N);
    us = time_us(s, reps, [&](int) { smooth_step(s, A, dinv.p, b.p, x0.p, x1.p, 0.7); });
    report("smooth_step x0->x1 repeated", us, fmt + 32.0 * N);
    us = time_us(s, reps, [&](int) { spmv(s, A, x0.p, x1.p); });
    report("spmv x0->x1 repeated", us, fmt + 16.0 * N);
    us = time_us(s, reps, [&](int r) { (r & 1) ? spmv(s, A, x1.p, x0.p) : spmv(s, A, x0.p, x1.p); });
    report("spmv ping-pong", us, fmt + 16.0 * N);
    us = time_us(s, reps, [&](int) { spmv_residual(s, A, x0.p, b.p, x1.p); });
    report("spmv_residual x0->x1 repeated", us, fmt + 24.0 * N);
    us = time_us(s, reps, [&](int) { smooth_step(s, A, b.p, b.p, x0.p, x1.p, 0.7); });
    report("smooth_step with dinv == b (one stream fewer)", us, fmt + 24.0 * N);
    us = time_us(s, reps, [&](int) { smooth_step(s, A, x0.p, x0.p, x0.p, x1.p, 0.7); });
    report("smooth_step with dinv == b == x (three streams fewer)", us, fmt + 16.0 * N);
    us = time_us(s, reps, [&](int) { smooth_step(s, A, dinv.p, b.p, x0.p, x2.p, 0.7); spmv(s, A, x0.p, x1.p); });
    report("smooth_step + spmv pair (sum)", us, 2 * fmt + 48.0 * N);
    const int wpr = 7;
    us = time_us(s, reps, [&](int) { hipLaunchKernelGGL(stream_mix_kernel, dim3(grid), dim3(256), 0, s, N, A.sell_code.p, x0.p, b.p, dinv.p, x1.p, wpr); });
    report("stream_mix: 7 code words + 3 vectors in, 1 out (coalesced)", us, (4.0 * wpr + 32.0) * N);
    us = time_us(s, reps, [&](int) { hipLaunchKernelGGL(stream_mix_kernel, dim3(grid), dim3(256), 0, s, N, A.sell_code.p, x0.p, x0.p, x0.p, x1.p, wpr); });
    report("stream_mix with one vector in", us, (4.0 * wpr + 16.0) * N);
    us = time_us(s, reps, [&](int) { hipLaunchKernelGGL(stream_mix_kernel, dim3(grid), dim3(256), 0, s, N, A.sell_code.p, x0.p, b.p, dinv.p, x1.p, 0); });
    report("stream_mix without codes (3 vectors in, 1 out)", us, 32.0 * N);
    {
        const int nblocks = (int)((N + 255) / 256), per_xcd = (nblocks + 7) / 8;
#define LAB(FLAGS, NAME)                                                                                                      \
        us = time_us(s, reps, [&](int) {                                                                                      \
            hipLaunchKernelGGL((lab_kernel<FLAGS>), dim3(((FLAGS) & L_XCD) ? per_xcd * 8 : nblocks), dim3(256), 0, s, (int)N, nblocks, \
                               per_xcd, A.sell_ptr.p, A.sell_ntab.p, A.sell_tab.p, A.sell_vtab.p, A.sell_code.p, x0.p, b.p, dinv.p, x1.p); \
        });                                                                                                                   \
        report(NAME, us, fmt + 32.0 * N);
        LAB(0, "lab: codes + b, dinv, x in, y out (no table, no fma)")
        LAB(L_NOCODES, "lab: the same without the code loads")
        LAB(L_SCALAR, "lab: + scalar slice descriptors")
        LAB(L_SCALAR | L_TABLE, "lab: + slice table -> LDS, 27 table reads")
        LAB(L_SCALAR | L_TABLE | L_FMA, "lab: + 27 fma")
        LAB(L_SCALAR | L_TABLE | L_FMA | L_LDSX, "lab: + x from LDS (27 ds_read_b64)")
        LAB(L_SCALAR | L_TABLE | L_FMA | L_GATHER, "lab: + 27 global gathers instead")
        LAB(L_SCALAR | L_TABLE | L_FMA | L_GATHER | L_XCD, "lab: + XCD-contiguous blocks")
        LAB(L_FMA | L_GATHER | L_XCD | L_TABLE, "lab: gathers, table, no scalar descriptors")
    }
    return 0;
}
