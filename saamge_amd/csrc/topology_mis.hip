// Minimal intersection sets on the device (agg_construct_mises_local, amg/src/aggregates.cpp:501-653, and
// the tables derived from them, :776-777): the same tables, bit for bit, as the host build in
// topology.hip -- that one stays for the small coarse levels, for aggregates with arbitration and as the
// fallback; this one takes ~2 ms instead of ~170 ms of host time at 17 M dofs, which had become the longest
// thing beside the level-0 eigenproblems and was replicated on every rank.
//
// A MIS is the set of dofs with the same AE list.  Numbering = order of first appearance scanning the dofs
// upwards, i.e. the rank of the group's smallest dof among all group representatives; dofs inside a MIS
// ascending.  On the device:
//   1. representative of every dof: single-AE dofs by an atomicMin per AE; the others through a hash table
//      keyed by a 64-bit hash of the (ascending) AE list holding the smallest dof of the group.  The hash is
//      not trusted: every dof then compares its whole AE list with its representative's, and any mismatch
//      (a 64-bit collision) sends the level back to the host path.
//   2. MIS ids = exclusive scan over the representative flags, gathered through the representative.
//   3. mis_to_dof = stable radix sort of the dofs by MIS id (hipcub): ascending dofs inside a MIS.
//   4. mis_to_AE = the representative's AE list; AE_to_mis = stable sort of the (MIS, AE) pairs by AE.
//   5. pair_loc: AE-local index of every MIS dof for every AE of the MIS.
#include <hipcub/hipcub.hpp>

#include "topology.h"

namespace saamge_amd {


namespace {

constexpr unsigned long long EMPTY_KEY = ~0ull;

__device__ inline unsigned long long hash_ae_row(const int *row, int rs) {
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)rs;
    for (int t = 0; t < rs; ++t) {
        h ^= (unsigned long long)(unsigned)row[t] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 1099511628211ull;
    }
    h ^= h >> 29;
    return h == EMPTY_KEY ? 0ull : h;
}

__global__ __launch_bounds__(256) void mis_fill_kernel(long n, int v, int *p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void mis_fill64_kernel(long n, unsigned long long v, unsigned long long *p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void mis_iota_kernel(long n, int *p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (int)i;
}

__global__ __launch_bounds__(256) void mis_multi_flag_kernel(int ND, const int *__restrict__ I, int *__restrict__ flag) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < ND) flag[i] = (I[i + 1] - I[i] > 1) ? 1 : 0;
}

__global__ __launch_bounds__(256) void mis_insert_kernel(int ND, const int *__restrict__ I, const int *__restrict__ J,
                                                         int *__restrict__ single_min, unsigned long long *__restrict__ keys,
                                                         int *__restrict__ minv, unsigned mask, int *__restrict__ hslot) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ND) return;
    const int rs = I[i + 1] - I[i];
    if (rs == 1) {
        atomicMin(&single_min[J[I[i]]], (int)i);
        return;
    }
    if (rs == 0) { hslot[i] = -1; return; }
    const unsigned long long h = hash_ae_row(J + I[i], rs);
    unsigned slot = (unsigned)(h >> 7) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, h);
        if (prev == EMPTY_KEY || prev == h) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&minv[slot], (int)i);
    hslot[i] = (int)slot;
}

__global__ __launch_bounds__(256) void mis_rep_kernel(int ND, const int *__restrict__ I, const int *__restrict__ J,
                                                      const int *__restrict__ single_min, const int *__restrict__ minv,
                                                      const int *__restrict__ hslot, int *__restrict__ rep_of,
                                                      int *__restrict__ is_rep, int *__restrict__ err) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ND) return;
    const int rs = I[i + 1] - I[i];
    int rep = (int)i;
    if (rs == 1) rep = single_min[J[I[i]]];
    else if (rs > 1) rep = minv[hslot[i]];
    else atomicExch(err, 2);                     // a dof in no agglomerate: not a valid partition
    if (rs > 1 && rep != (int)i) {               // the hash is not trusted
        bool same = (I[rep + 1] - I[rep]) == rs;
        for (int t = 0; same && t < rs; ++t) same = J[I[rep] + t] == J[I[i] + t];
        if (!same) atomicExch(err, 1);
    }
    rep_of[i] = rep;
    is_rep[i] = (rep == (int)i) ? 1 : 0;
}

__global__ __launch_bounds__(256) void mis_ids_kernel(int ND, const int *__restrict__ rep_of, const int *__restrict__ rank,
                                                      const int *__restrict__ is_rep, int *__restrict__ mises,
                                                      int *__restrict__ reps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ND) return;
    mises[i] = rank[rep_of[i]];
    if (is_rep[i]) reps[rank[i]] = (int)i;
}

__global__ __launch_bounds__(256) void mis_hist_kernel(long n, const int *__restrict__ key, int *__restrict__ cnt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[key[i]], 1);
}

__global__ __launch_bounds__(256) void mis_rowpos_kernel(int ND, const int *__restrict__ mis2d_I, const int *__restrict__ mis2d_J,
                                                         const int *__restrict__ mises, int *__restrict__ row_in_mis) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= ND) return;
    const int dof = mis2d_J[k];
    row_in_mis[dof] = (int)k - mis2d_I[mises[dof]];
}

__global__ __launch_bounds__(256) void mis_ae_count_kernel(int nm, const int *__restrict__ reps, const int *__restrict__ I,
                                                           int *__restrict__ cnt) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m < nm) cnt[m] = I[reps[m] + 1] - I[reps[m]];
}
__global__ __launch_bounds__(256) void mis_ae_fill_kernel(int nm, const int *__restrict__ reps, const int *__restrict__ I,
                                                          const int *__restrict__ J, const int *__restrict__ mis2ae_I,
                                                          int *__restrict__ mis2ae_J, int *__restrict__ mis_of_pair) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= nm) return;
    const int r = reps[m], rs = I[r + 1] - I[r], o = mis2ae_I[m];
    for (int t = 0; t < rs; ++t) {
        mis2ae_J[o + t] = J[I[r] + t];
        mis_of_pair[o + t] = (int)m;
    }
}
__global__ __launch_bounds__(256) void mis_ae2mis_kernel(long npairs, const int *__restrict__ ae_pair,
                                                         const int *__restrict__ mis_of_pair, int *__restrict__ ae2mis_J) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < npairs) ae2mis_J[k] = mis_of_pair[ae_pair[k]];
}
__global__ __launch_bounds__(256) void mis_pair_size_kernel(long npairs, const int *__restrict__ mis_of_pair,
                                                            const int *__restrict__ mis2d_I, int *__restrict__ sz) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q < npairs) { const int m = mis_of_pair[q]; sz[q] = mis2d_I[m + 1] - mis2d_I[m]; }
}
__global__ __launch_bounds__(256) void mis_widen_kernel(long n, const int *__restrict__ in, int64_t *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void mis_pair_loc_kernel(int ND, const int *__restrict__ I, const int *__restrict__ dof_id_inAE,
                                                           const int *__restrict__ mises, const int *__restrict__ row_in_mis,
                                                           const int *__restrict__ mis2ae_I, const int64_t *__restrict__ pair_loc_off,
                                                           int *__restrict__ pair_loc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= ND) return;
    const int m = mises[i], k = row_in_mis[i], q0 = mis2ae_I[m];
    // every dof of the MIS has the same AE list: the t-th AE of the dof's row is the t-th AE of the MIS
    for (int t = 0; t < I[i + 1] - I[i]; ++t) pair_loc[pair_loc_off[q0 + t] + k] = dof_id_inAE[I[i] + t];
}

int bits_for(int n) {
    int b = 1;
    while ((1ll << b) < n) ++b;
    return b;
}

// stable sort of the pairs (key[i], i) by key; out_vals = the permutation
void stable_sort_by_key(hipStream_t s, int n, const int *key, int nkeys, int *out_vals) {
    if (n == 0) return;
    DBuf<int> iota((size_t)n), keys_out((size_t)n);
    hipLaunchKernelGGL(mis_iota_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, (long)n, iota.p);
    size_t tmp_bytes = 0;
    SA_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, keys_out.p, iota.p, out_vals, n, 0, bits_for(nkeys), s));
    DBuf<char> tmp(tmp_bytes + 16);
    SA_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs((void *)tmp.p, tmp_bytes, key, keys_out.p, iota.p, out_vals, n, 0, bits_for(nkeys), s));
    SA_HIP_CHECK(hipStreamSynchronize(s));      // the temporaries go out of scope
}

template <class T>
void fetch(hvec<T> &dst, const DBuf<T> &src, size_t n, hipStream_t s) {
    dst.resize(n);
    if (n) SA_HIP_CHECK(hipMemcpyAsync(dst.data(), src.p, sizeof(T) * n, hipMemcpyDeviceToHost, s));
}

}  // namespace

// Needs d.d2ae_I / d2ae_J / dof_id_inAE on the device.  Fills the MIS half of `d` and the host tables of `r`
// that host code reads (mises, mis_to_dof, mis_to_AE, AE_to_mis).  Returns false (nothing kept) when the
// level must take the host path.
bool build_relations_mis_device(Relations &r, DevRelations &d, hipStream_t s) {
    const int ND = r.ND, nparts = r.nparts;
    if (ND == 0 || !d.d2ae_I.p || !d.d2ae_J.p || !d.dof_id_inAE.p) return false;
    const int g = div_up(ND, 256);
    const int *I = d.d2ae_I.p, *J = d.d2ae_J.p;
    // ---- 1. representatives ----
    DBuf<int> flag((size_t)ND + 1), pos((size_t)ND + 2);
    hipLaunchKernelGGL(mis_multi_flag_kernel, dim3(g), dim3(256), 0, s, ND, I, flag.p);
    exclusive_scan_int(s, ND, flag.p, pos.p);
    int n_multi = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&n_multi, pos.p + ND, sizeof(int), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    unsigned cap = 1024;
    while (cap < 2u * (unsigned)std::max(n_multi, 1)) cap <<= 1;
    DBuf<unsigned long long> keys((size_t)cap);
    DBuf<int> minv((size_t)cap), single_min((size_t)nparts), hslot((size_t)ND), rep_of((size_t)ND), err(1);
    hipLaunchKernelGGL(mis_fill64_kernel, dim3(div_up(cap, 256)), dim3(256), 0, s, (long)cap, EMPTY_KEY, keys.p);
    hipLaunchKernelGGL(mis_fill_kernel, dim3(div_up(cap, 256)), dim3(256), 0, s, (long)cap, 0x7fffffff, minv.p);
    hipLaunchKernelGGL(mis_fill_kernel, dim3(div_up(nparts, 256)), dim3(256), 0, s, (long)nparts, 0x7fffffff, single_min.p);
    err.zero(s);
    hipLaunchKernelGGL(mis_insert_kernel, dim3(g), dim3(256), 0, s, ND, I, J, single_min.p, keys.p, minv.p, cap - 1, hslot.p);
    hipLaunchKernelGGL(mis_rep_kernel, dim3(g), dim3(256), 0, s, ND, I, J, single_min.p, minv.p, hslot.p, rep_of.p, flag.p, err.p);
    SA_HIP_CHECK(hipGetLastError());
    // ---- 2. MIS ids ----
    exclusive_scan_int(s, ND, flag.p, pos.p);
    int nm = 0, herr = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&nm, pos.p + ND, sizeof(int), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipMemcpyAsync(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    if (herr) return false;      // hash collision (1) or a dof outside every agglomerate (2): host path decides
    d.mises.alloc((size_t)ND);
    DBuf<int> reps((size_t)nm);
    hipLaunchKernelGGL(mis_ids_kernel, dim3(g), dim3(256), 0, s, ND, rep_of.p, pos.p, flag.p, d.mises.p, reps.p);
    // ---- 3. mis_to_dof ----
    d.mis2d_I.alloc((size_t)nm + 1);
    d.mis2d_J.alloc((size_t)ND);
    {
        DBuf<int> cnt((size_t)nm + 1);
        cnt.zero(s);
        hipLaunchKernelGGL(mis_hist_kernel, dim3(g), dim3(256), 0, s, (long)ND, d.mises.p, cnt.p);
        exclusive_scan_int(s, nm, cnt.p, d.mis2d_I.p);
    }
    stable_sort_by_key(s, ND, d.mises.p, nm, d.mis2d_J.p);
    d.dof_row_in_mis.alloc((size_t)ND);
    hipLaunchKernelGGL(mis_rowpos_kernel, dim3(g), dim3(256), 0, s, ND, d.mis2d_I.p, d.mis2d_J.p, d.mises.p, d.dof_row_in_mis.p);
    // ---- 4. mis_to_AE, AE_to_mis ----
    d.mis2ae_I.alloc((size_t)nm + 1);
    {
        DBuf<int> cnt((size_t)nm + 1);
        hipLaunchKernelGGL(mis_ae_count_kernel, dim3(div_up(nm, 256)), dim3(256), 0, s, nm, reps.p, I, cnt.p);
        exclusive_scan_int(s, nm, cnt.p, d.mis2ae_I.p);
    }
    int npairs = 0;
    SA_HIP_CHECK(hipMemcpyAsync(&npairs, d.mis2ae_I.p + nm, sizeof(int), hipMemcpyDeviceToHost, s));
    SA_HIP_CHECK(hipStreamSynchronize(s));
    d.mis2ae_J.alloc((size_t)npairs);
    DBuf<int> mis_of_pair((size_t)npairs);
    hipLaunchKernelGGL(mis_ae_fill_kernel, dim3(div_up(nm, 256)), dim3(256), 0, s, nm, reps.p, I, J, d.mis2ae_I.p, d.mis2ae_J.p,
                       mis_of_pair.p);
    d.ae2mis_I.alloc((size_t)nparts + 1);
    d.ae2mis_J.alloc((size_t)npairs);
    d.ae_pair.alloc((size_t)npairs);
    {
        DBuf<int> cnt((size_t)nparts + 1);
        cnt.zero(s);
        hipLaunchKernelGGL(mis_hist_kernel, dim3(div_up(npairs, 256)), dim3(256), 0, s, (long)npairs, d.mis2ae_J.p, cnt.p);
        exclusive_scan_int(s, nparts, cnt.p, d.ae2mis_I.p);
    }
    stable_sort_by_key(s, npairs, d.mis2ae_J.p, nparts, d.ae_pair.p);
    hipLaunchKernelGGL(mis_ae2mis_kernel, dim3(div_up(npairs, 256)), dim3(256), 0, s, (long)npairs, d.ae_pair.p, mis_of_pair.p, d.ae2mis_J.p);
    // ---- 5. pair_loc ----
    d.pair_loc_off.alloc((size_t)npairs + 1);
    int total = 0;
    {
        DBuf<int> sz((size_t)npairs + 1), off((size_t)npairs + 2);
        hipLaunchKernelGGL(mis_pair_size_kernel, dim3(div_up(npairs, 256)), dim3(256), 0, s, (long)npairs, mis_of_pair.p, d.mis2d_I.p, sz.p);
        exclusive_scan_int(s, npairs, sz.p, off.p);      // (the total is the number of (dof, AE) incidences: fits 32 bits with d2ae_J)
        SA_HIP_CHECK(hipMemcpyAsync(&total, off.p + npairs, sizeof(int), hipMemcpyDeviceToHost, s));
        hipLaunchKernelGGL(mis_widen_kernel, dim3(div_up(npairs + 1, 256)), dim3(256), 0, s, (long)npairs + 1, off.p, d.pair_loc_off.p);
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    d.pair_loc.alloc((size_t)total);
    hipLaunchKernelGGL(mis_pair_loc_kernel, dim3(g), dim3(256), 0, s, ND, I, d.dof_id_inAE.p, d.mises.p, d.dof_row_in_mis.p,
                       d.mis2ae_I.p, d.pair_loc_off.p, d.pair_loc.p);
    SA_HIP_CHECK(hipGetLastError());
    // ---- host copies of what host code reads ----
    r.num_mises = nm;
    fetch(r.mises, d.mises, (size_t)ND, s);
    r.mis_to_dof.ncols = ND;
    fetch(r.mis_to_dof.I, d.mis2d_I, (size_t)nm + 1, s);
    fetch(r.mis_to_dof.J, d.mis2d_J, (size_t)ND, s);
    r.mis_to_AE.ncols = nparts;
    fetch(r.mis_to_AE.I, d.mis2ae_I, (size_t)nm + 1, s);
    fetch(r.mis_to_AE.J, d.mis2ae_J, (size_t)npairs, s);
    r.AE_to_mis.ncols = nm;
    fetch(r.AE_to_mis.I, d.ae2mis_I, (size_t)nparts + 1, s);
    fetch(r.AE_to_mis.J, d.ae2mis_J, (size_t)npairs, s);
    SA_HIP_CHECK(hipStreamSynchronize(s));
    return true;
}

}  // namespace saamge_amd
