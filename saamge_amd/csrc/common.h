// Shared host-side utilities for the saamge_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <vector>

namespace saamge_amd {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define SA_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            throw ::saamge_amd::Error(2, std::string(__FILE__) + ":" +                  \
                                             std::to_string(__LINE__) + " " + #expr +   \
                                             " -> " + hipGetErrorString(e_));           \
    } while (0)

// The reference aborts through SA_ASSERT (amg/inc/common.hpp:635-647); we throw and
// translate to an error code at the C ABI.
#define SA_REQUIRE(cond, msg)                                                           \
    do {                                                                                \
        if (!(cond))                                                                    \
            throw ::saamge_amd::Error(1, std::string(__FILE__) + ":" +                  \
                                             std::to_string(__LINE__) + " " + (msg));   \
    } while (0)

// ---- device affinity of helper threads -------------------------------------------------
// HIP's current device is a per-host-thread property and defaults to device 0.  One process
// drives one GPU (rank r on device LOCAL_RANK), so every helper thread that may touch HIP
// (hipMalloc, copies, pinned allocations, launches) adopts the device of the thread that
// spawned it, and the helper streams are kept per device.
inline int current_device() {
    int d = 0;
    SA_HIP_CHECK(hipGetDevice(&d));
    return d;
}
inline void adopt_device(int dev) { SA_HIP_CHECK(hipSetDevice(dev)); }
// non-blocking helper stream number `slot` of the calling thread's current device
hipStream_t side_stream(int slot);

// ---- pinned host memory: pooled allocator for the big host-side tables -----------------
// Pageable <-> device copies run at a few GB/s; page-locked ones at PCIe speed.  hipHostMalloc
// itself is slow (it pins pages), so freed blocks are kept in a size-class pool and reused
// by the next level / hierarchy.
void *pinned_alloc(size_t bytes);
void pinned_free(void *p, size_t bytes);
void pinned_pool_release();

template <class T>
struct PinnedAlloc {
    typedef T value_type;
    PinnedAlloc() {}
    template <class U>
    PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n) { return (T *)pinned_alloc(n * sizeof(T)); }
    void deallocate(T *p, size_t n) { pinned_free((void *)p, n * sizeof(T)); }
    template <class U>
    bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <class U>
    bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
template <class T>
using hvec = std::vector<T, PinnedAlloc<T>>;

inline bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear: plain host memory reports an error
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ---- options of the library (saamge_amd_options, include/saamge_amd.h): process-wide, set through the C ABI ----------
// What is left of the ~50 environment switches of rounds 1-3: the variants that were measured without gain are gone, the
// ones tests need to reach a code path (or a caller may want) are fields here.  Environment variables that remain:
// SAAMGE_AMD_TIMING, SAAMGE_AMD_SERIAL (diagnostics), SAAMGE_AMD_POOL_MAX_GB, SAAMGE_AMD_THREADS (resources).
struct Options {
    int eig_strict = 0;               // few-eigenpairs path: a fallback to the dense path is an error (tests of that path)
    int eig_certify = 1;              // the count #{lambda < theta} certified by the inertia of C - theta I
    int eig_min_n = 64;               // smallest agglomerate of a batch that takes the few-eigenpairs path
    int eig_force_fallback = 0;       // tests: every k-th matrix takes the per-matrix dense fallback
    int eig_dense_only = 0;           // saamge_amd_lower_eigens_batched: the dense path (a hierarchy: saamge_amd_params.eigensolver)
    int eig_dense_one_stage = 0;      // dense path: one-stage blocked Householder reduction instead of the two-stage one
    int eig_nullcheck = 1;            // known-null-vector shortcut (ss_nullcheck_kernel)
    int eig_keep_inertia_factor = 1;  // wide-band matrices with certified count 0 keep the factor of the inertia pass
    int band_assembly = 1;            // coarse-level agglomerate matrices assembled inside their band
    int eig_dedupe = 1;               // bitwise identical agglomerate matrices of a batch are solved once
    int eig_outer_panels = 8;         // 16-column panels per outer block of the wide-band factorisations (2: the right-looking two-panel walk)
    int overlap = 15;                 // bit 0 subspace iteration beside the next chunk, 1 halo exchange beside the interior rows, 2 Galerkin product beside the next level, 3 fine operator data beside the AE tables
    int sell = 31;                    // bit 0 coded slices at all, 1 pair coding, 2 short-chain kernel path, 3 operator-level dictionary, 4 node blocks, 5 (off) coded smoother diagonal
    int spmv_sell = 0;                // saamge_amd_spmv / spmv64 build and use the SELL copy
    int debug = 0;                    // bit 0 iteration traces of the few-eigenpairs path, 1 operator format census, 2 level tags in the kernel profile
    int host_heap_pad_mb = 256;       // > 0: glibc never trims its heap, serves blocks up to 32 MB from it and grows it in steps of this size (0: allocator left alone)
};
// Applied once, by the first hierarchy of the process (capi.hip): see Options::host_heap_pad_mb and DESIGN.md section 7.0.
void host_heap_policy();
Options &options();
bool env_timing();      // SAAMGE_AMD_TIMING
bool env_timing_host(); // SAAMGE_AMD_TIMING=host
bool env_serial();      // SAAMGE_AMD_SERIAL: no worker threads in the setup (counter passes)

// ---- device memory: caching allocator behind every DBuf -----------------------------------
// hipMalloc costs 10 us - 1 ms and hipFree additionally waits for the whole device, which stalls the
// host between setup kernels and serialises streams that are meant to overlap.  Freed blocks are kept
// and handed out again: at once to the stream that freed them (stream order makes that safe), to any
// other stream once the event recorded at the free has completed.  The freeing stream is the calling
// thread's `thread_stream()` (set by every API entry and every worker thread); a thread without one
// frees through hipFree as before.  Idle blocks beyond SAAMGE_AMD_POOL_MAX_GB (default 64) are returned
// to the driver, all of them by dev_pool_release() or when a hipMalloc fails.
void set_thread_stream(hipStream_t s);
void unset_thread_stream();      // the thread frees through hipFree again (its stream may be destroyed next)
hipStream_t thread_stream();
bool thread_stream_is_set();
void *dev_alloc(size_t bytes);
void dev_free(void *p) noexcept;
void dev_pool_release();
size_t dev_pool_idle_bytes();
// device bytes the library holds in DBufs right now and their high-water mark since the last reset
void dev_memory_stats(size_t *live, size_t *peak, bool reset_peak);
// requests of the pool that went to the driver since the last reset (hipMalloc calls and their bytes, hipFree of cached blocks)
void dev_pool_counts(long *n_malloc, long *n_free, size_t *malloc_bytes, bool reset);
struct ThreadStreamScope {       // the calling thread's stream for a scope (restored at its end)
    hipStream_t prev;
    bool had;
    explicit ThreadStreamScope(hipStream_t s) : prev(thread_stream()), had(thread_stream_is_set()) { set_thread_stream(s); }
    ~ThreadStreamScope() { if (had) set_thread_stream(prev); else unset_thread_stream(); }
};
// Close the batch of frees `s` is filling NOW (its event is recorded while the stream is certainly alive) -- called
// when a hierarchy is destroyed: its caller may destroy the stream next, and an open batch would later record its
// event on a dead handle.
void dev_pool_close_stream(hipStream_t s);

// RAII device buffer.
template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    DBuf() {}
    explicit DBuf(size_t n_) { alloc(n_); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    DBuf(DBuf &&o) noexcept : p(o.p), n(o.n), owned(o.owned) { o.p = nullptr; o.n = 0; }
    DBuf &operator=(DBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p; n = o.n; owned = o.owned;
            o.p = nullptr; o.n = 0;
        }
        return *this;
    }
    ~DBuf() { release(); }
    void release() {
        if (p && owned) dev_free(p);
        p = nullptr; n = 0; owned = true;
    }
    void alloc(size_t n_) {
        release();
        n = n_;
        if (n) p = (T *)dev_alloc(n * sizeof(T));
    }
    void zero(hipStream_t s = 0) {
        if (n) SA_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
    // copy from a host OR device pointer
    void assign(const T *src, size_t n_, hipStream_t s = 0) {
        alloc(n_);
        if (!n) return;
        SA_HIP_CHECK(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyDefault, s));
    }
    template <class V>
    void from_host(const V &v, hipStream_t s = 0) {
        alloc(v.size());
        if (!n) return;
        const size_t bytes = n * sizeof(T);
        // A large pageable vector goes through a page-locked block of the library's own (never returned to the system): handed
        // over as it is, the runtime registers the caller's pages with the GPU for the transfer and keeps the registration
        // cached -- pages that go back to the kernel when the vector dies (host_heap_policy, topology.hip).
        const bool pinned = std::is_same<typename V::allocator_type, PinnedAlloc<typename V::value_type>>::value;
        if (!pinned && bytes >= (256u << 10)) {
            void *stage = pinned_alloc(bytes);
            std::memcpy(stage, v.data(), bytes);
            const hipError_t e = hipMemcpyAsync(p, stage, bytes, hipMemcpyHostToDevice, s);
            const hipError_t e2 = hipStreamSynchronize(s);
            pinned_free(stage, bytes);
            SA_HIP_CHECK(e);
            SA_HIP_CHECK(e2);
            return;
        }
        SA_HIP_CHECK(hipMemcpyAsync(p, v.data(), bytes, hipMemcpyHostToDevice, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));  // v may be a temporary
    }
    hvec<T> to_host(hipStream_t s = 0) const {
        hvec<T> v(n);
        if (n) {
            SA_HIP_CHECK(hipMemcpyAsync(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
            SA_HIP_CHECK(hipStreamSynchronize(s));
        }
        return v;
    }
    // non-owning view of an existing device pointer
    void view(T *ptr, size_t n_) {
        release();
        p = ptr; n = n_; owned = false;
    }
};

// Bring an input array (host or device pointer) to the device: device pointers are
// viewed in place (zero copy), host pointers are uploaded once.
template <class T>
inline void import_array(DBuf<T> &dst, const T *src, size_t n, hipStream_t s) {
    if (is_device_ptr(src))
        dst.view(const_cast<T *>(src), n);
    else
        dst.assign(src, n, s);
}

// Fetch an input array to the host (for host-side topology).
template <class T>
inline hvec<T> fetch_host(const T *src, size_t n, hipStream_t s) {
    hvec<T> v(n);
    if (n) {
        SA_HIP_CHECK(hipMemcpyAsync(v.data(), src, n * sizeof(T), hipMemcpyDefault, s));
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
    return v;
}

// Row offsets of every CSR / SELL operator are 64-bit: column indices and dimensions are int32 like the
// reference's hypre/MFEM types, but the stored entries of one operator may exceed 2^31 (Q2 elasticity at
// 96^3: 4.2e9) -- the reference splits such an operator over MPI ranks, here one GPU holds it.
typedef int64_t roff_t;

constexpr int SELL_SEG_MAX = 16;      // segments per staged tile
constexpr int SELL_STAGE_CAP = 3584;  // doubles of x one tile may stage (28 KB of LDS)

// Device CSR matrix.
struct DCsr {
    int nrows = 0, ncols = 0;
    int64_t nnz = 0;
    DBuf<roff_t> rowptr;
    DBuf<int> col;
    DBuf<double> val;
    int lanes_per_row = 8;  // SpMV launch shape, chosen from the average row length
    mutable int max_row = -1;  // longest row (computed on first use by the fused AE assembly)
    // optional SELL-64 copy for the SpMV family: slice s = rows 64s..64s+63, entry (k, lane)
    // at sell_ptr[s] + 64 k + lane (padded with zero values), fully coalesced per wavefront
    bool has_sell = false;
    int nslices = 0;
    int64_t sell_size = 0;
    DBuf<roff_t> sell_ptr;
    DBuf<int> sell_col;
    DBuf<double> sell_val;
    // coded slices: <= 64 distinct offsets col - row -> sell_tab[64 s + code], one byte per entry
    // in sell_code (four consecutive entries of a row per word); sell_ntab[s] = -1: plain slice
    // pair-coded slices (sell_ntab >= 256): <= 64 distinct (offset, VALUE) pairs, sell_vtab holds the values
    DBuf<int> sell_ntab, sell_tab;
    DBuf<unsigned> sell_code;
    DBuf<double> sell_vtab;
    // census of the copy (build_sell): slices and stored entries per format [pair-coded, offset-coded, plain], the
    // bytes of matrix data one application streams in the formats in use, and whether the short-chain path of the
    // pair-coded slices may be used (32-bit byte offsets into x)
    int64_t sell_class_slices[3] = {0, 0, 0}, sell_class_entries[3] = {0, 0, 0};
    double sell_stream_bytes = 0.0;
    bool sell_fast_ok = false;
    // operator-level pair dictionary (sell_gdict_kernel): an operator none of whose slices could be coded per slice but
    // whose (offset, value) pairs repeat across the WHOLE operator (a uniform high-order mesh: Q2 elasticity has 243
    // entries per row and ~4 900 distinct pairs) stores a 16-bit code per entry into one table of 16-byte pairs:
    // 2 B instead of 12 B per stored entry.  sell_gcode: four codes of a row per 8-byte word, laid out like sell_code.
    bool sell_gpair = false;
    bool sell_bs3 = false;         // 3 x 3 node blocks: the lanes of a node share their gathers of x (sell_gpair_kernel)
    int sell_nirr = 0;             // rows outside regular node blocks ...
    DBuf<int> sell_irr;            // ... listed: sell_gpair3_fix_kernel redoes them
    int sell_ng = 0;
    DBuf<unsigned long long> sell_gcode;
    DBuf<double2> sell_gtab;       // {offset (as the low 32 bits of .x's pattern), value}: see GPair in sparse.hip
    // x-staging of the pair-coded slices (sell_stage_kernel): a TILE = 4 consecutive slices = the 256 rows of one workgroup.
    // Where the column offsets of a tile cluster into few runs (a stencil: 9), the x-entries those runs touch are
    // contiguous segments: sell_tile_nseg[t] of them (0: not staged), sell_tile_seg[SELL_SEG_MAX t + s] = {first offset
    // relative to the tile's first row, doubles to load}; the workgroup loads them into LDS with wide coalesced loads
    // and the products read LDS instead of gathering from global memory.  sell_stage_cap = doubles of the largest tile.
    DBuf<int> sell_tile_nseg;
    DBuf<int2> sell_tile_seg;
    int sell_stage_cap = 0;
    bool sell_one_table = false;   // every staged tile shares one pair table among its four slices
    // sell_staged2_kernel (one-table operators): the staged tiles' code words once more in a regular layout (word q of thread t
    // of tile T at (T sell_wq + q) 256 + t) and one descriptor word per tile (segments | the four slice widths)
    int sell_wq = 0;
    DBuf<unsigned> sell_codeR;
    // the smoother's diagonal factor as byte codes into a table of <= 256 values (operators whose rows repeat: a 256-row tile
    // then reads 256 bytes of it instead of 2 KB); sell_dsrc: the array the codes were made from (build_dinv_codes)
    DBuf<unsigned char> sell_dcode;
    DBuf<unsigned long long> sell_dtab;
    const double *sell_dsrc = nullptr;
    DBuf<int> sell_tile_desc;
    DBuf<int> sell_unstaged;       // tiles left to the gather kernel (sell_nunstaged of them)
    int sell_nunstaged = 0;
};

// exclusive scans (mis.hip); out has n + 1 entries
void exclusive_scan_int(hipStream_t s, int n, const int *in, int *out);
// the same without the host synchronisation: `tile_sums` is the caller's scratch of div_up(n, 1024) + 1 ints
void exclusive_scan_int_async(hipStream_t s, int n, const int *in, int *out, int *tile_sums);
void exclusive_scan_off(hipStream_t s, int n, const int *in, roff_t *out);

// Row offsets crossing the C ABI (sparse.hip).  In: int32 (the reference's HYPRE_Int; widened on the device)
// or int64 (device pointers viewed in place); host or device pointer either way.  Out: narrowed to int32 for
// the callers of the 32-bit getters (an operator beyond 2^31 entries is an error there).
void import_rowptr(DBuf<roff_t> &dst, const void *src, int bits, size_t n, hipStream_t s);
void export_rowptr32(int *dst_host, const DBuf<roff_t> &src, size_t n, hipStream_t s);

inline int pick_lanes_per_row(int64_t nnz, int nrows) {
    double avg = nrows ? double(nnz) / nrows : 1.0;
    int l = 1;
    while (l < 64 && l * 2 <= avg * 0.75 + 1.0) l *= 2;  // ~ half-full last pass at worst
    return l;
}

inline int div_up(int64_t a, int64_t b) { return int((a + b - 1) / b); }
// Home slot of `key` in an open-addressing table of `size` slots (a power of two >= 2): the HIGH bits of the multiplicative
// hash.  (Until round 4 the tables took the low bits, which are a permutation of the key's own low bits: the dofs of a box
// of a lexicographically numbered grid -- x + 257 y + 66049 z -- collide in lattices, 4.2 probes per look-up instead of 1.0
// for the 405 dofs of the headline's agglomerates in 1 024 slots.)
__host__ __device__ inline unsigned hash_home(unsigned key, unsigned size) {
    return (key * 2654435761u) >> (__builtin_clz(size) + 1);
}

// ---- optional per-kernel timing (bench.py's roofline leg) -------------------------
struct KernelStat {
    std::string name;
    double ms = 0.0;
    int64_t launches = 0;
    double bytes = 0.0;  // ALGORITHMIC bytes summed over launches (the CSR stream of the reference)
    double flops = 0.0;  // ALGORITHMIC flops summed over launches
    double fmt_bytes = 0.0;  // bytes the kernel has to move IN THE FORMAT IT RUNS (coded SELL slices, tables, vectors); 0: same as bytes
};

struct Profiler {
    bool enabled = false;
    std::vector<KernelStat> stats;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int level_tag = 0;  // SAAMGE_AMD_PROFILE_LEVELS=1: setup kernels of level l > 0 are listed as name@Ll
    KernelStat &get(const std::string &name) {
        for (auto &s : stats)
            if (s.name == name) return s;
        stats.push_back(KernelStat());
        stats.back().name = name;
        return stats.back();
    }
    void begin(hipStream_t s) {
        if (!enabled) return;
        if (!e0) {
            SA_HIP_CHECK(hipEventCreate(&e0));
            SA_HIP_CHECK(hipEventCreate(&e1));
        }
        SA_HIP_CHECK(hipEventRecord(e0, s));
    }
    void end(hipStream_t s, const char *name, double bytes, double flops, double fmt_bytes = 0.0) {
        if (!enabled) return;
        SA_HIP_CHECK(hipEventRecord(e1, s));
        SA_HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        SA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        KernelStat &k = level_tag > 0 ? get(std::string(name) + "@L" + std::to_string(level_tag)) : get(name);
        k.ms += ms;
        k.launches += 1;
        k.bytes += bytes;
        k.flops += flops;
        k.fmt_bytes += fmt_bytes > 0.0 ? fmt_bytes : bytes;
    }
};

Profiler &profiler();

}  // namespace saamge_amd
