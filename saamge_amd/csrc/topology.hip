// Integer topology of one level (single rank).  Output is bit-identical to the reference's
// tables; the reference's O(#MIS x ND) MIS loop (amg/src/aggregates.cpp:541-607) is replaced
// by a signature hash with the same numbering (first appearance scanning dofs upward, dofs
// inside a MIS ascending).  The tables that drive the numbering are built on the host with a
// small thread pool; dof_to_elem and elem_ldof (only consumed by the assembly kernel) are
// built on the device.
#include "topology.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <malloc.h>
#include <functional>
#include <thread>
#include <unordered_map>

#include <map>
#include <mutex>

namespace saamge_amd {

// ---------------------------------------------------------------------------------------
// pinned host memory pool
// ---------------------------------------------------------------------------------------
namespace {
std::mutex g_pin_mu;
std::multimap<size_t, void *> g_pin_free;
constexpr size_t PIN_MIN = 256 * 1024;
size_t pin_class(size_t bytes) {
    size_t c = PIN_MIN;
    while (c < bytes) c <<= 1;
    return c;
}
}  // namespace

void *pinned_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    if (bytes < PIN_MIN) return std::malloc(bytes);
    const size_t c = pin_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_free.find(c);
        if (it != g_pin_free.end()) {
            void *p = it->second;
            g_pin_free.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, c, hipHostMallocPortable) != hipSuccess || !p) {
        (void)hipGetLastError();
        throw std::bad_alloc();
    }
    return p;
}
void pinned_free(void *p, size_t bytes) {
    if (!p) return;
    if (bytes < PIN_MIN) {
        std::free(p);
        return;
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_free.insert(std::make_pair(pin_class(bytes), p));
}
void pinned_pool_release() {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto &kv : g_pin_free) (void)hipHostFree(kv.second);
    g_pin_free.clear();
}

// ---------------------------------------------------------------------------------------
// device memory pool (see common.h)
// ---------------------------------------------------------------------------------------
namespace {
// Blocks freed on one stream share an EVENT PER BATCH instead of one each: the frees of a hierarchy teardown
// (hundreds of blocks) used to put hundreds of markers into the stream, and on some processes the first kernel
// after them started 10 - 30 ms late (GPU idle in the kernel trace).  A batch ("epoch") stays open while its
// stream keeps freeing; its event is recorded -- on the freeing stream, after everything submitted to it so far,
// hence after the work that preceded every free of the batch -- when the batch is full or when another stream
// first asks for one of its blocks.  The freeing stream itself reuses its blocks at once, without any event.
struct Epoch {
    hipEvent_t ev = nullptr;
    hipStream_t stream = nullptr;
    int dev = 0;
    int refs = 0;           // idle blocks that belong to the batch
    bool recorded = false;
};
struct IdleBlock {
    void *p;
    Epoch *ep;
};
constexpr int POOL_EPOCH_BLOCKS = 64;
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, IdleBlock> idle;              // by size
    std::unordered_map<void *, size_t> live;            // blocks handed out -> size
    std::vector<hipEvent_t> events;                     // spare events
    std::map<std::pair<int, hipStream_t>, Epoch *> open;   // the batch each (device, stream) is filling
    size_t idle_bytes = 0, max_idle = 0;
    size_t live_bytes = 0, peak_bytes = 0;              // handed out now / high-water mark (dev_memory_stats)
    long n_malloc = 0, n_free = 0;                      // requests that went to the driver (dev_pool_counts)
    size_t malloc_bytes = 0;
    bool enabled = true;
    DevPool() {
        const char *e = std::getenv("SAAMGE_AMD_POOL_MAX_GB");
        const double gb = e ? std::atof(e) : 64.0;
        max_idle = (size_t)(gb * (double)(1ull << 30));
        enabled = gb > 0.0;
    }
};
DevPool &dev_pool() {
    static DevPool *p = new DevPool;      // never destroyed: static DBufs are released after main() returns
    return *p;
}
thread_local hipStream_t tl_stream = nullptr;
thread_local bool tl_stream_set = false;
inline size_t pool_round(size_t bytes) { return bytes <= (1u << 20) ? (bytes + 511) / 512 * 512 : (bytes + 65535) / 65536 * 65536; }
// pool lock held: one block leaves its batch; a recorded batch without blocks gives its event back
void epoch_unref(DevPool &P, Epoch *ep) {
    if (--ep->refs > 0 || !ep->recorded) return;
    P.events.push_back(ep->ev);
    delete ep;
}
// pool lock held: close the batch (record its event now) so that other streams can wait for it
bool epoch_record(DevPool &P, Epoch *ep) {
    if (ep->recorded) return true;
    if (hipEventRecord(ep->ev, ep->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    ep->recorded = true;
    auto it = P.open.find(std::make_pair(ep->dev, ep->stream));
    if (it != P.open.end() && it->second == ep) P.open.erase(it);
    return true;
}
// pool lock held: hipFree one idle block (hipFree waits for the device: whatever used the block is done)
std::multimap<size_t, IdleBlock>::iterator pool_drop_block(DevPool &P, std::multimap<size_t, IdleBlock>::iterator it) {
    (void)hipFree(it->second.p);
    ++P.n_free;
    Epoch *ep = it->second.ep;
    P.idle_bytes -= it->first;
    it = P.idle.erase(it);
    if (ep->refs == 1 && !ep->recorded) {      // the last block of an open batch: the batch goes with it
        auto o = P.open.find(std::make_pair(ep->dev, ep->stream));
        if (o != P.open.end() && o->second == ep) P.open.erase(o);
        ep->recorded = true;
    }
    epoch_unref(P, ep);
    return it;
}
// hipFree the idle blocks for which keep() is false; pool lock held
template <class F>
void pool_drop(DevPool &P, F keep) {
    for (auto it = P.idle.begin(); it != P.idle.end();) {
        if (keep(it)) { ++it; continue; }
        it = pool_drop_block(P, it);
    }
}
}  // namespace

Options &options() {
    static Options o;
    return o;
}
static int timing_mode() {      // 0 off, 1 SAAMGE_AMD_TIMING set, 2 SAAMGE_AMD_TIMING=host
    static const int v = [] {
        const char *e = std::getenv("SAAMGE_AMD_TIMING");
        return !e ? 0 : (e[0] == 'h' ? 2 : 1);
    }();
    return v;
}
bool env_timing() { return timing_mode() == 1; }
// SAAMGE_AMD_TIMING=host: the phases' host times WITHOUT synchronising the stream at their ends
bool env_timing_host() { return timing_mode() == 2; }
// The setup builds some tens of MB of host tables per hierarchy in std::vectors, copies some of them to and from the device
// (pageable memory: the runtime registers the pages with the GPU for the transfer and keeps such registrations cached) and
// releases them with the hierarchy.  What glibc then does with the memory decides how the NEXT setup starts:
//  * from the heap, with the default trimming: the top of the heap goes back to the kernel at every release and is grown
//    again by the next hierarchy.  Unmapping pages the GPU driver still knows invalidates its registration, and the
//    process's queues are stopped and restored: the first kernel of the next setup starts ~20 ms late (GPU trace: the queue
//    idle with the kernel submitted; 127 -> 150-160 ms per setup in half of the processes -- those in which glibc's sliding
//    mmap threshold had moved the tables onto the heap);
//  * from anonymous mappings of their own (the threshold frozen at its initial 128 KB): mapped and unmapped at recurring
//    addresses under the runtime's registration cache -- measured once: GPU memory access faults;
//  * from a heap that is never trimmed: neither.  That is what this asks for, once per process: blocks up to glibc's
//    maximum of 32 MB from the heap, no trimming, the heap grown in steps of host_heap_pad_mb.
// MPI libraries with registration caches set the same three parameters for the same reason.  DESIGN.md section 7.0.
void host_heap_policy() {
    static std::once_flag once;
    std::call_once(once, [] {
        const int mb = options().host_heap_pad_mb;
        if (mb <= 0) return;
        (void)mallopt(M_MMAP_THRESHOLD, 32 << 20);
        (void)mallopt(M_TRIM_THRESHOLD, 0x7ff00000);
        (void)mallopt(M_TOP_PAD, (int)std::min<long>((long)mb << 20, 0x7ff00000l));
    });
}
bool env_serial() {
    static const bool v = std::getenv("SAAMGE_AMD_SERIAL") != nullptr;
    return v;
}

void set_thread_stream(hipStream_t s) { tl_stream = s; tl_stream_set = true; }
void unset_thread_stream() { tl_stream = nullptr; tl_stream_set = false; }
hipStream_t thread_stream() { return tl_stream; }
bool thread_stream_is_set() { return tl_stream_set; }

void *dev_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    DevPool &P = dev_pool();
    const size_t want = pool_round(bytes);
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        // smallest idle block that fits without wasting more than an eighth; same stream, or idle for certain
        const size_t limit = want + want / 8 + 4096;
        for (auto it = P.idle.lower_bound(want); it != P.idle.end() && it->first <= limit; ++it) {
            Epoch *ep = it->second.ep;
            if (ep->dev != dev) continue;
            const bool same = tl_stream_set && ep->stream == tl_stream;
            if (!same) {
                if (!epoch_record(P, ep)) continue;
                if (hipEventQuery(ep->ev) != hipSuccess) { (void)hipGetLastError(); continue; }
            }
            void *p = it->second.p;
            P.live[p] = it->first;
            P.live_bytes += it->first;
            P.peak_bytes = std::max(P.peak_bytes, P.live_bytes);
            P.idle_bytes -= it->first;
            P.idle.erase(it);
            if (ep->refs == 1 && !ep->recorded) {      // an open batch that has just lost its last block stays open
                --ep->refs;
            } else {
                epoch_unref(P, ep);
            }
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {          // out of memory: give the cached blocks back and try once more
        (void)hipGetLastError();
        dev_pool_release();
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) throw Error((int)e, std::string("hipMalloc of ") + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[p] = want;
    P.live_bytes += want;
    P.peak_bytes = std::max(P.peak_bytes, P.live_bytes);
    ++P.n_malloc;
    P.malloc_bytes += want;
    return p;
}
void dev_pool_counts(long *n_malloc, long *n_free, size_t *malloc_bytes, bool reset) {
    DevPool &P = dev_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (n_malloc) *n_malloc = P.n_malloc;
    if (n_free) *n_free = P.n_free;
    if (malloc_bytes) *malloc_bytes = P.malloc_bytes;
    if (reset) { P.n_malloc = P.n_free = 0; P.malloc_bytes = 0; }
}

void dev_free(void *p) noexcept {
    if (!p) return;
    DevPool &P = dev_pool();
    size_t size = 0;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) { size = it->second; P.live.erase(it); P.live_bytes -= size; }
    }
    int dev = 0;
    if (!size || !P.enabled || !tl_stream_set || size > P.max_idle || hipGetDevice(&dev) != hipSuccess) {
        (void)hipFree(p);
        return;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    Epoch *&cur = P.open[std::make_pair(dev, tl_stream)];
    if (!cur) {
        hipEvent_t ev = nullptr;
        if (!P.events.empty()) { ev = P.events.back(); P.events.pop_back(); }
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            P.open.erase(std::make_pair(dev, tl_stream));
            (void)hipFree(p);
            return;
        }
        cur = new Epoch;
        cur->ev = ev;
        cur->stream = tl_stream;
        cur->dev = dev;
    }
    Epoch *ep = cur;
    ++ep->refs;
    P.idle.insert(std::make_pair(size, IdleBlock{p, ep}));
    P.idle_bytes += size;
    if (ep->refs >= POOL_EPOCH_BLOCKS) (void)epoch_record(P, ep);     // (full: closed, the next free opens a new one)
    while (P.idle_bytes > P.max_idle && !P.idle.empty())       // over the cap: the largest blocks go back to the driver
        (void)pool_drop_block(P, std::prev(P.idle.end()));
}

void dev_pool_close_stream(hipStream_t s) {
    DevPool &P = dev_pool();
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.open.find(std::make_pair(dev, s));
    if (it == P.open.end()) return;
    Epoch *ep = it->second;
    if (ep->refs == 0) {                  // an open batch without blocks: it just goes
        P.events.push_back(ep->ev);
        P.open.erase(it);
        delete ep;
        return;
    }
    if (!epoch_record(P, ep)) {           // (cannot record: wait for the stream instead, then the blocks are idle for certain)
        (void)hipStreamSynchronize(s);
        ep->recorded = true;
        P.open.erase(it);
    }
}

void dev_pool_release() {
    DevPool &P = dev_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    pool_drop(P, [](std::multimap<size_t, IdleBlock>::iterator) { return false; });
}

void dev_memory_stats(size_t *live, size_t *peak, bool reset_peak) {
    DevPool &P = dev_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    if (live) *live = P.live_bytes;
    if (peak) *peak = P.peak_bytes;
    if (reset_peak) P.peak_bytes = P.live_bytes;
}

size_t dev_pool_idle_bytes() {
    DevPool &P = dev_pool();
    std::lock_guard<std::mutex> lk(P.mu);
    return P.idle_bytes;
}

// ---------------------------------------------------------------------------------------
// tiny fork-join helper
// ---------------------------------------------------------------------------------------
static int num_threads() {
    static int nt = 0;
    if (!nt) {
        const char *e = std::getenv("SAAMGE_AMD_THREADS");
        nt = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
        if (nt < 1) nt = 1;
        if (nt > 16) nt = 16;
    }
    return nt;
}

// fn(begin, end, tid) over [0, n) split into contiguous ranges
static void parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int64_t grain = 4096) {
    int T = num_threads();
    if (n < grain * 2) T = 1;
    if (T == 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    const int dev = current_device();   // pinned allocations inside fn: see adopt_device
    for (int t = 0; t < T; ++t) {
        const int64_t b = n * t / T, e = n * (t + 1) / T;
        th.emplace_back([=, &fn]() {
            (void)hipSetDevice(dev);
            fn(b, e, t);
        });
    }
    for (auto &x : th) x.join();
}

hipStream_t side_stream(int slot) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, hipStream_t> streams;
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(mu);
    auto it = streams.find(std::make_pair(dev, slot));
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    SA_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    streams[std::make_pair(dev, slot)] = s;
    return s;
}

Table table_transpose(const Table &T) {
    Table R;
    const int nr = T.nrows();
    R.ncols = nr;
    R.I.assign((size_t)T.ncols + 1, 0);
    for (int v : T.J) R.I[(size_t)v + 1]++;
    for (int i = 0; i < T.ncols; ++i) R.I[i + 1] += R.I[i];
    R.J.resize(T.J.size());
    std::vector<int> pos(R.I.begin(), R.I.end() - 1);
    for (int i = 0; i < nr; ++i)
        for (int k = T.I[i]; k < T.I[i + 1]; ++k) R.J[pos[T.J[k]]++] = i;
    return R;
}

Table table_mult(const Table &A, const Table &B) {
    Table C;
    const int nr = A.nrows();
    C.ncols = B.ncols;
    C.I.assign((size_t)nr + 1, 0);
    std::vector<int> stamp((size_t)B.ncols, -1);
    for (int i = 0; i < nr; ++i) {
        int cnt = 0;
        for (int k = A.I[i]; k < A.I[i + 1]; ++k) {
            const int j = A.J[k];
            for (int q = B.I[j]; q < B.I[j + 1]; ++q) {
                const int c = B.J[q];
                if (stamp[c] != i) { stamp[c] = i; ++cnt; }
            }
        }
        C.I[i + 1] = C.I[i] + cnt;
    }
    C.J.resize((size_t)C.I[nr]);
    std::fill(stamp.begin(), stamp.end(), -1);
    for (int i = 0; i < nr; ++i) {
        int p = C.I[i];
        for (int k = A.I[i]; k < A.I[i + 1]; ++k) {
            const int j = A.J[k];
            for (int q = B.I[j]; q < B.I[j + 1]; ++q) {
                const int c = B.J[q];
                if (stamp[c] != i) { stamp[c] = i; C.J[p++] = c; }
            }
        }
    }
    return C;
}

// Stable parallel counting sort of items 0..n-1 by key(item) in [0, nkeys): returns the CSR
// (I, J) with J listing the items of each key in ascending item order (== mfem::Transpose).
static void counting_sort_rows(int64_t n, int nkeys, const std::function<int(int64_t)> &key,
                               hvec<int> &I, hvec<int> &J) {
    int T = num_threads();
    if (n < 1 << 16 || (int64_t)nkeys * T > ((int64_t)1 << 27)) T = 1;
    I.assign((size_t)nkeys + 1, 0);
    J.resize((size_t)n);
    if (T == 1) {
        for (int64_t i = 0; i < n; ++i) I[(size_t)key(i) + 1]++;
        for (int k = 0; k < nkeys; ++k) I[k + 1] += I[k];
        std::vector<int> pos(I.begin(), I.end() - 1);
        for (int64_t i = 0; i < n; ++i) J[pos[key(i)]++] = (int)i;
        return;
    }
    std::vector<std::vector<int>> cnt(T, std::vector<int>((size_t)nkeys, 0));
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            const int64_t b = n * t / T, e = n * (t + 1) / T;
            int *c = cnt[t].data();
            for (int64_t i = b; i < e; ++i) c[key(i)]++;
        });
    for (auto &x : th) x.join();
    th.clear();
    int run = 0;
    for (int k = 0; k < nkeys; ++k) {
        I[k] = run;
        for (int t = 0; t < T; ++t) {
            const int c = cnt[t][k];
            cnt[t][k] = run;
            run += c;
        }
    }
    I[nkeys] = run;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
            const int64_t b = n * t / T, e = n * (t + 1) / T;
            int *c = cnt[t].data();
            for (int64_t i = b; i < e; ++i) J[c[key(i)]++] = (int)i;
        });
    for (auto &x : th) x.join();
}

static inline uint64_t hash_row(const int *r, int n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (int i = 0; i < n; ++i) {
        h ^= (uint64_t)(uint32_t)r[i] + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 1099511628211ull;
    }
    return h;
}

void build_relations_ae(Relations &r, Table &&elem_to_dof, const hvec<int> &partitioning,
                        int nparts, int ND, const signed char *bdr) {
    const bool timing = env_timing();
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "TIMING:     topology %-22s %8.3f ms\n", what,
                     std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    r.ND = ND;
    r.nparts = nparts;
    r.elem_to_dof = std::move(elem_to_dof);
    r.elem_to_dof.ncols = ND;
    r.NE = r.elem_to_dof.nrows();
    const Table &e2d = r.elem_to_dof;
    SA_REQUIRE((int)partitioning.size() == r.NE, "partitioning size != number of elements");
    r.partitioning = partitioning;
    {
        std::atomic<int> bad(0);
        parallel_for(r.NE, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i)
                if (partitioning[i] < 0 || partitioning[i] >= nparts) bad = 1;
        });
        SA_REQUIRE(!bad, "partition id out of range");
        parallel_for((int64_t)e2d.J.size(), [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i)
                if (e2d.J[i] < 0 || e2d.J[i] >= ND) bad = 1;
        });
        SA_REQUIRE(!bad, "elem_to_dof entry out of range");
    }
    lap("checks");
    // AE_to_elem (agg_construct_tables_from_arr): elements of each AE, ascending
    r.AE_to_elem.ncols = r.NE;
    counting_sort_rows(r.NE, nparts, [&](int64_t e) { return partitioning[e]; }, r.AE_to_elem.I,
                       r.AE_to_elem.J);
    for (int p = 0; p < nparts; ++p) SA_REQUIRE(r.AE_to_elem.row_size(p) > 0, "empty agglomerate");
    lap("AE_to_elem");
    // AE_to_dof = AE_to_elem x elem_to_dof in first-encounter order, one hash set per thread
    r.AE_to_dof.ncols = ND;
    r.AE_to_dof.I.assign((size_t)nparts + 1, 0);
    std::vector<std::vector<int>> rows((size_t)nparts);
    parallel_for(nparts, [&](int64_t pb, int64_t pe, int) {
        std::vector<int> table;
        for (int64_t p = pb; p < pe; ++p) {
            size_t cand = 0;
            for (int k = r.AE_to_elem.I[p]; k < r.AE_to_elem.I[p + 1]; ++k) cand += (size_t)e2d.row_size(r.AE_to_elem.J[k]);
            size_t cap = 16;
            while (cap < 2 * cand) cap <<= 1;
            table.assign(cap, -1);
            std::vector<int> &out = rows[(size_t)p];
            out.reserve(cand / 2 + 8);
            for (int k = r.AE_to_elem.I[p]; k < r.AE_to_elem.I[p + 1]; ++k) {
                const int el = r.AE_to_elem.J[k];
                for (int q = e2d.I[el]; q < e2d.I[el + 1]; ++q) {
                    const int d = e2d.J[q];
                    size_t h = hash_home((unsigned)d, (unsigned)cap);
                    while (table[h] != -1 && table[h] != d) h = (h + 1) & (cap - 1);
                    if (table[h] == -1) {
                        table[h] = d;
                        out.push_back(d);
                    }
                }
            }
        }
    }, 8);
    for (int p = 0; p < nparts; ++p) r.AE_to_dof.I[p + 1] = r.AE_to_dof.I[p] + (int)rows[p].size();
    r.AE_to_dof.J.resize((size_t)r.AE_to_dof.I[nparts]);
    parallel_for(nparts, [&](int64_t pb, int64_t pe, int) {
        for (int64_t p = pb; p < pe; ++p)
            std::copy(rows[(size_t)p].begin(), rows[(size_t)p].end(), r.AE_to_dof.J.begin() + r.AE_to_dof.I[p]);
    }, 8);
    rows.clear();
    rows.shrink_to_fit();
    lap("AE_to_dof");
    // dof_to_AE = transpose, rows ascending in AE: atomic counts + cursors, then sort the short rows
    const int64_t nconn = (int64_t)r.AE_to_dof.J.size();
    r.dof_to_AE.ncols = nparts;
    r.dof_to_AE.I.assign((size_t)ND + 1, 0);
    {
        std::vector<int> cnt((size_t)ND, 0);
        parallel_for(nconn, [&](int64_t b, int64_t e, int) {
            for (int64_t k = b; k < e; ++k) __atomic_fetch_add(&cnt[r.AE_to_dof.J[k]], 1, __ATOMIC_RELAXED);
        });
        for (int i = 0; i < ND; ++i) {
            SA_REQUIRE(cnt[i] > 0, "dof without any element");
            r.dof_to_AE.I[i + 1] = r.dof_to_AE.I[i] + cnt[i];
        }
        r.dof_to_AE.J.resize((size_t)nconn);
        std::fill(cnt.begin(), cnt.end(), 0);
        parallel_for(nparts, [&](int64_t pb, int64_t pe, int) {
            for (int64_t p = pb; p < pe; ++p)
                for (int k = r.AE_to_dof.I[p]; k < r.AE_to_dof.I[p + 1]; ++k) {
                    const int d = r.AE_to_dof.J[k];
                    const int pos = __atomic_fetch_add(&cnt[d], 1, __ATOMIC_RELAXED);
                    r.dof_to_AE.J[(size_t)r.dof_to_AE.I[d] + pos] = (int)p;
                }
        }, 8);
        parallel_for(ND, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i)
                if (r.dof_to_AE.I[i + 1] - r.dof_to_AE.I[i] > 1)
                    std::sort(r.dof_to_AE.J.begin() + r.dof_to_AE.I[i], r.dof_to_AE.J.begin() + r.dof_to_AE.I[i + 1]);
        });
    }
    lap("dof_to_AE");
    // dof_id_inAE (agg_build_glob_to_AE_id_map, :1202-1244)
    r.dof_id_inAE.assign((size_t)nconn, -1);
    parallel_for(nparts, [&](int64_t pb, int64_t pe, int) {
        for (int64_t p = pb; p < pe; ++p)
            for (int k = r.AE_to_dof.I[p]; k < r.AE_to_dof.I[p + 1]; ++k) {
                const int d = r.AE_to_dof.J[k];
                const int *row = r.dof_to_AE.row(d);
                const int rs = r.dof_to_AE.row_size(d);
                const int t = (int)(std::lower_bound(row, row + rs, (int)p) - row);
                r.dof_id_inAE[(size_t)r.dof_to_AE.I[d] + t] = k - r.AE_to_dof.I[p];
            }
    }, 8);
    lap("dof_id_inAE");
    // flags (agg_construct_agg_flags, :198-216)
    r.agg_flags.assign((size_t)ND, 0);
    parallel_for(ND, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            signed char f = bdr ? bdr[i] : 0;
            if (r.dof_to_AE.row_size((int)i) > 1) f |= FLAG_BETWEEN_AES;
            r.agg_flags[(size_t)i] = f;
        }
    });
    lap("flags");
}

// Part 2 (MIS tables): independent of the device work on the AE matrices, so the caller runs
// it on a host thread while the GPU solves the local eigenproblems.
// agg_construct_aggregate_mises (amg/src/aggregates.cpp:324-487) + Arbitrator::suggest
// (amg/src/arbitrator.cpp:93-204): the greedy distribution is sequential by construction (every
// decision sees the earlier ones), so it runs on one host thread over the rows of the interface
// dofs only.  Ties: first maximum in the stored column order of A's row.
static void arbitrate_aggregates(Relations &r, const HostCsr &A) {
    const int ND = r.ND, nparts = r.nparts;
    SA_REQUIRE(A.nrows == ND, "aggregates: level matrix / dof count mismatch");
    r.mises.assign((size_t)ND, -2);
    std::vector<int> size((size_t)nparts, 0);
    std::vector<double> diag((size_t)ND, 0.0);
    parallel_for(ND, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i)
            for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
                if (A.col[k] == (int)i) diag[(size_t)i] = A.val[k];
    });
    for (int i = 0; i < ND; ++i)
        if (r.dof_to_AE.row_size(i) == 1) {
            const int p = r.dof_to_AE.row(i)[0];
            r.mises[(size_t)i] = p;
            ++size[(size_t)p];
        }
    for (int i = 0; i < ND; ++i) {
        if (r.mises[(size_t)i] != -2) continue;
        const int *parts = r.dof_to_AE.row(i);
        const int np = r.dof_to_AE.row_size(i);
        int agg = -1;
        double max_stren = -1.0;
        if (A.rowptr[i + 1] - A.rowptr[i] > 1)
            for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
                const int nb = A.col[k];
                const int a = r.mises[(size_t)nb];
                if (nb == i || a < 0 || std::find(parts, parts + np, a) == parts + np) continue;
                const double strength = std::fabs(A.val[k]) / std::sqrt(diag[(size_t)i] * diag[(size_t)nb]);
                if (strength > max_stren) { max_stren = strength; agg = a; }
            }
        if (max_stren < 0.0) {
            agg = parts[0];
            for (int j = 1; j < np; ++j)
                if (size[(size_t)agg] > size[(size_t)parts[j]]) agg = parts[j];
        }
        r.mises[(size_t)i] = agg;
        ++size[(size_t)agg];
    }
}

void build_relations_mis(Relations &r, const HostCsr *aggregates_A) {
    const bool timing = env_timing();
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "TIMING:     topology %-22s %8.3f ms\n", what,
                     std::chrono::duration<double, std::milli>(t - t_last).count());
        t_last = t;
    };
    const int ND = r.ND, nparts = r.nparts;
    const bool aggregates = aggregates_A != nullptr;
    std::vector<int> reps;
    if (aggregates) {
        // ---- last coarsening with do_aggregates: one "MIS" per AE, mis_to_AE = identity ----
        arbitrate_aggregates(r, *aggregates_A);
        r.num_mises = nparts;
        lap("aggregates (arbitration)");
    } else {
    // ---- MISes: groups of dofs with identical AE lists, numbered by first appearance ----
    // group representative = smallest dof of the group
    std::vector<int> rep_of((size_t)ND, -1);        // dof -> representative dof
    {
        const int T = num_threads();
        // single-AE dofs: representative = min dof per AE
        std::vector<std::vector<int>> mins(T, std::vector<int>((size_t)nparts, ND));
        parallel_for(ND, [&](int64_t b, int64_t e, int t) {
            int *m = mins[t].data();
            for (int64_t i = b; i < e; ++i)
                if (r.dof_to_AE.row_size((int)i) == 1) {
                    const int p = r.dof_to_AE.row((int)i)[0];
                    if ((int)i < m[p]) m[p] = (int)i;
                }
        }, 1);
        std::vector<int> single((size_t)nparts, ND);
        for (int t = 0; t < T; ++t)
            for (int p = 0; p < nparts; ++p) single[p] = std::min(single[p], mins[t][p]);
        mins.clear();
        // multi-AE dofs: hash of the AE list; dofs are bucketed by hash (stable counting sort, so
        // each bucket lists its dofs in ascending order) and every bucket is grouped by one thread
        std::vector<uint64_t> hv((size_t)ND, 0);
        parallel_for(ND, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i) {
                const int rs = r.dof_to_AE.row_size((int)i);
                if (rs > 1) hv[(size_t)i] = hash_row(r.dof_to_AE.row((int)i), rs);
                else rep_of[(size_t)i] = single[r.dof_to_AE.row((int)i)[0]];
            }
        });
        const int NBK = 4 * T;
        hvec<int> bk_I, bk_J;
        counting_sort_rows(ND, NBK + 1, [&](int64_t i) {
            return r.dof_to_AE.row_size((int)i) > 1 ? (int)((hv[(size_t)i] >> 40) % (uint64_t)NBK) : NBK;
        }, bk_I, bk_J);
        std::atomic<int> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                for (;;) {
                    const int bk = next.fetch_add(1);
                    if (bk >= NBK) break;
                    std::unordered_map<uint64_t, std::vector<int>> groups;  // hash -> representatives
                    groups.reserve((size_t)(bk_I[bk + 1] - bk_I[bk]) / 4 + 16);
                    for (int q = bk_I[bk]; q < bk_I[bk + 1]; ++q) {
                        const int i = bk_J[q];
                        const int rs = r.dof_to_AE.row_size(i);
                        std::vector<int> &cands = groups[hv[(size_t)i]];
                        const int *row = r.dof_to_AE.row(i);
                        int rep = -1;
                        for (int c : cands)
                            if (r.dof_to_AE.row_size(c) == rs && std::equal(row, row + rs, r.dof_to_AE.row(c))) { rep = c; break; }
                        if (rep < 0) { rep = i; cands.push_back(i); }   // ascending scan: first seen = smallest
                        rep_of[(size_t)i] = rep;
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    // MIS ids = rank of the representative among all representatives (first appearance order)
    for (int i = 0; i < ND; ++i)
        if (rep_of[i] == i) reps.push_back(i);
    r.num_mises = (int)reps.size();
    std::vector<int> mis_of_rep((size_t)ND, -1);
    for (int m = 0; m < r.num_mises; ++m) mis_of_rep[reps[m]] = m;
    r.mises.assign((size_t)ND, -1);
    parallel_for(ND, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) r.mises[(size_t)i] = mis_of_rep[rep_of[(size_t)i]];
    });
    lap("MIS ids");
    }
    // mis_to_dof: dofs of each MIS ascending
    r.mis_to_dof.ncols = ND;
    counting_sort_rows(ND, r.num_mises, [&](int64_t i) { return r.mises[(size_t)i]; }, r.mis_to_dof.I,
                       r.mis_to_dof.J);
    r.dof_row_in_mis.assign((size_t)ND, 0);
    parallel_for(r.num_mises, [&](int64_t mb, int64_t me, int) {
        for (int64_t m = mb; m < me; ++m)
            for (int k = r.mis_to_dof.I[m]; k < r.mis_to_dof.I[m + 1]; ++k)
                r.dof_row_in_mis[(size_t)r.mis_to_dof.J[k]] = k - r.mis_to_dof.I[m];
    }, 64);
    lap("mis_to_dof");
    // mis_to_AE = mis_to_dof x dof_to_AE == the (ascending) AE list of any member dof
    r.mis_to_AE.ncols = nparts;
    r.mis_to_AE.I.assign((size_t)r.num_mises + 1, 0);
    for (int m = 0; m < r.num_mises; ++m)
        r.mis_to_AE.I[m + 1] = r.mis_to_AE.I[m] + (aggregates ? 1 : r.dof_to_AE.row_size(reps[m]));
    r.mis_to_AE.J.resize((size_t)r.mis_to_AE.I[r.num_mises]);
    parallel_for(r.num_mises, [&](int64_t mb, int64_t me, int) {
        for (int64_t m = mb; m < me; ++m)
            if (aggregates)
                r.mis_to_AE.J[(size_t)m] = (int)m;     // IdentityTable, aggregates.cpp:770-771
            else
                std::copy(r.dof_to_AE.row(reps[m]), r.dof_to_AE.row(reps[m]) + r.dof_to_AE.row_size(reps[m]),
                          r.mis_to_AE.J.begin() + r.mis_to_AE.I[m]);
    }, 64);
    // AE_to_mis = transpose (ascending MIS ids == the "sorted" order of elmat.cpp:121-123);
    // ae_pair keeps, for each (AE, mis) entry, the id of the (mis, AE) pair
    const int64_t npairs = (int64_t)r.mis_to_AE.J.size();
    r.AE_to_mis.ncols = r.num_mises;
    counting_sort_rows(npairs, nparts, [&](int64_t q) { return r.mis_to_AE.J[(size_t)q]; }, r.AE_to_mis.I,
                       r.ae_pair);
    r.AE_to_mis.J.resize((size_t)npairs);
    {
        // pair q belongs to the MIS whose row contains q
        std::vector<int> mis_of_pair((size_t)npairs);
        parallel_for(r.num_mises, [&](int64_t mb, int64_t me, int) {
            for (int64_t m = mb; m < me; ++m)
                for (int q = r.mis_to_AE.I[m]; q < r.mis_to_AE.I[m + 1]; ++q) mis_of_pair[(size_t)q] = (int)m;
        }, 64);
        parallel_for(npairs, [&](int64_t b, int64_t e, int) {
            for (int64_t k = b; k < e; ++k) r.AE_to_mis.J[(size_t)k] = mis_of_pair[(size_t)r.ae_pair[(size_t)k]];
        });
    }
    lap("mis_to_AE/AE_to_mis");
    // (MIS, AE) pairs: AE-local indices of the MIS dofs
    r.pair_loc_off.assign((size_t)npairs + 1, 0);
    for (int m = 0; m < r.num_mises; ++m) {
        const int sz = r.mis_to_dof.row_size(m);
        for (int q = r.mis_to_AE.I[m]; q < r.mis_to_AE.I[m + 1]; ++q)
            r.pair_loc_off[(size_t)q + 1] = r.pair_loc_off[q] + sz;
    }
    r.pair_loc.resize((size_t)r.pair_loc_off[npairs]);
    parallel_for(r.num_mises, [&](int64_t mb, int64_t me, int) {
        for (int64_t m = mb; m < me; ++m) {
            const int na = r.mis_to_AE.row_size((int)m);
            const int sz = r.mis_to_dof.row_size((int)m);
            for (int t = 0; t < na; ++t) {
                const int q = r.mis_to_AE.I[m] + t;
                int *dst = r.pair_loc.data() + r.pair_loc_off[q];
                for (int k = 0; k < sz; ++k) {
                    const int dof = r.mis_to_dof.J[(size_t)r.mis_to_dof.I[m] + k];
                    // every dof of the MIS has the same AE list: the t-th AE of the dof's row
                    // (aggregates: the position of AE m in the dof's row)
                    int tt = t;
                    if (aggregates) {
                        const int *row = r.dof_to_AE.row(dof);
                        tt = (int)(std::find(row, row + r.dof_to_AE.row_size(dof), (int)m) - row);
                    }
                    dst[k] = r.dof_id_inAE[(size_t)r.dof_to_AE.I[dof] + tt];
                }
            }
        }
    }, 64);
    lap("pairs");
}

// ---------------------------------------------------------------------------------------
// device-side pieces: dof_to_elem (transpose with ascending rows) and elem_ldof
// ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void d2e_count_kernel(long nconn, const int *__restrict__ e2d_J,
                                                        int *__restrict__ cnt) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k < nconn) atomicAdd(&cnt[e2d_J[k]], 1);
}
__global__ __launch_bounds__(256) void d2e_fill_kernel(int NE, const int *__restrict__ e2d_I,
                                                       const int *__restrict__ e2d_J,
                                                       const int *__restrict__ d2e_I,
                                                       int *__restrict__ cursor, int *__restrict__ d2e_J) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= NE) return;
    for (int k = e2d_I[e]; k < e2d_I[e + 1]; ++k) {
        const int d = e2d_J[k];
        d2e_J[d2e_I[d] + atomicAdd(&cursor[d], 1)] = (int)e;
    }
}
__global__ __launch_bounds__(256) void d2e_sort_kernel(int ND, const int *__restrict__ d2e_I,
                                                       int *__restrict__ d2e_J) {
    const long d = (long)blockIdx.x * 256 + threadIdx.x;
    if (d >= ND) return;
    const int b = d2e_I[d], e = d2e_I[d + 1];
    for (int i = b + 1; i < e; ++i) {  // insertion sort, rows are short
        const int v = d2e_J[i];
        int j = i - 1;
        while (j >= b && d2e_J[j] > v) { d2e_J[j + 1] = d2e_J[j]; --j; }
        d2e_J[j + 1] = v;
    }
}
__global__ __launch_bounds__(256) void elem_ldof_kernel(int NE, const int *__restrict__ e2d_I,
                                                        const int *__restrict__ e2d_J,
                                                        const int *__restrict__ part,
                                                        const int *__restrict__ d2ae_I,
                                                        const int *__restrict__ d2ae_J,
                                                        const int *__restrict__ dof_id_inAE,
                                                        int *__restrict__ elem_ldof) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= NE) return;
    const int p = part[e];
    for (int k = e2d_I[e]; k < e2d_I[e + 1]; ++k) {
        const int d = e2d_J[k];
        int idx = d2ae_I[d];
        while (d2ae_J[idx] != p) ++idx;  // every element dof lies in the element's AE
        elem_ldof[k] = dof_id_inAE[idx];
    }
}

void upload_relations_ae(DevRelations &d, const Relations &r, hipStream_t s) {
    d.e2d_I.from_host(r.elem_to_dof.I, s);
    d.e2d_J.from_host(r.elem_to_dof.J, s);
    d.part.from_host(r.partitioning, s);
    d.ae2d_I.from_host(r.AE_to_dof.I, s);
    d.ae2d_J.from_host(r.AE_to_dof.J, s);
    d.d2ae_I.from_host(r.dof_to_AE.I, s);
    d.d2ae_J.from_host(r.dof_to_AE.J, s);
    d.dof_id_inAE.from_host(r.dof_id_inAE, s);
    d.flags.from_host(r.agg_flags, s);
    // dof_to_elem on the device
    const long nconn = (long)r.elem_to_dof.J.size();
    d.d2e_I.alloc((size_t)r.ND + 1);
    d.d2e_J.alloc((size_t)nconn);
    d.elem_ldof.alloc((size_t)nconn);
    {
        DBuf<int> cnt((size_t)r.ND);
        cnt.zero(s);
        hipLaunchKernelGGL(d2e_count_kernel, dim3(div_up(nconn, 256)), dim3(256), 0, s, nconn, d.e2d_J.p, cnt.p);
        exclusive_scan_int(s, r.ND, cnt.p, d.d2e_I.p);
        cnt.zero(s);
        hipLaunchKernelGGL(d2e_fill_kernel, dim3(div_up(r.NE, 256)), dim3(256), 0, s, r.NE, d.e2d_I.p,
                           d.e2d_J.p, d.d2e_I.p, cnt.p, d.d2e_J.p);
        hipLaunchKernelGGL(d2e_sort_kernel, dim3(div_up(r.ND, 256)), dim3(256), 0, s, r.ND, d.d2e_I.p, d.d2e_J.p);
        hipLaunchKernelGGL(elem_ldof_kernel, dim3(div_up(r.NE, 256)), dim3(256), 0, s, r.NE, d.e2d_I.p,
                           d.e2d_J.p, d.part.p, d.d2ae_I.p, d.d2ae_J.p, d.dof_id_inAE.p, d.elem_ldof.p);
        SA_HIP_CHECK(hipGetLastError());
        SA_HIP_CHECK(hipStreamSynchronize(s));
    }
}

// ---------------------------------------------------------------------------------------
// Device build of the AE tables (level 0 with device-resident inputs): the same tables as
// build_relations_ae, bit for bit, without moving elem_to_dof to the host.
//   AE_to_elem   count + scan + atomic fill, rows sorted in LDS (bitonic)       ascending elements
//   AE_to_dof    one workgroup per AE: LDS hash set keyed by dof holding the MIN rank of its
//                appearances (rank = element position * nde + slot); a dof is kept at its minimum
//                rank, so compacting the ranks in order gives mfem::Mult's first-encounter order
//   dof_to_AE    count + scan + atomic fill, short rows sorted per dof (with dof_id_inAE payload)
// ---------------------------------------------------------------------------------------
constexpr int TOPO_MAXK = 4096;   // candidate (element, slot) pairs per AE handled in LDS

__global__ __launch_bounds__(256) void topo_check_kernel(long NE, long nconn, int nparts, int ND,
                                                         const int *__restrict__ part,
                                                         const int *__restrict__ e2d_J, int *__restrict__ err) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < NE && (part[i] < 0 || part[i] >= nparts)) atomicOr(err, 1);
    if (i < nconn && (e2d_J[i] < 0 || e2d_J[i] >= ND)) atomicOr(err, 2);
}
__global__ __launch_bounds__(256) void iota_scaled_kernel(long n, int scale, int *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int)(i * scale);
}
__global__ __launch_bounds__(256) void key_count_kernel(long n, const int *__restrict__ key, int *__restrict__ cnt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[key[i]], 1);
}
// Count and fill for keys that come in runs (the partition of lexicographically numbered elements: eight neighbours share an
// agglomerate): one atomic per run of equal keys in a wavefront instead of one per element -- the 64 atomics of a wavefront fell
// on eight counters.  The order inside a row is whatever the atomics give, as before (row_sort_kernel follows).
__device__ inline void run_of_lane(int k, int lane, int &head, int &len) {
    const int prev = __shfl_up(k, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || k != prev);
    head = 63 - __clzll(heads & (~0ull >> (63 - lane)));
    const unsigned long long after = head == 63 ? 0ull : heads >> (head + 1);
    len = after ? __ffsll((long long)after) : 64 - head;
}
__global__ __launch_bounds__(256) void key_count_runs_kernel(long n, const int *__restrict__ key, int *__restrict__ cnt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int k = i < n ? key[i] : -1 - lane;
    int head, len;
    run_of_lane(k, lane, head, len);
    if (lane == head && i < n) atomicAdd(&cnt[k], len);
}
__global__ __launch_bounds__(256) void key_fill_runs_kernel(long n, const int *__restrict__ key,
                                                            const int *__restrict__ rowI, int *__restrict__ cursor,
                                                            int *__restrict__ J) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int k = i < n ? key[i] : -1 - lane;
    int head, len;
    run_of_lane(k, lane, head, len);
    int base = 0;
    if (lane == head && i < n) base = rowI[k] + atomicAdd(&cursor[k], len);
    base = __shfl(base, head, 64);
    if (i < n) J[base + (lane - head)] = (int)i;
}
__global__ __launch_bounds__(256) void any_zero_kernel(long n, const int *__restrict__ cnt, int bit, int *__restrict__ err) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && cnt[i] == 0) atomicOr(err, bit);
}
// ascending sort of every row (row length <= cap, a power of two <= 4096), one workgroup per row
__global__ __launch_bounds__(256) void row_sort_kernel(const int *__restrict__ I, int *__restrict__ J, int cap) {
    extern __shared__ int sh[];
    const int p = blockIdx.x;
    const int b = I[p], len = I[p + 1] - b;
    for (int i = threadIdx.x; i < cap; i += 256) sh[i] = (i < len) ? J[b + i] : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= cap; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < cap; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int a = sh[i], c = sh[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { sh[i] = c; sh[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < len; i += 256) J[b + i] = sh[i];
}

// mode 0: rowcnt[p] = number of distinct dofs; mode 1: write them at out[outI[p] ...] in
// first-encounter order
__global__ __launch_bounds__(256) void ae_pack_lists_kernel(int nde, const int *__restrict__ ae2e_I, const int *__restrict__ outI,
                                                            const int *__restrict__ padded, int *__restrict__ out) {
    const int p = blockIdx.x, n = outI[p + 1] - outI[p];
    const int *src = padded + (size_t)ae2e_I[p] * nde;
    int *dst = out + outI[p];
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void ae_to_dof_kernel(int mode, int nde, int HS,
                                                        const int *__restrict__ ae2e_I,
                                                        const int *__restrict__ ae2e_J,
                                                        const int *__restrict__ e2d_J, int *__restrict__ rowcnt,
                                                        const int *__restrict__ outI, int *__restrict__ out) {
    extern __shared__ int sh[];
    __shared__ int wsum[4], total;
    const int p = blockIdx.x, tid = threadIdx.x;
    const int eb = ae2e_I[p], ne = ae2e_I[p + 1] - eb;
    const int K = ne * nde;
    const int Kpad = (K + 255) & ~255;
    int *keys = sh, *ranks = sh + HS, *first = sh + 2 * HS, *vals = first + Kpad;
    for (int i = tid; i < HS; i += 256) { keys[i] = -1; ranks[i] = 0x7fffffff; }
    for (int i = tid; i < Kpad; i += 256) first[i] = 0;
    // (the dofs first, into LDS: the loads of a thread's entries are independent of the table and go out together; with the
    // inserts in the same loop every entry waited for two dependent global loads behind the previous entry's atomics)
    for (int idx = tid; idx < K; idx += 256) {
        const int q = idx / nde, t = idx - q * nde;
        vals[idx] = e2d_J[(size_t)ae2e_J[eb + q] * nde + t];
    }
    __syncthreads();
    for (int idx = tid; idx < K; idx += 256) {
        const int d = vals[idx];
        unsigned h = hash_home((unsigned)d, (unsigned)HS);
        for (;;) {
            const int prev = atomicCAS(&keys[h], -1, d);
            if (prev == -1 || prev == d) { atomicMin(&ranks[h], idx); break; }
            h = (h + 1) & (unsigned)(HS - 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < HS; i += 256)
        if (keys[i] != -1) first[ranks[i]] = 1;
    __syncthreads();
    // exclusive scan of first[0..Kpad): each thread owns a contiguous run
    const int per = Kpad / 256;
    int run = 0;
    for (int u = 0; u < per; ++u) run += first[tid * per + u];
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 4; ++w) { const int v = wsum[w]; wsum[w] = acc; acc += v; }
        total = acc;
    }
    __syncthreads();
    if (mode == 0) {
        if (tid == 0) rowcnt[p] = total;
        return;
    }
    // mode 2: the count AND the list, the list into a padded buffer (K slots per AE at eb * nde: the offsets of the packed
    // lists are not known yet) -- one pass over the elements instead of a counting pass and a filling pass
    if (mode == 2 && tid == 0) rowcnt[p] = total;
    int pos = wsum[tid >> 6] + incl - run;
    int *o = mode == 2 ? out + (size_t)eb * nde : out + outI[p];
    for (int u = 0; u < per; ++u) {
        const int idx = tid * per + u;
        if (first[idx]) o[pos++] = vals[idx];
    }
}

__global__ __launch_bounds__(256) void d2ae_fill_kernel(const int *__restrict__ ae2d_I, const int *__restrict__ ae2d_J,
                                                        const int *__restrict__ d2ae_I, int *__restrict__ cursor,
                                                        int *__restrict__ d2ae_J, int *__restrict__ did) {
    const int p = blockIdx.x;
    const int b = ae2d_I[p], e = ae2d_I[p + 1];
    for (int k = b + threadIdx.x; k < e; k += 256) {
        const int d = ae2d_J[k];
        const int pos = d2ae_I[d] + atomicAdd(&cursor[d], 1);
        d2ae_J[pos] = p;
        did[pos] = k - b;
    }
}
__global__ __launch_bounds__(256) void d2ae_sort_flags_kernel(int ND, const int *__restrict__ d2ae_I,
                                                              int *__restrict__ d2ae_J, int *__restrict__ did,
                                                              const signed char *__restrict__ bdr,
                                                              signed char *__restrict__ flags) {
    const long d = (long)blockIdx.x * 256 + threadIdx.x;
    if (d >= ND) return;
    const int b = d2ae_I[d], e = d2ae_I[d + 1];
    for (int i = b + 1; i < e; ++i) {  // insertion sort by AE id, rows are short
        const int v = d2ae_J[i], w = did[i];
        int j = i - 1;
        while (j >= b && d2ae_J[j] > v) { d2ae_J[j + 1] = d2ae_J[j]; did[j + 1] = did[j]; --j; }
        d2ae_J[j + 1] = v;
        did[j + 1] = w;
    }
    signed char f = bdr ? bdr[d] : 0;
    if (e - b > 1) f |= FLAG_BETWEEN_AES;
    flags[d] = f;
}

template <class T>
static void download(hvec<T> &dst, const DBuf<T> &src, size_t n, hipStream_t s) {
    dst.resize(n);
    if (n) SA_HIP_CHECK(hipMemcpyAsync(dst.data(), src.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
}

bool build_relations_ae_device(Relations &r, DevRelations &d, const int *e2d_dev, int NE, int nde,
                               const int *part_dev, int nparts, int ND, const signed char *bdr_dev,
                               hipStream_t s) {
    const long nconn = (long)NE * nde;
    r.ND = ND;
    r.NE = NE;
    r.nparts = nparts;
    d.e2d_J.view(const_cast<int *>(e2d_dev), (size_t)nconn);
    d.part.view(const_cast<int *>(part_dev), (size_t)NE);
    d.e2d_I.alloc((size_t)NE + 1);
    hipLaunchKernelGGL(iota_scaled_kernel, dim3(div_up((long)NE + 1, 256)), dim3(256), 0, s, (long)NE + 1, nde, d.e2d_I.p);
    DBuf<int> err(1), cnt((size_t)std::max(nparts, ND) + 1);
    err.zero(s);
    hipLaunchKernelGGL(topo_check_kernel, dim3(div_up(nconn, 256)), dim3(256), 0, s, (long)NE, nconn, nparts, ND,
                       part_dev, e2d_dev, err.p);
    {
        auto h = err.to_host(s);
        SA_REQUIRE(!(h[0] & 1), "partition id out of range");
        SA_REQUIRE(!(h[0] & 2), "elem_to_dof entry out of range");
    }
    // dof_to_elem (for the assembly kernels, as in upload_relations_ae) depends on the elements alone: its five kernels --
    // counters and cursors of global atomics, latency-bound -- run on a side stream beside the agglomerate tables
    hipStream_t s2 = side_stream(7);
    DBuf<int> cnt2((size_t)ND + 1), sums2((size_t)div_up(ND, 1024) + 1);
    d.d2e_I.alloc((size_t)ND + 1);
    d.d2e_J.alloc((size_t)nconn);
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    SA_HIP_CHECK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    SA_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    {
        SA_HIP_CHECK(hipEventRecord(ev_fork, s));
        SA_HIP_CHECK(hipStreamWaitEvent(s2, ev_fork, 0));
        SA_HIP_CHECK(hipMemsetAsync(cnt2.p, 0, sizeof(int) * (size_t)ND, s2));
        hipLaunchKernelGGL(d2e_count_kernel, dim3(div_up(nconn, 256)), dim3(256), 0, s2, nconn, d.e2d_J.p, cnt2.p);
        exclusive_scan_int_async(s2, ND, cnt2.p, d.d2e_I.p, sums2.p);
        SA_HIP_CHECK(hipMemsetAsync(cnt2.p, 0, sizeof(int) * (size_t)ND, s2));
        hipLaunchKernelGGL(d2e_fill_kernel, dim3(div_up(NE, 256)), dim3(256), 0, s2, NE, d.e2d_I.p, d.e2d_J.p, d.d2e_I.p,
                           cnt2.p, d.d2e_J.p);
        hipLaunchKernelGGL(d2e_sort_kernel, dim3(div_up(ND, 256)), dim3(256), 0, s2, ND, d.d2e_I.p, d.d2e_J.p);
        SA_HIP_CHECK(hipGetLastError());
        SA_HIP_CHECK(hipEventRecord(ev_join, s2));
    }
    struct JoinSide {      // (every way out of this function waits for the side stream: its buffers are released here)
        hipStream_t s2;
        hipEvent_t a, b;
        ~JoinSide() {
            (void)hipStreamSynchronize(s2);
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
        }
    } join_side{s2, ev_fork, ev_join};
    // ---- AE_to_elem ----
    DBuf<int> ae2e_I((size_t)nparts + 1), ae2e_J((size_t)NE);
    SA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(int) * (size_t)nparts, s));
    hipLaunchKernelGGL(key_count_runs_kernel, dim3(div_up(NE, 256)), dim3(256), 0, s, (long)NE, part_dev, cnt.p);
    hipLaunchKernelGGL(any_zero_kernel, dim3(div_up(nparts, 256)), dim3(256), 0, s, (long)nparts, cnt.p, 4, err.p);
    exclusive_scan_int(s, nparts, cnt.p, ae2e_I.p);
    auto h_ae2e_I = ae2e_I.to_host(s);
    SA_REQUIRE(!(err.to_host(s)[0] & 4), "empty agglomerate");
    int max_ne = 0;
    for (int p = 0; p < nparts; ++p) max_ne = std::max(max_ne, h_ae2e_I[p + 1] - h_ae2e_I[p]);
    if ((long)max_ne * nde > TOPO_MAXK || max_ne > 4096) return false;   // host path handles it
    SA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(int) * (size_t)nparts, s));
    hipLaunchKernelGGL(key_fill_runs_kernel, dim3(div_up(NE, 256)), dim3(256), 0, s, (long)NE, part_dev, ae2e_I.p, cnt.p,
                       ae2e_J.p);
    int cap = 2;
    while (cap < max_ne) cap <<= 1;
    hipLaunchKernelGGL(row_sort_kernel, dim3(nparts), dim3(256), sizeof(int) * (size_t)cap, s, ae2e_I.p, ae2e_J.p, cap);
    // ---- AE_to_dof ----
    const int K = max_ne * nde;
    int HS = 64;
    while (HS < 2 * K) HS <<= 1;
    const size_t lds = sizeof(int) * ((size_t)2 * HS + 2 * (size_t)((K + 255) & ~255));
    DBuf<int> rowcnt((size_t)nparts);
    d.ae2d_I.alloc((size_t)nparts + 1);
    // (one pass: counts + lists into a padded buffer of NE * nde slots, then the packed copy)
    DBuf<int> padded((size_t)NE * nde);
    if (lds > 64 * 1024)
        SA_HIP_CHECK(hipFuncSetAttribute((const void *)ae_to_dof_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipLaunchKernelGGL(ae_to_dof_kernel, dim3(nparts), dim3(256), lds, s, 2, nde, HS, ae2e_I.p, ae2e_J.p, e2d_dev,
                       rowcnt.p, nullptr, padded.p);
    exclusive_scan_int(s, nparts, rowcnt.p, d.ae2d_I.p);
    download(r.AE_to_dof.I, d.ae2d_I, (size_t)nparts + 1, s);
    SA_HIP_CHECK(hipStreamSynchronize(s));
    const long nae2d = r.AE_to_dof.I[nparts];
    d.ae2d_J.alloc((size_t)nae2d);
    hipLaunchKernelGGL(ae_pack_lists_kernel, dim3(nparts), dim3(256), 0, s, nde, ae2e_I.p, d.ae2d_I.p, padded.p, d.ae2d_J.p);
    // ---- dof_to_AE, dof_id_inAE, flags ----
    d.d2ae_I.alloc((size_t)ND + 1);
    d.d2ae_J.alloc((size_t)nae2d);
    d.dof_id_inAE.alloc((size_t)nae2d);
    d.flags.alloc((size_t)ND);
    SA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(int) * (size_t)ND, s));
    hipLaunchKernelGGL(key_count_kernel, dim3(div_up(nae2d, 256)), dim3(256), 0, s, nae2d, d.ae2d_J.p, cnt.p);
    hipLaunchKernelGGL(any_zero_kernel, dim3(div_up(ND, 256)), dim3(256), 0, s, (long)ND, cnt.p, 8, err.p);
    exclusive_scan_int(s, ND, cnt.p, d.d2ae_I.p);
    SA_HIP_CHECK(hipMemsetAsync(cnt.p, 0, sizeof(int) * (size_t)ND, s));
    hipLaunchKernelGGL(d2ae_fill_kernel, dim3(nparts), dim3(256), 0, s, d.ae2d_I.p, d.ae2d_J.p, d.d2ae_I.p, cnt.p,
                       d.d2ae_J.p, d.dof_id_inAE.p);
    hipLaunchKernelGGL(d2ae_sort_flags_kernel, dim3(div_up(ND, 256)), dim3(256), 0, s, ND, d.d2ae_I.p, d.d2ae_J.p,
                       d.dof_id_inAE.p, bdr_dev, d.flags.p);
    SA_HIP_CHECK(hipGetLastError());
    // (host copies of these tables: on demand, fetch_relations_ae_host)
    r.ae_host_pending = true;
    r.AE_to_dof.ncols = ND;
    r.dof_to_AE.ncols = nparts;
    // elem_ldof for the assembly kernels
    d.elem_ldof.alloc((size_t)nconn);
    SA_HIP_CHECK(hipStreamWaitEvent(s, ev_join, 0));
    hipLaunchKernelGGL(elem_ldof_kernel, dim3(div_up(NE, 256)), dim3(256), 0, s, NE, d.e2d_I.p, d.e2d_J.p, d.part.p,
                       d.d2ae_I.p, d.d2ae_J.p, d.dof_id_inAE.p, d.elem_ldof.p);
    SA_HIP_CHECK(hipGetLastError());
    SA_HIP_CHECK(hipStreamSynchronize(s));
    SA_REQUIRE(!(err.to_host(s)[0] & 8), "dof without any element");
    return true;
}

void fetch_relations_ae_host(Relations &r, const DevRelations &d, hipStream_t s) {
    if (!r.ae_host_pending) return;
    const size_t nae2d = (size_t)r.AE_to_dof.I[(size_t)r.nparts];
    download(r.AE_to_dof.J, d.ae2d_J, nae2d, s);
    download(r.dof_to_AE.I, d.d2ae_I, (size_t)r.ND + 1, s);
    download(r.dof_to_AE.J, d.d2ae_J, nae2d, s);
    download(r.dof_id_inAE, d.dof_id_inAE, nae2d, s);
    download(r.agg_flags, d.flags, (size_t)r.ND, s);
    SA_HIP_CHECK(hipStreamSynchronize(s));
    r.ae_host_pending = false;
}

void upload_relations_mis(DevRelations &d, const Relations &r, hipStream_t s) {
    d.mis2d_I.from_host(r.mis_to_dof.I, s);
    d.mis2d_J.from_host(r.mis_to_dof.J, s);
    d.mis2ae_I.from_host(r.mis_to_AE.I, s);
    d.mis2ae_J.from_host(r.mis_to_AE.J, s);
    d.ae2mis_I.from_host(r.AE_to_mis.I, s);
    d.ae2mis_J.from_host(r.AE_to_mis.J, s);
    d.ae_pair.from_host(r.ae_pair, s);
    d.mises.from_host(r.mises, s);
    d.dof_row_in_mis.from_host(r.dof_row_in_mis, s);
    d.pair_loc_off.from_host(r.pair_loc_off, s);
    d.pair_loc.from_host(r.pair_loc, s);
}

}  // namespace saamge_amd
