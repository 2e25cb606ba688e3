// Native collectives of the multi-GPU path: RCCL over xGMI, one process per GPU, called from the C++
// setup / solve loops and enqueued on the hierarchy's stream -- no Python, no host synchronisation in the
// solve loop.  Reference counterpart: hypre's halo exchange inside every HypreParMatrix::Mult of the
// V-cycle (amg/src/tg.cpp:91-132), MPI_Allreduce of the PCG inner products
// (amg/src/mfem_addons.cpp:106-248), SharedEntityCommunication of the setup
// (amg/inc/SharedEntityCommunication.hpp:74-245).
//
// The three primitives have the signatures of the saamge_amd_params callbacks, so an MPI host code can
// still plug its own (include/saamge_amd.h); saamge_amd_params_set_comm installs these.
//   allgather   in place, variable parts: grouped ncclSend / ncclRecv of every rank's part to every peer
//   allreduce   ncclAllReduce(ncclDouble, ncclSum), in place
//   alltoallv   grouped ncclSend / ncclRecv (the halo exchange: two neighbours per rank for slab partitions;
//               xGMI is point-to-point, one direct link per peer pair)
// librccl is resolved at run time (dlopen): a process that already holds an RCCL (PyTorch-ROCm bundles one
// under the same SONAME) shares it, and the library loads on machines without RCCL as long as no
// communicator is asked for.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <mutex>

#include "../../include/saamge_amd.h"
#include "common.h"

namespace saamge_amd {

namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        // ONE RCCL per process: if any librccl is already mapped (PyTorch-ROCm ships its own copy under its own
        // path and SONAME), that one is used; a second copy in the same process breaks both at teardown
        std::string loaded;
        dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *data) -> int {
            if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) { *(std::string *)data = info->dlpi_name; return 1; }
            return 0;
        }, &loaded);
        void *h = loaded.empty() ? nullptr : dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD);
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        api.lib = h;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
        api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    });
    SA_REQUIRE(api.lib && api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.Send && api.Recv &&
                   api.GroupStart && api.GroupEnd,
               "RCCL (librccl.so.1) could not be loaded");
    return api;
}

#define SA_NCCL_CHECK(expr)                                                                         \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            throw ::saamge_amd::Error(3, std::string(__FILE__) + ":" + std::to_string(__LINE__) + " " + #expr + " -> " + \
                                             (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"));         \
    } while (0)
}  // namespace

}  // namespace saamge_amd

using namespace saamge_amd;

struct saamge_amd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    std::string err;
};

static std::string g_comm_error;

// ---- the three primitives, with the callback signatures of saamge_amd_params ------------------------
static int native_allreduce(void *ctx, double *buf, long long count) {
    saamge_amd_comm *c = (saamge_amd_comm *)ctx;
    try {
        if (count > 0) SA_NCCL_CHECK(rccl().AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
        return 0;
    } catch (const std::exception &e) { c->err = e.what(); std::fprintf(stderr, "saamge_amd: %s\n", e.what()); return 5; }
}
static int native_alltoallv(void *ctx, const void *send, const long long *soff, void *recv, const long long *roff) {
    saamge_amd_comm *c = (saamge_amd_comm *)ctx;
    try {
        RcclApi &api = rccl();
        SA_NCCL_CHECK(api.GroupStart());
        for (int r = 0; r < c->world; ++r) {
            if (r == c->rank) continue;
            const long long ns = soff[r + 1] - soff[r], nr = roff[r + 1] - roff[r];
            if (ns > 0) SA_NCCL_CHECK(api.Send((const char *)send + soff[r], (size_t)ns, ncclChar, r, c->comm, c->stream));
            if (nr > 0) SA_NCCL_CHECK(api.Recv((char *)recv + roff[r], (size_t)nr, ncclChar, r, c->comm, c->stream));
        }
        SA_NCCL_CHECK(api.GroupEnd());
        // (a rank's own part: the library never sends to itself, soff[rank+1] == soff[rank])
        return 0;
    } catch (const std::exception &e) { c->err = e.what(); std::fprintf(stderr, "saamge_amd: %s\n", e.what()); return 5; }
}
static int native_allgather(void *ctx, void *buf, const long long *off) {
    saamge_amd_comm *c = (saamge_amd_comm *)ctx;
    try {
        RcclApi &api = rccl();
        const long long mine = off[c->rank + 1] - off[c->rank];
        SA_NCCL_CHECK(api.GroupStart());
        for (int r = 0; r < c->world; ++r) {
            if (r == c->rank) continue;
            const long long theirs = off[r + 1] - off[r];
            if (mine > 0) SA_NCCL_CHECK(api.Send((const char *)buf + off[c->rank], (size_t)mine, ncclChar, r, c->comm, c->stream));
            if (theirs > 0) SA_NCCL_CHECK(api.Recv((char *)buf + off[r], (size_t)theirs, ncclChar, r, c->comm, c->stream));
        }
        SA_NCCL_CHECK(api.GroupEnd());
        // the setup reads gathered data on the host side of the same stream order; several callers sync anyway
        return 0;
    } catch (const std::exception &e) { c->err = e.what(); std::fprintf(stderr, "saamge_amd: %s\n", e.what()); return 5; }
}

extern "C" {

const char *saamge_amd_comm_last_error(void) { return g_comm_error.c_str(); }

int saamge_amd_comm_unique_id(char id[128]) {
    try {
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId u;
        SA_NCCL_CHECK(rccl().GetUniqueId(&u));
        std::memcpy(id, &u, 128);
        return 0;
    } catch (const std::exception &e) { g_comm_error = e.what(); return 1; }
}

int saamge_amd_comm_create(int rank, int world, const char id[128], void *stream, saamge_amd_comm **out) {
    try {
        SA_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, "bad argument");
        ncclUniqueId u;
        std::memcpy(&u, id, 128);
        saamge_amd_comm *c = new saamge_amd_comm;
        c->rank = rank;
        c->world = world;
        c->stream = (hipStream_t)stream;
        ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) {
            delete c;
            SA_NCCL_CHECK(r);
        }
        *out = c;
        return 0;
    } catch (const std::exception &e) { g_comm_error = e.what(); return 1; }
}

void saamge_amd_comm_destroy(saamge_amd_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
}

// For saamge_amd_ml_produce_data: is `p` wired to a native communicator, and on which stream does that one enqueue?
// (the collectives are stream-ordered with the hierarchy only if it is the hierarchy's stream)
extern "C" int saamge_amd_comm_native_stream(const saamge_amd_params *p, void **stream) {
    if (!p || p->allgather != native_allgather || !p->allgather_ctx) return 0;
    if (stream) *stream = (void *)((saamge_amd_comm *)p->allgather_ctx)->stream;
    return 1;
}

int saamge_amd_params_set_comm(saamge_amd_params *p, saamge_amd_comm *c) {
    if (!p || !c) return 1;
    p->rank = c->rank;
    p->world = c->world;
    p->allgather = native_allgather;
    p->allreduce_sum = native_allreduce;
    p->alltoallv = native_alltoallv;
    p->allgather_ctx = c;
    p->comm_stream_ordered = 1;       // everything is enqueued on the communicator's stream = the hierarchy's
    return 0;
}

// One all-reduce, one all-gather and one all-to-all of known data through the communicator (what the multi-rank
// paths use), checked on the host: returns 0 when every result is what the arithmetic says.
int saamge_amd_comm_selftest(saamge_amd_comm *c) {
    try {
        SA_REQUIRE(c, "bad argument");
        const int W = c->world, R = c->rank;
        hipStream_t s = c->stream;
        // all-reduce: sum over ranks of (rank + 1) * (i + 1)
        std::vector<double> h(4);
        for (int i = 0; i < 4; ++i) h[i] = (R + 1.0) * (i + 1.0);
        DBuf<double> d;
        d.from_host(h, s);
        SA_REQUIRE(native_allreduce(c, d.p, 4) == 0, c->err);
        auto got = d.to_host(s);
        for (int i = 0; i < 4; ++i) SA_REQUIRE(got[i] == 0.5 * W * (W + 1.0) * (i + 1.0), "all-reduce gave a wrong sum");
        // all-gather: rank r owns r + 1 doubles of value r
        std::vector<long long> off((size_t)W + 1, 0);
        for (int r = 0; r < W; ++r) off[(size_t)r + 1] = off[r] + 8ll * (r + 1);
        std::vector<double> g((size_t)off[W] / 8, -1.0);
        for (long long k = off[R] / 8; k < off[R + 1] / 8; ++k) g[(size_t)k] = R;
        DBuf<double> dg;
        dg.from_host(g, s);
        SA_REQUIRE(native_allgather(c, dg.p, off.data()) == 0, c->err);
        auto gg = dg.to_host(s);
        for (int r = 0; r < W; ++r)
            for (long long k = off[r] / 8; k < off[r + 1] / 8; ++k) SA_REQUIRE(gg[(size_t)k] == r, "all-gather gave a wrong part");
        // all-to-all: rank r sends the value 100 r + q to rank q
        std::vector<long long> so((size_t)W + 1, 0), ro((size_t)W + 1, 0);
        for (int r = 0; r < W; ++r) { so[(size_t)r + 1] = so[r] + (r == R ? 0 : 8); ro[(size_t)r + 1] = ro[r] + (r == R ? 0 : 8); }
        std::vector<double> sb((size_t)W + 1, 0.0), rb((size_t)W + 1, -1.0);
        for (int q = 0; q < W; ++q) if (q != R) sb[(size_t)so[q] / 8] = 100.0 * R + q;
        DBuf<double> ds, dr;
        ds.from_host(sb, s);
        dr.from_host(rb, s);
        SA_REQUIRE(native_alltoallv(c, ds.p, so.data(), dr.p, ro.data()) == 0, c->err);
        auto rr = dr.to_host(s);
        for (int q = 0; q < W; ++q) if (q != R) SA_REQUIRE(rr[(size_t)ro[q] / 8] == 100.0 * q + R, "all-to-all gave a wrong entry");
        return 0;
    } catch (const std::exception &e) { g_comm_error = e.what(); return 1; }
}

}  // extern "C"
