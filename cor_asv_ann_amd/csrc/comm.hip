// Multi-GPU leg of the C ABI: lines shard with no data-path collective; what crosses the GPUs is
// ONE all-gather of fixed-width result records per batch, RCCL over xGMI.  RCCL is opened at the first casv_comm_* call
// (dlopen), not linked: a process that never shards pays nothing, and a host program that brings its own RCCL (PyTorch
// does) does not get a second copy mapped by this library.
#include "engine.h"
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only; the library itself is opened at run time

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail(CASV_ERR_STATE, "RCCL not found (librccl.so.1): %s", dlerror());
#define SYM(field, name) *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return fail(CASV_ERR_STATE, "RCCL symbol %s missing", name);
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce") SYM(CommDestroy, "ncclCommDestroy") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.lib = h;
    return 0;
}
#define NCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(CASV_ERR_HIP, "%s failed: %s", #x, g_rccl.GetErrorString(r_)); } while (0)

}  // namespace

extern "C" int casv_comm_unique_id(void* out) {
    if (!out) return fail(CASV_ERR_ARG, "null argument");
    if (int rc = load_rccl()) return rc;
    NCHK(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(out)));
    return CASV_OK;
}

extern "C" int casv_comm_init(casv_model* m, int32_t rank, int32_t world, const void* unique_id) {
    if (!m || !unique_id) return fail(CASV_ERR_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(CASV_ERR_ARG, "rank %d of %d", rank, world);
    if (m->comm) return fail(CASV_ERR_STATE, "communicator already initialised");
    if (int rc = load_rccl()) return rc;
    HIPCHK(hipSetDevice(m->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    NCHK(g_rccl.CommInitRank(&comm, world, id, rank));
    m->comm = comm;
    m->comm_rank = rank; m->comm_world = world;
    return CASV_OK;
}

extern "C" int casv_comm_all_gather(casv_model* m, const void* send, void* recv, int64_t bytes_per_rank) {
    if (!m || !send || !recv) return fail(CASV_ERR_ARG, "null argument");
    if (!m->comm) return fail(CASV_ERR_STATE, "casv_comm_init first");
    if (bytes_per_rank < 0) return fail(CASV_ERR_ARG, "negative size");
    if (bytes_per_rank == 0) return CASV_OK;
    HIPCHK(hipSetDevice(m->device));
    const size_t n = (size_t)bytes_per_rank;
    if (int rc = m->comm_send.ensure(n)) return rc;
    if (int rc = m->comm_recv.ensure(n * m->comm_world)) return rc;
    HIPCHK(hipMemcpyAsync(m->comm_send.p, send, n, hipMemcpyHostToDevice, m->stream));
    NCHK(g_rccl.AllGather(m->comm_send.p, m->comm_recv.p, n, ncclChar, reinterpret_cast<ncclComm_t>(m->comm), m->stream));
    HIPCHK(hipMemcpyAsync(recv, m->comm_recv.p, n * m->comm_world, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return CASV_OK;
}

// The all-gather of SURVEY.md section 8e straight from the device: every rank's record buffer (casv_records_reset /
// casv_records_append: packed by a kernel where the decode results lie) -> all ranks' records on the host, rank-major.
// No host-side packing and no host-to-device copy in front of the collective.
extern "C" int casv_comm_all_gather_records(casv_model* m, int32_t* recv) {
    if (!m || !recv) return fail(CASV_ERR_ARG, "null argument");
    if (!m->comm) return fail(CASV_ERR_STATE, "casv_comm_init first");
    if (!m->rec.p || !m->rec_rows) return fail(CASV_ERR_STATE, "casv_records_reset first");
    HIPCHK(hipSetDevice(m->device));
    const size_t n = (size_t)m->rec_rows * (2 * m->rec_S + 4) * 4;
    {   // every rank must bring the same record shape: a mismatch would hang or corrupt the collective -- check it first
        // (max of (rows, S, -rows, -S) over the ranks: all equal iff max == -max of the negatives)
        double shape[4] = {(double)m->rec_rows, (double)m->rec_S, -(double)m->rec_rows, -(double)m->rec_S};
        if (int rc = m->comm_send.ensure(sizeof(shape))) return rc;
        HIPCHK(hipMemcpyAsync(m->comm_send.p, shape, sizeof(shape), hipMemcpyHostToDevice, m->stream));
        NCHK(g_rccl.AllReduce(m->comm_send.p, m->comm_send.p, 4, ncclFloat64, ncclMax, reinterpret_cast<ncclComm_t>(m->comm), m->stream));
        HIPCHK(hipMemcpyAsync(shape, m->comm_send.p, sizeof(shape), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hipStreamSynchronize(m->stream));
        if (shape[0] != -shape[2] || shape[1] != -shape[3])
            return fail(CASV_ERR_STATE, "the ranks' record buffers differ in shape (casv_records_reset with the same rows and steps on every rank)");
    }
    if (int rc = m->comm_recv.ensure(n * m->comm_world)) return rc;
    NCHK(g_rccl.AllGather(m->rec.p, m->comm_recv.p, n, ncclChar, reinterpret_cast<ncclComm_t>(m->comm), m->stream));
    HIPCHK(hipMemcpyAsync(recv, m->comm_recv.p, n * m->comm_world, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return CASV_OK;
}

extern "C" int casv_comm_all_reduce_max(casv_model* m, double* value) {
    if (!m || !value) return fail(CASV_ERR_ARG, "null argument");
    if (!m->comm) return fail(CASV_ERR_STATE, "casv_comm_init first");
    HIPCHK(hipSetDevice(m->device));
    if (int rc = m->comm_send.ensure(8)) return rc;
    HIPCHK(hipMemcpyAsync(m->comm_send.p, value, 8, hipMemcpyHostToDevice, m->stream));
    NCHK(g_rccl.AllReduce(m->comm_send.p, m->comm_send.p, 1, ncclFloat64, ncclMax, reinterpret_cast<ncclComm_t>(m->comm), m->stream));
    HIPCHK(hipMemcpyAsync(value, m->comm_send.p, 8, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));      // also a barrier across the ranks
    return CASV_OK;
}

extern "C" int casv_comm_destroy(casv_model* m) {
    if (!m) return fail(CASV_ERR_ARG, "null argument");
    if (!m->comm) return CASV_OK;
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    NCHK(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(m->comm)));
    m->comm = nullptr;
    m->comm_send.release(); m->comm_recv.release();
    return CASV_OK;
}
