// ddcomm.hip -- the multi-GPU fuse of the C ABI: an in-place all-gatherv of the per-GPU compacted clouds over an RCCL
// communicator the CALLER owns (replaces, across GPUs, the np.concatenate of scripts/test.py:262-266).
//
// RCCL has no gatherv and xGMI is point-to-point (one link per peer): every rank posts, inside ONE ncclGroupStart /
// ncclGroupEnd, a send of its rows to each peer and a receive of each peer's rows straight into their final rows of the
// caller's global buffers.  No staging, no padding, nothing is copied locally -- the densify kernel has already written
// this rank's rows at rank_rows[rank] (dd_unproject_compact with *cursor_dev = rank_rows[rank]).
//
// libddcore.so does NOT link librccl: the entry points are resolved at first use from the RCCL runtime already loaded
// in the process (the one that created the caller's communicator), falling back to dlopen("librccl.so.1").  A process
// that never calls dd_allgatherv never touches RCCL.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>

#include "ddcore.h"

namespace {

thread_local char g_cerr[256] = "";

struct Rccl {
    ncclResult_t (*group_start)();
    ncclResult_t (*group_end)();
    ncclResult_t (*send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char *(*error_string)(ncclResult_t);
    bool ok;
};

const Rccl &rccl() {
    static const Rccl r = [] {
        Rccl x{};
        void *h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclSend")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        }
        x.group_start = reinterpret_cast<decltype(x.group_start)>(dlsym(h, "ncclGroupStart"));
        x.group_end = reinterpret_cast<decltype(x.group_end)>(dlsym(h, "ncclGroupEnd"));
        x.send = reinterpret_cast<decltype(x.send)>(dlsym(h, "ncclSend"));
        x.recv = reinterpret_cast<decltype(x.recv)>(dlsym(h, "ncclRecv"));
        x.error_string = reinterpret_cast<decltype(x.error_string)>(dlsym(h, "ncclGetErrorString"));
        x.ok = x.group_start && x.group_end && x.send && x.recv;
        return x;
    }();
    return r;
}

int cfail(int code, const char *msg) {
    snprintf(g_cerr, sizeof(g_cerr), "%s", msg);
    return code;
}

}  // namespace

extern "C" {

const char *dd_comm_last_error(void) { return g_cerr; }

int dd_allgatherv(void *comm, int32_t rank, int32_t world_size, const DDCloudOut *cloud, const int64_t *rank_rows,
                  int32_t dst, void *stream) {
    if (world_size < 1 || rank < 0 || rank >= world_size) return cfail(DD_ERR_INVALID_ARG, "rank / world_size out of range");
    if (dst < -1 || dst >= world_size) return cfail(DD_ERR_INVALID_ARG, "dst must be -1 (every rank receives) or a rank");
    if (!cloud || !rank_rows) return cfail(DD_ERR_INVALID_ARG, "cloud / rank_rows is NULL");
    if (rank_rows[0] < 0) return cfail(DD_ERR_INVALID_ARG, "rank_rows must start at a non-negative row");
    for (int r = 0; r < world_size; ++r)
        if (rank_rows[r + 1] < rank_rows[r]) return cfail(DD_ERR_INVALID_ARG, "rank_rows must be non-decreasing");
    const bool receives = dst < 0 || dst == rank;
    if (receives && cloud->capacity < rank_rows[world_size]) return cfail(DD_ERR_INVALID_ARG, "cloud->capacity is smaller than the fused cloud");
    if (world_size == 1) return DD_OK;
    if (!comm) return cfail(DD_ERR_INVALID_ARG, "comm is NULL");
    const Rccl &nc = rccl();
    if (!nc.ok) return cfail(DD_ERR_UNSUPPORTED, "RCCL is not loaded in this process and librccl.so.1 could not be opened");

    struct Field { void *base; size_t row_bytes; };
    const Field fields[] = {{cloud->xyz, 12}, {cloud->normal, 12}, {cloud->rgb, 3}, {cloud->pixel_index, 4},
                            {cloud->view_index, 4}, {cloud->xyz_rgba, 16}};
    ncclComm_t c = reinterpret_cast<ncclComm_t>(comm);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t own_lo = rank_rows[rank], own_n = rank_rows[rank + 1] - own_lo;
    // a pure sender of a gather-to-owner exchange holds only its own rows: its buffers start at its first row
    const int64_t base = receives ? 0 : own_lo;
    ncclResult_t rc = nc.group_start();
    for (int k = 1; k < world_size && rc == ncclSuccess; ++k) {      // peer order staggered per rank: no hot receiver
        const int to = (rank + k) % world_size, from = (rank - k + world_size) % world_size;
        for (const Field &f : fields) {
            if (!f.base || rc != ncclSuccess) continue;
            char *p = reinterpret_cast<char *>(f.base);
            if (own_n > 0 && (dst < 0 || to == dst))
                rc = nc.send(p + (size_t)(own_lo - base) * f.row_bytes, (size_t)own_n * f.row_bytes, ncclUint8, to, c, s);
            const int64_t lo = rank_rows[from], n = rank_rows[from + 1] - lo;
            if (rc == ncclSuccess && n > 0 && receives)
                rc = nc.recv(p + (size_t)lo * f.row_bytes, (size_t)n * f.row_bytes, ncclUint8, from, c, s);
        }
    }
    const ncclResult_t rc_end = nc.group_end();
    if (rc == ncclSuccess) rc = rc_end;
    if (rc != ncclSuccess) {
        snprintf(g_cerr, sizeof(g_cerr), "RCCL: %s", nc.error_string ? nc.error_string(rc) : "error");
        return DD_ERR_LAUNCH;
    }
    return DD_OK;
}

}  // extern "C"
