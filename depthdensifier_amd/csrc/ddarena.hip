// ddarena.hip -- HBM zone arena: where the cloud's large arrays live in the 288 GB of an MI355X (round 3).
//
// What it is for.  The densify kernel (ddcore.hip; scripts/test.py:203-244, 262-266 fused) writes two large row streams in
// lock step -- the points and the normals, 12 bytes per point each (the two list appends of scripts/test.py:238-240) -- beside
// its read streams.  Measured on MI355X (profiles/r03_placement_*.txt): the physical memory falls into THREE classes of
// about a third of the HBM each; two lock-step write streams inside one class run 20-26 % slower than in two different
// classes (a plain two-stream store kernel: 0.146 vs 0.116 ms per 2 x 384 MiB), which is 4-11 % of the whole densify
// kernel (2.94 vs 2.60 ms on 185 1080p views) and was the unexplained "box state" of rounds 1 and 2: a fresh process gets
// all its memory from one end of the device, i.e. from one class.  The relation is an equivalence (pairwise table in
// profiles/r03_zone_probe.txt), so a chunk's class is found by timing it against one anchor chunk per class.
//
// Measured and NOT adopted (profiles/r03_zone_interleave.txt): arrays whose consecutive 32-64 MiB pieces alternate between the
// three classes serve even a SINGLE row-store stream at 7.1 instead of 5.7-6.0 TB/s (128 MiB pieces: 6.6-7.0), which would
// lift the xyz-only workloads as well.  But hipMemMap takes no offset into a handle on this runtime, so every piece must be
// a handle of its own; pieces that small cannot be timed alone, and runs of consecutively created small handles are pure
// only on an untouched device -- after torch has allocated and freed they are mixtures (profiles/
// r03_arena_runs_of_32MiB_pieces_are_mixed_after_torch.txt), and re-mapping scouting ranges piece by piece made
// hipMemSetAccess fail intermittently (round 4 found why: a range that is mapped a second time keeps its first physical
// memory on this stack -- see VA_BLOCK_CHUNKS below).  Handles of 1 GiB always come from ONE class, in every process state seen.
//
// What it does.  Physical memory is taken in chunks (default 1 GiB) through the virtual-memory API (hipMemCreate), every
// new chunk is mapped at a fresh piece of address space and classified with the two-stream store probe below against the anchors;
// an allocation request names a GROUP per array, arrays of different groups are built from chunks of different classes
// (mapped back to back into a fresh virtual range, so the caller sees one contiguous array), chunks that are not needed
// go straight back to the driver.  Nothing here touches results: the arena only chooses physical pages.
//
// C ABI: dd_arena_* in include/ddcore.h.  Host-side memory management; the only kernel is the probe.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>

#include "ddcore.h"
#include "ddarena_plan.h"

namespace {

thread_local char g_aerr[320] = "";

int afail(int code, const char *msg) {
    snprintf(g_aerr, sizeof(g_aerr), "%s", msg);
    return code;
}
int afail_hip(const char *what, hipError_t e) {
    snprintf(g_aerr, sizeof(g_aerr), "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? DD_ERR_WORKSPACE : DD_ERR_LAUNCH;
}

typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef f32x3 f32x3_rows __attribute__((aligned(4)));

// Two row streams written in lock step, one 12-byte row per lane and stream: the store shape of the densify kernel's
// points + normals.  Write-only: the classes show more clearly than with read streams in the mix.
__global__ __launch_bounds__(256) void zone_pair_store(float *a, float *b, const size_t rows) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= rows) return;
    f32x3 p;
    p.x = (float)(unsigned)i; p.y = 1.0f; p.z = 2.0f;
    *reinterpret_cast<f32x3_rows *>(a + 3 * i) = p;
    *reinterpret_cast<f32x3_rows *>(b + 3 * i) = p;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

using ddarena_plan::MAX_CLASSES;
using ddarena_plan::best_assignment;
using ddarena_plan::plan_classes;
constexpr int SCOUT_BATCH = 8;
// NO VIRTUAL RANGE IS EVER MAPPED TWICE.  On this stack (ROCm 7.2, MI355X) a range that was mapped, unmapped and mapped again --
// with or without hipMemAddressFree + hipMemAddressReserve in between, which hands the same address out again -- keeps
// translating to the physical memory of its FIRST mapping: kernels and copies through it read and write the old handle's pages,
// whoever owns them by then (tools/experiments/ubench_vmm_remap.hip, profiles/r04_ubench_vmm_remap.txt: 20 of 20 repetitions; a
// bump pointer through one large reservation: 0 of 30).  Rounds 3 and 4 reused scouting slots and freed array ranges; that was
// right only as long as the chunk that came back to a range was the chunk that had left it, and ended in rare aborts and
// segmentation faults inside the runtime once clouds were allocated and freed by the hundred (the test suite).  So virtual
// addresses are taken from large reservations with a bump pointer and never given back before dd_arena_destroy: a scouted chunk
// costs one chunk of address space, an array its size (47 bits of address space: ~100 000 GiB-sized mappings per process;
// arrays that are freed and asked for again in the same shape come out of the mapping cache below and cost none).
constexpr size_t VA_BLOCK_CHUNKS = 1024;      // chunks per reservation (1 TiB at the default chunk size)

struct VaBlock {
    char *base;
    size_t bytes, used;
};

struct Chunk {
    hipMemGenericAllocationHandle_t h;
    int cls;        // -1 = not classified
    char *scout;    // where it is mapped for scouting, nullptr = mapped into an allocation (or pooled: mapped nowhere)
    bool live;      // handle not yet released
    bool used;      // part of an allocation
    bool anchor;
    bool mixed;     // two classes inside: handed out only when the device is exhausted (as class 0, the allocation reported as not apart),
                    // never pooled as a classified spare -- it goes back to the driver when its array is freed
};

struct Mapping {
    char *va;
    size_t bytes;               // a multiple of the chunk size
    std::vector<int> chunks;
    int32_t layout;             // what the caller asked for (a group, DD_ARENA_ROTATED + phase, DD_ARENA_BLOCKED)
    bool clean;                 // built by a request that had no conflicts: every chunk lies in the class its layout names
};
constexpr size_t CACHE_MAX_CHUNKS_PER_ARRAY = 8;     // arrays up to this many chunks are kept mapped when freed (the mapping cache)

}  // namespace

struct DDArena {
    int device;
    size_t chunk;
    size_t rows;                // probe window, rows of 12 bytes
    std::vector<VaBlock> va;          // reservations, consumed front to back, never reused
    std::vector<Chunk> chunks;
    std::vector<Mapping> maps;
    std::vector<Mapping> cache;       // freed arrays kept as they are -- mapped, their chunks marked used -- for the next request of
                                      // the same shape and layout: a caller that builds a cloud per scene then makes no
                                      // virtual-memory call at all and consumes no address space (oldest first out; dd_arena_trim empties it)
    int anchors[MAX_CLASSES];
    int n_classes;
    int group_class[MAX_CLASSES];     // class a group was given at its first use (-1 = not yet): sticky, so that arrays of
                                      // later calls land relative to the arrays of earlier ones
    float same_ms;              // probe level of two windows inside one chunk (same class by construction)
    float fast_ms;              // fastest pair seen
    hipStream_t stream;
    hipEvent_t e0, e1;
    hipMemAllocationProp prop;
    std::mutex mu;
    // statistics
    int64_t created, released, probes, mixed, cache_hits;
    bool debug;                 // DD_ARENA_DEBUG: one line per classified chunk on stderr
    bool trace;                 // DD_ARENA_TRACE: one line in front of every virtual-memory call of alloc / free on stderr (fault hunting)
    int pool_per_class;         // classified chunks of each class kept (unmapped) when arrays are freed or scouting leaves
                                // spares, so that the next allocation need not scout again; dd_arena_trim gives them back
    double seconds;
    int32_t degraded;           // allocations that could not be given distinct classes
};

namespace {

#define AHIP(call, what)                                   \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return afail_hip(what, e_);  \
    } while (0)

// `bytes` (a multiple of the chunk size) of address space that no mapping has ever used; nullptr = the reservation failed
char *va_take(DDArena *A, size_t bytes) {
    if (A->va.empty() || A->va.back().used + bytes > A->va.back().bytes) {
        VaBlock b;
        b.base = nullptr; b.used = 0;
        b.bytes = std::max(bytes, VA_BLOCK_CHUNKS * A->chunk);
        if (hipMemAddressReserve(reinterpret_cast<void **>(&b.base), b.bytes, 0, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            b.bytes = std::max(bytes, 16 * A->chunk);
            hipError_t e = hipMemAddressReserve(reinterpret_cast<void **>(&b.base), b.bytes, 0, nullptr, 0);
            if (e != hipSuccess) { (void)afail_hip("hipMemAddressReserve", e); return nullptr; }
        }
        A->va.push_back(b);
    }
    VaBlock &b = A->va.back();
    char *p = b.base + b.used;
    b.used += bytes;
    return p;
}

#define ATRACE(A, ...) do { if ((A)->trace) { fprintf(stderr, "[ddarena] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

// median-free timing: the minimum of `reps` runs after one warm-up (disturbances only ever add time)
int probe_pair(DDArena *A, float *a, float *b, float *ms_out) {
    const dim3 grid((unsigned)((A->rows + 255) / 256)), block(256);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        AHIP(hipEventRecord(A->e0, A->stream), "hipEventRecord");
        hipLaunchKernelGGL(zone_pair_store, grid, block, 0, A->stream, a, b, A->rows);
        AHIP(hipGetLastError(), "zone_pair_store launch");
        AHIP(hipEventRecord(A->e1, A->stream), "hipEventRecord");
        AHIP(hipEventSynchronize(A->e1), "hipEventSynchronize");
        float ms = 0.f;
        AHIP(hipEventElapsedTime(&ms, A->e0, A->e1), "hipEventElapsedTime");
        if (r > 0 && ms < best) best = ms;
    }
    A->probes += 1;
    *ms_out = best;
    return DD_OK;
}

// is a pair at `ms` inside one class?  Same-class pairs sit at the level of two windows inside one chunk (same_ms),
// cross-class pairs 20-26 % below it; the cut is the geometric middle once a fast pair has been seen, 9 % below same_ms before.
bool is_same_class(const DDArena *A, float ms) {
    float cut = 0.91f * A->same_ms;
    if (A->fast_ms < 0.9f * A->same_ms) cut = sqrtf(A->fast_ms * A->same_ms);
    return ms > cut;
}

int release_chunk(DDArena *A, int ci) {
    Chunk &c = A->chunks[ci];
    if (!c.live) return DD_OK;
    if (c.scout) {
        AHIP(hipMemUnmap(c.scout, A->chunk), "hipMemUnmap(scout)");
        c.scout = nullptr;
    }
    AHIP(hipMemRelease(c.h), "hipMemRelease");
    c.live = false;
    A->released += 1;
    return DD_OK;
}

// One more physical chunk, mapped at a fresh piece of address space and classified.  DD_ERR_WORKSPACE = the device is out of memory
// (or out of address space): the caller stops scouting.
int scout_one(DDArena *A, int *chunk_out) {
    Chunk c;
    memset(&c, 0, sizeof(c));
    c.cls = -1; c.scout = nullptr; c.live = true; c.used = false; c.anchor = false;
    hipError_t e = hipMemCreate(&c.h, A->chunk, &A->prop, 0);
    if (e != hipSuccess) return afail_hip("hipMemCreate", e);
    c.scout = va_take(A, A->chunk);                 // (address space is taken only once the memory is there)
    if (!c.scout) { (void)hipMemRelease(c.h); return DD_ERR_WORKSPACE; }
    e = hipMemMap(c.scout, A->chunk, 0, c.h, 0);
    if (e != hipSuccess) { (void)hipMemRelease(c.h); return afail_hip("hipMemMap(scout)", e); }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = A->prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(c.scout, A->chunk, &acc, 1);
    if (e != hipSuccess) { (void)hipMemUnmap(c.scout, A->chunk); (void)hipMemRelease(c.h); return afail_hip("hipMemSetAccess(scout)", e); }
    int ci = -1;                                     // entries of released chunks are reused: nothing refers to them any more
    for (size_t i = 0; i < A->chunks.size() && ci < 0; ++i) if (!A->chunks[i].live) ci = (int)i;
    if (ci < 0) { A->chunks.push_back(c); ci = (int)A->chunks.size() - 1; }
    else A->chunks[ci] = c;
    A->created += 1;
    char *const here = c.scout;
    float *w = reinterpret_cast<float *>(here);
    int rc;
    if (A->n_classes == 0) {
        // the very first chunk: anchor of class 0; windows inside it give the same-class level -- IF it lies inside one class.
        // The classes meet at two places of the physical memory that are not chunk-aligned: a first chunk that straddles one
        // would set same_ms to the CROSS-class level, every later pair would look "same class" and every allocation would end up
        // in class 0.  Three windows (first, middle, last) must agree pairwise; a chunk in which they do not is set aside as
        // mixed and the next chunk scouted takes its place.
        float *mid = reinterpret_cast<float *>(here + A->chunk / 2);
        float *last = reinterpret_cast<float *>(here + A->chunk - A->rows * 12);
        float ms, ms_fl, ms_ml;
        if ((rc = probe_pair(A, w, mid, &ms)) != DD_OK) return rc;
        if ((rc = probe_pair(A, w, last, &ms_fl)) != DD_OK) return rc;
        if ((rc = probe_pair(A, mid, last, &ms_ml)) != DD_OK) return rc;
        const float lo = fminf(ms, fminf(ms_fl, ms_ml)), hi = fmaxf(ms, fmaxf(ms_fl, ms_ml));
        if (lo < 0.91f * hi) {
            A->chunks[ci].cls = -1;
            A->chunks[ci].mixed = true;
            A->mixed += 1;
            if (A->debug) fprintf(stderr, "[ddarena] chunk %d: the first, windows %.4f / %.4f / %.4f ms -> two classes inside, not used\n", ci, ms, ms_fl, ms_ml);
            *chunk_out = ci;
            return DD_OK;
        }
        ms = hi;
        A->same_ms = ms;
        A->fast_ms = ms;
        if (A->debug) fprintf(stderr, "[ddarena] chunk %d: the first, its windows %.4f ms -> anchor of class 0\n", ci, ms);
        A->chunks[ci].cls = 0;
        A->chunks[ci].anchor = true;
        A->anchors[0] = ci;
        A->n_classes = 1;
        *chunk_out = ci;
        return DD_OK;
    }
    float t[MAX_CLASSES];
    int cls = -1;
    for (int k = 0; k < A->n_classes; ++k) {
        float *anchor = reinterpret_cast<float *>(A->chunks[A->anchors[k]].scout);
        if ((rc = probe_pair(A, anchor, w, &t[k])) != DD_OK) return rc;
        if (t[k] < A->fast_ms) A->fast_ms = t[k];
    }
    for (int k = 0; k < A->n_classes; ++k)          // decided after all probes: fast_ms may have moved
        if (is_same_class(A, t[k]) && (cls < 0 || t[k] > t[cls])) cls = k;
    // The probes above saw the chunk's first window only.  The classes meet at two places of the physical memory that
    // are not chunk-aligned, so a chunk can change class inside: its first against its last window must be a same-class pair.
    float t_self = 0.f;
    if ((rc = probe_pair(A, w, reinterpret_cast<float *>(here + A->chunk - A->rows * 12), &t_self)) != DD_OK) return rc;
    if (!is_same_class(A, t_self)) {             // a verdict that costs the chunk: timed once more (a clock step during the probe reads the same way)
        float again = 0.f;
        if ((rc = probe_pair(A, w, reinterpret_cast<float *>(here + A->chunk - A->rows * 12), &again)) != DD_OK) return rc;
        t_self = fmaxf(t_self, again);
    }
    if (!is_same_class(A, t_self)) {
        A->chunks[ci].cls = -1;                      // mixed: never handed out, goes back to the driver with the other spares
        A->chunks[ci].mixed = true;
        A->mixed += 1;
        if (A->debug) fprintf(stderr, "[ddarena] chunk %d: first vs last window %.4f ms -> two classes inside, not used\n", ci, t_self);
        *chunk_out = ci;
        return DD_OK;
    }
    if (cls < 0) {
        if (A->n_classes < MAX_CLASSES) {            // differs from every anchor: a new class, this chunk anchors it
            cls = A->n_classes++;
            A->anchors[cls] = ci;
            A->chunks[ci].anchor = true;
        } else {                                     // cannot be: three classes are all there is.  Take the slowest pairing.
            cls = 0;
            for (int k = 1; k < MAX_CLASSES; ++k) if (t[k] > t[cls]) cls = k;
        }
    }
    A->chunks[ci].cls = cls;
    if (A->debug) {
        fprintf(stderr, "[ddarena] chunk %d:", ci);
        for (int k = 0; k < A->n_classes; ++k) if (A->anchors[k] != ci) fprintf(stderr, " vs anchor %d %.4f ms", k, t[k]);
        fprintf(stderr, " -> class %d%s   (same %.4f, fast %.4f)\n", cls, A->chunks[ci].anchor ? " (its anchor)" : "", A->same_ms, A->fast_ms);
    }
    *chunk_out = ci;
    return DD_OK;
}

// a chunk nobody uses: into the pool (unmapped, its class remembered) while the pool of its class has room, else to the driver
int pool_or_release(DDArena *A, int ci) {
    Chunk &c = A->chunks[ci];
    if (!c.live || c.used || c.anchor) return DD_OK;
    int pooled = 0;
    for (const Chunk &o : A->chunks) if (&o != &c && o.live && !o.used && !o.anchor && !o.scout && o.cls == c.cls) ++pooled;
    if (c.cls >= 0 && !c.mixed && pooled < A->pool_per_class) {
        if (c.scout) {
            AHIP(hipMemUnmap(c.scout, A->chunk), "hipMemUnmap(scout)");
            c.scout = nullptr;
        }
        return DD_OK;
    }
    return release_chunk(A, ci);
}

// a cached mapping goes: unmapped for good, its chunks to the pool or the driver
int drop_cached(DDArena *A, size_t k) {
    Mapping m = A->cache[k];
    A->cache.erase(A->cache.begin() + (long)k);
    ATRACE(A, "cache: %p leaves (%zu bytes)", (void *)m.va, m.bytes);
    AHIP(hipMemUnmap(m.va, m.bytes), "hipMemUnmap(array)");
    for (int ci : m.chunks) {
        A->chunks[ci].used = false;
        int rc = pool_or_release(A, ci);
        if (rc != DD_OK) return rc;
    }
    return DD_OK;
}

int free_count(const DDArena *A, int cls) {
    int n = 0;
    for (const Chunk &c : A->chunks) if (c.live && !c.used && !c.anchor && c.cls == cls) ++n;
    return n;
}

// the layout planning (which class every chunk of every array comes from) is plain host logic: ddarena_plan.h, compiled
// into tests/c_client/arena_plan_test.cpp by the CPU suite as well
void free_counts(const DDArena *A, int avail[MAX_CLASSES]) {
    for (int c = 0; c < MAX_CLASSES; ++c) avail[c] = free_count(A, c);
}

int take_chunk(DDArena *A, int cls) {      // a free chunk of the class; -1 = none
    for (size_t i = 0; i < A->chunks.size(); ++i) {
        Chunk &c = A->chunks[i];
        if (c.live && !c.used && !c.anchor && c.cls == cls) return (int)i;
    }
    return -1;
}

}  // namespace

extern "C" {

const char *dd_arena_last_error(void) { return g_aerr; }

int dd_arena_create(int32_t device, int64_t chunk_bytes, DDArena **arena_out) {
    if (!arena_out) return afail(DD_ERR_INVALID_ARG, "arena_out is NULL");
    *arena_out = nullptr;
    if (chunk_bytes == 0) chunk_bytes = (int64_t)1 << 30;
    if (chunk_bytes < ((int64_t)64 << 20) || (chunk_bytes & (((int64_t)2 << 20) - 1)))
        return afail(DD_ERR_INVALID_ARG, "chunk_bytes must be a multiple of 2 MiB and at least 64 MiB");
    int ndev = 0;
    AHIP(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
    if (device < 0 || device >= ndev) return afail(DD_ERR_INVALID_ARG, "no such device");
    int prev = 0;
    AHIP(hipGetDevice(&prev), "hipGetDevice");
    AHIP(hipSetDevice(device), "hipSetDevice");
    DDArena *A = new DDArena();
    A->device = device;
    A->chunk = (size_t)chunk_bytes;
    A->rows = std::min<size_t>((size_t)32 << 20, A->chunk / 24);      // two windows fit one chunk (the same-class reference)
    A->n_classes = 0;
    A->same_ms = A->fast_ms = 0.f;
    A->created = A->released = A->probes = A->mixed = A->cache_hits = 0;
    A->debug = getenv("DD_ARENA_DEBUG") != nullptr;
    A->trace = getenv("DD_ARENA_TRACE") != nullptr;
    A->pool_per_class = 4;
    A->seconds = 0.0;
    A->degraded = 0;
    for (int k = 0; k < MAX_CLASSES; ++k) A->anchors[k] = A->group_class[k] = -1;
    memset(&A->prop, 0, sizeof(A->prop));
    A->prop.type = hipMemAllocationTypePinned;
    A->prop.location.type = hipMemLocationTypeDevice;
    A->prop.location.id = device;
    int rc = DD_OK;
    hipError_t e;
    A->stream = nullptr; A->e0 = A->e1 = nullptr;
    if (!va_take(A, A->chunk)) rc = DD_ERR_LAUNCH;          // the first reservation: fails here where the virtual-memory API is missing
    if (rc == DD_OK && (e = hipStreamCreateWithFlags(&A->stream, hipStreamNonBlocking)) != hipSuccess) rc = afail_hip("hipStreamCreate", e);
    if (rc == DD_OK && (e = hipEventCreate(&A->e0)) != hipSuccess) rc = afail_hip("hipEventCreate", e);
    if (rc == DD_OK && (e = hipEventCreate(&A->e1)) != hipSuccess) rc = afail_hip("hipEventCreate", e);
    (void)hipSetDevice(prev);
    if (rc != DD_OK) {
        if (A->e0) (void)hipEventDestroy(A->e0);
        if (A->e1) (void)hipEventDestroy(A->e1);
        if (A->stream) (void)hipStreamDestroy(A->stream);
        for (VaBlock &b : A->va) (void)hipMemAddressFree(b.base, b.bytes);
        delete A;
        return rc;
    }
    *arena_out = A;
    return DD_OK;
}

int dd_arena_alloc(DDArena *A, int32_t n, const int64_t *sizes, const int32_t *groups, int64_t max_scout_bytes, void **ptrs_out) {
    if (!A || !sizes || !groups || !ptrs_out || n <= 0 || n > 64) return afail(DD_ERR_INVALID_ARG, "arena / sizes / groups / ptrs_out is NULL or n outside 1..64");
    std::lock_guard<std::mutex> lock(A->mu);
    const double t_start = now_s();
    int need[MAX_CLASSES] = {0, 0, 0}, fixed[MAX_CLASSES] = {0, 0, 0};
    std::vector<int> nch(n);
    for (int i = 0; i < n; ++i) {
        if (sizes[i] <= 0) return afail(DD_ERR_INVALID_ARG, "sizes must be positive");
        const int L = groups[i];
        const bool pure = L >= 0 && L < MAX_CLASSES, rotated = L >= DD_ARENA_ROTATED && L < DD_ARENA_ROTATED + MAX_CLASSES, blocked = L == DD_ARENA_BLOCKED;
        if (!pure && !rotated && !blocked) return afail(DD_ERR_INVALID_ARG, "layouts must be a group 0..2, DD_ARENA_ROTATED + phase 0..2 or DD_ARENA_BLOCKED");
        nch[i] = (int)(((size_t)sizes[i] + A->chunk - 1) / A->chunk);
        if (pure) need[L] += nch[i];
        else if (blocked) for (int k = 0; k < nch[i]; ++k) fixed[std::min(MAX_CLASSES - 1, k * MAX_CLASSES / nch[i])] += 1;
        else for (int k = 0; k < nch[i]; ++k) fixed[(L - DD_ARENA_ROTATED + k) % MAX_CLASSES] += 1;
        ptrs_out[i] = nullptr;
    }
    {   // the mapping cache: every array of the request as it was freed before (same size in chunks, same layout, built cleanly)
        std::vector<int> hit(n, -1);
        std::vector<char> taken(A->cache.size(), 0);
        bool all = true;
        for (int i = 0; i < n && all; ++i) {
            for (size_t k = 0; k < A->cache.size() && hit[i] < 0; ++k)
                if (!taken[k] && A->cache[k].clean && A->cache[k].layout == groups[i] && A->cache[k].chunks.size() == (size_t)nch[i]) { hit[i] = (int)k; taken[k] = 1; }
            all = hit[i] >= 0;
        }
        if (all) {
            for (int i = 0; i < n; ++i) { A->maps.push_back(A->cache[hit[i]]); ptrs_out[i] = A->cache[hit[i]].va; }
            std::vector<Mapping> rest;
            for (size_t k = 0; k < A->cache.size(); ++k) if (!taken[k]) rest.push_back(A->cache[k]);
            A->cache.swap(rest);
            A->cache_hits += 1;
            A->seconds += now_s() - t_start;
            return DD_OK;
        }
    }
    int prev = 0;
    AHIP(hipGetDevice(&prev), "hipGetDevice");
    AHIP(hipSetDevice(A->device), "hipSetDevice");
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    AHIP(hipDeviceSynchronize(), "hipDeviceSynchronize");          // probes are timed: nothing else should be running
    while (!A->cache.empty()) {                                     // a request of another shape: what the cache holds becomes spare chunks
        int rc0 = drop_cached(A, 0);
        if (rc0 != DD_OK) return rc0;
    }
    const int total_need = need[0] + need[1] + need[2] + fixed[0] + fixed[1] + fixed[2];
    int perm[MAX_CLASSES] = {0, 1, 2};
    bool oom = false;
    int rc = DD_OK;
    std::vector<std::vector<int>> choice;
    int missing = 0, conflicts = 0;
    // scout until every array can be laid out without two lock-step arrays sharing a class, or the budget / the memory is spent
    for (;;) {
        int avail[MAX_CLASSES];
        free_counts(A, avail);
        best_assignment(avail, A->group_class, need, fixed, perm);
        plan_classes(avail, n, nch, groups, perm, choice, &missing, &conflicts);
        if (oom && missing > 0) {
            // the device is exhausted and the request still cannot be served: chunks set aside as "two classes inside" are memory too
            // (the arrays built from them are reported as not apart)
            bool any = false;
            for (Chunk &c : A->chunks) if (c.live && !c.used && !c.anchor && c.cls < 0) { c.cls = 0; any = true; }
            if (any) {
                free_counts(A, avail);
                best_assignment(avail, A->group_class, need, fixed, perm);
                plan_classes(avail, n, nch, groups, perm, choice, &missing, &conflicts);
                conflicts += 1;
            }
        }
        if ((missing == 0 && conflicts == 0) || oom) break;
        int live_free = 0;
        for (const Chunk &c : A->chunks) if (c.live && !c.used && !c.anchor && c.cls >= 0) ++live_free;
        // the budget counts chunks beyond what the request itself needs
        if ((int64_t)(live_free - total_need + SCOUT_BATCH) * (int64_t)A->chunk > max_scout_bytes && live_free >= total_need) break;
        for (int b = 0; b < SCOUT_BATCH; ++b) {
            int ci;
            rc = scout_one(A, &ci);
            if (rc == DD_ERR_WORKSPACE) { oom = true; rc = DD_OK; break; }
            if (rc != DD_OK) return rc;
        }
    }
    for (int g = 0; g < MAX_CLASSES; ++g) if (need[g] > 0) A->group_class[g] = perm[g];
    // take the chunks the plan names
    std::vector<std::vector<int>> chosen(n);
    const bool degraded = conflicts > 0;
    for (int i = 0; i < n && rc == DD_OK; ++i) {
        for (int k = 0; k < nch[i]; ++k) {
            const int ci = choice[i][k] >= 0 ? take_chunk(A, choice[i][k]) : -1;
            if (ci < 0) {
                int avail[MAX_CLASSES];
                free_counts(A, avail);
                snprintf(g_aerr, sizeof(g_aerr), "arena: out of device memory (%d chunks wanted, %d + %d + %d free after scouting; %lld created, %lld released, %lld mixed so far)",
                         total_need, avail[0], avail[1], avail[2], (long long)A->created, (long long)A->released, (long long)A->mixed);
                rc = DD_ERR_WORKSPACE;
                break;
            }
            A->chunks[ci].used = true;
            chosen[i].push_back(ci);
        }
    }
    if (rc != DD_OK) {
        for (auto &v : chosen) for (int ci : v) A->chunks[ci].used = false;
        for (size_t ci = 0; ci < A->chunks.size(); ++ci) if (A->chunks[ci].live && !A->chunks[ci].used && !A->chunks[ci].anchor) (void)release_chunk(A, (int)ci);
        return rc;
    }
    // build the arrays: chunks leave the addresses they were scouted at and are mapped back to back at addresses no mapping has used.  On any failure
    // everything this request has built is undone ("nothing stays allocated").
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = A->prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<Mapping> built;
    for (int i = 0; i < n && rc == DD_OK; ++i) {
        Mapping m;
        m.bytes = (size_t)nch[i] * A->chunk;
        m.va = nullptr;
        hipError_t e = hipSuccess;
        m.va = va_take(A, m.bytes);
        if (!m.va) { rc = DD_ERR_WORKSPACE; break; }
        ATRACE(A, "alloc: %p taken, mapping %d chunks", (void *)m.va, nch[i]);
        int mapped = 0;
        for (int k = 0; k < nch[i] && rc == DD_OK; ++k) {
            Chunk &c = A->chunks[chosen[i][k]];
            if (c.scout) {                  // fresh from scouting (a pooled chunk is not mapped anywhere)
                if ((e = hipMemUnmap(c.scout, A->chunk)) != hipSuccess) { rc = afail_hip("hipMemUnmap(scout)", e); break; }
                c.scout = nullptr;
            }
            if ((e = hipMemMap(m.va + (size_t)k * A->chunk, A->chunk, 0, c.h, 0)) != hipSuccess) { rc = afail_hip("hipMemMap(array)", e); break; }
            ++mapped;
        }
        ATRACE(A, "alloc: %p set access", (void *)m.va);
        if (rc == DD_OK && (e = hipMemSetAccess(m.va, m.bytes, &acc, 1)) != hipSuccess) rc = afail_hip("hipMemSetAccess(array)", e);
        ATRACE(A, "alloc: %p ready", (void *)m.va);
        if (rc != DD_OK) {
            if (mapped > 0) (void)hipMemUnmap(m.va, (size_t)mapped * A->chunk);
            break;
        }
        m.chunks = chosen[i];
        m.layout = groups[i];
        m.clean = !degraded;
        built.push_back(m);
    }
    if (rc != DD_OK) {
        char msg[sizeof(g_aerr)];
        snprintf(msg, sizeof(msg), "%s", g_aerr);
        for (Mapping &m : built) (void)hipMemUnmap(m.va, m.bytes);
        for (auto &v : chosen) for (int ci : v) A->chunks[ci].used = false;
        for (size_t ci = 0; ci < A->chunks.size(); ++ci) if (A->chunks[ci].live && !A->chunks[ci].used && !A->chunks[ci].anchor) (void)release_chunk(A, (int)ci);
        (void)hipGetLastError();
        snprintf(g_aerr, sizeof(g_aerr), "%s", msg);
        return rc;
    }
    for (int i = 0; i < n; ++i) {
        A->maps.push_back(built[i]);
        ptrs_out[i] = built[i].va;
    }
    // what was scouted and not needed goes back to the driver, but for a few spares per class (the anchors stay)
    // (the arrays are built and handed out from here on: a failure to unmap or release a SPARE chunk is not the caller's problem --
    // returning it would make the caller drop arrays that stay mapped; the chunk simply stays with the arena until the next trim)
    for (size_t ci = 0; ci < A->chunks.size(); ++ci)
        if (A->chunks[ci].live && !A->chunks[ci].used && !A->chunks[ci].anchor) {
            if (pool_or_release(A, (int)ci) != DD_OK) (void)hipGetLastError();
        }
    if (degraded) A->degraded += 1;
    A->seconds += now_s() - t_start;
    return degraded ? 1 : DD_OK;
}

int dd_arena_free(DDArena *A, void *ptr) {
    if (!A || !ptr) return afail(DD_ERR_INVALID_ARG, "arena or ptr is NULL");
    std::lock_guard<std::mutex> lock(A->mu);
    for (size_t i = 0; i < A->maps.size(); ++i) {
        if (A->maps[i].va != ptr) continue;
        int prev = 0;
        AHIP(hipGetDevice(&prev), "hipGetDevice");
        AHIP(hipSetDevice(A->device), "hipSetDevice");
        struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
        ATRACE(A, "free %p: sync", ptr);
        AHIP(hipDeviceSynchronize(), "hipDeviceSynchronize");      // kernels that still use the array must have drained
        Mapping m = A->maps[i];
        A->maps.erase(A->maps.begin() + (long)i);
        if (m.clean && m.chunks.size() <= CACHE_MAX_CHUNKS_PER_ARRAY && A->pool_per_class > 0) {
            ATRACE(A, "free %p: kept mapped (cache)", ptr);
            A->cache.push_back(m);
            size_t held = 0;
            for (const Mapping &c : A->cache) held += c.chunks.size();
            while (held > (size_t)(MAX_CLASSES * A->pool_per_class) && A->cache.size() > 1) {      // oldest first out
                held -= A->cache.front().chunks.size();
                int rc = drop_cached(A, 0);
                if (rc != DD_OK) return rc;
            }
            return DD_OK;
        }
        ATRACE(A, "free %p: unmap %zu bytes", ptr, m.bytes);
        AHIP(hipMemUnmap(m.va, m.bytes), "hipMemUnmap(array)");
        for (int ci : m.chunks) {
            A->chunks[ci].used = false;
            ATRACE(A, "free %p: chunk %d to the pool or the driver", ptr, ci);
            int rc = pool_or_release(A, ci);
            if (rc != DD_OK) return rc;
        }
        ATRACE(A, "free %p: done (its addresses are never used again)", ptr);
        return DD_OK;
    }
    return afail(DD_ERR_INVALID_ARG, "ptr was not allocated by this arena");
}

int dd_arena_set_pool(DDArena *A, int32_t pool_chunks_per_class) {
    if (!A) return afail(DD_ERR_INVALID_ARG, "arena is NULL");
    if (pool_chunks_per_class < 0) return afail(DD_ERR_INVALID_ARG, "pool_chunks_per_class is negative");
    std::lock_guard<std::mutex> lock(A->mu);
    A->pool_per_class = pool_chunks_per_class;      // (nothing is released, nothing is unmapped: only what happens to chunks freed from now on)
    return DD_OK;
}

int dd_arena_trim(DDArena *A, int32_t pool_chunks_per_class) {
    if (!A) return afail(DD_ERR_INVALID_ARG, "arena is NULL");
    std::lock_guard<std::mutex> lock(A->mu);
    if (pool_chunks_per_class >= 0) A->pool_per_class = pool_chunks_per_class;
    int prev = 0;
    AHIP(hipGetDevice(&prev), "hipGetDevice");
    AHIP(hipSetDevice(A->device), "hipSetDevice");
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    const int keep = A->pool_per_class;
    A->pool_per_class = 0;                     // (what leaves the cache goes to the driver, not to the pool)
    while (!A->cache.empty()) {
        int rc = drop_cached(A, A->cache.size() - 1);
        if (rc != DD_OK) { A->pool_per_class = keep; return rc; }
    }
    A->pool_per_class = keep;
    for (size_t ci = 0; ci < A->chunks.size(); ++ci) {
        Chunk &c = A->chunks[ci];
        if (c.live && !c.used && !c.anchor) {
            int rc = release_chunk(A, (int)ci);
            if (rc != DD_OK) return rc;
        }
    }
    return DD_OK;
}

int dd_arena_classes(DDArena *A, const void *ptr, int32_t *classes_out, int32_t capacity) {
    if (!A || !ptr || !classes_out) return afail(DD_ERR_INVALID_ARG, "arena / ptr / classes_out is NULL");
    std::lock_guard<std::mutex> lock(A->mu);
    for (const Mapping &m : A->maps) {
        if (m.va != ptr) continue;
        const int k = (int)m.chunks.size();
        for (int i = 0; i < k && i < capacity; ++i) classes_out[i] = A->chunks[m.chunks[i]].cls;
        return k;
    }
    return afail(DD_ERR_INVALID_ARG, "ptr was not allocated by this arena");
}

int dd_arena_probe(DDArena *A, void *a, void *b, float *ms_out) {
    if (!A || !a || !b || !ms_out) return afail(DD_ERR_INVALID_ARG, "arena / a / b / ms_out is NULL");
    std::lock_guard<std::mutex> lock(A->mu);
    int prev = 0;
    AHIP(hipGetDevice(&prev), "hipGetDevice");
    AHIP(hipSetDevice(A->device), "hipSetDevice");
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    AHIP(hipDeviceSynchronize(), "hipDeviceSynchronize");
    return probe_pair(A, reinterpret_cast<float *>(a), reinterpret_cast<float *>(b), ms_out);
}

int dd_arena_stats(DDArena *A, DDArenaStats *out) {
    if (!A || !out) return afail(DD_ERR_INVALID_ARG, "arena or out is NULL");
    std::lock_guard<std::mutex> lock(A->mu);
    memset(out, 0, sizeof(*out));
    out->chunk_bytes = (int64_t)A->chunk;
    out->probe_bytes = (int64_t)A->rows * 12;
    out->num_classes = A->n_classes;
    out->chunks_created = A->created;
    out->chunks_released = A->released;
    out->probes = A->probes;
    out->same_class_ms = A->same_ms;
    out->cross_class_ms = A->fast_ms;
    out->seconds = A->seconds;
    out->degraded_allocs = A->degraded;
    for (const Chunk &c : A->chunks)
        if (c.live && c.cls >= 0 && c.cls < MAX_CLASSES) {
            out->chunks_held[c.cls] += 1;
            if (!c.used && !c.anchor) out->chunks_pooled[c.cls] += 1;
        }
    for (const Mapping &m : A->cache)              // arrays kept mapped for the next request of their shape: spare capacity as well
        for (int ci : m.chunks) { const int cls = A->chunks[ci].cls; if (cls >= 0 && cls < MAX_CLASSES) out->chunks_pooled[cls] += 1; }
    return DD_OK;
}

int dd_arena_destroy(DDArena *A) {
    if (!A) return DD_OK;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(A->device);
    (void)hipDeviceSynchronize();
    for (Mapping &m : A->maps) (void)hipMemUnmap(m.va, m.bytes);
    for (Mapping &m : A->cache) (void)hipMemUnmap(m.va, m.bytes);
    for (Chunk &c : A->chunks) {
        if (!c.live) continue;
        if (c.scout) (void)hipMemUnmap(c.scout, A->chunk);
        (void)hipMemRelease(c.h);
    }
    for (VaBlock &b : A->va) (void)hipMemAddressFree(b.base, b.bytes);
    (void)hipEventDestroy(A->e0);
    (void)hipEventDestroy(A->e1);
    (void)hipStreamDestroy(A->stream);
    (void)hipSetDevice(prev);
    delete A;
    return DD_OK;
}

}  // extern "C"
