// ddingest.hip -- the host side of getting a view's maps into HBM (scripts/test.py:143-168, where the
// reference decodes the image, runs MoGe and pulls depth / normal / mask to the host).  With precomputed maps
// (depthdensifier_amd/depth_source.py, CachedSource) a 1080p view is 41 MB in four .npy files, and the loop around the densify
// kernel was bound by the Python that moved them: sixteen prefetch threads parsing headers and copying arrays under the
// interpreter lock made every call of the main thread wait for its turn (round 6: 1.0-1.3 ms per view inside begin_refine
// alone with 16 threads, 0.25 ms with none; profiles/r06_bench_pipeline.txt).  These entry points do that work without the
// interpreter -- ctypes releases its lock for the duration of a foreign call:
//   dd_npy_read       one .npy file, validated against what the caller expects, read straight into a caller buffer (a page-locked
//                     staging slot): open / pread / close, no allocation, no copy through a temporary;
//   dd_upload_async   n host -> device copies and an event behind them with ONE call.
#include <hip/hip_runtime.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ddcore.h"

namespace {

thread_local char g_ierr[320] = "";

int ifail(int code, const char *fmt, const char *a = "", const char *b = "") {
    snprintf(g_ierr, sizeof(g_ierr), fmt, a, b);
    return code;
}

struct NpyInfo {
    int dtype;              // DD_NPY_*
    int ndim;
    long long shape[4];
    long long data_offset;
    long long nbytes;
};

const char *find_key(const char *hdr, const char *key) {
    const char *p = strstr(hdr, key);
    if (!p) return nullptr;
    p = strchr(p + strlen(key), ':');
    return p ? p + 1 : nullptr;
}

// the header of a version 1.0 / 2.0 / 3.0 .npy file: {'descr': '<f4', 'fortran_order': False, 'shape': (1080, 1920), }
int parse_npy(int fd, const char *path, NpyInfo &info) {
    unsigned char pre[12];
    if (pread(fd, pre, 12, 0) != 12 || memcmp(pre, "\x93NUMPY", 6) != 0) return ifail(DD_ERR_INVALID_ARG, "%s: not a .npy file", path);
    const int major = pre[6];
    long long hlen, hoff;
    if (major == 1) { hlen = pre[8] | (pre[9] << 8); hoff = 10; }
    else if (major == 2 || major == 3) { hlen = (long long)pre[8] | ((long long)pre[9] << 8) | ((long long)pre[10] << 16) | ((long long)pre[11] << 24); hoff = 12; }
    else return ifail(DD_ERR_UNSUPPORTED, "%s: unknown .npy version", path);
    if (hlen <= 0 || hlen > 4096) return ifail(DD_ERR_UNSUPPORTED, "%s: .npy header too long", path);
    char hdr[4097];
    if (pread(fd, hdr, (size_t)hlen, hoff) != hlen) return ifail(DD_ERR_INVALID_ARG, "%s: truncated .npy header", path);
    hdr[hlen] = 0;
    const char *d = find_key(hdr, "'descr'");
    if (!d) return ifail(DD_ERR_INVALID_ARG, "%s: .npy header without 'descr'", path);
    while (*d == ' ') ++d;
    int size;
    if (!strncmp(d, "'<f4'", 5)) { info.dtype = DD_NPY_F32; size = 4; }
    else if (!strncmp(d, "'<f2'", 5)) { info.dtype = DD_NPY_F16; size = 2; }
    else if (!strncmp(d, "'|u1'", 5)) { info.dtype = DD_NPY_U8; size = 1; }
    else if (!strncmp(d, "'|b1'", 5)) { info.dtype = DD_NPY_BOOL; size = 1; }
    else return ifail(DD_ERR_UNSUPPORTED, "%s: element type not one of <f4, <f2, |u1, |b1", path);
    const char *f = find_key(hdr, "'fortran_order'");
    if (!f) return ifail(DD_ERR_INVALID_ARG, "%s: .npy header without 'fortran_order'", path);
    while (*f == ' ') ++f;
    if (strncmp(f, "False", 5) != 0) return ifail(DD_ERR_UNSUPPORTED, "%s: Fortran-ordered arrays are not supported", path);
    const char *s = find_key(hdr, "'shape'");
    if (!s) return ifail(DD_ERR_INVALID_ARG, "%s: .npy header without 'shape'", path);
    s = strchr(s, '(');
    if (!s) return ifail(DD_ERR_INVALID_ARG, "%s: malformed shape", path);
    ++s;
    info.ndim = 0;
    long long elems = 1;
    for (;;) {
        while (*s == ' ' || *s == ',') ++s;
        if (*s == ')' || *s == 0) break;
        if (info.ndim == 4) return ifail(DD_ERR_UNSUPPORTED, "%s: more than 4 dimensions", path);
        char *end;
        const long long v = strtoll(s, &end, 10);
        if (end == s || v < 0) return ifail(DD_ERR_INVALID_ARG, "%s: malformed shape", path);
        if (v > ((long long)1 << 40) || (v > 0 && elems > ((long long)1 << 48) / v)) return ifail(DD_ERR_UNSUPPORTED, "%s: array too large", path);
        info.shape[info.ndim++] = v;
        elems *= v;
        s = end;
    }
    info.data_offset = hoff + hlen;
    info.nbytes = elems * size;
    return DD_OK;
}

// one .npy file into dst: validate, then pread until done.  expect_dtype < 0 = any of the four element types; the type found -> *found
int read_npy_into(const char *path, int expect_dtype, int expect_ndim, const long long *expect_shape, void *dst, long long dst_bytes, int *found) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return ifail(DD_ERR_INVALID_ARG, "%s: %s", path, strerror(errno));
    NpyInfo info;
    int rc = parse_npy(fd, path, info);
    if (rc == DD_OK) {
        // (a bool map may be asked for as bytes and the other way round: both are one byte per element with values 0 / 1)
        const bool bytes = info.dtype == DD_NPY_U8 || info.dtype == DD_NPY_BOOL;
        const bool same = expect_dtype < 0 || info.dtype == expect_dtype || (bytes && (expect_dtype == DD_NPY_U8 || expect_dtype == DD_NPY_BOOL));
        bool ok = same && info.ndim == expect_ndim;
        for (int k = 0; ok && k < info.ndim; ++k) ok = info.shape[k] == expect_shape[k];
        if (!ok) rc = ifail(DD_ERR_INVALID_ARG, "%s: element type or shape differs from what the view needs", path);
        else if (info.nbytes > dst_bytes) rc = ifail(DD_ERR_WORKSPACE, "%s: destination buffer too small", path);
    }
    if (rc == DD_OK) {
        if (found) *found = info.dtype;
        char *out = reinterpret_cast<char *>(dst);
        long long done = 0;
        while (done < info.nbytes) {
            const ssize_t got = pread(fd, out + done, (size_t)(info.nbytes - done), info.data_offset + done);
            if (got < 0 && errno == EINTR) continue;
            if (got <= 0) { rc = ifail(DD_ERR_INVALID_ARG, "%s: %s", path, got == 0 ? "file shorter than its header says" : strerror(errno)); break; }
            done += got;
        }
    }
    close(fd);
    return rc;
}

}  // namespace

// ==================================================================================================
// The prefetcher: native threads that read the .npy files of the views to come into page-locked slots while the main thread enqueues
// the current view -- the job of the pipeline's Python thread pool, without Python.  (Sixteen Python prefetch threads took the
// interpreter lock whenever the main thread let go of it -- at every tensor or library call -- and the loop around the kernels spent
// 0.6-1.3 ms per view waiting for its turn; with native threads nobody else wants the lock.)  Job t uses slot t mod slots; a slot is
// refilled once its previous job has been released and the event given with the release (recorded by the caller behind its last
// read of the slot, e.g. by dd_upload_async) has passed.  Page-locked memory is taken lazily, by the worker that first fills a slot.
// ==================================================================================================
struct DDPrefetcher {
    struct File { std::string path; int dtype, ndim; long long shape[4]; long long offset; int found; };
    struct Job {
        long long ticket = -1;
        int state = 0;                  // 0 free (released), 1 queued, 2 running, 3 done
        int status = DD_OK;
        std::string error;
        std::vector<File> files;
        void *release_event = nullptr;  // of the slot's PREVIOUS job, synchronised before the slot is written again
    };
    int device = 0;
    int nslots = 0;
    long long slot_bytes = 0;
    std::vector<void *> slot_mem;
    std::vector<char> slot_pinned;     // 1 = from hipHostMalloc (freed with hipHostFree), 0 = from malloc
    std::vector<Job> jobs;              // by slot
    long long next_ticket = 0, next_to_run = 0;
    bool stop = false;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::thread> workers;

    void run() {
        (void)hipSetDevice(device);
        for (;;) {
            long long t;
            int slot;
            void *ev;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return stop || (next_to_run < next_ticket && jobs[next_to_run % nslots].ticket == next_to_run && jobs[next_to_run % nslots].state == 1); });
                if (stop) return;
                t = next_to_run++;
                slot = (int)(t % nslots);
                jobs[slot].state = 2;
                ev = jobs[slot].release_event;
                jobs[slot].release_event = nullptr;
            }
            int status = DD_OK;
            std::string error;
            if (ev && hipEventSynchronize((hipEvent_t)ev) != hipSuccess) (void)hipGetLastError();      // (an event of a dead stream: nothing reads the slot any more)
            if (!slot_mem[slot]) {
                void *mem = nullptr;
                slot_pinned[slot] = 1;
                if (hipHostMalloc(&mem, (size_t)slot_bytes, hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    mem = malloc((size_t)slot_bytes);            // ordinary memory: still correct, the upload is then the driver's pageable copy
                    slot_pinned[slot] = 0;
                }
                if (!mem) { status = DD_ERR_WORKSPACE; error = "out of host memory for a staging slot"; }
                slot_mem[slot] = mem;
            }
            if (status == DD_OK) {
                std::vector<File> &files = jobs[slot].files;      // (ours alone while the job is running)
                for (File &f : files) {
                    if (f.offset < 0 || f.offset > slot_bytes) { status = DD_ERR_INVALID_ARG; error = "file offset outside the slot"; break; }
                    status = read_npy_into(f.path.c_str(), f.dtype, f.ndim, f.shape, reinterpret_cast<char *>(slot_mem[slot]) + f.offset, slot_bytes - f.offset, &f.found);
                    if (status != DD_OK) { error = g_ierr; break; }
                }
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                jobs[slot].status = status;
                jobs[slot].error = error;
                jobs[slot].state = 3;
            }
            cv.notify_all();
        }
    }
};

extern "C" {

const char *dd_ingest_last_error(void) { return g_ierr; }

int dd_prefetch_create(int32_t threads, int32_t slots, int64_t slot_bytes, DDPrefetcher **out) {
    if (!out) return ifail(DD_ERR_INVALID_ARG, "out is NULL");
    if (threads < 1 || threads > 256 || slots < 1 || slots > 4096 || slot_bytes < 1) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_create: threads 1..256, slots 1..4096, slot_bytes >= 1");
    DDPrefetcher *p = new DDPrefetcher();
    if (hipGetDevice(&p->device) != hipSuccess) { (void)hipGetLastError(); p->device = 0; }
    p->nslots = slots; p->slot_bytes = slot_bytes;
    p->slot_mem.assign((size_t)slots, nullptr);
    p->slot_pinned.assign((size_t)slots, 0);
    p->jobs.resize((size_t)slots);
    for (int i = 0; i < threads; ++i) p->workers.emplace_back([p] { p->run(); });
    *out = p;
    return DD_OK;
}

int64_t dd_prefetch_submit(DDPrefetcher *p, int32_t n, const char *const *paths, const int32_t *expect_dtype, const int32_t *ndim,
                           const int64_t *shapes, const int64_t *offsets) {
    if (!p || n < 1 || n > 8 || !paths || !expect_dtype || !ndim || !shapes || !offsets) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_submit: NULL argument or n outside 1..8");
    std::unique_lock<std::mutex> lock(p->mu);
    const long long t = p->next_ticket;
    DDPrefetcher::Job &job = p->jobs[t % p->nslots];
    if (job.state != 0) return ifail(DD_ERR_WORKSPACE, "dd_prefetch_submit: every slot holds a job that has not been released (submit at most `slots` jobs ahead)");
    job.files.clear();
    for (int i = 0; i < n; ++i) {
        if (!paths[i] || ndim[i] < 1 || ndim[i] > 4) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_submit: NULL path or ndim outside 1..4");
        DDPrefetcher::File f;
        f.path = paths[i]; f.dtype = expect_dtype[i]; f.ndim = ndim[i]; f.offset = offsets[i]; f.found = -1;
        for (int k = 0; k < 4; ++k) f.shape[k] = k < ndim[i] ? shapes[4 * i + k] : 1;
        job.files.push_back(f);
    }
    job.ticket = t; job.state = 1; job.status = DD_OK; job.error.clear();
    p->next_ticket = t + 1;
    lock.unlock();
    p->cv.notify_all();
    return t;
}

int dd_prefetch_wait(DDPrefetcher *p, int64_t ticket, void **base_out, int32_t *dtypes_out) {
    if (!p) return ifail(DD_ERR_INVALID_ARG, "prefetcher is NULL");
    std::unique_lock<std::mutex> lock(p->mu);
    if (ticket < 0 || ticket >= p->next_ticket) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_wait: no such ticket");
    DDPrefetcher::Job &job = p->jobs[ticket % p->nslots];
    if (job.ticket != ticket) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_wait: the ticket's slot has been given to a later job");
    p->cv.wait(lock, [&] { return job.ticket != ticket || job.state == 3 || job.state == 0 || p->stop; });
    if (job.state != 3 || job.ticket != ticket) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_wait: the job was released or the prefetcher is shutting down");
    if (base_out) *base_out = p->slot_mem[ticket % p->nslots];
    if (dtypes_out) for (size_t i = 0; i < job.files.size(); ++i) dtypes_out[i] = job.files[i].found;
    if (job.status != DD_OK) snprintf(g_ierr, sizeof(g_ierr), "%s", job.error.c_str());
    return job.status;
}

int dd_prefetch_release(DDPrefetcher *p, int64_t ticket, void *event) {
    if (!p) return ifail(DD_ERR_INVALID_ARG, "prefetcher is NULL");
    {
        std::lock_guard<std::mutex> lock(p->mu);
        if (ticket < 0 || ticket >= p->next_ticket) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_release: no such ticket");
        DDPrefetcher::Job &job = p->jobs[ticket % p->nslots];
        if (job.ticket != ticket || job.state != 3) return ifail(DD_ERR_INVALID_ARG, "dd_prefetch_release: the job is not finished (dd_prefetch_wait first) or was released already");
        job.state = 0;
        job.release_event = event;
    }
    p->cv.notify_all();
    return DD_OK;
}

int dd_prefetch_destroy(DDPrefetcher *p) {
    if (!p) return DD_OK;
    {
        std::lock_guard<std::mutex> lock(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    for (std::thread &w : p->workers) w.join();
    for (size_t i = 0; i < p->slot_mem.size(); ++i) {
        void *m = p->slot_mem[i];
        if (!m) continue;
        if (!p->slot_pinned[i]) free(m);
        else if (hipHostFree(m) != hipSuccess) (void)hipGetLastError();      // (a runtime that is shutting down: the memory goes with the process)
    }
    delete p;
    return DD_OK;
}

int dd_npy_header(const char *path, int32_t *dtype_out, int32_t *ndim_out, int64_t *shape_out, int64_t *data_offset_out) {
    if (!path) return ifail(DD_ERR_INVALID_ARG, "path is NULL");
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return ifail(DD_ERR_INVALID_ARG, "%s: %s", path, strerror(errno));
    NpyInfo info;
    const int rc = parse_npy(fd, path, info);
    close(fd);
    if (rc != DD_OK) return rc;
    if (dtype_out) *dtype_out = info.dtype;
    if (ndim_out) *ndim_out = info.ndim;
    if (shape_out) for (int k = 0; k < info.ndim; ++k) shape_out[k] = info.shape[k];
    if (data_offset_out) *data_offset_out = info.data_offset;
    return DD_OK;
}

int dd_npy_read(const char *path, int32_t expect_dtype, int32_t expect_ndim, const int64_t *expect_shape, void *dst, int64_t dst_bytes) {
    if (!path || !dst || !expect_shape) return ifail(DD_ERR_INVALID_ARG, "path / dst / expect_shape is NULL");
    if (expect_ndim < 1 || expect_ndim > 4) return ifail(DD_ERR_INVALID_ARG, "expect_ndim outside 1..4");
    long long shape[4] = {1, 1, 1, 1};
    for (int k = 0; k < expect_ndim; ++k) shape[k] = expect_shape[k];
    return read_npy_into(path, expect_dtype, expect_ndim, shape, dst, dst_bytes, nullptr);
}

int dd_upload_async(int32_t n, const void *const *src_host, void *const *dst_dev, const int64_t *nbytes, void *event, void *stream) {
    if (n < 0 || (n > 0 && (!src_host || !dst_dev || !nbytes))) return ifail(DD_ERR_INVALID_ARG, "dd_upload_async: NULL array");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n; ++i) {
        if (nbytes[i] == 0) continue;
        if (!src_host[i] || !dst_dev[i] || nbytes[i] < 0) return ifail(DD_ERR_INVALID_ARG, "dd_upload_async: NULL pointer or negative size");
        if (hipMemcpyAsync(dst_dev[i], src_host[i], (size_t)nbytes[i], hipMemcpyHostToDevice, s) != hipSuccess) {
            (void)hipGetLastError();
            return ifail(DD_ERR_LAUNCH, "dd_upload_async: hipMemcpyAsync failed");
        }
    }
    if (event && hipEventRecord((hipEvent_t)event, s) != hipSuccess) {
        (void)hipGetLastError();
        return ifail(DD_ERR_LAUNCH, "dd_upload_async: hipEventRecord failed");
    }
    return DD_OK;
}

}  // extern "C"
