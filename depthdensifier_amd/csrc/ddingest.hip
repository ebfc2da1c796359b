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
        info.shape[info.ndim++] = v;
        elems *= v;
        s = end;
    }
    info.data_offset = hoff + hlen;
    info.nbytes = elems * size;
    return DD_OK;
}

}  // namespace

extern "C" {

const char *dd_ingest_last_error(void) { return g_ierr; }

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
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return ifail(DD_ERR_INVALID_ARG, "%s: %s", path, strerror(errno));
    NpyInfo info;
    int rc = parse_npy(fd, path, info);
    if (rc == DD_OK) {
        // (a bool map may be asked for as bytes and the other way round: both are one byte per element with values 0 / 1)
        const bool same = info.dtype == expect_dtype || ((info.dtype == DD_NPY_U8 || info.dtype == DD_NPY_BOOL) && (expect_dtype == DD_NPY_U8 || expect_dtype == DD_NPY_BOOL));
        bool ok = same && info.ndim == expect_ndim;
        for (int k = 0; ok && k < info.ndim; ++k) ok = info.shape[k] == expect_shape[k];
        if (!ok) rc = ifail(DD_ERR_INVALID_ARG, "%s: element type or shape differs from what the view needs", path);
        else if (info.nbytes > dst_bytes) rc = ifail(DD_ERR_WORKSPACE, "%s: destination buffer too small", path);
    }
    if (rc == DD_OK) {
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
