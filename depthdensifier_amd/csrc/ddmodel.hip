// ddmodel.hip -- the dense points as COLMAP points3D.bin records, formatted on the device.
//
// Replaces, for the dense cloud, the per-point Python loop `rec.add_point3D(xyz, Track(), color)` of
// scripts/test.py:355-358 and the serialisation of those points by write_binary (:363).  At full density the model file
// is the largest object of a run (51 bytes per point: 16.6 GB for a 185-view 1080p scan, 170 GB for 2000 views), and
// building it on the host needs the float64 copies and the record array in host memory at once.  Here a point's record is
// composed in LDS and leaves the GPU in its final byte layout, chunk by chunk; the host only moves bytes
// (depthdensifier_amd/model_writer.py: pinned double buffers, device->host copies overlapped with the file writes).
//
// Record (COLMAP's public binary layout, little-endian, no padding): point3D_id uint64 | xyz 3 x float64 | rgb 3 x uint8 |
// error float64 | track length uint64 (+ no track elements).  New points carry error = -1 and an empty track, like
// pycolmap's add_point3D.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ddcore.h"

namespace {

thread_local char g_merr[256] = "";
constexpr int REC = 51;           // bytes per record
constexpr int MT = 256;           // points per workgroup: 256 * 51 = 13056 bytes = 816 x 16 -> every workgroup starts 16-byte aligned

__device__ __forceinline__ void put64(unsigned char *p, const unsigned long long v) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = (unsigned char)(v >> (8 * k));
}

__global__ __launch_bounds__(MT) void format_points3d_kernel(const float *xyz, const uint8_t *rgb, const uint32_t *packed,
                                                             const long long n, const unsigned long long first_id,
                                                             unsigned char *out) {
    __shared__ __attribute__((aligned(16))) unsigned char sb[MT * REC];
    const long long base = (long long)blockIdx.x * MT;
    const long long i = base + threadIdx.x;
    if (i < n) {
        float x, y, z;
        unsigned c = 0;
        if (packed) {
            const uint4 r = *reinterpret_cast<const uint4 *>(packed + 4 * i);
            x = __uint_as_float(r.x); y = __uint_as_float(r.y); z = __uint_as_float(r.z); c = r.w;
        } else {
            x = xyz[3 * i]; y = xyz[3 * i + 1]; z = xyz[3 * i + 2];
            if (rgb) c = (unsigned)rgb[3 * i] | ((unsigned)rgb[3 * i + 1] << 8) | ((unsigned)rgb[3 * i + 2] << 16);
        }
        unsigned char *p = sb + threadIdx.x * REC;
        put64(p, first_id + (unsigned long long)i);
        put64(p + 8, (unsigned long long)__double_as_longlong((double)x));       // float32 -> float64 is exact
        put64(p + 16, (unsigned long long)__double_as_longlong((double)y));
        put64(p + 24, (unsigned long long)__double_as_longlong((double)z));
        p[32] = (unsigned char)c; p[33] = (unsigned char)(c >> 8); p[34] = (unsigned char)(c >> 16);
        put64(p + 35, 0xBFF0000000000000ull);                                     // error = -1.0
        put64(p + 43, 0ull);                                                      // empty track
    }
    __syncthreads();
    const long long left = n - base;
    const int bytes = (int)(left < MT ? left : MT) * REC;
    unsigned char *dst = out + base * REC;
    const int n16 = bytes >> 4;
    for (int k = threadIdx.x; k < n16; k += MT) reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(sb)[k];
    for (int k = (n16 << 4) + threadIdx.x; k < bytes; k += MT) dst[k] = sb[k];
}

}  // namespace

extern "C" {

const char *dd_model_last_error(void) { return g_merr; }

int dd_format_points3d(const float *xyz, const uint8_t *rgb, const uint32_t *xyz_rgba, int64_t n, uint64_t first_id,
                       uint8_t *out, void *stream) {
    auto fail = [](const char *m) { snprintf(g_merr, sizeof(g_merr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (n < 0) return fail("n is negative");
    if (n == 0) return DD_OK;
    if (n > ((int64_t)1 << 48)) return fail("too many points for one launch; split the call");
    if (!xyz && !xyz_rgba) return fail("xyz and xyz_rgba are both NULL");
    if (xyz_rgba && ((uintptr_t)xyz_rgba % 16) != 0) return fail("xyz_rgba must be 16-byte aligned");
    if (!out || ((uintptr_t)out % 16) != 0) return fail("out is NULL or not 16-byte aligned");
    const long long blocks = (n + MT - 1) / MT;
    if (blocks > 0x7fffffffll) return fail("too many points for one launch; split the call");
    hipLaunchKernelGGL(format_points3d_kernel, dim3((unsigned)blocks), dim3(MT), 0, (hipStream_t)stream, xyz, rgb, xyz_rgba,
                       (long long)n, (unsigned long long)first_id, out);
    if (hipGetLastError() != hipSuccess) { snprintf(g_merr, sizeof(g_merr), "format_points3d launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

}  // extern "C"
