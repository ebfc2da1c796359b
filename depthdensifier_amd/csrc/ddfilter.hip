// ddfilter.hip -- multi-view floater votes, MI355X / gfx950.
//
// Replaces the O(N*V) NumPy loop of scripts/test.py:273-328 (+ project_points :58-76): every fused
// point is projected into every cached view; it collects a vote when it lies clearly in front of
// that view's refined depth (z_proj < depth_threshold * d_view), is inside the image, in front of
// the camera and not seen at a grazing angle (dot(n, -dir) > 0.087).
//
// One lane per point, views in the inner loop: the camera block of a view is wave-uniform (scalar
// loads), the point stays in registers, the only memory traffic per (point, view) pair is one
// depth-map lookup -- and neighbouring lanes are neighbouring pixels of their source view, so the
// lookups of a wave land close together.  All decisions are taken in float64, like the reference
// (NumPy promotes everything to float64 there except `depth_threshold * depth`, a float32
// product under NEP 50), so the votes are reproduced exactly; cheap tests run first (behind the
// camera, outside the image) and most pairs never reach the square root.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ddcore.h"

namespace {

struct FArgs {
    const float *xyz;
    const float *normal;
    const float *depth;
    const uint8_t *mask;
    const double *cams;     // (V, 24)
    int32_t *votes;
    long long n;
    long long hw;
    int V, H, W;
    int accumulate;
    float depth_threshold;
    double grazing_cos;
};

__global__ __launch_bounds__(256) void floater_votes_kernel(const FArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
    const double nx = a.normal[3 * i], ny = a.normal[3 * i + 1], nz = a.normal[3 * i + 2];
    const double wlim = (double)a.W, hlim = (double)a.H;
    int votes = a.accumulate ? a.votes[i] : 0;
    for (int v = 0; v < a.V; ++v) {
        const double *c = a.cams + (size_t)v * 24;          // wave-uniform: scalar loads
        // scripts/test.py:63-68  camera-frame point and depth
        const double zc = c[8] * x + c[9] * y + c[10] * z + c[11];
        if (!(zc > 0.0)) continue;                           // :301 depths > 0
        const double xc = c[0] * x + c[1] * y + c[2] * z + c[3];
        const double yc = c[4] * x + c[5] * y + c[6] * z + c[7];
        // :71-75  normalise by (depth + 1e-8), apply K
        const double den = zc + 1e-8;
        const double xn = xc / den, yn = yc / den, zn = zc / den;
        const double u = c[12] * xn + c[13] * yn + c[14] * zn;
        const double w = c[15] * xn + c[16] * yn + c[17] * zn;
        if (!(u >= 0.0 && u < wlim && w >= 0.0 && w < hlim)) continue;      // :300-302
        // :284-295  grazing-angle test against the direction camera centre -> point
        double dx = x - c[18], dy = y - c[19], dz = z - c[20];
        const double len = sqrt(dx * dx + dy * dy + dz * dz);
        dx /= len; dy /= len; dz /= len;
        const double facing = nx * -dx + ny * -dy + nz * -dz;
        if (!(facing > a.grazing_cos)) continue;
        // :308-312  truncating lookup of the view's (mask-zeroed) refined depth
        const long long pix = (long long)v * a.hw + (long long)(int)w * a.W + (int)u;
        float seen = a.depth[pix];
        if (a.mask && a.mask[pix] == 0) seen = 0.0f;        // :194 refined_depth[~mask] = 0
        if (!(seen > 0.0f)) continue;                        // :315
        const float limit = a.depth_threshold * seen;        // :320 float32 product (NEP 50)
        if (zc < (double)limit) ++votes;                     // :319-328
    }
    a.votes[i] = votes;
}

thread_local char g_ferr[192] = "";

}  // namespace

extern "C" {

const char *dd_filter_last_error(void) { return g_ferr; }

int dd_floater_votes(const DDFilterViews *views, const float *xyz, const float *normal, int64_t n,
                     int32_t *votes_dev, int32_t accumulate, void *stream) {
    auto fail = [](const char *m) { snprintf(g_ferr, sizeof(g_ferr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (!views) return fail("views is NULL");
    if (views->num_views <= 0 || views->height <= 0 || views->width <= 0) return fail("num_views/height/width must be positive");
    if (!views->depth || !views->cams) return fail("depth / cams is NULL");
    if (n < 0) return fail("n is negative");
    if (n > 0 && (!xyz || !normal || !votes_dev)) return fail("xyz / normal / votes_dev is NULL");
    if (n == 0) return DD_OK;
    FArgs a;
    a.xyz = xyz; a.normal = normal; a.depth = views->depth; a.mask = views->mask; a.cams = views->cams;
    a.votes = votes_dev; a.n = n; a.hw = (long long)views->height * views->width;
    a.V = views->num_views; a.H = views->height; a.W = views->width; a.accumulate = accumulate;
    a.depth_threshold = views->depth_threshold; a.grazing_cos = views->grazing_cos;
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffll) return fail("too many points for one launch; split the call");
    hipLaunchKernelGGL(floater_votes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { snprintf(g_ferr, sizeof(g_ferr), "floater_votes launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

}  // extern "C"
