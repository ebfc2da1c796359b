// ddrefine.hip -- the per-pixel half of DepthRefiner as ONE kernel, gfx950.
//
// Replaces, for a whole view, the reference's `_apply_transformation` (src/depthdensifier/
// depth_refiner.py:180-205): masked gather -> sorted look-up table with linear interpolation
// (`_pchip_interpolate_optimized`, :141-178) -> scatter -> 3x3 median with replicated borders
// (:194-200) -> zero outside the mask (:203).  As stock tensor ops that is ~40 launches and ~100x
// the compulsory traffic; here a 32x32 tile (+1 halo) is transformed once into LDS, the median is a
// 19-exchange network on LDS values, and the view is read once (depth + mask) and written once.
// Memory-bound: 5 B read + 4 B written per pixel.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ddcore.h"

namespace {

constexpr int TW = 32, TH = 32;            // output tile
constexpr int HW_ = TW + 2, HH_ = TH + 2;  // with halo
constexpr int MAX_LDS_KNOTS = 2048;

struct RArgs {
    const void *depth;
    const uint8_t *mask;
    const float *kx, *ky;
    float *out;
    int H, W, n, f16, skip_smoothing;
};

__device__ __forceinline__ float read_depth(const RArgs &a, int idx) {
    return a.f16 ? (float)reinterpret_cast<const _Float16 *>(a.depth)[idx] : reinterpret_cast<const float *>(a.depth)[idx];
}

// torch.searchsorted(xs, d, right=False) clamped to [1, n-1], then the linear blend of :160-176
__device__ __forceinline__ float lut(const float *kx, const float *ky, int n, float d) {
#pragma clang fp contract(off)   // the reference rounds after the multiply (separate tensor ops): no FMA here
    if (d != d) return d;                   // torch.clamp / torch.maximum propagate a NaN depth (:168-176): the pixel later fails depth > 0
    int lo = 0, hi = n;                     // first index with kx[i] >= d
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kx[mid] < d) lo = mid + 1; else hi = mid;
    }
    int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const float x0 = kx[i - 1], x1 = kx[i], y0 = ky[i - 1], y1 = ky[i];
    float dx = x1 - x0;
    if (dx == 0.0f) dx = 1e-6f;
    float t = (d - x0) / dx;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return fmaxf(y0 + t * (y1 - y0), 1e-3f);
}

#define CSWAP(a, b) { const float lo_ = fminf(v[a], v[b]), hi_ = fmaxf(v[a], v[b]); v[a] = lo_; v[b] = hi_; }

__global__ __launch_bounds__(256) void refine_apply_kernel(const RArgs a) {
    __shared__ float s_val[HH_][HW_ + 1];
    __shared__ float s_kx[MAX_LDS_KNOTS], s_ky[MAX_LDS_KNOTS];
    const int tid = threadIdx.x;
    const bool lds_knots = a.n <= MAX_LDS_KNOTS;
    if (lds_knots)
        for (int i = tid; i < a.n; i += 256) { s_kx[i] = a.kx[i]; s_ky[i] = a.ky[i]; }
    __syncthreads();
    const float *kx = lds_knots ? s_kx : a.kx, *ky = lds_knots ? s_ky : a.ky;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    // transformed values of the tile + halo; replicate padding == clamped coordinates
    for (int i = tid; i < HH_ * HW_; i += 256) {
        const int ly = i / HW_, lx = i - ly * HW_;
        int gy = y0 + ly - 1, gx = x0 + lx - 1;
        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
        const int idx = gy * a.W + gx;
        const float d = read_depth(a, idx);
        const bool m = a.mask ? (a.mask[idx] != 0) : (d > 0.0f);      // :238-241
        s_val[ly][lx] = m ? lut(kx, ky, a.n, d) : 0.0f;                // :185-191
    }
    __syncthreads();
    for (int i = tid; i < TH * TW; i += 256) {
        const int ly = i / TW, lx = i - ly * TW;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= a.H || gx >= a.W) continue;
        float r;
        if (a.skip_smoothing) {
            r = s_val[ly + 1][lx + 1];
        } else {                                                       // median of the 3x3 window (:194-200)
            float v[9];
            bool has_nan = false;                                      // torch.median of a window holding a NaN is NaN
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) { v[dy * 3 + dx] = s_val[ly + dy][lx + dx]; has_nan |= v[dy * 3 + dx] != v[dy * 3 + dx]; }
            CSWAP(1, 2) CSWAP(4, 5) CSWAP(7, 8) CSWAP(0, 1) CSWAP(3, 4) CSWAP(6, 7) CSWAP(1, 2) CSWAP(4, 5) CSWAP(7, 8)
            CSWAP(0, 3) CSWAP(5, 8) CSWAP(4, 7) CSWAP(3, 6) CSWAP(1, 4) CSWAP(2, 5) CSWAP(4, 7) CSWAP(4, 2) CSWAP(6, 4) CSWAP(4, 2)
            r = has_nan ? __builtin_nanf("") : v[4];
        }
        const int idx = gy * a.W + gx;
        const bool m = a.mask ? (a.mask[idx] != 0) : (read_depth(a, idx) > 0.0f);
        a.out[idx] = m ? r : 0.0f;                                     // :203
    }
}

thread_local char g_rerr[192] = "";

}  // namespace

extern "C" {

const char *dd_refine_last_error(void) { return g_rerr; }

int dd_refine_apply(const void *depth, int32_t depth_dtype, const uint8_t *mask, int32_t height, int32_t width,
                    const float *knots_x, const float *knots_y, int32_t n_knots, int32_t skip_smoothing,
                    float *refined_out, void *stream) {
    auto fail = [](const char *m) { snprintf(g_rerr, sizeof(g_rerr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (!depth || !refined_out) return fail("depth / refined_out is NULL");
    if (height <= 0 || width <= 0) return fail("height/width must be positive");
    if (depth_dtype != DD_F32 && depth_dtype != DD_F16) return fail("depth_dtype must be DD_F32 or DD_F16");
    if (!knots_x || !knots_y || n_knots < 2) return fail("need at least two sorted knots");
    if ((long long)height * width >= (1ll << 31)) return fail("view too large");
    RArgs a;
    a.depth = depth; a.mask = mask; a.kx = knots_x; a.ky = knots_y; a.out = refined_out;
    a.H = height; a.W = width; a.n = n_knots; a.f16 = depth_dtype == DD_F16; a.skip_smoothing = skip_smoothing;
    const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH);
    hipLaunchKernelGGL(refine_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { snprintf(g_rerr, sizeof(g_rerr), "refine_apply launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

}  // extern "C"
