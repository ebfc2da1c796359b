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
#include <stdlib.h>

#include "ddcore.h"
#include "ddcore_lab.h"
#include "ddrefine_math.h"

__attribute__((visibility("hidden"))) uint32_t dd_lab_word();      // ddcore.hip: this thread's experiment switches (not exported)

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

using ddmath::lut;

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
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = s_val[ly + dy][lx + dx];
            r = ddmath::median9(v);
        }
        const int idx = gy * a.W + gx;
        const bool m = a.mask ? (a.mask[idx] != 0) : (read_depth(a, idx) > 0.0f);
        a.out[idx] = m ? r : 0.0f;                                     // :203
    }
}

// ---- round 5: the same map, 2x as fast.  The kernel above is bound by the vector ALU and by latencies, not by its 9 bytes per
// pixel (40 us per 1080p view = 0.06 of the HBM peak): every 32x32 tile loads the knots again, every value bisects all of them, every
// window is sorted on its own.  Here a workgroup keeps the knots AND a grid of buckets over them (ddrefine_math.h) for several 64x16
// tiles, takes four values per lane and step with their (short) bisections in lock step, and sorts the six columns that four
// consecutive windows share once.  Same bits: the interval of the curve is the one `lut` finds, and the median selects one of its
// inputs either way (tests/test_refiner.py compares both kernels with the tensor formulation).  More than 2048 knots, or
// DD_REFINE_APPLY_PLAIN=1 in the environment (A/B): the kernel above. ----
constexpr int PW = 64, PH = 16;              // output tile
constexpr int PHW = PW + 2, PHH = PH + 2;    // with halo
constexpr int PSTRIDE = PHW + 1;             // LDS row stride (67 words: rows start in different banks)

__global__ __launch_bounds__(256) void refine_apply_tiles(const RArgs a, const int tiles_x, const int tiles) {
    __shared__ float s_val[PHH * PSTRIDE];
    extern __shared__ float2 s_knots[];                // 3 n floats: n knots {x, y}, then the n reciprocal interval widths -- the workgroup's
    float2 *const s_kxy = s_knots;                     // LDS follows the number of knots (more workgroups per CU)
    float *const s_inv = reinterpret_cast<float *>(s_knots + a.n);
    __shared__ unsigned short s_grid[ddmath::CURVE_GRID_WORDS];
    const int tid = threadIdx.x;
    const ddmath::Curve curve = ddmath::curve_build(s_kxy, s_inv, s_grid, a.kx, a.ky, a.n, tid, 256);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int x0 = tx * PW, y0 = ty * PH;
        // transformed values of the tile + halo (replicate padding == clamped coordinates), four per lane and step
        bool any_nan = false;
        for (int i0 = tid; i0 < PHH * PHW; i0 += 4 * 256) {
            float d[4], val[4];
            bool mk[4];
            int at[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + 256 * k;
                const bool in = i < PHH * PHW;
                const int ly = in ? i / PHW : 0, lx = in ? i - ly * PHW : 0;
                int gy = y0 + ly - 1, gx = x0 + lx - 1;
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                const int idx = gy * a.W + gx;
                d[k] = read_depth(a, idx);
                mk[k] = in && (a.mask ? (a.mask[idx] != 0) : (d[k] > 0.0f));      // :238-241
                at[k] = in ? ly * PSTRIDE + lx : -1;
            }
            ddmath::curve_eval4(curve, d, mk, val);                               // :185-191
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (at[k] >= 0) { s_val[at[k]] = val[k]; any_nan |= val[k] != val[k]; }
        }
        const bool tile_nan = __syncthreads_or((int)any_nan) != 0;
        // four consecutive pixels of a row per lane and step
        for (int g = tid; g < PH * (PW / 4); g += 256) {
            const int ly = g / (PW / 4), lx = (g - ly * (PW / 4)) * 4;
            const int gy = y0 + ly, gx = x0 + lx;
            if (gy >= a.H || gx >= a.W) continue;
            const float *r0 = s_val + ly * PSTRIDE + lx, *r1 = r0 + PSTRIDE, *r2 = r1 + PSTRIDE;      // columns lx-1 .. lx+4 of the halo tile
            float r[4];
            if (a.skip_smoothing) {
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = r1[k + 1];
            } else if (!tile_nan) {
                ddmath::median9x4(r0, r1, r2, r);                               // :194-200
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v[9] = {r0[k], r0[k + 1], r0[k + 2], r1[k], r1[k + 1], r1[k + 2], r2[k], r2[k + 1], r2[k + 2]};
                    r[k] = ddmath::median9(v);
                }
            }
            float *o = a.out + (size_t)gy * a.W + gx;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (gx + k < a.W) o[k] = (r1[k + 1] != 0.0f) ? r[k] : 0.0f;     // :203 (a masked pixel's value is >= 1e-3 or NaN, never 0)
        }
        __syncthreads();                      // the next tile overwrites s_val
    }
}

// ==================================================================================================
// The correspondence half of DepthRefiner.refine_depth (src/depthdensifier/depth_refiner.py:244-299) as ONE launch
// per view: project the sparse COLMAP points (:92-115), keep those inside the image minus the edge margin with a
// positive depth (:247-254), sample the depth map bilinearly at the projections (F.grid_sample, bilinear, zero
// padding, align_corners=True, :260-272), keep positive samples (:281-288), drop ratio outliers by IQR (:117-139) --
// ~25 small tensor launches and several host synchronisations as stock tensor ops, the pipeline's main-thread
// bottleneck once image decoding runs on I/O threads.  One workgroup per view; order statistics by radix select
// (no size limit, no sort); both compactions are stable, so the correspondences come out in the order the reference's
// boolean-mask indexing gives them.  float32 arithmetic in the reference's operation order, no contraction.
// ==================================================================================================
constexpr int FIT_T = 1024;

struct FitArgs {
    const float *points;        // (n,3) world
    const void *depth;
    float *z_mono, *z_metric;   // (n) out: compacted correspondences (depth-map sample, COLMAP depth)
    float *ratio;               // (n) scratch
    int *meta;                  // out: [0] in bounds, [1] sampled > 0, [2] kept, [3] outliers removed, [4] float bits of the scale factor
    float E[12];
    float K[6];                 // rows 0 and 1 of the calibration matrix
    float outlier_threshold;
    int n, H, W, f16, edge_margin, robust;
    int half_io;                // FP16 mode of the reference (:85-86, 233-236): points and results are quantised to half, arithmetic stays float32
};

__device__ __forceinline__ unsigned fit_key(float f) {           // order-preserving float -> uint
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// k-th smallest (0-based) of vals[0..m): 4 passes of 8-bit histograms on the order-preserving key
__device__ float fit_select(const float *vals, int m, int k, unsigned *hist, int tid) {
    unsigned prefix = 0, mask = 0;
    __shared__ unsigned s_bin;
    __shared__ int s_k;
    if (tid == 0) s_k = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += FIT_T) hist[b] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += FIT_T) {
            const unsigned key = fit_key(vals[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int kk = s_k;
            unsigned b = 0;
            for (; b < 255; ++b) { if (kk < (int)hist[b]) break; kk -= (int)hist[b]; }
            s_bin = b; s_k = kk;
        }
        __syncthreads();
        prefix |= s_bin << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned key = prefix;
    const unsigned bits = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    return __uint_as_float(bits);
}

// torch.quantile(x, q, interpolation="linear") for float32: rank = q (n-1) in float32, torch.lerp between the neighbours
__device__ float fit_quantile(const float *vals, int m, float q, unsigned *hist, int tid) {
#pragma clang fp contract(off)
    const float rank = q * (float)(m - 1);
    const float below = floorf(rank);
    const int lo = (int)below, hi = (int)ceilf(rank);
    const float w = rank - below;
    const float a = fit_select(vals, m, lo, hist, tid);
    const float b = hi == lo ? a : fit_select(vals, m, hi, hist, tid);
    return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w);
}

// stable compaction step of one chunk of FIT_T candidates: returns this thread's output slot or -1
__device__ __forceinline__ int fit_slot(bool keep, int tid, int *s_wave, int *s_base) {
    const unsigned long long b = __ballot(keep);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int before = *s_base, total = 0;
    for (int w = 0; w < FIT_T / 64; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
    const int slot = keep ? before + __popcll(b & ((1ull << lane) - 1ull)) : -1;
    __syncthreads();
    if (tid == 0) *s_base += total;
    __syncthreads();
    return slot;
}

__global__ __launch_bounds__(FIT_T) void refine_fit_kernel(const FitArgs a) {
#pragma clang fp contract(off)
    __shared__ unsigned s_hist[256];
    __shared__ int s_wave[FIT_T / 64];
    __shared__ int s_base;
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    if (tid == 0) { s_base = 0; s_cnt = 0; }
    __syncthreads();
    const float wm = (float)(a.W - a.edge_margin), hm = (float)(a.H - a.edge_margin), em = (float)a.edge_margin;
    const float wden = (float)(a.W - 1), hden = (float)(a.H - 1);
    // ---- project, bounds, bilinear sample, keep positive samples (stable) ----
    for (int c0 = 0; c0 < a.n; c0 += FIT_T) {
        const int i = c0 + tid;
        bool keep = false, inb = false;
        float zs = 0.0f, zc = 0.0f;
        if (i < a.n) {
            float x = a.points[3 * i], y = a.points[3 * i + 1], z = a.points[3 * i + 2];
            if (a.half_io) { x = (float)(_Float16)x; y = (float)(_Float16)y; z = (float)(_Float16)z; }
            // (H @ [p;1]) row by row (:99-101)
            const float xc = a.E[0] * x + a.E[1] * y + a.E[2] * z + a.E[3];
            const float yc = a.E[4] * x + a.E[5] * y + a.E[6] * z + a.E[7];
            zc = a.E[8] * x + a.E[9] * y + a.E[10] * z + a.E[11];
            float u = 0.0f, v = 0.0f;
            if (zc > 0.0f) {                                       // :104-112
                const float xn = xc / zc, yn = yc / zc;
                u = (a.K[0] * xn + a.K[1] * yn) + a.K[2];
                v = (a.K[3] * xn + a.K[4] * yn) + a.K[5];
            }
            inb = (u >= em) && (u < wm) && (v >= em) && (v < hm) && (zc > 0.0f);      // :248-254
            if (inb) {
                // :262-263 normalised coordinates, then grid_sample's un-normalisation (align_corners=True)
                const float gx = (u / wden) * 2.0f - 1.0f, gy = (v / hden) * 2.0f - 1.0f;
                const float ix = ((gx + 1.0f) / 2.0f) * wden, iy = ((gy + 1.0f) / 2.0f) * hden;
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
                const float nw = ((fx0 + 1.0f) - ix) * ((fy0 + 1.0f) - iy), ne = (ix - fx0) * ((fy0 + 1.0f) - iy);
                const float sw = ((fx0 + 1.0f) - ix) * (iy - fy0), se = (ix - fx0) * (iy - fy0);
                auto at = [&](int yy, int xx) -> float {
                    if (xx < 0 || yy < 0 || xx >= a.W || yy >= a.H) return 0.0f;      // padding_mode="zeros"
                    const long long p = (long long)yy * a.W + xx;
                    return a.f16 ? (float)reinterpret_cast<const _Float16 *>(a.depth)[p] : reinterpret_cast<const float *>(a.depth)[p];
                };
                float acc = at(y0, x0) * nw;
                acc = acc + at(y0, x1) * ne;
                acc = acc + at(y1, x0) * sw;
                acc = acc + at(y1, x1) * se;
                zs = acc;
                if (a.half_io) { zs = (float)(_Float16)zs; zc = (float)(_Float16)zc; }
                keep = zs > 0.0f;                                   // :281
            }
        }
        const unsigned long long bb = __ballot(inb);
        if ((tid & 63) == 0 && bb) atomicAdd(&s_cnt, __popcll(bb));
        const int slot = fit_slot(keep, tid, s_wave, &s_base);
        if (slot >= 0) { a.z_mono[slot] = zs; a.z_metric[slot] = zc; }
    }
    __syncthreads();
    const int m1 = s_base;
    int m = m1, removed = 0;
    __threadfence_block();
    // ---- ratio outliers by IQR (:117-139), only if robust and more than 10 correspondences (:292) ----
    if (a.robust && m1 > 10) {
        for (int i = tid; i < m1; i += FIT_T) a.ratio[i] = a.z_metric[i] / (a.z_mono[i] + 1e-6f);
        __syncthreads();
        const float med = fit_select(a.ratio, m1, (m1 - 1) / 2, s_hist, tid);          // torch.median: the lower middle
        const float q75 = fit_quantile(a.ratio, m1, 0.75f, s_hist, tid);
        const float q25 = fit_quantile(a.ratio, m1, 0.25f, s_hist, tid);
        const float thr = a.outlier_threshold * (q75 - q25);
        if (tid == 0) s_base = 0;
        __syncthreads();
        // in-place stable compaction: a slot is never beyond the element it comes from, chunks are processed in order,
        // and every element of a chunk is read before any of that chunk is written
        for (int c0 = 0; c0 < m1; c0 += FIT_T) {
            const int i = c0 + tid;
            float zm = 0.0f, zt = 0.0f;
            bool keep = false;
            if (i < m1) { zm = a.z_mono[i]; zt = a.z_metric[i]; keep = fabsf(a.ratio[i] - med) < thr; }
            const int slot = fit_slot(keep, tid, s_wave, &s_base);
            if (slot >= 0) { a.z_mono[slot] = zm; a.z_metric[slot] = zt; }
        }
        __syncthreads();
        m = s_base;
        removed = m1 - m;
    }
    // ---- effective scale: median of z_metric / (z_mono + 1e-6) over the kept correspondences (:315) ----
    float scale = 1.0f;
    if (m > 0) {
        for (int i = tid; i < m; i += FIT_T) a.ratio[i] = a.z_metric[i] / (a.z_mono[i] + 1e-6f);
        __syncthreads();
        scale = fit_select(a.ratio, m, (m - 1) / 2, s_hist, tid);
    }
    if (tid == 0) { a.meta[0] = s_cnt; a.meta[1] = m1; a.meta[2] = m; a.meta[3] = removed; a.meta[4] = (int)__float_as_uint(scale); }
}

// Sort the knots by x (torch.argsort at :149-151) in one workgroup: bitonic network in LDS, up to 4096 knots.  The key is
// the pair (x, input index): equal x keep their input order (a stable sort), and the padding to a power of two
// (+inf, index >= n) sorts strictly behind every real knot -- a real correspondence whose sampled depth is +inf (a bilinear
// sample that touches an inf pixel passes the "> 0" test) stays inside the first n outputs with its own y.
constexpr int SORT_MAX = 4096;
__global__ __launch_bounds__(1024) void sort_knots_kernel(const float *x, const float *y, int n, float *xs, float *ys) {
    __shared__ float s_x[SORT_MAX], s_y[SORT_MAX];
    __shared__ unsigned short s_i[SORT_MAX];
    const int tid = threadIdx.x;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 1024) { s_x[i] = i < n ? x[i] : __builtin_inff(); s_y[i] = i < n ? y[i] : 0.0f; s_i[i] = (unsigned short)i; }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const float a0 = s_x[i], a1 = s_x[p];
                    const unsigned short i0 = s_i[i], i1 = s_i[p];
                    const bool greater = a0 > a1 || (a0 == a1 && i0 > i1);          // (x, index) order; indices are distinct
                    if (greater == up) {
                        s_x[i] = a1; s_x[p] = a0;
                        s_i[i] = i1; s_i[p] = i0;
                        const float t = s_y[i]; s_y[i] = s_y[p]; s_y[p] = t;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 1024) { xs[i] = s_x[i]; ys[i] = s_y[i]; }
}

thread_local char g_rerr[192] = "";

// the number of masked pixels of a view (mask != 0, or depth > 0 without a mask: depth_refiner.py:238-241), clamped to 2^31 - 1
// (what DepthRefiner needs to choose between the curve and the degenerate branches of :143-154)
__global__ __launch_bounds__(256) void count_masked_kernel(const uint8_t *mask, const void *depth, const int f16, const long long hw, int *out) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 64;
    int cnt = 0;
    for (int k = 0; k < 64 && i + k < hw; ++k) {
        const long long j = i + k;
        cnt += mask ? (mask[j] != 0) : ((f16 ? (float)reinterpret_cast<const _Float16 *>(depth)[j] : reinterpret_cast<const float *>(depth)[j]) > 0.0f);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) {
        const int before = atomicAdd(out, cnt);
        if (before < 0 || before + cnt < 0) atomicExch(out, 0x7fffffff);      // (saturates: views of 2^31 pixels do not exist on this path)
    }
}

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
    RArgs a{};
    a.depth = depth; a.mask = mask; a.kx = knots_x; a.ky = knots_y; a.out = refined_out;
    a.H = height; a.W = width; a.n = n_knots; a.f16 = depth_dtype == DD_F16; a.skip_smoothing = skip_smoothing;
    const uint32_t lab = dd_lab_word();          // the experiment switches of ddcore_lab.h (zero in production)
    const bool plain = (lab & DD_LAB_APPLY_PLAIN) != 0u;
    if (n_knots <= MAX_LDS_KNOTS && !plain) {
        const int tiles_x = (width + PW - 1) / PW, tiles = tiles_x * ((height + PH - 1) / PH);
        const int lab_wgs = (int)((lab >> 8) & 0x1fffu), max_wgs = lab_wgs > 0 ? lab_wgs : 4096;
        const int wgs = tiles < max_wgs ? tiles : max_wgs;     // (the knots and their grid are built once per workgroup)
        hipLaunchKernelGGL(refine_apply_tiles, dim3(wgs), dim3(256), (size_t)n_knots * 3 * sizeof(float), (hipStream_t)stream, a, tiles_x, tiles);
    } else {
        const dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH);
        hipLaunchKernelGGL(refine_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    if (hipGetLastError() != hipSuccess) { snprintf(g_rerr, sizeof(g_rerr), "refine_apply launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

int dd_refine_fit(const float *points, int32_t n, const float *cam_from_world, const float *calibration, const void *depth,
                  int32_t depth_dtype, int32_t height, int32_t width, int32_t edge_margin, int32_t robust, float outlier_threshold,
                  int32_t half_precision_io, float *z_mono_out, float *z_metric_out, float *scratch, int32_t *meta_out, void *stream) {
    auto fail = [](const char *m) { snprintf(g_rerr, sizeof(g_rerr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (n < 0) return fail("n is negative");
    if (!cam_from_world || !calibration || !depth || !meta_out) return fail("cam_from_world / calibration / depth / meta_out is NULL");
    if (n > 0 && (!points || !z_mono_out || !z_metric_out || !scratch)) return fail("points / outputs / scratch is NULL");
    if (height <= 1 || width <= 1) return fail("height/width must be at least 2");
    if (depth_dtype != DD_F32 && depth_dtype != DD_F16) return fail("depth_dtype must be DD_F32 or DD_F16");
    FitArgs a{};
    a.points = points; a.depth = depth; a.z_mono = z_mono_out; a.z_metric = z_metric_out; a.ratio = scratch; a.meta = meta_out;
    for (int k = 0; k < 12; ++k) a.E[k] = cam_from_world[k];
    for (int k = 0; k < 6; ++k) a.K[k] = calibration[k];
    a.outlier_threshold = outlier_threshold; a.n = n; a.H = height; a.W = width; a.f16 = depth_dtype == DD_F16;
    a.edge_margin = edge_margin; a.robust = robust; a.half_io = half_precision_io;
    hipLaunchKernelGGL(refine_fit_kernel, dim3(1), dim3(FIT_T), 0, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) { snprintf(g_rerr, sizeof(g_rerr), "refine_fit launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

int dd_refine_fit_async(const float *points_host, int32_t n, const float *cam_from_world, const float *calibration, const void *depth,
                        int32_t depth_dtype, int32_t height, int32_t width, int32_t edge_margin, int32_t robust, float outlier_threshold,
                        int32_t half_precision_io, const uint8_t *mask, float *work, int32_t *meta_dev, int32_t *meta_host,
                        void *ready_event, void *stream) {
    auto fail = [](const char *m) { snprintf(g_rerr, sizeof(g_rerr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (n < 0) return fail("n is negative");
    if (!meta_dev || !meta_host) return fail("meta_dev / meta_host is NULL");
    if (n > 0 && (!points_host || !work)) return fail("points_host / work is NULL");
    hipStream_t s = (hipStream_t)stream;
    // one stream, in order: the sparse points up, the result words cleared, the fit, the number of masked pixels, the words down, the event
    if (n > 0 && hipMemcpyAsync(work, points_host, (size_t)n * 12, hipMemcpyHostToDevice, s) != hipSuccess) { (void)hipGetLastError(); snprintf(g_rerr, sizeof(g_rerr), "hipMemcpyAsync(points) failed"); return DD_ERR_LAUNCH; }
    if (hipMemsetAsync(meta_dev, 0, 32, s) != hipSuccess) { (void)hipGetLastError(); snprintf(g_rerr, sizeof(g_rerr), "hipMemsetAsync(meta) failed"); return DD_ERR_LAUNCH; }
    float *const z_mono = work + (size_t)3 * n, *const z_metric = z_mono + n, *const scratch = z_metric + n;
    int rc = dd_refine_fit(work, n, cam_from_world, calibration, depth, depth_dtype, height, width, edge_margin, robust, outlier_threshold,
                           half_precision_io, z_mono, z_metric, scratch, meta_dev, stream);
    if (rc != DD_OK) return rc;
    const long long hw = (long long)height * width;
    hipLaunchKernelGGL(count_masked_kernel, dim3((unsigned)((hw + 16383) / 16384)), dim3(256), 0, s, mask, depth, depth_dtype == DD_F16 ? 1 : 0, hw, meta_dev + 5);
    if (hipGetLastError() != hipSuccess) { snprintf(g_rerr, sizeof(g_rerr), "count_masked launch failed"); return DD_ERR_LAUNCH; }
    if (hipMemcpyAsync(meta_host, meta_dev, 32, hipMemcpyDeviceToHost, s) != hipSuccess) { (void)hipGetLastError(); snprintf(g_rerr, sizeof(g_rerr), "hipMemcpyAsync(meta) failed"); return DD_ERR_LAUNCH; }
    if (ready_event && hipEventRecord((hipEvent_t)ready_event, s) != hipSuccess) { (void)hipGetLastError(); snprintf(g_rerr, sizeof(g_rerr), "hipEventRecord failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

int dd_sort_knots(const float *x, const float *y, int32_t n, float *x_sorted, float *y_sorted, void *stream) {
    auto fail = [](const char *m) { snprintf(g_rerr, sizeof(g_rerr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (n < 0 || n > SORT_MAX) return fail("dd_sort_knots handles 0..4096 knots");
    if (n > 0 && (!x || !y || !x_sorted || !y_sorted)) return fail("a knot array is NULL");
    if (n == 0) return DD_OK;
    hipLaunchKernelGGL(sort_knots_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, n, x_sorted, y_sorted);
    if (hipGetLastError() != hipSuccess) { snprintf(g_rerr, sizeof(g_rerr), "sort_knots launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

}  // extern "C"
