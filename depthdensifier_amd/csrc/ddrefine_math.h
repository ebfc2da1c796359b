// ddrefine_math.h -- the per-pixel arithmetic of DepthRefiner shared by dd_refine_apply (csrc/ddrefine.hip) and by the
// densify kernel's fused refine stage (csrc/ddcore.hip): both must produce the same bits.
//   src/depthdensifier/depth_refiner.py:141-178  sorted look-up table, linear interpolation, clamped, floored at 1e-3
//   src/depthdensifier/depth_refiner.py:194-200  3x3 median (torch.median: a window holding a NaN gives NaN)
#ifndef DDREFINE_MATH_H
#define DDREFINE_MATH_H

#include <hip/hip_runtime.h>

namespace ddmath {

// the linear blend of :160-176 on the interval torch.searchsorted(xs, d, right=False) selects: `lo` = first index with kx[lo] >= d
__device__ __forceinline__ float lut_blend(const float *kx, const float *ky, int n, float d, int lo) {
#pragma clang fp contract(off)   // the reference rounds after the multiply (separate tensor ops): no FMA here
    const int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const float x0 = kx[i - 1], x1 = kx[i], y0 = ky[i - 1], y1 = ky[i];
    float dx = x1 - x0;
    if (dx == 0.0f) dx = 1e-6f;
    float t = (d - x0) / dx;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return fmaxf(y0 + t * (y1 - y0), 1e-3f);
}

// torch.searchsorted(xs, d, right=False) clamped to [1, n-1], then the blend
__device__ __forceinline__ float lut(const float *kx, const float *ky, int n, float d) {
    if (d != d) return d;                   // torch.clamp / torch.maximum propagate a NaN depth (:168-176): the pixel later fails depth > 0
    int lo = 0, hi = n;                     // first index with kx[i] >= d
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kx[mid] < d) lo = mid + 1; else hi = mid;
    }
    return lut_blend(kx, ky, n, d, lo);
}

// The same value found faster (the densify kernel's fused refine stage evaluates the curve 16 000 times per tile): a grid of
// LUT_BUCKETS equal buckets over [kx[0], kx[n-1]] holds, per bucket, a knot index that is known to lie at or before the answer;
// the bisection then runs over three buckets' worth of knots (~1 knot per bucket).  The interval is the one `lut` finds --
// lower_bound is monotone, and both ends are lower_bounds of values a whole bucket away from any depth of the bucket, far more
// than the rounding of either side (the grid is switched off, inv = 0, when a bucket is not much wider than an ulp of the knots).
#ifndef DD_LUT_BUCKETS
#define DD_LUT_BUCKETS 512          // (1024: the same speed, 1 KiB more LDS per workgroup)
#endif
constexpr int LUT_BUCKETS = DD_LUT_BUCKETS;

struct LutGrid {
    float x0, inv;                          // bucket of d: (d - x0) * inv, clamped to [0, LUT_BUCKETS - 1]
};
__device__ __forceinline__ LutGrid lut_grid(const float *kx, int n) {
    const float x0 = kx[0], xn = kx[n - 1], range = xn - x0;
    const bool ok = range > fmaxf(fabsf(x0), fabsf(xn)) * (1.0f / 128.0f) && range < 3.0e38f;
    return LutGrid{x0, ok ? (float)LUT_BUCKETS / range : 0.0f};
}
// entry j of the grid: the first knot >= the lower edge of bucket j - 1 (0 for the first bucket and for a grid switched off)
__device__ __forceinline__ unsigned short lut_grid_entry(const float *kx, int n, const LutGrid g, int j) {
    if (j < 1 || g.inv == 0.0f) return 0;
    const float edge = g.x0 + (float)(j - 1) * ((kx[n - 1] - g.x0) * (1.0f / (float)LUT_BUCKETS));
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kx[mid] < edge) lo = mid + 1; else hi = mid;
    }
    return (unsigned short)lo;
}
// Four values at once, their bisections in lock step (the four chains of dependent LDS reads overlap; the trip count is what the
// slowest lane of the wave needs -- two or three steps with a grid, not ten).  For each value the answer lies between the entry of
// its bucket (every knot before it is < d) and the entry three buckets on (the first knot >= an edge a whole bucket above d).
// out[k] = 0 where !mk[k].
__device__ __forceinline__ void lut_grid_eval4(const float *kx, const float *ky, int n, const LutGrid g, const unsigned short *grid,
                                               const float (&d)[4], const bool (&mk)[4], float (&out)[4]) {
    int base[4], len[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = (int)fminf(fmaxf((d[k] - g.x0) * g.inv, 0.0f), (float)(LUT_BUCKETS - 1));      // (NaN -> bucket 0; its result is d itself)
        base[k] = grid[b];
        len[k] = ((g.inv != 0.0f && b + 3 < LUT_BUCKETS) ? (int)grid[b + 3] : n) - base[k];
        if (!mk[k] || d[k] != d[k]) len[k] = 0;
    }
    // (tried: the range scanned four knots per step with independent reads instead of halved -- 20.7 against 19.6 us per view; the
    //  count of knots below d built from its top bit down with wave-uniform steps, 8 instead of 13 instructions per step -- 19.8)
    while (__any((len[0] | len[1] | len[2] | len[3]) > 0)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                  // lower_bound of d[k] in [base, base + len]
            const int half = len[k] >> 1;
            const bool right = len[k] > 0 && kx[base[k] + half] < d[k];
            base[k] = right ? base[k] + half + 1 : base[k];
            len[k] = right ? len[k] - half - 1 : half;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = !mk[k] ? 0.0f : (d[k] != d[k]) ? d[k] : lut_blend(kx, ky, n, d[k], base[k]);
}

// median of 9; NaN in -> NaN out like torch.median.  Default: the three-instruction column sort + v_med3 form -- sort
// each triple (min3 / med3 / max3), then median = med3(max of the minima, med of the medians, min of the maxima):
// 13 VALU instructions instead of the 38 of the 19-exchange network (DD_MEDIAN_NETWORK=1 keeps that one for A/B).
// Either selects one of its inputs, so the result is the same bits.
#ifndef DD_MEDIAN_NETWORK
#define DD_MEDIAN_NETWORK 0
#endif
__device__ __forceinline__ float median9(float (&v)[9]) {
    bool has_nan = false;
#pragma unroll
    for (int k = 0; k < 9; ++k) has_nan |= v[k] != v[k];
#if !DD_MEDIAN_NETWORK
    {
        float lo[3], mid[3], hi[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a = v[3 * r], b = v[3 * r + 1], c = v[3 * r + 2];
            lo[r] = fminf(fminf(a, b), c);
            hi[r] = fmaxf(fmaxf(a, b), c);
            mid[r] = __builtin_amdgcn_fmed3f(a, b, c);
        }
        const float m = __builtin_amdgcn_fmed3f(fmaxf(fmaxf(lo[0], lo[1]), lo[2]), __builtin_amdgcn_fmed3f(mid[0], mid[1], mid[2]),
                                                fminf(fminf(hi[0], hi[1]), hi[2]));
        return has_nan ? __builtin_nanf("") : m;
    }
#endif
#define DD_CSWAP(a, b) { const float lo_ = fminf(v[a], v[b]), hi_ = fmaxf(v[a], v[b]); v[a] = lo_; v[b] = hi_; }
    DD_CSWAP(1, 2) DD_CSWAP(4, 5) DD_CSWAP(7, 8) DD_CSWAP(0, 1) DD_CSWAP(3, 4) DD_CSWAP(6, 7) DD_CSWAP(1, 2) DD_CSWAP(4, 5) DD_CSWAP(7, 8)
    DD_CSWAP(0, 3) DD_CSWAP(5, 8) DD_CSWAP(4, 7) DD_CSWAP(3, 6) DD_CSWAP(1, 4) DD_CSWAP(2, 5) DD_CSWAP(4, 7) DD_CSWAP(4, 2) DD_CSWAP(6, 4) DD_CSWAP(4, 2)
#undef DD_CSWAP
    return has_nan ? __builtin_nanf("") : v[4];
}

// The medians of FOUR horizontally consecutive 3x3 windows at once (windows that lie inside the image and hold no NaN): the six
// columns they share are sorted once (min3 / med3 / max3), every window is the med3 of (max of its three column minima, med of the
// medians, min of the maxima).  r0 / r1 / r2: the three rows, six values each (columns x-1 .. x+4).  Selects inputs like median9:
// the same bits (no -0 can occur: a value is 0, >= 1e-3 or NaN).
__device__ __forceinline__ void median9x4(const float *r0, const float *r1, const float *r2, float (&out)[4]) {
    float lo[6], mid[6], hi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float a = r0[c], b = r1[c], d = r2[c];
        lo[c] = fminf(fminf(a, b), d);
        hi[c] = fmaxf(fmaxf(a, b), d);
        mid[c] = __builtin_amdgcn_fmed3f(a, b, d);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        out[k] = __builtin_amdgcn_fmed3f(fmaxf(fmaxf(lo[k], lo[k + 1]), lo[k + 2]), __builtin_amdgcn_fmed3f(mid[k], mid[k + 1], mid[k + 2]),
                                         fminf(fminf(hi[k], hi[k + 1]), hi[k + 2]));
}

}  // namespace ddmath

#endif
