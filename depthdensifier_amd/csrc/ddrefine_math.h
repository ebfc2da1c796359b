// ddrefine_math.h -- the per-pixel arithmetic of DepthRefiner shared by dd_refine_apply (csrc/ddrefine.hip) and by the
// densify kernel's fused refine stage (csrc/ddcore.hip): both must produce the same bits.
//   src/depthdensifier/depth_refiner.py:141-178  sorted look-up table, linear interpolation, clamped, floored at 1e-3
//   src/depthdensifier/depth_refiner.py:194-200  3x3 median (torch.median: a window holding a NaN gives NaN)
#ifndef DDREFINE_MATH_H
#define DDREFINE_MATH_H

#include <hip/hip_runtime.h>

namespace ddmath {

// torch.searchsorted(xs, d, right=False) clamped to [1, n-1], then the linear blend of :160-176
__device__ __forceinline__ float lut(const float *kx, const float *ky, int n, float d) {
#pragma clang fp contract(off)   // the reference rounds after the multiply (separate tensor ops): no FMA here
    if (d != d) return d;                   // torch.clamp / torch.maximum propagate a NaN depth (:168-176): the pixel later fails depth > 0
    int lo = 0, hi = n;                     // first index with kx[i] >= d
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kx[mid] < d) lo = mid + 1; else hi = mid;
    }
    const int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const float x0 = kx[i - 1], x1 = kx[i], y0 = ky[i - 1], y1 = ky[i];
    float dx = x1 - x0;
    if (dx == 0.0f) dx = 1e-6f;
    float t = (d - x0) / dx;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return fmaxf(y0 + t * (y1 - y0), 1e-3f);
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

}  // namespace ddmath

#endif
