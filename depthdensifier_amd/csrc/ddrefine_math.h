// ddrefine_math.h -- the per-pixel arithmetic of DepthRefiner shared by dd_refine_apply (csrc/ddrefine.hip) and by the
// densify kernel's fused refine stage (csrc/ddcore.hip): both must produce the same bits.
//   src/depthdensifier/depth_refiner.py:141-178  sorted look-up table, linear interpolation, clamped, floored at 1e-3
//   src/depthdensifier/depth_refiner.py:194-200  3x3 median (torch.median: a window holding a NaN gives NaN)
#ifndef DDREFINE_MATH_H
#define DDREFINE_MATH_H

#include <hip/hip_runtime.h>

namespace ddmath {

// The linear blend of :160-176 between the knots (x0, y0) and (x1, y1).  Round 6: t = (d - x0) * (1 / dx) instead of (d - x0) / dx --
// the reciprocal of an interval is a property of the CURVE (looked up once per interval, `Curve::inv`), the division was a tenth of
// the fused refine stage's per-pixel instructions.  Differs from the reference's quotient by at most two ulps of t, i.e. by
// 1.2e-7 |y1 - y0| in the value: inside the 4e-6-of-range float32 budget of tests/test_refiner.py (the golden is the bar);
// every kernel of the library evaluates the curve through these two functions, so they agree with each other bit for bit.
__device__ __forceinline__ float inv_dx(float x0, float x1) {
    float dx = x1 - x0;
    if (dx == 0.0f) dx = 1e-6f;             // :165-166
    return 1.0f / dx;
}
__device__ __forceinline__ float blend(float d, float x0, float y0, float y1, float inv) {
#pragma clang fp contract(off)   // the reference rounds after every tensor op: no FMA here
    float t = (d - x0) * inv;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return fmaxf(y0 + t * (y1 - y0), 1e-3f);
}

// torch.searchsorted(xs, d, right=False) clamped to [1, n-1], then the blend: the plain form (any number of knots, global or LDS arrays)
__device__ __forceinline__ float lut(const float *kx, const float *ky, int n, float d) {
    if (d != d) return d;                   // torch.clamp / torch.maximum propagate a NaN depth (:168-176): the pixel later fails depth > 0
    int lo = 0, hi = n;                     // first index with kx[i] >= d
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kx[mid] < d) lo = mid + 1; else hi = mid;
    }
    const int i = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    return blend(d, kx[i - 1], ky[i - 1], ky[i], inv_dx(kx[i - 1], kx[i]));
}

// ==================================================================================================
// The curve as the kernels keep it in LDS (round 6): the knots interleaved {x, y} (one 16-byte read fetches both ends of an
// interval), the reciprocal width of every interval, and a grid of CURVE_BUCKETS equal buckets over [x_first, x_last] that says
// where in the knots to look: entry j = the number of knots whose BUCKET is below j.  The bucket function is monotone in its
// argument (a rounded subtraction of a constant, a rounded multiplication by a positive constant, a clamp, a truncation), so for a
// depth d of bucket b every knot of a lower bucket is < d and every knot of a higher bucket is > d: lower_bound(d) lies in
// [grid[b], grid[b + 1]] -- exactly the knots that share d's bucket (about one on average), whatever the rounding does.
// (Round 5 derived the grid from bucket EDGES computed on their own and had to search three buckets to be safe.)
// ==================================================================================================
#ifndef DD_CURVE_BUCKETS
#define DD_CURVE_BUCKETS 768
#endif
constexpr int CURVE_BUCKETS = DD_CURVE_BUCKETS;
constexpr int CURVE_GRID_WORDS = CURVE_BUCKETS + 2;             // entries 0 .. CURVE_BUCKETS, padded to an even count

struct Curve {
    const float2 *kxy;            // n knots {x, y}, ascending x
    const float *inv;             // inv[i] = inv_dx(x[i-1], x[i]) for i in [1, n)
    const unsigned short *grid;   // CURVE_BUCKETS + 1 entries
    int n;
    float x0, scale;              // bucket of d: (d - x0) * scale, clamped to [0, CURVE_BUCKETS - 1]; scale 0 = everything in bucket 0
};
__device__ __forceinline__ int curve_bucket(float d, float x0, float scale) {
    return (int)fminf(fmaxf((d - x0) * scale, 0.0f), (float)(CURVE_BUCKETS - 1));      // (NaN -> bucket 0: fmaxf returns the other operand)
}
// the grid is useless where a bucket is not much wider than an ulp of the knots, or the range is not finite: one bucket then
__device__ __forceinline__ float curve_scale(float x_first, float x_last) {
    const float range = x_last - x_first;
    const bool ok = range > fmaxf(fabsf(x_first), fabsf(x_last)) * (1.0f / 128.0f) && range < 3.0e38f;
    return ok ? (float)CURVE_BUCKETS / range : 0.0f;
}
// Builds the curve in LDS from the sorted knots in memory; all `nthreads` threads of the workgroup call it (it contains barriers).
__device__ __forceinline__ Curve curve_build(float2 *s_kxy, float *s_inv, unsigned short *s_grid, const float *kx, const float *ky, int n,
                                             int tid, int nthreads) {
    for (int i = tid; i < n; i += nthreads) s_kxy[i] = make_float2(kx[i], ky[i]);
    __syncthreads();
    Curve c;
    c.kxy = s_kxy; c.inv = s_inv; c.grid = s_grid; c.n = n;
    c.x0 = s_kxy[0].x;
    c.scale = curve_scale(c.x0, s_kxy[n - 1].x);
    for (int i = tid + 1; i < n; i += nthreads) s_inv[i] = inv_dx(s_kxy[i - 1].x, s_kxy[i].x);
    for (int j = tid; j <= CURVE_BUCKETS; j += nthreads) {      // knots with a bucket below j: lower_bound over the (ascending) buckets of the knots
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (curve_bucket(s_kxy[mid].x, c.x0, c.scale) < j) lo = mid + 1; else hi = mid;
        }
        s_grid[j] = (unsigned short)lo;
    }
    __syncthreads();
    return c;
}
__device__ __forceinline__ float curve_blend(const Curve &c, float d, int lo) {
    const int i = lo < 1 ? 1 : (lo > c.n - 1 ? c.n - 1 : lo);
    const float2 a = c.kxy[i - 1], b = c.kxy[i];
    return blend(d, a.x, a.y, b.y, c.inv[i]);
}
// Four values at once, their (short) bisections in lock step: the four chains of dependent LDS reads overlap, and the trip count is
// what the slowest lane of the wave needs.  out[k] = 0 where !mk[k]; a NaN depth comes back as it is.
template <int N>
__device__ __forceinline__ void curve_evalN(const Curve &c, const float (&d)[N], const bool (&mk)[N], float (&out)[N]) {
    int base[N], len[N];
    int any = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int b = curve_bucket(d[k], c.x0, c.scale);
        base[k] = c.grid[b];
        len[k] = (int)c.grid[b + 1] - base[k];
        if (!mk[k] || d[k] != d[k]) len[k] = 0;
        any |= len[k];
    }
    while (__any(any > 0)) {
        any = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {                  // lower_bound of d[k] in [base, base + len]
            const int half = len[k] >> 1;
            const bool right = len[k] > 0 && c.kxy[base[k] + half].x < d[k];
            base[k] = right ? base[k] + half + 1 : base[k];
            len[k] = right ? len[k] - half - 1 : half;
            any |= len[k];
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = !mk[k] ? 0.0f : (d[k] != d[k]) ? d[k] : curve_blend(c, d[k], base[k]);
}
__device__ __forceinline__ void curve_eval4(const Curve &c, const float (&d)[4], const bool (&mk)[4], float (&out)[4]) { curve_evalN<4>(c, d, mk, out); }
__device__ __forceinline__ float curve_eval(const Curve &c, float d) {      // one value (ragged ends)
    if (d != d) return d;
    const int b = curve_bucket(d, c.x0, c.scale);
    int lo = c.grid[b], hi = c.grid[b + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c.kxy[mid].x < d) lo = mid + 1; else hi = mid;
    }
    return curve_blend(c, d, lo);
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
template <typename Row>
__device__ __forceinline__ void median9x4(const Row &r0, const Row &r1, const Row &r2, float (&out)[4]) {
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
