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
//
// What is in this file (DDFilterViews.mode; every form gives the same votes, bit for bit):
//   floater_votes_kernel2           float64, image bounds without the division, reciprocal only before the lookup (mode 1)
//   floater_votes_kernel_cull       kernel2's pair test behind a two-level view cull: a mask of visible views per 65 536
//                                   points, then the workgroup's own bounding sphere (mode 3; mode 4 = default of the host
//                                   layer: votes_cull_estimate samples the workgroups and kernel2 or the cull runs)
//   (a float32 first pass with error bounds existed in round 2 and measured slower -- v_fma_f64 issues as fast as a
//    scalar-per-lane v_fma_f32 on gfx950; removed in round 3, the packed form's ceiling is in tools/experiments)
//   compact_* kernels               dd_compact_cloud: stable compaction of the cloud by votes < threshold

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ddcore.h"

#ifndef DD_VOTES_FAST
#define DD_VOTES_FAST 2      // filtered predicates (identical decisions): 1 = one IEEE reciprocal instead of three divisions +
                             // division-free grazing test; 2 = additionally rcp + two Newton steps instead of the IEEE
                             // division and a squared grazing comparison instead of the sqrt; 0 = always the exact formulation
#endif
#ifndef DD_VOTES_STAGE
#define DD_VOTES_STAGE 0     // diagnostic builds stop after stage 1 (depth sign), 2 (bounds), 3 (grazing)
#endif

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

// A camera block whose first 12 doubles ([R|t]) were fetched ahead of time into scalar registers (see the view loop of
// floater_votes_kernel); the rest is read through the constant address space when a pair gets that far.
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef const f64x4 __attribute__((address_space(4))) cam_vec4;
typedef const double __attribute__((address_space(4))) cam_double;
struct CamAhead {
    f64x4 r0, r1, r2;
    cam_double *rest;
    __device__ __forceinline__ double operator[](int k) const {
        return k < 4 ? r0[k] : k < 8 ? r1[k - 4] : k < 12 ? r2[k - 8] : rest[k];
    }
};

// One (point, view) pair: true when the view votes against the point.  `c` is the view's camera block.
template <typename Cam>
__device__ __forceinline__ bool pair_votes(const FArgs &a, const Cam &c, const int v, const double x, const double y,
                                           const double z, const double nx, const double ny, const double nz,
                                           const double wlim, const double hlim) {
    // scripts/test.py:63-68  camera-frame point and depth
    const double zc = c[8] * x + c[9] * y + c[10] * z + c[11];
    if (!(zc > 0.0)) return false;                           // :301 depths > 0
#if DD_VOTES_STAGE == 1
    return true;
#endif
    const double xc = c[0] * x + c[1] * y + c[2] * z + c[3];
    const double yc = c[4] * x + c[5] * y + c[6] * z + c[7];
    // :71-75  normalise by (depth + 1e-8), apply K.  Filtered predicate: the decisions below depend on u, w
    // only through comparisons with integers (image bounds, truncation to a pixel), so a cheaper
    // evaluation (one reciprocal instead of three divisions; error ~1e-15 relative) decides whenever
    // it is farther than 1e-9 from every integer; inside that band the exact formulation is used.
    const double den = zc + 1e-8;
    double u, w;
#if DD_VOTES_FAST
    {
#if DD_VOTES_FAST >= 2
        double rden = __builtin_amdgcn_rcp(den);             // seed, then two Newton steps: relative error ~1e-16,
        rden = fma(fma(-den, rden, 1.0), rden, rden);        // seven orders below the 1e-9 guard band
        rden = fma(fma(-den, rden, 1.0), rden, rden);        // (inf / NaN propagate to u, w and fail `safe`)
#else
        const double rden = 1.0 / den;
#endif
        const double xf = xc * rden, yf = yc * rden, zf = zc * rden;
        u = c[12] * xf + c[13] * yf + c[14] * zf;
        w = c[15] * xf + c[16] * yf + c[17] * zf;
        const double gu = 1e-9 * (fabs(u) + 1.0), gw = 1e-9 * (fabs(w) + 1.0);
        const bool safe = fabs(u - rint(u)) > gu && fabs(w - rint(w)) > gw && fabs(u) < 1e12 && fabs(w) < 1e12;
        if (!safe) {
            const double xn = xc / den, yn = yc / den, zn = zc / den;
            u = c[12] * xn + c[13] * yn + c[14] * zn;
            w = c[15] * xn + c[16] * yn + c[17] * zn;
        }
    }
#else
    {
        const double xn = xc / den, yn = yc / den, zn = zc / den;
        u = c[12] * xn + c[13] * yn + c[14] * zn;
        w = c[15] * xn + c[16] * yn + c[17] * zn;
    }
#endif
    if (!(u >= 0.0 && u < wlim && w >= 0.0 && w < hlim)) return false;      // :300-302
#if DD_VOTES_STAGE == 2
    return true;
#endif
    // :284-295  grazing-angle test against the direction camera centre -> point.  Same idea: with
    // d = p - centre, facing = -(n . d)/|d| > g  <=>  -(n . d) > g |d|; decided without the three
    // divisions unless the two sides are within 1e-9 of each other.
    double dx = x - c[18], dy = y - c[19], dz = z - c[20];
#if DD_VOTES_FAST >= 2
    {
        // t > g |d| with g > 0  <=>  t > 0 and t^2 > g^2 |d|^2: no square root unless the two sides are within
        // 2.5e-9 of each other (or anything is not finite) -- then the exact formulation decides.
        const double len2 = dx * dx + dy * dy + dz * dz;
        const double t = -(nx * dx + ny * dy + nz * dz), g = a.grazing_cos;
        const double s = t * t, q = g * g * len2;
        if (g > 0.0 && t <= 0.0) return false;              // facing <= 0 < g (0/0 = NaN for a zero direction fails too)
        if (g > 0.0 && fabs(s - q) > 2.5e-9 * (s + q) && s < 1e300) {
            if (!(s > q)) return false;
        } else {
            const double len = sqrt(len2);
            dx /= len; dy /= len; dz /= len;
            const double facing = nx * -dx + ny * -dy + nz * -dz;
            if (!(facing > a.grazing_cos)) return false;
        }
    }
#elif DD_VOTES_FAST
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    {
        const double t = -(nx * dx + ny * dy + nz * dz), rhs = a.grazing_cos * len;
        const double guard = 1e-9 * (fabs(t) + fabs(rhs));
        if (fabs(t - rhs) > guard && fabs(t) < 1e300) {
            if (!(t > rhs)) return false;
        } else {
            dx /= len; dy /= len; dz /= len;
            const double facing = nx * -dx + ny * -dy + nz * -dz;
            if (!(facing > a.grazing_cos)) return false;
        }
    }
#else
    const double len = sqrt(dx * dx + dy * dy + dz * dz);
    dx /= len; dy /= len; dz /= len;
    const double facing = nx * -dx + ny * -dy + nz * -dz;
    if (!(facing > a.grazing_cos)) return false;
#endif
#if DD_VOTES_STAGE == 3
    return true;
#endif
    // :308-312  truncating lookup of the view's (mask-zeroed) refined depth
    const long long pix = (long long)v * a.hw + (long long)(int)w * a.W + (int)u;
    float seen = a.depth[pix];
    if (a.mask && a.mask[pix] == 0) seen = 0.0f;        // :194 refined_depth[~mask] = 0
    if (!(seen > 0.0f)) return false;                        // :315
    const float limit = a.depth_threshold * seen;        // :320 float32 product (NEP 50)
    return zc < (double)limit;                           // :319-328
}

// ==================================================================================================
// float64 kernel, second form (round 2): the image-bounds test WITHOUT the division, the grazing test second, and the
// reciprocal only for the pairs that reach the depth lookup.  Stage times of the first form on the 96-view ring scene
// (16.2 G pairs, DD_VOTES_STAGE builds): depth sign 5.3 ms, + projection and bounds 18.0 ms (every pair in front of the
// camera paid xc, yc, a reciprocal with two Newton steps, six more FMAs and the integer guards before the four
// comparisons), + grazing 10.4 ms (74 % of the pairs), + lookup 16.0 ms (34 %).
//
// With M = K [R|t] (rows 0 and 1; a per-view table built by votes_prepare64 in the caller's workspace) the reference's
//   u = K00 xc/den + K01 yc/den + K02 zc/den,  0 <= u < W      (den = zc + 1e-8 > 0, scripts/test.py:71-75, 300-302)
// is, in real arithmetic,  nu = M0 . (p, 1) >= 0  and  W den - nu > 0.  The float64 evaluation of nu and the reference's
// own float64 evaluation of u both stay within ~20 eps64 A of the real values, A = sum of the magnitudes of every
// product and intermediate sum involved <= coef S, with  coef = max_j (sum_i |K0i| |[R|t]_ij| + W |[R|t]_2j|)  (+ the
// 1e-8 term) per view and S = |x| + |y| + |z| + 1 per point.  A comparison is accepted when the value is farther than
// band = 1e-9 coef S from zero -- 4.5e5 times the rounding it could hide -- and the pair takes the exact formulation
// (the reference's three divisions) otherwise; NaN / inf compare false on both sides and land there too.  The pixel
// of a surviving pair is trunc(nu / den) by reciprocal + Newton unless within band / den of an integer (then exact).
// Same votes as the first form on every test (oracle, the reference's vote-loop fixture, 16.2 G-pair checksums).
// ==================================================================================================
struct Cam2 {
    f64x4 m0, m1, zr;           // M0, M1, [R|t] row z: fetched one view ahead into scalar registers
    cam_double *rest;           // the 32-double table row: [0..23] the caller's block, [21] band coefficient
};

__global__ __launch_bounds__(64) void votes_prepare64(const double *cams, double *tab, int V, int H, int W) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const double *c = cams + (size_t)v * 24;
    double *t = tab + (size_t)v * 32;
    for (int k = 0; k < 24; ++k) t[k] = c[k];
    double coef = 0.0;
    for (int j = 0; j < 4; ++j) {
        t[24 + j] = c[12] * c[j] + c[13] * c[4 + j] + c[14] * c[8 + j];
        t[28 + j] = c[15] * c[j] + c[16] * c[4 + j] + c[17] * c[8 + j];
        const double a0 = fabs(c[12]) * fabs(c[j]) + fabs(c[13]) * fabs(c[4 + j]) + fabs(c[14]) * fabs(c[8 + j]);
        const double a1 = fabs(c[15]) * fabs(c[j]) + fabs(c[16]) * fabs(c[4 + j]) + fabs(c[17]) * fabs(c[8 + j]);
        const double az = fabs(c[8 + j]) + (j == 3 ? 1e-8 : 0.0);
        coef = fmax(coef, fmax(a0 + (double)W * az, a1 + (double)H * az));
    }
    t[21] = 1.0000001e-9 * coef + 1e-300;      // NaN / inf camera entries make every band test fail -> exact formulation
}

__device__ __forceinline__ void exact_uw(cam_double *c, const double x, const double y, const double z, const double zc,
                                         const double den, double &u, double &w) {
    const double xc = c[0] * x + c[1] * y + c[2] * z + c[3];
    const double yc = c[4] * x + c[5] * y + c[6] * z + c[7];
    const double xn = xc / den, yn = yc / den, zn = zc / den;
    u = c[12] * xn + c[13] * yn + c[14] * zn;
    w = c[15] * xn + c[16] * yn + c[17] * zn;
}

__device__ __forceinline__ bool pair_votes2(const FArgs &a, const Cam2 &c, const int v, const double x, const double y,
                                            const double z, const double nx, const double ny, const double nz, const double S,
                                            const double wlim, const double hlim) {
    const double zc = c.zr[0] * x + c.zr[1] * y + c.zr[2] * z + c.zr[3];
    if (!(zc > 0.0)) return false;                           // :301 depths > 0
    const double den = zc + 1e-8;
    const double nu = c.m0[0] * x + c.m0[1] * y + c.m0[2] * z + c.m0[3];
    const double nw = c.m1[0] * x + c.m1[1] * y + c.m1[2] * z + c.m1[3];
    const double band = c.rest[21] * S;
    const double ru = fma(wlim, den, -nu), rw = fma(hlim, den, -nw);
    const double lo = fmin(fmin(nu, nw), fmin(ru, rw));      // NaN-free minimum is all that matters: NaN operands are caught below
    bool exact = false;
    double u = 0.0, w = 0.0;
    if (!(nu > band && nw > band && ru > band && rw > band)) {
        if (lo < -band) return false;                        // certainly outside the image
        exact_uw(c.rest, x, y, z, zc, den, u, w);            // inside a band (or not finite): the reference's formulation decides
        if (!(u >= 0.0 && u < wlim && w >= 0.0 && w < hlim)) return false;      // :300-302
        exact = true;
    }
    // :284-295  grazing-angle test (the filtered form of the first kernel)
    {
        double dx = x - c.rest[18], dy = y - c.rest[19], dz = z - c.rest[20];
        const double len2 = dx * dx + dy * dy + dz * dz;
        const double t = -(nx * dx + ny * dy + nz * dz), g = a.grazing_cos;
        const double s = t * t, q = g * g * len2;
        // one exit for the whole test (every `return` inside the view loop costs exec-mask bookkeeping on the scalar unit):
        // decided by the squared comparison unless the two sides are within 2.5e-9 of each other or not finite
        const bool sure = g > 0.0 && (t <= 0.0 || (fabs(s - q) > 2.5e-9 * (s + q) && s < 1e300));
        bool pass = t > 0.0 && s > q;
        if (!sure) {
            const double len = sqrt(len2);
            dx /= len; dy /= len; dz /= len;
            const double facing = nx * -dx + ny * -dy + nz * -dz;
            pass = facing > a.grazing_cos;
        }
        if (!pass) return false;
    }
    if (!exact) {
        double rden = __builtin_amdgcn_rcp(den);
        rden = fma(fma(-den, rden, 1.0), rden, rden);
        rden = fma(fma(-den, rden, 1.0), rden, rden);
        u = nu * rden; w = nw * rden;
        const double g = fma(band, rden, 1e-9);              // |u - u_reference| <= ~20 eps64 A / den, far inside band / den
        if (!(fabs(u - rint(u)) > g && fabs(w - rint(w)) > g)) exact_uw(c.rest, x, y, z, zc, den, u, w);
    }
    // :308-312  truncating lookup of the view's (mask-zeroed) refined depth
    const long long pix = (long long)v * a.hw + (long long)(int)w * a.W + (int)u;
    float seen = a.depth[pix];
    if (a.mask && a.mask[pix] == 0) seen = 0.0f;        // :194 refined_depth[~mask] = 0
    if (!(seen > 0.0f)) return false;                        // :315
    const float limit = a.depth_threshold * seen;        // :320 float32 product (NEP 50)
    return zc < (double)limit;                           // :319-328
}

// `decide`: NULL, or two counters written by votes_cull_estimate -- [0] (workgroup, view) cells that survive the cull,
// [1] cells tested.  The plain kernel runs when culling would remove less than 10 % of the cells, the culling kernel
// otherwise; both are launched and the one not chosen returns at once (no host round trip).
__device__ __forceinline__ bool cull_chosen(const unsigned long long *decide) {
    return decide[0] * 10ull < decide[1] * 9ull;      // the culling kernel costs 2-5 % when it removes nothing
}

__global__ __launch_bounds__(256) void floater_votes_kernel2(const FArgs a, const double *tab, const unsigned long long *decide) {
    if (decide && cull_chosen(decide)) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double wlim = (double)a.W, hlim = (double)a.H;
    const double x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
    const double nx = a.normal[3 * i], ny = a.normal[3 * i + 1], nz = a.normal[3 * i + 2];
    const double S = fabs(x) + fabs(y) + fabs(z) + 1.0;
    int votes = a.accumulate ? a.votes[i] : 0;
    auto fetch = [&](int v, Cam2 &c) {
        cam_vec4 *p = (cam_vec4 *)(tab + (size_t)v * 32);
        c.zr = p[2]; c.m0 = p[6]; c.m1 = p[7];
        c.rest = (cam_double *)(tab + (size_t)v * 32);
    };
    // two views per trip with the two register sets taking turns: `cur = nxt` cost twelve s_mov_b64 per view, and the
    // scalar unit (one per CU) was issuing as many instructions as the vector units (profiles/r02_pmc_votes.json)
    Cam2 ca, cb;
    fetch(0, ca);
    int v = 0;
    for (; v + 1 < a.V; v += 2) {
        fetch(v + 1, cb);
        votes += pair_votes2(a, ca, v, x, y, z, nx, ny, nz, S, wlim, hlim) ? 1 : 0;
        fetch(v + 2 < a.V ? v + 2 : v + 1, ca);
        votes += pair_votes2(a, cb, v + 1, x, y, z, nx, ny, nz, S, wlim, hlim) ? 1 : 0;
    }
    if (v < a.V) votes += pair_votes2(a, ca, v, x, y, z, nx, ny, nz, S, wlim, hlim) ? 1 : 0;
    a.votes[i] = votes;
}

// ==================================================================================================
// Per-workgroup view culling (round 2; the compile-time experiment of round 1 made a run-time choice).
//
// The exact per-pair conditions "in front of the camera and inside the image" (zc > 0, 0 <= u < W, 0 <= w < H with
// u = (K0 . Pcam) / (zc + 1e-8)) are, for zc > 0, linear inequalities in the world point p4 = (x, y, z, 1):
//   Z.p4 > 0;  U.p4 >= 0;  (U - W Z).p4 - W 1e-8 < 0;  Wv.p4 >= 0;  (Wv - H Z).p4 - H 1e-8 < 0
// (U, Wv, Z: rows K0 [R|t], K1 [R|t], [R|t]_z).  A workgroup's 256 points are 256 consecutive surviving pixels of one
// source view -- a short stretch of surface -- and lie inside a small sphere; when one of the five planes has the failing
// sign over the whole sphere, with a 1e-6 relative safety margin (ten orders of magnitude above the rounding of the
// exact evaluation), no point of the workgroup can collect a vote from that view and the view is skipped for all of
// them.  NaN / inf anywhere compares false and keeps the view, so the exact test still decides.  Votes are unchanged
// by construction (tests: equality with the un-culled kernels on every scene, the oracle, the reference's fixture).
//
// On an inward-facing ring every view sees almost every point: nothing to cull, and the culling kernel is 18-23 % slower
// than the plain one (LDS view lists, barriers, no camera prefetch).  On a scan whose views look at different parts of
// the scene -- what 2000 views of anything larger than a table top are -- most (workgroup, view) cells drop out.
// votes_cull_estimate measures the surviving fraction on a sample of workgroups and the kernels choose on the device.
// ==================================================================================================
constexpr int PLANE_STRIDE = 32;      // doubles per view: 5 planes x (nx, ny, nz, d) [0..19], |normal| [20..24], |d| [25..29]

__global__ __launch_bounds__(64) void votes_prepare_planes(const double *cams, double *planes, int V, int H, int W) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V) return;
    const double *c = cams + (size_t)v * 24;
    double *o = planes + (size_t)v * PLANE_STRIDE;
    const double wlim = (double)W, hlim = (double)H;
    double P[5][4];
    for (int k = 0; k < 4; ++k) {
        const double X = c[k], Y = c[4 + k], Z = c[8 + k];
        const double U = c[12] * X + c[13] * Y + c[14] * Z, Wv = c[15] * X + c[16] * Y + c[17] * Z;
        P[0][k] = Z; P[1][k] = U; P[2][k] = U - wlim * Z; P[3][k] = Wv; P[4][k] = Wv - hlim * Z;
    }
    P[2][3] -= wlim * 1e-8;
    P[4][3] -= hlim * 1e-8;
    for (int k = 0; k < 5; ++k) {
        for (int j = 0; j < 4; ++j) o[4 * k + j] = P[k][j];
        o[20 + k] = sqrt(P[k][0] * P[k][0] + P[k][1] * P[k][1] + P[k][2] * P[k][2]);
        o[25 + k] = fabs(P[k][3]);
    }
    o[30] = 0.0; o[31] = 0.0;
}

// can ANY point of the sphere (centre, radius r; cnorm = |cx| + |cy| + |cz|) pass the five conditions of view `o`?
__device__ __forceinline__ bool sphere_may_project(const double *o, const double cx, const double cy, const double cz,
                                                   const double r, const double cnorm) {
    bool out = false;
    // NOT unrolled: with the five planes' 30 doubles loaded up front the culling kernel needed 108 VGPRs (4 waves per
    // SIMD); this test runs once per workgroup and 256 views, the pair loop that follows needs the occupancy
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
        const double an = o[20 + k];
        const double f = o[4 * k] * cx + o[4 * k + 1] * cy + o[4 * k + 2] * cz + o[4 * k + 3];
        const double margin = 1e-6 * (an * (cnorm + r) + o[25 + k]);
        if (k == 0 || k == 1 || k == 3) out = out || (f + an * r < -margin);      // needs value > 0 / >= 0: negative everywhere
        else out = out || (f - an * r > margin);                                  // needs value < 0: positive everywhere
    }
    return !out;
}

// bounding sphere of the workgroup's points (fminf / fmaxf skip NaN: idle lanes and NaN points do not count -- a NaN
// point never votes anyway; an infinite coordinate makes r infinite or NaN and every view is kept)
__device__ __forceinline__ void tile_sphere(const float fx, const float fy, const float fz, float (*s_box)[6], double &cx, double &cy,
                                            double &cz, double &r, double &cnorm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lo0 = fx, lo1 = fy, lo2 = fz, hi0 = fx, hi1 = fy, hi2 = fz;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo0 = fminf(lo0, __shfl_xor(lo0, o)); lo1 = fminf(lo1, __shfl_xor(lo1, o)); lo2 = fminf(lo2, __shfl_xor(lo2, o));
        hi0 = fmaxf(hi0, __shfl_xor(hi0, o)); hi1 = fmaxf(hi1, __shfl_xor(hi1, o)); hi2 = fmaxf(hi2, __shfl_xor(hi2, o));
    }
    if (lane == 0) { s_box[wave][0] = lo0; s_box[wave][1] = lo1; s_box[wave][2] = lo2; s_box[wave][3] = hi0; s_box[wave][4] = hi1; s_box[wave][5] = hi2; }
    __syncthreads();
    double bl[3], bh[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bl[k] = fminf(fminf(s_box[0][k], s_box[1][k]), fminf(s_box[2][k], s_box[3][k]));
        bh[k] = fmaxf(fmaxf(s_box[0][k + 3], s_box[1][k + 3]), fmaxf(s_box[2][k + 3], s_box[3][k + 3]));
    }
    cx = 0.5 * (bl[0] + bh[0]); cy = 0.5 * (bl[1] + bh[1]); cz = 0.5 * (bl[2] + bh[2]);
    const double ex = bh[0] - bl[0], ey = bh[1] - bl[1], ez = bh[2] - bl[2];
    r = 0.5 * sqrt(ex * ex + ey * ey + ez * ez) * (1.0 + 1e-9);                   // half diagonal: every point is inside
    cnorm = fabs(cx) + fabs(cy) + fabs(cz);
}

// a sample of the workgroups (every `stride`-th): how many (workgroup, view) cells survive the cull?
__global__ __launch_bounds__(256) void votes_cull_estimate(const FArgs a, const double *planes, unsigned long long *decide, const long long stride) {
    __shared__ float s_box[4][6];
    const long long i = (long long)blockIdx.x * stride * 256 + threadIdx.x;
    const bool live = i < a.n;
    const float qnan = __builtin_nanf("");
    const float fx = live ? a.xyz[3 * i] : qnan, fy = live ? a.xyz[3 * i + 1] : qnan, fz = live ? a.xyz[3 * i + 2] : qnan;
    double cx, cy, cz, r, cnorm;
    tile_sphere(fx, fy, fz, s_box, cx, cy, cz, r, cnorm);
    unsigned kept = 0;
    for (int v = (int)threadIdx.x; v < a.V; v += 256) kept += sphere_may_project(planes + (size_t)v * PLANE_STRIDE, cx, cy, cz, r, cnorm) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&decide[0], (unsigned long long)kept);
        if (threadIdx.x == 0) atomicAdd(&decide[1], (unsigned long long)a.V);
    }
}

// Level 1 of the cull (scans of thousands of views, where V sphere tests per workgroup would start to dominate): a
// SUPER-TILE is 256 workgroups = 65 536 consecutive points (a band of a few dozen image rows of one source view); its
// bounding sphere is tested against every view once and the survivors are kept as a bit mask of V bits.  A workgroup
// then tests its own, much smaller sphere only against the views of its super-tile's mask, and skips chunks of 256
// views whose four mask words are zero without touching the plane table.
constexpr int SUPER = 256;            // workgroups per super-tile

__global__ __launch_bounds__(256) void votes_supertile_masks(const FArgs a, const double *planes, unsigned long long *masks,
                                                             const int words, const unsigned long long *decide) {
    if (decide && !cull_chosen(decide)) return;
    __shared__ float s_box[4][6];
    const long long base = (long long)blockIdx.x * SUPER * 256;
    const float qnan = __builtin_nanf("");
    float lo0 = qnan, lo1 = qnan, lo2 = qnan, hi0 = qnan, hi1 = qnan, hi2 = qnan;
    for (int k = 0; k < SUPER; ++k) {
        const long long i = base + (long long)k * 256 + threadIdx.x;
        if (i >= a.n) break;
        const float fx = a.xyz[3 * i], fy = a.xyz[3 * i + 1], fz = a.xyz[3 * i + 2];
        lo0 = fminf(lo0, fx); lo1 = fminf(lo1, fy); lo2 = fminf(lo2, fz);
        hi0 = fmaxf(hi0, fx); hi1 = fmaxf(hi1, fy); hi2 = fmaxf(hi2, fz);
    }
    // tile_sphere reduces one point per lane: feed it the two corners of this lane's box in turn and join the results
    double c0[3], c1[3], r0, r1, n0, n1;
    tile_sphere(lo0, lo1, lo2, s_box, c0[0], c0[1], c0[2], r0, n0);
    __syncthreads();
    const float l0 = fminf(fminf(s_box[0][0], s_box[1][0]), fminf(s_box[2][0], s_box[3][0]));
    const float l1 = fminf(fminf(s_box[0][1], s_box[1][1]), fminf(s_box[2][1], s_box[3][1]));
    const float l2 = fminf(fminf(s_box[0][2], s_box[1][2]), fminf(s_box[2][2], s_box[3][2]));
    __syncthreads();
    tile_sphere(hi0, hi1, hi2, s_box, c1[0], c1[1], c1[2], r1, n1);
    const float h0 = fmaxf(fmaxf(s_box[0][3], s_box[1][3]), fmaxf(s_box[2][3], s_box[3][3]));
    const float h1 = fmaxf(fmaxf(s_box[0][4], s_box[1][4]), fmaxf(s_box[2][4], s_box[3][4]));
    const float h2 = fmaxf(fmaxf(s_box[0][5], s_box[1][5]), fmaxf(s_box[2][5], s_box[3][5]));
    const double cx = 0.5 * ((double)l0 + h0), cy = 0.5 * ((double)l1 + h1), cz = 0.5 * ((double)l2 + h2);
    const double ex = (double)h0 - l0, ey = (double)h1 - l1, ez = (double)h2 - l2;
    const double r = 0.5 * sqrt(ex * ex + ey * ey + ez * ez) * (1.0 + 1e-9);
    const double cnorm = fabs(cx) + fabs(cy) + fabs(cz);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v0 = 0; v0 < a.V; v0 += 256) {
        const int mine = v0 + (int)threadIdx.x;
        // an empty box (no finite coordinate at all: NaN centre) compares false everywhere -> every view is kept
        const bool keep = mine < a.V && sphere_may_project(planes + (size_t)mine * PLANE_STRIDE, cx, cy, cz, r, cnorm);
        const unsigned long long b = __ballot(keep);
        const int word = (v0 >> 6) + wave;
        if (lane == 0 && word < words) masks[(size_t)blockIdx.x * words + word] = b;
    }
}

__global__ __launch_bounds__(256) void floater_votes_kernel_cull(const FArgs a, const double *tab, const double *planes,
                                                                 const unsigned long long *decide, const unsigned long long *masks,
                                                                 const int words) {
    if (decide && !cull_chosen(decide)) return;
    __shared__ float s_box[4][6];
    __shared__ unsigned short s_views[256];
    __shared__ int s_cnt[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.n;
    const double wlim = (double)a.W, hlim = (double)a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float qnan = __builtin_nanf("");
    const float fx = live ? a.xyz[3 * i] : qnan, fy = live ? a.xyz[3 * i + 1] : qnan, fz = live ? a.xyz[3 * i + 2] : qnan;
    const double x = fx, y = fy, z = fz;
    double nx = 0, ny = 0, nz = 0;
    if (live) { nx = a.normal[3 * i]; ny = a.normal[3 * i + 1]; nz = a.normal[3 * i + 2]; }
    const double S = fabs(x) + fabs(y) + fabs(z) + 1.0;
    __shared__ double s_sphere[5];
    {
        double cx, cy, cz, r, cnorm;
        tile_sphere(fx, fy, fz, s_box, cx, cy, cz, r, cnorm);
        // parked in LDS and re-read by phase A of every chunk: held in registers across the pair loop, the five
        // (uniform) doubles cost 10 VGPRs per lane and a step of occupancy
        if (threadIdx.x == 0) { s_sphere[0] = cx; s_sphere[1] = cy; s_sphere[2] = cz; s_sphere[3] = r; s_sphere[4] = cnorm; }
    }
    __syncthreads();
    int votes = (live && a.accumulate) ? a.votes[i] : 0;
    const unsigned long long *my_masks = masks ? masks + (size_t)(blockIdx.x / SUPER) * words : nullptr;
    for (int v0 = 0; v0 < a.V; v0 += 256) {
        // level 1: the views this workgroup's super-tile can see at all (wave-uniform words -> scalar loads)
        unsigned long long m = ~0ull;
        if (my_masks) {
            const int w0 = v0 >> 6;
            const unsigned long long m0 = my_masks[w0], m1 = w0 + 1 < words ? my_masks[w0 + 1] : 0ull;
            const unsigned long long m2 = w0 + 2 < words ? my_masks[w0 + 2] : 0ull, m3 = w0 + 3 < words ? my_masks[w0 + 3] : 0ull;
            if ((m0 | m1 | m2 | m3) == 0ull) continue;       // the same words for the whole workgroup: nobody reaches a barrier
            m = wave == 0 ? m0 : wave == 1 ? m1 : wave == 2 ? m2 : m3;
        }
        // phase A: one lane per view of this chunk decides whether the workgroup can see it at all
        const int mine = v0 + (int)threadIdx.x;
        const bool keep = mine < a.V && ((m >> lane) & 1ull) &&
                          sphere_may_project(planes + (size_t)mine * PLANE_STRIDE, s_sphere[0], s_sphere[1], s_sphere[2], s_sphere[3], s_sphere[4]);
        const unsigned long long b = __ballot(keep);
        if (keep) s_views[wave * 64 + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)threadIdx.x;
        if (lane == 0) s_cnt[wave] = __popcll(b);
        __syncthreads();
        // phase B: every lane runs the exact test against the surviving views only
        if (live) {
            for (int w = 0; w < 4; ++w) {
                const int cnt = __builtin_amdgcn_readfirstlane(s_cnt[w]);
                for (int k = 0; k < cnt; ++k) {
                    const int v = v0 + __builtin_amdgcn_readfirstlane((int)s_views[w * 64 + k]);      // uniform: scalar loads
                    cam_vec4 *p = (cam_vec4 *)(tab + (size_t)v * 32);
                    Cam2 c;
                    c.zr = p[2]; c.m0 = p[6]; c.m1 = p[7];
                    c.rest = (cam_double *)(tab + (size_t)v * 32);
                    votes += pair_votes2(a, c, v, x, y, z, nx, ny, nz, S, wlim, hlim) ? 1 : 0;
                }
            }
        }
        __syncthreads();
    }
    if (live) a.votes[i] = votes;
}

// (Round 2 also carried a float32 first pass with rigorous error bounds whose undecided pairs were resolved in float64: the same
// votes bit for bit, but 0.7-0.8x the rate of the float64 kernels -- on gfx950 v_fma_f64 issues at 0.88x the rate of a scalar-
// per-lane v_fma_f32, so an evaluation that needs twice the instructions for its bounds has nothing to collect
// (tools/experiments/ubench_fma.hip).  Removed in round 3; the history is in git, the measurements in DESIGN.md section 7.)

// ==================================================================================================
// Stable compaction of the fused cloud by the vote test (scripts/test.py:330-332:
// `keep = votes < vote_threshold; points = points[keep]; colors = colors[keep]`), every per-point field.
// Two passes, no inter-workgroup dependency: (1) kept rows per 4096-row tile, (2) two small scans,
// (3) one lane per INPUT row: each wave owns 1024 consecutive rows, ranks its kept rows with ballots
// and copies them to their final position (reads perfectly coalesced, writes contiguous up to the
// gaps the dropped rows leave).  New view offsets = number of kept rows before each old offset.
// ==================================================================================================
constexpr int C_BLOCK = 256, C_WAVES = 4, C_PER_LANE = 16, C_WSPAN = 64 * C_PER_LANE, C_TILE = C_WAVES * C_WSPAN;
constexpr int C_GROUP = 1024;       // tiles per first-level scan group

typedef float cf3 __attribute__((ext_vector_type(3)));
typedef unsigned cu1u __attribute__((aligned(1)));

struct CArgs {
    const int32_t *votes;
    const float *xyz, *normal;
    const uint8_t *rgb;
    const int32_t *pix, *view;
    float *o_xyz, *o_normal;
    uint8_t *o_rgb;
    int32_t *o_pix, *o_view;
    uint32_t *o_packed;                  // (capacity,4) dwords x, y, z, r|g<<8|b<<16|255<<24 or NULL
    long long capacity;                  // rows at or beyond it are counted, not written
    long long n;
    int thr;
    unsigned num_tiles, num_groups;
    unsigned *tile_cnt, *tile_off;       // kept rows per tile; offset of the tile inside its group
    long long *group_tot, *group_off;    // kept rows per group; absolute offset of the group
    long long *kept;                     // (1) total kept rows
    const long long *old_offsets;        // (V+1) or NULL
    long long *new_offsets;
    int V;
};

__global__ __launch_bounds__(C_BLOCK) void compact_count(const CArgs a) {
    __shared__ unsigned s_w[C_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * C_TILE + wave * C_WSPAN;
    unsigned cnt = 0;
#pragma unroll
    for (int i = 0; i < C_PER_LANE; ++i) {
        const long long r = base + i * 64 + lane;
        const bool keep = r < a.n && a.votes[r] < a.thr;
        cnt += (unsigned)__popcll(__ballot(keep));
    }
    if (lane == 0) s_w[wave] = cnt;
    __syncthreads();
    if (tid == 0) a.tile_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of up to C_GROUP tile counts per workgroup (one group each) + the group's total
__global__ __launch_bounds__(C_BLOCK) void compact_scan_groups(const CArgs a) {
    __shared__ unsigned s_w[C_WAVES];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned g0 = blockIdx.x * C_GROUP;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (unsigned b = 0; b < C_GROUP; b += C_BLOCK) {
        const unsigned i = g0 + b + tid;
        const unsigned x = i < a.num_tiles ? a.tile_cnt[i] : 0u;
        unsigned incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        unsigned pre = s_carry;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        if (i < a.num_tiles) a.tile_off[i] = pre + incl - x;
        __syncthreads();
        if (tid == C_BLOCK - 1) s_carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0) a.group_tot[blockIdx.x] = (long long)s_carry;
}

// exclusive scan of the group totals (one workgroup), total kept rows, and the new view offsets
__global__ __launch_bounds__(C_BLOCK) void compact_scan_top(const CArgs a) {
    __shared__ long long s_w[C_WAVES];
    __shared__ long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (unsigned b = 0; b < a.num_groups; b += C_BLOCK) {
        const unsigned i = b + tid;
        const long long x = i < a.num_groups ? a.group_tot[i] : 0ll;
        long long incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        long long pre = s_carry;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        if (i < a.num_groups) a.group_off[i] = pre + incl - x;
        __syncthreads();
        if (tid == C_BLOCK - 1) s_carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0) *a.kept = s_carry;
}

// new_offsets[v] = kept rows in [0, old_offsets[v]): the scanned tile prefix plus the kept rows of the partial tile,
// counted by one wave per view offset (coalesced loads, ballots) -- a serial walk over up to 4095 rows cost
// 280 us per call for a dozen views
__global__ __launch_bounds__(64) void compact_view_offsets(const CArgs a) {
    const int v = blockIdx.x, lane = threadIdx.x;
    const long long row = a.old_offsets[v];
    const long long t = row / C_TILE;
    long long kept;
    if (t >= (long long)a.num_tiles) {
        kept = *a.kept;
    } else {
        kept = a.group_off[t / C_GROUP] + a.tile_off[t];
        for (long long r0 = t * C_TILE; r0 < row; r0 += 64) {
            const long long r = r0 + lane;
            kept += __popcll(__ballot(r < row && a.votes[r] < a.thr));
        }
    }
    if (lane == 0) a.new_offsets[v] = kept;
}

__global__ __launch_bounds__(C_BLOCK) void compact_scatter(const CArgs a) {
    __shared__ unsigned s_w[C_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned t = blockIdx.x;
    const long long base = (long long)t * C_TILE + wave * C_WSPAN;
    unsigned long long bal[C_PER_LANE];
    unsigned cnt = 0;
#pragma unroll
    for (int i = 0; i < C_PER_LANE; ++i) {
        const long long r = base + i * 64 + lane;
        bal[i] = __ballot(r < a.n && a.votes[r] < a.thr);
        cnt += (unsigned)__popcll(bal[i]);
    }
    if (lane == 0) s_w[wave] = cnt;
    __syncthreads();
    long long dst = a.group_off[t / C_GROUP] + a.tile_off[t];
    for (int w = 0; w < wave; ++w) dst += s_w[w];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < C_PER_LANE; ++i) {
        const long long r = base + i * 64 + lane;
        if ((bal[i] >> lane) & 1ull) {
            const long long o = dst + __popcll(bal[i] & lt);
            if (o < a.capacity) {
                const cf3 p = *reinterpret_cast<const cf3 *>(a.xyz + r * 3);
                if (a.o_xyz) *reinterpret_cast<cf3 *>(a.o_xyz + o * 3) = p;
                if (a.o_normal) *reinterpret_cast<cf3 *>(a.o_normal + o * 3) = *reinterpret_cast<const cf3 *>(a.normal + r * 3);
                unsigned c = 0;
                if (a.rgb && (a.o_rgb || a.o_packed))
                    c = r ? (*reinterpret_cast<const cu1u *>(a.rgb + r * 3 - 1) >> 8)
                          : (*reinterpret_cast<const cu1u *>(a.rgb) & 0xffffffu);
                if (a.o_rgb) {
                    uint8_t *d = a.o_rgb + o * 3;
                    d[0] = (uint8_t)c; d[1] = (uint8_t)(c >> 8); d[2] = (uint8_t)(c >> 16);
                }
                if (a.o_packed) {
                    uint4 rec; rec.x = __float_as_uint(p.x); rec.y = __float_as_uint(p.y); rec.z = __float_as_uint(p.z); rec.w = c | 0xff000000u;
                    *reinterpret_cast<uint4 *>(a.o_packed + o * 4) = rec;
                }
                if (a.o_pix) a.o_pix[o] = a.pix[r];
                if (a.o_view) a.o_view[o] = a.view[r];
            }
        }
        dst += __popcll(bal[i]);
    }
}

thread_local char g_ferr[192] = "";

}  // namespace

extern "C" {

const char *dd_filter_last_error(void) { return g_ferr; }

int64_t dd_votes_workspace_bytes(int32_t num_views, int64_t n_points) {
    if (num_views <= 0 || n_points < 0) return DD_ERR_INVALID_ARG;
    if (n_points > ((int64_t)1 << 48)) return DD_ERR_UNSUPPORTED;      // (beyond any memory: keeps the arithmetic below inside 64 bits)
    const int64_t blocks = (n_points + 255) / 256;
    // two 256-byte tables per view, the decision counters, one mask of V bits per 65 536 points (level 1 of the cull)
    return (int64_t)num_views * 512 + 64 + ((blocks + SUPER - 1) / SUPER) * (((int64_t)num_views + 63) / 64) * 8;
}

int dd_floater_votes(const DDFilterViews *views, const float *xyz, const float *normal, int64_t n,
                     int32_t *votes_dev, int32_t accumulate, void *stream) {
    auto fail = [](const char *m) { snprintf(g_ferr, sizeof(g_ferr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (!views) return fail("views is NULL");
    if (views->num_views <= 0 || views->height <= 0 || views->width <= 0) return fail("num_views/height/width must be positive");
    if (!views->depth || !views->cams) return fail("depth / cams is NULL");
    if (n < 0) return fail("n is negative");
    if (n > ((int64_t)1 << 48) || views->num_views > (1 << 24)) return fail("more than 2^48 points or 2^24 views");      // (keeps the size arithmetic below inside its types)
    if (views->mode < 0 || views->mode > 4) return fail("mode must be 0 .. 4");
    if (views->mode == 2) return fail("mode 2 (the float32 first pass and its verify build) was removed in ABI 9");
    if (n > 0 && (!xyz || !normal || !votes_dev)) return fail("xyz / normal / votes_dev is NULL");
    if (n == 0) return DD_OK;
    FArgs a{};
    a.xyz = xyz; a.normal = normal; a.depth = views->depth; a.mask = views->mask; a.cams = views->cams;
    a.votes = votes_dev; a.n = n; a.hw = (long long)views->height * views->width;
    a.V = views->num_views; a.H = views->height; a.W = views->width; a.accumulate = accumulate;
    a.depth_threshold = views->depth_threshold; a.grazing_cos = views->grazing_cos;
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffll) return fail("too many points for one launch; split the call");
    hipStream_t s = (hipStream_t)stream;
    const int V = views->num_views;
    // mode 0 (a zero-initialised struct) = the best the workspace allows: 4 (culling chosen on the device) with
    // 512 * V + 64 bytes, 1 (table kernel) with 256 * V
    int mode = views->mode;
    if (mode == 0) mode = (views->workspace && views->workspace_bytes >= (int64_t)V * 512 + 64) ? 4 : 1;
    if (mode == 3 || mode == 4) {
        // float64 with per-workgroup view culling: 3 = always, 4 = chosen on the device from a sample of the workgroups
        if (!views->workspace || views->workspace_bytes < (int64_t)V * 512 + 64) return fail("modes 3 / 4 need a workspace of 512 * num_views + 64 bytes");
        if (((uintptr_t)views->workspace % 32) != 0) return fail("workspace must be 32-byte aligned");
        double *tab = reinterpret_cast<double *>(views->workspace);
        double *planes = tab + (size_t)V * 32;
        unsigned long long *decide = reinterpret_cast<unsigned long long *>(planes + (size_t)V * PLANE_STRIDE);
        // with room for them: one mask of V bits per super-tile of 65 536 points (level 1 of the cull)
        const long long supers = (blocks + SUPER - 1) / SUPER;
        const int words = (V + 63) / 64;
        unsigned long long *masks = decide + 8;
        const bool two_level = views->workspace_bytes >= (int64_t)V * 512 + 64 + supers * words * 8;
        hipLaunchKernelGGL(votes_prepare64, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, s, views->cams, tab, V, views->height, views->width);
        hipLaunchKernelGGL(votes_prepare_planes, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, s, views->cams, planes, V, views->height, views->width);
        if (mode == 4) {
            if (hipMemsetAsync(decide, 0, 16, s) != hipSuccess) { snprintf(g_ferr, sizeof(g_ferr), "hipMemsetAsync(decide) failed"); return DD_ERR_LAUNCH; }
            const long long sample = blocks < 512 ? blocks : 512, stride = blocks / sample;
            hipLaunchKernelGGL(votes_cull_estimate, dim3((unsigned)sample), dim3(256), 0, s, a, (const double *)planes, decide, stride);
            hipLaunchKernelGGL(floater_votes_kernel2, dim3((unsigned)blocks), dim3(256), 0, s, a, (const double *)tab, (const unsigned long long *)decide);
        } else decide = nullptr;
        if (two_level) hipLaunchKernelGGL(votes_supertile_masks, dim3((unsigned)supers), dim3(256), 0, s, a, (const double *)planes, masks, words,
                                          (const unsigned long long *)decide);
        hipLaunchKernelGGL(floater_votes_kernel_cull, dim3((unsigned)blocks), dim3(256), 0, s, a, (const double *)tab, (const double *)planes,
                           (const unsigned long long *)decide, (const unsigned long long *)(two_level ? masks : nullptr), words);
    } else if (views->workspace && views->workspace_bytes >= (int64_t)V * 256) {
        if (((uintptr_t)views->workspace % 32) != 0) return fail("workspace must be 32-byte aligned");
        double *tab = reinterpret_cast<double *>(views->workspace);
        hipLaunchKernelGGL(votes_prepare64, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, s, views->cams, tab, V, views->height, views->width);
        hipLaunchKernelGGL(floater_votes_kernel2, dim3((unsigned)blocks), dim3(256), 0, s, a, (const double *)tab, (const unsigned long long *)nullptr);
    } else
        return fail("a workspace of at least 256 * num_views bytes is required (ABI 10: the table-free kernel of round 1 is gone)");
    if (hipGetLastError() != hipSuccess) { snprintf(g_ferr, sizeof(g_ferr), "floater_votes launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

int64_t dd_compact_workspace_bytes(int64_t n) {
    if (n < 0) return DD_ERR_INVALID_ARG;
    if (n > ((int64_t)1 << 48)) return DD_ERR_UNSUPPORTED;
    const int64_t tiles = (n + C_TILE - 1) / C_TILE, groups = (tiles + C_GROUP - 1) / C_GROUP;
    return 16 + tiles * 8 + groups * 16;
}

int dd_compact_cloud(const DDCloudOut *in, int64_t n, const int32_t *votes_dev, int32_t vote_threshold,
                     const DDCloudOut *out, int64_t *kept_dev, const int64_t *old_view_offsets_dev,
                     int64_t *new_view_offsets_dev, int32_t num_views, void *workspace, int64_t workspace_bytes,
                     void *stream) {
    auto fail = [](const char *m) { snprintf(g_ferr, sizeof(g_ferr), "%s", m); return DD_ERR_INVALID_ARG; };
    if (!in || !out || !in->xyz || (!out->xyz && !out->xyz_rgba)) return fail("in / out / xyz is NULL");
    if (out->xyz_rgba && ((uintptr_t)out->xyz_rgba % 16) != 0) return fail("out->xyz_rgba must be 16-byte aligned");
    if (n < 0 || !kept_dev) return fail("n is negative or kept_dev is NULL");
    if (n > 0 && !votes_dev) return fail("votes_dev is NULL");
    if ((out->normal && !in->normal) || (out->rgb && !in->rgb) || (out->pixel_index && !in->pixel_index) ||
        (out->view_index && !in->view_index)) return fail("an output field has no input field");
    if (out->capacity < 0) return fail("out->capacity is negative");
    if ((old_view_offsets_dev == nullptr) != (new_view_offsets_dev == nullptr)) return fail("old/new view offsets must be given together");
    const int64_t need = dd_compact_workspace_bytes(n);
    if (need < 0) return fail("more than 2^48 rows");
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 16)) {
        snprintf(g_ferr, sizeof(g_ferr), "workspace is NULL, mis-aligned or smaller than dd_compact_workspace_bytes()");
        return DD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (n + C_TILE - 1) / C_TILE, groups = (tiles + C_GROUP - 1) / C_GROUP;
    if (tiles > 0x7fffffffll) return fail("too many rows for one call");
    CArgs a{};
    a.votes = votes_dev; a.xyz = in->xyz; a.normal = in->normal; a.rgb = in->rgb; a.pix = in->pixel_index; a.view = in->view_index;
    a.o_xyz = out->xyz; a.o_normal = out->normal; a.o_rgb = out->rgb; a.o_pix = out->pixel_index; a.o_view = out->view_index;
    a.o_packed = out->xyz_rgba; a.capacity = out->capacity;
    a.n = n; a.thr = vote_threshold; a.num_tiles = (unsigned)tiles; a.num_groups = (unsigned)groups;
    char *w = reinterpret_cast<char *>(workspace) + 16;
    a.tile_cnt = reinterpret_cast<unsigned *>(w);
    a.tile_off = a.tile_cnt + tiles;
    a.group_tot = reinterpret_cast<long long *>(w + tiles * 8);
    a.group_off = a.group_tot + groups;
    a.kept = reinterpret_cast<long long *>(kept_dev);
    a.old_offsets = reinterpret_cast<const long long *>(old_view_offsets_dev);
    a.new_offsets = reinterpret_cast<long long *>(new_view_offsets_dev);
    a.V = num_views;
    if (n == 0) {
        if (hipMemsetAsync(kept_dev, 0, 8, s) != hipSuccess) return DD_ERR_LAUNCH;
        if (new_view_offsets_dev && hipMemsetAsync(new_view_offsets_dev, 0, 8 * (size_t)(num_views + 1), s) != hipSuccess) return DD_ERR_LAUNCH;
        return DD_OK;
    }
    hipLaunchKernelGGL(compact_count, dim3((unsigned)tiles), dim3(C_BLOCK), 0, s, a);
    hipLaunchKernelGGL(compact_scan_groups, dim3((unsigned)groups), dim3(C_BLOCK), 0, s, a);
    hipLaunchKernelGGL(compact_scan_top, dim3(1), dim3(C_BLOCK), 0, s, a);
    if (new_view_offsets_dev)
        hipLaunchKernelGGL(compact_view_offsets, dim3((unsigned)(num_views + 1)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(compact_scatter, dim3((unsigned)tiles), dim3(C_BLOCK), 0, s, a);
    if (hipGetLastError() != hipSuccess) { snprintf(g_ferr, sizeof(g_ferr), "compact_cloud launch failed"); return DD_ERR_LAUNCH; }
    return DD_OK;
}

}  // extern "C"
