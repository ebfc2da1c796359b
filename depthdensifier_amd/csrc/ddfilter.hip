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

// One (point, view) pair: true when the view votes against the point.  `c` is the view's camera block.
__device__ __forceinline__ bool pair_votes(const FArgs &a, const double *c, const int v, const double x, const double y,
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

#ifndef DD_VOTES_CULL
#define DD_VOTES_CULL 0      // 1 = per-workgroup view culling (bounding sphere vs frustum).  Measured on MI355X: identical
                             // votes but 18-23 % SLOWER on the ring scene (rejected pairs are already cheap in the exact
                             // test and skip a whole wave at a time; the cull adds LDS-indexed view lookups + barriers)
#endif

#if DD_VOTES_CULL
// Can ANY point of the sphere (centre cx,cy,cz, radius r) pass "in front of the camera and inside the image" of the
// view with camera block c?  The exact per-pair conditions (zc > 0, 0 <= u < W, 0 <= w < H with
// u = (K0 . Pcam) / (zc + 1e-8)) are, for zc > 0, linear inequalities in the world point:
//   zc > 0;  U.p >= 0;  (U - W Z).p - W 1e-8 < 0;  Wv.p >= 0;  (Wv - H Z).p - H 1e-8 < 0
// (U, Wv, Z: rows K0 [R|t], K1 [R|t], [R|t]_z as 4-vectors, p = (x,y,z,1)).  A plane whose value over the whole
// sphere has the failing sign -- with a 1e-6 relative safety margin, ten orders of magnitude above the rounding
// of the exact evaluation -- rejects the view for every point of the workgroup.  NaN / inf anywhere compares
// false and keeps the view, so the exact test still decides.
__device__ __forceinline__ bool sphere_may_project(const double *c, const double cx, const double cy, const double cz,
                                                   const double r, const double cnorm, const double wlim, const double hlim) {
    double P[5][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double X = c[k], Y = c[4 + k], Z = c[8 + k];
        const double U = c[12] * X + c[13] * Y + c[14] * Z, Wv = c[15] * X + c[16] * Y + c[17] * Z;
        P[0][k] = Z; P[1][k] = U; P[2][k] = U - wlim * Z; P[3][k] = Wv; P[4][k] = Wv - hlim * Z;
    }
    P[2][3] -= wlim * 1e-8;
    P[4][3] -= hlim * 1e-8;
    bool out = false;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double an = sqrt(P[k][0] * P[k][0] + P[k][1] * P[k][1] + P[k][2] * P[k][2]);
        const double f = P[k][0] * cx + P[k][1] * cy + P[k][2] * cz + P[k][3];
        const double margin = 1e-6 * (an * (cnorm + r) + fabs(P[k][3]));
        if (k == 0 || k == 1 || k == 3) out = out || (f + an * r < -margin);      // needs value > 0 / >= 0: all negative
        else out = out || (f - an * r > margin);                                  // needs value < 0: all positive
    }
    return !out;
}
#endif

__global__ __launch_bounds__(256) void floater_votes_kernel(const FArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.n;
    const double wlim = (double)a.W, hlim = (double)a.H;
#if DD_VOTES_CULL
    __shared__ float s_box[4][6];
    __shared__ unsigned short s_views[256];
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float qnan = __builtin_nanf("");
    const float fx = live ? a.xyz[3 * i] : qnan, fy = live ? a.xyz[3 * i + 1] : qnan, fz = live ? a.xyz[3 * i + 2] : qnan;
    const double x = fx, y = fy, z = fz;
    double nx = 0, ny = 0, nz = 0;
    if (live) { nx = a.normal[3 * i]; ny = a.normal[3 * i + 1]; nz = a.normal[3 * i + 2]; }
    // bounding box of the workgroup's points (fminf / fmaxf skip NaN: idle lanes and NaN points do not count;
    // a NaN point never votes anyway)
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
    const double cx = 0.5 * (bl[0] + bh[0]), cy = 0.5 * (bl[1] + bh[1]), cz = 0.5 * (bl[2] + bh[2]);
    const double ex = bh[0] - bl[0], ey = bh[1] - bl[1], ez = bh[2] - bl[2];
    const double r = 0.5 * sqrt(ex * ex + ey * ey + ez * ez) * (1.0 + 1e-9);     // half diagonal: every point is inside
    const double cnorm = fabs(cx) + fabs(cy) + fabs(cz);
    int votes = (live && a.accumulate) ? a.votes[i] : 0;
    for (int v0 = 0; v0 < a.V; v0 += 256) {
        // phase A: one lane per view of this chunk decides whether the workgroup can see it at all
        const int mine = v0 + (int)threadIdx.x;
        const bool keep = mine < a.V && sphere_may_project(a.cams + (size_t)mine * 24, cx, cy, cz, r, cnorm, wlim, hlim);
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
                    votes += pair_votes(a, a.cams + (size_t)v * 24, v, x, y, z, nx, ny, nz, wlim, hlim) ? 1 : 0;
                }
            }
        }
        __syncthreads();
    }
    if (live) a.votes[i] = votes;
#else
    if (!live) return;
    const double x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
    const double nx = a.normal[3 * i], ny = a.normal[3 * i + 1], nz = a.normal[3 * i + 2];
    int votes = a.accumulate ? a.votes[i] : 0;
    for (int v = 0; v < a.V; ++v)
        votes += pair_votes(a, a.cams + (size_t)v * 24, v, x, y, z, nx, ny, nz, wlim, hlim) ? 1 : 0;   // wave-uniform block: scalar loads
    a.votes[i] = votes;
#endif
}

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

int64_t dd_compact_workspace_bytes(int64_t n) {
    if (n < 0) return DD_ERR_INVALID_ARG;
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
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 16)) {
        snprintf(g_ferr, sizeof(g_ferr), "workspace is NULL, mis-aligned or smaller than dd_compact_workspace_bytes()");
        return DD_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (n + C_TILE - 1) / C_TILE, groups = (tiles + C_GROUP - 1) / C_GROUP;
    if (tiles > 0x7fffffffll) return fail("too many rows for one call");
    CArgs a;
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
