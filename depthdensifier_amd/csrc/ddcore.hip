// ddcore.hip -- MI355X (gfx950 / CDNA4) kernels and C ABI of the densification core.
//
// Hot path replaced (reference citations relative to OpsiClear/DepthDensifier):
//   scripts/test.py:194        mask fold-in                -> validity predicate
//   scripts/test.py:205-212    strided grid + depth > 0    -> validity predicate + stable compaction
//   scripts/test.py:215-220    colour / normal gather      -> attribute gather of surviving pixels
//   scripts/test.py:79-90      pinhole unprojection        -> fused ray_to_world (R^T K^-1)
//   scripts/test.py:233        cam_from_world.inverse()*p  -> fused centre (-R^T t)
//   scripts/test.py:238-240, 264-266  list append + concatenate -> one global scan, each tile
//                              writes at its final slot (no concatenate copy)
//   src/depthdensifier/visualizer.py:291-376  package formulation (mask-only validity, general K,
//                              rotated + renormalised normals) -> flags of the same kernel.
//
// Design: a memory-bound map + stable compaction, no MFMA.  One 256-thread
// workgroup per tile of visited pixels; tiles take a ticket (so a tile only ever waits on tiles
// that already run), count their valid pixels with wave ballots, publish the count in an 8-byte
// {status,value} granule and obtain their global slot by decoupled look-back over predecessor
// granules (relaxed agent-scope atomics: the granule is the whole payload, so no fence is needed).
// Surviving pixels are listed in LDS in output order, then one lane per output point computes
// xyz and copies the attributes, so stores of consecutive lanes hit consecutive output rows.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ddcore.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

constexpr unsigned long long ST_AGG = 1ull << 62;   // value = number of valid pixels of the tile
constexpr unsigned long long ST_INCL = 2ull << 62;  // value = slot after the tile's last point
constexpr unsigned long long VAL_MASK = (1ull << 62) - 1;
constexpr unsigned SPIN_LIMIT = 1u << 21;

struct WsHeader {          // 16 bytes at the start of the workspace, zeroed every call
    unsigned int ticket;
    int error;
    unsigned int pad[2];
};

struct KArgs {
    const void *depth;
    const uint8_t *mask;
    const void *conf;
    const float *normal;
    const uint8_t *rgb;
    const DDViewParams *params;
    float *out_xyz;
    float *out_normal;
    uint8_t *out_rgb;
    int32_t *out_pix;
    int32_t *out_view;
    long long capacity;
    long long *view_offsets;
    const long long *cursor;
    WsHeader *hdr;
    unsigned long long *tile_state;
    unsigned long long *counts;   // count kernel only
    long long hw;                 // H*W
    int V, H, W, stride;
    int Hs, Ws;                   // visited grid
    unsigned P;                   // Hs*Ws visited pixels per view
    unsigned tiles_per_view;
    unsigned num_tiles;
    float conf_thr;
    unsigned flags;
    int conf_f16;
    int view_base;
};

using gu64 = __attribute__((address_space(1))) unsigned long long;

__device__ __forceinline__ unsigned long long ld_state(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- element loads ---------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float cvt(float x) { return x; }
};
template <> struct Elem<_Float16> {
    static __device__ __forceinline__ float cvt(_Float16 x) { return (float)x; }
};

// VEC contiguous elements, address aligned to min(16, VEC*sizeof(T)).
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, float (&out)[VEC]) {
    constexpr int BYTES = VEC * (int)sizeof(T);
    if constexpr (BYTES >= 16) {
        constexpr int PER = 16 / (int)sizeof(T);
        typedef T vt __attribute__((ext_vector_type(PER)));
#pragma unroll
        for (int i = 0; i < VEC / PER; ++i) {
            vt v = *reinterpret_cast<const vt *>(p + i * PER);
#pragma unroll
            for (int k = 0; k < PER; ++k) out[i * PER + k] = Elem<T>::cvt(v[k]);
        }
    } else if constexpr (VEC == 1) {
        out[0] = Elem<T>::cvt(*p);
    } else {
        typedef T vt __attribute__((ext_vector_type(VEC)));
        vt v = *reinterpret_cast<const vt *>(p);
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = Elem<T>::cvt(v[k]);
    }
}

template <int VEC>
__device__ __forceinline__ void load_bytes(const uint8_t *p, unsigned (&out)[VEC]) {
    if constexpr (VEC == 1) {
        out[0] = *p;
    } else if constexpr (VEC == 4) {
        unsigned w = *reinterpret_cast<const unsigned *>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (w >> (8 * k)) & 0xffu;
    } else {
        static_assert(VEC == 8, "VEC");
        uint2 w = *reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (w.x >> (8 * k)) & 0xffu;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[4 + k] = (w.y >> (8 * k)) & 0xffu;
    }
}

// ---- validity of the chunks a thread owns ------------------------------------------------------
// Visited pixel q of a view (row-major over the Hs x Ws strided grid) lives at source pixel
// (qy*stride, qx*stride).  A thread owns CHUNKS groups of VEC consecutive q; group c starts at
// q0 + (c*BLOCK + tid)*VEC, so every load instruction of a wave is one contiguous run.
template <typename DepthT, int VEC, int CHUNKS, bool CONTIG>
__device__ __forceinline__ void load_and_test(const KArgs &a, int v, unsigned q0, int tid,
                                              float (&dval)[CHUNKS][VEC], unsigned (&vbits)[CHUNKS]) {
    const long long vbase = (long long)v * a.hw;
    const DepthT *depth = reinterpret_cast<const DepthT *>(a.depth) + vbase;
    const bool use_depth = a.flags & DD_VALID_DEPTH_POSITIVE;
    const bool use_mask = a.flags & DD_VALID_MASK;
    const bool use_conf = a.flags & DD_VALID_CONF;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const unsigned qb = q0 + (unsigned)(c * BLOCK + tid) * VEC;
        unsigned bits = 0;
        if constexpr (CONTIG) {
            if (qb < a.P) {   // P % VEC == 0 in this mode: a group is entirely inside or outside
                load_vec<DepthT, VEC>(depth + qb, dval[c]);
                bits = (1u << VEC) - 1;
                if (use_mask) {
                    unsigned m[VEC];
                    load_bytes<VEC>(a.mask + vbase + qb, m);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) if (m[k] == 0) bits &= ~(1u << k);
                }
                if (use_conf) {
                    float cf[VEC];
                    if (a.conf_f16) load_vec<_Float16, VEC>(reinterpret_cast<const _Float16 *>(a.conf) + vbase + qb, cf);
                    else load_vec<float, VEC>(reinterpret_cast<const float *>(a.conf) + vbase + qb, cf);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) if (!(cf[k] > a.conf_thr)) bits &= ~(1u << k);
                }
                if (use_depth) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) if (!(dval[c][k] > 0.0f)) bits &= ~(1u << k);
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) dval[c][k] = 0.0f;
            }
        } else {
            static_assert(CONTIG || VEC == 1, "strided mode is scalar");
            dval[c][0] = 0.0f;
            if (qb < a.P) {
                const unsigned qy = qb / (unsigned)a.Ws;
                const unsigned qx = qb - qy * (unsigned)a.Ws;
                const long long off = vbase + (long long)(qy * (unsigned)a.stride) * a.W + qx * (unsigned)a.stride;
                const float d = Elem<DepthT>::cvt(reinterpret_cast<const DepthT *>(a.depth)[off]);
                dval[c][0] = d;
                bool ok = true;
                if (use_mask) ok = ok && (a.mask[off] != 0);
                if (use_conf) {
                    const float cf = a.conf_f16 ? (float)reinterpret_cast<const _Float16 *>(a.conf)[off]
                                                : reinterpret_cast<const float *>(a.conf)[off];
                    ok = ok && (cf > a.conf_thr);
                }
                if (use_depth) ok = ok && (d > 0.0f);
                bits = ok ? 1u : 0u;
            }
        }
        vbits[c] = bits;
    }
}

// Rank of this lane's first valid pixel of a chunk among the wave's pixels of that chunk
// (lane-major, then k) and the wave's total, from one ballot per k.
template <int VEC>
__device__ __forceinline__ void wave_rank(unsigned bits, int lane, int &lane_prefix, int &wave_total) {
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const unsigned long long b = __ballot((bits >> k) & 1u);
        pre += __popcll(b & lt);
        tot += __popcll(b);
    }
    lane_prefix = pre;
    wave_total = tot;
}

// ---- decoupled look-back (wave 0 of the workgroup, all 64 lanes) ---------------------------------
__device__ __forceinline__ long long lookback(unsigned long long *state, unsigned t, long long agg,
                                              long long base, int lane, int *err) {
    if (lane == 0) st_state(&state[t], ST_AGG | (unsigned long long)agg);
    long long excl = 0;
    long long look = (long long)t - 1;   // lane 0 inspects the nearest predecessor
    unsigned spins = 0;
    for (;;) {
        const long long idx = look - lane;
        // tiles "before the first" hold the cursor the batch starts from
        const unsigned long long s = (idx >= 0) ? ld_state(&state[idx]) : (ST_INCL | (unsigned long long)base);
        const unsigned st = (unsigned)(s >> 62);
        const unsigned long long incl_b = __ballot(st == 2u);
        const unsigned long long empty_b = __ballot(st == 0u);
        const int first_incl = incl_b ? __builtin_ctzll(incl_b) : 64;
        const unsigned long long need = (first_incl >= 63) ? ~0ull : ((2ull << first_incl) - 1ull);
        if (empty_b & need) {            // a predecessor we depend on has not published yet
            if (++spins > SPIN_LIMIT) {  // bounded: report and let successors proceed
                if (lane == 0) atomicExch(err, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        long long val = (lane <= first_incl) ? (long long)(s & VAL_MASK) : 0ll;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off);
        excl += val;
        if (incl_b) break;
        look -= 64;
    }
    if (lane == 0) st_state(&state[t], ST_INCL | (unsigned long long)(excl + agg));
    return excl;
}

// ---- the hot kernel ------------------------------------------------------------------------------
template <typename DepthT, int VEC, int CHUNKS, bool CONTIG>
__global__ __launch_bounds__(BLOCK) void unproject_compact_kernel(const KArgs a) {
    constexpr int TILE = BLOCK * VEC * CHUNKS;
    static_assert(TILE <= 65536, "tile-local index is 16 bit");
    __shared__ float s_d[TILE];
    __shared__ unsigned short s_q[TILE];
    __shared__ int s_tot[CHUNKS][WAVES];
    __shared__ long long s_excl;
    __shared__ unsigned s_ticket;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (tid == 0) s_ticket = atomicAdd(&a.hdr->ticket, 1u);
    __syncthreads();
    const unsigned t = __builtin_amdgcn_readfirstlane(s_ticket);
    if (t >= a.num_tiles) return;
    const int v = (int)(t / a.tiles_per_view);
    const unsigned tv = t - (unsigned)v * a.tiles_per_view;
    const unsigned q0 = tv * (unsigned)TILE;

    float dval[CHUNKS][VEC];
    unsigned vbits[CHUNKS];
    load_and_test<DepthT, VEC, CHUNKS, CONTIG>(a, v, q0, tid, dval, vbits);

    int lane_pre[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        int tot;
        wave_rank<VEC>(vbits[c], lane, lane_pre[c], tot);
        if (lane == 0) s_tot[c][wave] = tot;
    }
    __syncthreads();

    // tile-local rank of each owned pixel: chunks in order, waves in order inside a chunk
    int n = 0;
    int base_c[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            if (w == wave) base_c[c] = n;
            n += s_tot[c][w];
        }
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        int r = base_c[c] + lane_pre[c];
        const unsigned ql = (unsigned)(c * BLOCK + tid) * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if ((vbits[c] >> k) & 1u) {
                s_q[r] = (unsigned short)(ql + k);
                s_d[r] = dval[c][k];
                ++r;
            }
        }
    }

    if (wave == 0) {
        const long long base = *a.cursor;
        const long long excl = lookback(a.tile_state, t, n, base, lane, &a.hdr->error);
        if (lane == 0) {
            s_excl = excl;
            if (tv == 0) a.view_offsets[v] = excl;
            if (t == a.num_tiles - 1) a.view_offsets[a.V] = excl + n;
        }
    }
    __syncthreads();
    const long long excl = s_excl;

    // ---- one lane per output point ----
    const DDViewParams *vp = a.params + v;
    const float m00 = vp->ray_to_world[0], m01 = vp->ray_to_world[1], m02 = vp->ray_to_world[2];
    const float m10 = vp->ray_to_world[3], m11 = vp->ray_to_world[4], m12 = vp->ray_to_world[5];
    const float m20 = vp->ray_to_world[6], m21 = vp->ray_to_world[7], m22 = vp->ray_to_world[8];
    const float c0 = vp->centre[0], c1 = vp->centre[1], c2 = vp->centre[2];
    const bool rotate = a.flags & DD_ROTATE_NORMALS;
    const long long vbase = (long long)v * a.hw;

    for (int j = tid; j < n; j += BLOCK) {
        const long long slot = excl + j;
        if (slot >= a.capacity) break;
        const unsigned q = q0 + s_q[j];
        const float d = s_d[j];
        const unsigned qy = q / (unsigned)a.Ws;
        const unsigned qx = q - qy * (unsigned)a.Ws;
        const unsigned y = qy * (unsigned)a.stride, x = qx * (unsigned)a.stride;
        const int pix = (int)(y * (unsigned)a.W + x);
        const float fx = (float)x, fy = (float)y;
        const float rx = fmaf(m00, fx, fmaf(m01, fy, m02));
        const float ry = fmaf(m10, fx, fmaf(m11, fy, m12));
        const float rz = fmaf(m20, fx, fmaf(m21, fy, m22));
        float *o = a.out_xyz + slot * 3;
        o[0] = fmaf(d, rx, c0);
        o[1] = fmaf(d, ry, c1);
        o[2] = fmaf(d, rz, c2);
        if (a.out_normal) {
            const float *np = a.normal + (vbase + pix) * 3;
            float n0 = np[0], n1 = np[1], n2 = np[2];
            if (rotate) {
                const float w0 = vp->rot[0] * n0 + vp->rot[1] * n1 + vp->rot[2] * n2;
                const float w1 = vp->rot[3] * n0 + vp->rot[4] * n1 + vp->rot[5] * n2;
                const float w2 = vp->rot[6] * n0 + vp->rot[7] * n1 + vp->rot[8] * n2;
                const float inv = 1.0f / (sqrtf(w0 * w0 + w1 * w1 + w2 * w2) + 1e-8f);
                n0 = w0 * inv; n1 = w1 * inv; n2 = w2 * inv;
            }
            float *on = a.out_normal + slot * 3;
            on[0] = n0; on[1] = n1; on[2] = n2;
        }
        if (a.out_rgb) {
            const uint8_t *cp = a.rgb + (vbase + pix) * 3;
            uint8_t *oc = a.out_rgb + slot * 3;
            oc[0] = cp[0]; oc[1] = cp[1]; oc[2] = cp[2];
        }
        if (a.out_pix) a.out_pix[slot] = pix;
        if (a.out_view) a.out_view[slot] = a.view_base + v;
    }
}

// ---- per-view counts (scripts/test.py:210-212 "valid_pixels", without producing points) ----------
template <typename DepthT, int VEC, int CHUNKS, bool CONTIG>
__global__ __launch_bounds__(BLOCK) void count_valid_kernel(const KArgs a) {
    constexpr int TILE = BLOCK * VEC * CHUNKS;
    __shared__ int s_part[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned t = blockIdx.x;
    const int v = (int)(t / a.tiles_per_view);
    const unsigned tv = t - (unsigned)v * a.tiles_per_view;
    float dval[CHUNKS][VEC];
    unsigned vbits[CHUNKS];
    load_and_test<DepthT, VEC, CHUNKS, CONTIG>(a, v, tv * (unsigned)TILE, tid, dval, vbits);
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) cnt += __popc(vbits[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) s_part[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < WAVES; ++w) tot += s_part[w];
        if (tot) atomicAdd(&a.counts[v], (unsigned long long)tot);
    }
}

// ---- host side -----------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

struct Plan {
    bool f16;
    bool contig;
    int vec, chunks, tile;
};

int make_plan(const DDViewBatch *b, KArgs &a, Plan &p) {
    if (!b) return fail(DD_ERR_INVALID_ARG, "batch is NULL");
    if (b->num_views <= 0 || b->height <= 0 || b->width <= 0) return fail(DD_ERR_INVALID_ARG, "num_views/height/width must be positive");
    if (b->stride < 1) return fail(DD_ERR_INVALID_ARG, "stride must be >= 1");
    if (!b->depth) return fail(DD_ERR_INVALID_ARG, "depth is NULL");
    if (!b->params) return fail(DD_ERR_INVALID_ARG, "params is NULL");
    if (b->depth_dtype != DD_F32 && b->depth_dtype != DD_F16) return fail(DD_ERR_INVALID_ARG, "depth_dtype must be DD_F32 or DD_F16");
    if ((b->flags & DD_VALID_MASK) && !b->mask) return fail(DD_ERR_INVALID_ARG, "DD_VALID_MASK set but mask is NULL");
    if (b->flags & DD_VALID_CONF) {
        if (!b->conf) return fail(DD_ERR_INVALID_ARG, "DD_VALID_CONF set but conf is NULL");
        if (b->conf_dtype != DD_F32 && b->conf_dtype != DD_F16) return fail(DD_ERR_INVALID_ARG, "conf_dtype must be DD_F32 or DD_F16");
    }
    if (!(b->flags & (DD_VALID_MASK | DD_VALID_CONF | DD_VALID_DEPTH_POSITIVE)))
        return fail(DD_ERR_INVALID_ARG, "flags select no validity rule");
    const long long hw = (long long)b->height * b->width;
    if (hw >= (1ll << 31)) return fail(DD_ERR_UNSUPPORTED, "views of 2^31 pixels or more are not supported");

    memset(&a, 0, sizeof(a));
    a.depth = b->depth; a.mask = b->mask; a.conf = b->conf; a.normal = b->normal; a.rgb = b->rgb;
    a.params = b->params;
    a.hw = hw; a.V = b->num_views; a.H = b->height; a.W = b->width; a.stride = b->stride;
    a.Hs = (b->height + b->stride - 1) / b->stride;
    a.Ws = (b->width + b->stride - 1) / b->stride;
    a.P = (unsigned)((long long)a.Hs * a.Ws);
    a.conf_thr = b->conf_threshold; a.flags = b->flags; a.conf_f16 = (b->conf_dtype == DD_F16);
    a.view_base = b->view_index_base;

    p.f16 = (b->depth_dtype == DD_F16);
    const int vec = p.f16 ? 8 : 4;
    const uintptr_t dp = (uintptr_t)b->depth, mp = (uintptr_t)b->mask, cp = (uintptr_t)b->conf;
    bool ok = (b->stride == 1) && (hw % vec == 0) && (dp % 16 == 0);
    if (b->flags & DD_VALID_MASK) ok = ok && (mp % vec == 0);
    if (b->flags & DD_VALID_CONF) ok = ok && (cp % 16 == 0);
    if (b->tuning & 1u) ok = false;   // tuning bit 0: force the scalar path (testing)
    p.contig = ok;
    p.vec = ok ? vec : 1;
    p.chunks = ok ? (p.f16 ? 1 : 2) : 8;
    p.tile = BLOCK * p.vec * p.chunks;
    a.tiles_per_view = (a.P + p.tile - 1) / p.tile;
    const unsigned long long nt = (unsigned long long)a.tiles_per_view * (unsigned)a.V;
    if (nt >= (1ull << 31)) return fail(DD_ERR_UNSUPPORTED, "too many tiles in one batch; split the batch");
    a.num_tiles = (unsigned)nt;
    return DD_OK;
}

int launch_compact(const Plan &p, const KArgs &a, hipStream_t s) {
    const dim3 grid(a.num_tiles), block(BLOCK);
    if (p.contig) {
        if (p.f16) hipLaunchKernelGGL((unproject_compact_kernel<_Float16, 8, 1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((unproject_compact_kernel<float, 4, 2, true>), grid, block, 0, s, a);
    } else {
        if (p.f16) hipLaunchKernelGGL((unproject_compact_kernel<_Float16, 1, 8, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((unproject_compact_kernel<float, 1, 8, false>), grid, block, 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? DD_OK : DD_ERR_LAUNCH;
}

int launch_count(const Plan &p, const KArgs &a, hipStream_t s) {
    const dim3 grid(a.num_tiles), block(BLOCK);
    if (p.contig) {
        if (p.f16) hipLaunchKernelGGL((count_valid_kernel<_Float16, 8, 1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((count_valid_kernel<float, 4, 2, true>), grid, block, 0, s, a);
    } else {
        if (p.f16) hipLaunchKernelGGL((count_valid_kernel<_Float16, 1, 8, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((count_valid_kernel<float, 1, 8, false>), grid, block, 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? DD_OK : DD_ERR_LAUNCH;
}

int64_t ws_bytes(const KArgs &a) {
    return (int64_t)sizeof(WsHeader) + (int64_t)a.num_tiles * 8;
}

}  // namespace

extern "C" {

int dd_abi_version(void) { return DD_ABI_VERSION; }

const char *dd_last_error(void) { return g_err; }

int dd_count_valid(const DDViewBatch *batch, int64_t *counts_dev, void *stream) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if (!counts_dev) return fail(DD_ERR_INVALID_ARG, "counts_dev is NULL");
    hipStream_t s = (hipStream_t)stream;
    a.counts = reinterpret_cast<unsigned long long *>(counts_dev);
    if (hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * (size_t)a.V, s) != hipSuccess)
        return fail(DD_ERR_LAUNCH, "hipMemsetAsync(counts) failed");
    rc = launch_count(p, a, s);
    if (rc != DD_OK) return fail(rc, "count_valid kernel launch failed");
    return DD_OK;
}

int64_t dd_workspace_bytes(const DDViewBatch *batch) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    return ws_bytes(a);
}

int dd_unproject_compact(const DDViewBatch *batch, const DDCloudOut *out, int64_t *view_offsets_dev,
                         int64_t *cursor_dev, void *workspace, int64_t workspace_bytes, void *stream) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if (!out || !out->xyz) return fail(DD_ERR_INVALID_ARG, "out / out->xyz is NULL");
    if (out->capacity < 0) return fail(DD_ERR_INVALID_ARG, "capacity is negative");
    if (out->normal && !batch->normal) return fail(DD_ERR_INVALID_ARG, "out->normal requested but batch->normal is NULL");
    if (out->rgb && !batch->rgb) return fail(DD_ERR_INVALID_ARG, "out->rgb requested but batch->rgb is NULL");
    if (!view_offsets_dev || !cursor_dev) return fail(DD_ERR_INVALID_ARG, "view_offsets_dev / cursor_dev is NULL");
    if (!workspace || ((uintptr_t)workspace % 16) != 0) return fail(DD_ERR_WORKSPACE, "workspace is NULL or not 16-byte aligned");
    const int64_t need = ws_bytes(a);
    if (workspace_bytes < need) return fail(DD_ERR_WORKSPACE, "workspace too small (see dd_workspace_bytes)");

    hipStream_t s = (hipStream_t)stream;
    a.out_xyz = out->xyz; a.out_normal = out->normal; a.out_rgb = out->rgb;
    a.out_pix = out->pixel_index; a.out_view = out->view_index; a.capacity = out->capacity;
    a.view_offsets = reinterpret_cast<long long *>(view_offsets_dev);
    a.cursor = reinterpret_cast<const long long *>(cursor_dev);
    a.hdr = reinterpret_cast<WsHeader *>(workspace);
    a.tile_state = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(workspace) + sizeof(WsHeader));

    if (hipMemsetAsync(workspace, 0, (size_t)need, s) != hipSuccess)
        return fail(DD_ERR_LAUNCH, "hipMemsetAsync(workspace) failed");
    rc = launch_compact(p, a, s);
    if (rc != DD_OK) return fail(rc, "unproject_compact kernel launch failed");
    // cursor <- slot after the batch (kept out of the kernel: tiles read the old cursor)
    if (hipMemcpyAsync(cursor_dev, view_offsets_dev + a.V, sizeof(int64_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(DD_ERR_LAUNCH, "hipMemcpyAsync(cursor) failed");
    return DD_OK;
}

}  // extern "C"
