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
// Design: a memory-bound map + stable compaction, no MFMA.  Every tile re-derives
// its validity bits, ranks its survivors with wave ballots, lists them in LDS in output order and one
// lane per output point computes xyz and copies the attributes, so that consecutive lanes write
// consecutive rows of the (N,3) outputs at their final position.  The tile's first row comes either
//   * from a ticket + decoupled look-back over 8-byte {status,value} granules (relaxed agent-scope
//     atomics; the granule is the whole payload, so no fence) -- dd_unproject_compact on aligned
//     stride-1 maps: 12288-pixel tiles (12 waves), a 16-granule window (polling loads cross XCDs, so few of them),
//     and the other waves issue their first gathers while wave 0 looks back; or
//   * from a counting pass + two small scans (dd_plan), after which the scatter pass (dd_scatter) has
//     no inter-workgroup dependency at all -- the exact-allocation API and the generic path.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "ddcore.h"
#include "ddcore_lab.h"
#include "ddrefine_math.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

// A look-back granule: [63:62] status | [61:44] epoch of the call that wrote it | [43:0] value.  The epoch makes every granule
// of an earlier call read as "not published" without anybody zeroing it: a single-pass call is ONE stream operation (round 5;
// rounds 1-4 zeroed the granules with a memset in front of every kernel and copied the cursor behind it).
constexpr int EPOCH_SHIFT = 44, EPOCH_BITS = 18;
constexpr unsigned EPOCH_MAX = (1u << EPOCH_BITS) - 1u;
constexpr unsigned long long ST_AGG = 1ull << 62;   // value = number of valid pixels of the tile
constexpr unsigned long long ST_INCL = 2ull << 62;  // value = slot after the tile's last point
constexpr unsigned long long POISON = 1ull << (EPOCH_SHIFT - 1);      // inclusive granule of a tile that does not know its first row (a look-back
                                                                      // gave up somewhere before it): whoever reads it does not know its own either
constexpr unsigned long long VAL_MASK = POISON - 1;                   // rows below 2^43
constexpr int CHAIN_SHIFT = 44;                   // chain word: [63:44] sequence (mod 2^20), [43:0] the row the next call starts from
constexpr unsigned CHAIN_SEQ_MASK = (1u << 20) - 1u;
constexpr long long POISON_ROW = 1ll << 42;   // the cursor a call leaves behind when its last tile does not know the row after the batch: beyond any
                                              // capacity, so that calls chained behind it count their points and write nothing (the error word is set)
constexpr unsigned TAG_AGG = 1u << EPOCH_BITS, TAG_INCL = 2u << EPOCH_BITS;   // granule >> EPOCH_SHIFT == TAG_x | epoch
constexpr unsigned SPIN_LIMIT = 1u << 21;

struct WsHeader {          // 64 bytes at the start of the workspace
    // first 16 bytes: STICKY -- never zeroed by the library.  `error` (int32 word 1 of the workspace) is set when an
    // in-kernel look-back gives up and stays set over later calls on the same workspace until the caller clears
    // it, so a caller that chains many batches through one workspace checks it once at the end.
    unsigned int pad0;
    int error;
    unsigned int pad1[2];
    // state of the single-pass kernel, all zero between calls except `epoch`: the workgroup that is the LAST of a call to
    // finish its look-back writes the cursor, puts ticket and done back to zero and advances the epoch (when the epoch
    // wraps -- every 2^18 calls -- it also zeroes every record of the workspace, so a stale tag can never match)
    unsigned int ticket;
    unsigned int done;
    unsigned int epoch;
    unsigned int pad2;
    long long pad3[4];
};
static_assert(sizeof(WsHeader) == 64, "workspace header");

// behind the header: the look-back granules of the single-pass kernels, 8 bytes per tile, PACKED (a polling lane fetches two
// of them with one 16-byte load: a poll costs per request, not per byte -- profiles/r05_streaming_poll_width.txt), then the
// count / first row of every tile for the two-pass kernels, then 8 bytes per view.  Calls with different tilings lay these
// regions out differently on the same workspace; whatever a two-pass call leaves where a later single-pass call has its
// granules reads as "not published": a {count, row} pair has its top two bits clear (rows < 2^30), and so has a view total.
struct TileCO {
    unsigned cnt;             // two-pass: valid pixels of the tile
    unsigned off;             // two-pass: first row of the tile inside its view (< 2^30: see above)
};
static_assert(sizeof(TileCO) == 8, "tile record");

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
    float *refined_out;           // DD_REFINE: (V,H,W) refined map or NULL
    int raw_f16;                  // DD_REFINE: the raw depth is float16 (the kernel's own depth type is float32 then)
    uint32_t *out_packed;         // (capacity,4) dwords: x, y, z (float bits), r | g<<8 | b<<16 | 255<<24 -- the 16-byte gather record
    long long capacity;
    long long *view_offsets;
    const long long *cursor;
    long long *cursor_out;        // two-pass: scan_views stores the row after the batch here (may be NULL)
    WsHeader *hdr;
    unsigned long long *gran;     // look-back granules, one per tile of the single-pass tiling
    unsigned long long *pref;     // scan service: the first row of every tile, written by the service workgroup (same tags as the granules)
    int scan_service;             // single-pass lean kernel: one workgroup of the launch scans the tiles' counts, the tiles poll their own row
    unsigned long long *chain;    // ABI 12: the word that chains this call behind the previous one of the same cloud on another stream (or NULL)
    unsigned chain_seq;           // ... and the sequence number the word must show before this call's scan may start
    int chain_gated;              // ... the call is preceded by chain_gate (too large to wait inside its own workgroups)
    TileCO *tiles;                // two-pass: count and first row of every tile (NULL in dd_count_valid)
    unsigned long long ws_words;  // 8-byte words of the caller's workspace behind the header (what a wrap of the epoch zeroes)
    unsigned long long *counts;   // per-view counts (dd_count_valid)
    long long *view_tot;          // two-pass: valid pixels per view
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
    int align_runs;               // lean kernels: shift the sweeps so that wave runs start on 128-byte lines (default on)
    unsigned spin_limit;          // look-back polls before a tile gives up (SPIN_LIMIT; 0 with the fault-injection tuning bit 64)
    int dense_ok;                 // lean kernels: tiles whose pixels all survive take the list-free path (dense_wave)
    unsigned order_regions;       // two-pass scatter: 1 = tiles in order, K > 1 = K stretches of the batch interleaved (tuning bits 8-13)
    int static_tiles;             // single-pass: tile = workgroup index instead of a ticket (tuning bit 22)
    unsigned lb_lanes;            // single-pass: lanes that poll in a look-back round (window of 16 / 32 / 64 tiles)
    int assume_dense;             // scatter pass: the plan was written WITHOUT counting, as if every visited pixel were valid (tuning bit 17);
                                  // a tile that finds otherwise sets the workspace's error word to 2 (the rows are then invalid: redo)
};


typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// two neighbouring granules (16-byte aligned pair) with ONE agent-scope load
__device__ __forceinline__ u32x4_t ld_state_pair(const unsigned long long *p) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long ld_state(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_state(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- element conversion ------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float cvt(float x) { return x; }
};
template <> struct Elem<_Float16> {
    static __device__ __forceinline__ float cvt(_Float16 x) { return (float)x; }
};

// ==================================================================================================
// Generic path: any downsample_density, any alignment (scripts/test.py:205-206 default stride is 32,
// i.e. ~1/1000 of the pixels -- this path is about semantics, not bandwidth).  Visited pixel q of a
// view (row-major over the Hs x Ws strided grid) lives at source pixel (qy*stride, qx*stride); a
// thread owns G_CHUNKS visited pixels, chunk c at q0 + c*BLOCK + tid (scalar, lane-contiguous loads).
// ==================================================================================================
constexpr int G_CHUNKS = 8;
constexpr int G_TILE = BLOCK * G_CHUNKS;

template <typename DepthT>
__device__ __forceinline__ void generic_load_test(const KArgs &a, int v, unsigned q0, int tid,
                                                  float (&dval)[G_CHUNKS], unsigned (&vbits)[G_CHUNKS]) {
    const long long vbase = (long long)v * a.hw;
    const bool use_depth = a.flags & DD_VALID_DEPTH_POSITIVE;
    const bool use_mask = a.flags & DD_VALID_MASK;
    const bool use_conf = a.flags & DD_VALID_CONF;
#pragma unroll
    for (int c = 0; c < G_CHUNKS; ++c) {
        const unsigned q = q0 + (unsigned)(c * BLOCK + tid);
        dval[c] = 0.0f;
        vbits[c] = 0;
        if (q < a.P) {
            const unsigned qy = q / (unsigned)a.Ws;
            const unsigned qx = q - qy * (unsigned)a.Ws;
            const long long off = vbase + (long long)(qy * (unsigned)a.stride) * a.W + qx * (unsigned)a.stride;
            const float d = Elem<DepthT>::cvt(reinterpret_cast<const DepthT *>(a.depth)[off]);
            dval[c] = d;
            bool ok = true;
            if (use_mask) ok = ok && (a.mask[off] != 0);
            if (use_conf) {
                const float cf = a.conf_f16 ? (float)reinterpret_cast<const _Float16 *>(a.conf)[off]
                                            : reinterpret_cast<const float *>(a.conf)[off];
                ok = ok && (cf > a.conf_thr);
            }
            if (use_depth) ok = ok && (d > 0.0f);
            vbits[c] = ok ? 1u : 0u;
        }
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

// ---- the end of a single-pass call, inside the kernel ----------------------------------------------
// The batch's LAST tile (the one with the last ticket) closes the call as soon as its own look-back is over: it writes the
// cursor, puts the ticket back to zero and advances the epoch.  That is safe because by then every tile of the call
//   * has drawn its ticket (tickets are drawn in order),
//   * has read the epoch (its aggregate granule, which the last tile's look-back has seen or summed over, carries it), and
//   * has read the cursor (load_cursor: the load is complete before the tile publishes anything),
// and nothing else of what the close changes is read inside a call.  No atomic, no fence, nobody waits.
// Only when the epoch wraps (every 2^18 calls) the granules themselves must be zeroed, and that has to wait until the LAST
// look-back of the call is over: in that one call every tile counts itself done (tile_done) and whoever counts last zeroes
// every record the workspace holds and closes the call.
__device__ __forceinline__ long long load_cursor(const long long *cursor) {
    const long long base = *cursor;
    // the value in scalar registers HERE, and no later memory operation moved in front of this point
    asm volatile("" ::"s"((unsigned)(unsigned long long)base), "s"((unsigned)((unsigned long long)base >> 32)) : "memory");
    return base;
}
__device__ __forceinline__ void close_call(WsHeader *h, long long *cursor_out, long long row_after_batch, unsigned epoch) {
    *cursor_out = row_after_batch;          // (POISON_ROW if a look-back gave up: the error word says so)
    h->ticket = 0u;
    h->epoch = epoch == EPOCH_MAX ? 0u : epoch + 1u;
}
// wave 0 of every tile of the call in which the epoch wraps, all 64 lanes, after the tile's look-back
__device__ __forceinline__ void wrap_call(WsHeader *h, unsigned long long *gran, unsigned long long ws_words, long long *cursor_out,
                                          unsigned T, int lane) {
    unsigned before = 0u;
    if (lane == 0) before = __hip_atomic_fetch_add(&h->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)before) != T - 1u) return;
    unsigned long long last;                // the row after the batch: the inclusive granule of the batch's last tile (stored before
    do {                                    // that tile counted itself done: on its way if it is not there yet)
        last = ld_state(&gran[T - 1u]);
    } while ((unsigned)(last >> EPOCH_SHIFT) != (TAG_INCL | EPOCH_MAX));
    for (unsigned long long i = (unsigned long long)lane; i < ws_words; i += 64ull) gran[i] = 0ull;
    if (lane == 0) {
        h->done = 0u;
        close_call(h, cursor_out, (last & POISON) ? POISON_ROW : (long long)(last & VAL_MASK), EPOCH_MAX);
    }
}

// ---- decoupled look-back (wave 0 of the workgroup, all 64 lanes) ---------------------------------
__device__ __forceinline__ long long lookback(unsigned long long *gran, unsigned t, long long agg,
                                              long long base, int lane, int *err, const unsigned epoch) {
    const unsigned long long tag_agg = (unsigned long long)(TAG_AGG | epoch) << EPOCH_SHIFT;
    const unsigned long long tag_incl = (unsigned long long)(TAG_INCL | epoch) << EPOCH_SHIFT;
    if (lane == 0) st_state(&gran[t], tag_agg | (unsigned long long)agg);
    long long excl = 0;
    long long look = (long long)t - 1;   // lane 0 inspects the nearest predecessor
    unsigned spins = 0;
    bool poisoned = false;               // -> returns -1: the tile's first row is unknown, it must not write anything
    for (;;) {
        const long long idx = look - lane;
        // tiles "before the first" hold the cursor the batch starts from
        const unsigned long long s = (idx >= 0) ? ld_state(&gran[idx]) : (tag_incl | (unsigned long long)base);
        const unsigned tag = (unsigned)(s >> EPOCH_SHIFT);
        const unsigned st = tag == (TAG_INCL | epoch) ? 2u : tag == (TAG_AGG | epoch) ? 1u : 0u;   // another call's granule: not published
        const unsigned long long incl_b = __ballot(st == 2u);
        const unsigned long long empty_b = __ballot(st == 0u);
        const int first_incl = incl_b ? __builtin_ctzll(incl_b) : 64;
        const unsigned long long need = (first_incl >= 63) ? ~0ull : ((2ull << first_incl) - 1ull);
        if (empty_b & need) {            // a predecessor we depend on has not published yet
            if (++spins > SPIN_LIMIT) {  // bounded: report and let successors proceed
                if (lane == 0) atomicExch(err, 1);
                poisoned = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        long long val = (lane <= first_incl) ? (long long)(s & VAL_MASK) : 0ll;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) val += __shfl_xor(val, off);
        excl += val;
        if (incl_b) {
            poisoned = (__ballot(lane == first_incl && (s & POISON) != 0ull) != 0ull);
            break;
        }
        look -= 64;
    }
    if (lane == 0) st_state(&gran[t], tag_incl | (poisoned ? POISON : 0ull) | ((unsigned long long)(excl + agg) & VAL_MASK));
    return poisoned ? -1ll : excl;
}

// ---- generic scatter kernel (two-pass by default; SINGLE_PASS = ticket + look-back variant) -------
template <typename DepthT, bool SINGLE_PASS>
__global__ __launch_bounds__(BLOCK) void compact_generic(const KArgs a) {
    __shared__ float s_d[G_TILE];
    __shared__ unsigned short s_q[G_TILE];
    __shared__ int s_tot[G_CHUNKS][WAVES];
    __shared__ long long s_excl;
    __shared__ unsigned s_ticket;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    unsigned t, epoch = 0u;
    long long base = 0;
    if constexpr (SINGLE_PASS) {
        epoch = a.hdr->epoch;
        base = load_cursor(a.cursor);
        if (tid == 0) s_ticket = atomicAdd(&a.hdr->ticket, 1u);
        __syncthreads();
        t = __builtin_amdgcn_readfirstlane(s_ticket);
        if (t >= a.num_tiles) return;          // (never: the grid is num_tiles workgroups)
    } else {
        t = blockIdx.x;
    }
    const int v = (int)(t / a.tiles_per_view);
    const unsigned tv = t - (unsigned)v * a.tiles_per_view;
    const unsigned q0 = tv * (unsigned)G_TILE;

    float dval[G_CHUNKS];
    unsigned vbits[G_CHUNKS];
    generic_load_test<DepthT>(a, v, q0, tid, dval, vbits);

    int lane_pre[G_CHUNKS];
#pragma unroll
    for (int c = 0; c < G_CHUNKS; ++c) {
        int tot;
        wave_rank<1>(vbits[c], lane, lane_pre[c], tot);
        if (lane == 0) s_tot[c][wave] = tot;
    }
    __syncthreads();

    // tile-local rank of each owned pixel: chunks in order, waves in order inside a chunk
    int n = 0;
    int base_c[G_CHUNKS];
#pragma unroll
    for (int c = 0; c < G_CHUNKS; ++c) {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            if (w == wave) base_c[c] = n;
            n += s_tot[c][w];
        }
    }
#pragma unroll
    for (int c = 0; c < G_CHUNKS; ++c) {
        if (vbits[c]) {
            const int r = base_c[c] + lane_pre[c];
            s_q[r] = (unsigned short)(c * BLOCK + tid);
            s_d[r] = dval[c];
        }
    }

    if constexpr (SINGLE_PASS) {
        if (wave == 0) {
            const long long e = lookback(a.gran, t, n, base, lane, &a.hdr->error, epoch);
            if (lane == 0) {
                s_excl = e;
                if (tv == 0) a.view_offsets[v] = e;
                if (t == a.num_tiles - 1) a.view_offsets[a.V] = e + n;
            }
            if (epoch == EPOCH_MAX) wrap_call(a.hdr, a.gran, a.ws_words, a.cursor_out, a.num_tiles, lane);
            else if (lane == 0 && t == a.num_tiles - 1) close_call(a.hdr, a.cursor_out, e < 0 ? POISON_ROW : e + n, epoch);
        }
    }
    __syncthreads();
    long long excl;
    if constexpr (SINGLE_PASS) excl = s_excl;
    else excl = a.view_offsets[v] + (long long)a.tiles[t].off;

    // ---- one lane per output point ----
    const DDViewParams *vp = a.params + v;
    const float m00 = vp->ray_to_world[0], m01 = vp->ray_to_world[1], m02 = vp->ray_to_world[2];
    const float m10 = vp->ray_to_world[3], m11 = vp->ray_to_world[4], m12 = vp->ray_to_world[5];
    const float m20 = vp->ray_to_world[6], m21 = vp->ray_to_world[7], m22 = vp->ray_to_world[8];
    const float c0 = vp->centre[0], c1 = vp->centre[1], c2 = vp->centre[2];
    const bool rotate = a.flags & DD_ROTATE_NORMALS;
    const long long vbase = (long long)v * a.hw;

    if (excl < 0) return;             // single-pass: the tile's first row is unknown (a look-back gave up): nothing is written
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
        const float ox = fmaf(d, rx, c0), oy = fmaf(d, ry, c1), oz = fmaf(d, rz, c2);
        if (a.out_xyz) {
            float *o = a.out_xyz + slot * 3;
            o[0] = ox; o[1] = oy; o[2] = oz;
        }
        if (a.out_packed) {
            unsigned c = 0xff000000u;
            if (a.rgb) {
                const uint8_t *cp = a.rgb + (vbase + pix) * 3;
                c |= (unsigned)cp[0] | ((unsigned)cp[1] << 8) | ((unsigned)cp[2] << 16);
            }
            uint4 rec; rec.x = __float_as_uint(ox); rec.y = __float_as_uint(oy); rec.z = __float_as_uint(oz); rec.w = c;
            *reinterpret_cast<uint4 *>(a.out_packed + slot * 4) = rec;
        }
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

// ==================================================================================================
// Lean path (stride 1, any view size; see lean_load_test for ragged / element-aligned views).  Sized for latency
// tolerance: <= 80 VGPRs and 24 KiB of LDS per 256-thread workgroup -> 6 workgroups (24 waves) per CU in the
// two-pass kernels (the single-pass variant runs 12-wave workgroups on 12288-pixel tiles, 2 per CU), every load of a tile issued
// back to back, gathers of the next group of points in flight while the current group is stored.
//
//  * a tile is 4096 (single-pass: 12288) consecutive pixels of one view; wave w owns the contiguous span
//    [q0 + 1024 w, +1024); lane l owns CH groups of VEC pixels, group ch at
//    q0 + 1024 w + (64 ch + l) VEC, so each load instruction of a wave is one contiguous run;
//  * validity -> wave ballots -> tile-local rank; surviving pixels are listed in LDS (16-bit pixel
//    index + depth) in output order; then ONE LANE PER OUTPUT POINT computes xyz and copies the
//    attributes, so consecutive lanes write consecutive rows of the (N,3) outputs;
//  * SINGLE_PASS: the tile's first output row comes from a ticket + decoupled look-back;
//    otherwise (two-pass) from the offsets produced by count_lean + the two scan kernels, and
//    the workgroup has no dependency on any other workgroup.
// ==================================================================================================
#ifndef DD_L_PXT
#define DD_L_PXT 16
#endif
#ifndef DD_NT_STORE
#define DD_NT_STORE 0
#endif
#ifndef DD_NT_LOAD
#define DD_NT_LOAD 1
#endif
#ifndef DD_XCD_SWIZZLE
#define DD_XCD_SWIZZLE 0
#endif
#ifndef DD_REVERSE_SCATTER
#define DD_REVERSE_SCATTER 1
#endif
#ifndef DD_LEAN_WGS
#define DD_LEAN_WGS 6
#endif
#ifndef DD_REFINE_EV
#define DD_REFINE_EV 4
#endif
constexpr int L_PXT = DD_L_PXT;              // pixels per lane per tile
constexpr int L_WSPAN = 64 * L_PXT;          // 1024 pixels per wave
constexpr int L_TILE = WAVES * L_WSPAN;      // 4096
#ifndef DD_SP_WAVES
#define DD_SP_WAVES 12
#endif
constexpr int SP_WAVES = DD_SP_WAVES;         // single-pass variant: 12 waves, 12288-pixel tiles (8: +3 %, 16: +3 %, 4: +7 % time)
#ifndef DD_SP_PXT_SMALL
#define DD_SP_PXT_SMALL 8
#endif
#ifndef DD_SP_SMALL_BATCH_TILES
#define DD_SP_SMALL_BATCH_TILES 3072
#endif
// A small batch (a streamed view or two: scripts/test.py:131 densifies one view per loop iteration)
// does not fill the chip even once, so a tile's lifetime IS the kernel's: 8 pixels per lane instead of 16 -- 8192-pixel tiles of 16 waves -- halve
// the chain of gather / store sweeps every lane walks through and double the workgroups (profiles/r05_stamps_small_batches.txt:
// of the 17 us a 12288-pixel tile of a one-view launch lives, 8.5 are those sweeps, 16 times one memory latency that nothing hides).
// In a CHAIN of calls (what a streaming caller makes) the small tile wins up to 16 views of 1080p per call and draws at 32
// (profiles/r05_streaming_sweep_9_service_ticket_vs_index.txt: 8 views 0.677 against 0.644, 16 views 0.717 against 0.704); a single
// launch between two synchronisations prefers the large tile from four views on (r05_ab_scan_service_2.txt) -- the chain decides.
constexpr int SP_PXT_SMALL = DD_SP_PXT_SMALL;
#ifndef DD_SP_WAVES_SMALL
#define DD_SP_WAVES_SMALL 16
#endif
// ... in workgroups of 16 waves (8192-pixel tiles): with 52 registers and 49 KiB of LDS two of them -- 32 waves -- fit a CU, where the
// 12-wave workgroup leaves it at 24; 8 waves (four per CU) do the same.  A chain of one-view calls 0.43 -> 0.51 of the roofline,
// two views 0.52 -> 0.60, four 0.62 -> 0.66, eight 0.68 -> 0.69 (profiles/r05_streaming_small_tile_waves.txt)
constexpr int SP_WAVES_SMALL = DD_SP_WAVES_SMALL;
constexpr unsigned long long SP_SMALL_BATCH_TILES = DD_SP_SMALL_BATCH_TILES;   // batches of up to this many 12288-pixel tiles take the small tile
static_assert(DD_SP_WAVES_SMALL * 64 * DD_SP_PXT_SMALL >= 4096 && DD_SP_WAVES * 64 * DD_SP_PXT_SMALL >= 4096, "the workspace holds one record per 4096 pixels: the small single-pass tile must not be finer");

typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_unaligned __attribute__((aligned(1)));
typedef unsigned u32_unaligned __attribute__((aligned(1)));

template <typename DepthT> __device__ __forceinline__ float raw_depth(const uint4 &d, int k);
template <> __device__ __forceinline__ float raw_depth<float>(const uint4 &d, int k) {
    return __uint_as_float(k == 0 ? d.x : k == 1 ? d.y : k == 2 ? d.z : d.w);
}
template <> __device__ __forceinline__ float raw_depth<_Float16>(const uint4 &d, int k) {
    const unsigned w = (k >> 1) == 0 ? d.x : (k >> 1) == 1 ? d.y : (k >> 1) == 2 ? d.z : d.w;
    const unsigned short h = (unsigned short)((k & 1) ? (w >> 16) : (w & 0xffffu));
    _Float16 f;
    __builtin_memcpy(&f, &h, 2);
    return (float)f;
}

// element-aligned (not vector-aligned) wide loads: a view of H*W pixels starts wherever the previous one ended, and
// gfx950 serves dword- / byte-aligned dwordx4 accesses in hardware (tools/experiments/ubench.hip: -25 % at worst on a pure
// store stream; nothing measurable on aligned addresses, where the instruction is the same)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(2)));
typedef u32x2 u32x2_unaligned __attribute__((aligned(2)));

// once-read streams (depth, mask, confidence): non-temporal unless built with -DDD_NT_LOAD=0.  A macro, not a
// function template: the pointee's reduced alignment (element-aligned wide loads) must reach the builtin.
#if DD_NT_LOAD
#define LD_STREAM(p) __builtin_nontemporal_load(p)
#else
#define LD_STREAM(p) (*(p))
#endif

template <typename DepthT> __device__ __forceinline__ void set_raw(uint4 &d, int k, unsigned bits);
template <> __device__ __forceinline__ void set_raw<float>(uint4 &d, int k, unsigned bits) {
    if (k == 0) d.x = bits; else if (k == 1) d.y = bits; else if (k == 2) d.z = bits; else d.w = bits;
}
template <> __device__ __forceinline__ void set_raw<_Float16>(uint4 &d, int k, unsigned bits) {
    unsigned &w = (k >> 1) == 0 ? d.x : (k >> 1) == 1 ? d.y : (k >> 1) == 2 ? d.z : d.w;
    w = (k & 1) ? ((w & 0x0000ffffu) | (bits << 16)) : ((w & 0xffff0000u) | (bits & 0xffffu));
}

// Loads + validity bits of the CH groups (vectors of VEC pixels) a lane owns.  Every load is unconditional so that
// the compiler issues them all before the first use: a vector that would cross the end of the view is read from
// the view's last VEC pixels instead (P >= VEC), and -- only in the wave that holds the end of a view whose pixel
// count is not a multiple of VEC, a wave-uniform branch -- the one partial vector is re-read element by element.
template <typename DepthT, bool HAS_MASK, bool NEED_DEPTH, int PXT = L_PXT>
__device__ __forceinline__ void lean_load_test(const KArgs &a, long long vbase, unsigned qw, int lane,
                                               uint4 (&d)[PXT / (16 / (int)sizeof(DepthT))],
                                               unsigned (&bits)[PXT / (16 / (int)sizeof(DepthT))]) {
    // NEED_DEPTH = false: pass 1 under a mask-only validity rule touches 1 B/px instead of 5
    constexpr int VEC = 16 / (int)sizeof(DepthT), CH = PXT / VEC;
    static_assert(PXT % VEC == 0 && CH >= 1, "a lane owns whole vectors");
    unsigned mk[CH][VEC / 4];
    unsigned nvalid[CH];                      // pixels of the vector that exist: VEC, 0, or 1..VEC-1 at a ragged view end
    long long eload[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const unsigned qb = qw + (unsigned)(ch * 64 + lane) * VEC;
        nvalid[ch] = qb < a.P ? (a.P - qb < (unsigned)VEC ? a.P - qb : (unsigned)VEC) : 0u;
        const long long e = vbase + (nvalid[ch] == (unsigned)VEC ? qb : a.P - VEC);
        eload[ch] = e;
        if constexpr (!NEED_DEPTH) { d[ch].x = d[ch].y = d[ch].z = d[ch].w = 0u; }
#if DD_NT_LOAD
        if constexpr (NEED_DEPTH) {
            const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(reinterpret_cast<const DepthT *>(a.depth) + e));
            d[ch].x = w.x; d[ch].y = w.y; d[ch].z = w.z; d[ch].w = w.w;
        }
        if constexpr (HAS_MASK) {
#pragma unroll
            for (int i = 0; i < VEC / 4; ++i) mk[ch][i] = __builtin_nontemporal_load(reinterpret_cast<const u32_unaligned *>(a.mask + e + 4 * i));
        }
#else
        if constexpr (NEED_DEPTH) {
            const u32x4 w = *reinterpret_cast<const u32x4_unaligned *>(reinterpret_cast<const DepthT *>(a.depth) + e);
            d[ch].x = w.x; d[ch].y = w.y; d[ch].z = w.z; d[ch].w = w.w;
        }
        if constexpr (HAS_MASK) {
#pragma unroll
            for (int i = 0; i < VEC / 4; ++i) mk[ch][i] = *reinterpret_cast<const u32_unaligned *>(a.mask + e + 4 * i);
        }
#endif
    }
    // confidence map (BASELINE config 4): vector loads issued with the others, under a uniform branch
    const bool use_conf = a.flags & DD_VALID_CONF;
    uint4 cfa[CH], cfb[CH];
    if (use_conf) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const long long e = eload[ch];
            if (a.conf_f16) {
                const unsigned short *c = reinterpret_cast<const unsigned short *>(a.conf) + e;
                if constexpr (VEC == 8) { const u32x4 w = LD_STREAM(reinterpret_cast<const u32x4_unaligned *>(c)); cfa[ch].x = w.x; cfa[ch].y = w.y; cfa[ch].z = w.z; cfa[ch].w = w.w; }
                else { const u32x2 w = LD_STREAM(reinterpret_cast<const u32x2_unaligned *>(c)); cfa[ch].x = w.x; cfa[ch].y = w.y; }
            } else {
                const float *c = reinterpret_cast<const float *>(a.conf) + e;
                { const u32x4 w = LD_STREAM(reinterpret_cast<const u32x4_unaligned *>(c)); cfa[ch].x = w.x; cfa[ch].y = w.y; cfa[ch].z = w.z; cfa[ch].w = w.w; }
                if constexpr (VEC == 8) { const u32x4 w = LD_STREAM(reinterpret_cast<const u32x4_unaligned *>(c + 4)); cfb[ch].x = w.x; cfb[ch].y = w.y; cfb[ch].z = w.z; cfb[ch].w = w.w; }
            }
        }
    }
    // ragged end of a view (P % VEC != 0): this wave holds it iff its span crosses P.  One lane re-reads its
    // partial vector element by element (indices clamped to the view, validity comes from nvalid).
    if ((a.P % (unsigned)VEC) != 0u && qw < a.P && qw + (unsigned)(64 * PXT) > a.P) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            if (nvalid[ch] != 0u && nvalid[ch] != (unsigned)VEC) {
                const unsigned qb = qw + (unsigned)(ch * 64 + lane) * VEC;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const unsigned q = qb + (unsigned)k < a.P ? qb + (unsigned)k : a.P - 1u;
                    const long long e = vbase + q;
                    if constexpr (NEED_DEPTH) {
                        unsigned raw;
                        if constexpr (sizeof(DepthT) == 2) raw = reinterpret_cast<const unsigned short *>(a.depth)[e];
                        else raw = reinterpret_cast<const unsigned *>(a.depth)[e];
                        set_raw<DepthT>(d[ch], k, raw);
                    }
                    if constexpr (HAS_MASK) {
                        unsigned &w = mk[ch][k >> 2];
                        w = (w & ~(0xffu << (8 * (k & 3)))) | ((unsigned)a.mask[e] << (8 * (k & 3)));
                    }
                    if (use_conf) {
                        if (a.conf_f16) set_raw<_Float16>(cfa[ch], k, reinterpret_cast<const unsigned short *>(a.conf)[e]);
                        else set_raw<float>(k < 4 ? cfa[ch] : cfb[ch], k & 3, reinterpret_cast<const unsigned *>(a.conf)[e]);
                    }
                }
            }
        }
    }
    const bool use_depth = NEED_DEPTH && (a.flags & DD_VALID_DEPTH_POSITIVE);
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        unsigned b = (1u << nvalid[ch]) - 1u;
        if constexpr (HAS_MASK) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (((mk[ch][k >> 2] >> (8 * (k & 3))) & 0xffu) == 0) b &= ~(1u << k);
        }
        if (use_depth) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (!(raw_depth<DepthT>(d[ch], k) > 0.0f)) b &= ~(1u << k);
        }
        if (use_conf) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float cf;
                if (a.conf_f16) cf = raw_depth<_Float16>(cfa[ch], k);
                else cf = raw_depth<float>(k < 4 ? cfa[ch] : cfb[ch], k & 3);
                if (!(cf > a.conf_thr)) b &= ~(1u << k);
            }
        }
        bits[ch] = b;
    }
}

#ifndef DD_LB_SLEEP
#define DD_LB_SLEEP 1
#endif
// ==================================================================================================
// The scan service (round 5).  In the decoupled look-back every tile's wave 0 polls its predecessors' granules through the memory
// pipeline of a CU whose other 23 waves are streaming rows: a polling round takes ~0.5 us whatever the window and whichever cache
// answers (16 / 32 / 64 lanes, pairs per 16-byte load, granules kept in the XCD's own L2: profiles/r05_ab_paired_polls.txt,
// r05_ab_xcd_runs.txt), the inclusive prefix advances 16 tiles per round, and a tile spends a third to a half of its life waiting
// for it.  Here ONE workgroup of the launch -- ticket 0 -- does nothing but the scan: its first wave reads the tiles' counts 256
// at a time as they are published (coalesced loads, four in flight), adds them up with a wave scan and publishes every tile's
// FIRST ROW; a tile publishes its count and polls one word, its own.  The chain of dependent round trips is gone: a tile's row is
// two hand-overs behind its count, whatever the tiles before it are doing.
// ==================================================================================================
__device__ __forceinline__ void scan_service(const KArgs &a, const unsigned epoch, const long long base, const int lane) {
    constexpr int K = 4;                              // loads in flight per lane: 256 tiles per round
    const unsigned T = a.num_tiles;
    const unsigned tag_a = TAG_AGG | epoch;
    const unsigned long long itag = (unsigned long long)(TAG_INCL | epoch) << EPOCH_SHIFT;
    const unsigned limit = a.spin_limit ? a.spin_limit : SPIN_LIMIT;
    unsigned f = 0;                                   // tiles [0, f) have their first row
    long long running = base;                         // first row of tile f
    unsigned spins = 0;
    bool poison = false;                              // from here on the rows are unknown (a tile never published; fault injection)
    if (a.chain) {
        // chained behind the previous call of this cloud, which runs on another stream and may not have finished its scan yet: the
        // batch starts where that call says it ends.  (This call's tiles load and count meanwhile; nothing of the previous call waits
        // for anything of this one, so waiting here cannot dead-lock as long as this launch cannot occupy every slot: ddcore.h)
        unsigned long long w;
        for (;;) {
            w = ld_state(a.chain);
            if ((unsigned)(w >> CHAIN_SHIFT) == (a.chain_seq & CHAIN_SEQ_MASK)) break;
            if (++spins > limit) { poison = true; break; }
            __builtin_amdgcn_s_sleep(DD_LB_SLEEP);
        }
        spins = 0;
        running = (long long)(w & ((1ull << CHAIN_SHIFT) - 1ull));
        if (running >= POISON_ROW) poison = true;     // the previous call did not know where it ended: neither does this one
        if (poison) {
            for (unsigned i = (unsigned)lane; i < T; i += 64u) st_state(&a.pref[i], itag | POISON);
            f = T;
        }
    }
    while (f < T) {
        unsigned long long s[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned idx = f + (unsigned)(k * 64 + lane);
            s[k] = idx < T ? ld_state(&a.gran[idx]) : 0ull;
        }
        unsigned got = 0;
        bool more = true;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (more) {
                const unsigned first = f + (unsigned)(k * 64);
                const unsigned cnt = first >= T ? 0u : (T - first < 64u ? T - first : 64u);
                const bool ready = (unsigned)lane < cnt && (unsigned)(s[k] >> EPOCH_SHIFT) == tag_a;
                const unsigned long long rb = __ballot(ready);
                const unsigned nr = rb == ~0ull ? 64u : (unsigned)__builtin_ctzll(~rb);      // tiles of this group ready IN ORDER
                if (nr) {
                    if (a.spin_limit == 0u && first + nr > 1u) poison = true;      // tuning bit 64: behave as if tile 1 had never published
                    const unsigned val = (unsigned)lane < nr ? ((unsigned)s[k] & 0xffffu) : 0u;
                    unsigned incl = val;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned y = __shfl_up(incl, o);
                        if (lane >= o) incl += y;
                    }
                    const bool bad = poison && !(a.spin_limit == 0u && first + (unsigned)lane == 0u);      // (injection: tile 0 keeps its row)
                    if ((unsigned)lane < nr)
                        st_state(&a.pref[first + (unsigned)lane], itag | (bad ? POISON : 0ull) | ((unsigned long long)(running + (long long)(incl - val)) & VAL_MASK));
                    running += (long long)(unsigned)__shfl((int)incl, (int)nr - 1);
                    got += nr;
                }
                more = nr == cnt && cnt == 64u;
            }
        }
        f += got;
        if (got) { spins = 0; continue; }
        if (++spins > limit) {                        // a tile never published its count: everything from here on is unknown
            poison = true;
            for (unsigned i = f + (unsigned)lane; i < T; i += 64u) st_state(&a.pref[i], itag | POISON);
            break;
        }
        __builtin_amdgcn_s_sleep(DD_LB_SLEEP);
    }
    if (poison && lane == 0) atomicExch(&a.hdr->error, 1);
    // the end of the call: nobody reads the cursor or the epoch any more (every tile has published its count with this epoch's tag);
    // when the epoch wraps, every word of the workspace is zeroed first -- once every tile has READ its row (tiles count themselves done)
    if (epoch == EPOCH_MAX) {
        unsigned patience = 0;
        while (__hip_atomic_load(&a.hdr->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T && ++patience < (1u << 24)) __builtin_amdgcn_s_sleep(8);
        for (unsigned long long i = (unsigned long long)lane; i < a.ws_words; i += 64ull) a.gran[i] = 0ull;
        if (lane == 0) a.hdr->done = 0u;
    }
    if (lane == 0) {
        const long long end = poison ? POISON_ROW : running;
        a.view_offsets[a.V] = poison ? -1ll : running;
        if (a.chain) {
            // the cursor for whoever reads it after the streams are joined -- written through, and acknowledged BEFORE the chain word
            // releases the next call (whose own cursor store must not be overtaken by this one) ...
            __hip_atomic_store(a.cursor_out, end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a.hdr->ticket = 0u;
            a.hdr->epoch = epoch == EPOCH_MAX ? 0u : epoch + 1u;
            st_state(a.chain, ((unsigned long long)((a.chain_seq + 1u) & CHAIN_SEQ_MASK) << CHAIN_SHIFT) | (unsigned long long)end);
        } else {
            close_call(a.hdr, a.cursor_out, end, epoch);
        }
    }
}
// a tile's side of it: wave 0 polls the tile's own word (all lanes the same address: one request)
__device__ __forceinline__ long long service_row(const KArgs &a, const unsigned t, const unsigned epoch, const int lane) {
    const unsigned tag_i = TAG_INCL | epoch;
    const unsigned limit = a.spin_limit ? a.spin_limit : SPIN_LIMIT;
    unsigned spins = 0;
    unsigned long long g;
    for (;;) {
        g = ld_state(&a.pref[t]);
        if ((unsigned)(g >> EPOCH_SHIFT) == tag_i) break;
        if (++spins > limit) {                        // the service never answered (its workgroup was parked for ~2 s)
            if (lane == 0) atomicExch(&a.hdr->error, 1);
            g = POISON;
            break;
        }
        __builtin_amdgcn_s_sleep(DD_LB_SLEEP);
    }
    if (epoch == EPOCH_MAX && lane == 0) __hip_atomic_fetch_add(&a.hdr->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (g & POISON) ? -1ll : (long long)(g & VAL_MASK);
}

// Look-back over tile aggregates of at most 14 bits (tiles of <= 12288 pixels).  Measured on MI355X in rounds 1-4: a poll costs per
// REQUEST (one per polling lane), whatever the window: 64 lanes x 1 granule 3.57 ms, x 4 5.50, x 8 6.09 (185 x 1080p, round 1);
// 16 lanes 3.02; 8 / 4 lanes -7 / -9 %; 32 / 64 lanes: the round takes proportionally longer (round 4).  Round 5: the BYTES of a
// request are free (a 16-byte poll costs what an 8-byte poll costs, profiles/r05_streaming_poll_width.txt), but fetching an aligned
// PAIR of the packed granules per lane -- 32 tiles per round with 16 requests, -DDD_LB_PAIR=1 -- is 0.5-2.7 % SLOWER on every
// workload, and 8 lanes x 2 equals 16 x 1 (profiles/r05_ab_paired_polls.txt): it is the 16 NEAREST predecessors that a round can
// use, however they are fetched.  DD_LB_PAIR stays 0.
// Aggregates fit 14 bits: their wave sum is taken with bit-sliced ballots (scalar popcounts) and the
// single inclusive value with a readlane -- no cross-lane data movement.
#ifndef DD_LB_LANES
#define DD_LB_LANES 16
#endif
#ifndef DD_LB_PAIR
#define DD_LB_PAIR 0
#endif
#ifndef DD_LB_SLEEP
#define DD_LB_SLEEP 1
#endif
constexpr int LB_LANES = DD_LB_LANES;   // lanes that poll in the steady state of a large batch (window = LB_LANES * LB_K tiles; KArgs.lb_lanes)

__device__ __forceinline__ long long lookback13(unsigned long long *gran, unsigned t, unsigned agg, long long base, int lane, int *err,
                                                const unsigned spin_limit, const unsigned epoch, const int lb_lanes, const bool inject_give_up) {
    // Every polling lane inspects one granule per round (DD_LB_PAIR = 0: the default), or fetches an aligned PAIR with one 16-byte
    // load, so that a round inspects 2 * lb_lanes tiles: `top` is then the odd index of lane 0's pair, the pair of lane l is
    // (top - 2l - 1, top - 2l), nearest tile first; in the first round lane 0's pair may hold the tile's own granule (t even): skipped.
    const unsigned tag_a = TAG_AGG | epoch, tag_i = TAG_INCL | epoch;
    const unsigned long long none = (unsigned long long)tag_a << EPOCH_SHIFT;      // "aggregate, 0 points": contributes nothing, never blocks
    const unsigned long long start = ((unsigned long long)tag_i << EPOCH_SHIFT) | (unsigned long long)base;   // tiles "before the first" hold the cursor
    const long long hi = (long long)t - 1;    // nearest predecessor
#if DD_LB_PAIR
    long long top = hi | 1ll;
#else
    long long top = hi;
#endif
    long long excl = 0;
    unsigned spins = 0;
    bool poisoned = false;                    // -> returns -1: the tile's first row is unknown, it must not write anything
    for (;;) {
        unsigned long long s[2] = {none, none};
#if !DD_LB_PAIR     // (A/B: one 8-byte granule per polling lane, as in rounds 1-4)
        if (lane < lb_lanes) s[0] = (top - lane >= 0) ? ld_state(gran + (top - lane)) : start;
#else
        if (lane < lb_lanes) {
            const long long i1 = top - 2ll * lane, i0 = i1 - 1;      // i1 odd, i0 even
            if (i1 >= 0) {
                const u32x4_t v = ld_state_pair(gran + i0);
                s[0] = ((unsigned long long)v.w << 32) | v.z;         // granule i1 (nearer)
                s[1] = ((unsigned long long)v.y << 32) | v.x;         // granule i0
                if (i1 > hi) s[0] = none;                             // the tile's own granule
            } else {
                s[0] = start;
            }
        }
#endif
        int incl_k = 2;                       // first inclusive granule among this lane's (nearest first)
        bool empty_before = false;            // an unpublished granule in front of it
        unsigned agg_before = 0;
        unsigned long long incl_val = 0;
#pragma unroll
        for (int k = 1; k >= 0; --k) {
            if ((unsigned)(s[k] >> EPOCH_SHIFT) == tag_i) { incl_k = k; incl_val = s[k] & (VAL_MASK | POISON); }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < incl_k) {
                const bool is_agg = (unsigned)(s[k] >> EPOCH_SHIFT) == tag_a;      // anything else (zero, another call's tag): not published
                empty_before |= !is_agg;
                agg_before += is_agg ? (unsigned)s[k] : 0u;   // aggregates: value bits only (< 2^14 each)
            }
        }
        const unsigned long long incl_b = __ballot(incl_k < 2);
        const int L = incl_b ? __builtin_ctzll(incl_b) : 64;
        const unsigned long long need = (L >= 63) ? ~0ull : ((2ull << L) - 1ull);   // lanes 0..L
        if (__ballot(empty_before) & need) {
            if (++spins > spin_limit) {
                if (lane == 0) atomicExch(err, 1);
                poisoned = true;
                break;
            }
            __builtin_amdgcn_s_sleep(DD_LB_SLEEP);
            continue;
        }
        const unsigned mine = (lane <= L) ? agg_before : 0u;       // <= 2 * 12288 < 2^16
        unsigned sum = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) sum += (unsigned)__popcll(__ballot((mine >> b) & 1u)) << b;
        excl += sum;
        if (incl_b) {
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)incl_val, L);
            const unsigned hi32 = __builtin_amdgcn_readlane((unsigned)(incl_val >> 32), L);
            const unsigned long long iv = ((unsigned long long)hi32 << 32) | lo;
            poisoned = (iv & POISON) != 0ull;
            excl += (long long)(iv & VAL_MASK);
            break;
        }
        top -= (DD_LB_PAIR ? 2ll : 1ll) * lb_lanes;
    }
    if (inject_give_up) {                     // tuning bit 64: this tile behaves as if its look-back had given up
        if (lane == 0) atomicExch(err, 1);
        poisoned = true;
    }
    if (lane == 0) st_state(&gran[t], ((unsigned long long)tag_i << EPOCH_SHIFT) | (poisoned ? POISON : 0ull) | ((unsigned long long)(excl + agg) & VAL_MASK));
    return poisoned ? -1ll : excl;
}

// ==================================================================================================
// Dense tiles.  Where every pixel of a tile survives (the inside of a mask blob, an unmasked depth map: all of BASELINE
// configs[4] and most tiles of a real scan) the compaction is the identity: point j of the tile is pixel j.  Such a
// tile builds no list.  Each wave keeps its own 1024 pixels:
//   * xyz: the raw depth vectors go to LDS as they were loaded (one ds_write_b128 per lane), are read back one
//     pixel per lane (x advances by 64 per step: no division), and the 12-byte rows are staged in LDS at the byte
//     phase they have in HBM, so that they leave as line-aligned 16-byte pieces: one wave store = 1 KiB = 8 whole
//     128-byte lines (the shape a plain fill reaches 6.9-7.0 TB/s with where 12-byte row stores reach 5.75,
//     profiles/r03_zone_interleave.txt); the bytes that share a line with the next chunk are carried over in LDS;
//   * normals / colours: the tile's rows are one contiguous run of the source map -- a shifted copy, 16 bytes
//     per lane, aligned on the store side;
//   * pixel / view indices: generated.
// Same arithmetic as the list path (same FMA chain), so the rows are bit-identical (tests/test_dense_tiles.py).
// ==================================================================================================
#ifndef DD_DENSE
#define DD_DENSE 1
#endif
#ifndef DD_GATHER_DEPTH
#define DD_GATHER_DEPTH 2         // point sets in flight per lane in the list path's sweeps (3: measured in round 4)
#endif
constexpr int DENSE_CHUNK = 256;                                  // rows staged per flush: 3072 B = 3 wave stores of 1 KiB
constexpr int DENSE_STAGE = DENSE_CHUNK * 12 + 128 + 16;          // + the line phase (< 128 B) + slack to a 16-byte multiple
constexpr int DENSE_LDS_PER_WAVE = DENSE_STAGE + 1024;            // + one 16-byte depth vector per lane
static_assert(L_WSPAN % DENSE_CHUNK == 0 && DENSE_STAGE % 16 == 0, "dense path geometry");

typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));

// a workgroup-uniform 64-bit value that came through LDS or memory, moved to SGPRs (addresses derived from it stay scalar)
__device__ __forceinline__ long long uniform64(long long x) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)x);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)x >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// LDS operations of one wave execute in issue order; this keeps the COMPILER from moving the LDS accesses of a
// lane across a point where the wave's lanes exchange data through LDS (no instruction is emitted).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Experiment build (-DDD_X_STAMPS): per tile, lane 0 of two waves writes shader-clock stamps of the phases to the buffer passed
// in DDViewBatch.refined_out (16 x 8 bytes per tile): where a tile's time goes (tools/ab_builds.py --stamps).
#ifdef DD_X_STAMPS
#define STAMP(slot) do { if (lane == 0 && a.refined_out) reinterpret_cast<unsigned long long *>(a.refined_out)[(size_t)t * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// Experiment knob (tools/ab_builds.py): at most DD_STORE_CAP vector-memory operations of a wave stay in flight behind a store.
#ifndef DD_STORE_CAP
#define DD_STORE_CAP -1
#endif
__device__ __forceinline__ void cap_stores() {
    if constexpr (DD_STORE_CAP >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DD_STORE_CAP) : "memory");
}

// One wave's contiguous run of one output array in "span coordinates": byte 0 is the 128-byte line that holds the
// run's first byte, the run is [lo, hi) with lo < 128.
struct Span {
    unsigned char *line;
    unsigned lo, hi;
};
__device__ __forceinline__ Span make_span(void *base, long long row0, unsigned row_bytes, unsigned rows) {
#ifdef DD_X_SINK    // experiment: every span lands in the first MiB of its array (stores stay in L2): what the kernel costs without its HBM writes
    unsigned char *g = reinterpret_cast<unsigned char *>(base) + ((row0 * (long long)row_bytes) & 0xFFFFCll);
#else
    unsigned char *g = reinterpret_cast<unsigned char *>(base) + row0 * (long long)row_bytes;
#endif
    const unsigned h = (unsigned)(reinterpret_cast<uintptr_t>(g) & 127u);
    Span s; s.line = g - h; s.lo = h; s.hi = h + rows * row_bytes;
    return s;
}
// the 16-byte piece at span offset o (a multiple of 16): whole if it lies inside the run, else its dwords / bytes that do
template <int GRAN>
__device__ __forceinline__ void put16(const Span &s, unsigned o, const uint4 &v) {
    if (o >= s.lo && o + 16u <= s.hi) { *reinterpret_cast<uint4 *>(s.line + o) = v; return; }
    if (o + 16u <= s.lo || o >= s.hi) return;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (GRAN == 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const unsigned b = o + 4u * c; if (b >= s.lo && b < s.hi) *reinterpret_cast<unsigned *>(s.line + b) = w[c]; }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) { const unsigned b = o + (unsigned)c; if (b >= s.lo && b < s.hi) s.line[b] = (unsigned char)(w[c >> 2] >> (8 * (c & 3))); }
    }
}
// sweep i of a span = its bytes [1024 i, 1024 (i + 1)): one 16-byte piece per lane.  `gen(o)` yields the piece at span offset o.
template <int GRAN, typename Gen>
__device__ __forceinline__ void span_sweep(const Span &s, int i, int lane, Gen gen) {
    const unsigned o = (unsigned)(i * 64 + lane) * 16u;
    const unsigned b0 = (unsigned)i * 1024u;
    if (b0 >= s.hi || b0 + 1024u <= s.lo) return;                               // wave-uniform: nothing of this sweep is ours
    if (b0 >= s.lo && b0 + 1024u <= s.hi) { *reinterpret_cast<uint4 *>(s.line + o) = gen(o); cap_stores(); return; }   // wave-uniform: all of it
    if (o + 16u > s.lo && o < s.hi) put16<GRAN>(s, o, gen(o));
    cap_stores();
}
// shifted copy: source byte 0 corresponds to span byte s.lo; NSW sweeps cover the nominal run + the line phase.
// Loads of a group of 4 sweeps are issued together.  A piece is loaded only if some of it is inside the run; a
// straddling piece loads its 16 bytes from a clamped address and shifts them (the source run has >= 16 bytes).
template <int GRAN, int NSW, typename VecT>
__device__ __forceinline__ void span_copy(const Span &s, const unsigned char *__restrict__ src, int lane) {
#pragma unroll 1
    for (int g = 0; g < NSW; g += 4) {          // a real loop: 4 loads in flight per lane, not 13
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (g + k < NSW) {
                const unsigned o = (unsigned)((g + k) * 64 + lane) * 16u;
                v[k] = make_uint4(0u, 0u, 0u, 0u);
                if (o >= s.lo && o + 16u <= s.hi) {
                    const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const VecT *>(src + (o - s.lo)));
                    v[k] = make_uint4(w.x, w.y, w.z, w.w);
                } else if (o + 16u > s.lo && o < s.hi) {          // straddles an end of the run: element by element
                    unsigned w[4] = {0u, 0u, 0u, 0u};
                    if constexpr (GRAN == 4) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const unsigned b = o + 4u * c; if (b >= s.lo && b < s.hi) w[c] = *reinterpret_cast<const unsigned *>(src + (b - s.lo)); }
                    } else {
#pragma unroll
                        for (int c = 0; c < 16; ++c) { const unsigned b = o + (unsigned)c; if (b >= s.lo && b < s.hi) w[c >> 2] |= (unsigned)src[b - s.lo] << (8 * (c & 3)); }
                    }
                    v[k] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (g + k < NSW) {
                const uint4 val = v[k];
                span_sweep<GRAN>(s, g + k, lane, [&](unsigned) { return val; });
            }
        }
    }
}

struct CamBlock { float m00, m01, m02, m10, m11, m12, m20, m21, m22, c0, c1, c2; };

// One wave's 1024 pixels of a dense tile.  `d`: the raw depth vectors as loaded (lane l, group ch: pixels
// qw + (64 ch + l) VEC ...); `lds`: this wave's DENSE_LDS_PER_WAVE bytes (16-byte aligned); `row0`: output row of pixel qw.
template <typename DepthT, bool HAS_NORMAL, bool HAS_RGB>
__device__ __forceinline__ void dense_wave(const KArgs &a, const uint4 (&d)[L_PXT / (16 / (int)sizeof(DepthT))], unsigned char *lds,
                                           const CamBlock &cam, unsigned v, long long vbase, unsigned q0, unsigned qw, long long row0, int lane) {
    constexpr int VEC = 16 / (int)sizeof(DepthT), CH = L_PXT / VEC, CSPAN = 64 * VEC, SUBS = CSPAN / DENSE_CHUNK;
    static_assert(CSPAN % DENSE_CHUNK == 0, "a depth vector group is a whole number of staged chunks");
    const long long room = a.capacity - row0;
    if (room <= 0) return;
    const unsigned rows = room < (long long)L_WSPAN ? (unsigned)room : (unsigned)L_WSPAN;   // < L_WSPAN only where the capacity cuts the cloud

    // ---- xyz ----
    if (a.out_xyz) {
        float *const stage_f = reinterpret_cast<float *>(lds);
        unsigned *const stage_u = reinterpret_cast<unsigned *>(lds);
        uint4 *const dvec = reinterpret_cast<uint4 *>(lds + DENSE_STAGE);
        const DepthT *const delem = reinterpret_cast<const DepthT *>(lds + DENSE_STAGE);
        const Span sx = make_span(a.out_xyz, row0, 12u, rows);
        const unsigned hq = sx.lo >> 2;                       // line phase in dwords (< 32)
        // (x, y) of this lane's first pixel: one division per lane and tile, then x += 64 per step (W >= 64)
        const unsigned W = (unsigned)a.W;
        const unsigned y0 = q0 / W;
        const unsigned r = qw + (unsigned)lane - y0 * W;     // < tile + W
        unsigned yo = (unsigned)((float)r * (1.0f / (float)W));
        if (yo * W > r) --yo; else if ((yo + 1) * W <= r) ++yo;
        float fy = (float)(y0 + yo), fx = (float)(r - yo * W);
        const float Wf = (float)W;
        int cc = 0;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            wave_lds_sync();
            dvec[lane] = d[ch];
            wave_lds_sync();
#pragma unroll
            for (int sub = 0; sub < SUBS; ++sub) {
                constexpr int STEPS = DENSE_CHUNK / 64;
                float dd[STEPS];
#pragma unroll
                for (int s = 0; s < STEPS; ++s) dd[s] = Elem<DepthT>::cvt(delem[sub * DENSE_CHUNK + s * 64 + lane]);   // pixel inside the group's span
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    float *o = stage_f + hq + (unsigned)(s * 64 + lane) * 3u;
                    o[0] = fmaf(dd[s], fmaf(cam.m00, fx, fmaf(cam.m01, fy, cam.m02)), cam.c0);
                    o[1] = fmaf(dd[s], fmaf(cam.m10, fx, fmaf(cam.m11, fy, cam.m12)), cam.c1);
                    o[2] = fmaf(dd[s], fmaf(cam.m20, fx, fmaf(cam.m21, fy, cam.m22)), cam.c2);
                    const float nx = fx + 64.0f;
                    const bool wrap = nx >= Wf;
                    fx = wrap ? nx - Wf : nx;
                    fy = wrap ? fy + 1.0f : fy;
                }
                wave_lds_sync();
                const unsigned base = (unsigned)cc * (unsigned)(DENSE_CHUNK * 12);
                Span sc;                                        // this chunk's window of the span, in LDS coordinates
                sc.line = sx.line + base;
                sc.lo = sx.lo > base ? sx.lo - base : 0u;
                sc.hi = sx.hi > base ? sx.hi - base : 0u;
#pragma unroll
                for (int i = 0; i < DENSE_CHUNK * 12 / 1024; ++i)
                    span_sweep<4>(sc, i, lane, [&](unsigned o) { return *reinterpret_cast<const uint4 *>(lds + o); });
                // the bytes behind the last whole line of this chunk open the next chunk's first line
                unsigned carry = 0;
                if ((unsigned)lane < hq) carry = stage_u[DENSE_CHUNK * 3 + lane];
                wave_lds_sync();
                if ((unsigned)lane < hq) stage_u[lane] = carry;
                ++cc;
            }
        }
        if (hq) {                                               // the last line's head
            wave_lds_sync();
            const unsigned base = (unsigned)(L_WSPAN * 12);
            Span sc;
            sc.line = sx.line + base;
            sc.lo = 0u;
            sc.hi = sx.hi > base ? sx.hi - base : 0u;
            const unsigned o = (unsigned)lane * 16u;
            if (o < 128u && o < sc.hi) put16<4>(sc, o, *reinterpret_cast<const uint4 *>(lds + o));
        }
    }
    // ---- normals: a shifted copy of 12 KiB (camera-frame normals pass through, scripts/test.py:220) ----
    if constexpr (HAS_NORMAL) {
        const Span sn = make_span(a.out_normal, row0, 12u, rows);
        span_copy<4, L_WSPAN * 12 / 1024 + 1, u32x4_a4>(sn, reinterpret_cast<const unsigned char *>(a.normal + (vbase + qw) * 3), lane);
    }
    // ---- colours: a shifted copy of 3 KiB, byte-granular ends ----
    if (HAS_RGB && a.out_rgb != nullptr) {
        const Span sc = make_span(a.out_rgb, row0, 3u, rows);
        span_copy<1, L_WSPAN * 3 / 1024 + 1, u32x4_a1>(sc, a.rgb + (vbase + qw) * 3, lane);
    }
    // ---- indices: generated ----
    if (a.out_pix) {
        const Span sp = make_span(a.out_pix, row0, 4u, rows);
#pragma unroll
        for (int i = 0; i < L_WSPAN * 4 / 1024 + 1; ++i)
            span_sweep<4>(sp, i, lane, [&](unsigned o) {
                const unsigned e = qw + (unsigned)((int)(o - sp.lo) >> 2);   // element of the piece's first dword (negative in front of the run: not stored)
                return make_uint4(e, e + 1u, e + 2u, e + 3u);
            });
    }
    if (a.out_view) {
        const Span sv = make_span(a.out_view, row0, 4u, rows);
        const unsigned id = (unsigned)(a.view_base + (int)v);
#pragma unroll
        for (int i = 0; i < L_WSPAN * 4 / 1024 + 1; ++i)
            span_sweep<4>(sv, i, lane, [&](unsigned) { return make_uint4(id, id, id, id); });
    }
}

// NW = waves per workgroup: 4 (4096-pixel tiles) for the two-pass scatter; the single-pass variant uses 12
// (12288-pixel tiles, 2 workgroups x 12 waves per CU) so that one look-back is amortised over three times the work.
constexpr int REFINE_MAX_KNOTS = 512;

// PXT = pixels per lane: 16, or 8 in the single-pass instantiation for small batches (a lane then makes half as many sweeps
// over the tile's points: the sweeps are a chain of memory latencies that nothing hides when a CU holds one workgroup)
template <typename DepthT, bool HAS_MASK, bool SINGLE_PASS, bool HAS_NORMAL, bool HAS_RGB, int NW, bool REFINE = false, int PXT = L_PXT>
__global__ __launch_bounds__(64 * NW, DD_LEAN_WGS) void compact_lean(const KArgs a) {
    constexpr int BT = 64 * NW;             // threads per workgroup
    constexpr int WSPAN = 64 * PXT;         // pixels per wave
    constexpr int LT = NW * WSPAN;          // pixels per tile
    constexpr int VEC = 16 / (int)sizeof(DepthT), CH = PXT / VEC, CSPAN = 64 * VEC;
    // rows written past the L2 (non-temporal) in the small-batch instantiation: what a kernel leaves dirty in the L2s is written
    // back at its end, and a chain of small calls pays that once per call (1 us of 27 per one-view call, profiles/r05_streaming_*.txt)
    // (... and in the fused refine instantiation, whose L2 also serves the halo's re-reads: 17.1 -> 17.0 us per view, profiles/r06_fused_refine.txt)
    constexpr bool NT_ROWS = DD_NT_STORE != 0 || PXT != L_PXT || REFINE;
    // the point list: depth (float) + 16-bit pixel per listed point.  One raw block, because the fused refine stage uses
    // the same bytes, BEFORE the list exists, for the transformed values of the tile and its halo (up to LT*6/4 floats)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LT * 6];
    float *const s_d = reinterpret_cast<float *>(s_raw);                              // 4 B per pixel
    unsigned short *const s_q = reinterpret_cast<unsigned short *>(s_raw + LT * 4);  // 2 B per pixel
    // the fused refine stage's transfer curve (ddrefine_math.h): knots {x, y}, reciprocal widths of the intervals, the grid of buckets.
    // (LDS is the budget: 73728 + 7680 + ... = 81.5 KB -- two workgroups still share a CU's 160 KiB)
    __shared__ float2 s_kxy[REFINE ? REFINE_MAX_KNOTS : 1];
    __shared__ float s_inv[REFINE ? REFINE_MAX_KNOTS : 1];
    __shared__ unsigned short s_grid[REFINE ? ddmath::CURVE_GRID_WORDS : 2];
    __shared__ unsigned s_tot[NW];
    __shared__ long long s_excl;
    __shared__ unsigned s_ticket;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

#ifdef DD_X_STAMPS
    const unsigned long long entry_clk = __builtin_amdgcn_s_memtime(), entry_rt = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned t, epoch = 0u;
    long long base = 0;
    if constexpr (SINGLE_PASS) {
        epoch = a.hdr->epoch;                  // the tag of this call's granules (advanced by close_call)
        base = load_cursor(a.cursor);          // the row the batch starts from, read before this tile publishes anything
        if (a.static_tiles) {
            t = blockIdx.x;
        } else {
            if (tid == 0) s_ticket = atomicAdd(&a.hdr->ticket, 1u);
            __syncthreads();
            t = __builtin_amdgcn_readfirstlane(s_ticket);
        }
        if (a.scan_service) {                  // the launch has one workgroup more than tiles: the first one is the scan service
            if (t == 0u) {
                if (wave == 0) scan_service(a, epoch, base, lane);
                return;
            }
            t -= 1u;
        }
        if (t >= a.num_tiles) return;          // (never: the grid is num_tiles workgroups)
#ifdef DD_REFINE_XCD_RUN
        // experiment: runs of DD_REFINE_XCD_RUN consecutive tiles on one XCD (workgroup b runs on XCD b mod 8), so that the halo rows two
        // neighbouring tiles both read come out of that XCD's L2 the second time (profiles/r06_fused_refine.txt: the halo's re-reads
        // reach the HBM today, 1.6 B/px).  A tile then waits for up to 8 (R - 1) workgroups dispatched behind it.
        if constexpr (REFINE) {
            constexpr unsigned R = DD_REFINE_XCD_RUN, BLK = 8u * R;
            if (a.static_tiles && a.scan_service && t < a.num_tiles - a.num_tiles % BLK) {
                const unsigned blk = t / BLK, j = t % BLK, x = (j + 1u) & 7u, i = j >> 3;      // (tile workgroups start at blockIdx 1: XCD (j + 1) mod 8)
                t = blk * BLK + x * R + i;
            }
        }
#endif
    } else {
#if DD_XCD_SWIZZLE
        {   // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of tiles
            const unsigned nt = a.num_tiles, b = blockIdx.x, x = b & 7u, q = nt >> 3, r = nt & 7u;
            t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        }
#elif DD_REVERSE_SCATTER
        t = a.num_tiles - 1u - blockIdx.x;   // pass 1 read tiles 0..T-1: start where its reads are still in the Infinity Cache
#else
        t = blockIdx.x;
#endif
        if (a.order_regions > 1u) {
            // tiles of K distant stretches of the batch taken in turn: consecutive workgroups write K regions of the output
            // that lie a K-th of the cloud apart -- in different 1 GiB chunks, i.e. (placed by the arena) in different classes
            // of HBM.  Only the scatter pass can do this: its rows are known before it starts.
            const unsigned K = a.order_regions, per = (a.num_tiles + K - 1u) / K;
            const unsigned b = blockIdx.x, r = b % K, i = b / K;
            t = r * per + i;
            if (t >= a.num_tiles) return;
        }
    }
    const unsigned v = t / a.tiles_per_view;
    const unsigned tv = t - v * a.tiles_per_view;
    const long long vbase = (long long)v * a.hw;
    const unsigned q0 = tv * (unsigned)LT;
    const unsigned qw = q0 + (unsigned)wave * WSPAN;

    uint4 d[CH];
    unsigned bits[CH];
    if (wave == 0) STAMP(0);
    if (wave == 5) STAMP(8);
#ifdef DD_X_STAMPS
    if (wave == 0 && lane == 0 && a.refined_out) {
        unsigned long long *st = reinterpret_cast<unsigned long long *>(a.refined_out) + (size_t)t * 16;
        st[5] = entry_clk; st[6] = entry_rt;
    }
#define STAMP_END() do { if (wave == 0 && lane == 0 && a.refined_out) reinterpret_cast<unsigned long long *>(a.refined_out)[(size_t)t * 16 + 7] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP_END() do { } while (0)
#endif
    if constexpr (REFINE) {
        static_assert(sizeof(DepthT) == 4 && SINGLE_PASS, "the fused refine stage runs in the float32 single-pass instantiation");
        // ---- src/depthdensifier/depth_refiner.py:180-205 on this tile, from the RAW depth (scripts/test.py:179-194 fused in) ----
        const DDViewParams *vp = a.params + v;
        const int nk = vp->n_knots < REFINE_MAX_KNOTS ? vp->n_knots : REFINE_MAX_KNOTS;      // wave-uniform; 2..512 by contract (the clamp keeps a bad value inside the LDS arrays)
        // transformed values of the tile plus W + 1 pixels either side (a 3x3 window of a pixel of the tile reaches one
        // row up and down); replicate padding = clamped coordinates, so nothing outside the view is ever needed
        float *const s_val = reinterpret_cast<float *>(s_raw);
        const unsigned Wd = (unsigned)a.W, Hd = (unsigned)a.H;
#ifdef DD_X_NO_HALO      // experiment (tools/bench_fused_refine.py --variants): what the halo's look-ups cost (wrong medians at the tile's ends)
        const unsigned lo = (q0 > Wd + 1u ? q0 - (Wd + 1u) : 0u) & ~3u;
        const unsigned hi = (q0 + (unsigned)LT + Wd + 1u) < a.P ? (q0 + (unsigned)LT + Wd + 1u) : a.P;
        const unsigned lo_eval = q0, hi_eval = (q0 + (unsigned)LT) < a.P ? q0 + (unsigned)LT : a.P;
#else
        const unsigned hi = (q0 + (unsigned)LT + Wd + 1u) < a.P ? (q0 + (unsigned)LT + Wd + 1u) : a.P;
        // (the window starts on a multiple of four values where that fits the LDS -- it always does below 3070 columns: a lane's four
        //  pixels then sit on a 16-byte boundary of s_val whenever the width is a multiple of four, and are read with one LDS access)
        const unsigned lo_raw = q0 > Wd + 1u ? q0 - (Wd + 1u) : 0u;
        const unsigned lo = hi - (lo_raw & ~3u) <= (unsigned)(LT * 6 / 4) ? (lo_raw & ~3u) : lo_raw;
        const unsigned lo_eval = lo, hi_eval = hi;
#endif
        // Rounds 5-6: this stage is bound by the vector ALU, not by memory (24 us per 1080p view where the plain kernel takes 14).  The
        // curve's interval is found through a grid of buckets (the knots that share the depth's bucket: ~1), the blend reads both knots
        // with one LDS access and multiplies by the interval's reciprocal width instead of dividing (ddrefine_math.h), four values per
        // lane and step with their bisections in lock step; four consecutive windows share their sorted columns (median9x4).
#ifdef DD_X_NO_BUILD     // experiment (with DD_X_NO_LUT): what building the curve per tile costs
        ddmath::Curve curve; curve.kxy = s_kxy; curve.inv = s_inv; curve.grid = s_grid; curve.n = nk; curve.x0 = 0.0f; curve.scale = 0.0f;
#else
        const ddmath::Curve curve = ddmath::curve_build(s_kxy, s_inv, s_grid, vp->knots_x, vp->knots_y, nk, tid, BT);
#endif
        // (tried in round 2: batches of 7 pixels per lane with their loads issued together and a fixed-trip lockstep search --
        // 6 % SLOWER on the same box; the other waves of the CU already hide these latencies and the early-exit search does less work)
        bool any_nan = false;
        {
            // four consecutive values per lane and step: one wide load of the depth, one of the mask, four look-ups in lock step.
            // Scalar bases + 32-bit byte offsets (the window is < 2^16 values): no 64-bit address arithmetic per lane.
            const unsigned cnt = hi_eval - lo;
            const unsigned char *const dbase = reinterpret_cast<const unsigned char *>(a.depth) + (vbase + lo) * (a.raw_f16 ? 2 : 4);
            const unsigned char *const mbase = a.mask ? a.mask + vbase + lo : nullptr;
            constexpr int EV = DD_REFINE_EV;             // values per lane and step (4, or 8: two wide loads and eight look-ups in lock step)
            for (unsigned e0 = (unsigned)tid * (unsigned)EV + ((lo_eval - lo) & ~3u); e0 < cnt; e0 += (unsigned)BT * (unsigned)EV) {
                float raw[EV], val[EV];
                bool mk[EV];
#pragma unroll
                for (int g = 0; g < EV; g += 4) {
                    const unsigned e4 = e0 + (unsigned)g;
                    if (e4 + 4u <= cnt) {
                        if (a.raw_f16) {
                            const u32x2 w = *reinterpret_cast<const u32x2_unaligned *>(dbase + e4 * 2u);
                            raw[g + 0] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w.x & 0xffffu)); raw[g + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w.x >> 16));
                            raw[g + 2] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w.y & 0xffffu)); raw[g + 3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w.y >> 16));
                        } else {
                            const u32x4 w = *reinterpret_cast<const u32x4_unaligned *>(dbase + e4 * 4u);
                            raw[g + 0] = __uint_as_float(w.x); raw[g + 1] = __uint_as_float(w.y); raw[g + 2] = __uint_as_float(w.z); raw[g + 3] = __uint_as_float(w.w);
                        }
                        const unsigned m4 = mbase ? *reinterpret_cast<const u32_unaligned *>(mbase + e4) : 0u;
#pragma unroll
                        for (int k = 0; k < 4; ++k) mk[g + k] = mbase ? ((m4 >> (8 * k)) & 0xffu) != 0u : (raw[g + k] > 0.0f);      // depth_refiner.py:238-241
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool in = e4 + (unsigned)k < cnt;
                            const unsigned e = in ? e4 + (unsigned)k : 0u;
                            raw[g + k] = a.raw_f16 ? (float)*reinterpret_cast<const _Float16 *>(dbase + e * 2u) : *reinterpret_cast<const float *>(dbase + e * 4u);
                            mk[g + k] = in && (mbase ? (mbase[e] != 0) : (raw[g + k] > 0.0f));
                        }
                    }
                }
#ifdef DD_X_NO_LUT       // experiment: what the look-ups cost
#pragma unroll
                for (int k = 0; k < EV; ++k) val[k] = mk[k] ? raw[k] : 0.0f;
#else
                ddmath::curve_evalN<EV>(curve, raw, mk, val);                                 // :185-191
#endif
#pragma unroll
                for (int g = 0; g < EV; g += 4) {
                    const unsigned e4 = e0 + (unsigned)g;
                    any_nan |= (val[g] != val[g]) | (val[g + 1] != val[g + 1]) | (val[g + 2] != val[g + 2]) | (val[g + 3] != val[g + 3]);
                    if (e4 + 4u <= cnt) *reinterpret_cast<float4 *>(s_val + e4) = make_float4(val[g], val[g + 1], val[g + 2], val[g + 3]);
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (e4 + (unsigned)k < cnt) s_val[e4 + k] = val[g + k];
                    }
                }
            }
        }
        const bool tile_nan = __syncthreads_or((int)any_nan) != 0;            // (a NaN anywhere in the tile or its halo: the windows take the careful path)
#ifdef DD_X_NO_MEDIAN    // experiment: what the windows cost
        const bool smooth = false;
#else
        const bool smooth = vp->skip_smoothing == 0;
#endif
        float ref[CH][VEC];
        const bool quads = (lo & 3u) == 0u && (Wd & 3u) == 0u;      // wave-uniform: every lane's four pixels are one image row's, 16-byte aligned in s_val
        unsigned y0g = 0u, x0g = 0u;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const unsigned qb = qw + (unsigned)(ch * 64 + lane) * VEC;
            unsigned y, x;
            if (ch == 0) { y0g = qb / Wd; x0g = qb - y0g * Wd; }             // (one division per lane: the lane's next group lies CSPAN pixels on)
            else { x0g += (unsigned)CSPAN; while (x0g >= Wd) { x0g -= Wd; ++y0g; } }
            y = y0g; x = x0g;
            unsigned b = 0;
            // four windows of one image row, no NaN in the tile: their six columns are sorted once.  At the row's ends the outer
            // columns are the replicate padding (clamped coordinates) -- no separate path for them: a wave holds half a row or more,
            // so nearly every wave has a lane at a row's end, and a divergent slow path would be executed by all of them (round 6)
            const bool fast = smooth && !tile_nan && VEC == 4 && qb + 4u <= a.P && x + 4u <= Wd;
            if (fast) {
                const unsigned ym = y ? y - 1u : 0u, yp = y + 1u < Hd ? y + 1u : Hd - 1u;
                const unsigned c0 = x ? x - 1u : 0u, c5 = x + 4u < Wd ? x + 4u : Wd - 1u;
                float v0[6], v1[6], v2[6];
                const unsigned b0 = ym * Wd - lo, b1 = y * Wd - lo, b2 = yp * Wd - lo;      // (mod 2^32: a row may start in front of the window, the sums below never do)
                if (quads) {
                    // the four own columns of each row with ONE 16-byte LDS read (dword reads at a lane stride of four collide four ways in
                    // the banks); the outer two columns are the neighbouring lanes' -- a wave shift -- and come from the LDS only at the
                    // wave's two ends; at a row's ends they are the replicate padding, i.e. the lane's own outer columns
                    const float4 q0v = *reinterpret_cast<const float4 *>(s_val + (b0 + x)), q1v = *reinterpret_cast<const float4 *>(s_val + (b1 + x)),
                                 q2v = *reinterpret_cast<const float4 *>(s_val + (b2 + x));
                    v0[1] = q0v.x; v0[2] = q0v.y; v0[3] = q0v.z; v0[4] = q0v.w;
                    v1[1] = q1v.x; v1[2] = q1v.y; v1[3] = q1v.z; v1[4] = q1v.w;
                    v2[1] = q2v.x; v2[2] = q2v.y; v2[3] = q2v.z; v2[4] = q2v.w;
                    auto from_left = [&](float mine, unsigned at) {     // column x - 1: lane - 1 holds it as its fourth column
                        float v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x138, 0xF, 0xF, false));     // wave_shr:1
                        if (lane == 0) v = s_val[at];
                        return v;
                    };
                    auto from_right = [&](float mine, unsigned at) {    // column x + 4: lane + 1 holds it as its first column
                        float v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x130, 0xF, 0xF, false));     // wave_shl:1
                        if (lane == 63) v = s_val[at];
                        return v;
                    };
                    const bool first = x == 0u, last = x + 4u == Wd;
                    const float l0 = from_left(v0[4], b0 + c0), l1 = from_left(v1[4], b1 + c0), l2 = from_left(v2[4], b2 + c0);
                    const float r0 = from_right(v0[1], b0 + c5), r1 = from_right(v1[1], b1 + c5), r2 = from_right(v2[1], b2 + c5);
                    v0[0] = first ? v0[1] : l0; v1[0] = first ? v1[1] : l1; v2[0] = first ? v2[1] : l2;
                    v0[5] = last ? v0[4] : r0; v1[5] = last ? v1[4] : r1; v2[5] = last ? v2[4] : r2;
                } else {
                    v0[0] = s_val[b0 + c0]; v1[0] = s_val[b1 + c0]; v2[0] = s_val[b2 + c0];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v0[k + 1] = s_val[b0 + x + (unsigned)k]; v1[k + 1] = s_val[b1 + x + (unsigned)k]; v2[k + 1] = s_val[b2 + x + (unsigned)k]; }
                    v0[5] = s_val[b0 + c5]; v1[5] = s_val[b1 + c5]; v2[5] = s_val[b2 + c5];
                }
                float m4[4];
                ddmath::median9x4(v0, v1, v2, m4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float r = (v1[k + 1] != 0.0f) ? m4[k] : 0.0f;       // :203
                    if (r > 0.0f) b |= 1u << k;
                    ref[ch][k] = r;
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if (fast) break;
                const unsigned q = qb + (unsigned)k;
                float r = 0.0f;
                if (q < a.P) {
                    const float c = s_val[q - lo];
                    r = c;
                    if (smooth) {                                             // :194-200
                        const unsigned ym = y ? y - 1u : 0u, yp = y + 1u < Hd ? y + 1u : Hd - 1u;
                        const unsigned xm = x ? x - 1u : 0u, xp = x + 1u < Wd ? x + 1u : Wd - 1u;
                        float w9[9];
                        w9[0] = s_val[ym * Wd + xm - lo]; w9[1] = s_val[ym * Wd + x - lo]; w9[2] = s_val[ym * Wd + xp - lo];
                        w9[3] = s_val[y * Wd + xm - lo];  w9[4] = c;                        w9[5] = s_val[y * Wd + xp - lo];
                        w9[6] = s_val[yp * Wd + xm - lo]; w9[7] = s_val[yp * Wd + x - lo]; w9[8] = s_val[yp * Wd + xp - lo];
                        r = ddmath::median9(w9);
                    }
                    r = (c != 0.0f) ? r : 0.0f;      // :203 zero outside the mask (a masked pixel's value is >= 1e-3 or NaN, never 0)
                    if (r > 0.0f) b |= 1u << k;       // scripts/test.py:194 + :210: mask AND refined depth > 0
                }
                ref[ch][k] = r;
                if (++x == Wd) { x = 0; ++y; }
            }
            bits[ch] = b;
            d[ch].x = __float_as_uint(ref[ch][0]); d[ch].y = __float_as_uint(ref[ch][1]);
            d[ch].z = __float_as_uint(ref[ch][2]); d[ch].w = __float_as_uint(ref[ch][3]);
#ifdef DD_X_NO_REFOUT
            if (false) {
#else
            if (a.refined_out) {                       // the filter's cache (scripts/test.py:197-201)
#endif
                float *o = a.refined_out + vbase + qb;
                // (written once, read by the filter much later: past the L2 -- 17.7 -> 17.0 us per 1080p view, profiles/r06_fused_refine.txt)
                if (qb + (unsigned)VEC <= a.P && ((vbase + qb) & 3) == 0) __builtin_nontemporal_store(u32x4{d[ch].x, d[ch].y, d[ch].z, d[ch].w}, reinterpret_cast<u32x4 *>(o));
                else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) if (qb + (unsigned)k < a.P) o[k] = ref[ch][k];
                }
            }
        }
        __syncthreads();                               // s_val is dead from here: the point list takes its place
    } else {
        lean_load_test<DepthT, HAS_MASK, true, PXT>(a, vbase, qw, lane, d, bits);
    }

    int lane_pre[CH], tot[CH], m = 0;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        wave_rank<VEC>(bits[ch], lane, lane_pre[ch], tot[ch]);
        m += tot[ch];
    }
    if (lane == 0) s_tot[wave] = (unsigned)m;
    if (wave == 0) STAMP(1);
    if (wave == 5) STAMP(9);
    __syncthreads();   // #1
    if (wave == 0) STAMP(2);

    unsigned wbase = 0, n = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wave) wbase += s_tot[w];
        n += s_tot[w];
    }
    if constexpr (SINGLE_PASS) {                       // publish the aggregate as early as possible
        if (tid == 0) st_state(&a.gran[t], ((unsigned long long)(TAG_AGG | epoch) << EPOCH_SHIFT) | (unsigned long long)n);
    } else {
        // tuning bit 17: the rows of this tile were assigned without a count, on the assumption that every pixel is valid --
        // this pass reads the validity inputs anyway, so it is also the check; one miss voids the batch (error word 2)
        if (a.assume_dense) {
            const unsigned there = a.P - q0 < (unsigned)LT ? a.P - q0 : (unsigned)LT;
            if (n != there) {
                if (tid == 0) atomicExch(&a.hdr->error, 2);
                return;
            }
        }
    }
    // wave 0: the tile's first output row by ticket order (decoupled look-back), the view offsets it defines
    auto look_back = [&]() {
        // (fault injection, tuning bit 64: spin_limit is 0 -- a tile that would have to wait gives up at once -- and every eighth
        // tile behaves as if it had.  A tile whose look-back gave up, or that read the row of one that did, gets -1: it writes
        // nothing, so the rows of everything that was appended BEFORE stay intact, and its successors learn the same from its granule)
        if (a.scan_service) {                  // the tile's row comes from the service workgroup, which also closes the call
            const long long e = service_row(a, t, epoch, lane);
            if (lane == 0) {
                s_excl = e;
                if (tv == 0) a.view_offsets[v] = e;
            }
            return;
        }
        const long long e = lookback13(a.gran, t, n, base, lane, &a.hdr->error, a.spin_limit, epoch, (int)a.lb_lanes,
                                       a.spin_limit == 0u && (t & 7u) == 1u);
        if (lane == 0) {
            s_excl = e;
            if (tv == 0) a.view_offsets[v] = e;
            if (t == a.num_tiles - 1) a.view_offsets[a.V] = e + n;
        }
        // the end of the call (see close_call): the batch's last tile closes it -- or, when the epoch wraps, whoever finishes last
        if (epoch == EPOCH_MAX) wrap_call(a.hdr, a.gran, a.ws_words, a.cursor_out, a.num_tiles, lane);
        else if (lane == 0 && t == a.num_tiles - 1) close_call(a.hdr, a.cursor_out, e < 0 ? POISON_ROW : e + n, epoch);
    };
#if DD_DENSE
    if constexpr (!REFINE && PXT == L_PXT) {
        if (n == (unsigned)LT && a.dense_ok) {         // workgroup-uniform: every pixel of the tile survives -> no list (dense_wave)
            static_assert(DENSE_LDS_PER_WAVE * NW <= LT * 6, "the dense path's staging fits the list's LDS");
            long long e0;
            if constexpr (SINGLE_PASS) {
                if (wave == 0) { look_back(); STAMP(3); }
                __syncthreads();
                if (wave == 5) STAMP(10);
                e0 = uniform64(s_excl);
                if (e0 < 0) return;                      // the tile's first row is unknown (a look-back gave up): nothing is written
            } else {
                e0 = uniform64(a.view_offsets[v] + (long long)a.tiles[t].off);
            }
            const float *cp = reinterpret_cast<const float *>(a.params + v);
            CamBlock cam;
            cam.m00 = cp[0]; cam.m01 = cp[1]; cam.m02 = cp[2]; cam.m10 = cp[3]; cam.m11 = cp[4]; cam.m12 = cp[5];
            cam.m20 = cp[6]; cam.m21 = cp[7]; cam.m22 = cp[8]; cam.c0 = cp[9]; cam.c1 = cp[10]; cam.c2 = cp[11];
            dense_wave<DepthT, HAS_NORMAL, HAS_RGB>(a, d, s_raw + wave * DENSE_LDS_PER_WAVE, cam, v, vbase, q0, qw,
                                                    e0 + (long long)wave * L_WSPAN, lane);
            if (wave == 0) STAMP(4);
            if (wave == 5) STAMP(11);
            STAMP_END();
            return;
        }
    }
#endif
    {   // tile-local list of surviving pixels, in output order
        int cum = (int)wbase;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            int r = cum + lane_pre[ch];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                if ((bits[ch] >> k) & 1u) {
                    s_q[r] = (unsigned short)(wave * WSPAN + ch * CSPAN + lane * VEC + k);
                    s_d[r] = raw_depth<DepthT>(d[ch], k);
                    ++r;
                }
            }
            cum += tot[ch];
        }
    }
    long long excl = 0;     // first output row of the tile; known after the look-back / from pass 1
    // ---- one lane per output point; gathers of group i+1 in flight while group i is stored ----
    // Gathers are unconditional and branch-free (lanes past the end re-read point 0) so that the
    // loop is straight-line code and the compiler can keep counted vmcnt waits.
    const float *par = reinterpret_cast<const float *>(a.params + v);
    const float m00 = par[0], m01 = par[1], m02 = par[2], m10 = par[3], m11 = par[4], m12 = par[5];
    const float m20 = par[6], m21 = par[7], m22 = par[8], c0 = par[9], c1 = par[10], c2 = par[11];
    const float r00 = par[12], r01 = par[13], r02 = par[14], r10 = par[15], r11 = par[16], r12 = par[17];
    const float r20 = par[18], r21 = par[19], r22 = par[20];
    const bool rotate = a.flags & DD_ROTATE_NORMALS;
    const unsigned W = (unsigned)a.W;
    const unsigned y0 = q0 / W;
    const unsigned rowstart = y0 * W;
    const float invW = 1.0f / (float)W;
    const float *__restrict__ nsrc = a.normal + vbase * 3;
    const unsigned char *__restrict__ csrc = a.rgb + vbase * 3;

    // Which listed point a lane handles in sweep i.  Sweep 0 takes the tile's first BT points as they come (its
    // gathers are issued before the tile's first row is known).  From sweep 1 on the points are shifted by `hrot`
    // rows so that every wave's run of 64 rows starts on a 128-byte line of the (N,3) float32 outputs (32 rows = 3
    // lines): a wave's store instruction then covers 6 whole lines instead of touching 7 (tools/experiments/ubench_tile.hip:
    // 5.5 vs 4.8 TB/s of row stores).  The hrot points skipped behind sweep 0 are taken by the lanes that the
    // shift pushes past the end of the list.
    int hrot = 0;
    auto point_of = [&](int i) -> int {
        if (i == 0) return tid < (int)n ? tid : -1;
        const int p = i * BT + tid + hrot;
        if (p < (int)n) return p;
        const int end = (int)n > BT + hrot ? (int)n : BT + hrot;
        const int g = BT + (p - end);
        return (g < BT + hrot && g < (int)n) ? g : -1;
    };
    // what a lane holds of a point between its gathers and its stores: the depth and the pixel (xyz is computed when the row is
    // stored: two registers less per point in flight than carrying x, y, z), the gathered normal and colour
    struct Pt { float d; f32x3 nr; unsigned rgbw; unsigned q; int j; };
    auto prep = [&](int i, Pt &p) {
        const int j = point_of(i);
        p.j = j;
        const int jj = j >= 0 ? j : 0;
        const unsigned q = q0 + s_q[jj];
        p.d = s_d[jj];
        p.q = q;
        // scalar base + 32-bit byte offset (a view has < 2^28 pixels on this path): no 64-bit address arithmetic per lane
        if constexpr (HAS_NORMAL) p.nr = *reinterpret_cast<const f32x3 *>(reinterpret_cast<const unsigned char *>(nsrc) + q * 12u);
        if constexpr (HAS_RGB) {
            // the 3 colour bytes with ONE unaligned dword load that stays inside the view's image:
            // bytes [3q-1, 3q+3) for q > 0 (colour in the upper 3 bytes), [0, 4) for q == 0
            p.rgbw = *reinterpret_cast<const u32_unaligned *>(csrc + (q * 3u - (q ? 1u : 0u)));
        }
    };
    // the tile's rows start at scalar addresses (excl is workgroup-uniform); a lane adds a 32-bit byte offset
    unsigned char *xb = nullptr, *nb = nullptr, *cb = nullptr, *kb = nullptr, *ib = nullptr, *vb = nullptr;
    unsigned cap_rows = 0;            // rows of this tile the capacity leaves room for
    auto set_row_bases = [&]() {
#ifdef DD_X_SINK
        xb = reinterpret_cast<unsigned char *>(a.out_xyz) + ((excl * 12) & 0xFFFFCll);
        nb = reinterpret_cast<unsigned char *>(a.out_normal) + ((excl * 12) & 0xFFFFCll);
        cb = reinterpret_cast<unsigned char *>(a.out_rgb) + ((excl * 3) & 0xFFFFFll);
#else
        xb = reinterpret_cast<unsigned char *>(a.out_xyz) + excl * 12;
        nb = reinterpret_cast<unsigned char *>(a.out_normal) + excl * 12;
        cb = reinterpret_cast<unsigned char *>(a.out_rgb) + excl * 3;
#endif
        kb = reinterpret_cast<unsigned char *>(a.out_packed) + excl * 16;
        ib = reinterpret_cast<unsigned char *>(a.out_pix) + excl * 4;
        vb = reinterpret_cast<unsigned char *>(a.out_view) + excl * 4;
        const long long room = a.capacity - excl;
        cap_rows = room <= 0 ? 0u : room < (long long)LT ? (unsigned)room : (unsigned)LT;
    };
    auto emit = [&](const Pt &p) {
        const int j = p.j;
        const unsigned uj = (unsigned)j;
        const bool act = (j >= 0) && (uj < cap_rows);
        const unsigned r = p.q - rowstart;               // < 12288 + W
        unsigned yo = (unsigned)((float)r * invW);
        if (yo * W > r) --yo; else if ((yo + 1) * W <= r) ++yo;
        const float fy = (float)(y0 + yo), fx = (float)(r - yo * W);
        const float px = fmaf(p.d, fmaf(m00, fx, fmaf(m01, fy, m02)), c0);
        const float py = fmaf(p.d, fmaf(m10, fx, fmaf(m11, fy, m12)), c1);
        const float pz = fmaf(p.d, fmaf(m20, fx, fmaf(m21, fy, m22)), c2);
        if (act) {
            f32x3 o; o.x = px; o.y = py; o.z = pz;
            if (a.out_xyz) {
                if constexpr (NT_ROWS) __builtin_nontemporal_store(o, reinterpret_cast<f32x3 *>(xb + uj * 12u));
                else *reinterpret_cast<f32x3 *>(xb + uj * 12u) = o;
            }
            if (a.out_packed) {     // one aligned 16-byte store per point: a wave writes 1 KiB of whole lines
                unsigned c = 0xff000000u;
                if constexpr (HAS_RGB) c |= p.q ? (p.rgbw >> 8) : (p.rgbw & 0xffffffu);
                uint4 rec; rec.x = __float_as_uint(px); rec.y = __float_as_uint(py); rec.z = __float_as_uint(pz); rec.w = c;
                *reinterpret_cast<uint4 *>(kb + uj * 16u) = rec;
            }
            if (a.out_pix) *reinterpret_cast<int *>(ib + uj * 4u) = (int)p.q;
            if (a.out_view) *reinterpret_cast<int *>(vb + uj * 4u) = a.view_base + (int)v;
        }
        if constexpr (HAS_NORMAL) {
            f32x3 nv = p.nr;
            if (rotate) {
                const float w0 = r00 * nv.x + r01 * nv.y + r02 * nv.z;
                const float w1 = r10 * nv.x + r11 * nv.y + r12 * nv.z;
                const float w2 = r20 * nv.x + r21 * nv.y + r22 * nv.z;
                const float inv = 1.0f / (sqrtf(w0 * w0 + w1 * w1 + w2 * w2) + 1e-8f);
                nv.x = w0 * inv; nv.y = w1 * inv; nv.z = w2 * inv;
            }
            if constexpr (NT_ROWS) { if (act) __builtin_nontemporal_store(nv, reinterpret_cast<f32x3 *>(nb + uj * 12u)); }
            else { if (act) *reinterpret_cast<f32x3 *>(nb + uj * 12u) = nv; }
        }
        if (HAS_RGB && a.out_rgb != nullptr) {
            // rows of 4 consecutive lanes -> 12 contiguous bytes, stored by the quad's first lane
            const unsigned c0w = p.q ? (p.rgbw >> 8) : (p.rgbw & 0xffffffu);
            const unsigned c1w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0w, 0xF9, 0xF, 0xF, false);  // quad_perm [1,2,3,3]
            const unsigned c2w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0w, 0xFE, 0xF, 0xF, false);  // quad_perm [2,3,3,3]
            const unsigned c3w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c0w, 0xFF, 0xF, 0xF, false);  // quad_perm [3,3,3,3]
            // a quad stores together only if its four lanes hold four consecutive rows (not so where the shifted
            // sweep wraps around to the skipped points)
            const int jq = __builtin_amdgcn_update_dpp(0, j, 0x00, 0xF, 0xF, false);               // quad_perm [0,0,0,0]
            const unsigned long long am = __ballot(act && j == jq + (lane & 3));
            const unsigned quad = (unsigned)(am >> (lane & ~3)) & 0xFu;
            unsigned char *dst = cb + uj * 3u;
            if (quad == 0xFu) {
                if ((lane & 3) == 0) {
                    u32x3 w;
                    w.x = c0w | (c1w << 24);
                    w.y = (c1w >> 8) | (c2w << 16);
                    w.z = (c2w >> 16) | (c3w << 8);
                    *reinterpret_cast<u32x3_unaligned *>(dst) = w;
                }
            } else if (act) {        // ragged end of the run (or capacity cut): byte stores
                dst[0] = (unsigned char)c0w; dst[1] = (unsigned char)(c0w >> 8); dst[2] = (unsigned char)(c0w >> 16);
            }
        }
        cap_stores();
    };

    constexpr int NI = LT / BT;       // 16 point slots per lane at most
    Pt pa, pb;
    if constexpr (SINGLE_PASS) {
        // Wave 0 looks back while the other waves already issue their first gathers (the list is complete
        // after this barrier; neither the list nor the gathers need the tile's first row).
        if (wave == 5) STAMP(12);
        __syncthreads();
        if (wave == 0) {
            look_back();
            STAMP(3);
        } else if (n != 0) {
            prep(0, pa);
        }
        __syncthreads();
        if (wave == 5) STAMP(10);
        excl = uniform64(s_excl);
        if (n == 0 || excl < 0) return;                      // (excl < 0: the tile's first row is unknown -- a look-back gave up: nothing is written)
        if (wave == 0) prep(0, pa);
    } else {
        __syncthreads();
        excl = uniform64(a.view_offsets[v] + (long long)a.tiles[t].off);
        if (n == 0) return;
        prep(0, pa);
    }
    set_row_bases();
#ifdef DD_X_PRIO      // experiment: the row phase's waves issue in front of the co-resident workgroup's compute phases
    if constexpr (REFINE) __builtin_amdgcn_s_setprio(DD_X_PRIO);
#endif
    if (a.align_runs) hrot = (int)((0ll - excl) & 31ll);
#if DD_GATHER_DEPTH == 3
    // the gathers of TWO sweeps in flight behind the sweep that is being stored (three point sets taking turns)
    Pt pc;
    prep(1, pb);
#pragma unroll
    for (int i = 0; i < NI; i += 3) {
        if (i + 2 < NI) prep(i + 2, pc);
        emit(pa);
        if ((i + 1) * BT >= (int)n) break;
        if (i + 3 < NI) prep(i + 3, pa);
        emit(pb);
        if ((i + 2) * BT >= (int)n) break;
        if (i + 4 < NI) prep(i + 4, pb);
        emit(pc);
        if ((i + 3) * BT >= (int)n) break;
    }
#else
#pragma unroll
    for (int i = 0; i < NI; i += 2) {
        prep(i + 1, pb);
        emit(pa);
        if ((i + 1) * BT >= (int)n) break;
        if (i + 2 < NI) prep(i + 2, pa);
        emit(pb);
        if ((i + 2) * BT >= (int)n) break;
    }
#endif
    if (wave == 0) STAMP(4);
    if (wave == 5) STAMP(11);
    STAMP_END();
}

// ---- the gate of a chained call that is too large to wait inside its own workgroups (DDViewBatch.chain): ONE wave, in front of the
// call on the call's stream, returns when the previous call of the cloud -- on the other stream -- has finished its SCAN.  The call's
// kernel then starts beside the previous call's last tiles (which are still writing rows) instead of behind them, and none of its
// workgroups ever holds a slot while it waits for another launch.  (Gives up after ~2 s like every wait here; the call's own scan
// then waits once more, and poisons the call if the word still does not come.) ----
__global__ __launch_bounds__(64) void chain_gate(const unsigned long long *chain, const unsigned seq) {
    unsigned spins = 0;
    while ((unsigned)(__hip_atomic_load(chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> CHAIN_SHIFT) != (seq & CHAIN_SEQ_MASK)) {
        if (++spins > SPIN_LIMIT) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

// ---- do two streams run side by side?  (dd_streams_overlap)  HIP deals its streams to a few hardware queues, and two streams on ONE
// queue run strictly one after the other: calls chained across them would gain nothing (measured: 26 instead of 17 us per one-view
// call, profiles/r05_streaming_queue_collision.txt).  `overlap_wait` on stream A waits up to 1 ms of wall clock for the flag that
// `overlap_set`, launched behind it on stream B, raises: it sees the flag only if B's kernel ran while A's was running. ----
__global__ __launch_bounds__(64) void overlap_wait(int *w) {
    if (threadIdx.x) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    int seen = 0;
    for (;;) {
        if (__hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { seen = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000ull) break;
        __builtin_amdgcn_s_sleep(8);
    }
    w[1] = seen;
}
__global__ __launch_bounds__(64) void overlap_set(int *w) {
    if (threadIdx.x == 0) __hip_atomic_store(&w[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the count-free plan (tuning bit 17): every visited pixel taken as valid -- tile t of a view starts tv * tile rows
// into the view, view v starts v * P rows behind the cursor.  The scatter pass verifies it (assume_dense above). ----
__global__ __launch_bounds__(256) void plan_dense(const KArgs a, const unsigned tile) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < a.num_tiles) {
        const unsigned v = i / a.tiles_per_view, tv = i - v * a.tiles_per_view;
        a.tiles[i].off = tv * tile;
        a.tiles[i].cnt = a.P - tv * tile < tile ? a.P - tv * tile : tile;
    }
    if (blockIdx.x == 0) {                            // (the cursor is read by this workgroup only, and written after it has read it)
        const long long c = *a.cursor;
        __syncthreads();
        for (int v = threadIdx.x; v <= a.V; v += 256) a.view_offsets[v] = c + (long long)v * a.P;
        for (int v = threadIdx.x; v < a.V; v += 256) a.view_tot[v] = a.P;
        if (threadIdx.x == 0 && a.cursor_out) *a.cursor_out = c + (long long)a.V * a.P;
    }
}

// ---- pass 1 of the two-pass mode (and dd_count_valid on stride-1 maps) ----------------------------
template <typename DepthT, bool HAS_MASK, bool NEED_DEPTH>
__global__ __launch_bounds__(BLOCK) void count_lean(const KArgs a) {
    constexpr int VEC = 16 / (int)sizeof(DepthT), CH = L_PXT / VEC;
    __shared__ unsigned s_tot[WAVES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned t = blockIdx.x;
    const unsigned v = t / a.tiles_per_view;
    const unsigned tv = t - v * a.tiles_per_view;
    uint4 d[CH];
    unsigned bits[CH];
    lean_load_test<DepthT, HAS_MASK, NEED_DEPTH>(a, (long long)v * a.hw, tv * (unsigned)L_TILE + (unsigned)wave * L_WSPAN, lane, d, bits);
    unsigned cnt = 0;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) cnt += __popc(bits[ch]);
    unsigned tot = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) tot += (unsigned)__popcll(__ballot((cnt >> b) & 1u)) << b;   // cnt <= 16
    if (lane == 0) s_tot[wave] = tot;
    __syncthreads();
    if (tid == 0) {
        unsigned n = 0;
        for (int w = 0; w < WAVES; ++w) n += s_tot[w];
        if (a.tiles) a.tiles[t].cnt = n;
        if (a.counts && n) atomicAdd(&a.counts[v], (unsigned long long)n);
    }
}

// Exclusive scan of one view's tile counts (one workgroup per view) + the view's total.
__global__ __launch_bounds__(BLOCK) void scan_view_tiles(const KArgs a) {
    __shared__ unsigned s_w[WAVES];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned v = blockIdx.x;
    TileCO *rec = a.tiles + (size_t)v * a.tiles_per_view;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (unsigned b = 0; b < a.tiles_per_view; b += BLOCK) {
        const unsigned i = b + tid;
        const unsigned x = i < a.tiles_per_view ? rec[i].cnt : 0u;
        unsigned incl = x;                           // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        unsigned pre = s_carry;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        if (i < a.tiles_per_view) rec[i].off = pre + incl - x;
        __syncthreads();
        if (tid == BLOCK - 1) s_carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0) a.view_tot[v] = (long long)s_carry;
}

// Exclusive scan of the per-view totals (one workgroup): absolute first row of every view.
__global__ __launch_bounds__(BLOCK) void scan_views(const KArgs a) {
    __shared__ long long s_w[WAVES];
    __shared__ long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = *a.cursor;
    __syncthreads();
    for (int b = 0; b < a.V; b += BLOCK) {
        const int i = b + tid;
        const long long x = i < a.V ? a.view_tot[i] : 0ll;
        long long incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        long long pre = s_carry;
        for (int w = 0; w < wave; ++w) pre += s_w[w];
        if (i < a.V) a.view_offsets[i] = pre + incl - x;
        __syncthreads();
        if (tid == BLOCK - 1) s_carry = pre + incl;
        __syncthreads();
    }
    if (tid == 0) {
        a.view_offsets[a.V] = s_carry;
        if (a.cursor_out) *a.cursor_out = s_carry;      // read at the top, written last, by the only workgroup
    }
}

// Small batches (a single view streamed per call): both scans in ONE workgroup, one launch less.
__global__ __launch_bounds__(BLOCK) void scan_small(const KArgs a) {
    __shared__ unsigned s_w[WAVES];
    __shared__ unsigned s_carry;
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = *a.cursor;
    for (int v = 0; v < a.V; ++v) {
        TileCO *rec = a.tiles + (size_t)v * a.tiles_per_view;
        if (tid == 0) s_carry = 0;
        __syncthreads();
        for (unsigned b = 0; b < a.tiles_per_view; b += BLOCK) {
            const unsigned i = b + tid;
            const unsigned x = i < a.tiles_per_view ? rec[i].cnt : 0u;
            unsigned incl = x;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned y = __shfl_up(incl, o);
                if (lane >= o) incl += y;
            }
            if (lane == 63) s_w[wave] = incl;
            __syncthreads();
            unsigned pre = s_carry;
            for (int w = 0; w < wave; ++w) pre += s_w[w];
            if (i < a.tiles_per_view) rec[i].off = pre + incl - x;
            __syncthreads();
            if (tid == BLOCK - 1) s_carry = pre + incl;
            __syncthreads();
        }
        if (tid == 0) {
            a.view_offsets[v] = s_base;
            s_base += (long long)s_carry;
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.view_offsets[a.V] = s_base;
        if (a.cursor_out) *a.cursor_out = s_base;
    }
}

// ---- generic pass 1 / dd_count_valid (scripts/test.py:210-212 "valid_pixels", no points produced) -
template <typename DepthT>
__global__ __launch_bounds__(BLOCK) void count_generic(const KArgs a) {
    __shared__ int s_part[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned t = blockIdx.x;
    const int v = (int)(t / a.tiles_per_view);
    const unsigned tv = t - (unsigned)v * a.tiles_per_view;
    float dval[G_CHUNKS];
    unsigned vbits[G_CHUNKS];
    generic_load_test<DepthT>(a, v, tv * (unsigned)G_TILE, tid, dval, vbits);
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < G_CHUNKS; ++c) cnt += (int)vbits[c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) s_part[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < WAVES; ++w) tot += s_part[w];
        if (a.tiles) a.tiles[t].cnt = (unsigned)tot;
        if (a.counts && tot) atomicAdd(&a.counts[v], (unsigned long long)tot);
    }
}

// ---- host side -----------------------------------------------------------------------------------
thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

// DDViewBatch.tuning bits (0 = defaults): the caller's choices, DD_TUNE_* of ddcore.h
constexpr unsigned TUNE_FORCE_GENERIC = DD_TUNE_GENERIC;     // scalar kernels even on aligned stride-1 maps
constexpr unsigned TUNE_TWO_PASS = DD_TUNE_TWO_PASS;          // dd_unproject_compact: plan + scatter even on the lean path
constexpr unsigned TUNE_SINGLE_PASS = DD_TUNE_SINGLE_PASS;    // dd_unproject_compact: the single-pass kernel (default on the lean path)
constexpr unsigned TUNE_ASSUME_DENSE = DD_TUNE_ASSUME_DENSE;  // dd_unproject_compact, lean path: scatter against a count-free plan, verified by the scatter

// The experiment switches (ddcore_lab.h): a thread-local word, zero unless a test or an A/B tool has set it on this thread.
thread_local uint32_t g_lab = 0u;

// (every member has an initialiser, and make_plan starts from Plan{}: up to round 5 `chain_gate` was written for chained calls only, an
// unchained call read whatever its caller's stack held there, and a non-zero byte launched the gate kernel with a NULL chain word --
// the rare "Memory access fault ... address (nil)" of the three-rank rehearsal; tests/c_client/plan_stack_test.cpp)
struct Plan {
    bool refine = false;    // DD_REFINE: the fused refine stage (single-pass lean kernel, float32)
    bool f16 = false;
    bool lean = false;      // stride-1 maps (any size >= one vector) -> lean kernels; otherwise the generic scalar kernels
    bool single = false;    // dd_unproject_compact runs the single-pass kernel
    int tile = 0;
    int sp_pxt = 0;         // lean single-pass kernel: pixels per lane (the tile is SP_WAVES * 64 * sp_pxt pixels)
    bool chain_gate = false;  // a chained call too large to wait inside its own workgroups: chain_gate runs in front of it
};

// Workgroups an UNGATED chained call may have: three quarters of the slots the current device offers the instantiation such a call
// runs with (12 waves, 8 pixels per lane; the variant with the most registers), from the runtime's occupancy calculator and the
// device's CU count -- a partitioned GPU (CPX: 32 CUs) or another build setting (DD_LEAN_WGS, DD_SP_WAVES) moves the limit with
// it.  A device property, looked up once per device; 0 = unknown: every chained call is gated.
unsigned chain_wg_limit() {
    static std::atomic<int> cache[64];               // per device ordinal: 0 = not looked up yet, -1 = unknown, else the limit
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0u; }
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess
            || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, compact_lean<float, true, true, true, true, SP_WAVES, false, SP_PXT_SMALL>, 64 * SP_WAVES, 0) != hipSuccess
            || cus < 1 || per_cu < 1) {
            (void)hipGetLastError();
            v = -1;
        } else {
            v = (int)(((long long)cus * per_cu * 3) / 4);
            if (v < 1) v = -1;
        }
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0 ? (unsigned)v : 0u;
}

int make_plan(const DDViewBatch *b, KArgs &a, Plan &p) {
    p = Plan{};
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
    if (b->tuning & ~(uint32_t)DD_TUNE_ALL) return fail(DD_ERR_INVALID_ARG, "tuning: reserved bits set (the experiment switches moved to ddcore_lab.h with ABI 14)");
    const uint32_t lab = g_lab;
    const long long hw = (long long)b->height * b->width;
    if (hw >= (1ll << 31)) return fail(DD_ERR_UNSUPPORTED, "views of 2^31 pixels or more are not supported");

    memset(&a, 0, sizeof(a));
    a.depth = b->depth; a.mask = b->mask; a.conf = b->conf; a.normal = b->normal; a.rgb = b->rgb;
    a.params = b->params;
    a.hw = hw; a.V = b->num_views; a.H = b->height; a.W = b->width; a.stride = b->stride;
    a.Hs = (int)(((long long)b->height + b->stride - 1) / b->stride);      // (64-bit: height + stride may pass 2^31 -- tests/c_client/host_limits_test.cpp)
    a.Ws = (int)(((long long)b->width + b->stride - 1) / b->stride);
    a.P = (unsigned)((long long)a.Hs * a.Ws);
    a.conf_thr = b->conf_threshold; a.flags = b->flags; a.conf_f16 = (b->conf_dtype == DD_F16);
    a.view_base = b->view_index_base;

    p.f16 = (b->depth_dtype == DD_F16);
    p.refine = (b->flags & DD_REFINE) != 0;
#ifdef DD_X_STAMPS
    a.refined_out = b->refined_out;
#endif
    if (p.refine) {
        // the fused refine stage: float32 single-pass lean kernel, raw depth read element by element
        if (b->stride != 1) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE needs stride 1 (at a coarser density refine with dd_refine_apply first)");
        if (b->flags & DD_VALID_CONF) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE does not combine with DD_VALID_CONF");
        if (b->width > 3071 || hw < 4) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE handles widths up to 3071 (tile + halo must fit the LDS)");
        if (b->tuning & (TUNE_FORCE_GENERIC | TUNE_TWO_PASS)) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE runs in the single-pass kernel only");
        if ((b->flags & DD_VALID_MASK) && !b->mask) return fail(DD_ERR_INVALID_ARG, "DD_VALID_MASK set but mask is NULL");
        a.refined_out = b->refined_out;
        a.raw_f16 = p.f16;
        if (!(b->flags & DD_VALID_MASK)) a.mask = nullptr;      // mask = raw depth > 0 (depth_refiner.py:241)
        p.f16 = false;
    }
    const int vec = p.f16 ? 8 : 4;
    // lean kernels: stride 1, any view size of at least one vector; pointers need element alignment only (views of
    // H*W % vec != 0 pixels start off the 16-byte grid anyway; the wide loads are element-aligned)
    // (a lane addresses a view's attributes with 32-bit byte offsets: views of 2^28 pixels or more take the generic path)
    bool aligned = (b->stride == 1) && (hw >= vec) && (hw < (1ll << 28)) && (p.refine || (uintptr_t)b->depth % (p.f16 ? 2 : 4) == 0);
    if (b->flags & DD_VALID_CONF) aligned = aligned && ((uintptr_t)b->conf % (b->conf_dtype == DD_F16 ? 2 : 4) == 0);
    if (b->tuning & TUNE_FORCE_GENERIC) aligned = false;
    p.lean = aligned;
    // fused call: single-pass on the lean path (measured 8-23 % faster than plan + scatter on MI355X with
    // 12288-pixel tiles and a 16-granule look-back window), two-pass on the generic path
    // (the scalar kernels run single-pass only on request, and only on views below 2^30 pixels: what a two-pass call leaves in a
    // workspace must read as "not published" to a later single-pass call -- see TileCO)
    p.single = ((b->tuning & TUNE_SINGLE_PASS) != 0 && (p.lean || hw < (1ll << 30))) || (p.lean && (p.refine || !(b->tuning & (TUNE_TWO_PASS | TUNE_ASSUME_DENSE))));
    a.align_runs = (lab & DD_LAB_LIST_ORDER) == 0;  // lab: rows in list order (A/B of the line-aligned sweeps)
    // DD_TUNE_DENSE_TILES: tiles whose pixels all survive take the list-free path (dense_wave).  Off by default: measured in the
    // real kernel it is the same rows with a third of the instructions, and 0-5 % SLOWER.
    // The dense path steps x by 64 per lane (width >= 64) and derives rows in float32 (width < 2^23).
    a.dense_ok = (b->tuning & DD_TUNE_DENSE_TILES) != 0 && b->width >= 64 && b->width < (1 << 23);
    a.order_regions = 1u + ((b->tuning >> 8) & 63u);       // DD_TUNE_INTERLEAVE: the scatter pass interleaves 2 .. 64 stretches of tiles
    a.spin_limit = (lab & DD_LAB_FAULT_INJECT) ? 0u : SPIN_LIMIT;   // lab, fault injection: a tile that would have to wait for a
                                                          // predecessor gives up at once, and every eighth tile behaves as if it had:
                                                          // wrong rows, the workspace's error word set (tests of the caller's recovery)
    // The lean single-pass kernel (round 5).  The tiles' first rows come from the scan service: one workgroup of the launch -- the one
    // that draws ticket 0 -- scans the counts, every tile polls its own row (scan_service).  Tiles are drawn by TICKET, so a tile only
    // ever waits for workgroups that are running: safe whatever else shares the GPU.  (tuning bit 22 takes tiles by workgroup index
    // instead -- the ticket counter is one address that every workgroup of the launch hits: 2.5 % on 185 x 1080p -- but then a
    // resident tile can wait for a workgroup that has no slot yet, and two such launches on one GPU, two processes or two streams,
    // can hold each other's slots until the spin limit ends it: three ranks sharing a box's GPU did exactly that.)
    // tuning bit 26 = the decoupled look-back of rounds 1-4 instead of the service (bits 20-21: 2 / 3 = 32 / 64 polling lanes instead
    // of 16).  The tile goes by the size of the batch: 16 pixels per lane, 12288-pixel tiles -- or, for a streamed view or a few
    // (scripts/test.py:131 densifies one view per loop iteration: the launch does not fill the chip and a tile's lifetime IS the
    // kernel's), 8 pixels per lane, 8192-pixel tiles of 16 waves, rows written past the L2.  tuning bits 18-19: 1 / 3 force the small / the large
    // tile.  profiles/r05_ab_scan_service*.txt (service + workgroup index against the look-back of round 4): 1.02-1.03x on 185 x 1080p,
    // 1.06x on a per-pixel confidence cull, 1.09x on 100 x 12 MP, 1.05-1.2x on 8 views, 1.14x on one.
    {
        const unsigned long long big_tiles = (unsigned long long)((a.P + SP_WAVES * L_WSPAN - 1) / (SP_WAVES * L_WSPAN)) * (unsigned)a.V;
        const unsigned tsel = (b->tuning >> 18) & 3u, wsel = (lab & DD_LAB_POLL_LANES_64) ? 3u : (lab & DD_LAB_POLL_LANES_32) ? 2u : 0u;
        const bool small = (tsel == 1u || (tsel == 0u && big_tiles <= SP_SMALL_BATCH_TILES)) && !p.refine;
        p.sp_pxt = small ? SP_PXT_SMALL : L_PXT;
        a.scan_service = p.lean && (lab & DD_LAB_LOOKBACK) == 0u;
        a.static_tiles = (int)((b->tuning >> 22) & 1u);
        a.lb_lanes = wsel == 2u ? 32u : wsel == 3u ? 64u : (unsigned)LB_LANES;
        if (b->chain) {
            // a chained call's workgroups occupy slots while its scan waits for the previous call: it must never be able to occupy
            // all of them, whatever else of this cloud is in flight -- at most three quarters of the device's slots for this
            // instantiation (chain_wg_limit: 384 of the 512 of a whole MI355X, 48 of the 64 of a CPX partition), the scan included
            // (counted in the 12-wave small tile a call that waits in its scan runs with: 6144 pixels)
            const unsigned tile_px = p.sp_pxt == SP_PXT_SMALL ? SP_WAVES * 64 * SP_PXT_SMALL : SP_WAVES * 64 * L_PXT;
            const unsigned long long wgs = (unsigned long long)((a.P + tile_px - 1) / tile_px) * (unsigned)a.V + 1ull;
            if (!p.lean || !a.scan_service || p.refine || !p.single)
                return fail(DD_ERR_UNSUPPORTED, "DDViewBatch.chain needs the single-pass kernel with the scan service (stride-1 maps, no DD_REFINE, tuning without 1 / 4 / bit 17 / bit 26)");
            // a larger call gets a GATE in front of it (chain_gate): one wave that returns when the previous call's scan is over, so
            // that this call's workgroups exist only once they have nothing to wait for
            p.chain_gate = wgs > (unsigned long long)chain_wg_limit();
            a.chain_gated = p.chain_gate ? 1 : 0;
            a.chain = reinterpret_cast<unsigned long long *>(b->chain);
            a.chain_seq = (unsigned)(b->chain_seq & (int64_t)CHAIN_SEQ_MASK);
        }
    }
    p.tile = p.lean ? L_TILE : G_TILE;      // tiling of the two-pass kernels (count / plan / scatter); the finest single-pass tiling is the same
    a.tiles_per_view = (a.P + p.tile - 1) / p.tile;
    const unsigned long long nt = (unsigned long long)a.tiles_per_view * (unsigned)a.V;
    if (nt >= (1ull << 31) - 1ull) return fail(DD_ERR_UNSUPPORTED, "too many tiles in one batch; split the batch");      // (+ 1 workgroup: the scan service)
    a.num_tiles = (unsigned)nt;
    return DD_OK;
}

// workspace: [WsHeader: 16 B sticky + 48 B single-pass state][8 B per tile of the FINEST tiling, padded to 16: look-back granules][the same again: the
// tiles' first rows from the scan service][8 B per tile: count, first row][8 B per view]
int64_t ws_bytes(const KArgs &a) {
    return (int64_t)sizeof(WsHeader) + 2 * (((int64_t)a.num_tiles * 8 + 15) & ~(int64_t)15) + (int64_t)a.num_tiles * 8 + (int64_t)a.V * 8;
}

int bind_workspace(KArgs &a, void *workspace, int64_t workspace_bytes) {
    if (!workspace || ((uintptr_t)workspace % 16) != 0) return fail(DD_ERR_WORKSPACE, "workspace is NULL or not 16-byte aligned");
    if (workspace_bytes < ws_bytes(a)) return fail(DD_ERR_WORKSPACE, "workspace too small (see dd_workspace_bytes)");
    char *w = reinterpret_cast<char *>(workspace);
    a.hdr = reinterpret_cast<WsHeader *>(w);
    const size_t gran_bytes = ((size_t)a.num_tiles * 8 + 15) & ~(size_t)15;
    a.gran = reinterpret_cast<unsigned long long *>(w + sizeof(WsHeader));
    a.pref = reinterpret_cast<unsigned long long *>(w + sizeof(WsHeader) + gran_bytes);
    a.tiles = reinterpret_cast<TileCO *>(w + sizeof(WsHeader) + 2 * gran_bytes);
    a.view_tot = reinterpret_cast<long long *>(w + sizeof(WsHeader) + 2 * gran_bytes + (size_t)a.num_tiles * 8);
    a.ws_words = (unsigned long long)((workspace_bytes - (int64_t)sizeof(WsHeader)) / 8);   // all of it: what other calls left behind, too
    return DD_OK;
}

int bind_output(KArgs &a, const DDViewBatch *batch, const DDCloudOut *out) {
    if (!out || (!out->xyz && !out->xyz_rgba)) return fail(DD_ERR_INVALID_ARG, "out is NULL or neither out->xyz nor out->xyz_rgba is given");
    if (out->xyz_rgba && ((uintptr_t)out->xyz_rgba % 16) != 0) return fail(DD_ERR_INVALID_ARG, "out->xyz_rgba must be 16-byte aligned");
    if (out->capacity < 0) return fail(DD_ERR_INVALID_ARG, "capacity is negative");
    if (out->normal && !batch->normal) return fail(DD_ERR_INVALID_ARG, "out->normal requested but batch->normal is NULL");
    if (out->rgb && !batch->rgb) return fail(DD_ERR_INVALID_ARG, "out->rgb requested but batch->rgb is NULL");
    a.out_xyz = out->xyz; a.out_normal = out->normal; a.out_rgb = out->rgb;
    a.out_pix = out->pixel_index; a.out_view = out->view_index; a.capacity = out->capacity;
    a.out_packed = out->xyz_rgba;
    if (a.out_normal && (a.flags & DD_ROTATE_NORMALS)) a.dense_ok = 0;   // rotated normals are computed per point: list path
    if (a.out_packed) a.dense_ok = 0;                                    // the 16-byte record is one aligned store per point already: list path
    return DD_OK;
}

template <typename DepthT, bool SP, bool HM, bool HN, int NW, int PXT>
void launch_lean4(const KArgs &a, hipStream_t s) {
    const unsigned K = SP ? 1u : a.order_regions;
    const dim3 grid(K * ((a.num_tiles + K - 1u) / K) + (SP && a.scan_service ? 1u : 0u)), block(64 * NW);
    if (a.rgb && (a.out_rgb || a.out_packed)) hipLaunchKernelGGL((compact_lean<DepthT, HM, SP, HN, true, NW, false, PXT>), grid, block, 0, s, a);   // colours gathered
    else hipLaunchKernelGGL((compact_lean<DepthT, HM, SP, HN, false, NW, false, PXT>), grid, block, 0, s, a);
}

// sp_pxt: pixels per lane of the single-pass kernel (the tile is SP_WAVES * 64 * sp_pxt pixels): L_PXT for a batch that fills
// the chip several times over, SP_PXT_SMALL for a streamed view or a few (make_plan)
template <typename DepthT, bool SP, bool HM, bool HN>
void launch_lean3(const KArgs &a, hipStream_t s, int sp_pxt) {
    if constexpr (SP) {
        // (a chained call that waits inside its scan workgroup -- a single view -- keeps the 12-wave workgroup: beside the previous
        // call's tail it is the faster of the two, 3.2 against 3.6-4.0 ms per 185 calls; everywhere else 16 waves win)
        if (sp_pxt == SP_PXT_SMALL && a.chain && !a.chain_gated) launch_lean4<DepthT, SP, HM, HN, SP_WAVES, SP_PXT_SMALL>(a, s);
        else if (sp_pxt == SP_PXT_SMALL) launch_lean4<DepthT, SP, HM, HN, SP_WAVES_SMALL, SP_PXT_SMALL>(a, s);
        else launch_lean4<DepthT, SP, HM, HN, SP_WAVES, L_PXT>(a, s);
    } else {
        launch_lean4<DepthT, SP, HM, HN, WAVES, L_PXT>(a, s);
    }
}

template <typename DepthT, bool SP>
void launch_lean(const KArgs &a, hipStream_t s, int sp_pxt) {
    const bool hm = a.flags & DD_VALID_MASK, hn = a.out_normal != nullptr;
    if (hm && hn) launch_lean3<DepthT, SP, true, true>(a, s, sp_pxt);
    else if (hm) launch_lean3<DepthT, SP, true, false>(a, s, sp_pxt);
    else if (hn) launch_lean3<DepthT, SP, false, true>(a, s, sp_pxt);
    else launch_lean3<DepthT, SP, false, false>(a, s, sp_pxt);
}

// the fused refine instantiation's shape: the tile is the large single-pass tile (12288 pixels) whichever way it is cut into waves
#ifndef DD_REFINE_NW
#define DD_REFINE_NW DD_SP_WAVES
#endif
#ifndef DD_REFINE_PXT
#define DD_REFINE_PXT DD_L_PXT
#endif
constexpr int RF_NW = DD_REFINE_NW, RF_PXT = DD_REFINE_PXT;
static_assert(RF_NW * RF_PXT == SP_WAVES * L_PXT && RF_NW <= 16, "the fused refine stage keeps the large tile's size (make_plan counts tiles by it)");

void launch_refine(const KArgs &a, hipStream_t s) {
    const dim3 grid(a.num_tiles + (a.scan_service ? 1u : 0u)), block(64 * RF_NW);
    const bool hn = a.out_normal != nullptr, hc = a.rgb && (a.out_rgb || a.out_packed);
    if (hn && hc) hipLaunchKernelGGL((compact_lean<float, false, true, true, true, RF_NW, true, RF_PXT>), grid, block, 0, s, a);
    else if (hn) hipLaunchKernelGGL((compact_lean<float, false, true, true, false, RF_NW, true, RF_PXT>), grid, block, 0, s, a);
    else if (hc) hipLaunchKernelGGL((compact_lean<float, false, true, false, true, RF_NW, true, RF_PXT>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((compact_lean<float, false, true, false, false, RF_NW, true, RF_PXT>), grid, block, 0, s, a);
}

template <bool SP>
void launch_scatter(const Plan &p, const KArgs &a, hipStream_t s) {
    if (p.refine) {
        launch_refine(a, s);
        return;
    }
    if (p.lean) {
        if (p.f16) launch_lean<_Float16, SP>(a, s, p.sp_pxt); else launch_lean<float, SP>(a, s, p.sp_pxt);
    } else {
        const dim3 grid(a.num_tiles), block(BLOCK);
        if (p.f16) hipLaunchKernelGGL((compact_generic<_Float16, SP>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((compact_generic<float, SP>), grid, block, 0, s, a);
    }
}

void launch_count(const Plan &p, const KArgs &a, hipStream_t s) {
    const dim3 grid(a.num_tiles), block(BLOCK);
    const bool hm = a.flags & DD_VALID_MASK;
    if (p.lean) {
        const bool nd = (a.flags & DD_VALID_DEPTH_POSITIVE) != 0;   // mask(/conf)-only rules never read depth in pass 1
        if (hm && !nd) {
            if (p.f16) hipLaunchKernelGGL((count_lean<_Float16, true, false>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((count_lean<float, true, false>), grid, block, 0, s, a);
        } else if (p.f16) {
            if (hm) hipLaunchKernelGGL((count_lean<_Float16, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((count_lean<_Float16, false, true>), grid, block, 0, s, a);
        } else {
            if (hm) hipLaunchKernelGGL((count_lean<float, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((count_lean<float, false, true>), grid, block, 0, s, a);
        }
    } else {
        if (p.f16) hipLaunchKernelGGL((count_generic<_Float16>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((count_generic<float>), grid, block, 0, s, a);
    }
}

int check_launch(const char *what) {
    if (hipGetLastError() == hipSuccess) return DD_OK;
    snprintf(g_err, sizeof(g_err), "%s: kernel launch failed", what);
    return DD_ERR_LAUNCH;
}

// pass 1: tile counts -> per-view tile offsets + view totals -> absolute view offsets
int enqueue_plan(const Plan &p, const KArgs &a, hipStream_t s) {
    launch_count(p, a, s);
    const unsigned long long serial_steps = (unsigned long long)a.V * ((a.tiles_per_view + BLOCK - 1) / BLOCK);
    if (serial_steps <= 16) {
        hipLaunchKernelGGL(scan_small, dim3(1), dim3(BLOCK), 0, s, a);
    } else {
        hipLaunchKernelGGL(scan_view_tiles, dim3(a.V), dim3(BLOCK), 0, s, a);
        hipLaunchKernelGGL(scan_views, dim3(1), dim3(BLOCK), 0, s, a);
    }
    return check_launch("dd_plan");
}

}  // namespace

// (for the library's other translation units: dd_refine_apply reads its two switches here)
__attribute__((visibility("hidden"))) uint32_t dd_lab_word() { return g_lab; }

extern "C" {

int dd_abi_version(void) { return DD_ABI_VERSION; }

uint32_t dd_debug_tuning(uint32_t bits) {
    const uint32_t before = g_lab;
    g_lab = bits;
    return before;
}

int32_t dd_chain_workgroup_limit(void) { return (int32_t)chain_wg_limit(); }

int dd_debug_plan(const DDViewBatch *batch, int32_t out[8]) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if (!out) return fail(DD_ERR_INVALID_ARG, "out is NULL");
    out[0] = p.lean; out[1] = p.single; out[2] = p.refine; out[3] = p.f16; out[4] = p.tile; out[5] = p.sp_pxt; out[6] = p.chain_gate;
    out[7] = (int32_t)a.num_tiles;
    return DD_OK;
}

const char *dd_last_error(void) { return g_err; }

int dd_count_valid(const DDViewBatch *batch, int64_t *counts_dev, void *stream) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if (!counts_dev) return fail(DD_ERR_INVALID_ARG, "counts_dev is NULL");
    if (p.refine) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE is handled by dd_unproject_compact only");
    hipStream_t s = (hipStream_t)stream;
    a.counts = reinterpret_cast<unsigned long long *>(counts_dev);
    if (hipMemsetAsync(counts_dev, 0, sizeof(int64_t) * (size_t)a.V, s) != hipSuccess)
        return fail(DD_ERR_LAUNCH, "hipMemsetAsync(counts) failed");
    launch_count(p, a, s);
    return check_launch("dd_count_valid");
}

int64_t dd_workspace_bytes(const DDViewBatch *batch) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    return ws_bytes(a);
}

int dd_plan(const DDViewBatch *batch, const int64_t *cursor_dev, int64_t *view_offsets_dev,
            void *workspace, int64_t workspace_bytes, void *stream) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if (!view_offsets_dev || !cursor_dev) return fail(DD_ERR_INVALID_ARG, "view_offsets_dev / cursor_dev is NULL");
    if (p.refine) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE is handled by dd_unproject_compact only");
    if ((rc = bind_workspace(a, workspace, workspace_bytes)) != DD_OK) return rc;
    a.view_offsets = reinterpret_cast<long long *>(view_offsets_dev);
    a.cursor = reinterpret_cast<const long long *>(cursor_dev);
    return enqueue_plan(p, a, (hipStream_t)stream);
}

int dd_scatter(const DDViewBatch *batch, const DDCloudOut *out, const int64_t *view_offsets_dev,
               void *workspace, int64_t workspace_bytes, void *stream) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if ((rc = bind_output(a, batch, out)) != DD_OK) return rc;
    if (!view_offsets_dev) return fail(DD_ERR_INVALID_ARG, "view_offsets_dev is NULL");
    if (p.refine) return fail(DD_ERR_UNSUPPORTED, "DD_REFINE is handled by dd_unproject_compact only");
    if ((rc = bind_workspace(a, workspace, workspace_bytes)) != DD_OK) return rc;
    a.view_offsets = const_cast<long long *>(reinterpret_cast<const long long *>(view_offsets_dev));
    launch_scatter<false>(p, a, (hipStream_t)stream);
    return check_launch("dd_scatter");
}

int dd_stream_fork(void *event, void *from_stream, void *to_stream) {
    if (!event) return fail(DD_ERR_INVALID_ARG, "event is NULL");
    // A stream with nothing pending has nothing to wait for: the fork then costs one query on the host and no barrier packet in
    // either queue (a stream that is being captured cannot be queried; there the fork is recorded as it is).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusActive;
    if (hipStreamIsCapturing((hipStream_t)from_stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
    if (cap == hipStreamCaptureStatusNone) {
        if (hipStreamQuery((hipStream_t)from_stream) == hipSuccess) return DD_OK;
        (void)hipGetLastError();                       // (hipErrorNotReady must not stay behind as the thread's last error)
    }
    if (hipEventRecord((hipEvent_t)event, (hipStream_t)from_stream) != hipSuccess) return fail(DD_ERR_LAUNCH, "hipEventRecord failed");
    if (hipStreamWaitEvent((hipStream_t)to_stream, (hipEvent_t)event, 0) != hipSuccess) return fail(DD_ERR_LAUNCH, "hipStreamWaitEvent failed");
    return DD_OK;
}

int dd_stream_wait(void *stream, void *event) {
    if (!event) return fail(DD_ERR_INVALID_ARG, "event is NULL");
    if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) != hipSuccess) { (void)hipGetLastError(); return fail(DD_ERR_LAUNCH, "hipStreamWaitEvent failed"); }
    return DD_OK;
}

int dd_streams_overlap(void *stream_a, void *stream_b, int32_t *scratch_dev, int32_t *overlap_out) {
    if (!scratch_dev || !overlap_out) return fail(DD_ERR_INVALID_ARG, "scratch_dev / overlap_out is NULL");
    if (stream_a == stream_b) { *overlap_out = 0; return DD_OK; }
    hipStream_t sa = (hipStream_t)stream_a, sb = (hipStream_t)stream_b;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    for (hipStream_t s : {sa, sb})
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return fail(DD_ERR_INVALID_ARG, "dd_streams_overlap synchronises both streams: not inside a stream capture");
        }
    int *w = reinterpret_cast<int *>(scratch_dev);
    if (hipMemsetAsync(w, 0, 8, sa) != hipSuccess || hipStreamSynchronize(sa) != hipSuccess || hipStreamSynchronize(sb) != hipSuccess)
        return fail(DD_ERR_LAUNCH, "dd_streams_overlap: clearing the scratch words failed");
    overlap_wait<<<1, 64, 0, sa>>>(w);
    overlap_set<<<1, 64, 0, sb>>>(w);
    int32_t seen = 0;
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sb) != hipSuccess
        || hipMemcpyAsync(&seen, w + 1, 4, hipMemcpyDeviceToHost, sa) != hipSuccess || hipStreamSynchronize(sa) != hipSuccess)
        return fail(DD_ERR_LAUNCH, "dd_streams_overlap: the probe kernels failed");
    *overlap_out = seen;
    return DD_OK;
}

int dd_unproject_compact(const DDViewBatch *batch, const DDCloudOut *out, int64_t *view_offsets_dev,
                         int64_t *cursor_dev, void *workspace, int64_t workspace_bytes, void *stream) {
    KArgs a; Plan p;
    int rc = make_plan(batch, a, p);
    if (rc != DD_OK) return rc;
    if ((rc = bind_output(a, batch, out)) != DD_OK) return rc;
    if (!view_offsets_dev || !cursor_dev) return fail(DD_ERR_INVALID_ARG, "view_offsets_dev / cursor_dev is NULL");
    if ((rc = bind_workspace(a, workspace, workspace_bytes)) != DD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    a.view_offsets = reinterpret_cast<long long *>(view_offsets_dev);
    a.cursor = reinterpret_cast<const long long *>(cursor_dev);

    a.cursor_out = reinterpret_cast<long long *>(cursor_dev);   // the kernels advance the cursor themselves
    if (p.single) {
        if (p.lean) {   // the single-pass lean kernel works on tiles of 12 waves x 16 pixels per lane (12288), or 16 x 8 (8192) for a small batch
            const unsigned tile_px = p.sp_pxt != SP_PXT_SMALL ? SP_WAVES * 64 * L_PXT
                                   : (a.chain && !a.chain_gated) ? SP_WAVES * 64 * SP_PXT_SMALL : SP_WAVES_SMALL * 64 * SP_PXT_SMALL;
            a.tiles_per_view = (a.P + tile_px - 1) / tile_px;
            a.num_tiles = a.tiles_per_view * (unsigned)a.V;
        }
        // ONE stream operation: the granules are tagged with the workspace's call epoch (nothing is zeroed), and the last tile
        // to finish its look-back writes the cursor and closes the call (close_call)
        if (p.chain_gate && a.chain) hipLaunchKernelGGL(chain_gate, dim3(1), dim3(64), 0, s, a.chain, a.chain_seq);
        launch_scatter<true>(p, a, s);
        return check_launch("dd_unproject_compact");
    }
    if ((batch->tuning & TUNE_ASSUME_DENSE) && p.lean && !p.refine) {
        // speculation for inputs that are expected to be dense (a depth map without holes and no mask): no counting pass.
        // The plan is arithmetic, the scatter pass -- which reads the validity inputs anyway -- verifies every tile.
        a.assume_dense = 1;
        hipLaunchKernelGGL(plan_dense, dim3((a.num_tiles + 255u) / 256u), dim3(256), 0, s, a, (unsigned)p.tile);
    } else if ((rc = enqueue_plan(p, a, s)) != DD_OK) return rc;
    launch_scatter<false>(p, a, s);
    return check_launch("dd_unproject_compact");
}

}  // extern "C"
