/*
 * ddcore.h -- C ABI of libddcore.so, the MI355X (gfx950) densification core.
 *
 * The reference (OpsiClear/DepthDensifier) has no FFI / operator boundary: its
 * depth->points path is inline NumPy in scripts/test.py:203-244 (+ :262-266) and
 * a second formulation in src/depthdensifier/visualizer.py:246-376.  This header
 * is the boundary a maintainer would bind instead of that inline block (see
 * INTEGRATION.md for the ctypes stub).  Plain pointers and sizes only; no torch
 * or HIP types.  Conventions:
 *
 *  - every function returns DD_OK (0) or a negative DD_ERR_* code and never
 *    throws; dd_last_error() returns a thread-local message for the last error;
 *  - the caller owns every buffer (device memory unless stated); the compute entry
 *    points allocate nothing and keep no global mutable state; calls are re-entrant
 *    (the one allocator is the explicit DDArena object at the end of this header; the
 *    one thing remembered is a property of the device, dd_chain_workgroup_limit);
 *  - the switches of the library's own A/B experiments and fault-injection tests are
 *    NOT here: include/ddcore_lab.h (a thread-local debug word, zero in production);
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream).  Nothing synchronises the host.
 */
#ifndef DDCORE_H
#define DDCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_ABI_VERSION 14

enum {
    DD_OK = 0,
    DD_ERR_INVALID_ARG = -1,  /* NULL / out-of-range field, see dd_last_error() */
    DD_ERR_LAUNCH = -2,       /* HIP reported an error at enqueue time */
    DD_ERR_WORKSPACE = -3,    /* workspace too small or mis-aligned */
    DD_ERR_UNSUPPORTED = -4
};

/* element type of a depth / confidence map */
enum { DD_F32 = 0, DD_F16 = 1 };

/* DDViewBatch.flags */
#define DD_VALID_DEPTH_POSITIVE 0x1u /* depth > 0            scripts/test.py:210, visualizer.py:314 */
#define DD_VALID_MASK           0x2u /* mask  != 0           scripts/test.py:194, visualizer.py:312 */
#define DD_VALID_CONF           0x4u /* conf  > threshold    (build-defined)          */
#define DD_ROTATE_NORMALS       0x8u /* n_w = R^T n / (|R^T n| + 1e-8)   visualizer.py:363-374      */
#define DD_REFINE               0x10u /* depth is the RAW map; refine it per DDViewParams.knots_* inside the densify kernel
                                        (src/depthdensifier/depth_refiner.py:180-205 fused with scripts/test.py:194-233).
                                        dd_unproject_compact only, stride 1, width <= 3071, no DD_VALID_CONF; the validity
                                        rule is the script's: mask (or raw depth > 0 without one) AND refined depth > 0 */

/* DDViewBatch.tuning: what a caller may choose (0 = the library decides).  The rows written are the same for every value. */
#define DD_TUNE_GENERIC      0x1u        /* the scalar any-stride kernels even on stride-1 maps */
#define DD_TUNE_TWO_PASS     0x4u        /* dd_unproject_compact as count + scans + scatter: no workgroup waits for another */
#define DD_TUNE_SINGLE_PASS  0x8u        /* dd_unproject_compact as ONE kernel with an in-kernel scan (the default on stride-1 maps) */
#define DD_TUNE_DENSE_TILES  0x80u       /* tiles whose pixels all survive skip the point list (line-aligned 16-byte stores; not with
                                            xyz_rgba or rotated normals) */
#define DD_TUNE_INTERLEAVE(k) ((((uint32_t)(k) - 1u) & 63u) << 8)   /* two-pass scatter: tiles of k = 2..64 stretches of the batch in turn, so
                                            that consecutive workgroups write k distant regions (a DD_ARENA_BLOCKED array: 0.80 vs 0.66) */
#define DD_TUNE_ASSUME_DENSE (1u << 17)  /* stride-1 maps without holes: no counting pass; the scatter verifies every tile and a miss sets
                                            the workspace's error word to 2 -- rows, offsets and cursor are then void: redo with TWO_PASS */
#define DD_TUNE_TILE_SMALL   (1u << 18)  /* single pass: 8192-pixel tiles (default up to 3072 tiles of 12288: a streamed view or a dozen) */
#define DD_TUNE_TILE_LARGE   (3u << 18)  /* single pass: 12288-pixel tiles (default above) */
#define DD_TUNE_BY_INDEX     (1u << 22)  /* single pass: tiles by workgroup index instead of by ticket -- 2-8 % faster, only on a GPU this
                                            stream has to itself (two such launches sharing a GPU can hold each other's slots) */
#define DD_TUNE_ALL (DD_TUNE_GENERIC | DD_TUNE_TWO_PASS | DD_TUNE_SINGLE_PASS | DD_TUNE_DENSE_TILES | (63u << 8) | DD_TUNE_ASSUME_DENSE | DD_TUNE_TILE_LARGE | DD_TUNE_BY_INDEX)

/*
 * Per-view camera block, 32 floats (128 B), device memory, built on the host in
 * float64 and rounded once.  With cam_from_world = [R|t] and intrinsics K:
 *   ray_to_world = R^T * K^-1   (row-major 3x3)
 *   centre       = -R^T t
 *   rot          = R^T          (row-major 3x3, used only with DD_ROTATE_NORMALS)
 * so that  p_world = depth * (ray_to_world * [u, v, 1]^T) + centre, which is
 * scripts/test.py:79-90 followed by :233 (and visualizer.py:320-334) fused.
 */
typedef struct DDViewParams {
    float ray_to_world[9];
    float centre[3];
    float rot[9];
    /* Optional depth->depth transfer curve of this view (DD_REFINE, see DDViewBatch.flags): the kernel then reads the RAW
     * monocular depth and refines it on the fly exactly like dd_refine_apply -- sorted look-up table with linear
     * interpolation (depth_refiner.py:141-178), 3x3 median (:194-200), zero outside the mask (:203). */
    int32_t n_knots;          /* 2..512 */
    int32_t skip_smoothing;
    int32_t reserved0;
    const float *knots_x;     /* (n_knots) device, ascending */
    const float *knots_y;     /* (n_knots) device */
    float reserved[4];
} DDViewParams;

/*
 * A stack of V equally sized views, each map contiguous row-major, view v at
 * element offset v*H*W (x3 for normal / rgb).  Replaces the per-image locals of
 * scripts/test.py:166-168 (depth, normal, mask), :215 (rgb) and :172-178 (camera).
 */
typedef struct DDViewBatch {
    int32_t num_views;
    int32_t height;
    int32_t width;
    int32_t stride;           /* downsample_density, scripts/test.py:37,206; >= 1 */
    const void *depth;        /* (V,H,W) DD_F32 or DD_F16 */
    const uint8_t *mask;      /* (V,H,W) bytes, non-zero = keep; required iff DD_VALID_MASK */
    const void *conf;         /* (V,H,W) DD_F32 or DD_F16;      required iff DD_VALID_CONF */
    const float *normal;      /* (V,H,W,3) camera-frame unit normals, or NULL */
    const uint8_t *rgb;       /* (V,H,W,3) or NULL */
    const DDViewParams *params; /* (V) */
    int32_t depth_dtype;
    int32_t conf_dtype;
    float conf_threshold;
    uint32_t flags;
    int32_t view_index_base;  /* added to the per-point view index (global view id of view 0) */
    uint32_t tuning;          /* 0 = the library chooses; else an OR of DD_TUNE_* above -- what a caller may decide.  Every
                                 choice writes the same rows; reserved bits must be zero (DD_ERR_INVALID_ARG) */
    float *refined_out;       /* DD_REFINE: NULL, or (V,H,W) float32 receiving the refined, mask-zeroed depth -- the map the
                                 reference caches for the multi-view filter (scripts/test.py:194, 197-201) */
    int64_t *chain;           /* ABI 12: NULL, or (1) int64 device word that chains consecutive calls of ONE cloud across TWO streams, so
                                 that call n + 1 runs beside the tail of call n instead of behind it (a streaming caller's small calls --
                                 scripts/test.py:131: one view per iteration -- are bound by the launch-to-launch latency of a stream, not by
                                 their kernels).  The word holds [63:44] a sequence number, [43:0] the row the next call starts from; the
                                 caller initialises it to (0, first row).  A call with chain != NULL takes its first row from the word --
                                 its scan workgroup polls until the sequence equals chain_seq (mod 2^20) -- not from *cursor_dev, and when
                                 its scan is over stores (chain_seq + 1, row after the batch) there, and the row in *cursor_dev as usual.
                                 Calls that may be in flight together need a workspace and a view_offsets array each; two in flight at
                                 most, on a GPU the process has to itself.  A call of fewer than dd_chain_workgroup_limit() tiles of 6144 pixels
                                 (383 on a whole MI355X: a 1080p view) starts at once and waits inside its scan workgroup -- its tiles load
                                 and count meanwhile, and could never occupy every slot; a larger one is preceded by a one-wave gate kernel on its stream that returns
                                 when the earlier call's scan is over, so that its workgroups never hold a slot while they wait for
                                 another launch.  Stride-1 maps and the scan service only (DD_ERR_UNSUPPORTED otherwise) */
    int64_t chain_seq;
} DDViewBatch;

/*
 * The fused cloud (scripts/test.py:264-266 final_point_cloud / final_colors /
 * final_normals; visualizer.py:71-80 PointCloud), array-of-structs rows exactly
 * like the reference's (N,3) arrays.  NULL members are not produced.  Points
 * whose slot is >= capacity are counted but not written.
 */
typedef struct DDCloudOut {
    float *xyz;           /* (capacity,3) */
    float *normal;        /* (capacity,3); needs DDViewBatch.normal */
    uint8_t *rgb;         /* (capacity,3); needs DDViewBatch.rgb */
    int32_t *pixel_index; /* (capacity)  y*W + x inside the view */
    int32_t *view_index;  /* (capacity)  view_index_base + v */
    int64_t capacity;
    uint32_t *xyz_rgba;   /* (capacity,4) dwords, 16-B aligned, or NULL: x, y, z as float32 bits and
                             r | g<<8 | b<<16 | 255<<24 (0 colour without DDViewBatch.rgb) -- the compact 16-byte
                             record of a point for the multi-GPU gather (SURVEY.md 8e ii: what the model writer
                             of scripts/test.py:355-358 needs).  xyz may be NULL when this is given. */
} DDCloudOut;

int dd_abi_version(void);
const char *dd_last_error(void);

/* Per-view number of valid visited pixels (the N of scripts/test.py:210-212).
 * counts_dev: (V) int64, overwritten. */
int dd_count_valid(const DDViewBatch *batch, int64_t *counts_dev, void *stream);

/* Bytes of device scratch the calls below need for this batch (>= 0), or a negative error. */
int64_t dd_workspace_bytes(const DDViewBatch *batch);

/*
 * Pass 1 of the hot path: count the valid visited pixels of every tile (one streaming read of
 * depth / mask / conf) and scan the counts, so that every view -- and, in the workspace, every
 * tile -- knows the row of its first point.  Lets the caller size the cloud exactly
 * (view_offsets_dev[V] - *cursor_dev points) before anything is written.
 *
 *  cursor_dev       (1) int64, device, read: number of points already in the cloud.
 *  view_offsets_dev (V+1) int64, device, out: row of the first point of each view;
 *                   [V] = row after the batch.
 *  workspace        dd_workspace_bytes() bytes, 16-B aligned; must be handed unchanged to
 *                   dd_scatter for the same batch.
 */
int dd_plan(const DDViewBatch *batch, const int64_t *cursor_dev, int64_t *view_offsets_dev,
            void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Pass 2: cull, unproject (scripts/test.py:79-90), camera-to-world (:233), gather colour/normal
 * (:215-220) and write every surviving pixel at its final row -- stable order: views in batch
 * order, row-major inside a view (:205-212), i.e. exactly the order of the reference's
 * np.concatenate (:264-266).  No workgroup depends on another one.
 */
int dd_scatter(const DDViewBatch *batch, const DDCloudOut *out, const int64_t *view_offsets_dev,
               void *workspace, int64_t workspace_bytes, void *stream);

/*
 * The whole hot path for one batch, appended to the cloud, enqueued on `stream` without a host
 * round trip, so chaining calls fuses any number of batches (scripts/test.py:238-240 list append +
 * :264-266 concatenate).  On stride-1 maps (any view size) this is ONE pass over the inputs (ticket +
 * decoupled look-back inside the kernel); otherwise dd_plan + dd_scatter.  Same rows either way.
 *
 *  cursor_dev       (1) int64, device, in/out: advanced by the batch's number of points.
 *  view_offsets_dev (V+1) int64, device, out.
 *  workspace        as above.  ((int32_t*)workspace)[1] != 0 after the stream has drained means the
 *                   in-kernel look-back gave up after ~2 s of polling (1: should never happen) or a batch
 *                   run with tuning bit 17 was not dense (2); the rows, offsets and the cursor of THAT batch
 *                   are then invalid (rows of earlier batches are untouched: a tile that does not know its
 *                   place writes nothing) -- zero the whole workspace and redo the batch with tuning = 4.
 *                   The first 16 bytes of the workspace are STICKY: the library never zeroes them, so a set
 *                   error word survives any number of later calls on the same workspace until the caller
 *                   clears it.
 *
 * ABI 11: the single-pass call is ONE stream operation (one kernel launch; up to ABI 10 a memset of the look-back
 * granules went in front of it and a copy of the cursor behind it).  The caller zeroes the workspace ONCE, before its
 * first use; from then on the workspace carries a call epoch that tags the granules (a granule of an earlier call reads as
 * "not published"), and the batch's last tile writes the cursor and advances the epoch.  One workspace serves one stream
 * at a time, calls of any kind and size in any order (dd_plan + dd_scatter pairs uninterrupted).
 */
int dd_unproject_compact(const DDViewBatch *batch, const DDCloudOut *out,
                         int64_t *view_offsets_dev, int64_t *cursor_dev,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* ABI 12, for calls chained across two streams (DDViewBatch.chain): everything enqueued on `to_stream` from here on runs behind
 * everything enqueued on `from_stream` so far -- hipEventRecord(event, from_stream) + hipStreamWaitEvent(to_stream, event) in one
 * call (`event`: a hipEvent_t of the caller, as void*).  The maps of a chained call are produced on the caller's stream and read
 * on a side stream; when the cloud is read, the caller's stream is ordered behind the side streams the same way.  When
 * `from_stream` has nothing pending (hipStreamQuery == hipSuccess, outside a stream capture) there is nothing to order and
 * nothing is enqueued: a chain of small calls on resident maps then carries no barrier packet at all. */
int dd_stream_fork(void *event, void *from_stream, void *to_stream);

/* `stream` waits (on the device) for `event`, a hipEvent_t recorded elsewhere -- e.g. by dd_upload_async on a copy stream. */
int dd_stream_wait(void *stream, void *event);
/* ABI 13: do kernels of `stream_a` and `stream_b` run side by side?  The HIP runtime deals streams to a few hardware queues, and two
 * streams that share one run strictly in order -- calls chained across them (DDViewBatch.chain) would then be SLOWER than on one
 * stream.  A one-thread kernel on `stream_a` waits (at most 1 ms) for a flag that a kernel launched behind it on `stream_b` raises;
 * *overlap_out = 1 if it saw the flag.  `scratch_dev`: 8 bytes of device memory.  Synchronises both streams (about 50 us; call it
 * once per pair of side streams, outside any stream capture) and no other. */
int dd_streams_overlap(void *stream_a, void *stream_b, int32_t *scratch_dev, int32_t *overlap_out);

/* Workgroups (tiles + the scan) up to which a chained call (DDViewBatch.chain) runs WITHOUT a gate kernel on the current device: three
 * quarters of the workgroup slots the device offers that kernel (occupancy x compute units: 384 on a whole MI355X, 48 on a CPX
 * partition), so that a call waiting for its predecessor's scan can never hold every slot.  0 = unknown (every chained call is gated). */
int32_t dd_chain_workgroup_limit(void);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU fuse: what scripts/test.py:262-266 (np.concatenate of the per-view arrays)
 * becomes when the views are sharded over the GPUs of a node, one process per GPU.
 *
 * In-place all-gatherv of the per-GPU compacted clouds over an RCCL communicator the CALLER owns.  Rank r has
 * already written its rows [rank_rows[r], rank_rows[r+1]) of every non-NULL field of `cloud` -- the GLOBAL buffers,
 * capacity >= rank_rows[world_size] -- e.g. by dd_unproject_compact with *cursor_dev = rank_rows[r].  The call posts
 * ONE ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd on `stream`: its rows to every peer, every peer's rows
 * straight into place (xGMI is point-to-point: one link per peer, all driven at once; nothing is staged or copied
 * locally).  Chain it per chunk of views on a second stream to overlap the exchange with the next chunk's kernel.
 *
 *  comm        ncclComm_t of the caller (as void*); unused when world_size == 1.
 *  rank_rows   (world_size+1) int64 on the HOST: first global row of each rank's slice, [world_size] = total.
 *  dst         -1: every rank receives (replicated cloud); r: gather-to-owner, only rank r receives and a pure
 *              sender's buffers hold just its own rows (their row 0 is global row rank_rows[rank]).
 * libddcore.so does not link RCCL: the entry points are resolved at first use from the RCCL runtime already loaded in
 * the process.  Returns DD_ERR_UNSUPPORTED when none can be found, DD_ERR_LAUNCH on an RCCL error
 * (dd_comm_last_error() has the RCCL message).
 * ------------------------------------------------------------------------------------------- */
int dd_allgatherv(void *comm, int32_t rank, int32_t world_size, const DDCloudOut *cloud, const int64_t *rank_rows,
                  int32_t dst, void *stream);
const char *dd_comm_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * multi-view floater votes, scripts/test.py:269-335 (+ project_points :58-76).
 * ------------------------------------------------------------------------------------------- */

/* The cached views of scripts/test.py:197-201 as device stacks. */
typedef struct DDFilterViews {
    int32_t num_views;
    int32_t height;
    int32_t width;
    int32_t reserved;       /* 0 */
    const float *depth;     /* (V,H,W) refined depth */
    const uint8_t *mask;    /* (V,H,W) or NULL; mask == 0 reads as depth 0 (scripts/test.py:194) */
    const double *cams;     /* (V,24) float64 per view: cam_from_world 3x4 row-major [0..11]
                               (image.cam_from_world().matrix(), :63), calibration rows 0 and 1 [12..17]
                               (camera.calibration_matrix(), :73), projection centre -R^T t [18..20] (:284) */
    double grazing_cos;     /* 0.087, scripts/test.py:295 */
    float depth_threshold;  /* FilteringConfig.depth_threshold = 0.7, scripts/test.py:45-46, 320 */
    float reserved2;
    void *workspace;        /* device scratch, 32-byte aligned, at least 256 * num_views bytes (ABI 10; the table-free kernel that a
                               NULL workspace selected up to ABI 9 was removed): the float64 kernel builds a per-view table there
                               (K [R|t] and a band coefficient) and tests the image bounds without the division.
                               >= 512 * num_views + 64 bytes: per-workgroup view culling becomes
                               possible (a view is skipped for 256 consecutive points when their bounding sphere cannot touch
                               its frustum -- conservative, same votes); with ceil(n / 65536) * ceil(num_views / 64) * 8 bytes
                               more (dd_votes_workspace_bytes() covers it) the cull is two-level: a mask of visible views per
                               65 536 consecutive points first */
    int64_t workspace_bytes;
    int32_t mode;           /* 0 = the best the workspace allows (4, else 1); 1 = float64 with
                               the table, no culling; 3 = culling always; 4 = 1 or 3, chosen on the device from a sample of
                               the workgroups (culling when it removes more than 10 % of the workgroup x view cells);
                               2 = removed (the float32 first pass of ABI 8) */
    int32_t reserved3;
} DDFilterViews;

/* votes_dev[i] (= or +=, by `accumulate`) the number of views in which point i is a floater
 * (scripts/test.py:273-328).  xyz / normal: (n,3) float32 rows of the fused cloud; decisions are taken
 * in float64 exactly as NumPy promotes them.  The caller then keeps votes < vote_threshold (:330). */
int dd_floater_votes(const DDFilterViews *views, const float *xyz, const float *normal, int64_t n,
                     int32_t *votes_dev, int32_t accumulate, void *stream);
int64_t dd_votes_workspace_bytes(int32_t num_views, int64_t n_points);
const char *dd_filter_last_error(void);

/* Stable compaction of every per-point field by the vote test of scripts/test.py:330-332
 * (keep = votes < vote_threshold).  in / out: the field pointers of DDCloudOut (NULL output fields are
 * skipped; kept rows at or beyond out->capacity are counted, not written; out->xyz_rgba packs xyz + colour
 * into the 16-byte gather record); kept_dev: (1) int64 out; old/new_view_offsets_dev: (V+1) int64 each or both
 * NULL -- the new offsets are the kept rows before each old offset.  workspace:
 * dd_compact_workspace_bytes(n) bytes, 16-B aligned. */
int64_t dd_compact_workspace_bytes(int64_t n);
int dd_compact_cloud(const DDCloudOut *in, int64_t n, const int32_t *votes_dev, int32_t vote_threshold,
                     const DDCloudOut *out, int64_t *kept_dev, const int64_t *old_view_offsets_dev,
                     int64_t *new_view_offsets_dev, int32_t num_views, void *workspace, int64_t workspace_bytes,
                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * getting a view's precomputed maps into HBM (the stand-in for scripts/test.py:143-168 where MoGe is not
 * installed) without the interpreter: ctypes releases its lock during a foreign call, so prefetch threads that call these do not
 * make the main thread queue for it.  Host functions; dd_ingest_last_error() has the message of a negative return.
 * ------------------------------------------------------------------------------------------- */
enum { DD_NPY_F32 = 0, DD_NPY_F16 = 1, DD_NPY_U8 = 2, DD_NPY_BOOL = 3 };
/* Header of a .npy file (format versions 1-3, C order, up to 4 dimensions, one of the four element types above). */
int dd_npy_header(const char *path, int32_t *dtype_out, int32_t *ndim_out, int64_t *shape_out /* [4] */, int64_t *data_offset_out);
/* Reads the array of a .npy file into dst (e.g. a page-locked staging buffer) after checking element type and shape against what
 * the caller expects (U8 and BOOL are interchangeable): open / pread / close, no allocation, no temporary. */
int dd_npy_read(const char *path, int32_t expect_dtype, int32_t expect_ndim, const int64_t *expect_shape, void *dst, int64_t dst_bytes);
/* n host -> device copies on `stream` and, behind them, `event` (a hipEvent_t or NULL) -- one call instead of n + 1. */
int dd_upload_async(int32_t n, const void *const *src_host, void *const *dst_dev, const int64_t *nbytes, void *event, void *stream);
const char *dd_ingest_last_error(void);
/* The prefetcher: `threads` native threads read the .npy files of views to come into `slots` page-locked staging slots of slot_bytes
 * each (taken lazily).  dd_prefetch_submit queues a job of n <= 8 files -- file i, expected as in dd_npy_read (expect_dtype < 0: any of
 * the four types), goes to byte offset offsets[i] of the job's slot -- and returns its ticket (0, 1, 2 ...; job t uses slot t mod
 * slots).  dd_prefetch_wait blocks until the job is done and returns its status, the slot's address and the element types found;
 * dd_prefetch_release gives the slot back: it is refilled once `event` (a hipEvent_t the caller recorded behind its last read of the
 * slot, e.g. through dd_upload_async; NULL = at once) has passed.  Jobs are released by their owner, at most `slots` in flight.
 * dd_prefetch_destroy joins the threads and frees the slots: no call on `p` may be in flight or follow. */
typedef struct DDPrefetcher DDPrefetcher;
int dd_prefetch_create(int32_t threads, int32_t slots, int64_t slot_bytes, DDPrefetcher **out);
int64_t dd_prefetch_submit(DDPrefetcher *p, int32_t n, const char *const *paths, const int32_t *expect_dtype, const int32_t *ndim,
                           const int64_t *shapes /* n x 4 */, const int64_t *offsets);
int dd_prefetch_wait(DDPrefetcher *p, int64_t ticket, void **base_out, int32_t *dtypes_out);
int dd_prefetch_release(DDPrefetcher *p, int64_t ticket, void *event);
int dd_prefetch_destroy(DDPrefetcher *p);

/* ---------------------------------------------------------------------------------------------
 * the per-pixel half of DepthRefiner as one kernel --
 * src/depthdensifier/depth_refiner.py:180-205 (_apply_transformation) with :141-178
 * (_pchip_interpolate_optimized): refined = mask ? median3x3( mask ? LUT(depth) : 0 ) : 0, where LUT is
 * the piecewise-linear curve through the sorted knots (knots_x ascending, >= 2 of them), clamped at the
 * end knots and floored at 1e-3.  mask == NULL means depth > 0 (:241).  float32 arithmetic.
 * ------------------------------------------------------------------------------------------- */
int dd_refine_apply(const void *depth, int32_t depth_dtype, const uint8_t *mask, int32_t height, int32_t width,
                    const float *knots_x, const float *knots_y, int32_t n_knots, int32_t skip_smoothing,
                    float *refined_out, void *stream);
const char *dd_refine_last_error(void);

/* The correspondence half of DepthRefiner.refine_depth, src/depthdensifier/depth_refiner.py:244-299, in ONE launch:
 * project the n sparse points (:92-115; cam_from_world = 12 floats [R|t] row-major, calibration = rows 0 and 1 of K, 6
 * floats, both on the HOST), keep projections inside the image minus edge_margin with positive depth (:247-254), sample
 * the depth map bilinearly there (grid_sample, zero padding, align_corners=True, :260-272), keep positive samples
 * (:281-288), and -- if robust and more than 10 remain -- drop ratio outliers by outlier_threshold x IQR (:117-139).
 * z_mono_out / z_metric_out: (n) float32 device, the kept correspondences in the reference's order; scratch: (n) float32
 * device; meta_out: (8) int32 device: [0] projections in bounds, [1] positive samples, [2] kept, [3] outliers removed,
 * [4] float32 bits of the effective scale median(z_metric / (z_mono + 1e-6)) (:315).  float32 arithmetic;
 * half_precision_io != 0 mirrors the reference's FP16 mode (:85-86): points and correspondences are quantised to half. */
int dd_refine_fit(const float *points, int32_t n, const float *cam_from_world, const float *calibration, const void *depth,
                  int32_t depth_dtype, int32_t height, int32_t width, int32_t edge_margin, int32_t robust, float outlier_threshold,
                  int32_t half_precision_io, float *z_mono_out, float *z_metric_out, float *scratch, int32_t *meta_out, void *stream);
/* dd_refine_fit for a streaming caller, everything of a view's fit enqueued by ONE call (round 6: the loop around the kernels is bound by
 * the number of calls the host makes): the n sparse points are copied up from points_host ((n,3) float32, page-locked for the copy to
 * be asynchronous), the fit runs as in dd_refine_fit with z_mono / z_metric / scratch inside `work` (device, 6 n floats: the points,
 * then the three arrays), the number of masked pixels (mask != 0, or depth > 0 if mask is NULL; clamped to 2^31 - 1) goes to
 * meta_dev[5], the eight result words are copied to meta_host (page-locked) and ready_event (a hipEvent_t or NULL) is recorded. */
int dd_refine_fit_async(const float *points_host, int32_t n, const float *cam_from_world, const float *calibration, const void *depth,
                        int32_t depth_dtype, int32_t height, int32_t width, int32_t edge_margin, int32_t robust, float outlier_threshold,
                        int32_t half_precision_io, const uint8_t *mask, float *work, int32_t *meta_dev, int32_t *meta_host,
                        void *ready_event, void *stream);
/* The knots sorted by x (torch.argsort, :149-151), n <= 4096, one launch. */
int dd_sort_knots(const float *x, const float *y, int32_t n, float *x_sorted, float *y_sorted, void *stream);

/* ---------------------------------------------------------------------------------------------
 * the dense points as points3D.bin records, formatted on the device (ABI 8) -- replaces, for the
 * dense cloud, the per-point loop rec.add_point3D(xyz, Track(), color) of scripts/test.py:355-358 and their
 * serialisation by rec.write_binary (:363).  Per point 51 bytes, little-endian, no padding (COLMAP's public layout):
 * point3D_id uint64 = first_id + i | xyz 3 x float64 (the float32 coordinate widened exactly) | rgb 3 x uint8 |
 * error float64 = -1.0 | track length uint64 = 0.  Input: xyz (n,3) float32 + rgb (n,3) uint8 (NULL = black), or the
 * 16-byte records xyz_rgba (n,4) of DDCloudOut (then xyz / rgb are ignored).  out: n * 51 bytes of device memory,
 * 16-byte aligned; the caller copies them behind the records of the sparse points (chunk by chunk: a chunk that
 * starts at point i0 passes first_id + i0).
 * ------------------------------------------------------------------------------------------- */
int dd_format_points3d(const float *xyz, const uint8_t *rgb, const uint32_t *xyz_rgba, int64_t n, uint64_t first_id,
                       uint8_t *out, void *stream);
const char *dd_model_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * HBM zone arena (ABI 9): where the large arrays of the fused cloud (scripts/test.py:264-266 final_point_cloud /
 * final_normals / final_colors) live in the 288 GB of an MI355X.  The physical memory falls into three classes of about a
 * third each; two row-store streams written in lock step (points + normals) run at 5.8 TB/s inside one class and at
 * 7.1 TB/s in two, which is 4-11 % of the densify kernel's time -- and a fresh process gets all its memory from one class
 * (profiles/r03_placement_*.txt, r03_zone_*.txt).  The arena takes physical chunks (1 GiB) through the
 * virtual-memory API, classifies each with a two-stream store probe against one anchor chunk per class, and maps every
 * requested array from chunks of the classes its layout names.  Host-side memory management: it changes addresses, never
 * results.  Not for buffers handed to RCCL (dd_allgatherv): memory from the virtual-memory API is not IPC-exportable.
 * ------------------------------------------------------------------------------------------- */
typedef struct DDArena DDArena;

/* layout of one array of dd_arena_alloc (the `groups` argument): 0, 1, 2 = class-pure, in the class of that GROUP (arrays
 * of one group share a class, different groups get different classes; a group keeps its class over later calls);
 * DD_ARENA_ROTATED + phase (phase 0..2) = the rotated arrays of ONE request never share a class at equal chunk index (two
 * arrays written in lock step then run at 7.1 TB/s, like two pure arrays of different groups); chunk k of an array prefers
 * class (phase + k) mod 3, and takes the class with the most free chunks otherwise -- so an even supply gives the exact
 * rotation and two plentiful classes give two class-pure arrays.  Phases 0 and 1 are the ones that must differ (the
 * lock-step store streams); phase 2 (the colours) may share a class when only two are at hand */
#define DD_ARENA_ROTATED 8
/* DD_ARENA_BLOCKED = the first, middle and last third of the array's chunks each from a class of its own: for ONE large row
 * array (a cloud of points only) whose scatter pass takes tiles of the three thirds in turn (DDViewBatch.tuning bits 8-11),
 * so that consecutive workgroups write three classes at once */
#define DD_ARENA_BLOCKED 16

typedef struct DDArenaStats {
    int64_t chunk_bytes;
    int64_t probe_bytes;        /* bytes each of the two probe streams writes */
    int32_t num_classes;        /* classes discovered so far (<= 3) */
    int32_t degraded_allocs;    /* allocations in which two lock-step arrays had to share a class somewhere */
    int64_t chunks_created;
    int64_t chunks_released;
    int64_t probes;
    int64_t chunks_held[3];     /* live chunks per class (anchors, allocations, pool) */
    int64_t chunks_pooled[3];   /* of those: spare classified chunks kept for the next allocation, unmapped or in cached arrays (see dd_arena_trim) */
    float same_class_ms;        /* probe level inside one chunk */
    float cross_class_ms;       /* fastest pair seen */
    double seconds;             /* time spent inside dd_arena_alloc */
} DDArenaStats;

/* An empty arena on `device`; chunk_bytes = 0 means 1 GiB (multiple of 2 MiB, >= 64 MiB). */
int dd_arena_create(int32_t device, int64_t chunk_bytes, DDArena **arena_out);
/* n arrays at once: sizes[i] bytes (rounded up to whole chunks), groups[i] = the layout code of array i (above).  Scouts
 * (creates + classifies) physical chunks until the layouts can be honoured, holding at most max_scout_bytes beyond the
 * request; chunks not needed go back to the driver (or the pool) before the call returns.  Synchronises the device (the
 * probes are timed).  Returns DD_OK, 1 = allocated but two lock-step arrays share a class somewhere (budget or memory too
 * small), or a negative error (DD_ERR_WORKSPACE = out of device memory; nothing of the request stays allocated). */
int dd_arena_alloc(DDArena *arena, int32_t n, const int64_t *sizes, const int32_t *groups, int64_t max_scout_bytes, void **ptrs_out);
/* Gives back one array of dd_arena_alloc; synchronises the device first.  An array of up to 8 chunks whose request was
 * served without conflicts stays MAPPED in a cache (12 chunks at most, oldest first out): the next request for arrays of
 * the same sizes in chunks and the same layouts gets them back as they are -- no virtual-memory call, no address space.
 * Otherwise the array is unmapped and its chunks go back to the driver, except that up to 4 chunks per class (dd_arena_trim
 * changes that) are kept as classified spares so that the next allocation need not scout again (scouting costs ~27 ms per
 * GiB looked at).  The arena never maps a virtual range twice (a re-mapped range keeps translating to the physical memory of
 * its first mapping on this stack, ddarena.hip): address space is consumed, not recycled, until dd_arena_destroy. */
int dd_arena_free(DDArena *arena, void *ptr);
/* Empties the cache and gives every spare chunk back to the driver; pool_chunks_per_class >= 0 also sets how many are kept from now on. */
int dd_arena_trim(DDArena *arena, int32_t pool_chunks_per_class);
/* Only sets how many spare chunks per class are kept from now on (ABI 14): nothing is released, the cache stays mapped.  For a process in
 * which released physical memory does not come back to the device (under rocprofv3): keep everything, give nothing away. */
int dd_arena_set_pool(DDArena *arena, int32_t pool_chunks_per_class);
/* Class of every chunk behind an array: returns the number of chunks, writes min(that, capacity) entries. */
int dd_arena_classes(DDArena *arena, const void *ptr, int32_t *classes_out, int32_t capacity);
/* The probe on two windows (probe_bytes each) chosen by the caller, milliseconds (diagnostics, tests). */
int dd_arena_probe(DDArena *arena, void *a, void *b, float *ms_out);
int dd_arena_stats(DDArena *arena, DDArenaStats *out);
int dd_arena_destroy(DDArena *arena);
const char *dd_arena_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DDCORE_H */
