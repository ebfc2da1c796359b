/*
 * ddcore_lab.h -- the EXPERIMENT switches of libddcore.so: A/B variants kept for the measurements in profiles/ and the fault
 * injection the recovery tests need.  Not part of the product ABI (include/ddcore.h): nothing a caller of the densification core
 * has a reason to set.  Used by tests/, tools/ab_builds.py, tools/bench_streaming.py and tools/bench_fused_refine.py only.
 *
 * The switches live in a THREAD-LOCAL debug word (zero in production): dd_debug_tuning(bits) sets it for the calls this thread
 * makes from then on and returns the previous word; every variant writes the same rows as the default.
 */
#ifndef DDCORE_LAB_H
#define DDCORE_LAB_H

#include <stdint.h>

struct DDViewBatch;

#ifdef __cplusplus
extern "C" {
#endif

#define DD_LAB_LIST_ORDER      0x1u   /* rows in list order: wave runs NOT shifted onto 128-byte lines */
#define DD_LAB_FAULT_INJECT    0x2u   /* an in-kernel scan behaves as if a tile had never published: wrong rows are never written, the
                                         workspace's error word is set (what a workgroup parked for ~2 s causes; tests of the redo) */
#define DD_LAB_POLL_LANES_32   0x4u   /* decoupled look-back: 32 polling lanes instead of 16 */
#define DD_LAB_POLL_LANES_64   0x8u   /* ... 64 */
#define DD_LAB_LOOKBACK        0x10u  /* single pass: the decoupled look-back of rounds 1-4 instead of the scan service */
#define DD_LAB_APPLY_PLAIN     0x80u  /* dd_refine_apply: the one-tile-per-workgroup kernel of round 4 */
#define DD_LAB_APPLY_WGS(n)    (((uint32_t)(n) & 0x1fffu) << 8)   /* dd_refine_apply: at most n workgroups (0 = the default, 4096) */

uint32_t dd_debug_tuning(uint32_t bits);

/* What the host side decides for a batch, without a GPU: out[0] lean kernels (1) or the generic ones (0), out[1] single pass,
 * out[2] DD_REFINE stage, out[3] float16 depth, out[4] tile of the two-pass kernels, out[5] pixels per lane of the single-pass
 * kernel, out[6] a gate kernel precedes the call (chained calls only), out[7] tiles of the two-pass tiling.  Host logic tests. */
int dd_debug_plan(const struct DDViewBatch *batch, int32_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
