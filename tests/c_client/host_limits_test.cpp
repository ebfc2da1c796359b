// host_limits_test.cpp -- the host side of libddcore at the edges of its domain, for a build of csrc/ddcore.hip whose HOST code carries
// AddressSanitizer + UBSan (tests/test_host_cpu.py::test_host_planning_at_its_limits_under_sanitizers): plans and workspace sizes of
// batches from one pixel to the largest view (2^31 - 1 pixels) and the largest tile count, every stride, every tuning a caller may
// choose, chained and not -- no overflow in the size arithmetic, a sane answer or a clean refusal everywhere.  No GPU: dd_debug_plan
// and dd_workspace_bytes launch nothing.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ddcore.h"
#include "ddcore_lab.h"

int main() {
    const int32_t dims[] = {1, 2, 3, 7, 8, 63, 64, 120, 1080, 1920, 3071, 3072, 4096, 46340, 46341, 65535, 1 << 20, (1 << 30), 2147483647};
    const int32_t views[] = {1, 2, 16, 185, 2000, 1 << 20, 2147483647};
    const int32_t strides[] = {1, 2, 3, 32, 1 << 20};
    const uint32_t tunings[] = {0u, DD_TUNE_GENERIC, DD_TUNE_TWO_PASS, DD_TUNE_SINGLE_PASS, DD_TUNE_DENSE_TILES, DD_TUNE_ASSUME_DENSE, DD_TUNE_TILE_SMALL,
                                DD_TUNE_TILE_LARGE, DD_TUNE_BY_INDEX, DD_TUNE_TWO_PASS | DD_TUNE_INTERLEAVE(8), DD_TUNE_BY_INDEX | DD_TUNE_TILE_LARGE};
    long ok = 0, refused = 0;
    for (int32_t H : dims) for (int32_t W : dims) for (int32_t V : views) for (int32_t s : strides) for (uint32_t t : tunings) for (int chained = 0; chained < 2; ++chained)
        for (int f16 = 0; f16 < 2; ++f16) {
            DDViewBatch b; memset(&b, 0, sizeof(b));
            b.num_views = V; b.height = H; b.width = W; b.stride = s;
            b.depth = (const void *)0x10000; b.params = (const DDViewParams *)0x20000; b.mask = (const uint8_t *)0x30000;
            b.depth_dtype = f16 ? DD_F16 : DD_F32; b.flags = DD_VALID_MASK | DD_VALID_DEPTH_POSITIVE; b.tuning = t;
            if (chained) { b.chain = (int64_t *)0x40000; b.chain_seq = 0x7fffffffffffffffll; }
            int32_t out[8];
            const int rc = dd_debug_plan(&b, out);
            const int64_t ws = dd_workspace_bytes(&b);
            if (rc == DD_OK) {
                ++ok;
                if (ws < 64 || out[7] < 1 || out[4] < 1) { printf("H %d W %d V %d stride %d tuning %#x: plan ok but workspace %lld, tiles %d, tile %d\n", H, W, V, s, t, (long long)ws, out[7], out[4]); return 1; }
                // the workspace holds at least 8 bytes per tile and per view
                if (ws < (int64_t)out[7] * 8 + (int64_t)V * 8) { printf("H %d W %d V %d stride %d tuning %#x: workspace %lld below its records (%d tiles)\n", H, W, V, s, t, (long long)ws, out[7]); return 1; }
            } else {
                ++refused;
                if (rc != DD_ERR_INVALID_ARG && rc != DD_ERR_UNSUPPORTED) { printf("H %d W %d V %d: rc %d\n", H, W, V, rc); return 1; }
                if (!dd_last_error()[0]) { printf("a refusal without a message\n"); return 1; }
            }
        }
    // the filter's workspace sizes: any number of views and points
    const int64_t counts[] = {0, 1, 255, 256, 65536, (int64_t)1 << 31, (int64_t)1 << 40, (int64_t)1 << 48, ((int64_t)1 << 48) + 1, INT64_MAX - 1, INT64_MAX};
    for (int32_t V : views) for (int64_t n : counts) {
        const int64_t a = dd_votes_workspace_bytes(V, n), c = dd_compact_workspace_bytes(n);
        if (n <= ((int64_t)1 << 48) ? (a < 64 || c < 16) : (a >= 0 || c >= 0)) { printf("workspace bytes for %d views, %lld points: %lld / %lld\n", V, (long long)n, (long long)a, (long long)c); return 1; }
    }
    if (dd_votes_workspace_bytes(0, 10) >= 0 || dd_votes_workspace_bytes(3, -1) >= 0 || dd_compact_workspace_bytes(-1) >= 0) { printf("negative sizes accepted\n"); return 1; }
    printf("host limits OK: %ld plans, %ld clean refusals\n", ok, refused);
    return 0;
}
