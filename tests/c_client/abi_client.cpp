// Stand-alone client of the C ABI (include/ddcore.h): no Python, no torch.  Links libddcore.so, runs
// dd_plan + dd_scatter and the fused dd_unproject_compact on a small synthetic stack and checks both
// against a scalar float64 loop that follows scripts/test.py:194-233 (mask, depth > 0, row-major order,
// pinhole unprojection, R^T (p - t)).   hipcc -I include tests/c_client/abi_client.cpp -L depthdensifier_amd -lddcore
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "ddcore.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define DD(x) do { int r_ = (x); if (r_ < 0) { printf("ddcore error %d: %s (line %d)\n", r_, dd_last_error(), __LINE__); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) / 16777216.0f; }

int main() {
    const int V = 3, H = 72, W = 128;               // H*W % 8 == 0 -> the aligned (lean) kernels
    const size_t P = (size_t)H * W, N = (size_t)V * P;
    if (dd_abi_version() != DD_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }

    std::vector<float> depth(N), normal(N * 3);
    std::vector<uint8_t> mask(N), rgb(N * 3);
    for (size_t i = 0; i < N; ++i) {
        depth[i] = frand() < 0.05f ? 0.0f : 0.5f + 4.5f * frand();
        mask[i] = frand() < 0.8f;
        for (int c = 0; c < 3; ++c) { normal[3 * i + c] = frand() - 0.5f; rgb[3 * i + c] = (uint8_t)(frand() * 255.0f); }
    }
    // cameras: K = [fx 0 cx; 0 fy cy; 0 0 1], cam_from_world = [Rz(a) | t]
    std::vector<DDViewParams> params(V);
    std::vector<double> R(V * 9), T(V * 3), Kp(V * 4);
    for (int v = 0; v < V; ++v) {
        const double a = 0.3 + 0.7 * v, c = cos(a), s = sin(a);
        const double Rv[9] = {c, -s, 0, s, c, 0, 0, 0, 1}, t[3] = {0.1 * v, -0.2, 0.3 + v};
        const double fx = 100 + v, fy = 101 + v, cx = W / 2.0, cy = H / 2.0;
        memcpy(&R[9 * v], Rv, sizeof(Rv)); memcpy(&T[3 * v], t, sizeof(t));
        Kp[4 * v] = fx; Kp[4 * v + 1] = fy; Kp[4 * v + 2] = cx; Kp[4 * v + 3] = cy;
        const double Kinv[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};
        memset(&params[v], 0, sizeof(DDViewParams));
        for (int i = 0; i < 3; ++i) {
            double ct = 0;
            for (int j = 0; j < 3; ++j) {
                double m = 0;
                for (int k = 0; k < 3; ++k) m += Rv[3 * k + i] * Kinv[3 * k + j];     // (R^T K^-1)[i][j]
                params[v].ray_to_world[3 * i + j] = (float)m;
                params[v].rot[3 * i + j] = (float)Rv[3 * j + i];
                ct += Rv[3 * j + i] * t[j];
            }
            params[v].centre[i] = (float)-ct;
        }
    }
    // reference rows (float64)
    std::vector<double> ref;
    std::vector<int> ref_pix;
    std::vector<int64_t> ref_off(V + 1, 0);
    for (int v = 0; v < V; ++v) {
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)v * P + (size_t)y * W + x;
            if (!mask[i] || !(depth[i] > 0.0f)) continue;
            const double d = depth[i];
            const double pc[3] = {(x - Kp[4 * v + 2]) / Kp[4 * v] * d, (y - Kp[4 * v + 3]) / Kp[4 * v + 1] * d, d};
            for (int r = 0; r < 3; ++r) {
                double w = 0;
                for (int k = 0; k < 3; ++k) w += R[9 * v + 3 * k + r] * (pc[k] - T[3 * v + k]);
                ref.push_back(w);
            }
            ref_pix.push_back(y * W + x);
        }
        ref_off[v + 1] = (int64_t)ref_pix.size();
    }
    const int64_t n_ref = (int64_t)ref_pix.size();

    void *d_depth, *d_mask, *d_normal, *d_rgb, *d_params, *d_xyz, *d_nrm, *d_col, *d_pix, *d_off, *d_cur, *d_ws;
    CK(hipMalloc(&d_depth, N * 4)); CK(hipMalloc(&d_mask, N)); CK(hipMalloc(&d_normal, N * 12)); CK(hipMalloc(&d_rgb, N * 3));
    CK(hipMalloc(&d_params, V * sizeof(DDViewParams)));
    CK(hipMemcpy(d_depth, depth.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mask, mask.data(), N, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_normal, normal.data(), N * 12, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rgb, rgb.data(), N * 3, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_params, params.data(), V * sizeof(DDViewParams), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_xyz, N * 12)); CK(hipMalloc(&d_nrm, N * 12)); CK(hipMalloc(&d_col, N * 3)); CK(hipMalloc(&d_pix, N * 4));
    CK(hipMalloc(&d_off, (V + 1) * 8)); CK(hipMalloc(&d_cur, 8));

    DDViewBatch b; memset(&b, 0, sizeof(b));
    b.num_views = V; b.height = H; b.width = W; b.stride = 1;
    b.depth = d_depth; b.mask = (const uint8_t *)d_mask; b.normal = (const float *)d_normal; b.rgb = (const uint8_t *)d_rgb;
    b.params = (const DDViewParams *)d_params; b.depth_dtype = DD_F32; b.flags = DD_VALID_DEPTH_POSITIVE | DD_VALID_MASK;
    DDCloudOut out; memset(&out, 0, sizeof(out));
    out.xyz = (float *)d_xyz; out.normal = (float *)d_nrm; out.rgb = (uint8_t *)d_col; out.pixel_index = (int32_t *)d_pix; out.capacity = (int64_t)N;
    const int64_t wsb = dd_workspace_bytes(&b);
    if (wsb < 0) { printf("workspace query failed: %s\n", dd_last_error()); return 3; }
    CK(hipMalloc(&d_ws, (size_t)wsb)); CK(hipMemset(d_ws, 0, (size_t)wsb));

    hipStream_t stream; CK(hipStreamCreate(&stream));
    std::vector<float> xyz(N * 3), nrm(N * 3); std::vector<uint8_t> col(N * 3); std::vector<int> pix(N); std::vector<int64_t> off(V + 1);
    // mode 2 writes into arrays handed out by the HBM zone arena (dd_arena_*): ordinary device pointers for every entry point
    DDArena *arena = NULL;
    void *aptr[3] = {NULL, NULL, NULL};
    for (int mode = 0; mode < 5; ++mode) {       // 0: dd_plan + dd_scatter, 1: fused dd_unproject_compact, 2: fused, into arena arrays,
                                                 // 3: the reference's loop (scripts/test.py:131): ONE VIEW PER CALL, chained through one
                                                 //    cursor and one workspace that is never zeroed again (ABI 11: a call is one kernel launch)
                                                 // 4: the same loop with consecutive calls on TWO streams (ABI 12 / 13): DDViewBatch.chain,
                                                 //    a workspace per stream, the pair of streams probed with dd_streams_overlap first
        if (mode == 2) {
            if (dd_arena_create(0, 0, &arena) != DD_OK) { printf("dd_arena_create: %s\n", dd_arena_last_error()); return 6; }
            const int64_t sizes[3] = {(int64_t)N * 12, (int64_t)N * 12, (int64_t)N * 3};
            const int32_t layouts[3] = {DD_ARENA_ROTATED + 0, DD_ARENA_ROTATED + 1, DD_ARENA_ROTATED + 2};
            const int rc = dd_arena_alloc(arena, 3, sizes, layouts, (int64_t)16 << 30, aptr);
            if (rc < 0) { printf("dd_arena_alloc: %s\n", dd_arena_last_error()); return 6; }
            int32_t cls[8]; DDArenaStats st;
            if (dd_arena_classes(arena, aptr[0], cls, 8) != 1 || dd_arena_stats(arena, &st) != DD_OK || st.chunk_bytes != ((int64_t)1 << 30)) { printf("arena bookkeeping is wrong\n"); return 6; }
            if (dd_arena_free(arena, d_ws) != DD_ERR_INVALID_ARG) { printf("dd_arena_free accepted a foreign pointer\n"); return 6; }
            d_xyz = aptr[0]; d_nrm = aptr[1]; d_col = aptr[2];
            out.xyz = (float *)d_xyz; out.normal = (float *)d_nrm; out.rgb = (uint8_t *)d_col;
            printf("arena: %s, %lld chunks created, classes found %d\n", rc == 1 ? "allocated (not apart)" : "allocated apart", (long long)st.chunks_created, st.num_classes);
        }
        CK(hipMemsetAsync(d_cur, 0, 8, stream)); CK(hipMemsetAsync(d_xyz, 0xff, N * 12, stream));
        if (mode == 0) {
            DD(dd_plan(&b, (const int64_t *)d_cur, (int64_t *)d_off, d_ws, wsb, stream));
            DD(dd_scatter(&b, &out, (const int64_t *)d_off, d_ws, wsb, stream));
        } else if (mode == 3) {
            for (int rep = 0; rep < 3; ++rep) {          // three scans back to back: 3 V calls on the same workspace
                CK(hipMemsetAsync(d_cur, 0, 8, stream));
                for (int v = 0; v < V; ++v) {
                    DDViewBatch one = b;
                    one.num_views = 1; one.view_index_base = v;
                    one.depth = (const char *)d_depth + (size_t)v * P * 4; one.mask = (const uint8_t *)d_mask + (size_t)v * P;
                    one.normal = (const float *)d_normal + (size_t)v * P * 3; one.rgb = (const uint8_t *)d_rgb + (size_t)v * P * 3;
                    one.params = (const DDViewParams *)d_params + v;
                    // a view's two offsets land in d_off[v], d_off[v + 1]: the next call rewrites d_off[v + 1] with the same row
                    DD(dd_unproject_compact(&one, &out, (int64_t *)d_off + v, (int64_t *)d_cur, d_ws, wsb, stream));
                }
            }
            CK(hipStreamSynchronize(stream));
            int32_t hdr[16]; CK(hipMemcpy(hdr, d_ws, 64, hipMemcpyDeviceToHost));
            if (hdr[1] != 0 || hdr[4] != 0) { printf("mode 3: workspace header after the chain: error %d, ticket %d\n", hdr[1], hdr[4]); return 4; }
            int64_t cur; CK(hipMemcpy(&cur, d_cur, 8, hipMemcpyDeviceToHost));
            if (cur != n_ref) { printf("mode 3: cursor %lld != %lld\n", (long long)cur, (long long)n_ref); return 4; }
        } else if (mode == 4) {
            hipStream_t side[2]; CK(hipStreamCreateWithFlags(&side[0], hipStreamNonBlocking));
            void *d_scratch; CK(hipMalloc(&d_scratch, 8));
            int32_t seen = 0, tried = 0;
            std::vector<hipStream_t> spare;
            for (; tried < 8 && !seen; ++tried) {        // the runtime may have put both streams on one hardware queue: take another second stream
                CK(hipStreamCreateWithFlags(&side[1], hipStreamNonBlocking));
                DD(dd_streams_overlap(side[0], side[1], (int32_t *)d_scratch, &seen));
                if (!seen) spare.push_back(side[1]);
            }
            if (!seen) { printf("mode 4: no pair of streams that run side by side in 8 tries: chaining skipped\n"); }
            else {
                void *d_chain, *d_ws2[2]; hipEvent_t ev;
                CK(hipMalloc(&d_chain, 8)); CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                for (int k = 0; k < 2; ++k) { CK(hipMalloc(&d_ws2[k], (size_t)wsb)); CK(hipMemset(d_ws2[k], 0, (size_t)wsb)); }
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(d_cur, 0, 8, stream)); CK(hipMemsetAsync(d_chain, 0, 8, stream));      // sequence 0 starts at row 0
                    for (int k = 0; k < 2; ++k) DD(dd_stream_fork(ev, stream, side[k]));                      // both side streams behind that
                    for (int v = 0; v < V; ++v) {
                        DDViewBatch one = b;
                        one.num_views = 1; one.view_index_base = v;
                        one.depth = (const char *)d_depth + (size_t)v * P * 4; one.mask = (const uint8_t *)d_mask + (size_t)v * P;
                        one.normal = (const float *)d_normal + (size_t)v * P * 3; one.rgb = (const uint8_t *)d_rgb + (size_t)v * P * 3;
                        one.params = (const DDViewParams *)d_params + v;
                        one.chain = (int64_t *)d_chain; one.chain_seq = v;
                        DD(dd_unproject_compact(&one, &out, (int64_t *)d_off + v, (int64_t *)d_cur, d_ws2[v & 1], wsb, side[v & 1]));
                    }
                    for (int k = 0; k < 2; ++k) DD(dd_stream_fork(ev, side[k], stream));                      // join: the caller's stream behind both
                    CK(hipStreamSynchronize(stream));      // (the next repetition rewrites the chain word: dd_stream_fork may skip an idle stream, so be sure)
                }
                int64_t cur; CK(hipMemcpy(&cur, d_cur, 8, hipMemcpyDeviceToHost));
                if (cur != n_ref) { printf("mode 4: cursor %lld != %lld\n", (long long)cur, (long long)n_ref); return 4; }
                for (int k = 0; k < 2; ++k) {
                    int32_t hdr[16]; CK(hipMemcpy(hdr, d_ws2[k], 64, hipMemcpyDeviceToHost));
                    if (hdr[1] != 0) { printf("mode 4: error word %d in workspace %d\n", hdr[1], k); return 4; }
                    CK(hipFree(d_ws2[k]));
                }
                CK(hipFree(d_chain)); CK(hipEventDestroy(ev));
                printf("mode 4: %d views chained across two streams (%d pair(s) of streams probed)\n", V, tried);
            }
            for (hipStream_t sp : spare) CK(hipStreamDestroy(sp));
            if (seen) CK(hipStreamDestroy(side[1]));
            CK(hipStreamDestroy(side[0])); CK(hipFree(d_scratch));
            if (!seen) DD(dd_unproject_compact(&b, &out, (int64_t *)d_off, (int64_t *)d_cur, d_ws, wsb, stream));
        } else {
            DD(dd_unproject_compact(&b, &out, (int64_t *)d_off, (int64_t *)d_cur, d_ws, wsb, stream));
        }
        CK(hipStreamSynchronize(stream));
        CK(hipMemcpy(off.data(), d_off, (V + 1) * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(xyz.data(), d_xyz, N * 12, hipMemcpyDeviceToHost));
        CK(hipMemcpy(nrm.data(), d_nrm, N * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(col.data(), d_col, N * 3, hipMemcpyDeviceToHost));
        CK(hipMemcpy(pix.data(), d_pix, N * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v <= V; ++v) if (off[v] != ref_off[v]) { printf("mode %d: view offset %d is %lld, expected %lld\n", mode, v, (long long)off[v], (long long)ref_off[v]); return 4; }
        double worst = 0;
        for (int64_t i = 0; i < n_ref; ++i) {
            if (pix[i] != ref_pix[i]) { printf("mode %d: pixel index mismatch at row %lld\n", mode, (long long)i); return 4; }
            int v = 0; while (i >= ref_off[v + 1]) ++v;
            const size_t src = (size_t)v * P + (size_t)pix[i];
            for (int c = 0; c < 3; ++c) {
                const double e = fabs((double)xyz[3 * i + c] - ref[3 * i + c]) / fmax(1.0, fabs(ref[3 * i + c]));
                if (e > worst) worst = e;
                if (nrm[3 * i + c] != normal[3 * src + c] || col[3 * i + c] != rgb[3 * src + c]) { printf("mode %d: attribute mismatch at row %lld\n", mode, (long long)i); return 4; }
            }
        }
        if (worst > 1e-4) { printf("mode %d: xyz relative error %g\n", mode, worst); return 4; }
        if (mode == 1) { int64_t cur; CK(hipMemcpy(&cur, d_cur, 8, hipMemcpyDeviceToHost)); if (cur != n_ref) { printf("cursor %lld != %lld\n", (long long)cur, (long long)n_ref); return 4; } }
        printf("mode %d: %lld points, xyz max rel err %.2e\n", mode, (long long)n_ref, worst);
    }
    // (the arena's arrays stay in use below: the records are formatted from them)
    // the cloud as points3D.bin records (dd_format_points3d): 51 bytes per point, checked field by field
    {
        void *d_rec;
        CK(hipMalloc(&d_rec, (size_t)n_ref * 51 + 16));
        const uint64_t first_id = 1000000007ull;
        if (dd_format_points3d((const float *)d_xyz, (const uint8_t *)d_col, NULL, n_ref, first_id, (uint8_t *)d_rec, stream) != DD_OK) {
            printf("dd_format_points3d: %s\n", dd_model_last_error()); return 3; }
        CK(hipStreamSynchronize(stream));
        std::vector<unsigned char> recs((size_t)n_ref * 51);
        CK(hipMemcpy(recs.data(), d_rec, recs.size(), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n_ref; ++i) {
            const unsigned char *r = recs.data() + (size_t)i * 51;
            uint64_t id, track; double p[3], err;
            memcpy(&id, r, 8); memcpy(p, r + 8, 24); memcpy(&err, r + 35, 8); memcpy(&track, r + 43, 8);
            bool ok = id == first_id + (uint64_t)i && err == -1.0 && track == 0;
            for (int c = 0; c < 3; ++c) ok = ok && p[c] == (double)xyz[3 * i + c] && r[32 + c] == col[3 * i + c];
            if (!ok) { printf("points3D record %lld is wrong\n", (long long)i); return 4; }
        }
        if (dd_format_points3d(NULL, NULL, NULL, 1, 1, (uint8_t *)d_rec, stream) != DD_ERR_INVALID_ARG) { printf("error convention broken\n"); return 5; }
        printf("points3D records OK\n");
    }
    // error convention: invalid argument -> negative code + message, nothing thrown
    b.stride = 0;
    if (dd_workspace_bytes(&b) != DD_ERR_INVALID_ARG || strstr(dd_last_error(), "stride") == NULL) { printf("error convention broken\n"); return 5; }
    for (int k = 0; k < 3; ++k) if (dd_arena_free(arena, aptr[k]) != DD_OK) { printf("dd_arena_free: %s\n", dd_arena_last_error()); return 6; }
    if (dd_arena_trim(arena, 0) != DD_OK || dd_arena_destroy(arena) != DD_OK) { printf("arena teardown failed\n"); return 6; }
    printf("C ABI OK\n");
    return 0;
}
