// One rank of a C++ client of the multi-GPU fuse through the C ABI alone (no Python, no torch):
//   dd_count_valid -> ncclAllGather of the counts -> dd_unproject_compact with cursor = rank_rows[rank] into the GLOBAL
//   buffers -> dd_allgatherv (in place) -> every rank holds the cloud a single process computes, bit for bit.
// usage: abi_allgatherv <rank> <world> <unique-id file>      (built and launched by tests/test_fuse_gpu.py)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <vector>

#include "ddcore.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define NC(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { printf("RCCL error %s at line %d\n", ncclGetErrorString(r_), __LINE__); return 2; } } while (0)
#define DD(x) do { int r_ = (x); if (r_ != DD_OK) { printf("ddcore error %d at line %d: %s / %s\n", r_, __LINE__, dd_last_error(), dd_comm_last_error()); return 3; } } while (0)

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main(int argc, char **argv) {
    if (argc < 4) { printf("usage: %s rank world idfile\n", argv[0]); return 1; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    const char *idfile = argv[3];
    int ndev = 0; CK(hipGetDeviceCount(&ndev));
    CK(hipSetDevice(rank % ndev));
    ncclUniqueId id;
    if (rank == 0) {
        NC(ncclGetUniqueId(&id));
        char tmp[512]; snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE *f = fopen(tmp, "wb"); fwrite(&id, sizeof id, 1, f); fclose(f); rename(tmp, idfile);
    } else {
        FILE *f = NULL;
        for (int i = 0; i < 600 && !(f = fopen(idfile, "rb")); ++i) usleep(100000);
        if (!f || fread(&id, sizeof id, 1, f) != 1) { printf("rank %d: no unique id\n", rank); return 2; }
        fclose(f);
    }
    ncclComm_t comm; NC(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream; CK(hipStreamCreate(&stream));

    // the same seeded scene on every rank: VPR views per rank, H x W, depth / mask / rgb, one camera block per view
    const int VPR = 3, V = VPR * world, H = 64, W = 96, P = H * W;
    std::vector<float> depth((size_t)V * P); std::vector<uint8_t> mask((size_t)V * P), rgb((size_t)V * P * 3);
    std::vector<DDViewParams> params(V);
    uint32_t s = 12345u;
    for (size_t i = 0; i < depth.size(); ++i) { depth[i] = 0.5f + (lcg(s) % 4096) / 1024.0f; mask[i] = (lcg(s) % 100) < 70; }
    for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (uint8_t)lcg(s);
    memset(params.data(), 0, V * sizeof(DDViewParams));
    for (int v = 0; v < V; ++v) {          // identity rotation, focal 80, principal point at the centre, centre shifted per view
        float *m = params[v].ray_to_world;
        m[0] = 1.0f / 80; m[2] = -W / 160.0f; m[4] = 1.0f / 80; m[5] = -H / 160.0f; m[8] = 1.0f;
        params[v].centre[0] = 0.25f * v; params[v].rot[0] = params[v].rot[4] = params[v].rot[8] = 1.0f;
    }
    void *d_depth, *d_mask, *d_rgb, *d_params;
    CK(hipMalloc(&d_depth, depth.size() * 4)); CK(hipMalloc(&d_mask, mask.size())); CK(hipMalloc(&d_rgb, rgb.size())); CK(hipMalloc(&d_params, V * sizeof(DDViewParams)));
    CK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rgb, rgb.data(), rgb.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_params, params.data(), V * sizeof(DDViewParams), hipMemcpyHostToDevice));

    auto batch = [&](int v0, int nv) {
        DDViewBatch b; memset(&b, 0, sizeof b);
        b.num_views = nv; b.height = H; b.width = W; b.stride = 1;
        b.depth = (const float *)d_depth + (size_t)v0 * P; b.mask = (const uint8_t *)d_mask + (size_t)v0 * P; b.rgb = (const uint8_t *)d_rgb + (size_t)v0 * P * 3;
        b.params = (const DDViewParams *)d_params + v0; b.depth_dtype = DD_F32; b.flags = DD_VALID_DEPTH_POSITIVE | DD_VALID_MASK; b.view_index_base = v0;
        return b;
    };
    // 1. counts of my views, all-gathered (device), then the row layout on the host
    int64_t *d_counts; CK(hipMalloc((void **)&d_counts, V * 8));
    DDViewBatch mine = batch(rank * VPR, VPR);
    DD(dd_count_valid(&mine, d_counts + rank * VPR, stream));
    NC(ncclAllGather(d_counts + rank * VPR, d_counts, VPR, ncclInt64, comm, stream));
    CK(hipStreamSynchronize(stream));
    std::vector<int64_t> counts(V), rank_rows(world + 1, 0);
    CK(hipMemcpy(counts.data(), d_counts, V * 8, hipMemcpyDeviceToHost));
    for (int r = 0; r < world; ++r) { rank_rows[r + 1] = rank_rows[r]; for (int k = 0; k < VPR; ++k) rank_rows[r + 1] += counts[r * VPR + k]; }
    const int64_t N = rank_rows[world];

    // 2. reference: the whole scene in one call on this GPU
    void *r_xyz, *r_rgb, *r_rec, *d_off, *d_cur, *d_ws;
    CK(hipMalloc(&r_xyz, N * 12 + 16)); CK(hipMalloc(&r_rgb, N * 3 + 16)); CK(hipMalloc(&r_rec, N * 16 + 16)); CK(hipMalloc(&d_off, (V + 1) * 8)); CK(hipMalloc(&d_cur, 8));
    DDViewBatch all = batch(0, V);
    const int64_t wsb = dd_workspace_bytes(&all);
    CK(hipMalloc(&d_ws, wsb)); CK(hipMemset(d_ws, 0, wsb)); CK(hipMemset(d_cur, 0, 8));
    DDCloudOut ref; memset(&ref, 0, sizeof ref);
    ref.xyz = (float *)r_xyz; ref.rgb = (uint8_t *)r_rgb; ref.xyz_rgba = (uint32_t *)r_rec; ref.capacity = N;
    DD(dd_unproject_compact(&all, &ref, (int64_t *)d_off, (int64_t *)d_cur, d_ws, wsb, stream));
    CK(hipStreamSynchronize(stream));
    int64_t cur; CK(hipMemcpy(&cur, d_cur, 8, hipMemcpyDeviceToHost));
    if (cur != N) { printf("rank %d: reference wrote %lld rows, counts say %lld\n", rank, (long long)cur, (long long)N); return 4; }

    // 3. my shard, written at rank_rows[rank] of the GLOBAL buffers, then the in-place all-gatherv
    void *g_xyz, *g_rgb, *g_rec;
    CK(hipMalloc(&g_xyz, N * 12 + 16)); CK(hipMalloc(&g_rgb, N * 3 + 16)); CK(hipMalloc(&g_rec, N * 16 + 16));
    CK(hipMemset(g_xyz, 0xee, N * 12)); CK(hipMemset(g_rgb, 0xee, N * 3)); CK(hipMemset(g_rec, 0xee, N * 16));
    DDCloudOut glob; memset(&glob, 0, sizeof glob);
    glob.xyz = (float *)g_xyz; glob.rgb = (uint8_t *)g_rgb; glob.xyz_rgba = (uint32_t *)g_rec; glob.capacity = N;
    CK(hipMemcpy(d_cur, &rank_rows[rank], 8, hipMemcpyHostToDevice));
    DD(dd_unproject_compact(&mine, &glob, (int64_t *)d_off, (int64_t *)d_cur, d_ws, wsb, stream));
    DD(dd_allgatherv(comm, rank, world, &glob, rank_rows.data(), -1, stream));
    CK(hipStreamSynchronize(stream));
    std::vector<char> a(N * 16), b(N * 16);
    struct { void *g, *r; size_t bytes; const char *name; } cmp[] = {{g_xyz, r_xyz, (size_t)N * 12, "xyz"}, {g_rgb, r_rgb, (size_t)N * 3, "rgb"}, {g_rec, r_rec, (size_t)N * 16, "xyz_rgba"}};
    for (auto &c : cmp) {
        CK(hipMemcpy(a.data(), c.g, c.bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), c.r, c.bytes, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), c.bytes) != 0) { printf("rank %d: fused %s differs from the one-process cloud\n", rank, c.name); return 4; }
    }
    // 4. gather-to-owner of the 16-byte records: only the last rank receives, the others hold just their own rows
    const int owner = world - 1;
    const int64_t own_n = rank_rows[rank + 1] - rank_rows[rank];
    DDCloudOut part; memset(&part, 0, sizeof part);
    void *p_rec = NULL;
    if (rank == owner) { CK(hipMemset(g_rec, 0xdd, N * 16)); part.xyz_rgba = (uint32_t *)g_rec; part.capacity = N; CK(hipMemcpy(d_cur, &rank_rows[rank], 8, hipMemcpyHostToDevice)); }
    else { CK(hipMalloc(&p_rec, own_n * 16 + 16)); part.xyz_rgba = (uint32_t *)p_rec; part.capacity = own_n; CK(hipMemset(d_cur, 0, 8)); }
    DD(dd_unproject_compact(&mine, &part, (int64_t *)d_off, (int64_t *)d_cur, d_ws, wsb, stream));
    DD(dd_allgatherv(comm, rank, world, &part, rank_rows.data(), owner, stream));
    CK(hipStreamSynchronize(stream));
    if (rank == owner) {
        CK(hipMemcpy(a.data(), g_rec, N * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), r_rec, N * 16, hipMemcpyDeviceToHost));
        if (memcmp(a.data(), b.data(), N * 16) != 0) { printf("rank %d: gather-to-owner differs\n", rank); return 4; }
    }
    // error conventions
    if (dd_allgatherv(comm, rank, world, &glob, rank_rows.data(), world, stream) != DD_ERR_INVALID_ARG || !strstr(dd_comm_last_error(), "dst")) { printf("error convention broken\n"); return 5; }
    printf("rank %d/%d: C ABI all-gatherv OK, %lld points\n", rank, world, (long long)N);
    ncclCommDestroy(comm);
    return 0;
}
