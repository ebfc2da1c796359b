// Round 3: does the REAL densify kernel (xyz-only, f16 depth in: BASELINE configs[4]) gain from an output array whose
// consecutive pieces alternate between the three HBM classes, like the one-shot row-store micro-benchmark does
// (zone_interleave.hip: 5.7 -> 7.1 TB/s at 32-64 MiB pieces)?  Same classification as there (fresh process: runs of small
// handles are class-pure), then dd_unproject_compact of libddcore.so on 20 dense 12 MP views into a class-pure and into an
// interleaved 3 GiB array.                                                                          GPU box only.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude -o build/zone_kernel tools/experiments/zone_kernel.hip -Ldepthdensifier_amd -lddcore -Wl,-rpath,$PWD/depthdensifier_amd
//   build/zone_kernel [piece MiB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "ddcore.h"
typedef float f3 __attribute__((ext_vector_type(3)));
typedef f3 f3u __attribute__((aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void w2(float *a, float *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 p = {(float)i, 1.0f, 2.0f};
    *reinterpret_cast<f3u *>(a + 3 * i) = p; *reinterpret_cast<f3u *>(b + 3 * i) = p;
}
__global__ void w1(float *a, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 p = {(float)i, 1.0f, 2.0f};
    *reinterpret_cast<f3u *>(a + 3 * i) = p;
}
__global__ void fill4(f4 *a, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) a[i] = f4{1.f, 2.f, 3.f, 4.f};
}
__global__ void copy4(const f4 *a, f4 *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ void read4(const f4 *a, float *sink, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { const f4 v = a[i]; if (v.x == 123.456f) sink[0] = v.y; }
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <class F> float best(F f, int reps = 5) {
        f();
        float b = 1e30f;
        for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); b = std::min(b, ms); }
        return b;
    }
};

int main(int argc, char **argv) {
    const size_t CHUNK = (size_t)1 << 30;
    const size_t PIECE = (size_t)(argc > 1 ? atoi(argv[1]) : 2) << 20;
    const int NC = 96;
    const size_t ROWS = (size_t)32 << 20;            // classification window
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    // hipMemMap takes no offset into a handle on this runtime: every piece is a handle of its own; a "chunk" is a run of
    // PPC consecutively created pieces (assumed to come from one neighbourhood of physical memory, checked by the probe)
    const size_t PPC = CHUNK / PIECE;
    char *scout; CK(hipMemAddressReserve((void **)&scout, NC * CHUNK, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs((size_t)NC * PPC);
    for (size_t i = 0; i < hs.size(); ++i) { CK(hipMemCreate(&hs[i], PIECE, &prop, 0)); CK(hipMemMap(scout + i * PIECE, PIECE, 0, hs[i], 0)); }
    CK(hipMemSetAccess(scout, NC * CHUNK, &acc, 1));
    printf("%zu handles of %zu MiB created and mapped\n", hs.size(), PIECE >> 20); fflush(stdout);
    Timer T;
    const dim3 g2((unsigned)((ROWS + 255) / 256)), blk(256);
    auto pair = [&](int i, int j) { return T.best([&] { w2<<<g2, blk>>>((float *)(scout + i * CHUNK), (float *)(scout + j * CHUNK + (i == j ? CHUNK / 2 : 0)), ROWS); }, 3); };
    // classes: against chunk 0, then against the first fast chunk, the rest is the third class
    std::vector<int> cls(NC, -1);
    std::vector<float> t0(NC);
    float lo = 1e9f, hi = 0;
    for (int j = 1; j < NC; ++j) { t0[j] = pair(0, j); lo = std::min(lo, t0[j]); hi = std::max(hi, t0[j]); }
    const float cut = std::sqrt(lo * hi);
    cls[0] = 0;
    int r1 = -1;
    for (int j = 1; j < NC; ++j) { if (t0[j] > cut) cls[j] = 0; else if (r1 < 0) r1 = j; }
    if (r1 < 0 || hi / lo < 1.1f) { printf("only one class among %d chunks (levels %.4f .. %.4f)\n", NC, lo, hi); return 0; }
    cls[r1] = 1;
    for (int j = 1; j < NC; ++j) if (cls[j] < 0) cls[j] = pair(r1, j) > cut ? 1 : 2;
    int cnt[3] = {0, 0, 0};
    for (int j = 0; j < NC; ++j) cnt[cls[j]]++;
    printf("%d chunks: class sizes %d %d %d (levels %.4f / %.4f ms)\n   ", NC, cnt[0], cnt[1], cnt[2], lo, hi);
    for (int j = 0; j < NC; ++j) putchar('A' + cls[j]);
    printf("\n"); fflush(stdout);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(scout, NC * CHUNK));
    std::vector<int> byc[3];
    for (int j = 0; j < NC; ++j) byc[cls[j]].push_back(j);
    if (byc[0].size() < 9 || byc[1].size() < 6 || byc[2].size() < 3) { printf("not enough chunks per class for the layouts\n"); return 0; }
    auto takec = [&](int c) { int j = byc[c].back(); byc[c].pop_back(); return j; };

    const size_t ARR = 3 * CHUNK;
    // class-pure array: three chunks of one class back to back
    auto pure = [&](int c) {
        char *va; CK(hipMemAddressReserve((void **)&va, ARR, 0, nullptr, 0));
        for (int k = 0; k < 3; ++k) { const int c3 = takec(c); for (size_t q = 0; q < PPC; ++q) CK(hipMemMap(va + k * CHUNK + q * PIECE, PIECE, 0, hs[(size_t)c3 * PPC + q], 0)); }
        CK(hipMemSetAccess(va, ARR, &acc, 1));
        return va;
    };
    // interleaved array: piece k comes from the chunk of class (k + phase) % 3, at offset (k / 3) * PIECE inside it
    bool offset_ok = true;
    auto inter = [&](int phase) -> char * {
        char *va; CK(hipMemAddressReserve((void **)&va, ARR, 0, nullptr, 0));
        int ch[3] = {takec(0), takec(1), takec(2)};
        const size_t np = ARR / PIECE;
        for (size_t k = 0; k < np; ++k) {
            CK(hipMemMap(va + k * PIECE, PIECE, 0, hs[(size_t)ch[(k + phase) % 3] * PPC + k / 3], 0));
        }
        CK(hipMemSetAccess(va, ARR, &acc, 1));
        return va;
    };
    char *P0 = pure(0), *P0b = pure(0), *P1 = pure(1);
    char *IA = inter(0);
    char *IB = offset_ok ? inter(1) : nullptr;
    char *IC = offset_ok ? inter(0) : nullptr;
    if (!offset_ok) { printf("sub-chunk mapping offsets are not supported: interleaving needs piece-sized handles\n"); return 0; }
    printf("piece %zu MiB; arrays of 3 GiB: P0, P0b (class A), P1 (class B), IA (phase 0), IB (phase 1), IC (phase 0)\n", PIECE >> 20);

    // ---- the real kernel: 20 dense 12 MP views, f16 depth -> xyz rows (2.93 GB) ----
    const int V = 20, H = 3024, W = 4032;
    const size_t N = (size_t)V * H * W;
    void *d_depth; CK(hipMalloc(&d_depth, N * 2));
    { std::vector<unsigned short> h((size_t)H * W, 0x4200); for (int v = 0; v < V; ++v) CK(hipMemcpy((char *)d_depth + (size_t)v * H * W * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }   // 3.0 everywhere
    std::vector<DDViewParams> params(V);
    for (int v = 0; v < V; ++v) { memset(&params[v], 0, sizeof(DDViewParams)); params[v].ray_to_world[0] = 1e-3f; params[v].ray_to_world[4] = 1e-3f; params[v].ray_to_world[8] = 1.f; params[v].rot[0] = params[v].rot[4] = params[v].rot[8] = 1.f; }
    void *d_params; CK(hipMalloc(&d_params, V * sizeof(DDViewParams))); CK(hipMemcpy(d_params, params.data(), V * sizeof(DDViewParams), hipMemcpyHostToDevice));
    void *d_off, *d_cur; CK(hipMalloc(&d_off, (V + 1) * 8)); CK(hipMalloc(&d_cur, 8));
    DDViewBatch b; memset(&b, 0, sizeof(b));
    b.num_views = V; b.height = H; b.width = W; b.stride = 1; b.depth = d_depth; b.params = (const DDViewParams *)d_params;
    b.depth_dtype = DD_F16; b.flags = DD_VALID_DEPTH_POSITIVE;
    const int64_t wsb = dd_workspace_bytes(&b);
    if (wsb < 0) { printf("dd_workspace_bytes: %s\n", dd_last_error()); return 1; }
    void *d_ws; CK(hipMalloc(&d_ws, (size_t)wsb)); CK(hipMemset(d_ws, 0, (size_t)wsb));
    if (N * 12 > ARR) { printf("cloud does not fit the arrays\n"); return 1; }
    struct { const char *n; char *p; } outs[] = {{"P0 (class A)", P0}, {"P1 (class B)", P1}, {"IA (interleaved)", IA}, {"IB (interleaved, other phase)", IB}, {"P0 again", P0}};
    for (auto &o : outs) {
        DDCloudOut out; memset(&out, 0, sizeof(out));
        out.xyz = (float *)o.p; out.capacity = (int64_t)N;
        const float t = T.best([&] {
            CK(hipMemsetAsync(d_cur, 0, 8, 0));
            if (dd_unproject_compact(&b, &out, (int64_t *)d_off, (int64_t *)d_cur, d_ws, wsb, nullptr) != DD_OK) { printf("dd_unproject_compact: %s\n", dd_last_error()); exit(1); }
        }, 6);
        long long cur = 0; CK(hipMemcpy(&cur, d_cur, 8, hipMemcpyDeviceToHost));
        printf("  densify kernel, xyz into %-30s %.3f ms  = %.1f GB/s algorithmic (%lld points)\n", o.n, t, (N * 14.0) / (t * 1e-3) / 1e9, cur);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
