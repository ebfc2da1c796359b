// Round 3 prototype of the HBM zone arena: physical chunks from the virtual-memory API (hipMemCreate), all mapped into one
// range, classified pairwise with a small lock-step two-stream write kernel.  Questions: (1) does a plain two-stream kernel
// see the classes the densify kernel sees (placement9.py)?  with or without a read stream?  (2) do VMM chunks behave like
// hipMalloc memory?  (3) what do create / map cost?                                                        GPU box only.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/zone_probe tools/experiments/zone_probe.hip && /tmp/zone_probe [chunk MiB] [max GiB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// two row streams written in lock step, 12 bytes per row each (the densify kernel's points + normals)
// (rows are 3 floats at a 12-byte pitch: an ext_vector_type(3) pointer would step 16 bytes)
typedef f3 f3u __attribute__((aligned(4)));
__global__ void w2(float *a, float *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 p = {(float)i, 1.0f, 2.0f};
    *reinterpret_cast<f3u *>(a + 3 * i) = p; *reinterpret_cast<f3u *>(b + 3 * i) = p;
}
// the same with the reads of the densify kernel in front: 4 + 1 bytes per row and a 12-byte row
__global__ void m2(const float *d, const unsigned char *m, const float *nin, float *a, float *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dd = d[i] + (float)m[i];
    const f3 nn = *reinterpret_cast<const f3u *>(nin + 3 * i);
    f3 p = {dd, dd * 2.0f, dd * 3.0f};
    *reinterpret_cast<f3u *>(a + 3 * i) = p; *reinterpret_cast<f3u *>(b + 3 * i) = nn;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    template <class F> float med(F f, int reps = 5) {
        f();
        std::vector<float> ts;
        for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
        std::sort(ts.begin(), ts.end());
        return ts[reps / 2];
    }
};

static void show(const char *title, const std::vector<float> &t) {
    float lo = 1e9, hi = 0;
    for (float x : t) if (x > 0) { lo = std::min(lo, x); hi = std::max(hi, x); }
    printf("%s  min %.4f max %.4f ms (digit: 0 fast .. 9 slow, '-' = the reference itself)\n   ", title, lo, hi);
    for (size_t i = 0; i < t.size(); ++i) { if (t[i] <= 0) putchar('-'); else putchar('0' + std::min(9, (int)(10 * (t[i] - lo) / std::max(hi - lo, 1e-9f)))); if (i % 64 == 63) printf("\n   "); }
    printf("\n"); fflush(stdout);
}

int main(int argc, char **argv) {
    const size_t CHUNK = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
    const size_t MAXB = (size_t)(argc > 2 ? atoi(argv[2]) : 224) << 30;
    const size_t ROWS = std::min<size_t>(32u << 20, CHUNK / 12 / 2);      // window: up to 384 MiB, at most half a chunk
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f; chunk %zu MiB; probe window %zu rows = %.0f MiB per stream\n", fr / 1073741824.0, tot / 1073741824.0, CHUNK >> 20, ROWS, ROWS * 12 / 1048576.0);
    // fixed read region for m2 (hipMalloc, allocated first)
    float *d; unsigned char *mk; float *nin;
    CK(hipMalloc(&d, ROWS * 4)); CK(hipMalloc(&mk, ROWS)); CK(hipMalloc(&nin, ROWS * 12));
    CK(hipMemset(d, 0, ROWS * 4)); CK(hipMemset(mk, 1, ROWS)); CK(hipMemset(nin, 0, ROWS * 12));

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    CK(hipMemGetInfo(&fr, &tot));
    const size_t total = std::min(MAXB, fr - ((size_t)12 << 30)) / CHUNK * CHUNK;
    const int NC = (int)(total / CHUNK);
    char *va; CK(hipMemAddressReserve((void **)&va, total, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs(NC);
    double t0 = now();
    for (int i = 0; i < NC; ++i) CK(hipMemCreate(&hs[i], CHUNK, &prop, 0));
    double t1 = now();
    for (int i = 0; i < NC; ++i) CK(hipMemMap(va + (size_t)i * CHUNK, CHUNK, 0, hs[i], 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    double t2 = now();
    printf("granularity %zu; %d chunks = %.0f GiB: create %.3f s, map + access %.3f s\n", gran, NC, total / 1073741824.0, t1 - t0, t2 - t1); fflush(stdout);

    Timer T;
    const dim3 grid((unsigned)((ROWS + 255) / 256)), block(256);
    auto win = [&](int c, int half) { return (float *)(va + (size_t)c * CHUNK + (half ? CHUNK / 2 : 0)); };
    auto pw = [&](int i, int j) { return T.med([&] { w2<<<grid, block>>>(win(i, 0), win(j, i == j), ROWS); }); };
    auto pm = [&](int i, int j) { return T.med([&] { m2<<<grid, block>>>(d, mk, nin, win(i, 0), win(j, i == j), ROWS); }); };

    std::vector<float> s(NC);
    double t3 = now();
    for (int j = 0; j < NC; ++j) s[j] = j == 0 ? -1.f : pw(0, j);
    double t4 = now();
    show("w2 (write-only pair), reference chunk 0:", s);
    printf("   sweep took %.3f s\n", t4 - t3);
    std::vector<float> sm(NC);
    for (int j = 0; j < NC; ++j) sm[j] = j == 0 ? -1.f : pm(0, j);
    show("m2 (reads + pair), reference chunk 0:", sm);
    // the first clearly fast chunk becomes the second reference
    float lo = 1e9, hi = 0; for (int j = 1; j < NC; ++j) { lo = std::min(lo, sm[j]); hi = std::max(hi, sm[j]); }
    int r1 = -1; for (int j = 1; j < NC; ++j) if (sm[j] < lo + 0.25f * (hi - lo)) { r1 = j; break; }
    printf("second reference: chunk %d\n", r1);
    if (r1 > 0) {
        for (int j = 0; j < NC; ++j) s[j] = j == r1 ? -1.f : pw(r1, j);
        show("w2, reference = second:", s);
        for (int j = 0; j < NC; ++j) sm[j] = j == r1 ? -1.f : pm(r1, j);
        show("m2, reference = second:", sm);
    }
    // same-chunk reference levels (both windows inside one chunk)
    printf("same-chunk pairs (w2 / m2): ");
    for (int c : {0, NC / 4, NC / 2, 3 * NC / 4, NC - 1}) printf(" c%d %.4f/%.4f", c, pw(c, c), pm(c, c));
    printf("\n");
    // pairwise table over every 4th chunk (w2)
    {
        std::vector<int> ids; for (int i = 0; i < NC; i += 4) ids.push_back(i);
        const int n = (int)ids.size();
        std::vector<float> Tm((size_t)n * n);
        float lo2 = 1e9, hi2 = 0;
        for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) { const float t = T.med([&] { w2<<<grid, block>>>(win(ids[a], 0), win(ids[b], ids[a] == ids[b]), ROWS); }, 3); Tm[(size_t)a * n + b] = t; lo2 = std::min(lo2, t); hi2 = std::max(hi2, t); }
        printf("w2 pairwise, every 4th chunk (rows: first stream's chunk), min %.4f max %.4f ms\n", lo2, hi2);
        for (int a = 0; a < n; ++a) { printf("   c%03d ", ids[a]); for (int b = 0; b < n; ++b) putchar('0' + std::min(9, (int)(10 * (Tm[(size_t)a * n + b] - lo2) / (hi2 - lo2)))); printf("\n"); }
    }
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(va, total));
    for (int i = 0; i < NC; ++i) CK(hipMemRelease(hs[i]));
    CK(hipMemAddressFree(va, total));
    CK(hipFree(d)); CK(hipFree(mk)); CK(hipFree(nin));
    printf("done\n");
    return 0;
}
