// Does hipMemMap take an OFFSET into a physical allocation on this stack?  (GPU box only)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_vmm_offset tools/experiments/ubench_vmm_offset.hip && tools/experiments/ubench_vmm_offset
// If it does, an array striped between the classes of HBM below the arena's 1 GiB chunk is a matter of
// mapping pieces of three classified chunks in turn; if not, striping needs one physical handle per piece.
// Also: what creating and mapping many small handles costs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void stamp(unsigned *p, size_t n, unsigned tag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = tag + (unsigned)i;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("allocation granularity %zu bytes\n", gran);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    const size_t CH = (size_t)1 << 30, PIECE = (size_t)32 << 20, NP = CH / PIECE;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, CH, &prop, 0));
    char *va = nullptr, *vb = nullptr;
    CK(hipMemAddressReserve((void **)&va, CH, 0, nullptr, 0));
    CK(hipMemAddressReserve((void **)&vb, CH, 0, nullptr, 0));
    // (1) the whole handle at va: the reference picture
    CK(hipMemMap(va, CH, 0, h, 0));
    CK(hipMemSetAccess(va, CH, &acc, 1));
    stamp<<<(unsigned)(CH / 4 / 256), 256>>>((unsigned *)va, CH / 4, 7u);
    CK(hipDeviceSynchronize());
    // (2) its pieces in reverse order at vb, mapped with offsets
    bool ok = true;
    for (size_t k = 0; k < NP && ok; ++k) {
        hipError_t e = hipMemMap(vb + k * PIECE, PIECE, (NP - 1 - k) * PIECE, h, 0);
        if (e != hipSuccess) { printf("hipMemMap with offset %zu MiB: %s\n", ((NP - 1 - k) * PIECE) >> 20, hipGetErrorString(e)); (void)hipGetLastError(); ok = false; }
    }
    if (ok) {
        hipError_t e = hipMemSetAccess(vb, CH, &acc, 1);
        if (e != hipSuccess) { printf("hipMemSetAccess on the piecewise range: %s\n", hipGetErrorString(e)); ok = false; }
    }
    if (ok) {
        std::vector<unsigned> host(4);
        size_t bad = 0;
        for (size_t k = 0; k < NP; ++k) {
            CK(hipMemcpy(host.data(), vb + k * PIECE, 16, hipMemcpyDeviceToHost));
            const unsigned want = 7u + (unsigned)(((NP - 1 - k) * PIECE) / 4);
            if (host[0] != want) ++bad;
        }
        printf("offset mapping: accepted; %zu of %zu pieces show the wrong bytes -> %s\n", bad, NP, bad ? "NOT usable" : "usable");
    } else {
        printf("offset mapping: NOT supported\n");
    }
    // (3) many small handles: cost of create + map + set-access, and of unmap + release
    for (size_t piece : {(size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
        const size_t total = (size_t)8 << 30, n = total / piece;
        std::vector<hipMemGenericAllocationHandle_t> hs(n);
        char *v = nullptr;
        CK(hipMemAddressReserve((void **)&v, total, 0, nullptr, 0));
        double t0 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&hs[i], piece, &prop, 0));
        double t1 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemMap(v + i * piece, piece, 0, hs[i], 0));
        double t2 = now();
        CK(hipMemSetAccess(v, total, &acc, 1));
        double t3 = now();
        stamp<<<(unsigned)(total / 4 / 256), 256>>>((unsigned *)v, total / 4, 1u);
        CK(hipDeviceSynchronize());
        double t4 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemUnmap(v + i * piece, piece));
        double t5 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemRelease(hs[i]));
        double t6 = now();
        CK(hipMemAddressFree(v, total));
        printf("8 GiB in %5zu handles of %2zu MiB: create %.1f ms, map %.1f ms, set access %.1f ms, first touch %.1f ms, unmap %.1f ms, release %.1f ms\n",
               n, piece >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3);
    }
    // (4) how much address space can be reserved: 1 TiB ranges until the runtime says no (the arena never reuses addresses)
    {
        std::vector<void *> rs;
        for (int i = 0; i < 512; ++i) {
            void *r = nullptr;
            if (hipMemAddressReserve(&r, (size_t)1 << 40, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            rs.push_back(r);
        }
        printf("%zu reservations of 1 TiB succeeded (stopped at 512)\n", rs.size());
        for (void *r : rs) CK(hipMemAddressFree(r, (size_t)1 << 40));
    }
    printf("done\n");
    return 0;
}
