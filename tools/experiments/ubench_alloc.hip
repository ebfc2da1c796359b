// Does the physical placement the driver picks for an allocation change streaming bandwidth, and does the virtual-memory
// API (one physical handle per buffer / per 1 GiB / per 2 MiB) behave differently from hipMalloc?  (GPU box only.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_alloc tools/experiments/ubench_alloc.hip && tools/experiments/ubench_alloc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void copy4(const f4 *a, f4 *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) o[i] = a[i]; }
// a mixed stream like the scatter kernel's: read 5 B (as 4+1), read 12, write 12+12 per element
__global__ void mixed(const float *d, const unsigned char *m, const float *nin, float *x, float *nout, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dd = d[i] + (float)m[i];
    const f3 nn = *reinterpret_cast<const f3 *>(nin + 3 * i);
    f3 p = {dd, dd * 2.0f, dd * 3.0f};
    *reinterpret_cast<f3 *>(x + 3 * i) = p;
    *reinterpret_cast<f3 *>(nout + 3 * i) = nn;
}

static double time_copy(void *a, void *b, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n = bytes / 16;
    for (int i = 0; i < 2; ++i) copy4<<<dim3((unsigned)((n + 255) / 256)), 256>>>((const f4 *)a, (f4 *)b, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 6; ++i) copy4<<<dim3((unsigned)((n + 255) / 256)), 256>>>((const f4 *)a, (f4 *)b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * bytes * 6 / (ms * 1e-3) / 1e9;
}
static double time_mixed(char *base, size_t n) {     // carve d, m, nin, x, nout from one region
    float *d = (float *)base; unsigned char *m = (unsigned char *)(base + n * 4); float *nin = (float *)(base + n * 5 + 256 - (n * 5) % 256);
    float *x = nin + 3 * n, *nout = x + 3 * n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) mixed<<<dim3((unsigned)((n + 255) / 256)), 256>>>(d, m, nin, x, nout, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 6; ++i) mixed<<<dim3((unsigned)((n + 255) / 256)), 256>>>(d, m, nin, x, nout, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 41.0 * n * 6 / (ms * 1e-3) / 1e9;
}

static void *vmm_alloc(size_t bytes, size_t chunk, std::vector<hipMemGenericAllocationHandle_t> &hs) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    bytes = (bytes + chunk - 1) / chunk * chunk;
    void *va; CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
    for (size_t off = 0; off < bytes; off += chunk) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)va + off, chunk, 0, h, 0));
        hs.push_back(h);
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    return va;
}
static void vmm_free(void *va, size_t bytes, size_t chunk, std::vector<hipMemGenericAllocationHandle_t> &hs) {
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk < gran) chunk = gran; chunk = (chunk + gran - 1) / gran * gran; bytes = (bytes + chunk - 1) / chunk * chunk;
    CK(hipMemUnmap(va, bytes));
    for (auto h : hs) CK(hipMemRelease(h));
    hs.clear();
    CK(hipMemAddressFree(va, bytes));
}

int main() {
    const size_t BYTES = (size_t)4 << 30;                 // per copy buffer
    const size_t NMIX = 200u * 1000 * 1000;               // mixed stream: 41 B per element = 8.2 GB region
    const size_t MIXB = NMIX * 41 + 4096;
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0; hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum); hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("VMM granularity: minimum %zu, recommended %zu\n", gmin, grec);
    std::vector<void *> ballast;
    for (int r = 0; r < 6; ++r) {
        void *a, *b, *m; CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&b, BYTES)); CK(hipMalloc(&m, MIXB));
        CK(hipMemset(a, 1, BYTES)); CK(hipMemset(m, 0, MIXB));
        printf("hipMalloc round %d: copy %7.1f GB/s   mixed %7.1f GB/s\n", r, time_copy(a, b, BYTES), time_mixed((char *)m, NMIX));
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(m));
        void *x; CK(hipMalloc(&x, (size_t)(r + 1) * 700 * 1000 * 1000)); ballast.push_back(x);     // perturb the next placement
    }
    for (void *x : ballast) CK(hipFree(x));
    for (size_t chunk : {(size_t)0, (size_t)64 << 10, (size_t)512 << 10, (size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        for (int r = 0; r < 4; ++r) {
            std::vector<hipMemGenericAllocationHandle_t> ha, hb, hm;
            const size_t ca = chunk ? chunk : BYTES, cm = chunk ? chunk : MIXB;
            void *a = vmm_alloc(BYTES, ca, ha), *b = vmm_alloc(BYTES, ca, hb), *m = vmm_alloc(MIXB, cm, hm);
            CK(hipMemset(a, 1, BYTES)); CK(hipMemset(m, 0, MIXB));
            char nm[32]; if (chunk == 0) snprintf(nm, sizeof nm, "whole buffer"); else snprintf(nm, sizeof nm, "%zu KiB", chunk >> 10);
            printf("VMM handle size %-12s round %d: copy %7.1f GB/s   mixed %7.1f GB/s\n", nm, r, time_copy(a, b, BYTES), time_mixed((char *)m, NMIX));
            fflush(stdout);
            vmm_free(a, BYTES, ca, ha); vmm_free(b, BYTES, ca, hb); vmm_free(m, MIXB, cm, hm);
        }
    }
    return 0;
}
