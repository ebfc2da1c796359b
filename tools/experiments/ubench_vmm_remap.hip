// Does a kernel see the NEW physical memory after a virtual range was unmapped and mapped again?  (GPU box only)
//   hipcc --offload-arch=gfx950 -O2 -o tools/experiments/ubench_vmm_remap tools/experiments/ubench_vmm_remap.hip && tools/experiments/ubench_vmm_remap
// tests/c_client/arena_churn.cpp read the arena's probe pattern out of an array it had just filled, in the first round that reused
// a virtual range: the kernel's writes had gone somewhere else.  Here without the arena: handles h0, h1; map h0 at X, write A;
// unmap, map h1 at X, write B through X; then look at both handles through fresh ranges.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__global__ void fill(unsigned *p, size_t n, unsigned tag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = tag;
}
__global__ void peek(const unsigned *p, unsigned *out) { out[0] = p[5]; out[1] = p[(64u << 20) / 4 + 5]; }

static unsigned look_copy(void *va) { unsigned v = 0; CK(hipMemcpy(&v, (unsigned *)va + 5, 4, hipMemcpyDeviceToHost)); return v; }
static unsigned look_kernel(void *va, unsigned *scratch) {
    peek<<<1, 1>>>((const unsigned *)va, scratch);
    CK(hipDeviceSynchronize());
    unsigned v[2]; CK(hipMemcpy(v, scratch, 8, hipMemcpyDeviceToHost)); return v[0];
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;     // 0: unmap + map at the same reserved range; 1: address free + reserve again in between; 2: + hipDeviceSynchronize after the unmap
                                                       // 3: no virtual range is ever used twice: every mapping gets the next piece of ONE large reservation (a bump pointer)
    const size_t CH = (size_t)1 << 30, N = CH / 4;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned *scratch; CK(hipMalloc(&scratch, 64));
    int bad = 0;
    char *big = nullptr; size_t bump = 0;
    if (mode == 3) CK(hipMemAddressReserve((void **)&big, (size_t)128 << 30, 0, nullptr, 0));
    auto take = [&](char **out) { if (mode == 3) { *out = big + bump; bump += CH; } else CK(hipMemAddressReserve((void **)out, CH, 0, nullptr, 0)); };
    auto give = [&](char *va) { if (mode != 3) CK(hipMemAddressFree(va, CH)); };
    for (int rep = 0; rep < (mode == 3 ? 30 : 20); ++rep) {
        hipMemGenericAllocationHandle_t h0, h1;
        CK(hipMemCreate(&h0, CH, &prop, 0)); CK(hipMemCreate(&h1, CH, &prop, 0));
        char *X = nullptr, *Y = nullptr, *Z = nullptr;
        take(&X);
        CK(hipMemMap(X, CH, 0, h0, 0)); CK(hipMemSetAccess(X, CH, &acc, 1));
        fill<<<(unsigned)(N / 256), 256>>>((unsigned *)X, N, 0xA0000000u + rep);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(X, CH));
        if (mode >= 2) CK(hipDeviceSynchronize());
        char *X2 = X;
        if (mode == 1 || mode == 2) { CK(hipMemAddressFree(X, CH)); CK(hipMemAddressReserve((void **)&X2, CH, 0, nullptr, 0)); }
        if (mode == 3) take(&X2);
        CK(hipMemMap(X2, CH, 0, h1, 0)); CK(hipMemSetAccess(X2, CH, &acc, 1));
        fill<<<(unsigned)(N / 256), 256>>>((unsigned *)X2, N, 0xB0000000u + rep);
        CK(hipDeviceSynchronize());
        const unsigned viaX_copy = look_copy(X2), viaX_kernel = look_kernel(X2, scratch);
        // both handles through fresh ranges
        take(&Y); CK(hipMemMap(Y, CH, 0, h0, 0)); CK(hipMemSetAccess(Y, CH, &acc, 1));
        CK(hipMemUnmap(X2, CH));
        take(&Z); CK(hipMemMap(Z, CH, 0, h1, 0)); CK(hipMemSetAccess(Z, CH, &acc, 1));
        const unsigned h0_copy = look_copy(Y), h0_kernel = look_kernel(Y, scratch), h1_copy = look_copy(Z), h1_kernel = look_kernel(Z, scratch);
        const bool ok = viaX_copy == 0xB0000000u + rep && viaX_kernel == viaX_copy && h0_copy == 0xA0000000u + rep && h0_kernel == h0_copy && h1_copy == 0xB0000000u + rep && h1_kernel == h1_copy;
        if (!ok || rep < 2)
            printf("mode %d rep %2d  X %s  through X: copy %08x kernel %08x | h0: copy %08x kernel %08x (want a...) | h1: copy %08x kernel %08x (want b...)  %s\n",
                   mode, rep, X2 == X ? "same range" : "new range ", viaX_copy, viaX_kernel, h0_copy, h0_kernel, h1_copy, h1_kernel, ok ? "ok" : "WRONG");
        bad += !ok;
        CK(hipMemUnmap(Y, CH)); CK(hipMemUnmap(Z, CH));
        give(X2); give(Y); give(Z);
        CK(hipMemRelease(h0)); CK(hipMemRelease(h1));
    }
    printf("mode %d: %d of 20 repetitions wrong\n", mode, bad);
    return bad ? 1 : 0;
}
