// Why does a one-shot grid of 256-thread workgroups, each storing ONE aligned 4 KiB chunk with dwordx4, write at 6.9 TB/s
// when every other store shape stays near 5.6?  Hypothesis: workgroups are dealt round-robin over the 8 XCDs and
// physical memory is interleaved over the HBM stacks in 4 KiB units, so workgroup i (XCD i % 8) writes the stack next
// to its XCD.  Test: permute which chunk a workgroup writes.  (GPU box only.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_affinity tools/experiments/ubench_affinity.hip && tools/experiments/ubench_affinity
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0 identity; 1 chunk = i+1; 2 chunk = i^1; 3 chunk = i+4 inside its group of 8; 4 same slot of the neighbour group (i^8);
// 5 XCD-matched: chunk % 8 == hardware XCC id (identity if dispatch is round-robin); 6 XCD-mismatched: chunk % 8 == (xcc+3) % 8
template <int MODE>
__global__ __launch_bounds__(256) void oneshot(f4 *out, unsigned nchunks, unsigned *hist) {
    const unsigned i = blockIdx.x;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    unsigned c = i;
    if (MODE == 1) c = (i + 1) % nchunks;
    if (MODE == 2) c = i ^ 1u;
    if (MODE == 3) c = (i & ~7u) | ((i + 4) & 7u);
    if (MODE == 4) c = i ^ 8u;
    if (MODE == 5) c = (i & ~7u) | (xcc & 7u);
    if (MODE == 6) c = (i & ~7u) | ((xcc + 3) & 7u);
    if (hist && threadIdx.x == 0) atomicAdd(&hist[(i & 7u) * 16 + xcc], 1u);
    if (c < nchunks) { f4 v = {1.0f, (float)i, (float)threadIdx.x, 2.0f}; out[(size_t)c * 256 + threadIdx.x] = v; }
}

template <typename F> void run(const char *name, double bytes, F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); const int it = 6; for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %8.1f GB/s  (%.3f ms)\n", name, bytes * it / (ms * 1e-3) / 1e9, ms / it); }

int main() {
    const size_t BYTES = (size_t)3 << 30;
    const unsigned nchunks = (unsigned)(BYTES / 4096);
    void *o; unsigned *hist; CK(hipMalloc(&o, BYTES + 65536)); CK(hipMemset(o, 0, BYTES)); CK(hipMalloc(&hist, 128 * 4)); CK(hipMemset(hist, 0, 512));
    oneshot<0><<<nchunks, 256>>>((f4 *)o, nchunks, hist); CK(hipDeviceSynchronize());
    unsigned h[128]; CK(hipMemcpy(h, hist, 512, hipMemcpyDeviceToHost));
    printf("blockIdx %% 8 (rows) x XCC id (cols) histogram:\n");
    for (int r = 0; r < 8; ++r) { for (int c = 0; c < 8; ++c) printf("%8u", h[r * 16 + c]); printf("\n"); }
    for (int rep = 0; rep < 2; ++rep) {
        run("identity (chunk = workgroup)", BYTES, [&] { oneshot<0><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
        run("chunk = i + 1", BYTES, [&] { oneshot<1><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
        run("chunk = i ^ 1", BYTES, [&] { oneshot<2><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
        run("chunk = i + 4 within its group of 8", BYTES, [&] { oneshot<3><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
        run("chunk = i ^ 8 (same slot, neighbour group)", BYTES, [&] { oneshot<4><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
        run("chunk % 8 = XCC id", BYTES, [&] { oneshot<5><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
        run("chunk % 8 = XCC id + 3", BYTES, [&] { oneshot<6><<<nchunks, 256>>>((f4 *)o, nchunks, nullptr); });
    }
    return 0;
}
