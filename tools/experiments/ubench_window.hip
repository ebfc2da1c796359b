// Does the width of the chip-wide write window matter?  (GPU box only)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_window tools/experiments/ubench_window.hip && tools/experiments/ubench_window
// Every workgroup of T threads writes one contiguous tile of T*R rows of 12 B as R sweeps of dwordx3 row stores;
// `lds` bytes of dynamic LDS cap the workgroups per CU.  Rows in flight per CU = waves/CU * 64 * R: the tile kernel
// (compact_lean<single-pass>) sits at T=768, R=16, 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f3 __attribute__((ext_vector_type(3)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int T>
__global__ __launch_bounds__(T) void sweep(float *out, int R, size_t rows) {
    extern __shared__ char smem[];
    const size_t r0 = (size_t)blockIdx.x * T * R;
    for (int i = 0; i < R; ++i) {
        const size_t r = r0 + (size_t)i * T + threadIdx.x;
        if (r < rows) { f3 v = {1.0f, (float)i, (float)threadIdx.x}; *reinterpret_cast<f3 *>(out + r * 3) = v; }
    }
}
// transposed order inside the tile: thread owns R consecutive rows (what a lane would write WITHOUT the LDS list)
template <int T>
__global__ __launch_bounds__(T) void sweep_t(float *out, int R, size_t rows) {
    const size_t r0 = (size_t)blockIdx.x * T * R + (size_t)threadIdx.x * R;
    for (int i = 0; i < R; ++i) {
        const size_t r = r0 + i;
        if (r < rows) { f3 v = {1.0f, (float)i, (float)threadIdx.x}; *reinterpret_cast<f3 *>(out + r * 3) = v; }
    }
}

// the same sweeps as aligned 16-byte stores (T*R*16 B per workgroup)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int T>
__global__ __launch_bounds__(T) void sweep4(f4 *out, int R, size_t n4) {
    extern __shared__ char smem[];
    const size_t r0 = (size_t)blockIdx.x * T * R;
    for (int i = 0; i < R; ++i) {
        const size_t r = r0 + (size_t)i * T + threadIdx.x;
        if (r < n4) { f4 v = {1.0f, (float)i, (float)threadIdx.x, 2.0f}; out[r] = v; }
    }
}
// each WAVE owns a contiguous stretch of R KiB (instead of the workgroup's waves interleaving)
template <int T>
__global__ __launch_bounds__(T) void sweep4_wave(f4 *out, int R, size_t n4) {
    const size_t r0 = ((size_t)blockIdx.x * (T / 64) + (threadIdx.x >> 6)) * 64 * R;
    for (int i = 0; i < R; ++i) {
        const size_t r = r0 + (size_t)i * 64 + (threadIdx.x & 63);
        if (r < n4) { f4 v = {1.0f, (float)i, (float)threadIdx.x, 2.0f}; out[r] = v; }
    }
}

template <typename F> void run(const char *name, double bytes, F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); const int it = 6; for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s %8.1f GB/s  (%.3f ms)\n", name, bytes * it / (ms * 1e-3) / 1e9, ms / it); }

int main() {
    const size_t rows = (size_t)24000 * 12288;          // 3.5 GB
    void *o; CK(hipMalloc(&o, rows * 12 + 4096)); CK(hipMemset(o, 0, rows * 12 + 4096));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep<768>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    char name[128];
    for (int rep = 0; rep < 2; ++rep) {
        for (int R : {1, 16}) {
            const unsigned g256 = (unsigned)((rows + 256 * (size_t)R - 1) / (256 * (size_t)R));
            snprintf(name, sizeof name, "T=256 R=%-2d 8 wg/CU (32 waves)", R);
            run(name, rows * 12.0, [&] { sweep<256><<<g256, 256, 0>>>((float *)o, R, rows); });
            snprintf(name, sizeof name, "T=256 R=%-2d 6 wg/CU (24 waves, 26 KiB LDS)", R);
            run(name, rows * 12.0, [&] { sweep<256><<<g256, 256, 26 * 1024>>>((float *)o, R, rows); });
            const unsigned g768 = (unsigned)((rows + 768 * (size_t)R - 1) / (768 * (size_t)R));
            snprintf(name, sizeof name, "T=768 R=%-2d 2 wg/CU (24 waves, 72 KiB LDS)", R);
            run(name, rows * 12.0, [&] { sweep<768><<<g768, 768, 72 * 1024>>>((float *)o, R, rows); });
            const unsigned g1024 = (unsigned)((rows + 1024 * (size_t)R - 1) / (1024 * (size_t)R));
            snprintf(name, sizeof name, "T=1024 R=%-2d 2 wg/CU (32 waves, 72 KiB LDS)", R);
            run(name, rows * 12.0, [&] { sweep<1024><<<g1024, 1024, 72 * 1024>>>((float *)o, R, rows); });
        }
        const size_t n4 = rows * 12 / 16;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep4<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep4<768>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        for (int R : {1, 2, 3, 4, 8, 16}) {
            const unsigned g256 = (unsigned)((n4 + 256 * (size_t)R - 1) / (256 * (size_t)R));
            snprintf(name, sizeof name, "x4 T=256 R=%-2d 8 wg/CU", R);
            run(name, rows * 12.0, [&] { sweep4<256><<<g256, 256, 0>>>((f4 *)o, R, n4); });
            snprintf(name, sizeof name, "x4 T=256 R=%-2d 6 wg/CU (26 KiB LDS)", R);
            run(name, rows * 12.0, [&] { sweep4<256><<<g256, 256, 26 * 1024>>>((f4 *)o, R, n4); });
            snprintf(name, sizeof name, "x4 T=256 R=%-2d 2 wg/CU (72 KiB LDS)", R);
            run(name, rows * 12.0, [&] { sweep4<256><<<g256, 256, 72 * 1024>>>((f4 *)o, R, n4); });
            const unsigned g768 = (unsigned)((n4 + 768 * (size_t)R - 1) / (768 * (size_t)R));
            snprintf(name, sizeof name, "x4 T=768 R=%-2d 2 wg/CU (72 KiB LDS)", R);
            run(name, rows * 12.0, [&] { sweep4<768><<<g768, 768, 72 * 1024>>>((f4 *)o, R, n4); });
            snprintf(name, sizeof name, "x4 T=256 R=%-2d wave-contiguous 8 wg/CU", R);
            run(name, rows * 12.0, [&] { sweep4_wave<256><<<g256, 256, 0>>>((f4 *)o, R, n4); });
        }
        for (int R : {4, 16}) {
            const unsigned g256 = (unsigned)((rows + 256 * (size_t)R - 1) / (256 * (size_t)R));
            snprintf(name, sizeof name, "T=256 R=%-2d lane-owns-R-rows (no LDS list)", R);
            run(name, rows * 12.0, [&] { sweep_t<256><<<g256, 256, 0>>>((float *)o, R, rows); });
        }
    }
    return 0;
}
