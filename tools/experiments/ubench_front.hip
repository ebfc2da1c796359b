// What caps a tiled store stream below the one-shot fill?  (GPU box only)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_front tools/experiments/ubench_front.hip && tools/experiments/ubench_front
//
// The densify kernel keeps 512 workgroups resident (2 per CU), each writing ITS OWN 147 KB tile over ~15 us in 16 sweeps of
// 9 KiB: at any instant the chip writes 512 pieces that lie 147 KB apart (a 75 MB "front").  A one-shot fill writes one
// dense 8 MB window.  This program writes the same bytes with the same resident workgroups and the same 1 KiB-per-wave
// aligned dwordx4 stores, and varies only WHICH piece a workgroup writes in sweep k:
//   tile    : workgroup w of round r writes tile (r G + w): piece k of its own tile              (the kernel today)
//   cohort C: C workgroups share C consecutive tiles; in sweep k member m writes piece (k C + m) of the C-tile run, so
//             the cohort's sweep is one contiguous run of C pieces (C = G: the whole chip sweeps one window)
// with `pace` ALU work between sweeps (the kernel computes ~1 us between the stores of a sweep) and optional jitter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// T threads; a piece = T * 16 B; a tile = S pieces; G resident workgroups (persistent, grid = G); C = cohort size (1 = own tile)
template <int K> __device__ __forceinline__ void cap_stores() {      // at most K vector-memory operations of this wave stay in flight
    if constexpr (K >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}

template <int T, int K = -1>
__global__ __launch_bounds__(T) void front(f4 *out, size_t n16, int S, int C, int pace, int jitter, float *sink) {
    extern __shared__ char smem[];
    const unsigned G = gridDim.x, w = blockIdx.x;
    const size_t tile16 = (size_t)S * T;
    const size_t tiles = n16 / tile16;
    const unsigned cohort = w / C, member = w % C, ncoh = G / C;
    float acc = (float)threadIdx.x;
    for (size_t round = 0;; ++round) {
        const size_t run0 = (round * ncoh + cohort) * (size_t)C;       // first tile of this cohort's run of C tiles
        if (run0 + C > tiles) break;
        for (int k = 0; k < S; ++k) {
            const size_t piece = run0 * S + (size_t)k * C + member;    // C == 1: run0 * S + k = piece k of the own tile
            f4 v = {acc, (float)k, 1.0f, 2.0f};
            out[piece * T + threadIdx.x] = v;
            cap_stores<K>();
            int spin = pace + (jitter ? (int)((w * 2654435761u + k * 40503u) >> 27) * jitter / 32 : 0);
            for (int i = 0; i < spin; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int T>
__global__ __launch_bounds__(T) void oneshot(f4 *out, size_t n16) {
    const size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    if (i < n16) out[i] = f4{1.f, 2.f, 3.f, 4.f};
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < 4; ++i) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)3 << 30;
    const size_t n16 = bytes / 16;
    void *o; float *sink;
    CK(hipMalloc(&o, bytes)); CK(hipMemset(o, 0, bytes)); CK(hipMalloc(&sink, 4));
#define ATTR(K) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&front<768, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&front<256, K>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    ATTR(-1) ATTR(0) ATTR(1) ATTR(3)
    {
        const float ms = timeit([&] { oneshot<256><<<(unsigned)(n16 / 256), 256>>>((f4 *)o, n16); });
        printf("one-shot fill, 256-thread workgroups                         %7.1f GB/s\n", bytes / ms / 1e6);
    }
    // 768-thread workgroups, 2 per CU (72 KiB LDS each): the kernel's geometry.  piece = 12 KiB, tile = 12 pieces = 144 KiB.
    // Round 1 of this experiment: cohort size (which piece a workgroup writes when) makes no difference, 5.47-5.61 TB/s for
    // C = 1 ... 512 -- even the exact linear sweep of the whole chip (C = 512) stays 20 % under the one-shot fill.  So it is
    // not the address pattern.  Round 2: is it the number of stores a wave keeps in flight?
    for (int C : {1, 512}) {
        const unsigned G = 512; const int S = 12;
        const size_t tile16 = (size_t)S * 768, tiles = n16 / tile16;
        const double wr = (double)(tiles / G * G) * tile16 * 16;
        float ms;
        ms = timeit([&] { front<768, -1><<<G, 768, 72 * 1024>>>((f4 *)o, n16, S, C, 0, 0, sink); });
        printf("T=768 2 wg/CU tile 144 KiB cohort %3d  stores in flight per wave: unlimited  %7.1f GB/s\n", C, wr / ms / 1e6);
        ms = timeit([&] { front<768, 3><<<G, 768, 72 * 1024>>>((f4 *)o, n16, S, C, 0, 0, sink); });
        printf("T=768 2 wg/CU tile 144 KiB cohort %3d  stores in flight per wave: <= 4       %7.1f GB/s\n", C, wr / ms / 1e6);
        ms = timeit([&] { front<768, 1><<<G, 768, 72 * 1024>>>((f4 *)o, n16, S, C, 0, 0, sink); });
        printf("T=768 2 wg/CU tile 144 KiB cohort %3d  stores in flight per wave: <= 2       %7.1f GB/s\n", C, wr / ms / 1e6);
        ms = timeit([&] { front<768, 0><<<G, 768, 72 * 1024>>>((f4 *)o, n16, S, C, 0, 0, sink); });
        printf("T=768 2 wg/CU tile 144 KiB cohort %3d  stores in flight per wave: 1          %7.1f GB/s\n", C, wr / ms / 1e6);
    }
    for (int wg_per_cu : {1, 2, 4, 8}) {
        const unsigned G = 256 * wg_per_cu; const int S = 36;
        const size_t lds = wg_per_cu == 8 ? 0 : wg_per_cu == 4 ? 36 * 1024 : wg_per_cu == 2 ? 72 * 1024 : 80 * 1024;
        const size_t tile16 = (size_t)S * 256, tiles = n16 / tile16;
        const double wr = (double)(tiles / G * G) * tile16 * 16;
        float ms;
        ms = timeit([&] { front<256, -1><<<G, 256, lds>>>((f4 *)o, n16, S, 1, 0, 0, sink); });
        printf("T=256 %d wg/CU tile 144 KiB  stores in flight per wave: unlimited  %7.1f GB/s\n", wg_per_cu, wr / ms / 1e6);
        ms = timeit([&] { front<256, 3><<<G, 256, lds>>>((f4 *)o, n16, S, 1, 0, 0, sink); });
        printf("T=256 %d wg/CU tile 144 KiB  stores in flight per wave: <= 4       %7.1f GB/s\n", wg_per_cu, wr / ms / 1e6);
        ms = timeit([&] { front<256, 1><<<G, 256, lds>>>((f4 *)o, n16, S, 1, 0, 0, sink); });
        printf("T=256 %d wg/CU tile 144 KiB  stores in flight per wave: <= 2       %7.1f GB/s\n", wg_per_cu, wr / ms / 1e6);
        ms = timeit([&] { front<256, 0><<<G, 256, lds>>>((f4 *)o, n16, S, 1, 0, 0, sink); });
        printf("T=256 %d wg/CU tile 144 KiB  stores in flight per wave: 1          %7.1f GB/s\n", wg_per_cu, wr / ms / 1e6);
    }
    printf("done\n");
    return 0;
}
