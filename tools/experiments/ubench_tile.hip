// Store-shape micro-benchmark in the scatter kernel's own launch geometry (GPU box only):
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_tile tools/experiments/ubench_tile.hip && tools/experiments/ubench_tile
// One 768-thread workgroup per tile of 12288 output rows of 12 B (2 workgroups per CU through a 72 KiB LDS
// allocation, like compact_lean<single-pass,12 waves>), 16 iterations, every wave writes a run of 64 rows
// (768 B) per iteration.  Question: do 16-byte aligned dwordx4 stores of the same bytes
// beat the 12-byte row stores in THIS geometry, with and without the f16 depth read (2 B per row) in front?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NW = 12, BT = 64 * NW, ROWS = 12288, NI = ROWS / BT;

// MODE 0: one dwordx3 store per lane (row = lane)            -- what compact_lean does today
// MODE 1: lanes 0..46/47 store aligned dwordx4 chunks of the wave's run, lane 0 / 63 the run's ragged ends
//         (tile start misaligned by `f` dwords, f = tile % 4: the in-register re-shape, no LDS)
// MODE 2: the tile's bytes as 1024-B runs of 64 aligned dwordx4 stores (full re-shape through LDS, f = 0)
template <int MODE, bool READ>
__global__ __launch_bounds__(BT) void tile_store(const unsigned short *depth, float *out, int ntiles, int shift, int pad_rows = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t t = blockIdx.x;
    float s = 1.0f;
    if (READ) {   // 16 rows per lane: two 16-B loads of f16 depth
        const u4 *p = reinterpret_cast<const u4 *>(depth + t * ROWS) + tid;
        const u4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + BT);
        s = __uint_as_float((a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w) & 0x3fffffffu);
        reinterpret_cast<float *>(smem)[tid] = s;     // keep a dependency like the LDS list
        __syncthreads();
        s = reinterpret_cast<float *>(smem)[tid ^ 1];
    }
    // first row of the tile: MODE 1 shifts tiles by `shift` rows in total so that starts are not 16-B aligned
    const size_t row0 = t * (size_t)(ROWS + pad_rows) + (MODE == 1 ? (size_t)shift : 0);   // pad_rows: tile pitch off the 144 KiB grid
    float *base = out + row0 * 3;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = i * BT + tid;
            f3 v = {s, s + 1.0f, (float)r};
            *reinterpret_cast<f3 *>(base + (size_t)r * 3) = v;
        }
    } else if (MODE == 1) {
        const unsigned f = (4u - ((unsigned)(row0 * 3) & 3u)) & 3u;     // dwords up to the first 16-B boundary of every run
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float *run = base + (size_t)(i * BT + wave * 64) * 3;
            const int nfull = f ? 47 : 48;
            if (lane < nfull) {
                f4 v = {s, s + 1.0f, (float)i, (float)lane};
                *reinterpret_cast<f4 *>(run + f + lane * 4) = v;
            }
            if (f) {
                if (lane == 0) { for (unsigned k = 0; k < f; ++k) run[k] = s; }
                if (lane == 63) { for (unsigned k = f + 188; k < 192; ++k) run[k] = s; }
            }
        }
    } else {
        constexpr int NI4 = ROWS * 12 / (BT * 16);   // 12
#pragma unroll
        for (int i = 0; i < NI4; ++i) {
            f4 v = {s, s + 1.0f, (float)i, (float)lane};
            *reinterpret_cast<f4 *>(base + (size_t)(i * BT + tid) * 4) = v;
        }
    }
}

template <typename F> double run(const char *name, double bytes, F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); const int it = 8; for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double gbps = bytes * it / (ms * 1e-3) / 1e9; printf("%-58s %8.1f GB/s  (%.3f ms)\n", name, gbps, ms / it); return gbps; }

int main() {
    const int ntiles = 24000;                       // 295 M rows = 3.5 GB of output, 0.59 GB of f16 depth
    const size_t rows = (size_t)ntiles * ROWS;
    void *d, *o; CK(hipMalloc(&d, rows * 2 + 256)); CK(hipMalloc(&o, rows * 12 + 4096 + (size_t)ntiles * 12 * 1024)); CK(hipMemset(d, 1, rows * 2)); CK(hipMemset(o, 0, rows * 12 + 4096));
    const size_t lds = 72 * 1024;
    dim3 g(ntiles), b(BT);
#define SETLDS(K) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    SETLDS((tile_store<0, false>)); SETLDS((tile_store<0, true>)); SETLDS((tile_store<1, false>)); SETLDS((tile_store<1, true>));
    SETLDS((tile_store<2, false>)); SETLDS((tile_store<2, true>));
    for (int rep = 0; rep < 2; ++rep) {
        run("x3 row stores, write only", rows * 12.0, [&] { tile_store<0, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0); });
        run("x4 in-register re-shape f=0, write only", rows * 12.0, [&] { tile_store<1, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0); });
        run("x4 in-register re-shape f=1 (+1 row), write only", rows * 12.0, [&] { tile_store<1, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 1); });
        run("x4 in-register re-shape f=2 (+2 rows), write only", rows * 12.0, [&] { tile_store<1, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 2); });
        for (int pad : {4, 20, 84, 340, 1024}) { char nm[96]; snprintf(nm, sizeof nm, "x3 row stores, tile pitch +%d rows, write only", pad);
            run(nm, rows * 12.0, [&] { tile_store<0, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0, pad); });
            snprintf(nm, sizeof nm, "x4 1024-B runs, tile pitch +%d rows, write only", pad);
            run(nm, rows * 12.0, [&] { tile_store<2, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0, pad); }); }
        run("x4 full 1024-B runs, write only", rows * 12.0, [&] { tile_store<2, false><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0); });
        run("x3 row stores + f16 depth read", rows * 14.0, [&] { tile_store<0, true><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0); });
        run("x4 in-register re-shape f=0 + f16 depth read", rows * 14.0, [&] { tile_store<1, true><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0); });
        run("x4 in-register re-shape f=1 + f16 depth read", rows * 14.0, [&] { tile_store<1, true><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 1); });
        run("x4 full 1024-B runs + f16 depth read", rows * 14.0, [&] { tile_store<2, true><<<g, b, lds>>>((const unsigned short *)d, (float *)o, ntiles, 0); });
    }
    return 0;
}
