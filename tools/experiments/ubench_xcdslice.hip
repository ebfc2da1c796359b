// Is the one-shot fill fast because every XCD then writes a FIXED eighth of the 4 KiB blocks?  (GPU box only)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_xcdslice tools/experiments/ubench_xcdslice.hip && tools/experiments/ubench_xcdslice
// Workgroups are dealt round-robin to the 8 XCDs, so in a one-shot fill with 4 KiB per workgroup XCD x writes exactly
// the blocks b with b % 8 == x (round 2's permutations -- chunk = i + 1, i ^ 1, xcc + 3 ... -- all kept that property,
// which is why none of them mattered).  ubench_oneshot.hip: 4 KiB per workgroup 6.7-7.0 TB/s whatever the workgroup
// size, 8 KiB 6.4-6.7, 16 KiB 6.3-6.7, 32 KiB 6.0, every multiple of 3 KiB 5.8-6.1, every persistent form 5.5.
// Test: (1) one-shot with the residue scrambled per XCD; (2) a persistent, ticket-driven grid whose workgroups read
// their XCC id and take only blocks of their own residue class; (3) which address bits form the class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

// one-shot, 256 threads, one 4 KiB block per workgroup.  MODE 0: block = workgroup; 1: residue scrambled (block % 8 =
// (i ^ (i >> 3)) % 8: every XCD writes all eight classes); 2: class = bits [13..15] of the address (8 KiB units);
// 3: class = bits [14..16] (16 KiB units); 4: 1 KiB pieces, class = bits [10..12]; 5: 2 KiB pieces, class = bits [11..13]
template <int MODE, bool READ>
__global__ __launch_bounds__(256) void shot(f4 *buf, unsigned nblocks, float *sink) {
    const unsigned i = blockIdx.x, g = i >> 3, x = i & 7u;
    size_t e;                                   // element (16 B) index this thread touches
    if (MODE == 0) e = (size_t)i * 256 + threadIdx.x;
    if (MODE == 1) e = (size_t)((i & ~7u) | ((i ^ (i >> 3)) & 7u)) * 256 + threadIdx.x;
    if (MODE == 2) e = (size_t)(((g >> 1) << 4) | (x << 1) | (g & 1u)) * 256 + threadIdx.x;
    if (MODE == 3) e = (size_t)(((g >> 2) << 5) | (x << 2) | (g & 3u)) * 256 + threadIdx.x;
    if (MODE == 4) { const unsigned w = threadIdx.x >> 6; e = (size_t)(((4u * g + w) << 3) | x) * 64 + (threadIdx.x & 63u); }
    if (MODE == 5) { const unsigned h = threadIdx.x >> 7; e = (size_t)(((2u * g + h) << 3) | x) * 128 + (threadIdx.x & 127u); }
    if (e >= (size_t)nblocks * 256) return;
    if (READ) { const f4 v = buf[e]; if (v.x == 123.456f) sink[0] = v.y; }
    else buf[e] = f4{1.0f, (float)i, (float)threadIdx.x, 2.0f};
}

// persistent grid; a ticket is a run of 16 * NB blocks (one counter serves ~88 M tickets/s: a ticket must be worth >= 64 KiB).
// AFFINE: one counter per XCC id, ticket t of XCC x = blocks (16 NB t + j) * 8 + x: only blocks of the XCD's own class;
// SCRAMBLE: one counter, blocks of all classes in an order that changes from ticket to ticket; else plain ticket order
// (16 NB consecutive blocks).  A workgroup of T threads writes NB = T / 256 blocks per sweep.
template <int T, bool AFFINE, bool SCRAMBLE>
__global__ __launch_bounds__(T) void persistent(f4 *out, unsigned nblocks, unsigned *tickets) {
    __shared__ unsigned s_t;
    constexpr unsigned NB = T / 256, RUN = 16 * NB;
    const unsigned x = xcc_id();
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(&tickets[AFFINE ? x * 32 : 0], 1u);      // counters on separate lines
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if ((size_t)t * RUN * (AFFINE ? 8u : 1u) >= nblocks) break;
#pragma unroll 4
        for (unsigned j = 0; j < 16; ++j) {
            const unsigned k = t * RUN + j * NB + threadIdx.x / 256;        // block inside this counter's sequence
            unsigned b;
            if (AFFINE) b = k * 8 + x;
            else if (SCRAMBLE) b = (k & ~7u) | ((k + (k >> 3) + x) & 7u);
            else b = k;
            if (b < nblocks) out[(size_t)b * 256 + (threadIdx.x & 255u)] = f4{1.f, (float)t, (float)threadIdx.x, 4.f};
        }
    }
}

// persistent, own class, TIGHT front: a workgroup draws a slot number inside its XCD once (atomic), then in round r writes
// own-class block r * nslots + slot -- all resident workgroups of an XCD write consecutive blocks of its class in every
// round: the one-shot fill's pattern without the dispatcher.  STRIDE = 8: own class; STRIDE = 1: the XCD's workgroups
// write a contiguous eighth of the array (all classes).
template <int T, int STRIDE>
__global__ __launch_bounds__(T) void slotted(f4 *out, unsigned nblocks, unsigned *counters, unsigned nslots) {
    __shared__ unsigned s_slot;
    constexpr unsigned NB = T / 256;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) s_slot = atomicAdd(&counters[x * 32], 1u);
    __syncthreads();
    const unsigned slot = s_slot;                       // < nslots if the grid is spread evenly (checked by the host through the counters)
    const unsigned per_class = nblocks / 8;
    for (unsigned r = 0;; ++r) {
        const unsigned k = (r * nslots + slot) * NB + threadIdx.x / 256;     // block inside the XCD's sequence
        if ((r * nslots) * NB >= per_class) break;
        if (k < per_class) {
            const unsigned b = STRIDE == 8 ? k * 8 + x : x * per_class + k;
            out[(size_t)b * 256 + (threadIdx.x & 255u)] = f4{1.f, (float)r, (float)threadIdx.x, 4.f};
        }
    }
}

// The densify kernel's shape with tiles dealt to XCDs in CHUNKS: workgroup slot s of XCD x writes, in round r, tile
// (r * 8 + x) * nslots + s as S sweeps of T * 16 B -- the nslots workgroups of an XCD cover nslots consecutive tiles (one
// contiguous region per XCD and round), each workgroup sweeping its own tile.  CHUNKED = false: tile = r * G + (slot * 8 + x),
// i.e. consecutive tiles on different XCDs (what ticket order on a round-robin dispatch gives today).
// MODE 0: consecutive tiles on different XCDs, tile % 8 == XCC id; 1: CHUNKED; 2: tile % 8 rotates against the XCC id with the slot;
// 3: tile = blockIdx.x + r * G (whatever XCD the dispatcher put the workgroup on; the histogram is printed); 4: tiles by a global ticket
template <int T, int MODE>
__global__ __launch_bounds__(T) void tiles(f4 *out, size_t n16, int S, unsigned *counters, unsigned nslots) {
    extern __shared__ char smem[];
    __shared__ unsigned s_slot;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) s_slot = atomicAdd(&counters[x * 32], 1u);
    __syncthreads();
    const unsigned slot = s_slot;
    const size_t tile16 = (size_t)S * T, ntiles = n16 / tile16;
    if (MODE == 3 && threadIdx.x == 0) atomicAdd(&counters[512 + (blockIdx.x & 7u) * 8 + x], 1u);
    for (unsigned r = 0;; ++r) {
        size_t t;
        if (MODE == 0) t = (size_t)r * 8 * nslots + (size_t)slot * 8 + x;
        if (MODE == 1) t = ((size_t)r * 8 + x) * nslots + slot;
        if (MODE == 2) t = (size_t)r * 8 * nslots + (size_t)slot * 8 + ((x + slot) & 7u);
        if (MODE == 3) t = (size_t)r * gridDim.x + blockIdx.x;
        if (MODE == 4) {
            __syncthreads();
            if (threadIdx.x == 0) s_slot = atomicAdd(&counters[256], 1u);
            __syncthreads();
            t = s_slot;
            if (t >= ntiles) break;
        }
        if ((size_t)r * 8 * nslots >= ntiles) break;
        if (t < ntiles)
            for (int k = 0; k < S; ++k) out[t * tile16 + (size_t)k * T + threadIdx.x] = f4{1.f, (float)r, (float)threadIdx.x, 4.f};
    }
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

int main() {
    const size_t BYTES = (size_t)3 << 30;
    const unsigned nblocks = (unsigned)(BYTES / 4096);
    void *o; float *sink; unsigned *tickets;
    CK(hipMalloc(&o, BYTES)); CK(hipMemset(o, 0, BYTES)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&tickets, 8 * 32 * 4));
    f4 *buf = (f4 *)o;
    const char *names[6] = {"block = workgroup (XCD x writes blocks b % 8 == x)", "residue scrambled: every XCD writes all 8 classes",
                            "class = address bits 13..15 (8 KiB units)", "class = address bits 14..16 (16 KiB units)",
                            "class = address bits 10..12 (1 KiB pieces)", "class = address bits 11..13 (2 KiB pieces)"};
    for (int rep = 0; rep < 2; ++rep) {
#define SHOT(M) { float ms = timeit([&] { shot<M, false><<<nblocks, 256>>>(buf, nblocks, sink); }); \
                  printf("one-shot write  %-54s %7.1f GB/s\n", names[M], BYTES / ms / 1e6); }
        SHOT(0) SHOT(1) SHOT(2) SHOT(3) SHOT(4) SHOT(5)
#define RD(M) { float ms = timeit([&] { shot<M, true><<<nblocks, 256>>>(buf, nblocks, sink); }); \
                printf("one-shot read   %-54s %7.1f GB/s\n", names[M], BYTES / ms / 1e6); }
        RD(0) RD(1) RD(2) RD(4)
    }
    for (int rep = 0; rep < 2; ++rep) {
        for (int wg : {2, 4, 8}) {
            const unsigned G = 256 * wg;
            float ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); persistent<256, true, false><<<G, 256>>>(buf, nblocks, tickets); });
            printf("persistent T=256 x %4u, blocks of the workgroup's own class (xcc)       %7.1f GB/s\n", G, BYTES / ms / 1e6);
            ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); persistent<256, false, false><<<G, 256>>>(buf, nblocks, tickets); });
            printf("persistent T=256 x %4u, blocks in ticket order                          %7.1f GB/s\n", G, BYTES / ms / 1e6);
            ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); persistent<256, false, true><<<G, 256>>>(buf, nblocks, tickets); });
            printf("persistent T=256 x %4u, residue scrambled                               %7.1f GB/s\n", G, BYTES / ms / 1e6);
        }
        for (int wg : {1, 2}) {
            const unsigned G = 256 * wg;
            float ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); persistent<768, true, false><<<G, 768>>>(buf, nblocks, tickets); });
            printf("persistent T=768 x %4u, 3 own-class blocks per sweep                     %7.1f GB/s\n", G, BYTES / ms / 1e6);
            ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); persistent<768, false, false><<<G, 768>>>(buf, nblocks, tickets); });
            printf("persistent T=768 x %4u, 3 consecutive blocks per sweep                  %7.1f GB/s\n", G, BYTES / ms / 1e6);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        for (int wg : {1, 2, 4, 8}) {
            const unsigned G = 256 * wg;
            float ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); slotted<256, 8><<<G, 256>>>(buf, nblocks, tickets, G / 8); });
            unsigned h[256]; CK(hipMemcpy(h, tickets, 1024, hipMemcpyDeviceToHost));
            printf("slotted T=256 x %4u own class, tight front (slots per XCD %u..%u of %u)   %7.1f GB/s\n", G, h[0] < h[224] ? h[0] : h[224], h[0] > h[224] ? h[0] : h[224], G / 8, BYTES / ms / 1e6);
            ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); slotted<256, 1><<<G, 256>>>(buf, nblocks, tickets, G / 8); });
            printf("slotted T=256 x %4u contiguous eighth per XCD (all classes)              %7.1f GB/s\n", G, BYTES / ms / 1e6);
        }
        for (int wg : {1, 2}) {
            const unsigned G = 256 * wg;
            float ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 1024)); slotted<768, 8><<<G, 768>>>(buf, nblocks, tickets, G / 8); });
            printf("slotted T=768 x %4u own class, tight front                               %7.1f GB/s\n", G, BYTES / ms / 1e6);
        }
    }
    CK(hipFree(tickets)); CK(hipMalloc(&tickets, 4096));
#define TATTR(T, M) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&tiles<T, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    TATTR(768, 0) TATTR(768, 1) TATTR(768, 2) TATTR(768, 3) TATTR(768, 4)
    const char *tn[5] = {"tile % 8 == XCC id", "64 consecutive tiles per XCD and round", "tile % 8 rotates against XCC id", "tile = blockIdx.x + r G", "tiles by global ticket"};
    for (int rep = 0; rep < 2; ++rep) {
        const unsigned G = 512; const int S = 12;
#define TRUN(M) { float ms = timeit([&] { CK(hipMemsetAsync(tickets, 0, 4096)); tiles<768, M><<<G, 768, 72 * 1024>>>(buf, BYTES / 16, S, tickets, G / 8); }); \
                  printf("tiles T=768 x 512 (2 per CU), tile 144 KiB, %-42s %7.1f GB/s\n", tn[M], BYTES / ms / 1e6); }
        TRUN(0) TRUN(1) TRUN(2) TRUN(3) TRUN(4)
        if (rep == 0) {
            unsigned h[1024]; CK(hipMemcpy(h, tickets, 4096, hipMemcpyDeviceToHost));
            printf("blockIdx %% 8 (rows) x XCC id (cols) of the 768-thread launch:\n");
            for (int r = 0; r < 8; ++r) { for (int c = 0; c < 8; ++c) printf("%6u", h[512 + r * 8 + c]); printf("\n"); }
        }
    }
    printf("done\n");
    return 0;
}
