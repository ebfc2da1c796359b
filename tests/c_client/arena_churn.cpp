// Churn of the zone arena without Python (GPU box): arrays of one chunk allocated, touched and freed over and over, what a
// caller that builds a cloud per scene does -- interleaved with ordinary hipMalloc / hipFree traffic like torch's allocator makes.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tests/c_client/arena_churn.cpp -Ldepthdensifier_amd -lddcore -Wl,-rpath,$PWD/depthdensifier_amd -o /tmp/arena_churn
//   DD_ARENA_TRACE=1 /tmp/arena_churn 3000 2> trace.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ddcore.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

__global__ void touch(unsigned *p, size_t n, unsigned tag) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = tag + (unsigned)i;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 1000;
    const int plain = argc > 2 ? atoi(argv[2]) : 1;          // 1: hipMalloc / hipFree of assorted sizes between the rounds
    DDArena *arena = nullptr;
    if (dd_arena_create(0, 0, &arena) != DD_OK) { printf("dd_arena_create: %s\n", dd_arena_last_error()); return 1; }
    std::vector<void *> side;
    for (int r = 0; r < rounds; ++r) {
        int64_t sizes[3] = {(int64_t)(12 + r % 7) << 20, (int64_t)(12 + r % 5) << 20, (int64_t)(3 + r % 3) << 20};
        if (r % 5 == 4) sizes[0] = ((int64_t)1 << 30) + ((int64_t)5 << 20);       // two chunks: another shape, the mapping cache misses and is emptied
        int32_t layouts[3] = {DD_ARENA_ROTATED, DD_ARENA_ROTATED + 1, DD_ARENA_ROTATED + 2};
        void *p[3] = {nullptr, nullptr, nullptr};
        const int rc = dd_arena_alloc(arena, 3, sizes, layouts, (int64_t)16 << 30, p);
        if (rc < 0) { printf("round %d: dd_arena_alloc: %s\n", r, dd_arena_last_error()); return 1; }
        for (int k = 0; k < 3; ++k) {
            const size_t n = (size_t)sizes[k] / 4;
            touch<<<(unsigned)((n + 255) / 256), 256>>>((unsigned *)p[k], n, (unsigned)r);
        }
        CK(hipGetLastError());
        if (plain) {
            void *q = nullptr;
            CK(hipMalloc(&q, (size_t)(1 + r % 9) << 20));
            side.push_back(q);
            if (side.size() > 6) { CK(hipFree(side.front())); side.erase(side.begin()); }
        }
        unsigned back = 0;
        CK(hipMemcpy(&back, (unsigned *)p[r % 3] + 5, 4, hipMemcpyDeviceToHost));
        if (back != (unsigned)r + 5u) { printf("round %d: array %d reads %u\n", r, r % 3, back); return 1; }
        for (int k = 0; k < 3; ++k)
            if (dd_arena_free(arena, p[(k + r) % 3]) != DD_OK) { printf("round %d: dd_arena_free: %s\n", r, dd_arena_last_error()); return 1; }
        if (r % 500 == 499) { printf("%d rounds\n", r + 1); fflush(stdout); }
    }
    DDArenaStats st;
    dd_arena_stats(arena, &st);
    printf("arena churn OK: %d rounds, %lld chunks created, %lld released\n", rounds, (long long)st.chunks_created, (long long)st.chunks_released);
    for (void *q : side) CK(hipFree(q));
    if (dd_arena_trim(arena, 0) != DD_OK || dd_arena_destroy(arena) != DD_OK) { printf("teardown failed\n"); return 1; }
    return 0;
}
