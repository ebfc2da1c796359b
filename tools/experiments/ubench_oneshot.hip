// Which property of the one-shot fill makes it 20-25 % faster than every tiled / persistent store stream?  (GPU box only)
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_oneshot tools/experiments/ubench_oneshot.hip && tools/experiments/ubench_oneshot
// ubench_front.hip showed: not the address pattern (a persistent grid sweeping ONE linear window: 5.6-5.75 TB/s against
// 6.9), not the stores a wave keeps in flight (1, 2, 4, unlimited: the same), not the occupancy (4 ... 32 waves per CU).
// Here: workgroup size, stores per thread, a delay or a load in front of the store, dispatcher-driven against
// ticket-driven assignment of the same 4 KiB chunks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one-shot: workgroup b writes pieces [b R, (b + 1) R) of T * 16 B each; optional sleep / load in front
template <int T, int R, int SLEEP, bool LOAD>
__global__ __launch_bounds__(T) void shot(f4 *out, const float *in, size_t n16) {
    extern __shared__ char smem[];
    float x = 1.0f;
    if constexpr (LOAD) x = in[((size_t)blockIdx.x * T + threadIdx.x) % (n16 / 4)];       // 4 B read per 16 R B written
    if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const size_t i = ((size_t)blockIdx.x * R + k) * T + threadIdx.x;
        if (i < n16) out[i] = f4{x, (float)k, (float)threadIdx.x, 4.f};
    }
}
// the same chunks handed out by a ticket counter to a resident grid
template <int T, int R>
__global__ __launch_bounds__(T) void ticketed(f4 *out, size_t n16, unsigned *ticket) {
    __shared__ unsigned s_t;
    const size_t chunks = n16 / ((size_t)T * R);
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if (t >= chunks) break;
#pragma unroll
        for (int k = 0; k < R; ++k) out[((size_t)t * R + k) * T + threadIdx.x] = f4{1.f, (float)k, (float)threadIdx.x, 4.f};
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

static f4 *g_out; static float *g_in; static size_t g_n16; static double g_bytes;

template <int T, int R, int SLEEP, bool LOAD> void one(const char *what, size_t lds = 0) {
    const unsigned grid = (unsigned)(g_n16 / ((size_t)T * R));
    const float ms = timeit([&] { shot<T, R, SLEEP, LOAD><<<grid, T, lds>>>(g_out, g_in, g_n16); });
    const double b = (double)grid * T * R * 16 * (LOAD ? 1.0 + 0.25 / R : 1.0);
    printf("one-shot T=%4d R=%2d %-34s %7.1f GB/s  (%.3f ms, %u workgroups)\n", T, R, what, b / ms / 1e6, ms, grid);
}

int main() {
    g_bytes = (double)((size_t)3 << 30);
    g_n16 = ((size_t)3 << 30) / 16;
    void *o, *in; unsigned *ticket;
    CK(hipMalloc(&o, (size_t)3 << 30)); CK(hipMemset(o, 0, (size_t)3 << 30));
    CK(hipMalloc(&in, (size_t)3 << 28)); CK(hipMemset(in, 0, (size_t)3 << 28));
    CK(hipMalloc(&ticket, 4));
    g_out = (f4 *)o; g_in = (float *)in;

    one<64, 1, 0, false>("");   one<128, 1, 0, false>(""); one<192, 1, 0, false>(""); one<256, 1, 0, false>("");
    one<384, 1, 0, false>("");  one<512, 1, 0, false>(""); one<768, 1, 0, false>(""); one<1024, 1, 0, false>("");
    one<64, 2, 0, false>("");   one<64, 4, 0, false>("");  one<64, 12, 0, false>("");
    one<128, 2, 0, false>("");  one<128, 3, 0, false>(""); one<128, 6, 0, false>("");
    one<256, 2, 0, false>("");  one<256, 3, 0, false>(""); one<256, 4, 0, false>(""); one<256, 8, 0, false>(""); one<256, 12, 0, false>("");
    one<768, 2, 0, false>("");  one<768, 4, 0, false>("");
    one<256, 1, 20, false>("sleep ~0.5 us first");  one<256, 1, 80, false>("sleep ~2 us first");
    one<256, 3, 20, false>("sleep ~0.5 us first");  one<256, 3, 80, false>("sleep ~2 us first");
    one<256, 1, 0, true>("4 B load per thread first");  one<256, 3, 0, true>("4 B load per thread first");
    one<256, 12, 0, true>("4 B load per thread first"); one<768, 4, 0, true>("4 B load per thread first");
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&shot<256, 1, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&shot<256, 3, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    one<256, 1, 0, false>("4 wg/CU (36 KiB LDS)", 36 * 1024);  one<256, 1, 0, false>("2 wg/CU (72 KiB LDS)", 72 * 1024);
    one<256, 3, 0, false>("4 wg/CU (36 KiB LDS)", 36 * 1024);  one<256, 3, 0, false>("2 wg/CU (72 KiB LDS)", 72 * 1024);
    for (int wg : {2, 4, 8}) {
        const unsigned G = 256 * wg;
        float ms = timeit([&] { CK(hipMemsetAsync(ticket, 0, 4)); ticketed<256, 1><<<G, 256>>>(g_out, g_n16, ticket); });
        printf("ticketed T= 256 R= 1 resident grid of %4u workgroups          %7.1f GB/s  (%.3f ms)\n", G, g_bytes / ms / 1e6, ms);
        ms = timeit([&] { CK(hipMemsetAsync(ticket, 0, 4)); ticketed<256, 3><<<G, 256>>>(g_out, g_n16, ticket); });
        printf("ticketed T= 256 R= 3 resident grid of %4u workgroups          %7.1f GB/s  (%.3f ms)\n", G, g_bytes / ms / 1e6, ms);
    }
    printf("done\n");
    return 0;
}
