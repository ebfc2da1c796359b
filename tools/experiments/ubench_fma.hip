// Issue rate of v_fma_f64 vs v_fma_f32 vs v_pk_fma_f32 on gfx950 (GPU box only):
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/ubench_fma tools/experiments/ubench_fma.hip && /tmp/ubench_fma
// 8 independent accumulators per lane, 4096 rounds; 256 CUs x 8 workgroups x 256 threads.  Answers whether a float32
// first pass of the floater-vote kernel has an arithmetic advantage over float64 at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ T mad(T a, T b, T c);
template <> __device__ __forceinline__ double mad(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ f2 mad(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <typename T> __global__ __launch_bounds__(256) void chain(T *out, T a, T b, int rounds) {
    T acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = a * (T)(float)(threadIdx.x + k);
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = mad<T>(acc[k], a, b);
    }
    T s = acc[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) s += acc[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T> void run(const char *name, int lanes_per_op) {
    const int grid = 256 * 8, rounds = 4096;
    T *o; CK(hipMalloc((void **)&o, (size_t)grid * 256 * sizeof(T)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    chain<T><<<grid, 256>>>(o, (T)1.0000001f, (T)0.5f, rounds);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) chain<T><<<grid, 256>>>(o, (T)1.0000001f, (T)0.5f, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double instr = (double)grid * 256 / 64 * rounds * 8;                    // wave-level FMA instructions
    const double per_simd_cycle = instr / (256.0 * 4) / (ms * 1e-3 * 2.4e9);       // instructions per SIMD per clock at 2.4 GHz
    printf("%-14s %.3f ms  %.1f TFLOP/s  %.2f cycles per wave instruction (at 2.4 GHz)\n", name, ms,
           instr * 64 * lanes_per_op * 2 / (ms * 1e-3) / 1e12, 1.0 / per_simd_cycle);
    CK(hipFree(o));
}

int main() {
    run<double>("v_fma_f64", 1);
    run<float>("v_fma_f32", 1);
    run<f2>("v_pk_fma_f32", 2);
    return 0;
}
