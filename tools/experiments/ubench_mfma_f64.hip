// Does v_mfma_f64_16x16x4_f64 take work off the vector ALU's issue port on gfx950 (GPU box only)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_mfma_f64 tools/experiments/ubench_mfma_f64.hip && tools/experiments/ubench_mfma_f64
// The floater-vote kernel (csrc/ddfilter.hip, scripts/test.py:273-328) is bound by VALU issue: 56.7 wave instructions per wave and
// view, 12 of them the float64 FMAs of three dot products of length 4 (zc, nu, nw) -- an (N x 4) x (4 x 3V) contraction, the
// shape of the K = 4 float64 matrix instruction.  One MFMA covers 16 points x 16 rows = 16 points x 4 views
// (three rows used of four): the same 64 pairs as one round of 12 FMAs.  Rounds of M matrix instructions + F float64 FMAs,
// independent accumulators, 8 waves per SIMD: if the matrix core runs beside the VALU, (1, 45) costs what (0, 45) costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int M, int F> __global__ __launch_bounds__(256) void mix(double *out, double a, double b, int rounds) {
    d4 acc[M > 0 ? M : 1];
    double f[F > 0 ? F : 1];
#pragma unroll
    for (int k = 0; k < (M > 0 ? M : 1); ++k) acc[k] = d4{a, b, a + k, b};
#pragma unroll
    for (int k = 0; k < (F > 0 ? F : 1); ++k) f[k] = a * (double)(threadIdx.x + k);
    const double av = a + threadIdx.x * 1e-9, bv = b + threadIdx.x * 1e-9;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < M; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < F; ++k) f[k] = __builtin_fma(f[k], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < (M > 0 ? M : 1); ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
#pragma unroll
    for (int k = 0; k < (F > 0 ? F : 1); ++k) s += f[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int M, int F> void run() {
    const int grid = 256 * 8, rounds = 1024;
    double *o; CK(hipMalloc((void **)&o, (size_t)grid * 256 * sizeof(double)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mix<M, F><<<grid, 256>>>(o, 1.0000001, 0.5, rounds);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) mix<M, F><<<grid, 256>>>(o, 1.0000001, 0.5, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double waves = (double)grid * 256 / 64;
    const double cyc = ms * 1e-3 * 2.4e9 * (256.0 * 4) / (waves * rounds);      // SIMD cycles per round of one wave (at 2.4 GHz)
    printf("M = %d matrix + F = %2d FMA per round: %.3f ms, %6.1f SIMD cycles per wave round", M, F, ms, cyc);
    if (M) printf("  (%.1f TFLOP/s of float64 MFMA)", waves * rounds * M * 2048.0 / (ms * 1e-3) / 1e12);
    printf("\n");
    CK(hipFree(o));
}

int main() {
    run<1, 0>(); run<4, 0>(); run<0, 12>(); run<0, 45>(); run<0, 57>();
    run<1, 12>(); run<1, 45>(); run<2, 45>(); run<1, 33>();
    return 0;
}
