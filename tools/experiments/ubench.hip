// Micro-benchmark of the memory access shapes used by the scatter kernel (GPU box only):
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/experiments/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef u3 u3u __attribute__((aligned(1)));
typedef unsigned u1u __attribute__((aligned(1)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each thread handles `per` elements strided by the grid (grid-stride keeps every instruction contiguous)
__global__ void w_x4(f4 *o, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f4 v = {1.f, 2.f, 3.f, (float)i}; o[i] = v; } }
__global__ void w_x3(float *o, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f3 v = {1.f, 2.f, (float)i}; *reinterpret_cast<f3 *>(o + i * 3) = v; } }
__global__ void w_x1(float *o, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = (float)i; }
__global__ void w_rgb(unsigned char *o, size_t n) {   // n points, 3 B each, stored as 12 B by every 4th lane (unaligned)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if ((threadIdx.x & 3) == 0) { u3 v = {(unsigned)i, 2u, 3u}; *reinterpret_cast<u3u *>(o + i * 3 + 1) = v; } } }
__global__ void w_rgb_bytes(unsigned char *o, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { o[i*3] = 1; o[i*3+1] = 2; o[i*3+2] = (unsigned char)i; } }
__global__ void r_x4(const f4 *in, float *o, size_t n) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f4 v = in[i]; s += v.x + v.y + v.z + v.w; } if (s == 123.f) o[0] = s; }
__global__ void r_x3(const float *in, float *o, size_t n) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f3 v = *reinterpret_cast<const f3 *>(in + i * 3); s += v.x + v.y + v.z; } if (s == 123.f) o[0] = s; }
__global__ void r_rgb(const unsigned char *in, float *o, size_t n) { unsigned s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { s += *reinterpret_cast<const u1u *>(in + i * 3 + 2); } if (s == 123u) o[0] = s; }
__global__ void c_x4(const f4 *in, f4 *o, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = in[i]; }
__global__ void c_x3(const float *in, float *o, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) *reinterpret_cast<f3 *>(o + i * 3) = *reinterpret_cast<const f3 *>(in + i * 3); }
// one-shot stores: huge grid, one 16-B (or 12-B) store per thread, no loop (torch's elementwise launch shape)
__global__ void w_x4_oneshot(f4 *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { f4 v = {1.f, 2.f, 3.f, (float)i}; o[i] = v; } }
__global__ void w_x3_oneshot(float *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { f3 v = {1.f, 2.f, (float)i}; *reinterpret_cast<f3 *>(o + i * 3) = v; } }
// each block owns a contiguous chunk and streams through it (4 stores in flight per thread)
__global__ void w_x4_chunk(f4 *o, size_t n, size_t per_block) {
  const size_t b0 = blockIdx.x * per_block; const size_t b1 = b0 + per_block < n ? b0 + per_block : n;
  for (size_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) { f4 v = {1.f, 2.f, 3.f, (float)i}; o[i] = v; } }
__global__ void w_x4_nt(f4 *o, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { f4 v = {1.f, 2.f, 3.f, (float)i}; __builtin_nontemporal_store(v, &o[i]); } }
__global__ void w_x4_const(f4 *o, size_t n) { const f4 v = {1.f, 1.f, 1.f, 1.f}; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = v; }
__global__ void c_x4_oneshot(const f4 *a, f4 *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) o[i] = a[i]; }
__global__ void c_x3_oneshot(const float *a, float *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) *reinterpret_cast<f3 *>(o + i * 3) = *reinterpret_cast<const f3 *>(a + i * 3); }
// cross shapes: which side of a copy pays for 12-byte accesses?
__global__ void c_r4w3_oneshot(const f4 *a, float *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { const f4 v = a[i]; f3 w = {v.x, v.y, v.z + v.w}; *reinterpret_cast<f3 *>(o + i * 3) = w; } }
__global__ void c_r3w4_oneshot(const float *a, f4 *o, size_t n) { const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) { const f3 v = *reinterpret_cast<const f3 *>(a + i * 3); f4 w = {v.x, v.y, v.z, 1.f}; o[i] = w; } }
// read 12-B elements, write the same stream as 16-B chunks (lane j of a 768-thread block reads 3 dwords, 576 lanes store dwordx4 via LDS)
__global__ __launch_bounds__(768) void c_x3_to_x4(const float *a, float *o, size_t npts) {
  __shared__ float st[768 * 3];
  const size_t p0 = blockIdx.x * (size_t)768;
  const unsigned t = threadIdx.x;
  if (p0 + t < npts) { f3 v = *reinterpret_cast<const f3 *>(a + (p0 + t) * 3); st[t * 3] = v.x; st[t * 3 + 1] = v.y; st[t * 3 + 2] = v.z; }
  __syncthreads();
  if (t < 576 && p0 + 768 <= npts) *reinterpret_cast<f4 *>(o + p0 * 3 + t * 4) = *reinterpret_cast<const f4 *>(st + t * 4);
}
// unrolled x4 copy, 4 loads in flight per thread
__global__ void c_x4u(const f4 *in, f4 *o, size_t n) { const size_t st = (size_t)gridDim.x * blockDim.x; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i + 3 * st < n; i += 4 * st) { f4 a = in[i], b = in[i + st], c = in[i + 2 * st], d = in[i + 3 * st]; o[i] = a; o[i + st] = b; o[i + 2 * st] = c; o[i + 3 * st] = d; } }

template <typename F> double run(const char *name, double bytes, F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); const int it = 8; for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double gbps = bytes * it / (ms * 1e-3) / 1e9; printf("%-34s %8.1f GB/s  (%.3f ms)\n", name, gbps, ms / it); return gbps; }

int main(int argc, char **argv) {
  const size_t BYTES = (size_t)3 << 30;   // 3 GiB per buffer
  void *a, *b; CK(hipMalloc(&a, BYTES + 256)); CK(hipMalloc(&b, BYTES + 256)); CK(hipMemset(a, 1, BYTES)); CK(hipMemset(b, 2, BYTES));
  for (int wg : {2048, 8192}) {
    printf("-- grid %d x 256\n", wg);
    dim3 g(wg), blk(256);
    run("write dwordx4 contiguous", BYTES, [&] { w_x4<<<g, blk>>>((f4 *)b, BYTES / 16); });
    run("write dwordx3 (12 B stride)", BYTES, [&] { w_x3<<<g, blk>>>((float *)b, BYTES / 12); });
    if (wg == 2048) {
      run("write dwordx4 one-shot grid", BYTES, [&] { w_x4_oneshot<<<dim3((unsigned)(BYTES / 16 / 256)), blk>>>((f4 *)b, BYTES / 16); });
      run("write dwordx3 one-shot grid", BYTES, [&] { w_x3_oneshot<<<dim3((unsigned)(BYTES / 12 / 256)), blk>>>((float *)b, BYTES / 12); });
      run("write dwordx4 one-shot +4B misaligned", BYTES, [&] { w_x4_oneshot<<<dim3((unsigned)(BYTES / 16 / 256) - 1), blk>>>((f4 *)((char *)b + 4), BYTES / 16 - 256); });
      run("copy dwordx4 one-shot", 2 * BYTES, [&] { c_x4_oneshot<<<dim3((unsigned)(BYTES / 16 / 256)), blk>>>((const f4 *)a, (f4 *)b, BYTES / 16); });
      run("copy dwordx3 one-shot", 2 * BYTES, [&] { c_x3_oneshot<<<dim3((unsigned)(BYTES / 12 / 256)), blk>>>((const float *)a, (float *)b, BYTES / 12); });
      { const size_t n = BYTES / 16; run("copy read x4 -> write x3 one-shot", n * 28, [&] { c_r4w3_oneshot<<<dim3((unsigned)(n / 256)), blk>>>((const f4 *)a, (float *)b, n); }); }
      { const size_t n = BYTES / 16; run("copy read x3 -> write x4 one-shot", n * 28, [&] { c_r3w4_oneshot<<<dim3((unsigned)(n / 256)), blk>>>((const float *)a, (f4 *)b, n); }); }
      run("copy read x3 -> LDS -> write x4", 2 * BYTES, [&] { c_x3_to_x4<<<dim3((unsigned)(BYTES / 12 / 768)), dim3(768)>>>((const float *)a, (float *)b, BYTES / 12); });
      run("write dwordx4 block chunks 64K", BYTES, [&] { w_x4_chunk<<<dim3((unsigned)(BYTES / 65536)), blk>>>((f4 *)b, BYTES / 16, 4096); });
      run("write dwordx4 block chunks 1M", BYTES, [&] { w_x4_chunk<<<dim3((unsigned)(BYTES / 1048576)), blk>>>((f4 *)b, BYTES / 16, 65536); });
      run("write dwordx4 nontemporal", BYTES, [&] { w_x4_nt<<<g, blk>>>((f4 *)b, BYTES / 16); });
      run("write dwordx4 constant value", BYTES, [&] { w_x4_const<<<g, blk>>>((f4 *)b, BYTES / 16); });
    }
    run("write dword contiguous", BYTES, [&] { w_x1<<<g, blk>>>((float *)b, BYTES / 4); });
    run("write rgb 12B/4 lanes unaligned", BYTES / 4, [&] { w_rgb<<<g, blk>>>((unsigned char *)b, BYTES / 12); });
    run("write rgb 3 byte stores", BYTES / 4, [&] { w_rgb_bytes<<<g, blk>>>((unsigned char *)b, BYTES / 12); });
    run("read  dwordx4 contiguous", BYTES, [&] { r_x4<<<g, blk>>>((const f4 *)a, (float *)b, BYTES / 16); });
    run("read  dwordx3 (12 B stride)", BYTES, [&] { r_x3<<<g, blk>>>((const float *)a, (float *)b, BYTES / 12); });
    run("read  rgb unaligned dword/3B", BYTES / 4, [&] { r_rgb<<<g, blk>>>((const unsigned char *)a, (float *)b, BYTES / 12); });
    run("copy  dwordx4", 2.0 * BYTES, [&] { c_x4<<<g, blk>>>((const f4 *)a, (f4 *)b, BYTES / 16); });
    run("copy  dwordx4 unrolled x4", 2.0 * BYTES, [&] { c_x4u<<<g, blk>>>((const f4 *)a, (f4 *)b, BYTES / 16); });
    run("copy  dwordx3", 2.0 * BYTES, [&] { c_x3<<<g, blk>>>((const float *)a, (float *)b, BYTES / 12); });
  }
  return 0;
}
