// Round trip of a flag between two workgroups: on the SAME XCD against on DIFFERENT XCDs, for every cache-scope pair of the
// store and the polling load (GPU box only).
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_xcd_pingpong tools/experiments/ubench_xcd_pingpong.hip && tools/experiments/ubench_xcd_pingpong
// The decoupled look-back of the densify kernel polls its predecessors' granules with agent-scope loads (sc1): DESIGN.md
// section 4 measures one polling round trip at ~0.48 us and half of a workgroup's life spent in it.  If two workgroups of ONE
// XCD can hand a flag over through that XCD's L2 (a load that may hit the L2 but not the CU's L1), tiles dealt to the XCDs in
// runs of consecutive tiles would look back at L2 latency and cross the fabric once per run.
// Workgroup b of a launch lands on XCD b mod 8 (checked here with the XCC_ID hardware register).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// scope bits of a global load / store: 0 = none (CU), 1 = sc0 (workgroup), 2 = sc1 (agent), 3 = sc0 sc1 (system), 4 = nt
template <int S> __device__ __forceinline__ unsigned ld(const unsigned *p) {
    unsigned v;
    if constexpr (S == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (S == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (S == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (S == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (S == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int S> __device__ __forceinline__ void st(unsigned *p, unsigned v) {
    if constexpr (S == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (S == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (S == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (S == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (S == 4) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}

// workgroups `a` and `b` of the launch play; everybody else leaves.  flag[0]: a -> b, flag[32]: b -> a (different lines).
template <int SS, int LS>
__global__ void pingpong(unsigned *flag, int a, int b, int iters, unsigned *xcc_out, long long *ticks_out, int *gave_up) {
    const int me = blockIdx.x;
    if (threadIdx.x == 0 && me < 64) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_out[me] = x & 0xf;
    }
    if (me != a && me != b) return;
    if (threadIdx.x != 0) return;
    unsigned *mine = flag + (me == a ? 0 : 32), *theirs = flag + (me == a ? 32 : 0);
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= iters; ++i) {
        if (me == a) st<SS>(mine, (unsigned)i);
        unsigned spins = 0;
        while (ld<LS>(theirs) < (unsigned)i) {
            if (++spins > (1u << 18)) { *gave_up = 1; return; }      // the scope pair does not carry the flag across
        }
        if (me == b) st<SS>(mine, (unsigned)i);
    }
    if (me == a) *ticks_out = __builtin_amdgcn_s_memrealtime() - t0;      // 100 MHz
}

int main() {
    unsigned *flag, *xcc;
    long long *ticks;
    int *gave;
    CK(hipMalloc(&flag, 4096));
    CK(hipMalloc(&xcc, 64 * 4));
    CK(hipMalloc(&ticks, 8));
    CK(hipMalloc(&gave, 4));
    const int iters = 2000;
    const char *sn[5] = {"none", "sc0", "sc1", "sc0 sc1", "nt"};
    printf("# one-way hand-over of a flag between two workgroups (half a ping-pong round trip), microseconds; '-' = the flag never arrived\n");
    printf("# store scope x load scope; pairs: workgroups 0 and 8 (same XCD), 0 and 1 (neighbouring XCDs), 0 and 4\n");
    for (int pair = 0; pair < 3; ++pair) {
        const int a = 0, b = pair == 0 ? 8 : pair == 1 ? 1 : 4;
        std::vector<unsigned> hx(64);
        for (int ss = 1; ss <= 3; ++ss) {
            for (int ls = 0; ls <= 4; ++ls) {
                CK(hipMemset(flag, 0, 4096));
                CK(hipMemset(gave, 0, 4));
                CK(hipMemset(ticks, 0, 8));
#define LAUNCH(SS, LS) if (ss == SS && ls == LS) pingpong<SS, LS><<<64, 64>>>(flag, a, b, iters, xcc, ticks, gave);
                LAUNCH(1, 0) LAUNCH(1, 1) LAUNCH(1, 2) LAUNCH(1, 3) LAUNCH(1, 4)
                LAUNCH(2, 0) LAUNCH(2, 1) LAUNCH(2, 2) LAUNCH(2, 3) LAUNCH(2, 4)
                LAUNCH(3, 0) LAUNCH(3, 1) LAUNCH(3, 2) LAUNCH(3, 3) LAUNCH(3, 4)
                CK(hipDeviceSynchronize());
                long long t; int g;
                CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&g, gave, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hx.data(), xcc, 256, hipMemcpyDeviceToHost));
                if (g) printf("wg %d (xcc %u) <-> wg %d (xcc %u)  store %-7s load %-7s      -\n", a, hx[a], b, hx[b], sn[ss], sn[ls]);
                else printf("wg %d (xcc %u) <-> wg %d (xcc %u)  store %-7s load %-7s  %6.3f us\n", a, hx[a], b, hx[b], sn[ss], sn[ls], t / 100.0 / iters / 2.0);
            }
        }
    }
    return 0;
}
