// Can a physical chunk of the kind the HBM zone arena takes (hipMemCreate) be created shareable and exported (GPU box only)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/ubench_vmm_export tools/experiments/ubench_vmm_export.hip && tools/experiments/ubench_vmm_export
// The arena's memory is not IPC-exportable through hipIpcGetMemHandle (what RCCL's peer-to-peer transport asks for), which is why the
// buffers of distributed.fuse_replicated are allocated plainly (DESIGN.md sections 3 and 6).  The virtual-memory API has its own
// route: requestedHandleTypes = hipMemHandleTypePosixFileDescriptor + hipMemExportToShareableHandle.  VERDICT r4 item 6: try it at world 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

static const char *err(hipError_t e) { return hipGetErrorString(e); }

int main() {
    const size_t CH = (size_t)1 << 30;
    for (int shareable = 0; shareable < 2; ++shareable) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        if (shareable) prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, CH, &prop, 0);
        printf("hipMemCreate(1 GiB, requestedHandleTypes = %s): %s\n", shareable ? "PosixFileDescriptor" : "None", err(e));
        if (e != hipSuccess) continue;
        int fd = -1;
        e = hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0);
        printf("  hipMemExportToShareableHandle -> %s (fd %d)\n", err(e), fd);
        if (e == hipSuccess) {
            hipMemGenericAllocationHandle_t h2;
            e = hipMemImportFromShareableHandle(&h2, (void *)(long)fd, hipMemHandleTypePosixFileDescriptor);
            printf("  hipMemImportFromShareableHandle (same process) -> %s\n", err(e));
            if (e == hipSuccess) {
                void *va = nullptr;
                hipMemAccessDesc acc = {};
                acc.location = prop.location;
                acc.flags = hipMemAccessFlagsProtReadWrite;
                hipError_t r = hipMemAddressReserve(&va, CH, 0, nullptr, 0);
                hipError_t m = r == hipSuccess ? hipMemMap(va, CH, 0, h2, 0) : r;
                hipError_t a = m == hipSuccess ? hipMemSetAccess(va, CH, &acc, 1) : m;
                hipError_t s = a == hipSuccess ? hipMemset(va, 0x5a, CH) : a;
                printf("  reserve / map / set access / memset of the imported handle -> %s / %s / %s / %s\n", err(r), err(m), err(a), err(s));
                hipIpcMemHandle_t ipc;
                hipError_t i = a == hipSuccess ? hipIpcGetMemHandle(&ipc, va) : a;
                printf("  hipIpcGetMemHandle on the mapped range (what RCCL's P2P transport calls) -> %s\n", err(i));
                if (m == hipSuccess) (void)hipMemUnmap(va, CH);
                (void)hipMemRelease(h2);
            }
            if (fd >= 0) close(fd);
        }
        (void)hipGetLastError();
        (void)hipMemRelease(h);
    }
    return 0;
}
