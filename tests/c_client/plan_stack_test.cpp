// plan_stack_test.cpp -- the host side's decisions must not depend on what the stack held before the call.  Round 6 found the
// "Memory access fault by GPU ... on address (nil)" of the three-rank rehearsal here: `Plan::chain_gate` was only written for chained
// calls, an unchained dd_unproject_compact read whatever the stack held -- and, when that was non-zero, launched the one-wave gate
// kernel with a NULL chain word (profiles/r06_fault_root_cause.txt).  No GPU needed: dd_debug_plan runs the same make_plan.
//   g++ -std=c++17 -I include tests/c_client/plan_stack_test.cpp -L depthdensifier_amd -lddcore -Wl,-rpath,depthdensifier_amd
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ddcore.h"
#include "ddcore_lab.h"

__attribute__((noinline)) static void paint_stack(unsigned char value) {
    volatile unsigned char buf[32768];
    for (size_t i = 0; i < sizeof(buf); ++i) buf[i] = value;
}

__attribute__((noinline)) static int plan(const DDViewBatch *b, int32_t *out) { return dd_debug_plan(b, out); }

int main() {
    DDViewBatch b; memset(&b, 0, sizeof(b));
    b.num_views = 2; b.height = 120; b.width = 200; b.stride = 1;
    b.depth = (const void *)0x10000; b.params = (const DDViewParams *)0x20000; b.mask = (const uint8_t *)0x30000;
    b.depth_dtype = DD_F32; b.flags = DD_VALID_MASK | DD_VALID_DEPTH_POSITIVE;
    int32_t first[8] = {0}, out[8];
    int bad = 0;
    const unsigned char paints[] = {0x00, 0xff, 0x01, 0xa5, 0x80};
    for (size_t k = 0; k < sizeof(paints); ++k) {
        paint_stack(paints[k]);
        memset(out, 0x55, sizeof(out));
        const int rc = plan(&b, out);
        if (rc != DD_OK) { printf("dd_debug_plan: %d %s\n", rc, dd_last_error()); return 2; }
        if (k == 0) memcpy(first, out, sizeof(out));
        if (memcmp(first, out, sizeof(out)) != 0 || out[6] != 0) {
            printf("stack painted 0x%02x: plan { lean %d single %d refine %d f16 %d tile %d pxt %d gate %d tiles %d } differs from the clean one (gate %d)\n",
                   paints[k], out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], first[6]);
            bad = 1;
        }
    }
    if (bad) return 1;
    printf("plan independent of the stack: lean %d single %d pxt %d gate %d tiles %d  OK\n", first[0], first[1], first[5], first[6], first[7]);
    return 0;
}
