// The peer schedule and row arithmetic of dd_allgatherv (csrc/ddcomm.hip; the fuse of scripts/test.py:262-266 across GPUs)
// executed for MORE THAN ONE RANK without a GPU and without RCCL: this program IS the "RCCL runtime already loaded in the
// process" -- it exports ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv / ncclGetErrorString, which libddcore.so
// resolves with dlsym(RTLD_DEFAULT) -- and plays all ranks of a communicator one after the other on host buffers
// (dd_allgatherv itself never touches the device: it only posts sends and receives).  Every call is logged; after all
// ranks have posted, the sends are matched with the receives the way NCCL does (in order per ordered pair of ranks)
// and the bytes are moved.  Checked for world in {2, 3, 8}, dst in {-1 (replicate), every rank (gather-to-owner)}, all six
// fields, ragged row counts with empty ranks:
//   * every send has a receive of equal size on the peer, and nothing is left over on either side;
//   * every receiving rank ends up with the whole fused cloud: the received ranges and its own rows tile [0, total) once;
//   * a pure sender's buffers start at its own first row (base offset honoured) and are never written.
//     g++ -std=c++17 -O1 -rdynamic -I include tests/c_client/fake_rccl.cpp -L depthdensifier_amd -lddcore -ldl -o fake_rccl
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ddcore.h"

namespace {
struct Post { int rank, peer; char *ptr; size_t bytes; bool used; };
std::vector<Post> g_sends, g_recvs;
int g_depth = 0, g_groups = 0, g_outside = 0;
int rank_of(void *comm) { return (int)(reinterpret_cast<uintptr_t>(comm) - 1); }
}  // namespace

extern "C" {
int ncclGroupStart() { ++g_depth; ++g_groups; return 0; }
int ncclGroupEnd() { --g_depth; return 0; }
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *) {
    if (g_depth != 1) ++g_outside;
    if (dtype != 1) { printf("ncclSend: dtype %d is not ncclUint8\n", dtype); exit(1); }        // ncclUint8 == 1
    g_sends.push_back({rank_of(comm), peer, const_cast<char *>(static_cast<const char *>(buf)), count, false});
    return 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *) {
    if (g_depth != 1) ++g_outside;
    if (dtype != 1) { printf("ncclRecv: dtype %d is not ncclUint8\n", dtype); exit(1); }
    g_recvs.push_back({rank_of(comm), peer, static_cast<char *>(buf), count, false});
    return 0;
}
const char *ncclGetErrorString(int) { return "fake"; }
}

#define REQUIRE(c, ...) do { if (!(c)) { printf("FAILED %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); exit(1); } } while (0)

static const size_t ROW[6] = {12, 12, 3, 4, 4, 16};      // xyz, normal, rgb, pixel_index, view_index, xyz_rgba

static unsigned char pattern(int field, int64_t row, size_t byte) { return (unsigned char)(17 * field + 31 * row + 7 * byte + 3); }

static void run(int world, int dst, const std::vector<int64_t> &counts, unsigned field_mask) {
    std::vector<int64_t> rows(world + 1, 5);                  // the fused cloud starts at global row 5 (a non-zero first row)
    for (int r = 0; r < world; ++r) rows[r + 1] = rows[r] + counts[r];
    const int64_t total = rows[world];
    g_sends.clear(); g_recvs.clear(); g_groups = 0; g_outside = 0;
    // buffers: receivers hold the whole cloud [0, total), pure senders just their own rows
    std::vector<std::vector<std::vector<unsigned char>>> buf(world, std::vector<std::vector<unsigned char>>(6));
    for (int r = 0; r < world; ++r) {
        const bool receives = dst < 0 || dst == r;
        const int64_t base = receives ? 0 : rows[r], n = receives ? total : counts[r];
        for (int f = 0; f < 6; ++f) {
            if (!((field_mask >> f) & 1u)) continue;
            buf[r][f].assign((size_t)n * ROW[f] + 64, 0xEE);                      // 64 guard bytes behind
            for (int64_t row = rows[r]; row < rows[r + 1]; ++row)                 // "the kernel wrote this rank's rows"
                for (size_t b = 0; b < ROW[f]; ++b) buf[r][f][(size_t)(row - base) * ROW[f] + b] = pattern(f, row, b);
        }
    }
    for (int r = 0; r < world; ++r) {
        const bool receives = dst < 0 || dst == r;
        DDCloudOut c{};
        auto p = [&](int f) -> void * { return buf[r][f].empty() ? nullptr : buf[r][f].data(); };
        c.xyz = (float *)p(0); c.normal = (float *)p(1); c.rgb = (uint8_t *)p(2); c.pixel_index = (int32_t *)p(3);
        c.view_index = (int32_t *)p(4); c.xyz_rgba = (uint32_t *)p(5);
        c.capacity = receives ? total : counts[r];
        const int rc = dd_allgatherv(reinterpret_cast<void *>((uintptr_t)(r + 1)), r, world, &c, rows.data(), dst, nullptr);
        REQUIRE(rc == DD_OK, "dd_allgatherv rank %d of %d: %d (%s)", r, world, rc, dd_comm_last_error());
    }
    REQUIRE(g_outside == 0, "sends / receives outside ncclGroupStart .. ncclGroupEnd");
    REQUIRE(g_groups == world, "one group per rank expected, saw %d for %d ranks", g_groups, world);
    // match: the k-th send a -> b with the k-th receive of b from a
    for (Post &s : g_sends) {
        bool found = false;
        for (Post &r : g_recvs) {
            if (r.used || r.rank != s.peer || r.peer != s.rank) continue;
            REQUIRE(r.bytes == s.bytes, "send %d -> %d of %zu bytes meets a receive of %zu", s.rank, s.peer, s.bytes, r.bytes);
            memcpy(r.ptr, s.ptr, s.bytes);
            r.used = s.used = found = true;
            break;
        }
        REQUIRE(found, "send %d -> %d (%zu bytes) has no receive", s.rank, s.peer, s.bytes);
    }
    for (const Post &r : g_recvs) REQUIRE(r.used, "receive on %d from %d (%zu bytes) has no send", r.rank, r.peer, r.bytes);
    int nfields = 0;
    for (int f = 0; f < 6; ++f) nfields += (field_mask >> f) & 1u;
    // what every rank must hold now
    for (int r = 0; r < world; ++r) {
        const bool receives = dst < 0 || dst == r;
        const int64_t base = receives ? 0 : rows[r], lo = receives ? rows[0] : rows[r], hi = receives ? total : rows[r + 1];
        for (int f = 0; f < 6; ++f) {
            if (buf[r][f].empty()) continue;
            const std::vector<unsigned char> &v = buf[r][f];
            const int64_t n = receives ? total : counts[r];
            for (int64_t row = base; row < base + n; ++row)
                for (size_t b = 0; b < ROW[f]; ++b) {
                    const unsigned char want = (row >= lo && row < hi) ? pattern(f, row, b) : 0xEE;     // rows in front of the cloud stay untouched
                    REQUIRE(v[(size_t)(row - base) * ROW[f] + b] == want, "world %d dst %d rank %d field %d row %lld byte %zu", world, dst, r, f, (long long)row, b);
                }
            for (size_t g = 0; g < 64; ++g) REQUIRE(v[(size_t)n * ROW[f] + g] == 0xEE, "world %d dst %d rank %d field %d: bytes behind the buffer written", world, dst, r, f);
        }
    }
    // traffic: a replicate moves (world - 1) x every rank's rows, a gather-to-owner every other rank's rows once
    size_t sent = 0, expect = 0;
    for (const Post &s : g_sends) sent += s.bytes;
    size_t row_bytes = 0;
    for (int f = 0; f < 6; ++f) if ((field_mask >> f) & 1u) row_bytes += ROW[f];
    for (int r = 0; r < world; ++r) expect += (size_t)counts[r] * row_bytes * (dst < 0 ? (size_t)(world - 1) : (r == dst ? 0u : 1u));
    REQUIRE(sent == expect, "world %d dst %d: %zu bytes sent, %zu expected", world, dst, sent, expect);
    (void)nfields;
}

int main() {
    int cases = 0;
    for (int world : {2, 3, 8}) {
        std::vector<std::vector<int64_t>> shapes;
        std::vector<int64_t> even(world, 1000), ragged(world), holes(world, 0);
        for (int r = 0; r < world; ++r) ragged[r] = 37 + 211 * ((r * 5 + 3) % 7);
        holes[world - 1] = 77; if (world > 2) holes[1] = 3;                       // ranks without a single row
        shapes = {even, ragged, holes};
        for (const auto &counts : shapes)
            for (unsigned mask : {0x3Fu, 0x01u, 0x20u, 0x07u})                   // all six fields; xyz only; the 16-byte record only; rows
                for (int dst = -1; dst < world; ++dst) { run(world, dst, counts, mask); ++cases; }
    }
    // error conventions that need no communicator
    DDCloudOut c{}; int64_t rows2[3] = {0, 4, 2};
    float dummy[64]; c.xyz = dummy; c.capacity = 4;
    if (dd_allgatherv((void *)1, 0, 2, &c, rows2, -1, nullptr) != DD_ERR_INVALID_ARG) { printf("decreasing rank_rows accepted\n"); return 1; }
    int64_t rows3[3] = {0, 4, 9};
    if (dd_allgatherv((void *)1, 0, 2, &c, rows3, -1, nullptr) != DD_ERR_INVALID_ARG) { printf("a capacity below the fused cloud accepted\n"); return 1; }
    if (dd_allgatherv(nullptr, 0, 2, &c, rows3, 1, nullptr) != DD_ERR_INVALID_ARG) { printf("NULL comm accepted\n"); return 1; }
    printf("fake RCCL: %d exchanges OK\n", cases);
    return 0;
}
