// prefetch_stress.cpp -- the native prefetcher (csrc/ddingest.hip) under ThreadSanitizer / AddressSanitizer on the CPU: a consumer that
// submits `ahead` jobs in front of itself, waits, checks every byte of every slot against what the files hold, releases, and submits the
// next -- against 1..8 worker threads and 2..9 slots, files of changing sizes, a missing file in between, a destroy with jobs queued.
// Built by tests/test_host_cpu.py::test_the_prefetcher_under_sanitizers (no GPU: slots are ordinary memory, events NULL).
//   usage: prefetch_stress <directory> <rounds>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ddcore.h"

static void write_npy(const std::string &path, const char *descr, long long h, long long w, const std::vector<unsigned char> &data) {
    char dict[256];
    int n = snprintf(dict, sizeof(dict), "{'descr': '%s', 'fortran_order': False, 'shape': (%lld, %lld), }", descr, h, w);
    int total = 10 + n + 1;
    int pad = (64 - total % 64) % 64;
    std::string hdr(dict);
    hdr.append((size_t)pad, ' ');
    hdr.push_back('\n');
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(hdr.size() & 255), (unsigned char)(hdr.size() >> 8)};
    fwrite(pre, 1, 10, f);
    fwrite(hdr.data(), 1, hdr.size(), f);
    fwrite(data.data(), 1, data.size(), f);
    fclose(f);
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL line %d: ", __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [%s]\n", dd_ingest_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string dir = argv[1];
    const int rounds = atoi(argv[2]);
    const int V = 23;
    std::vector<std::vector<unsigned char>> depth(V), mask(V);
    std::vector<long long> hs(V), ws(V);
    uint32_t x = 12345;
    auto rnd = [&] { x = x * 1664525u + 1013904223u; return x >> 8; };
    long long biggest = 0;
    for (int v = 0; v < V; ++v) {
        hs[v] = 3 + rnd() % 40; ws[v] = 5 + rnd() % 70;
        depth[v].resize((size_t)(hs[v] * ws[v] * 4)); mask[v].resize((size_t)(hs[v] * ws[v]));
        for (auto &b : depth[v]) b = (unsigned char)rnd();
        for (auto &b : mask[v]) b = (unsigned char)(rnd() & 1);
        write_npy(dir + "/v" + std::to_string(v) + "_depth.npy", "<f4", hs[v], ws[v], depth[v]);
        write_npy(dir + "/v" + std::to_string(v) + "_mask.npy", "|b1", hs[v], ws[v], mask[v]);
        if ((long long)depth[v].size() > biggest) biggest = (long long)depth[v].size();
    }
    const long long mask_at = (biggest + 63) & ~63LL, slot_bytes = mask_at + biggest / 4 + 64;
    long long checked = 0;
    for (int r = 0; r < rounds; ++r) {
        const int threads = 1 + r % 8, slots = 2 + (r * 3) % 8, ahead = 1 + (r % slots);
        DDPrefetcher *p = nullptr;
        CHECK(dd_prefetch_create(threads, slots, slot_bytes, &p) == 0, "create");
        std::vector<long long> tickets;
        auto submit = [&](int v, bool gone) {
            std::string a = dir + "/v" + std::to_string(v) + (gone ? "_nothing.npy" : "_depth.npy"), b = dir + "/v" + std::to_string(v) + "_mask.npy";
            const char *paths[2] = {a.c_str(), b.c_str()};
            int32_t codes[2] = {-1, DD_NPY_U8}, nds[2] = {2, 2};
            int64_t shapes[8] = {hs[v], ws[v], 1, 1, hs[v], ws[v], 1, 1}, offs[2] = {0, mask_at};
            return dd_prefetch_submit(p, 2, paths, codes, nds, shapes, offs);
        };
        const int total = 3 * V;
        int submitted = 0;
        auto gone = [&](int j) { return (j + r) % 11 == 7; };
        for (; submitted < ahead && submitted < total; ++submitted) tickets.push_back(submit(submitted % V, gone(submitted)));
        for (int j = 0; j < total; ++j) {
            const int v = j % V;
            CHECK(tickets[j] == j, "ticket %lld for job %d", tickets[j], j);
            void *base = nullptr; int32_t found[8];
            const int rc = dd_prefetch_wait(p, tickets[j], &base, found);
            if (gone(j)) CHECK(rc < 0 && strstr(dd_ingest_last_error(), "No such file"), "a missing file must fail the job, rc %d", rc);
            else {
                CHECK(rc == 0 && base, "wait %d", rc);
                CHECK(found[0] == DD_NPY_F32 && found[1] == DD_NPY_BOOL, "types %d %d", found[0], found[1]);
                CHECK(!memcmp(base, depth[v].data(), depth[v].size()), "depth bytes of job %d", j);
                CHECK(!memcmp((char *)base + mask_at, mask[v].data(), mask[v].size()), "mask bytes of job %d", j);
                checked += (long long)(depth[v].size() + mask[v].size());
            }
            if (j + slots < total || true) {
                // more jobs than slots without a release is refused -- tried now and then, when every slot is spoken for
                if (submitted - j == slots && submitted < total) CHECK(submit(submitted % V, false) < 0, "a job beyond the slots was accepted");
            }
            CHECK(dd_prefetch_release(p, tickets[j], nullptr) == 0, "release");
            CHECK(dd_prefetch_release(p, tickets[j], nullptr) < 0, "second release accepted");
            if (submitted < total) { tickets.push_back(submit(submitted % V, gone(submitted))); ++submitted; }
            if (r % 5 == 4 && j == total / 2) break;          // destroy with jobs queued, running and done
        }
        CHECK(dd_prefetch_destroy(p) == 0, "destroy");
    }
    // the header parser on damaged files: a valid file with a few header bytes changed, or cut short -- an error or a header, never a crash,
    // and a read into a buffer of the size the caller expects never writes past it (AddressSanitizer watches)
    {
        const std::string good = dir + "/v0_depth.npy", bad = dir + "/fuzz.npy";
        std::vector<unsigned char> bytes;
        { FILE *f = fopen(good.c_str(), "rb"); unsigned char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + n); fclose(f); }
        const char *snippets[] = {"(99999999999999999999, 9)", "(9223372036854775807, 9223372036854775807)", "(-3, 4)", "(,)", "((((", "'descr': '<f8'", "True", "'shape': 7", ""};
        std::vector<unsigned char> dst((size_t)(hs[0] * ws[0] * 4));
        long parsed = 0, refused = 0;
        for (int it = 0; it < 3000; ++it) {
            std::vector<unsigned char> m = bytes;
            const int kind = (int)(rnd() % 4);
            if (kind == 0) for (int k = 0; k < 1 + (int)(rnd() % 6); ++k) m[rnd() % 128] = (unsigned char)rnd();
            else if (kind == 1) m.resize(rnd() % 140);
            else if (kind == 2) { const char *sn = snippets[rnd() % 9]; size_t at = 10 + rnd() % 60; for (size_t k = 0; sn[k] && at + k < 128; ++k) m[at + k] = (unsigned char)sn[k]; }
            else { m[8] = (unsigned char)rnd(); m[9] = (unsigned char)rnd(); if (rnd() & 1) m[6] = (unsigned char)(1 + rnd() % 3); }
            FILE *f = fopen(bad.c_str(), "wb"); fwrite(m.data(), 1, m.size(), f); fclose(f);
            int32_t dt = -7, nd = -7; int64_t shape[4] = {-7, -7, -7, -7}, off = -7;
            const int rc = dd_npy_header(bad.c_str(), &dt, &nd, shape, &off);
            if (rc == 0) { ++parsed; CHECK(dt >= 0 && dt <= 3 && nd >= 0 && nd <= 4 && off >= 10, "a header that parses gives dtype %d ndim %d offset %lld", dt, nd, (long long)off); }
            else ++refused;
            int64_t want[4] = {hs[0], ws[0], 1, 1};
            (void)dd_npy_read(bad.c_str(), -1, 2, want, dst.data(), (int64_t)dst.size());
        }
        printf("header fuzz: %ld parsed, %ld refused\n", parsed, refused);
    }
    printf("prefetch_stress ok: %d rounds, %lld bytes compared\n", rounds, checked);
    return 0;
}
