// CPU test of the arena's layout planning (depthdensifier_amd/csrc/ddarena_plan.h): compiled with g++ by
// tests/test_host_cpu.py, no GPU and no HIP involved.  Prints "plan OK" and returns 0, or says what failed.
#include <stdio.h>

#include <vector>

#include "ddarena_plan.h"

using namespace ddarena_plan;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { printf("FAILED line %d: ", __LINE__); printf(__VA_ARGS__); printf("\n"); ++failures; } } while (0)

struct Plan { std::vector<std::vector<int>> choice; int missing, conflicts; };

static Plan plan(const int avail[3], const std::vector<int> &nch, const std::vector<int32_t> &layouts, const int perm[3]) {
    Plan p;
    plan_classes(avail, (int)nch.size(), nch, layouts.data(), perm, p.choice, &p.missing, &p.conflicts);
    return p;
}

int main() {
    const int ident[3] = {0, 1, 2}, none[3] = {-1, -1, -1};
    const int R = DD_ARENA_ROTATED;

    {   // supply that stays balanced: three rotated arrays get the exact rotation (ties go to (phase + k) mod 3)
        const int avail[3] = {6, 6, 6};
        Plan p = plan(avail, {6, 6, 6}, {R + 0, R + 1, R + 2}, ident);
        CHECK(p.missing == 0 && p.conflicts == 0, "missing %d conflicts %d", p.missing, p.conflicts);
        for (int k = 0; k < 6; ++k)
            for (int a = 0; a < 3; ++a) CHECK(p.choice[a][k] == (a + k) % 3, "array %d chunk %d in class %d", a, k, p.choice[a][k]);
    }
    {   // the cloud's shape (points 7, normals 7, colours 2 chunks) on an even supply: the lock-step pair never shares a class
        const int avail[3] = {7, 7, 7};
        Plan p = plan(avail, {7, 7, 2}, {R + 0, R + 1, R + 2}, ident);
        CHECK(p.missing == 0 && p.conflicts == 0, "missing %d conflicts %d", p.missing, p.conflicts);
        for (int k = 0; k < 7; ++k) CHECK(p.choice[0][k] != p.choice[1][k], "chunk %d shared class %d", k, p.choice[0][k]);
        for (int k = 0; k < 2; ++k) CHECK(p.choice[2][k] != p.choice[0][k] && p.choice[2][k] != p.choice[1][k], "colours chunk %d shares", k);
        int taken[3] = {0, 0, 0};
        for (auto &v : p.choice) for (int c : v) taken[c] += 1;
        for (int c = 0; c < 3; ++c) CHECK(taken[c] <= 7, "class %d gave %d of 7", c, taken[c]);
    }
    {   // two plentiful classes and an empty one: two class-pure arrays in DIFFERENT classes; the third array's sharing is harmless
        const int avail[3] = {9, 0, 9};
        Plan p = plan(avail, {7, 7, 2}, {R + 0, R + 1, R + 2}, ident);
        CHECK(p.missing == 0 && p.conflicts == 0, "missing %d conflicts %d", p.missing, p.conflicts);
        for (int k = 0; k < 7; ++k) CHECK(p.choice[0][k] != p.choice[1][k] && p.choice[0][k] != 1 && p.choice[1][k] != 1, "chunk %d: %d %d", k, p.choice[0][k], p.choice[1][k]);
    }
    {   // one class only: the first two arrays must share -> one conflict per chunk index, nothing missing
        const int avail[3] = {0, 20, 0};
        Plan p = plan(avail, {4, 4, 1}, {R + 0, R + 1, R + 2}, ident);
        CHECK(p.missing == 0, "missing %d", p.missing);
        CHECK(p.conflicts == 4, "conflicts %d (one per chunk index of the lock-step pair)", p.conflicts);
    }
    {   // not enough chunks at all: the shortfall is counted, the chunks that exist are still laid out
        const int avail[3] = {2, 2, 1};
        Plan p = plan(avail, {3, 3}, {R + 0, R + 1}, ident);
        CHECK(p.missing == 1, "missing %d", p.missing);
        int given = 0;
        for (auto &v : p.choice) for (int c : v) given += c >= 0;
        CHECK(given == 5, "given %d", given);
    }
    {   // scarce third class (1 chunk): the adaptive order keeps the lock-step pair apart at every index
        const int avail[3] = {10, 10, 1};
        Plan p = plan(avail, {7, 7, 2}, {R + 0, R + 1, R + 2}, ident);
        CHECK(p.missing == 0 && p.conflicts == 0, "missing %d conflicts %d", p.missing, p.conflicts);
        for (int k = 0; k < 7; ++k) CHECK(p.choice[0][k] != p.choice[1][k], "chunk %d shared class %d", k, p.choice[0][k]);
    }
    {   // class-pure groups follow the permutation; a group whose class has run dry takes another one and that is a conflict
        const int perm[3] = {2, 0, 1};
        const int avail[3] = {3, 3, 1};
        Plan p = plan(avail, {2, 2, 2}, {0, 1, 2}, perm);
        CHECK(p.choice[0][0] == 2 && p.choice[1][0] == 0 && p.choice[1][1] == 0 && p.choice[2][0] == 1 && p.choice[2][1] == 1, "pure arrays outside their classes");
        CHECK(p.choice[0][1] != 2 && p.conflicts == 1 && p.missing == 0, "second chunk of group 0: class %d, conflicts %d", p.choice[0][1], p.conflicts);
    }
    {   // best_assignment: serves the large group from the large class, and keeps a sticky group where it is
        const int avail[3] = {1, 8, 2}, need[3] = {8, 2, 1}, fixed[3] = {0, 0, 0};
        int perm[3] = {-1, -1, -1};
        const int served = best_assignment(avail, none, need, fixed, perm);
        CHECK(served == 11 && perm[0] == 1 && perm[1] == 2 && perm[2] == 0, "served %d perm %d %d %d", served, perm[0], perm[1], perm[2]);
        const int sticky[3] = {0, -1, -1};          // group 0 was given class 0 by an earlier call
        const int served2 = best_assignment(avail, sticky, need, fixed, perm);
        CHECK(perm[0] == 0 && served2 == 1 + 2 + 1, "sticky: served %d perm %d %d %d", served2, perm[0], perm[1], perm[2]);
        const int clash[3] = {0, 0, -1};            // cannot be: two groups in one class -> no permutation
        CHECK(best_assignment(avail, clash, need, fixed, perm) == -1, "clashing sticky classes accepted");
    }
    {   // rotated wants (fixed) count against the same supply as the groups
        const int avail[3] = {4, 4, 4}, need[3] = {3, 0, 0}, fixed[3] = {3, 1, 1};
        int perm[3];
        best_assignment(avail, none, need, fixed, perm);
        CHECK(perm[0] != 0, "group 0 sent to class %d where the rotated arrays already want 3 of 4", perm[0]);
    }
    {   // a blocked array: its thirds in three classes, the largest supply first; uneven supply spills over and is counted
        const int avail[3] = {5, 9, 7};
        Plan p = plan(avail, {9, 2}, {DD_ARENA_BLOCKED, 2}, ident);        // + a small class-pure array of group 2
        CHECK(p.missing == 0 && p.conflicts == 0, "missing %d conflicts %d", p.missing, p.conflicts);
        for (int k = 0; k < 9; ++k) CHECK(p.choice[0][k] == (k < 3 ? 1 : k < 6 ? 0 : 2), "blocked chunk %d in class %d", k, p.choice[0][k]);   // (the group-2 array took 2 of class 2 first: supplies 5, 9, 5)
        const int thin[3] = {1, 20, 1};
        Plan q = plan(thin, {9}, {DD_ARENA_BLOCKED}, ident);
        CHECK(q.missing == 0 && q.conflicts == 4, "thin supply: missing %d conflicts %d", q.missing, q.conflicts);   // thirds two and three find one chunk each
        const int two[3] = {1, 1, 0};
        Plan r = plan(two, {2}, {DD_ARENA_BLOCKED}, ident);                 // fewer chunks than classes: still laid out
        CHECK(r.missing == 0 && r.choice[0].size() == 2 && r.choice[0][0] != r.choice[0][1], "two-chunk blocked array");
    }
    if (failures) { printf("%d checks failed\n", failures); return 1; }
    printf("plan OK\n");
    return 0;
}
