// ddarena_plan.h -- the layout planning of the HBM zone arena (ddarena.hip): which class every chunk of every array of a
// request is taken from, given how many free chunks of each class are at hand.  Plain C++ without any HIP, so that the
// CPU suite can compile and test it (tests/c_client/arena_plan_test.cpp).
#pragma once

#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "ddcore.h"      // DD_ARENA_ROTATED, DD_ARENA_BLOCKED

namespace ddarena_plan {

constexpr int MAX_CLASSES = 3;

// Best assignment of groups to classes for the chunks at hand: maximise the chunks served from a group's own class; groups
// that were given a class by an earlier call keep it (`group_class[g]` >= 0).  `need[g]`: chunks the class-pure arrays of
// group g want; `fixed[c]`: chunks the rotated arrays of the request want from class c, whatever the assignment.
// Returns the chunks served, -1 when the sticky classes admit no permutation.
inline int best_assignment(const int avail[MAX_CLASSES], const int group_class[MAX_CLASSES], const int need[MAX_CLASSES],
                           const int fixed[MAX_CLASSES], int perm_out[MAX_CLASSES]) {
    int p[MAX_CLASSES] = {0, 1, 2}, best = -1;
    do {
        bool ok = true;
        for (int g = 0; g < MAX_CLASSES; ++g) if (group_class[g] >= 0 && group_class[g] != p[g]) ok = false;
        if (!ok) continue;
        int want[MAX_CLASSES];
        for (int c = 0; c < MAX_CLASSES; ++c) want[c] = fixed[c];
        for (int g = 0; g < MAX_CLASSES; ++g) want[p[g]] += need[g];
        int served = 0;
        for (int c = 0; c < MAX_CLASSES; ++c) served += std::min(want[c], avail[c]);
        if (served > best) { best = served; memcpy(perm_out, p, sizeof(p)); }
    } while (std::next_permutation(p, p + MAX_CLASSES));
    return best;
}

// Class-pure arrays (layout 0..2) take the class of their group (`perm`).  Rotated arrays (DD_ARENA_ROTATED + phase) are
// laid out index by index: at chunk index k the arrays, in phase order, each take the class with the most free chunks left
// that no other rotated array of the request uses at k (ties go to (phase + k) mod 3, so balanced supplies give the exact
// rotation, and two plentiful classes give two class-pure arrays in different classes).  `missing`: chunks nobody can
// supply yet; `conflicts`: chunks of a class-pure array taken outside its class, and chunk indices where one of the first
// TWO rotated arrays (the lock-step store streams) had to share a class -- the third (colours) sharing one is harmless.
inline void plan_classes(const int avail_in[MAX_CLASSES], int n, const std::vector<int> &nch, const int32_t *layouts,
                         const int perm[MAX_CLASSES], std::vector<std::vector<int>> &choice, int *missing, int *conflicts) {
    int avail[MAX_CLASSES];
    for (int c = 0; c < MAX_CLASSES; ++c) avail[c] = avail_in[c];
    choice.assign(n, std::vector<int>());
    *missing = 0; *conflicts = 0;
    auto any_class = [&]() { int b = -1; for (int c = 0; c < MAX_CLASSES; ++c) if (avail[c] > 0 && (b < 0 || avail[c] > avail[b])) b = c; return b; };
    for (int i = 0; i < n; ++i) {
        if (layouts[i] < 0 || layouts[i] >= MAX_CLASSES) continue;
        for (int k = 0; k < nch[i]; ++k) {
            int c = perm[layouts[i]];
            if (avail[c] <= 0) { c = any_class(); if (c >= 0) *conflicts += 1; }
            if (c < 0) { *missing += 1; choice[i].push_back(-1); continue; }
            avail[c] -= 1;
            choice[i].push_back(c);
        }
    }
    // Blocked arrays (DD_ARENA_BLOCKED): the first, middle and last third of the array's chunks each from a class of its own
    // (the thirds go to the classes in the order of their supply: the driver hands out long stretches of one class, so a
    // blocked array is what it gives most readily).  A chunk that has to come from another class than its third's counts
    // as a conflict (scouting goes on while the budget lasts).
    for (int i = 0; i < n; ++i) {
        if (layouts[i] != DD_ARENA_BLOCKED) continue;
        int order[MAX_CLASSES] = {0, 1, 2};
        std::sort(order, order + MAX_CLASSES, [&](int x, int y) { return avail[x] != avail[y] ? avail[x] > avail[y] : x < y; });
        for (int k = 0; k < nch[i]; ++k) {
            const int third = std::min(MAX_CLASSES - 1, (int)(((long long)k * MAX_CLASSES) / std::max(nch[i], 1)));
            int c = order[third];
            if (avail[c] <= 0) { c = any_class(); if (c >= 0) *conflicts += 1; }
            if (c < 0) { *missing += 1; choice[i].push_back(-1); continue; }
            avail[c] -= 1;
            choice[i].push_back(c);
        }
    }
    std::vector<int> rot;
    for (int ph = 0; ph < MAX_CLASSES; ++ph) for (int i = 0; i < n; ++i) if (layouts[i] == DD_ARENA_ROTATED + ph) rot.push_back(i);
    int kmax = 0;
    for (int i : rot) kmax = std::max(kmax, nch[i]);
    for (int k = 0; k < kmax; ++k) {
        bool used[MAX_CLASSES] = {false, false, false};
        int order = 0;
        for (int i : rot) {
            const int my = order++;
            if (k >= nch[i]) continue;
            const int pref = (layouts[i] - DD_ARENA_ROTATED + k) % MAX_CLASSES;
            int c = -1;
            for (int d = 0; d < MAX_CLASSES; ++d) {
                const int cand = (pref + d) % MAX_CLASSES;
                if (avail[cand] > 0 && !used[cand] && (c < 0 || avail[cand] > avail[c])) c = cand;
            }
            if (c < 0) {                     // every class that still has chunks is taken at this index
                c = any_class();
                if (c >= 0 && my < 2) *conflicts += 1;
            }
            if (c < 0) { *missing += 1; choice[i].push_back(-1); continue; }
            avail[c] -= 1;
            used[c] = true;
            choice[i].push_back(c);
        }
    }
}

}  // namespace ddarena_plan
