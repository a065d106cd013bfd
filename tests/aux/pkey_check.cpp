// Host check of the packed 4D sort keys (insmos_amd/csrc/common.h: pkey_make / pkey_expand): over random voxels inside the
// packed box, the 40-bit packed key orders EXACTLY like the canonical 64-bit key (key4_encode) and expands back to it.
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "../../insmos_amd/csrc/common.h"
using namespace insmos;
int main() {
    std::mt19937_64 rng(7);
    for (int B : {1, 2, 3, 8}) {
        std::vector<std::pair<uint64_t, uint64_t>> v;  // (packed, canonical)
        auto add = [&](int x, int y, int z, int tq, int b) {
            const uint64_t p = pkey_make((tq + 15) * B + b, x, y, z);
            const uint64_t c = key4_encode(x, y, z, tq * B + b);
            if (p >> 40) { printf("FAIL packed key wider than 40 bits\n"); exit(1); }
            if (pkey_expand(p, B) != c) { printf("FAIL expand B=%d (%d %d %d %d %d)\n", B, x, y, z, tq, b); exit(1); }
            v.push_back({p, c});
        };
        for (int i = 0; i < 200000; ++i)
            add((int)(rng() % 4096) - 2048, (int)(rng() % 4096) - 2048, (int)(rng() % 512) - 256, -(int)(rng() % 16), (int)(rng() % B));
        for (int x : {-2048, -1, 0, 2047}) for (int y : {-2048, -1, 0, 2047}) for (int z : {-256, -1, 0, 255})
            for (int tq : {-15, 0}) for (int b = 0; b < B; ++b) add(x, y, z, tq, b);
        // small dense cube: every bit boundary
        for (int x = -9; x < 9; ++x) for (int y = -9; y < 9; ++y) for (int z = -9; z < 9; ++z) add(x * 113, y * 57, z * 14, -(x & 15 ? (x + 9) % 16 : 0), (x + y + 18) % B);
        auto byp = v, byc = v;
        std::sort(byp.begin(), byp.end(), [](auto& a, auto& b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
        std::sort(byc.begin(), byc.end(), [](auto& a, auto& b) { return a.second < b.second || (a.second == b.second && a.first < b.first); });
        for (size_t i = 0; i < v.size(); ++i)
            if (byp[i] != byc[i]) { printf("FAIL order B=%d at %zu\n", B, i); return 1; }
    }
    printf("OK\n");
    return 0;
}
