// The task-parallel kd build must produce byte-identical arrays to the serial build, and the
// packet groups (maximal subtrees of <= group_max vertices, in position order) must tile the padded
// position space exactly.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../sdf_amd/csrc/kd_host.h"

static bool groups_ok(const sdfk::KdHost& k, int n) {
    std::vector<int> seen(k.pos_vert.size(), 0);
    std::vector<int> vseen(n, 0);
    auto mark = [&](uint32_t ref) {
        const uint32_t first = ((ref & sdfk::KD_REF_MASK) >> 4) << 2, chunks = ref & 15u;
        if (first + 4 * chunks > k.pos_vert.size()) return false;
        for (uint32_t i = first; i < first + 4 * chunks; ++i) seen[i]++;
        return true;
    };
    uint32_t next = 0;  // groups are listed in position order and tile the padded positions
    for (uint32_t ref : k.groups) {
        if (!(ref & sdfk::KD_LEAF_BIT) || !mark(ref)) return false;
        const uint32_t first = ((ref & sdfk::KD_REF_MASK) >> 4) << 2, chunks = ref & 15u;
        if (first != next) return false;
        next = first + 4 * chunks;
        int real = 0;
        for (uint32_t i = first; i < first + 4 * chunks; ++i) real += k.pos_vert[i] >= 0;
        // (an empty leaf, which nanoflann can produce on degenerate splits, is no group at all)
        if (real == 0 || real > (int)k.group_max) return false;
    }
    for (size_t i = 0; i < seen.size(); ++i)
        if (seen[i] != 1) return false;
    for (int32_t v : k.pos_vert)
        if (v >= 0) {
            if (v >= n || vseen[v]++) return false;
        }
    for (int i = 0; i < n; ++i)
        if (!vseen[i]) return false;
    return k.pos_vert.size() % 4 == 0;
}

int main() {
    std::mt19937 rg(5);
    std::uniform_real_distribution<float> U(-1, 1);
    for (int n : {7, 11, 500, 25000, 180000}) {
        for (int mode = 0; mode < 6; ++mode) {
            std::vector<float> p(3 * (size_t)n);
            for (auto& x : p) x = mode == 1 ? (float)(int)(U(rg) * 6) * 0.25f : U(rg);  // 1: lattice, many equal keys
            if (mode == 2)  // all points equal
                for (auto& x : p) x = 0.5f;
            if (mode == 3)  // sorted along x, two values of y
                for (int i = 0; i < n; ++i) p[3 * i] = (float)i / n, p[3 * i + 1] = (i & 1) ? 0.f : -0.f;
            if (mode == 4)  // reverse sorted along z, x and y constant
                for (int i = 0; i < n; ++i) p[3 * i] = 1.f, p[3 * i + 1] = 2.f, p[3 * i + 2] = (float)(n - i);
            if (mode == 5)  // two tight clusters and a few strays
                for (int i = 0; i < n; ++i)
                    for (int d = 0; d < 3; ++d)
                        p[3 * i + d] = (i % 97 == 0) ? U(rg) * 50 : ((i & 1) ? 10.f : -10.f) + U(rg) * 1e-3f;
            for (uint32_t g : {10u, 16u, 32u, 60u}) {
                sdfk::KdHost a, b;
                a.group_max = b.group_max = g;
                a.build(p.data(), n, 1);
                b.build(p.data(), n, g == 16u ? 3 : (g == 32u ? 16 : 8));
                b.build(p.data(), n, g == 16u ? 3 : (g == 32u ? 16 : 8));  // (a rebuild reuses the team)
                a.build_groups();
                b.build_groups();
                bool ok = a.root_ref == b.root_ref && a.depth == b.depth && a.vind == b.vind &&
                          a.pos_vert == b.pos_vert && a.split == b.split && a.nodes.size() == b.nodes.size() &&
                          !memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(sdfk::KdNodeH)) &&
                          !memcmp(a.bbox, b.bbox, sizeof(a.bbox)) && a.groups == b.groups;
                if (!ok) { printf("MISMATCH n=%d mode=%d g=%u\n", n, mode, g); return 1; }
                if (!groups_ok(a, n)) { printf("BAD GROUPS n=%d mode=%d g=%u\n", n, mode, g); return 1; }
            }
        }
    }
    printf("ok\n");
    return 0;
}
