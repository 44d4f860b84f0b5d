// The task-parallel kd build must produce byte-identical arrays to the serial build.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../sdf_amd/csrc/kd_host.h"
int main() {
    std::mt19937 rg(5);
    std::uniform_real_distribution<float> U(-1, 1);
    for (int n : {7, 11, 500, 25000, 180000}) {
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<float> p(3 * (size_t)n);
            for (auto& x : p) x = mode ? (float)(int)(U(rg) * 6) * 0.25f : U(rg);  // lattice: many equal keys
            sdfk::KdHost a, b;
            a.build(p.data(), n, 1);
            b.build(p.data(), n, 8);
            bool ok = a.root_ref == b.root_ref && a.depth == b.depth && a.n_leaves == b.n_leaves &&
                      a.vind == b.vind && a.pos_vert == b.pos_vert && a.split == b.split &&
                      a.nodes.size() == b.nodes.size() &&
                      !memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(sdfk::KdNodeH)) &&
                      !memcmp(a.bbox, b.bbox, sizeof(a.bbox));
            if (!ok) { printf("MISMATCH n=%d mode=%d\n", n, mode); return 1; }
        }
    }
    printf("ok\n");
    return 0;
}
