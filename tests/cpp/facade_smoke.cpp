// Compiles the C++ facade (no Eigen here) and calls it like a user of sdf::SDF would.
// exit 0: query ran and matched the analytic cube; exit 3: no GPU (the expected result in the
// CPU-only container: construction must fail loudly, there is no CPU fallback).
#include <cmath>
#include <cstdio>
#include <vector>

#include "sdf_amd/sdf.hpp"

int main() {
    const float h = 0.5f;
    std::vector<float> v = {-h, -h, -h, h, -h, -h, h, h, -h, -h, h, -h, -h, -h, h, h, -h, h, h, h, h, -h, h, h};
    std::vector<uint32_t> f = {0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4,
                               2, 3, 7, 2, 7, 6, 1, 2, 6, 1, 6, 5, 3, 0, 4, 3, 4, 7};
    try {
        sdf::SDF s(v.data(), 8, f.data(), 12, /*robust=*/true, /*copy=*/true);
        std::vector<float> p = {0.1f, 0.2f, 0.05f, 0.9f, 0.f, 0.f};
        auto d = s(p.data(), 2);
        auto c = s.contains(p.data(), 2);
        auto n = s.nn(p.data(), 2);
        std::printf("sdf %g %g contains %d %d nn %d %d area %g\n", d[0], d[1], c[0], c[1], n[0], n[1],
                    s.surface_area());
        const bool ok = std::fabs(d[0] - 0.3f) < 1e-6f && std::fabs(d[1] + 0.4f) < 1e-6f && c[0] == 1 &&
                        c[1] == 0 && s.robust && s.own_data && s.adj_faces(0).size() > 0;
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
}
