// Compiles the C++ facade (no Eigen here) and calls it like a user of sdf::SDF would.
// exit 0: SDF and Renderer queries ran and matched the analytic cube; exit 3: no GPU (the expected result in the
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
        // sdf::Renderer: the cube 3 units in front of a 64 x 64 pinhole camera
        std::vector<float> vc = v;
        for (size_t i = 2; i < vc.size(); i += 3) vc[i] += 3.0f;
        sdf::Renderer r(vc.data(), 8, f.data(), 12, 64, 64, 100.f, 100.f, 32.f, 32.f, /*copy=*/true);
        auto depth = r.render_depth_vec();
        auto mask = r.render_mask_vec();
        std::vector<float> q = {32.f, 32.f, 1.f, 1.f};
        auto dq = r(q.data(), 2);
        auto vq = r.nn(q.data(), 2, /*fill_outside=*/true);
        std::printf("renderer centre depth %g corner %g mask %d %d nn %d %d\n", dq[0], dq[1], mask[32 * 64 + 32],
                    mask[0], vq[0], vq[1]);
        const bool rok = depth.size() == 64 * 64 && std::fabs(dq[0] - 2.5f) < 1e-5f && dq[1] == 0.f &&
                         mask[32 * 64 + 32] == 1 && mask[0] == 0 && vq[0] >= 0 && vq[1] >= 0 && r.own_data;
        return ok && rok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 3;
    }
}
