// Times the host kd build (serial, previous task-parallel form is gone; threads = argv[1..]) on the
// 424 x 424 torus-sized vertex set and checks the parallel result against the serial one.
//   g++ -std=c++17 -O3 -pthread tools/kd_build_probe.cpp -o /tmp/kdprobe && /tmp/kdprobe 2 4 8 16
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../sdf_amd/csrc/kd_host.h"

int main(int argc, char** argv) {
    const int res = 424, n = res * res;
    std::vector<float> p(3 * (size_t)n);
    for (int i = 0; i < res; ++i)
        for (int j = 0; j < res; ++j) {
            const double u = 2 * M_PI * i / res, v = 2 * M_PI * j / res;
            float* q = &p[3 * ((size_t)i * res + j)];
            q[0] = (float)((0.7 + 0.24 * std::cos(v)) * std::cos(u));
            q[1] = (float)((0.7 + 0.24 * std::cos(v)) * std::sin(u));
            q[2] = (float)(0.24 * std::sin(v));
        }
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    sdfk::KdHost a;
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        a.build(p.data(), n, 1);
        best = std::min(best, ms(t0, std::chrono::steady_clock::now()));
    }
    {
        auto t1 = std::chrono::steady_clock::now();
        a.build_groups();
        std::printf("serial            %7.2f ms   groups %.2f ms\n", best, ms(t1, std::chrono::steady_clock::now()));
    }
    for (int k = 1; k < argc; ++k) {
        const int th = std::atoi(argv[k]);
        sdfk::KdHost b;
        best = 1e9;
        double bestq = 1e9;
        for (int r = 0; r < 8; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            b.build(p.data(), n, th);
            best = std::min(best, ms(t0, std::chrono::steady_clock::now()));
            auto t1 = std::chrono::steady_clock::now();
            b.build_groups();
            bestq = std::min(bestq, ms(t1, std::chrono::steady_clock::now()));
        }
        const bool same = a.vind == b.vind && a.pos_vert == b.pos_vert && a.split == b.split &&
                          a.nodes.size() == b.nodes.size() &&
                          !std::memcmp(a.nodes.data(), b.nodes.data(), a.nodes.size() * sizeof(sdfk::KdNodeH));
        std::printf("threads %2d        %7.2f ms   %s   (last: top %.2f ranges %.2f splice %.2f)  groups %.2f ms\n", th, best,
                    same ? "identical" : "MISMATCH", b.t_top_ms, b.t_ranges_ms, b.t_splice_ms, bestq);
    }
    return 0;
}
