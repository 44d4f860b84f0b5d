// obb_probe.cpp — CPU estimate of the packet walk's work per wave with axis-aligned child boxes (what the
// packet kernel uses) against oriented boxes (PCA frame of the child's vertices).  Tuning aid only.
//   g++ -O2 -std=c++17 -pthread -o /tmp/obb_probe tools/obb_probe.cpp && /tmp/obb_probe [res] [group]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <random>
#include <vector>

#include "../sdf_amd/csrc/kd_host.h"

using namespace sdfk;

struct Obb {
    double c[3], R[3][3], h[3];  // axes are the rows of R
};

static void jacobi3(double A[3][3], double V[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                double t = (th >= 0 ? 1 : -1) / (fabs(th) + sqrt(th * th + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq, A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk, A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq, V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

static const float* g_v;
static const KdHost* g_kd;

static Obb make_obb(uint32_t first, uint32_t count) {
    double m[3] = {0, 0, 0};
    int n = 0;
    for (uint32_t k = first; k < first + count; ++k) {
        int vi = g_kd->pos_vert[k];
        if (vi < 0) continue;
        for (int d = 0; d < 3; ++d) m[d] += g_v[3 * vi + d];
        ++n;
    }
    Obb o{};
    if (!n) return o;
    for (int d = 0; d < 3; ++d) m[d] /= n;
    double A[3][3] = {{0}};
    for (uint32_t k = first; k < first + count; ++k) {
        int vi = g_kd->pos_vert[k];
        if (vi < 0) continue;
        double x[3];
        for (int d = 0; d < 3; ++d) x[d] = g_v[3 * vi + d] - m[d];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) A[i][j] += x[i] * x[j];
    }
    double V[3][3];
    jacobi3(A, V);
    for (int a = 0; a < 3; ++a)
        for (int d = 0; d < 3; ++d) o.R[a][d] = V[d][a];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t k = first; k < first + count; ++k) {
        int vi = g_kd->pos_vert[k];
        if (vi < 0) continue;
        for (int a = 0; a < 3; ++a) {
            double p = 0;
            for (int d = 0; d < 3; ++d) p += o.R[a][d] * g_v[3 * vi + d];
            lo[a] = std::min(lo[a], p), hi[a] = std::max(hi[a], p);
        }
    }
    for (int a = 0; a < 3; ++a) o.h[a] = (hi[a] - lo[a]) / 2;
    for (int d = 0; d < 3; ++d) {
        o.c[d] = 0;
        for (int a = 0; a < 3; ++a) o.c[d] += o.R[a][d] * (lo[a] + hi[a]) / 2;
    }
    return o;
}

static double obb_lb(const Obb& o, const float* q) {
    double s = 0;
    for (int a = 0; a < 3; ++a) {
        double p = 0;
        for (int d = 0; d < 3; ++d) p += o.R[a][d] * (q[d] - o.c[d]);
        double g = fabs(p) - o.h[a];
        if (g > 0) s += g * g;
    }
    return s;
}
static double aabb_lb(const float* b, const float* q) {
    double s = 0;
    for (int d = 0; d < 3; ++d) {
        double g = std::max(std::max((double)b[d] - q[d], (double)q[d] - b[3 + d]), 0.0);
        s += g * g;
    }
    return s;
}

struct QInfo {
    uint32_t first[4], count[4];
    Obb obb[4];
};
static std::vector<QInfo> g_qi;

static void extent(uint32_t ref, uint32_t& first, uint32_t& count) {
    if (ref & KD_LEAF_BIT) {
        first = ((ref & KD_REF_MASK) >> 4) << 2, count = 4 * (ref & 15u);
        return;
    }
    const KdQuadH& q = g_kd->quads[ref];
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (int k = 0; k < 4; ++k) {
        if (q.ref[k] == KD_LEAF_BIT) continue;
        uint32_t f, c;
        extent(q.ref[k], f, c);
        g_qi[ref].first[k] = f, g_qi[ref].count[k] = c;
        g_qi[ref].obb[k] = make_obb(f, c);
        lo = std::min(lo, f), hi = std::max(hi, f + c);
    }
    first = lo, count = hi - lo;
}

struct Counters {
    double nodes = 0, groups = 0, slots = 0, boxtests = 0;
};
static int g_mode;  // 0 aabb, 1 obb, 2 max of both
static float g_pts[64][3];
static double g_best[64];

static void walk(uint32_t ref, Counters& C) {
    if (ref & KD_LEAF_BIT) {
        const uint32_t first = ((ref & KD_REF_MASK) >> 4) << 2, n = 4 * (ref & 15u);
        C.groups += 1;
        C.slots += n;
        for (uint32_t k = first; k < first + n; ++k) {
            int vi = g_kd->pos_vert[k];
            if (vi < 0) continue;
            for (int l = 0; l < 64; ++l) {
                double s = 0;
                for (int d = 0; d < 3; ++d) {
                    double g = g_pts[l][d] - g_v[3 * vi + d];
                    s += g * g;
                }
                if (s < g_best[l]) g_best[l] = s;
            }
        }
        return;
    }
    C.nodes += 1;
    const KdQuadH& q = g_kd->quads[ref];
    const QInfo& I = g_qi[ref];
    double lb[4][64];
    double key[4];
    bool act[4];
    for (int k = 0; k < 4; ++k) {
        act[k] = false;
        key[k] = 1e300;
        if (q.ref[k] == KD_LEAF_BIT) continue;
        C.boxtests += 1;
        double sum = 0;
        for (int l = 0; l < 64; ++l) {
            double a = aabb_lb(q.box[k], g_pts[l]);
            double o = obb_lb(I.obb[k], g_pts[l]);
            lb[k][l] = g_mode == 0 ? a : (g_mode == 1 ? o : std::max(a, o));
            if (lb[k][l] <= g_best[l]) act[k] = true;
            sum += lb[k][l];
        }
        key[k] = sum;
    }
    int ord[4] = {0, 1, 2, 3};
    std::sort(ord, ord + 4, [&](int a, int b) { return key[a] < key[b]; });
    for (int i = 0; i < 4; ++i) {
        const int k = ord[i];
        if (!act[k]) continue;
        bool still = false;
        for (int l = 0; l < 64; ++l)
            if (lb[k][l] <= g_best[l]) still = true;
        if (still) walk(q.ref[k], C);
    }
}


// ---- wide scheme: L1 node = W1 consecutive groups, tests against the packet's bounding sphere -------------
struct Grp { uint32_t ref, first, count; Obb obb; };
static std::vector<Grp> g_groups;
static void collect_groups(uint32_t ref) {
    if (ref & KD_LEAF_BIT) {
        if (ref == KD_LEAF_BIT) return;
        Grp g; g.ref = ref; g.first = ((ref & KD_REF_MASK) >> 4) << 2; g.count = 4 * (ref & 15u);
        g.obb = make_obb(g.first, g.count);
        g_groups.push_back(g);
        return;
    }
    for (int k = 0; k < 4; ++k) collect_groups(g_kd->quads[ref].ref[k]);
}

// distance to the box and its gradient at q (unit vector from the box towards q; zero inside)
static double obb_dist_grad(const Obb& o, const float* q, double n[3]) {
    double gl[3], s = 0;
    for (int a = 0; a < 3; ++a) {
        double p = 0;
        for (int d = 0; d < 3; ++d) p += o.R[a][d] * (q[d] - o.c[d]);
        double g = fabs(p) - o.h[a];
        gl[a] = g > 0 ? (p > 0 ? g : -g) : 0;
        s += gl[a] * gl[a];
    }
    s = sqrt(s);
    for (int d = 0; d < 3; ++d) {
        n[d] = 0;
        if (s > 0) for (int a = 0; a < 3; ++a) n[d] += o.R[a][d] * gl[a] / s;
    }
    return s;
}
static void wide_probe(int npk, double cell, const std::vector<std::array<float, 3>>& origins, int W1, bool sorted) {
    std::vector<Obb> l1;
    for (size_t g = 0; g < g_groups.size(); g += W1) {
        size_t e = std::min(g + W1, g_groups.size());
        l1.push_back(make_obb(g_groups[g].first, g_groups[e - 1].first + g_groups[e - 1].count - g_groups[g].first));
    }
    double n_l1 = 0, n_cand = 0, n_lin = 0, n_eval = 0, n_slots = 0, n_eval_lin = 0;
    std::mt19937 r2(11);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (int p = 0; p < npk; ++p) {
        float lo[3] = {1e9f, 1e9f, 1e9f}, hi[3] = {-1e9f, -1e9f, -1e9f};
        for (int l = 0; l < 64; ++l) {
            for (int d = 0; d < 3; ++d) {
                g_pts[l][d] = origins[p][d] + (float)cell * U(r2);
                lo[d] = std::min(lo[d], g_pts[l][d]), hi[d] = std::max(hi[d], g_pts[l][d]);
            }
            g_best[l] = 1e300;
        }
        float ctr[3]; double rho = 0;
        for (int d = 0; d < 3; ++d) ctr[d] = (lo[d] + hi[d]) / 2, rho += (hi[d] - lo[d]) * (hi[d] - lo[d]) / 4;
        rho = sqrt(rho);
        auto eval = [&](const Grp& G) {
            n_eval += 1; n_slots += G.count;
            for (uint32_t k = G.first; k < G.first + G.count; ++k) {
                int vi = g_kd->pos_vert[k];
                if (vi < 0) continue;
                for (int l = 0; l < 64; ++l) {
                    double s = 0;
                    for (int d = 0; d < 3; ++d) { double g = g_pts[l][d] - g_v[3 * vi + d]; s += g * g; }
                    if (s < g_best[l]) g_best[l] = s;
                }
            }
        };
        // greedy descent
        std::vector<std::pair<double, size_t>> o1;
        for (size_t j = 0; j < l1.size(); ++j) o1.push_back({obb_lb(l1[j], ctr), j});
        if (sorted) std::sort(o1.begin(), o1.end());
        size_t bj = std::min_element(o1.begin(), o1.end())->second;
        size_t bg = 0; double bd = 1e300;
        for (size_t g = bj * W1; g < std::min((bj + 1) * W1, g_groups.size()); ++g) { double d = obb_lb(g_groups[g].obb, ctr); if (d < bd) bd = d, bg = g; }
        eval(g_groups[bg]);
        auto thrP = [&]() { double T = 0; for (int l = 0; l < 64; ++l) T = std::max(T, g_best[l]); double r = sqrt(T) + rho; return r * r; };
        double tp = thrP();
        for (auto& e1 : o1) {
            if (e1.first > tp) continue;
            const size_t j = e1.second;
            n_l1 += 1;
            std::vector<std::pair<double, size_t>> o2;
            for (size_t g = j * W1; g < std::min((j + 1) * W1, g_groups.size()); ++g)
                if (g != bg) o2.push_back({obb_lb(g_groups[g].obb, ctr), g});
            if (sorted) std::sort(o2.begin(), o2.end());
            for (auto& e2 : o2) {
                if (e2.first > tp) continue;
                const size_t g = e2.second;
                n_cand += 1;
                double n[3];
                const double f = obb_dist_grad(g_groups[g].obb, ctr, n);
                bool lin = false, any = false;
                for (int l = 0; l < 64; ++l) {
                    double t = f;
                    for (int d = 0; d < 3; ++d) t += n[d] * (g_pts[l][d] - ctr[d]);
                    if (t <= sqrt(g_best[l])) lin = true;
                    if (obb_lb(g_groups[g].obb, g_pts[l]) <= g_best[l]) any = true;
                }
                if (!lin) continue;
                n_lin += 1;
                if (!any) { n_eval_lin += 1; if (g_mode == 1) continue; }
                eval(g_groups[g]);
                tp = thrP();
            }
        }
    }
    printf("wide W1=%d %s %s: %zu groups, %zu L1; per wave: L1 active %.1f  candidates %.1f  pass linear %.1f (exact test would drop %.1f)  evaluated %.1f  slots %.0f\n",
           W1, sorted ? "nearest-first" : "index order", g_mode == 1 ? "linear+exact" : "linear only", g_groups.size(), l1.size(), n_l1 / npk, n_cand / npk, n_lin / npk, n_eval_lin / npk, n_eval / npk, n_slots / npk);
}

int main(int argc, char** argv) {
    const int res = argc > 1 ? atoi(argv[1]) : 424;
    const int group = argc > 2 ? atoi(argv[2]) : 40;
    const double cell = argc > 3 ? atof(argv[3]) : 2.0 / 128;  // packet cube side
    const int npk = argc > 4 ? atoi(argv[4]) : 3000;
    // torus R = 0.7, r = 0.24 (sdf_amd/meshes.py)
    std::vector<float> v(3 * (size_t)res * res);
    for (int i = 0; i < res; ++i)
        for (int j = 0; j < res; ++j) {
            const double u = 2 * M_PI * i / res, w = 2 * M_PI * j / res;
            float* p = &v[3 * ((size_t)i * res + j)];
            p[0] = (float)((0.7 + 0.24 * cos(w)) * cos(u));
            p[1] = (float)((0.7 + 0.24 * cos(w)) * sin(u));
            p[2] = (float)(0.24 * sin(w));
        }
    KdHost kd;
    kd.group_max = group;
    kd.build(v.data(), (uint32_t)(res * res), 1);
    kd.build_quads();
    g_v = v.data();
    g_kd = &kd;
    g_qi.resize(kd.quads.size());
    uint32_t f, c;
    extent(kd.quad_root, f, c);
    printf("verts %d group %d: %zu quads, %zu positions, quad depth %d\n", res * res, group, kd.quads.size(),
           kd.pos_vert.size(), kd.quad_depth);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<std::array<float, 3>> origins(npk);
    for (auto& o : origins)
        for (int d = 0; d < 3; ++d) o[d] = -1.f + (2.f - (float)cell) * U(rng);
    for (g_mode = 0; g_mode < 3; ++g_mode) {
        Counters C;
        std::mt19937 r2(11);
        for (int p = 0; p < npk; ++p) {
            for (int l = 0; l < 64; ++l) {
                for (int d = 0; d < 3; ++d) g_pts[l][d] = origins[p][d] + (float)cell * U(r2);
                g_best[l] = 1e300;
            }
            walk(kd.quad_root, C);
        }
        printf("%s: per wave  quad nodes %.1f  box tests %.1f  groups %.1f  slots %.0f\n",
               g_mode == 0 ? "aabb" : (g_mode == 1 ? "obb " : "both"), C.nodes / npk, C.boxtests / npk,
               C.groups / npk, C.slots / npk);
    }
    collect_groups(kd.quad_root);
    g_mode = 0; wide_probe(npk, cell, origins, 64, false);
    wide_probe(npk, cell, origins, 64, true);
    g_mode = 1; wide_probe(npk, cell, origins, 64, true);
    return 0;
}
