// oracle/sdf_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the batched point-query path of
// sxyu/sdf (SDF::operator(), contains(), nn()).  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library; the product (sdf_amd/) never
// links, imports or calls it.
//
// Every function cites the reference file:line it restates.  "sdf.cpp" = src/sdf.cpp,
// "sdf.hpp" = include/sdf/sdf.hpp, "nanoflann.hpp" / "RTree.h" / "sdf_util.hpp" =
// include/sdf/internal/*, "util.cpp" = src/util.cpp, all under /root/reference.
//
// Parity status:
//   * kd-tree build + search (nn): PINNED.  Checked against the reference's real vendored
//     nanoflann.hpp (compiled by oracle/Makefile into oracle/_ref) and against the golden
//     vectors under tests/golden/ generated from it (tools/make_golden.py).
//   * ray candidate sets (R-tree Overlap semantics): PINNED the same way against the real
//     vendored RTree.h.
//   * Eigen fp32 expression association (dot / norm / row*3x3): PARITY UNPINNED.  Eigen is
//     absent from the image, so src/sdf.cpp itself cannot be built.  3-term reductions are
//     fixed here BY DEFINITION as c0 + (c1 + c2) (Eigen 3.3 redux_novec_unroller order, see
//     SURVEY.md §8 a8); nanoflann's metric is ((0 + c0) + c1) + c2 (nanoflann.hpp:432-440).
//
// Build: g++ -O3 -funroll-loops -ffp-contract=off (reference flags, CMakeLists.txt:33, plus
// contraction off because the host has FMA and the reference's x86-64 build has none).

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace orc {

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

static inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
// Eigen 3.3 3-term reduction order: c0 + (c1 + c2)   
static inline float sum3(float c0, float c1, float c2) { return c0 + (c1 + c2); }
static inline float dot(V3 a, V3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float sqnorm(V3 a) { return dot(a, a); }
// Eigen cross(): one subtraction of two products per component.
static inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// row-vector * 3x3 row-major matrix (lazy coefficient product): out_j = sum_k v_k R(k,j)
static inline V3 mulR(V3 v, const float* R) {
    return {sum3(v.x * R[0], v.y * R[3], v.z * R[6]), sum3(v.x * R[1], v.y * R[4], v.z * R[7]),
            sum3(v.x * R[2], v.y * R[5], v.z * R[8])};
}
static inline float stdmin(float a, float b) { return (b < a) ? b : a; }  // std::min
static inline float stdmax(float a, float b) { return (a < b) ? b : a; }  // std::max

// sdf.hpp:64-73  dist_point2lineseg (squared)
static inline float dist_point2lineseg(V3 p, V3 a, V3 b) {
    V3 ap = sub(p, a), ab = sub(b, a);
    float t = dot(ap, ab) / sqnorm(ab);
    t = stdmax(0.0f, stdmin(1.0f, t));
    V3 r = {ap.x - t * ab.x, ap.y - t * ab.y, ap.z - t * ab.z};
    return sqnorm(r);
}

// sdf.hpp:88-104 bary + sdf.hpp:110-127 dist_point2tri (squared)
static inline float dist_point2tri(V3 p, V3 a, V3 b, V3 c, V3 n, float area) {
    float area_pbc = dot(n, cross(sub(b, p), sub(c, p)));
    float area_pca = dot(n, cross(sub(c, p), sub(a, p)));
    float u = area_pbc / area;
    float v = area_pca / area;
    float w = (1.0f - u) - v;
    if (u < 0) return dist_point2lineseg(p, b, c);
    if (v < 0) return dist_point2lineseg(p, a, c);
    if (w < 0) return dist_point2lineseg(p, a, b);
    V3 r = {((u * a.x + v * b.x) + w * c.x) - p.x, ((u * a.y + v * b.y) + w * c.y) - p.y,
            ((u * a.z + v * b.z) + w * c.z) - p.z};
    return sqnorm(r);
}

// sdf.hpp:130-142 bary2d; returns true iff all three coordinates >= 0 (sdf.cpp:310)
static inline bool bary2d_inside(float px, float py, float ax, float ay, float bx, float by,
                                 float cx, float cy) {
    float v0x = bx - ax, v0y = by - ay, v1x = cx - ax, v1y = cy - ay, v2x = px - ax, v2y = py - ay;
    float invden = 1.f / (v0x * v1y - v1x * v0y);
    float b1 = (v2x * v1y - v1x * v2y) * invden;
    float b2 = (v0x * v2y - v2x * v0y) * invden;
    float b0 = 1.0f - (b1 + b2);
    return b0 >= 0.f && b1 >= 0.f && b2 >= 0.f;
}

// ---------------------------------------------------------------------------------------
// kd-tree: restatement of nanoflann v1.3.0 as configured by sdf_util.hpp:26-41
// (L2_Simple metric, DIM=3, IndexType=int, leaf_max_size=10).
// ---------------------------------------------------------------------------------------
struct KdNode {
    int child1 = -1, child2 = -1;  // -1/-1 => leaf
    int left = 0, right = 0;       // leaf: [left,right) into vind
    int divfeat = 0;
    float divlow = 0, divhigh = 0;
};
struct Interval {
    float low, high;
};
struct Box {
    Interval d[3];
};

struct KdTree {
    const float* pts = nullptr;  // [n,3]
    int n = 0;
    std::vector<int> vind;
    std::vector<KdNode> nodes;
    int root = -1;
    Box root_bbox;
    int max_depth = 0;
    static constexpr int LEAF_MAX = 10;

    float get(int idx, int dim) const { return pts[3 * (size_t)idx + dim]; }

    // nanoflann.hpp:836-848
    void minmax(const int* ind, int count, int dim, float& mn, float& mx) const {
        mn = mx = get(ind[0], dim);
        for (int i = 1; i < count; ++i) {
            float v = get(ind[i], dim);
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    }

    // nanoflann.hpp:967-1003  (IndexType = int; the "!right" guards are kept)
    void plane_split(int* ind, int count, int cutfeat, float cutval, int& lim1, int& lim2) const {
        int left = 0, right = count - 1;
        for (;;) {
            while (left <= right && get(ind[left], cutfeat) < cutval) ++left;
            while (right && left <= right && get(ind[right], cutfeat) >= cutval) --right;
            if (left > right || !right) break;
            std::swap(ind[left], ind[right]);
            ++left;
            --right;
        }
        lim1 = left;
        right = count - 1;
        for (;;) {
            while (left <= right && get(ind[left], cutfeat) <= cutval) ++left;
            while (right && left <= right && get(ind[right], cutfeat) > cutval) --right;
            if (left > right || !right) break;
            std::swap(ind[left], ind[right]);
            ++left;
            --right;
        }
        lim2 = left;
    }

    // nanoflann.hpp:909-956
    void middle_split(int* ind, int count, int& index, int& cutfeat, float& cutval,
                      const Box& bbox) const {
        const float EPS = 0.00001f;
        float max_span = bbox.d[0].high - bbox.d[0].low;
        for (int i = 1; i < 3; ++i) {
            float span = bbox.d[i].high - bbox.d[i].low;
            if (span > max_span) max_span = span;
        }
        float max_spread = -1;
        cutfeat = 0;
        for (int i = 0; i < 3; ++i) {
            float span = bbox.d[i].high - bbox.d[i].low;
            if (span > (1 - EPS) * max_span) {
                float mn, mx;
                minmax(ind, count, i, mn, mx);
                float spread = mx - mn;
                if (spread > max_spread) {
                    cutfeat = i;
                    max_spread = spread;
                }
            }
        }
        float split_val = (bbox.d[cutfeat].low + bbox.d[cutfeat].high) / 2;
        float mn, mx;
        minmax(ind, count, cutfeat, mn, mx);
        if (split_val < mn)
            cutval = mn;
        else if (split_val > mx)
            cutval = mx;
        else
            cutval = split_val;
        int lim1, lim2;
        plane_split(ind, count, cutfeat, cutval, lim1, lim2);
        if (lim1 > count / 2)
            index = lim1;
        else if (lim2 < count / 2)
            index = lim2;
        else
            index = count / 2;
    }

    // nanoflann.hpp:857-907
    int divide(int left, int right, Box& bbox, int depth) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        if (depth > max_depth) max_depth = depth;
        if ((right - left) <= LEAF_MAX) {
            nodes[id].left = left;
            nodes[id].right = right;
            for (int i = 0; i < 3; ++i) bbox.d[i].low = bbox.d[i].high = get(vind[left], i);
            for (int k = left + 1; k < right; ++k)
                for (int i = 0; i < 3; ++i) {
                    float v = get(vind[k], i);
                    if (bbox.d[i].low > v) bbox.d[i].low = v;
                    if (bbox.d[i].high < v) bbox.d[i].high = v;
                }
        } else {
            int idx, cutfeat;
            float cutval;
            middle_split(&vind[0] + left, right - left, idx, cutfeat, cutval, bbox);
            nodes[id].divfeat = cutfeat;
            Box lb = bbox;
            lb.d[cutfeat].high = cutval;
            int c1 = divide(left, left + idx, lb, depth + 1);
            Box rb = bbox;
            rb.d[cutfeat].low = cutval;
            int c2 = divide(left + idx, right, rb, depth + 1);
            nodes[id].child1 = c1;
            nodes[id].child2 = c2;
            nodes[id].divlow = lb.d[cutfeat].high;
            nodes[id].divhigh = rb.d[cutfeat].low;
            for (int i = 0; i < 3; ++i) {
                bbox.d[i].low = std::min(lb.d[i].low, rb.d[i].low);
                bbox.d[i].high = std::max(lb.d[i].high, rb.d[i].high);
            }
        }
        return id;
    }

    // nanoflann.hpp:1190-1202, 1307-1337
    void build(const float* p, int count) {
        pts = p;
        n = count;
        vind.resize(n);
        for (int i = 0; i < n; ++i) vind[i] = i;
        nodes.clear();
        root = -1;
        max_depth = 0;
        if (n == 0) return;
        for (int i = 0; i < 3; ++i) root_bbox.d[i].low = root_bbox.d[i].high = get(0, i);
        for (int k = 1; k < n; ++k)
            for (int i = 0; i < 3; ++i) {
                float v = get(k, i);
                if (v < root_bbox.d[i].low) root_bbox.d[i].low = v;
                if (v > root_bbox.d[i].high) root_bbox.d[i].high = v;
            }
        nodes.reserve(n / 3 + 16);
        root = divide(0, n, root_bbox, 0);
    }

    struct Result {
        size_t index = (size_t)-1;
        float dist = FLT_MAX;  // KNNResultSet::init, nanoflann.hpp:158-164
        // counters for the per-query work table; not part of the semantics
        long internal = 0, leaves = 0, dists = 0;
    };

    // nanoflann.hpp:1346-1409 (+ KNNResultSet<float>(1)::addPoint 175-202; eps = 0)
    void search_level(Result& rs, const float* vec, int node, float mindistsq, float* dists) const {
        const KdNode& nd = nodes[node];
        if (nd.child1 < 0 && nd.child2 < 0) {
            rs.leaves++;
            float worst = rs.dist;
            for (int i = nd.left; i < nd.right; ++i) {
                int index = vind[i];
                // nanoflann.hpp:432-440: result = 0; result += diff*diff, left to right
                float d = 0.f;
                for (int k = 0; k < 3; ++k) {
                    float diff = vec[k] - get(index, k);
                    d += diff * diff;
                }
                rs.dists++;
                if (d < worst) {
                    if (rs.dist > d) {  // strict: first visited wins ties
                        rs.dist = d;
                        rs.index = (size_t)index;
                    }
                }
            }
            return;
        }
        rs.internal++;
        int idx = nd.divfeat;
        float val = vec[idx];
        float diff1 = val - nd.divlow;
        float diff2 = val - nd.divhigh;
        int best, other;
        float cut;
        if ((diff1 + diff2) < 0) {
            best = nd.child1;
            other = nd.child2;
            cut = (val - nd.divhigh) * (val - nd.divhigh);
        } else {
            best = nd.child2;
            other = nd.child1;
            cut = (val - nd.divlow) * (val - nd.divlow);
        }
        search_level(rs, vec, best, mindistsq, dists);
        float dst = dists[idx];
        mindistsq = mindistsq + cut - dst;
        dists[idx] = cut;
        if (mindistsq * 1.0f <= rs.dist) search_level(rs, vec, other, mindistsq, dists);
        dists[idx] = dst;
    }

    // nanoflann.hpp:1221-1241 + computeInitialDistances 1005-1022
    Result find(const float* vec) const {
        Result rs;
        if (n == 0) return rs;
        float dists[3] = {0.f, 0.f, 0.f};
        float distsq = 0.f;
        for (int i = 0; i < 3; ++i) {
            if (vec[i] < root_bbox.d[i].low) {
                dists[i] = (vec[i] - root_bbox.d[i].low) * (vec[i] - root_bbox.d[i].low);
                distsq += dists[i];
            }
            if (vec[i] > root_bbox.d[i].high) {
                dists[i] = (vec[i] - root_bbox.d[i].high) * (vec[i] - root_bbox.d[i].high);
                distsq += dists[i];
            }
        }
        search_level(rs, vec, root, distsq, dists);
        return rs;
    }
};

// util.cpp:7-29 maybe_parallel_for: one atomic grab per index, n_threads-1 spawned + caller
static void parallel_for(const std::function<void(int)>& body, int n, int n_threads) {
    std::atomic<int> counter(-1);
    auto worker = [&]() {
        while (true) {
            int i = ++counter;
            if (i >= n) break;
            body(i);
        }
    };
    if (n >= 50) {
        std::vector<std::thread> th;
        for (int i = 1; i < n_threads; ++i) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
    } else {
        worker();
    }
}

// ---------------------------------------------------------------------------------------
// SDF::Impl restatement
// ---------------------------------------------------------------------------------------
struct Oracle {
    const float* verts;  // caller memory (re-read on update, like Eigen::Ref, sdf.cpp:339)
    const uint32_t* faces;
    int nv, nf;
    bool robust;
    std::vector<V3> face_normal;
    std::vector<float> face_area;
    std::vector<std::vector<int>> adj;
    float aabb[6];
    float R[9];
    KdTree kd;
    // rotated-frame per-face data: same values sdf.cpp:297-309 recomputes per query
    std::vector<V3> ra, rb, rc, rn, bmin, bmax;
    // candidate filter: uniform 2-D grids over the rotated frame.  _raycast's result depends
    // only on the SET of faces passing Overlap && facing && bary2d (XOR is order independent,
    // sdf.cpp:311), so any conservative superset + the exact leaf predicate is equivalent to
    // the R-tree (validated against the real RTree.h in oracle/_ref).
    struct Grid {
        int g = 1;
        float o0 = 0, o1 = 0, inv0 = 0, inv1 = 0;
        std::vector<int> start, items;
        int cell(float v, float o, float inv) const {
            float f = std::floor((v - o) * inv);
            if (!(f > 0.f)) return 0;
            if (f >= (float)(g - 1)) return g - 1;
            return (int)f;
        }
    } gxy, gyz;
    bool brute = false;

    V3 vert(uint32_t i) const { return {verts[3 * (size_t)i], verts[3 * (size_t)i + 1], verts[3 * (size_t)i + 2]}; }

    void build_grid(Grid& G, int c0, int c1) {
        int g = (int)std::sqrt((double)nf / 2.0);
        g = std::max(1, std::min(g, 2048));
        G.g = g;
        float lo0 = FLT_MAX, lo1 = FLT_MAX, hi0 = -FLT_MAX, hi1 = -FLT_MAX;
        for (int f = 0; f < nf; ++f) {
            lo0 = std::min(lo0, bmin[f][c0]);
            lo1 = std::min(lo1, bmin[f][c1]);
            hi0 = std::max(hi0, bmax[f][c0]);
            hi1 = std::max(hi1, bmax[f][c1]);
        }
        if (nf == 0) lo0 = lo1 = hi0 = hi1 = 0;
        G.o0 = lo0;
        G.o1 = lo1;
        G.inv0 = (hi0 > lo0) ? (float)g / (hi0 - lo0) : 0.f;
        G.inv1 = (hi1 > lo1) ? (float)g / (hi1 - lo1) : 0.f;
        std::vector<int> cnt((size_t)g * g + 1, 0);
        auto range = [&](int f, int& a0, int& a1, int& b0, int& b1) {
            a0 = G.cell(bmin[f][c0], G.o0, G.inv0);
            a1 = G.cell(bmax[f][c0], G.o0, G.inv0);
            b0 = G.cell(bmin[f][c1], G.o1, G.inv1);
            b1 = G.cell(bmax[f][c1], G.o1, G.inv1);
        };
        for (int f = 0; f < nf; ++f) {
            int a0, a1, b0, b1;
            range(f, a0, a1, b0, b1);
            for (int a = a0; a <= a1; ++a)
                for (int b = b0; b <= b1; ++b) cnt[(size_t)a * g + b + 1]++;
        }
        for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
        G.start = cnt;
        G.items.assign(cnt.back(), 0);
        std::vector<int> fill(cnt.begin(), cnt.end() - 1);
        for (int f = 0; f < nf; ++f) {  // ascending face id per cell
            int a0, a1, b0, b1;
            range(f, a0, a1, b0, b1);
            for (int a = a0; a <= a1; ++a)
                for (int b = b0; b <= b1; ++b) G.items[fill[(size_t)a * g + b]++] = f;
        }
    }

    // sdf.cpp:17-29 (adjacency) — built once; faces never change (sdf.hpp:162-163)
    void init_adj() {
        adj.assign(nv, {});
        for (int i = 0; i < nf; ++i)
            for (int j = 0; j < 3; ++j) adj[faces[3 * (size_t)i + j]].push_back(i);
    }

    // sdf.cpp:31-73
    void update(const float* Rin) {
        kd.build(verts, nv);
        for (int t = 0; t < 3; ++t) aabb[t] = FLT_MAX, aabb[t + 3] = -FLT_MAX;
        for (int i = 0; i < nv; ++i)
            for (int t = 0; t < 3; ++t) {
                float v = verts[3 * (size_t)i + t];
                if (v < aabb[t]) aabb[t] = v;
                if (v > aabb[t + 3]) aabb[t + 3] = v;
            }
        if (Rin) std::memcpy(R, Rin, sizeof(R));
        face_normal.resize(nf);
        face_area.resize(nf);
        ra.resize(nf), rb.resize(nf), rc.resize(nf), rn.resize(nf), bmin.resize(nf), bmax.resize(nf);
        for (int i = 0; i < nf; ++i) {
            V3 va = vert(faces[3 * (size_t)i]), vb = vert(faces[3 * (size_t)i + 1]),
               vc = vert(faces[3 * (size_t)i + 2]);
            // sdf.cpp:67-69
            V3 n = cross(sub(vb, va), sub(vc, va));
            float area = std::sqrt(sqnorm(n));
            face_area[i] = area;
            face_normal[i] = {n.x / area, n.y / area, n.z / area};
            if (robust) {
                // sdf.cpp:54-64 (cwiseMin/cwiseMax = std::min/std::max semantics)
                V3 a = mulR(va, R), b = mulR(vb, R), c = mulR(vc, R);
                V3 mn = a, mx = a;
                mn = {stdmin(mn.x, b.x), stdmin(mn.y, b.y), stdmin(mn.z, b.z)};
                mn = {stdmin(mn.x, c.x), stdmin(mn.y, c.y), stdmin(mn.z, c.z)};
                mx = {stdmax(mx.x, b.x), stdmax(mx.y, b.y), stdmax(mx.z, b.z)};
                mx = {stdmax(mx.x, c.x), stdmax(mx.y, c.y), stdmax(mx.z, c.z)};
                ra[i] = a, rb[i] = b, rc[i] = c, bmin[i] = mn, bmax[i] = mx;
                rn[i] = mulR(face_normal[i], R);  // sdf.cpp:299-300
            }
        }
        if (robust) {
            build_grid(gxy, 0, 1);
            build_grid(gyz, 1, 2);
        }
    }

    // leaf predicate of one ray for one face: RTree.h:1427-1438 Overlap + sdf.cpp:297-313
    // ax: 2 (z) or 0 (x); inv: negative direction
    inline bool ray_hits_face(const V3& q, int f, int ax, bool inv) const {
        float qmin[3] = {q.x, q.y, q.z}, qmax[3] = {q.x, q.y, q.z};
        if (inv)
            qmin[ax] = -FLT_MAX;
        else
            qmax[ax] = FLT_MAX;
        const V3 &mn = bmin[f], &mx = bmax[f];
        for (int d = 0; d < 3; ++d)
            if (qmin[d] > mx[d] || mn[d] > qmax[d]) return false;
        const V3& n = rn[f];
        V3 d = sub(q, ra[f]);
        float s = dot(n, d) * n[ax];
        if ((s > 0.f) != inv) return false;
        if (ax == 2)
            return bary2d_inside(q.x, q.y, ra[f].x, ra[f].y, rb[f].x, rb[f].y, rc[f].x, rc[f].y);
        return bary2d_inside(q.y, q.z, ra[f].y, ra[f].z, rb[f].y, rb[f].z, rc[f].y, rc[f].z);
    }

    int ray(const V3& q, int ax, bool inv, long* ncand = nullptr) const {
        int contained = 0;
        if (brute) {
            for (int f = 0; f < nf; ++f)
                if (ray_hits_face(q, f, ax, inv)) contained ^= 1;
            return contained;
        }
        const Grid& G = (ax == 2) ? gxy : gyz;
        float k0 = (ax == 2) ? q.x : q.y, k1 = (ax == 2) ? q.y : q.z;
        int a = G.cell(k0, G.o0, G.inv0), b = G.cell(k1, G.o1, G.inv1);
        size_t c = (size_t)a * G.g + b;
        for (int i = G.start[c]; i < G.start[c + 1]; ++i) {
            if (ncand) ++*ncand;
            if (ray_hits_face(q, G.items[i], ax, inv)) contained ^= 1;
        }
        return contained;
    }

    // sdf.cpp:276-328.  detail (optional): {z+, z-, x+ or -1}
    float raycast(V3 p, int* detail = nullptr) const {
        if (detail) detail[0] = detail[1] = detail[2] = -1;
        const float pc[3] = {p.x, p.y, p.z};
        for (int t = 0; t < 3; ++t)
            if (pc[t] < aabb[t] || pc[t] > aabb[t + 3]) return -1.f;
        V3 q = mulR(p, R);
        int zp = ray(q, 2, false), zm = ray(q, 2, true);
        int result = zp + zm;
        if (detail) detail[0] = zp, detail[1] = zm;
        if (result == 1) {
            int xp = ray(q, 0, false);
            if (detail) detail[2] = xp;
            result += xp;
        }
        return result > 1 ? 1.0f : -1.0f;
    }

    // sdf.cpp:93-158 body of the per-point lambda
    float calc_one(V3 p, bool trunc_aabb, int* nn_out = nullptr) const {
        const float DIST_EPS = robust ? 0.f : 1e-5f;
        float sign = 0.f;
        if (robust) sign = raycast(p);
        float min_dist = FLT_MAX;  // result.setConstant(FLT_MAX), sdf.cpp:96-97
        if (trunc_aabb) {
            const float pc[3] = {p.x, p.y, p.z};
            for (int t = 0; t < 3; ++t)
                if (pc[t] < aabb[t] || pc[t] > aabb[t + 3]) {
                    min_dist = -FLT_MAX;  // the `continue` only continues this loop: falls through
                    continue;
                }
        }
        const float pv[3] = {p.x, p.y, p.z};
        KdTree::Result rs = kd.find(pv);
        size_t nn = rs.index;
        if (nn_out) *nn_out = (int)nn;
        V3 avg = {0, 0, 0};
        if (nn != (size_t)-1) {
            for (int f : adj[nn]) {
                V3 a = vert(faces[3 * (size_t)f]), b = vert(faces[3 * (size_t)f + 1]),
                   c = vert(faces[3 * (size_t)f + 2]);
                V3 n = face_normal[f];
                float d = dist_point2tri(p, a, b, c, n, face_area[f]);
                if (d < min_dist - DIST_EPS) {
                    min_dist = d;
                    if (!robust) avg = n;
                } else if (!robust && d < min_dist + DIST_EPS) {
                    avg = {avg.x + n.x, avg.y + n.y, avg.z + n.z};
                }
            }
        }
        min_dist = std::sqrt(min_dist);
        if (robust) {
            min_dist *= sign;
        } else if (nn != (size_t)-1 && dot(avg, sub(p, vert((uint32_t)nn))) > 0) {
            min_dist = -min_dist;
        }
        return min_dist;
    }
};

}  // namespace orc

using orc::Oracle;
using orc::V3;

extern "C" {

// Unit-quaternion (w,x,y,z) -> row-major 3x3, Eigen QuaternionBase::toRotationMatrix order
// (sdf.cpp:45-48).  The quaternion is normalised first (Eigen normalize(): q / sqrt(|q|^2),
// 4-term squaredNorm; association not result-relevant because R is always injected).
void orc_quat_to_R(const float* q4, float* R) {
    float w = q4[0], x = q4[1], y = q4[2], z = q4[3];
    float nrm = std::sqrt((w * w + x * x) + (y * y + z * z));
    w /= nrm, x /= nrm, y /= nrm, z /= nrm;
    float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    float twx = tx * w, twy = ty * w, twz = tz * w;
    float txx = tx * x, txy = ty * x, txz = tz * x;
    float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
    R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
    R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}

// verts/faces are NOT copied (Eigen::Ref semantics); R must be given when robust.
void* orc_create(const float* verts, int nv, const uint32_t* faces, int nf, int robust,
                 const float* R) {
    Oracle* o = new Oracle();
    o->verts = verts, o->faces = faces, o->nv = nv, o->nf = nf, o->robust = robust != 0;
    static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(o->R, I, sizeof(I));
    o->init_adj();
    o->update(R);
    return o;
}
void orc_destroy(void* h) { delete (Oracle*)h; }
void orc_update(void* h, const float* R) { ((Oracle*)h)->update(R); }
void orc_set_brute(void* h, int b) { ((Oracle*)h)->brute = b != 0; }

void orc_nn(void* h, const float* pts, int n, int32_t* out, int n_threads) {
    Oracle* o = (Oracle*)h;
    orc::parallel_for([&](int i) { out[i] = (int32_t)o->kd.find(pts + 3 * (size_t)i).index; }, n,
                      n_threads);
}
void orc_calc(void* h, const float* pts, int n, int trunc_aabb, float* out, int n_threads) {
    Oracle* o = (Oracle*)h;
    orc::parallel_for(
        [&](int i) {
            const float* p = pts + 3 * (size_t)i;
            out[i] = o->calc_one({p[0], p[1], p[2]}, trunc_aabb != 0);
        },
        n, n_threads);
}
// sdf.cpp:160-174
void orc_contains(void* h, const float* pts, int n, uint8_t* out, int n_threads) {
    Oracle* o = (Oracle*)h;
    if (o->robust) {
        orc::parallel_for(
            [&](int i) {
                const float* p = pts + 3 * (size_t)i;
                out[i] = o->raycast({p[0], p[1], p[2]}) >= 0.0f;
            },
            n, n_threads);
    } else {
        orc::parallel_for(
            [&](int i) {
                const float* p = pts + 3 * (size_t)i;
                out[i] = o->calc_one({p[0], p[1], p[2]}, true) >= 0;
            },
            n, n_threads);
    }
}
// per-ray parities {z+, z-, x+ (or -1 if not cast)}; -1,-1,-1 if outside the mesh AABB
void orc_raycast_detail(void* h, const float* pts, int n, int32_t* out3, int n_threads) {
    Oracle* o = (Oracle*)h;
    orc::parallel_for(
        [&](int i) {
            const float* p = pts + 3 * (size_t)i;
            int d[3];
            o->raycast({p[0], p[1], p[2]}, d);
            out3[3 * (size_t)i] = d[0], out3[3 * (size_t)i + 1] = d[1], out3[3 * (size_t)i + 2] = d[2];
        },
        n, n_threads);
}
// exact leaf predicate for an explicit candidate list (used with the real R-tree's output)
int orc_ray_parity_candidates(void* h, const float* q_rot, int ax, int inv, const int32_t* cand,
                              int ncand) {
    Oracle* o = (Oracle*)h;
    V3 q = {q_rot[0], q_rot[1], q_rot[2]};
    int c = 0;
    for (int i = 0; i < ncand; ++i)
        if (o->ray_hits_face(q, cand[i], ax, inv != 0)) c ^= 1;
    return c;
}
void orc_rotate_point(void* h, const float* p, float* q) {
    Oracle* o = (Oracle*)h;
    V3 r = orc::mulR({p[0], p[1], p[2]}, o->R);
    q[0] = r.x, q[1] = r.y, q[2] = r.z;
}

void orc_get_face_normals(void* h, float* out) {
    Oracle* o = (Oracle*)h;
    std::memcpy(out, o->face_normal.data(), sizeof(float) * 3 * (size_t)o->nf);
}
void orc_get_face_areas(void* h, float* out) {
    Oracle* o = (Oracle*)h;
    std::memcpy(out, o->face_area.data(), sizeof(float) * (size_t)o->nf);
}
void orc_get_aabb(void* h, float* out6) { std::memcpy(out6, ((Oracle*)h)->aabb, 6 * sizeof(float)); }
void orc_get_face_boxes(void* h, float* mn, float* mx) {
    Oracle* o = (Oracle*)h;
    std::memcpy(mn, o->bmin.data(), sizeof(float) * 3 * (size_t)o->nf);
    std::memcpy(mx, o->bmax.data(), sizeof(float) * 3 * (size_t)o->nf);
}
int orc_adj_count(void* h, int v) { return (int)((Oracle*)h)->adj[v].size(); }
void orc_adj_get(void* h, int v, int32_t* out) {
    auto& a = ((Oracle*)h)->adj[v];
    std::memcpy(out, a.data(), a.size() * sizeof(int));
}

// kd-tree export in DFS pre-order (node id == creation order), for comparison with the real
// nanoflann tree and with the product's flattened tree.
int orc_kd_num_nodes(void* h) { return (int)((Oracle*)h)->kd.nodes.size(); }
int orc_kd_max_depth(void* h) { return ((Oracle*)h)->kd.max_depth; }
void orc_kd_export(void* h, int32_t* vind, int32_t* node_i /*[n,5]: c1,c2,left,right,feat*/,
                   float* node_f /*[n,2]: divlow,divhigh*/) {
    Oracle* o = (Oracle*)h;
    std::memcpy(vind, o->kd.vind.data(), sizeof(int) * (size_t)o->nv);
    for (size_t i = 0; i < o->kd.nodes.size(); ++i) {
        const orc::KdNode& nd = o->kd.nodes[i];
        node_i[5 * i] = nd.child1, node_i[5 * i + 1] = nd.child2, node_i[5 * i + 2] = nd.left;
        node_i[5 * i + 3] = nd.right, node_i[5 * i + 4] = nd.divfeat;
        node_f[2 * i] = nd.divlow, node_f[2 * i + 1] = nd.divhigh;
    }
}
// per-query work counters averaged over the batch: internal nodes, leaves, vertex distances
void orc_kd_stats(void* h, const float* pts, int n, double* out3) {
    Oracle* o = (Oracle*)h;
    long a = 0, b = 0, c = 0;
    for (int i = 0; i < n; ++i) {
        auto r = o->kd.find(pts + 3 * (size_t)i);
        a += r.internal, b += r.leaves, c += r.dists;
    }
    out3[0] = (double)a / n, out3[1] = (double)b / n, out3[2] = (double)c / n;
}

// Impl::sample_surface (sdf.cpp:176-234) with the random draws replaced by a counter-based
// stream: sample j takes its three uniforms from Philox4x32-10(key = seed, counter = j), mapped
// to [0, 1 - eps] like uniform_real_distribution<float>(0, 1 - eps) (sdf.cpp:182-183).  The
// reference draws from a time-seeded thread_local mt19937 (util.cpp:32-40), so only the
// distribution can match it; against this restatement the GPU kernel is bit-exact.
// face_area_cum is the reference's sequential fp32 running sum (sdf.cpp:196-201); total_area is
// passed in (the reference uses Eigen's vectorised sum, whose association is not restated).
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                 uint32_t k1, uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
static inline float u01(uint32_t x) {  // [0, 1 - eps]
    return ((float)(x >> 8) * 5.9604644775390625e-8f) * (1.0f - FLT_EPSILON);
}
void orc_sample_surface(void* h, int64_t n, uint64_t seed, float total_area, float* out) {
    Oracle* o = (Oracle*)h;
    const int nf = o->nf;
    if (nf == 0) return;
    std::vector<float> cum(nf);
    cum[0] = o->face_area[0];
    for (int i = 1; i < nf; ++i) cum[i] = cum[i - 1] + o->face_area[i];
    for (int64_t j = 0; j < n; ++j) {
        uint32_t x[4];
        philox4x32_10((uint32_t)j, (uint32_t)((uint64_t)j >> 32), 0u, 0u, (uint32_t)seed,
                      (uint32_t)(seed >> 32), x);
        const float r = u01(x[0]) * total_area;
        int i = (int)(std::lower_bound(cum.begin(), cum.end(), r) - cum.begin());
        i = std::max(std::min(i, nf - 1), 0);
        const V3 a = o->vert(o->faces[3 * (size_t)i]), b = o->vert(o->faces[3 * (size_t)i + 1]),
                 c = o->vert(o->faces[3 * (size_t)i + 2]);
        const V3 ab = orc::sub(b, a), ac = orc::sub(c, a), bc = orc::sub(c, b);
        V3 perp = orc::cross(bc, o->face_normal[i]);
        const float pn = std::sqrt(orc::sqnorm(perp));  // normalized(): v / norm() (no-op if 0)
        if (pn > 0.f) perp = {perp.x / pn, perp.y / pn, perp.z / pn};
        const bool a_dir = orc::dot(ab, perp) < 0.0f;
        const float u = u01(x[1]), w = u01(x[2]);
        V3 p = {(a.x + u * ab.x) + w * ac.x, (a.y + u * ab.y) + w * ac.y, (a.z + u * ab.z) + w * ac.z};
        const float bp = orc::dot(orc::sub(p, b), perp);
        if ((bp >= 0.0f) != a_dir)
            p = {p.x - (bp * perp.x) * 2.f, p.y - (bp * perp.y) * 2.f, p.z - (bp * perp.z) * 2.f};
        out[3 * j] = p.x, out[3 * j + 1] = p.y, out[3 * j + 2] = p.z;
    }
}

// standalone geometry entry points (for unit KATs)
float orc_dist_point2tri(const float* p, const float* a, const float* b, const float* c) {
    V3 A = {a[0], a[1], a[2]}, B = {b[0], b[1], b[2]}, C = {c[0], c[1], c[2]};
    V3 n = orc::cross(orc::sub(B, A), orc::sub(C, A));
    float area = std::sqrt(orc::sqnorm(n));
    n = {n.x / area, n.y / area, n.z / area};
    return orc::dist_point2tri({p[0], p[1], p[2]}, A, B, C, n, area);
}
float orc_dist_point2lineseg(const float* p, const float* a, const float* b) {
    return orc::dist_point2lineseg({p[0], p[1], p[2]}, {a[0], a[1], a[2]}, {b[0], b[1], b[2]});
}
int orc_bary2d_inside(const float* p, const float* a, const float* b, const float* c) {
    return orc::bary2d_inside(p[0], p[1], a[0], a[1], b[0], b[1], c[0], c[1]);
}

}  // extern "C"

// =========================================================================================
// sdf::Renderer (src/renderer.cpp, sdf.hpp:282-369): image-space raycaster.
// TEST INFRASTRUCTURE ONLY, like the rest of this file.
//
// Parity status:
//   * candidate faces of a 2-D point (RTree<int,float,2>::Search of a degenerate box,
//     renderer.cpp:170-173) and the 2-D nearest vertex of fill_outside (nanoflann DIM = 2,
//     renderer.cpp:196-204, 248-250): PINNED against the real vendored headers through
//     oracle/ref_shim.cpp (ref_rtree2_*, ref_kd2_*); tests/test_oracle_renderer.py.
//   * depth and mask do not depend on the visiting order (min / any).  The vertex id does when
//     two DIFFERENT vertices with exactly equal z compete (renderer.cpp:84 keeps the first
//     visited): orc_rend_eval_candidates applies the handlers in a GIVEN order (used with the
//     real R-tree's order); the restatement's own rule is order independent: lowest z, then
//     lowest face id.
//   * bary2d arithmetic (sdf.hpp:130-142) is plain float C++, no Eigen reduction beyond
//     tail<2>().sum() = b1 + b2; the projection (renderer.cpp:37-40) is elementwise.
// =========================================================================================
namespace orc {

struct Bary {
    float b0, b1, b2;
};
// sdf.hpp:130-142
static inline Bary bary2d(float px, float py, float ax, float ay, float bx, float by, float cx,
                          float cy) {
    float v0x = bx - ax, v0y = by - ay, v1x = cx - ax, v1y = cy - ay, v2x = px - ax, v2y = py - ay;
    float invden = 1.f / (v0x * v1y - v1x * v0y);
    Bary r;
    r.b1 = (v2x * v1y - v1x * v2y) * invden;
    r.b2 = (v0x * v2y - v2x * v0y) * invden;
    r.b0 = 1.0f - (r.b1 + r.b2);
    return r;
}

struct RendOracle {
    const float* verts;  // caller memory, re-read on update (Eigen::Ref, renderer.cpp:143)
    const uint32_t* faces;
    int nv, nf, width, height;
    float fx, fy, cx, cy;
    std::vector<float> cam;         // verts_cam [nv,2] (renderer.cpp:37-40)
    std::vector<float> bmin, bmax;  // per-face 2-D AABB (renderer.cpp:46-51)
    std::vector<uint8_t> valid;     // 0: NaN box, not inserted (renderer.cpp:52-56)
    std::vector<float> cam3;        // verts_cam embedded as (x, y, 0) for the kd restatement
    KdTree kd;
    bool kd_ready = false;
    // candidate filter (any superset + the exact per-face test gives the R-tree's hit set)
    int g = 1;
    float o0 = 0, o1 = 0, inv0 = 0, inv1 = 0;
    std::vector<int> start, items;

    int cell(float v, float o, float inv) const {
        float f = std::floor((v - o) * inv);
        if (!(f > 0.f)) return 0;
        if (f >= (float)(g - 1)) return g - 1;
        return (int)f;
    }

    void update() {
        kd_ready = false;
        cam.resize(2 * (size_t)nv);
        for (int i = 0; i < nv; ++i) {
            const float z = verts[3 * (size_t)i + 2];
            cam[2 * (size_t)i] = (verts[3 * (size_t)i] * fx) / z + cx;
            cam[2 * (size_t)i + 1] = (verts[3 * (size_t)i + 1] * fy) / z + cy;
        }
        bmin.assign(2 * (size_t)nf, 0.f);
        bmax.assign(2 * (size_t)nf, 0.f);
        valid.assign(nf, 0);
        float lo0 = FLT_MAX, lo1 = FLT_MAX, hi0 = -FLT_MAX, hi1 = -FLT_MAX;
        for (int f = 0; f < nf; ++f) {
            const float* a = &cam[2 * (size_t)faces[3 * (size_t)f]];
            const float* b = &cam[2 * (size_t)faces[3 * (size_t)f + 1]];
            const float* c = &cam[2 * (size_t)faces[3 * (size_t)f + 2]];
            for (int d = 0; d < 2; ++d) {
                float mn = a[d], mx = a[d];
                mn = stdmin(mn, b[d]), mn = stdmin(mn, c[d]);
                mx = stdmax(mx, b[d]), mx = stdmax(mx, c[d]);
                bmin[2 * (size_t)f + d] = mn, bmax[2 * (size_t)f + d] = mx;
            }
            valid[f] = !(std::isnan(bmin[2 * (size_t)f]) || std::isnan(bmin[2 * (size_t)f + 1]) ||
                         std::isnan(bmax[2 * (size_t)f]) || std::isnan(bmax[2 * (size_t)f + 1]));
            if (valid[f]) {
                for (int d = 0; d < 2; ++d) {
                    const float mn = bmin[2 * (size_t)f + d], mx = bmax[2 * (size_t)f + d];
                    float& lo = d ? lo1 : lo0;
                    float& hi = d ? hi1 : hi0;
                    if (std::fabs(mn) <= FLT_MAX && mn < lo) lo = mn;
                    if (std::fabs(mx) <= FLT_MAX && mx > hi) hi = mx;
                }
            }
        }
        if (!(lo0 <= hi0)) lo0 = hi0 = 0;
        if (!(lo1 <= hi1)) lo1 = hi1 = 0;
        g = std::max(1, std::min((int)std::sqrt((double)nf / 2.0), 2048));
        o0 = lo0, o1 = lo1;
        inv0 = (hi0 > lo0) ? (float)g / (hi0 - lo0) : 0.f;
        inv1 = (hi1 > lo1) ? (float)g / (hi1 - lo1) : 0.f;
        std::vector<int> cnt((size_t)g * g + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {
            for (int f = 0; f < nf; ++f) {
                if (!valid[f]) continue;
                const int a0 = cell(bmin[2 * (size_t)f], o0, inv0), a1 = cell(bmax[2 * (size_t)f], o0, inv0);
                const int b0 = cell(bmin[2 * (size_t)f + 1], o1, inv1), b1 = cell(bmax[2 * (size_t)f + 1], o1, inv1);
                for (int a = a0; a <= a1; ++a)
                    for (int b = b0; b <= b1; ++b) {
                        if (pass == 0)
                            cnt[(size_t)a * g + b + 1]++;
                        else
                            items[cnt[(size_t)a * g + b]++] = f;
                    }
            }
            if (pass == 0) {
                for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
                start = cnt;
                items.assign((size_t)start.back(), 0);
            }
        }
    }

    // one face against one point: RTree Overlap of the degenerate box (RTree.h:1427-1438, closed
    // intervals) && bary2d >= 0 (renderer.cpp:160-169).  Returns false if the face is not hit.
    bool hit(int f, float px, float py, Bary& b) const {
        if (!valid[f]) return false;
        if (px > bmax[2 * (size_t)f] || bmin[2 * (size_t)f] > px || py > bmax[2 * (size_t)f + 1] ||
            bmin[2 * (size_t)f + 1] > py)
            return false;
        const uint32_t* fc = faces + 3 * (size_t)f;
        b = bary2d(px, py, cam[2 * (size_t)fc[0]], cam[2 * (size_t)fc[0] + 1], cam[2 * (size_t)fc[1]],
                   cam[2 * (size_t)fc[1] + 1], cam[2 * (size_t)fc[2]], cam[2 * (size_t)fc[2] + 1]);
        return b.b0 >= 0.f && b.b1 >= 0.f && b.b2 >= 0.f;
    }
    // renderer.cpp:63-70
    float face_depth(int f, const Bary& b) const {
        const uint32_t* fc = faces + 3 * (size_t)f;
        return verts[3 * (size_t)fc[0] + 2] * b.b0 + verts[3 * (size_t)fc[1] + 2] * b.b1 +
               verts[3 * (size_t)fc[2] + 2] * b.b2;
    }
    // renderer.cpp:78-83
    int face_close_vert(int f, const Bary& b) const {
        const float bb[3] = {b.b0, b.b1, b.b2};
        int cv = 0;
        for (int i = 1; i < 3; ++i)
            if (bb[i] > bb[cv]) cv = i;
        return (int)faces[3 * (size_t)f + cv];
    }

    struct Pix {
        float depth = FLT_MAX;
        bool mask = false;
        int vertex = -1;
        int vface = -1;  // face that supplied `vertex` (order-independent rule)
    };
    // handlers applied to one hit, reference order semantics (renderer.cpp:63-87)
    void apply_ordered(Pix& p, int f, const Bary& b) const {
        p.depth = stdmin(p.depth, face_depth(f, b));
        p.mask = true;
        const int cand = face_close_vert(f, b);
        if (p.vertex == -1 || verts[3 * (size_t)cand + 2] < verts[3 * (size_t)p.vertex + 2]) p.vertex = cand;
    }
    // same, with the order-independent vertex rule: lowest z, then lowest face id
    void apply_canonical(Pix& p, int f, const Bary& b) const {
        p.depth = stdmin(p.depth, face_depth(f, b));
        p.mask = true;
        const int cand = face_close_vert(f, b);
        const float zc = verts[3 * (size_t)cand + 2];
        if (p.vertex == -1) {
            p.vertex = cand, p.vface = f;
        } else {
            const float zv = verts[3 * (size_t)p.vertex + 2];
            if (zc < zv || (zc == zv && f < p.vface)) p.vertex = cand, p.vface = f;
        }
    }
    Pix query(float px, float py) const {
        Pix p;
        const int c = cell(px, o0, inv0) * g + cell(py, o1, inv1);
        Bary b;
        for (int k = start[c]; k < start[c + 1]; ++k)
            if (hit(items[k], px, py, b)) apply_canonical(p, items[k], b);
        return p;
    }
    int nn2d(float px, float py) {
        const float q[3] = {px, py, 0.f};
        return (int)kd.find(q).index;
    }
    void ensure_kd() {  // renderer.cpp:110-113: lazy rebuild
        if (kd_ready) return;
        cam3.resize(3 * (size_t)nv);
        for (int i = 0; i < nv; ++i)
            cam3[3 * (size_t)i] = cam[2 * (size_t)i], cam3[3 * (size_t)i + 1] = cam[2 * (size_t)i + 1],
                            cam3[3 * (size_t)i + 2] = 0.f;
        kd.build(cam3.data(), nv);
        kd_ready = true;
    }
};

}  // namespace orc

extern "C" {

void* orc_rend_create(const float* verts, int nv, const uint32_t* faces, int nf, int width, int height,
                      float fx, float fy, float cx, float cy) {
    orc::RendOracle* r = new orc::RendOracle();
    r->verts = verts, r->faces = faces, r->nv = nv, r->nf = nf;
    r->width = width, r->height = height, r->fx = fx, r->fy = fy, r->cx = cx, r->cy = cy;
    r->update();
    return r;
}
void orc_rend_destroy(void* h) { delete (orc::RendOracle*)h; }
void orc_rend_update(void* h) { ((orc::RendOracle*)h)->update(); }
void orc_rend_get_cam(void* h, float* out) {
    orc::RendOracle* r = (orc::RendOracle*)h;
    std::memcpy(out, r->cam.data(), r->cam.size() * sizeof(float));
}
void orc_rend_get_face_boxes(void* h, float* mn, float* mx, uint8_t* valid) {
    orc::RendOracle* r = (orc::RendOracle*)h;
    std::memcpy(mn, r->bmin.data(), r->bmin.size() * sizeof(float));
    std::memcpy(mx, r->bmax.data(), r->bmax.size() * sizeof(float));
    std::memcpy(valid, r->valid.data(), r->valid.size());
}
// pts == nullptr: the image, pixel i = (float)(i % width), (float)(i / width) (renderer.cpp:187-189)
// kind 0 depth (float out, FLT_MAX -> 0), 1 mask (uint8), 2 vertex (int32; fill_outside: 2-D nn)
void orc_rend_query(void* h, int kind, const float* pts, int n, int fill_outside, void* out, int n_threads) {
    orc::RendOracle* r = (orc::RendOracle*)h;
    if (kind == 2 && fill_outside) r->ensure_kd();
    orc::parallel_for(
        [&](int i) {
            const float px = pts ? pts[2 * (size_t)i] : (float)(i % r->width);
            const float py = pts ? pts[2 * (size_t)i + 1] : (float)(i / r->width);
            const orc::RendOracle::Pix p = r->query(px, py);
            if (kind == 0) {
                ((float*)out)[i] = p.depth == FLT_MAX ? 0.0f : p.depth;  // renderer.cpp:193-194
            } else if (kind == 1) {
                ((uint8_t*)out)[i] = p.mask;
            } else {
                int v = p.vertex;
                if (fill_outside && v < 0) v = r->nn2d(px, py);  // renderer.cpp:196-204
                ((int32_t*)out)[i] = v;
            }
        },
        n, n_threads);
}
// the handlers applied to a GIVEN candidate list in its order (the real R-tree's search order):
// out3 = {depth bits, mask, vertex}.  The mask handler's early exit (renderer.cpp:72-76) does not
// change any of the three.
void orc_rend_eval_candidates(void* h, float px, float py, const int32_t* cand, int ncand, float* depth,
                              uint8_t* mask, int32_t* vertex) {
    orc::RendOracle* r = (orc::RendOracle*)h;
    orc::RendOracle::Pix p;
    orc::Bary b;
    for (int k = 0; k < ncand; ++k)
        if (r->hit(cand[k], px, py, b)) r->apply_ordered(p, cand[k], b);
    *depth = p.depth == FLT_MAX ? 0.0f : p.depth;
    *mask = p.mask;
    *vertex = p.vertex;
}

}  // extern "C"
