// oracle/ref_shim.cpp
//
// TEST INFRASTRUCTURE ONLY.  Thin extern "C" driver around the reference's REAL vendored
// spatial indices, compiled from where they lie (-I/root/reference/include):
//     sdf/internal/nanoflann.hpp  (nanoflann v1.3.0)
//     sdf/internal/RTree.h        (nushoin/RTree)
// Output goes to oracle/_ref/ (git-ignored).  The rest of the reference (src/sdf.cpp,
// sdf.hpp, sdf_util.hpp, pybind.cpp) needs Eigen 3, which is absent from this image, so
// it is unbuildable here; only these two self-contained headers are compiled.  The adaptor
// below provides what sdf_util.hpp:76-99 provides (point count, coefficient access, no
// custom bbox) with the same template arguments as sdf.cpp:260-262 / sdf_util.hpp:26-34:
// metric_L2_Simple, DIM = 3, IndexType = int, leaf_max_size = 10.
//
// Used by tools/make_golden.py to generate tests/golden/* and by tests to pin the oracle.

#include <cfloat>
#include <cstdint>
#include <cstring>
#include <vector>

#include "sdf/internal/RTree.h"
#include "sdf/internal/nanoflann.hpp"

namespace {

struct PointCloud {
    const float* p;
    size_t n;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, int dim) const { return p[3 * idx + dim]; }
    template <class BBOX>
    bool kdtree_get_bbox(BBOX&) const {
        return false;
    }
};

using Metric = nanoflann::metric_L2_Simple::traits<float, PointCloud>::distance_t;
using KdIndex = nanoflann::KDTreeSingleIndexAdaptor<Metric, PointCloud, 3, int>;

struct RefKd {
    PointCloud cloud;
    KdIndex* index;
};

struct RefRTree {
    RTree<int, float, 3> tree;
};

// 2-D variants as sdf::Renderer configures them (renderer.cpp:245-250): RTree<int, float, 2> and
// KDTreeEigenRefAdaptor<const Points2D, 2, metric_L2_Simple> (leaf_max_size 10, sdf_util.hpp:38)
struct PointCloud2 {
    const float* p;
    size_t n;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, int dim) const { return p[2 * idx + dim]; }
    template <class BBOX>
    bool kdtree_get_bbox(BBOX&) const {
        return false;
    }
};
using Metric2 = nanoflann::metric_L2_Simple::traits<float, PointCloud2>::distance_t;
using KdIndex2 = nanoflann::KDTreeSingleIndexAdaptor<Metric2, PointCloud2, 2, int>;
struct RefKd2 {
    PointCloud2 cloud;
    KdIndex2* index;
};
struct RefRTree2 {
    RTree<int, float, 2> tree;
};

}  // namespace

extern "C" {

void* ref_kd2_create(const float* pts2, int n) {
    RefKd2* r = new RefKd2();
    r->cloud.p = pts2;
    r->cloud.n = (size_t)n;
    r->index = new KdIndex2(2, r->cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10));
    r->index->buildIndex();
    return r;
}
void ref_kd2_destroy(void* h) {
    RefKd2* r = (RefKd2*)h;
    delete r->index;
    delete r;
}
// same call as renderer.cpp:197-204 / 229-236
void ref_kd2_nn(void* h, const float* q2, int n, int32_t* out) {
    RefKd2* r = (RefKd2*)h;
    for (int i = 0; i < n; ++i) {
        size_t index = (size_t)-1;
        float dist;
        nanoflann::KNNResultSet<float> rs(1);
        rs.init(&index, &dist);
        r->index->findNeighbors(rs, q2 + 2 * (size_t)i, nanoflann::SearchParams(10));
        out[i] = (int32_t)index;
    }
}
// R-tree over the per-face 2-D boxes, inserted in face order, NaN boxes skipped (renderer.cpp:42-59)
void* ref_rtree2_create(const float* bmin2, const float* bmax2, const uint8_t* valid, int nf) {
    RefRTree2* r = new RefRTree2();
    for (int i = 0; i < nf; ++i)
        if (valid[i]) r->tree.Insert(bmin2 + 2 * (size_t)i, bmax2 + 2 * (size_t)i, i);
    return r;
}
void ref_rtree2_destroy(void* h) { delete (RefRTree2*)h; }
// candidates of the degenerate box [q, q] IN SEARCH ORDER (renderer.cpp:170-173)
int ref_rtree2_point(void* h, const float* q2, int32_t* out, int cap) {
    RefRTree2* r = (RefRTree2*)h;
    float mn[2] = {q2[0], q2[1]}, mx[2] = {q2[0], q2[1]};
    int cnt = 0;
    r->tree.Search(mn, mx, [&](const int& id) -> bool {
        if (cnt < cap) out[cnt] = id;
        ++cnt;
        return true;
    });
    return cnt;
}


void* ref_kd_create(const float* pts, int n) {
    RefKd* r = new RefKd();
    r->cloud.p = pts;
    r->cloud.n = (size_t)n;
    r->index = new KdIndex(3, r->cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10));
    r->index->buildIndex();
    return r;
}
void ref_kd_destroy(void* h) {
    RefKd* r = (RefKd*)h;
    delete r->index;
    delete r;
}
// same call as sdf.cpp:80-86
void ref_kd_nn(void* h, const float* q, int n, int32_t* out, float* out_dist) {
    RefKd* r = (RefKd*)h;
    for (int i = 0; i < n; ++i) {
        size_t index = (size_t)-1;
        float dist;
        nanoflann::KNNResultSet<float> rs(1);
        rs.init(&index, &dist);
        r->index->findNeighbors(rs, q + 3 * (size_t)i, nanoflann::SearchParams(10));
        out[i] = (int32_t)index;
        if (out_dist) out_dist[i] = dist;
    }
}
void ref_kd_vind(void* h, int32_t* out) {
    RefKd* r = (RefKd*)h;
    std::memcpy(out, r->index->vind.data(), sizeof(int) * r->cloud.n);
}
static void walk(const KdIndex::Node* nd, std::vector<int32_t>& ni, std::vector<float>& nf) {
    // pre-order; record [is_leaf, left/divfeat, right/0] and [divlow, divhigh]
    bool leaf = !nd->child1 && !nd->child2;
    ni.push_back(leaf ? 1 : 0);
    if (leaf) {
        ni.push_back((int32_t)nd->node_type.lr.left);
        ni.push_back((int32_t)nd->node_type.lr.right);
        nf.push_back(0.f), nf.push_back(0.f);
    } else {
        ni.push_back(nd->node_type.sub.divfeat);
        ni.push_back(0);
        nf.push_back(nd->node_type.sub.divlow), nf.push_back(nd->node_type.sub.divhigh);
        walk(nd->child1, ni, nf);
        walk(nd->child2, ni, nf);
    }
}
// returns node count; if node_i/node_f non-null fills pre-order records
int ref_kd_export(void* h, int32_t* node_i, float* node_f) {
    RefKd* r = (RefKd*)h;
    std::vector<int32_t> ni;
    std::vector<float> nf;
    if (r->index->root_node) walk(r->index->root_node, ni, nf);
    if (node_i) std::memcpy(node_i, ni.data(), ni.size() * sizeof(int32_t));
    if (node_f) std::memcpy(node_f, nf.data(), nf.size() * sizeof(float));
    return (int)(ni.size() / 3);
}

// R-tree over per-face boxes (sdf.cpp:64 Insert); Search as sdf.cpp:316-322
void* ref_rtree_create(const float* bmin, const float* bmax, int nf) {
    RefRTree* r = new RefRTree();
    for (int i = 0; i < nf; ++i) r->tree.Insert(bmin + 3 * (size_t)i, bmax + 3 * (size_t)i, i);
    return r;
}
void ref_rtree_destroy(void* h) { delete (RefRTree*)h; }
// candidates of one ray from rotated point q: ax in {0,2}; inv = negative direction.
// returns the candidate count; writes up to cap ids
int ref_rtree_ray(void* h, const float* q, int ax, int inv, int32_t* out, int cap) {
    RefRTree* r = (RefRTree*)h;
    float mn[3] = {q[0], q[1], q[2]}, mx[3] = {q[0], q[1], q[2]};
    if (inv)
        mn[ax] = -FLT_MAX;
    else
        mx[ax] = FLT_MAX;
    int cnt = 0;
    r->tree.Search(mn, mx, [&](const int& id) -> bool {
        if (cnt < cap) out[cnt] = id;
        ++cnt;
        return true;
    });
    return cnt;
}

}  // extern "C"
