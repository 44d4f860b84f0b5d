// kd_host.h — host-side builder of the vertex kd-tree in "strict" tie-parity mode.
//
// nn() must return the vertex the reference returns, including on exact fp32 distance ties
// (up to 5.5 % of grid queries on the sample meshes).  The winner of a tie is the
// first vertex visited by nanoflann's near-child-first DFS, which depends on the exact tree:
// split dimension/value choice (nanoflann.hpp:909-956), the Hoare-style permutation of
// planeSplit (967-1003) and the tight-box divlow/divhigh (897-898).  This builder therefore
// reproduces that construction (leaf_max_size 10, sdf_util.hpp:38) and flattens it for the
// GPU:
//
//   * points are stored in permuted ("vind") order, so a leaf is a contiguous run and
//     the traversal needs no index indirection; `vind` maps the winner back once.
//     Tree positions are PADDED: leaf j owns positions [10 j, 10 j + 10), of which the first
//     `count` hold its vertices in nanoflann's order and the rest a far-away sentinel
//     (vind = -1), so that a whole leaf can be evaluated as ten straight-line slots.
//   * only internal nodes are materialised, 64 B each (one s_load_dwordx16 / four dwordx4):
//         { box1.lo[3], box1.hi[3], box2.lo[3], box2.hi[3],      tight boxes of the children
//           divlow, divhigh, ref1 | feat<<29, ref2 }
//     a child reference is either an internal node index or, with bit 31 set, a leaf
//     packed as  left<<4 | count  (count <= 10, left < 2^25).  `split[i]` = first tree-order
//     position of child 2's vertices (child 1 holds the positions below it).
//   * nodes are emitted in DFS pre-order, near the order queries walk them.
#pragma once
#include <cstdint>
#include <vector>

namespace sdfk {

// The tight child boxes are what nanoflann's divideTree returns through its bbox argument
// (nanoflann.hpp:889-903).  They are not part of nanoflann's own search; they serve the
// exact no-op skipping and the packet traversal.
struct KdNodeH {  // 64 B
    float lo1[3], hi1[3], lo2[3], hi2[3];
    float divlow, divhigh;
    uint32_t ref1, ref2;
};
static_assert(sizeof(KdNodeH) == 64, "device layout");

constexpr uint32_t KD_LEAF_BIT = 0x80000000u;
constexpr uint32_t KD_FEAT_SHIFT = 29;
constexpr uint32_t KD_REF_MASK = 0x1FFFFFFFu;  // strips leaf bit and feat bits
constexpr uint32_t KD_MAX_POINTS = 1u << 23;  // padded positions (<= 10 per vertex) fit 25 bits

struct KdHost {
    std::vector<int32_t> vind;     // nanoflann's permutation (dense, size n)
    std::vector<int32_t> pos_vert; // padded tree position -> vertex index, -1 = sentinel slot
    uint32_t n_leaves = 0;
    std::vector<KdNodeH> nodes;
    std::vector<uint32_t> split;  // parallel to nodes
    uint32_t root_ref = KD_LEAF_BIT;  // empty leaf
    int depth = 0;                    // max internal nodes on a root-to-leaf path
    float bbox[6] = {0, 0, 0, 0, 0, 0};  // min xyz, max xyz (= mesh AABB, sdf.cpp:36-37)

    void build(const float* pts, uint32_t n);

   private:
    struct Box {
        float lo[3], hi[3];
    };
    const float* p_ = nullptr;
    float at(int32_t i, int d) const { return p_[3 * (size_t)i + d]; }
    uint32_t divide(int32_t left, int32_t right, Box& box, int level);
};

inline uint32_t KdHost::divide(int32_t left, int32_t right, Box& box, int level) {
    const int32_t count = right - left;
    int32_t* ind = vind.data() + left;
    if (count <= 10) {
        // leaf: tight box of its points (nanoflann.hpp:862-879)
        for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = at(ind[0], d);
        for (int32_t k = 1; k < count; ++k)
            for (int d = 0; d < 3; ++d) {
                float v = at(ind[k], d);
                if (box.lo[d] > v) box.lo[d] = v;
                if (box.hi[d] < v) box.hi[d] = v;
            }
        const uint32_t base = 10u * n_leaves++;
        pos_vert.resize((size_t)base + 10, -1);
        for (int32_t k = 0; k < count; ++k) pos_vert[base + k] = ind[k];
        return KD_LEAF_BIT | (base << 4) | (uint32_t)count;
    }
    if (level + 1 > depth) depth = level + 1;

    // --- choose the cut (nanoflann.hpp:909-945) ---
    float span[3], max_span = box.hi[0] - box.lo[0];
    for (int d = 0; d < 3; ++d) {
        span[d] = box.hi[d] - box.lo[d];
        if (span[d] > max_span) max_span = span[d];
    }
    const float EPS = 0.00001f;
    float best_spread = -1;
    int feat = 0;
    float fmn = 0, fmx = 0;
    for (int d = 0; d < 3; ++d) {
        if (span[d] > (1 - EPS) * max_span) {
            float mn = at(ind[0], d), mx = mn;
            for (int32_t k = 1; k < count; ++k) {
                float v = at(ind[k], d);
                if (v < mn) mn = v;
                if (v > mx) mx = v;
            }
            float spread = mx - mn;
            if (spread > best_spread) feat = d, best_spread = spread, fmn = mn, fmx = mx;
        }
    }
    float cut = (box.lo[feat] + box.hi[feat]) / 2;
    if (cut < fmn)
        cut = fmn;
    else if (cut > fmx)
        cut = fmx;

    // --- two Hoare passes (nanoflann.hpp:967-1003), signed indices, "!hi" guards kept ---
    int32_t lo = 0, hi = count - 1;
    for (;;) {
        while (lo <= hi && at(ind[lo], feat) < cut) ++lo;
        while (hi && lo <= hi && at(ind[hi], feat) >= cut) --hi;
        if (lo > hi || !hi) break;
        int32_t t = ind[lo];
        ind[lo] = ind[hi];
        ind[hi] = t;
        ++lo, --hi;
    }
    const int32_t lim1 = lo;
    hi = count - 1;
    for (;;) {
        while (lo <= hi && at(ind[lo], feat) <= cut) ++lo;
        while (hi && lo <= hi && at(ind[hi], feat) > cut) --hi;
        if (lo > hi || !hi) break;
        int32_t t = ind[lo];
        ind[lo] = ind[hi];
        ind[hi] = t;
        ++lo, --hi;
    }
    const int32_t lim2 = lo;
    int32_t mid;
    if (lim1 > count / 2)
        mid = lim1;
    else if (lim2 < count / 2)
        mid = lim2;
    else
        mid = count / 2;

    // --- recurse; children return their tight boxes (nanoflann.hpp:889-903) ---
    const uint32_t self = (uint32_t)nodes.size();
    nodes.push_back({});
    split.push_back(0u);
    Box lb = box, rb = box;
    lb.hi[feat] = cut;
    rb.lo[feat] = cut;
    uint32_t r1 = divide(left, left + mid, lb, level + 1);
    split[self] = 10u * n_leaves;  // first padded position of child 2
    uint32_t r2 = divide(left + mid, right, rb, level + 1);
    KdNodeH& nd = nodes[self];
    nd.divlow = lb.hi[feat];
    nd.divhigh = rb.lo[feat];
    nd.ref1 = r1 | ((uint32_t)feat << KD_FEAT_SHIFT);
    nd.ref2 = r2;
    for (int d = 0; d < 3; ++d)
        nd.lo1[d] = lb.lo[d], nd.hi1[d] = lb.hi[d], nd.lo2[d] = rb.lo[d], nd.hi2[d] = rb.hi[d];
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = lb.lo[d] < rb.lo[d] ? lb.lo[d] : rb.lo[d];
        box.hi[d] = lb.hi[d] > rb.hi[d] ? lb.hi[d] : rb.hi[d];
    }
    return self;
}

inline void KdHost::build(const float* pts, uint32_t n) {
    p_ = pts;
    vind.resize(n);
    for (uint32_t i = 0; i < n; ++i) vind[i] = (int32_t)i;
    nodes.clear();
    nodes.reserve(n / 4 + 8);
    split.clear();
    split.reserve(n / 4 + 8);
    pos_vert.clear();
    n_leaves = 0;
    depth = 0;
    root_ref = KD_LEAF_BIT;
    for (int d = 0; d < 6; ++d) bbox[d] = 0;
    if (n == 0) return;
    Box box;
    for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = at(0, d);
    for (uint32_t k = 1; k < n; ++k)
        for (int d = 0; d < 3; ++d) {
            float v = at((int32_t)k, d);
            if (v < box.lo[d]) box.lo[d] = v;
            if (v > box.hi[d]) box.hi[d] = v;
        }
    for (int d = 0; d < 3; ++d) bbox[d] = box.lo[d], bbox[d + 3] = box.hi[d];
    root_ref = divide(0, (int32_t)n, box, 0);
}

}  // namespace sdfk
