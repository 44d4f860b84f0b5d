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
//   * only internal nodes are materialised, 16 B each (one dwordx4 load):
//         { divlow, divhigh, ref1 | feat<<29, ref2 }
//     a child reference is either an internal node index or, with bit 31 set, a leaf
//     packed as  left<<4 | count  (count <= 10, left < 2^25).
//   * nodes are emitted in DFS pre-order, near the order queries walk them.
#pragma once
#include <cstdint>
#include <vector>

namespace sdfk {

struct KdNodeH {
    float divlow, divhigh;
    uint32_t ref1, ref2;
};

constexpr uint32_t KD_LEAF_BIT = 0x80000000u;
constexpr uint32_t KD_FEAT_SHIFT = 29;
constexpr uint32_t KD_REF_MASK = 0x1FFFFFFFu;  // strips leaf bit and feat bits
constexpr uint32_t KD_MAX_POINTS = 1u << 25;

// tight AABBs of the two children of an internal node (what nanoflann's divideTree returns
// through its bbox argument, nanoflann.hpp:889-903): lo1 hi1 lo2 hi2.  Used only for exact
// no-op skipping / the packet traversal; not part of nanoflann's own search.
struct KdChildBoxesH {  // 64 B
    float lo1[3], hi1[3], lo2[3], hi2[3];
    uint32_t split;  // first tree-order position of child 2's vertices (child 1: [.., split))
    uint32_t pad[3];
};

static_assert(sizeof(KdChildBoxesH) == 64, "device layout");

struct KdHost {
    std::vector<int32_t> vind;
    std::vector<KdNodeH> nodes;
    std::vector<KdChildBoxesH> boxes;  // parallel to nodes
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
    uint32_t split(int32_t left, int32_t right, Box& box, int level);
};

inline uint32_t KdHost::split(int32_t left, int32_t right, Box& box, int level) {
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
        return KD_LEAF_BIT | ((uint32_t)left << 4) | (uint32_t)count;
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
    boxes.push_back({});
    Box lb = box, rb = box;
    lb.hi[feat] = cut;
    rb.lo[feat] = cut;
    uint32_t r1 = split(left, left + mid, lb, level + 1);
    uint32_t r2 = split(left + mid, right, rb, level + 1);
    KdNodeH& nd = nodes[self];
    nd.divlow = lb.hi[feat];
    nd.divhigh = rb.lo[feat];
    nd.ref1 = r1 | ((uint32_t)feat << KD_FEAT_SHIFT);
    nd.ref2 = r2;
    KdChildBoxesH& cb = boxes[self];
    cb.split = (uint32_t)(left + mid);
    cb.pad[0] = cb.pad[1] = cb.pad[2] = 0;
    for (int d = 0; d < 3; ++d)
        cb.lo1[d] = lb.lo[d], cb.hi1[d] = lb.hi[d], cb.lo2[d] = rb.lo[d], cb.hi2[d] = rb.hi[d];
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
    boxes.clear();
    boxes.reserve(n / 4 + 8);
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
    root_ref = split(0, (int32_t)n, box, 0);
}

}  // namespace sdfk
