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
//     the traversal needs no index indirection; `pos_vert` maps the winner back once.
//     Tree positions are PADDED per packet group: a maximal subtree with <= group_max
//     vertices ("group"; every leaf lies in exactly one) owns a block of positions that
//     starts at a multiple of 4 and spans ceil(count / 4) * 4 slots.  Its leaves sit in it
//     back to back in nanoflann's order; the tail slots hold a far-away sentinel
//     (pos_vert = -1).  The packet pass evaluates a group as straight-line chunks of four
//     slots (one 64-byte scalar load each); the strict search only ever sees the leaves.
//   * only internal nodes are materialised, 64 B each (one s_load_dwordx16 / four dwordx4):
//         { box1.lo[3], box1.hi[3], box2.lo[3], box2.hi[3],      tight boxes of the children
//           divlow, divhigh, ref1 | feat<<29, ref2 }
//     a child reference is either an internal node index or, with bit 31 set, a leaf
//     packed as  pos<<4 | count  (count <= 10, pos < 2^25).  `split[i]` = first tree
//     position of child 2's vertices (child 1 holds the positions below it).
//   * nodes are emitted in DFS pre-order, near the order queries walk them.
//
// The recursion is task-parallel: after a node's own split its two ranges are independent, so
// the top levels hand one child to another thread; sub-ranges are built into local arrays and
// spliced into the global pre-order numbering afterwards.  The result is identical to the
// serial build (it has to be: the numbering is part of the tie rule's data).
#pragma once
#include <cstdint>
#include <future>
#include <memory>
#include <thread>
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

// Two binary levels collapsed into one 4-wide node for the packet pass (half as many dependent
// load -> test -> branch steps per query wave).  Slots 0,1 come from binary child 1 (its two
// children, or itself in slot 0 if it is a leaf), slots 2,3 from child 2; an unused slot has
// an inverted infinite box, whose lower bound is +inf.  Search-order data (divlow/divhigh)
// is not needed here: the packet pass orders children by their box distances and settles
// ties on the binary tree.
struct KdQuadH {  // 128 B
    float box[4][6];  // lo xyz, hi xyz
    uint32_t ref[4];  // quad node index, or with bit 31 a packet group: (first pos / 4) << 4 | chunks
    uint32_t pad[4];
};
static_assert(sizeof(KdQuadH) == 128, "device layout");

constexpr uint32_t KD_LEAF_BIT = 0x80000000u;
constexpr uint32_t KD_FEAT_SHIFT = 29;
constexpr uint32_t KD_REF_MASK = 0x1FFFFFFFu;  // strips leaf bit and feat bits
constexpr uint32_t KD_MAX_POINTS = 1u << 23;   // padded positions (< 4 per vertex) fit 25 bits
constexpr uint32_t KD_GROUP_LIMIT = 60;        // chunks of a group fit 4 bits

struct KdHost {
    std::vector<int32_t> vind;      // nanoflann's permutation (dense, size n)
    std::vector<int32_t> pos_vert;  // padded tree position -> vertex index, -1 = sentinel slot
    uint32_t group_max = 24;        // vertices per packet group (10 = one group per leaf)
    std::vector<KdNodeH> nodes;
    std::vector<uint32_t> split;  // parallel to nodes
    uint32_t root_ref = KD_LEAF_BIT;  // empty leaf
    int depth = 0;                    // max internal nodes on a root-to-leaf path
    float bbox[6] = {0, 0, 0, 0, 0, 0};  // min xyz, max xyz (= mesh AABB, sdf.cpp:36-37)

    std::vector<KdQuadH> quads;  // 4-wide collapse of `nodes`, DFS pre-order
    uint32_t quad_root = KD_LEAF_BIT;
    int quad_depth = 0;

    // threads <= 1: serial.  Otherwise up to ~2*threads ranges are built concurrently.
    void build(const float* pts, uint32_t n, int threads = 0);
    void build_quads();

   private:
    struct Box {
        float lo[3], hi[3];
    };
    // a built sub-range with LOCAL numbering (node 0 / leaf 0 = its own first)
    struct Sub {
        std::vector<KdNodeH> nodes;
        std::vector<uint32_t> split;
        std::vector<int32_t> pos_vert;
        uint32_t npos = 0;  // position cursor; a multiple of 4 outside an open group
        uint32_t root = KD_LEAF_BIT;
        int depth = 0;
        // task-parallel inner node: nodes.size() == 1 (refs still to be set), children below
        std::unique_ptr<Sub> c1, c2;
        uint32_t total_nodes = 0, total_pos = 0;
    };
    struct Cut {
        int feat;
        int32_t mid;
    };
    const float* p_ = nullptr;
    float at(int32_t i, int d) const { return p_[3 * (size_t)i + d]; }
    Cut choose_and_partition(int32_t* ind, int32_t count, const Box& box, float& cutval) const;
    uint32_t divide(Sub& out, int32_t left, int32_t right, Box& box, int level, bool in_group);
    void divide_par(Sub& out, int32_t left, int32_t right, Box& box, int par_levels, int32_t cutoff);
    void splice(const Sub& s, uint32_t node_off, uint32_t pos_off);
    uint32_t quad_of(uint32_t ref, int level);
    std::vector<uint32_t> sub_first_, sub_count_;  // per binary node: first position, vertices
    void subtree_extent(uint32_t ref, uint32_t& first, uint32_t& count) const {
        if (ref & KD_LEAF_BIT)
            first = (ref & KD_REF_MASK) >> 4, count = ref & 15u;
        else
            first = sub_first_[ref & KD_REF_MASK], count = sub_count_[ref & KD_REF_MASK];
    }
    static uint32_t shift_ref(uint32_t ref, uint32_t node_off, uint32_t pos_off) {
        const uint32_t feat = ref & (3u << KD_FEAT_SHIFT);
        if (ref & KD_LEAF_BIT) {
            const uint32_t cnt = ref & 15u, pos = (ref & KD_REF_MASK) >> 4;
            return KD_LEAF_BIT | feat | ((pos + pos_off) << 4) | cnt;
        }
        return feat | ((ref & KD_REF_MASK) + node_off);
    }
};

// split choice (nanoflann.hpp:909-945) + the two Hoare passes (967-1003, signed indices, "!hi"
// guards kept) + the split index rule (950-955)
inline KdHost::Cut KdHost::choose_and_partition(int32_t* ind, int32_t count, const Box& box,
                                                float& cutval) const {
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
    cutval = cut;
    return {feat, mid};
}

// serial recursion into `out` (local numbering); returns the ref of the built range
inline uint32_t KdHost::divide(Sub& out, int32_t left, int32_t right, Box& box, int level,
                               bool in_group) {
    const int32_t count = right - left;
    int32_t* ind = vind.data() + left;
    if (!in_group && count <= (int32_t)group_max) {
        // this subtree is one packet group: its leaves take consecutive positions, the block is
        // closed (sentinel-filled to a multiple of 4) when the subtree is done
        const uint32_t r = divide(out, left, right, box, level, true);
        out.npos = (out.npos + 3u) & ~3u;
        out.pos_vert.resize(out.npos, -1);
        return r;
    }
    if (count <= 10) {
        // leaf: tight box of its points (nanoflann.hpp:862-879)
        for (int d = 0; d < 3; ++d) box.lo[d] = box.hi[d] = at(ind[0], d);
        for (int32_t k = 1; k < count; ++k)
            for (int d = 0; d < 3; ++d) {
                float v = at(ind[k], d);
                if (box.lo[d] > v) box.lo[d] = v;
                if (box.hi[d] < v) box.hi[d] = v;
            }
        const uint32_t base = out.npos;
        out.npos += (uint32_t)count;
        out.pos_vert.resize(out.npos, -1);
        for (int32_t k = 0; k < count; ++k) out.pos_vert[base + k] = ind[k];
        return KD_LEAF_BIT | (base << 4) | (uint32_t)count;
    }
    if (level + 1 > out.depth) out.depth = level + 1;
    float cut;
    const Cut c = choose_and_partition(ind, count, box, cut);
    // recurse; children return their tight boxes (nanoflann.hpp:889-903)
    const uint32_t self = (uint32_t)out.nodes.size();
    out.nodes.push_back({});
    out.split.push_back(0u);
    Box lb = box, rb = box;
    lb.hi[c.feat] = cut;
    rb.lo[c.feat] = cut;
    const uint32_t r1 = divide(out, left, left + c.mid, lb, level + 1, in_group);
    out.split[self] = out.npos;  // first padded position of child 2
    const uint32_t r2 = divide(out, left + c.mid, right, rb, level + 1, in_group);
    KdNodeH& nd = out.nodes[self];
    nd.divlow = lb.hi[c.feat];
    nd.divhigh = rb.lo[c.feat];
    nd.ref1 = r1 | ((uint32_t)c.feat << KD_FEAT_SHIFT);
    nd.ref2 = r2;
    for (int d = 0; d < 3; ++d)
        nd.lo1[d] = lb.lo[d], nd.hi1[d] = lb.hi[d], nd.lo2[d] = rb.lo[d], nd.hi2[d] = rb.hi[d];
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = lb.lo[d] < rb.lo[d] ? lb.lo[d] : rb.lo[d];
        box.hi[d] = lb.hi[d] > rb.hi[d] ? lb.hi[d] : rb.hi[d];
    }
    return self;
}

// top levels: split here, build the two ranges concurrently (they touch disjoint parts of vind)
inline void KdHost::divide_par(Sub& out, int32_t left, int32_t right, Box& box, int par_levels,
                               int32_t cutoff) {
    const int32_t count = right - left;
    if (par_levels <= 0 || count <= cutoff || count <= (int32_t)KD_GROUP_LIMIT) {
        out.root = divide(out, left, right, box, 0, false);
        out.total_nodes = (uint32_t)out.nodes.size();
        out.total_pos = out.npos;
        return;
    }
    float cut;
    const Cut c = choose_and_partition(vind.data() + left, count, box, cut);
    out.nodes.push_back({});
    out.c1.reset(new Sub());
    out.c2.reset(new Sub());
    Box lb = box, rb = box;
    lb.hi[c.feat] = cut;
    rb.lo[c.feat] = cut;
    auto fut = std::async(std::launch::async, [&]() {
        divide_par(*out.c1, left, left + c.mid, lb, par_levels - 1, cutoff);
    });
    divide_par(*out.c2, left + c.mid, right, rb, par_levels - 1, cutoff);
    fut.get();
    KdNodeH& nd = out.nodes[0];
    nd.divlow = lb.hi[c.feat];
    nd.divhigh = rb.lo[c.feat];
    nd.ref1 = (uint32_t)c.feat << KD_FEAT_SHIFT;  // child refs are set when spliced
    nd.ref2 = 0;
    for (int d = 0; d < 3; ++d)
        nd.lo1[d] = lb.lo[d], nd.hi1[d] = lb.hi[d], nd.lo2[d] = rb.lo[d], nd.hi2[d] = rb.hi[d];
    for (int d = 0; d < 3; ++d) {
        box.lo[d] = lb.lo[d] < rb.lo[d] ? lb.lo[d] : rb.lo[d];
        box.hi[d] = lb.hi[d] > rb.hi[d] ? lb.hi[d] : rb.hi[d];
    }
    out.total_nodes = 1 + out.c1->total_nodes + out.c2->total_nodes;
    out.total_pos = out.c1->total_pos + out.c2->total_pos;
    out.depth = 1 + (out.c1->depth > out.c2->depth ? out.c1->depth : out.c2->depth);
    out.root = 0;
}

// copy a built range into the global pre-order numbering
inline void KdHost::splice(const Sub& s, uint32_t node_off, uint32_t pos_off) {
    if (s.c1) {
        const uint32_t o1 = node_off + 1, o2 = o1 + s.c1->total_nodes;
        const uint32_t p2 = pos_off + s.c1->total_pos;
        KdNodeH nd = s.nodes[0];
        nd.ref1 |= shift_ref(s.c1->root, o1, pos_off);
        nd.ref2 = shift_ref(s.c2->root, o2, p2);
        nodes[node_off] = nd;
        split[node_off] = p2;
        splice(*s.c1, o1, pos_off);
        splice(*s.c2, o2, p2);
        return;
    }
    for (size_t i = 0; i < s.nodes.size(); ++i) {
        KdNodeH nd = s.nodes[i];
        nd.ref1 = shift_ref(nd.ref1, node_off, pos_off);
        nd.ref2 = shift_ref(nd.ref2, node_off, pos_off);
        nodes[node_off + i] = nd;
        split[node_off + i] = s.split[i] + pos_off;
    }
    for (size_t i = 0; i < s.pos_vert.size(); ++i) pos_vert[(size_t)pos_off + i] = s.pos_vert[i];
}

inline void KdHost::build(const float* pts, uint32_t n, int threads) {
    p_ = pts;
    vind.resize(n);
    for (uint32_t i = 0; i < n; ++i) vind[i] = (int32_t)i;
    nodes.clear();
    split.clear();
    pos_vert.clear();
    if (group_max < 10) group_max = 10;
    if (group_max > KD_GROUP_LIMIT) group_max = KD_GROUP_LIMIT;
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
    int par_levels = 0;
    if (threads > 1 && n >= 20000) {
        while ((1 << par_levels) < threads && par_levels < 6) ++par_levels;
        ++par_levels;  // ranges are uneven: a few more tasks than threads
    }
    Sub top;
    divide_par(top, 0, (int32_t)n, box, par_levels, (int32_t)(n / 64 + 1024));
    nodes.resize(top.total_nodes);
    split.resize(top.total_nodes);
    pos_vert.assign((size_t)top.total_pos, -1);
    splice(top, 0, 0);
    depth = top.depth;
    root_ref = top.c1 ? 0u : top.root;
}

inline uint32_t KdHost::quad_of(uint32_t ref, int level) {
    ref &= ~(3u << KD_FEAT_SHIFT);
    {   // a subtree of <= group_max vertices is one packet group (its block starts at a
        // multiple of 4: the builder opened it at exactly this subtree)
        uint32_t first, count;
        subtree_extent(ref, first, count);
        if (count <= group_max) return KD_LEAF_BIT | ((first >> 2) << 4) | ((count + 3u) >> 2);
    }
    if (level + 1 > quad_depth) quad_depth = level + 1;
    const uint32_t self = (uint32_t)quads.size();
    quads.push_back({});
    const float inf = __builtin_inff();
    KdQuadH q;
    for (int k = 0; k < 4; ++k) {
        for (int d = 0; d < 3; ++d) q.box[k][d] = inf, q.box[k][3 + d] = -inf;
        // an unused slot is an EMPTY GROUP, never a quad index: its lower bound is +inf, which
        // still passes `lb <= best * c` while best * c overflows (first descent; queries so far
        // away that every distance is +inf), and a stale index 0 would send the walk back to the
        // root for ever
        q.ref[k] = KD_LEAF_BIT;
        q.pad[k] = 0;
    }
    const KdNodeH nd = nodes[ref];
    const uint32_t child[2] = {nd.ref1 & ~(3u << KD_FEAT_SHIFT), nd.ref2};
    const float* cbox[2][2] = {{nd.lo1, nd.hi1}, {nd.lo2, nd.hi2}};
    uint32_t gref[4] = {0, 0, 0, 0};
    bool used[4] = {false, false, false, false};
    for (int side = 0; side < 2; ++side) {
        uint32_t cfirst, ccount;
        subtree_extent(child[side], cfirst, ccount);
        if (ccount <= group_max) {
            const int k = 2 * side;
            for (int d = 0; d < 3; ++d) q.box[k][d] = cbox[side][0][d], q.box[k][3 + d] = cbox[side][1][d];
            gref[k] = child[side];
            used[k] = true;
        } else {
            const KdNodeH c = nodes[child[side]];
            const int k = 2 * side;
            for (int d = 0; d < 3; ++d) {
                q.box[k][d] = c.lo1[d], q.box[k][3 + d] = c.hi1[d];
                q.box[k + 1][d] = c.lo2[d], q.box[k + 1][3 + d] = c.hi2[d];
            }
            gref[k] = c.ref1 & ~(3u << KD_FEAT_SHIFT);
            gref[k + 1] = c.ref2;
            used[k] = used[k + 1] = true;
        }
    }
    for (int k = 0; k < 4; ++k)
        if (used[k]) q.ref[k] = quad_of(gref[k], level + 1);
    quads[self] = q;
    return self;
}

inline void KdHost::build_quads() {
    // extents of every binary subtree (children follow their parent in pre-order)
    sub_first_.assign(nodes.size(), 0u);
    sub_count_.assign(nodes.size(), 0u);
    for (size_t i = nodes.size(); i-- > 0;) {
        uint32_t f1, c1, f2, c2;
        subtree_extent(nodes[i].ref1 & ~(3u << KD_FEAT_SHIFT), f1, c1);
        subtree_extent(nodes[i].ref2, f2, c2);
        sub_first_[i] = f1;
        sub_count_[i] = c1 + c2;
    }
    quads.clear();
    quads.reserve(nodes.size() / 2 + 8);
    quad_depth = 0;
    quad_root = quad_of(root_ref, 0);
}

}  // namespace sdfk
