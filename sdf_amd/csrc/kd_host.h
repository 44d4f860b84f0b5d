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
// The build is parallel in two ways, both byte-identical to the serial recursion (they have to be:
// the permutation and the numbering are part of the tie rule's data):
//   * the top levels are split breadth-first, every split itself data-parallel (KdTeam): min/max
//     and counts by blocks, and planeSplit's two Hoare passes as what they are — the k-th
//     misplaced element from the left swaps with the k-th from the right;
//   * the ranges below are independent: built concurrently into local arrays and spliced into
//     the global pre-order numbering afterwards.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
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

constexpr uint32_t KD_LEAF_BIT = 0x80000000u;
constexpr uint32_t KD_FEAT_SHIFT = 29;
constexpr uint32_t KD_REF_MASK = 0x1FFFFFFFu;  // strips leaf bit and feat bits
constexpr uint32_t KD_MAX_POINTS = 1u << 23;   // padded positions (< 4 per vertex) fit 25 bits
constexpr uint32_t KD_GROUP_LIMIT = 60;        // chunks of a group fit 4 bits

// Fork-join helper of the builder: `workers` persistent threads.  They sleep on a condition
// variable between builds and spin between the (many, short) phases of one build, where a
// condition-variable wake-up per phase would cost more than the phase.
class KdTeam {
   public:
    explicit KdTeam(int workers) {
        for (int i = 0; i < workers; ++i) th_.emplace_back([this] { loop(); });
    }
    ~KdTeam() {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    void begin() {  // wake the workers into spinning
        {
            std::lock_guard<std::mutex> l(m_);
            awake_ = true;
        }
        cv_.notify_all();
    }
    void end() {  // back to sleep: a generation of its own that carries no job
        std::lock_guard<std::mutex> l(m_);
        awake_ = false;
        const unsigned g = spin_.load(std::memory_order_relaxed) + 1u;
        sleep_gen_.store(g, std::memory_order_relaxed);
        spin_.store(g, std::memory_order_release);
    }
    // fn(b) for every b in [0, nb), dynamically shared among the workers and the caller
    void run(int nb, const std::function<void(int)>& fn) {
        if (nb <= 0) return;
        if (th_.empty() || nb == 1) {
            for (int b = 0; b < nb; ++b) fn(b);
            return;
        }
        fn_ = &fn;
        nb_ = nb;
        err_ = nullptr;
        next_.store(0, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        spin_.fetch_add(1, std::memory_order_release);  // publish the job
        work();
        while (done_.load(std::memory_order_acquire) != (int)th_.size()) std::this_thread::yield();
        // an exception in fn (std::bad_alloc in a range build) ends the job early and surfaces here, on
        // the caller's thread, once every worker has left the job
        if (err_) std::rethrow_exception(err_);
    }
    // begin() now, end() when the scope is left (also by an exception: the workers must not keep spinning)
    struct Session {
        explicit Session(KdTeam& t) : t_(t) { t_.begin(); }
        ~Session() { t_.end(); }
        Session(const Session&) = delete;
        Session& operator=(const Session&) = delete;
        KdTeam& t_;
    };

   private:
    void work() {
        for (;;) {
            const int b = next_.fetch_add(1, std::memory_order_relaxed);
            if (b >= nb_) break;
            try {
                (*fn_)(b);
            } catch (...) {
                std::lock_guard<std::mutex> l(m_);
                if (!err_) err_ = std::current_exception();
                next_.store(nb_, std::memory_order_relaxed);  // nobody starts another block
            }
        }
    }
    void loop() {
        unsigned seen = 0;  // spin_'s initial value: a job published before this thread first looks still counts
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return awake_ || quit_; });
                if (quit_) return;
            }
            for (;;) {  // spin while a build is running
                const unsigned g = spin_.load(std::memory_order_acquire);
                if (g == seen) {
                    std::this_thread::yield();
                    continue;
                }
                seen = g;  // every generation is handled once: a job (run waits for all workers) or end()
                if (sleep_gen_.load(std::memory_order_relaxed) == g) break;
                work();
                done_.fetch_add(1, std::memory_order_release);
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    bool awake_ = false, quit_ = false;
    std::atomic<unsigned> spin_{0}, sleep_gen_{0xFFFFFFFFu};
    std::atomic<int> next_{0}, done_{0};
    const std::function<void(int)>* fn_ = nullptr;
    std::exception_ptr err_;  // first exception of the running job (guarded by m_)
    int nb_ = 0;
};

struct KdHost {
    std::vector<int32_t> vind;      // nanoflann's permutation (dense, size n)
    std::vector<int32_t> pos_vert;  // padded tree position -> vertex index, -1 = sentinel slot
    uint32_t group_max = 24;        // vertices per packet group (10 = one group per leaf)
    std::vector<KdNodeH> nodes;
    std::vector<uint32_t> split;  // parallel to nodes
    uint32_t root_ref = KD_LEAF_BIT;  // empty leaf
    int depth = 0;                    // max internal nodes on a root-to-leaf path
    float bbox[6] = {0, 0, 0, 0, 0, 0};  // min xyz, max xyz (= mesh AABB, sdf.cpp:36-37)

    float t_top_ms = 0, t_ranges_ms = 0, t_splice_ms = 0;  // phases of the last parallel build

    // threads <= 1: serial.  Otherwise up to ~2*threads ranges are built concurrently.
    // bbox_in: the result of bounds() on the same points, if the caller already has it.
    void build(const float* pts, uint32_t n, int threads = 0, const float* bbox_in = nullptr);
    // min xyz, max xyz of the points (the root box of the tree == the mesh AABB, sdf.cpp:36-37);
    // zeros for n == 0.  Data-parallel for large n.
    void bounds(const float* pts, uint32_t n, int threads, float out[6]);
    // The packet groups in position order (= nanoflann's DFS order, spatially coherent): refs
    // KD_LEAF_BIT | (first pos / 4) << 4 | chunks.  The wide packet walk bounds each group, and
    // each run of 64 consecutive groups, by an oriented box (kernels_build.h k_group_obb).
    std::vector<uint32_t> groups;
    void build_groups();

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
    // a node of the breadth-first top of the tree
    struct Top {
        int32_t left = 0, right = 0;
        Box box;  // passed down (nanoflann.hpp:881-888); tight once its subtree is built
        int feat = 0;
        float cut = 0.f;
        int32_t mid = 0;
        int c1 = -1, c2 = -1;  // children in top_ (-1: this range is built by divide())
        Sub sub;               // the built range (leaves of the top)
    };
    std::vector<Top> top_;
    std::unique_ptr<KdTeam> team_;
    void split_level(const std::vector<int>& frontier);
    struct SpliceJob {
        const Sub* s;
        uint32_t node_off, pos_off;
    };
    std::vector<SpliceJob> splice_jobs_;
    std::vector<uint8_t> flag_;           // scratch of split_level, kept across builds
    std::vector<int32_t> scrL_, scrR_;
    void assemble(int t, Sub& out);
    void splice(const Sub& s, uint32_t node_off, uint32_t pos_off);
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
    if (best_spread < 0) {
        // no dimension qualified (all spans zero: every point equal): nanoflann still clamps the
        // cut to the elements of cutfeat = 0 (its second computeMinMax, nanoflann.hpp:936-939)
        fmn = fmx = at(ind[0], 0);
        for (int32_t k = 1; k < count; ++k) {
            const float v = at(ind[k], 0);
            if (v < fmn) fmn = v;
            if (v > fmx) fmx = v;
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

// One level of the top of the tree: every node of `frontier` is split exactly as
// choose_and_partition does it, all nodes at once and each node by blocks.
//   planeSplit's first pass (nanoflann.hpp:971-985) moves `lo` up to the first element >= cut and
//   `hi` down to the last one < cut, swaps, and repeats until they cross: the k-th misplaced
//   element counted from the left meets the k-th counted from the right.  With lim1 = #(< cut)
//   the misplaced ones are known without running the loop: left of lim1 those >= cut, right of
//   it those < cut.  The second pass (986-1001) does the same for <= cut on [lim1, count).
//   (The `!hi` guards only matter for unsigned indices: with hi == 0 the loop ends either way.)
inline void KdHost::split_level(const std::vector<int>& frontier) {
    struct Blk {
        int node;
        int32_t a, b;  // range within the node
        float mn[3], mx[3];
        int32_t n_lt = 0, n_le = 0, nL = 0, nR = 0, offL = 0, offR = 0;
        std::vector<int32_t> E;
    };
    constexpr int32_t BLK = 8192;
    std::vector<Blk> blk;
    std::vector<int> first_blk(frontier.size() + 1, 0);
    for (size_t f = 0; f < frontier.size(); ++f) {
        const Top& T = top_[frontier[f]];
        const int32_t count = T.right - T.left;
        const int32_t nb = (count + BLK - 1) / BLK;
        for (int32_t k = 0; k < nb; ++k) {
            Blk B;
            B.node = (int)f;
            B.a = k * BLK;
            B.b = (k + 1) * BLK < count ? (k + 1) * BLK : count;
            blk.push_back(std::move(B));
        }
        first_blk[f + 1] = (int)blk.size();
    }
    const int nblk = (int)blk.size();
    // A: min / max per dimension (exact, order independent)
    team_->run(nblk, [&](int bi) {
        Blk& B = blk[bi];
        const int32_t* ind = vind.data() + top_[frontier[B.node]].left;
        for (int d = 0; d < 3; ++d) B.mn[d] = B.mx[d] = at(ind[B.a], d);
        for (int32_t k = B.a + 1; k < B.b; ++k)
            for (int d = 0; d < 3; ++d) {
                const float v = at(ind[k], d);
                if (v < B.mn[d]) B.mn[d] = v;
                if (v > B.mx[d]) B.mx[d] = v;
            }
    });
    for (size_t f = 0; f < frontier.size(); ++f) {  // nanoflann.hpp:909-945
        Top& T = top_[frontier[f]];
        float mn[3], mx[3];
        for (int d = 0; d < 3; ++d) mn[d] = blk[first_blk[f]].mn[d], mx[d] = blk[first_blk[f]].mx[d];
        for (int bi = first_blk[f] + 1; bi < first_blk[f + 1]; ++bi)
            for (int d = 0; d < 3; ++d) {
                if (blk[bi].mn[d] < mn[d]) mn[d] = blk[bi].mn[d];
                if (blk[bi].mx[d] > mx[d]) mx[d] = blk[bi].mx[d];
            }
        float span[3], max_span = T.box.hi[0] - T.box.lo[0];
        for (int d = 0; d < 3; ++d) {
            span[d] = T.box.hi[d] - T.box.lo[d];
            if (span[d] > max_span) max_span = span[d];
        }
        const float EPS = 0.00001f;
        float best_spread = -1;
        int feat = 0;
        for (int d = 0; d < 3; ++d)
            if (span[d] > (1 - EPS) * max_span) {
                const float spread = mx[d] - mn[d];
                if (spread > best_spread) feat = d, best_spread = spread;
            }
        float cut = (T.box.lo[feat] + T.box.hi[feat]) / 2;
        if (cut < mn[feat])
            cut = mn[feat];
        else if (cut > mx[feat])
            cut = mx[feat];
        T.feat = feat;
        T.cut = cut;
    }
    // B: lim1 = #(< cut), lim2 = #(<= cut); the comparison results are kept for C (one indirect read less)
    if (flag_.size() < vind.size()) flag_.resize(vind.size()), scrL_.resize(vind.size()), scrR_.resize(vind.size());
    team_->run(nblk, [&](int bi) {
        Blk& B = blk[bi];
        const Top& T = top_[frontier[B.node]];
        const int32_t* ind = vind.data() + T.left;
        uint8_t* fl = flag_.data() + T.left;
        int32_t lt = 0, le = 0;
        for (int32_t k = B.a; k < B.b; ++k) {
            const float v = at(ind[k], T.feat);
            const bool l = v < T.cut;
            fl[k] = l;
            lt += l;
            le += v <= T.cut;
        }
        B.n_lt = lt, B.n_le = le;
    });
    std::vector<int32_t> lim1(frontier.size(), 0), lim2(frontier.size(), 0);
    for (size_t f = 0; f < frontier.size(); ++f)
        for (int bi = first_blk[f]; bi < first_blk[f + 1]; ++bi) lim1[f] += blk[bi].n_lt, lim2[f] += blk[bi].n_le;
    // C: the misplaced elements of the first pass, in ascending position, block by block into scratch
    team_->run(nblk, [&](int bi) {
        Blk& B = blk[bi];
        const Top& T = top_[frontier[B.node]];
        const uint8_t* fl = flag_.data() + T.left;
        int32_t* L = scrL_.data() + T.left + B.a;
        int32_t* R = scrR_.data() + T.left + B.a;
        const int32_t l1 = lim1[B.node];
        int32_t nl = 0, nr = 0;
        const int32_t mid = l1 < B.a ? B.a : (l1 > B.b ? B.b : l1);
        for (int32_t k = B.a; k < mid; ++k)
            if (!fl[k]) L[nl++] = k;
        for (int32_t k = mid; k < B.b; ++k)
            if (fl[k]) R[nr++] = k;
        B.nL = nl, B.nR = nr;
    });
    // D: k-th from the left <-> k-th from the right (block prefix tables instead of one long list)
    struct Job {
        int node;
        int32_t a, b, m;
    };
    std::vector<Job> jobs;
    for (size_t f = 0; f < frontier.size(); ++f) {
        int32_t m = 0, mr = 0;
        for (int bi = first_blk[f]; bi < first_blk[f + 1]; ++bi) {
            blk[bi].offL = m, blk[bi].offR = mr;
            m += blk[bi].nL, mr += blk[bi].nR;
        }
        // (m == mr: as many elements >= cut left of lim1 as elements < cut right of it)
        for (int32_t a = 0; a < m; a += BLK) jobs.push_back({(int)f, a, a + BLK < m ? a + BLK : m, m});
    }
    team_->run((int)jobs.size(), [&](int j) {
        const Job& J = jobs[j];
        const Top& T = top_[frontier[J.node]];
        int32_t* ind = vind.data() + T.left;
        const int b0 = first_blk[J.node], b1 = first_blk[J.node + 1];
        // cursors: block holding the k-th L (ascending) and the (m-1-k)-th R
        int bl = b0;
        while (bl + 1 < b1 && blk[bl + 1].offL <= J.a) ++bl;
        int32_t r = J.m - 1 - J.a;
        int br = b1 - 1;
        while (br > b0 && blk[br].offR > r) --br;
        for (int32_t k = J.a; k < J.b; ++k, --r) {
            while (k >= blk[bl].offL + blk[bl].nL) ++bl;
            while (r < blk[br].offR) --br;
            const int32_t lp = scrL_[(size_t)T.left + blk[bl].a + (k - blk[bl].offL)];
            const int32_t rp = scrR_[(size_t)T.left + blk[br].a + (r - blk[br].offR)];
            const int32_t t = ind[lp];
            ind[lp] = ind[rp];
            ind[rp] = t;
        }
    });
    // E: second pass on [lim1, count): elements <= cut there are == cut, normally a handful
    team_->run(nblk, [&](int bi) {
        Blk& B = blk[bi];
        const Top& T = top_[frontier[B.node]];
        const int32_t* ind = vind.data() + T.left;
        const int32_t l1 = lim1[B.node];
        for (int32_t k = B.a > l1 ? B.a : l1; k < B.b; ++k)
            if (at(ind[k], T.feat) <= T.cut) B.E.push_back(k);
    });
    for (size_t f = 0; f < frontier.size(); ++f) {
        Top& T = top_[frontier[f]];
        int32_t* ind = vind.data() + T.left;
        const int32_t count = T.right - T.left, l1 = lim1[f], l2 = lim2[f];
        std::vector<int32_t> E;
        for (int bi = first_blk[f]; bi < first_blk[f + 1]; ++bi) E.insert(E.end(), blk[bi].E.begin(), blk[bi].E.end());
        // left of lim2: positions not in E are misplaced (ascending); right of it: those in E (descending)
        size_t e = 0, r = E.size();
        for (int32_t pos = l1; pos < l2; ++pos) {
            if (e < E.size() && E[e] == pos) {
                ++e;
                continue;
            }
            --r;  // E[r] >= l2: there are exactly as many of them as gaps left of l2
            const int32_t t = ind[pos];
            ind[pos] = ind[E[r]];
            ind[E[r]] = t;
        }
        if (l1 > count / 2)  // nanoflann.hpp:950-955
            T.mid = l1;
        else if (l2 < count / 2)
            T.mid = l2;
        else
            T.mid = count / 2;
    }
}

// the Sub tree of top node t, bottom-up (same shape divide()'s recursion would give it)
inline void KdHost::assemble(int t, Sub& out) {
    Top& T = top_[t];
    if (T.c1 < 0) {
        out = std::move(T.sub);
        out.total_nodes = (uint32_t)out.nodes.size();
        out.total_pos = out.npos;
        return;
    }
    out.nodes.push_back({});
    out.c1.reset(new Sub());
    out.c2.reset(new Sub());
    assemble(T.c1, *out.c1);
    assemble(T.c2, *out.c2);
    const Box &lb = top_[T.c1].box, &rb = top_[T.c2].box;  // tight boxes of the children
    KdNodeH& nd = out.nodes[0];
    nd.divlow = lb.hi[T.feat];
    nd.divhigh = rb.lo[T.feat];
    nd.ref1 = (uint32_t)T.feat << KD_FEAT_SHIFT;  // child refs are set when spliced
    nd.ref2 = 0;
    for (int d = 0; d < 3; ++d)
        nd.lo1[d] = lb.lo[d], nd.hi1[d] = lb.hi[d], nd.lo2[d] = rb.lo[d], nd.hi2[d] = rb.hi[d];
    for (int d = 0; d < 3; ++d) {
        T.box.lo[d] = lb.lo[d] < rb.lo[d] ? lb.lo[d] : rb.lo[d];
        T.box.hi[d] = lb.hi[d] > rb.hi[d] ? lb.hi[d] : rb.hi[d];
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
    splice_jobs_.push_back({&s, node_off, pos_off});
}

inline void KdHost::bounds(const float* pts, uint32_t n, int threads, float out[6]) {
    for (int d = 0; d < 6; ++d) out[d] = 0.f;
    if (n == 0) return;
    constexpr uint32_t BLK = 16384;
    const uint32_t nb = (threads > 1 && n >= 20000) ? (n + BLK - 1) / BLK : 1u;
    std::vector<float> part(6 * (size_t)nb);
    auto one = [&](int b) {
        const uint32_t a = nb == 1 ? 0u : (uint32_t)b * BLK, e = nb == 1 ? n : (a + BLK < n ? a + BLK : n);
        float lo[3] = {pts[3 * (size_t)a], pts[3 * (size_t)a + 1], pts[3 * (size_t)a + 2]};
        float hi[3] = {lo[0], lo[1], lo[2]};
        for (uint32_t k = a + 1; k < e; ++k)
            for (int d = 0; d < 3; ++d) {
                const float v = pts[3 * (size_t)k + d];
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
        for (int d = 0; d < 3; ++d) part[6 * (size_t)b + d] = lo[d], part[6 * (size_t)b + 3 + d] = hi[d];
    };
    if (nb == 1) {
        one(0);
    } else {
        if (!team_ || team_->size() != threads) team_.reset(new KdTeam(threads - 1));
        KdTeam::Session awake(*team_);
        team_->run((int)nb, one);
    }
    for (int d = 0; d < 6; ++d) out[d] = part[d];
    for (uint32_t b = 1; b < nb; ++b)
        for (int d = 0; d < 3; ++d) {
            if (part[6 * (size_t)b + d] < out[d]) out[d] = part[6 * (size_t)b + d];
            if (part[6 * (size_t)b + 3 + d] > out[3 + d]) out[3 + d] = part[6 * (size_t)b + 3 + d];
        }
}

inline void KdHost::build(const float* pts, uint32_t n, int threads, const float* bbox_in) {
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
    if (bbox_in)
        for (int d = 0; d < 6; ++d) bbox[d] = bbox_in[d];
    else
        bounds(pts, n, threads, bbox);
    for (int d = 0; d < 3; ++d) box.lo[d] = bbox[d], box.hi[d] = bbox[d + 3];
    Sub top;
    if (threads > 1 && n >= 20000) {
        int par_levels = 0;
        while ((1 << par_levels) < threads && par_levels < 6) ++par_levels;
        par_levels += 2;  // ranges are uneven: ~4 tasks per thread, handed out largest first
        const int32_t cutoff = (int32_t)(n / 256 + 1024);
        if (!team_ || team_->size() != threads) team_.reset(new KdTeam(threads - 1));
        const auto tp0 = std::chrono::steady_clock::now();
        auto tp1 = tp0;
        {   // the team stays awake for the top levels and the ranges
        KdTeam::Session awake(*team_);
        top_.clear();
        top_.emplace_back();
        top_[0].left = 0, top_[0].right = (int32_t)n, top_[0].box = box;
        std::vector<int> frontier{0}, ranges;
        for (int level = 0; !frontier.empty(); ++level) {
            std::vector<int> split;
            for (int t : frontier) {
                const int32_t count = top_[t].right - top_[t].left;
                if (level >= par_levels || count <= cutoff || count <= (int32_t)KD_GROUP_LIMIT)
                    ranges.push_back(t);
                else
                    split.push_back(t);
            }
            if (!split.empty()) split_level(split);
            frontier.clear();
            for (int t : split) {
                // (copies: emplace_back below may move the array)
                const int32_t left = top_[t].left, right = top_[t].right, mid = top_[t].mid;
                const Box pbox = top_[t].box;
                const int feat = top_[t].feat;
                const float cut = top_[t].cut;
                for (int side = 0; side < 2; ++side) {
                    top_.emplace_back();
                    Top& C = top_.back();
                    C.left = side ? left + mid : left;
                    C.right = side ? right : left + mid;
                    C.box = pbox;  // parent box narrowed at the cut (nanoflann.hpp:881-888)
                    (side ? C.box.lo : C.box.hi)[feat] = cut;
                    frontier.push_back((int)top_.size() - 1);
                }
                top_[t].c1 = (int)top_.size() - 2, top_[t].c2 = (int)top_.size() - 1;
            }
        }
        tp1 = std::chrono::steady_clock::now();
        // the ranges below the top are independent (disjoint parts of vind): largest first
        std::sort(ranges.begin(), ranges.end(), [&](int x, int y) {
            return top_[x].right - top_[x].left > top_[y].right - top_[y].left;
        });
        team_->run((int)ranges.size(), [&](int i) {
            Top& T = top_[ranges[i]];
            const size_t cnt = (size_t)(T.right - T.left);
            T.sub.nodes.reserve(cnt / 4 + 16);
            T.sub.split.reserve(cnt / 4 + 16);
            T.sub.pos_vert.reserve(cnt + cnt / 4 + 64);
            T.sub.root = divide(T.sub, T.left, T.right, T.box, 0, false);
        });
        }
        const auto tp2 = std::chrono::steady_clock::now();
        t_top_ms = std::chrono::duration<float, std::milli>(tp1 - tp0).count();
        t_ranges_ms = std::chrono::duration<float, std::milli>(tp2 - tp1).count();
        assemble(0, top);
        box = top_[0].box;
        top_.clear();
    } else {
        top.root = divide(top, 0, (int32_t)n, box, 0, false);
        top.total_nodes = (uint32_t)top.nodes.size();
        top.total_pos = top.npos;
    }
    const auto ts0 = std::chrono::steady_clock::now();
    nodes.resize(top.total_nodes);
    split.resize(top.total_nodes);
    pos_vert.resize((size_t)top.total_pos);
    splice_jobs_.clear();
    splice(top, 0, 0);  // inner nodes of the top; the built ranges are copied below, concurrently
    auto copy_range = [&](int j) {
        const SpliceJob& J = splice_jobs_[j];
        const Sub& r = *J.s;
        for (size_t i = 0; i < r.nodes.size(); ++i) {
            KdNodeH nd = r.nodes[i];
            nd.ref1 = shift_ref(nd.ref1, J.node_off, J.pos_off);
            nd.ref2 = shift_ref(nd.ref2, J.node_off, J.pos_off);
            nodes[J.node_off + i] = nd;
            split[J.node_off + i] = r.split[i] + J.pos_off;
        }
        for (size_t i = 0; i < r.pos_vert.size(); ++i) pos_vert[(size_t)J.pos_off + i] = r.pos_vert[i];
    };
    if (team_ && splice_jobs_.size() > 1) {
        KdTeam::Session awake(*team_);
        team_->run((int)splice_jobs_.size(), copy_range);
    } else {
        for (size_t j = 0; j < splice_jobs_.size(); ++j) copy_range((int)j);
    }
    t_splice_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - ts0).count();
    depth = top.depth;
    root_ref = top.c1 ? 0u : top.root;
}

inline void KdHost::build_groups() {
    groups.clear();
    if (pos_vert.empty()) return;
    // extents of every binary subtree (children follow their parent in pre-order: one reverse pass)
    sub_first_.assign(nodes.size(), 0u);
    sub_count_.assign(nodes.size(), 0u);
    for (size_t i = nodes.size(); i-- > 0;) {
        uint32_t f1, c1, f2, c2;
        subtree_extent(nodes[i].ref1 & ~(3u << KD_FEAT_SHIFT), f1, c1);
        subtree_extent(nodes[i].ref2, f2, c2);
        sub_first_[i] = f1;
        sub_count_[i] = c1 + c2;
    }
    // maximal subtrees with <= group_max vertices, left to right (explicit stack: child 2 below child 1)
    std::vector<uint32_t> stack{root_ref & ~(3u << KD_FEAT_SHIFT)};
    while (!stack.empty()) {
        const uint32_t ref = stack.back();
        stack.pop_back();
        uint32_t first, count;
        subtree_extent(ref, first, count);
        if (count <= group_max) {
            if (count) groups.push_back(KD_LEAF_BIT | ((first >> 2) << 4) | ((count + 3u) >> 2));
            continue;
        }
        stack.push_back(nodes[ref].ref2);
        stack.push_back(nodes[ref].ref1 & ~(3u << KD_FEAT_SHIFT));
    }
}

}  // namespace sdfk
