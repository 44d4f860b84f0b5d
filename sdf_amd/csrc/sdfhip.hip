// sdfhip.hip — MI355X (gfx950) implementation of the batched point-query path of sxyu/sdf
// behind the C ABI of include/sdfhip.h.  Written from scratch for CDNA4; compile with
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
// (-ffp-contract=off is REQUIRED: results must be bit-identical to a non-FMA x86-64 build
// of the reference, CMakeLists.txt:33).  "file:line" citations are into /root/reference.
//
// Structures (all resident in HBM, sized to sit in L2 / Infinity Cache):
//   kd_pn      64 B / internal node of the strict nanoflann-equivalent vertex kd-tree
//              (tight child boxes + divlow/divhigh + child refs), kd_split 4 B / node
//   kd_pts4    16 B / padded tree position, in tree (vind) order: a leaf is a contiguous run
//   geo        64 B / face: a, b, c, unit normal, area (world frame) for dist_point2tri
//   rface/fbox 48 + 24 B / face: a', b', c', n' and the AABB in the rotated ray-cast frame
//   gr_*       column grid over (x', y') of that frame: cell -> faces whose AABB touches it
//              (candidates of the z+/z- parity rays, one O(1) lookup per query)
//   rnodes     32 B / node: threaded (stackless, pre-order + skip link) LBVH over the
//              rotated-frame face AABBs (x+ tie-break ray); Morton codes -> LDS radix sort ->
//              Karras hierarchy, refit on update()
//   kd_qn      128 B / 4-wide node of the packet pass (two kd levels collapsed, leaf groups)
// Two query kernels: k_query (one lane per query, exact nanoflann emulation, stack in LDS with
// [level][lane] layout) and k_query_packet (one wave per 64 Morton-adjacent queries, the
// dominant kernel).  DESIGN.md describes both and why their results are the reference's.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <future>
#include <mutex>
#include <thread>
#include <random>
#include <string>
#include <vector>

#include "../../include/sdfhip.h"
#include "geom.h"
#include "kd_host.h"

using sdfk::V3;

// =========================================================================================
// errors
// =========================================================================================
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPC(expr)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? SDFHIP_E_NOMEM : SDFHIP_E_HIP,          \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                        __LINE__);                                                          \
    } while (0)

// =========================================================================================
// device-side views
// =========================================================================================
struct RayNode {  // 32 B
    float mnx, mny, mxx, mxy;
    float mnz, mxz;
    uint32_t skip;  // next node when this subtree is skipped (pre-order index)
    uint32_t face;  // face id for a leaf, NOFACE for an internal node
};
constexpr uint32_t NOFACE = 0xFFFFFFFFu;

struct MeshDev {
    // vertex kd-tree: 64 B nodes {box1 lo/hi, box2 lo/hi, divlow, divhigh, ref1|feat, ref2}
    const float4* kd_pn;
    const uint32_t* kd_split;
    const float4* kd_qn;  // 4-wide collapse for the packet pass: 128 B {box[4][6], ref[4], pad}
    uint32_t kd_qroot;
    int fast;  // "fast_nn": GPU-built tree, exact distances, ties go to the lowest tree position
    const int32_t* kd_vind;
    uint32_t kd_root;
    float bb[6];  // mesh AABB == kd root bbox
    // packet traversal extras: vertices padded to 16 B, ambiguity factors c = 1 + W and ~c^2
    const float4* kd_pts4;
    uint32_t kd_npos;       // padded tree positions (sentinel slots included)
    uint32_t heavy_budget;  // packet pass: a wave that has evaluated more chunks than this gives up
    float amb_c, amb_c2;
    // adjacency + face geometry
    const uint32_t* adj_off;
    const int32_t* adj_faces;
    const float4* geo;
    // ray structure
    const float4* rnodes;
    uint32_t n_rnodes;
    const float4* rface;
    // column grid over (x', y') of the ray-cast frame: cell -> faces whose rotated AABB touches it
    const float* fbox;          // per face: rotated AABB min xyz, max xyz
    const uint32_t* gr_start;   // [g*g + 1]
    const uint32_t* gr_items;   // face ids, cell by cell
    const uint32_t* gr_big;     // faces spanning too many cells: tested by every query
    uint32_t gr_nbig;
    int ray_lbvh;  // packet kernel: walk the LBVH cooperatively instead of the grid (long lists)
    int gr_g;
    float gr_o0, gr_o1, gr_inv0, gr_inv1;
    float R[9];
};

constexpr uint32_t KD_LEAF = sdfk::KD_LEAF_BIT;
constexpr uint32_t KD_FEATM = 3u << sdfk::KD_FEAT_SHIFT;
constexpr uint32_t KD_MARK = sdfk::KD_REF_MASK;  // reserved internal index = restore marker

constexpr int QBLOCK = 128;       // threads per block of the query kernels (2 waves)
constexpr int LDS_DEPTH_MAX = 40; // deeper kd-trees use the global-memory stack variant

// =========================================================================================
// kd nearest vertex: exact iterative form of nanoflann's searchLevel
// (nanoflann.hpp:1346-1409) with KNNResultSet<float>(1) (142-205), eps = 0.
//
// The recursion's state is (node, mindistsq, dists[3]).  dists[] is kept in registers and
// restored exactly as the recursion does (nanoflann.hpp:1396-1407) by "restore markers":
//   push    when descending to the near child: {far ref | feat, mindistsq', cut_dist}
//   pop     a far entry is tested with `mindistsq' <= worst` at pop time, which is when the
//           recursion evaluates it (the near subtree has completely returned); if taken, its
//           slot is overwritten with a marker {MARK | feat, old dists[feat]} that restores
//           dists[feat] once the far subtree has returned.
// Hence at most one slot per internal node on the current root-to-leaf path: capacity =
// kd depth.  `stk` points at this lane's word of slot 0; consecutive words are `stride`
// dwords apart.
// =========================================================================================
// Exact no-op skipping: a subtree whose tight vertex box is at least `best` away (box_lb below:
// a rigorous lower bound of every computed vertex distance in it) cannot change (best, index)
// — nanoflann only replaces on dist < best — and dists[] is restored after any subtree, so not
// entering it leaves the search state exactly as nanoflann's own visit would.
__device__ __forceinline__ float box_lb(float qx, float qy, float qz, float lx, float ly, float lz,
                                        float hx, float hy, float hz) {
    const float gx = fmaxf(fmaxf(lx - qx, qx - hx), 0.f);
    const float gy = fmaxf(fmaxf(ly - qy, qy - hy), 0.f);
    const float gz = fmaxf(fmaxf(lz - qz, qz - hz), 0.f);
    return (gx * gx + gy * gy) + gz * gz;
}

template <class PTR>
__device__ __forceinline__ int kd_nearest(const MeshDev& M, float qx, float qy, float qz, PTR stk,
                                          uint32_t stride, float& best_out) {
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, mind = 0.f;
    // computeInitialDistances (nanoflann.hpp:1005-1022)
    if (qx < M.bb[0]) {
        d0 = (qx - M.bb[0]) * (qx - M.bb[0]);
        mind += d0;
    }
    if (qx > M.bb[3]) {
        d0 = (qx - M.bb[3]) * (qx - M.bb[3]);
        mind += d0;
    }
    if (qy < M.bb[1]) {
        d1 = (qy - M.bb[1]) * (qy - M.bb[1]);
        mind += d1;
    }
    if (qy > M.bb[4]) {
        d1 = (qy - M.bb[4]) * (qy - M.bb[4]);
        mind += d1;
    }
    if (qz < M.bb[2]) {
        d2 = (qz - M.bb[2]) * (qz - M.bb[2]);
        mind += d2;
    }
    if (qz > M.bb[5]) {
        d2 = (qz - M.bb[5]) * (qz - M.bb[5]);
        mind += d2;
    }
    float best = FLT_MAX;  // KNNResultSet::init (nanoflann.hpp:158-164)
    int bestpos = -1;
    uint32_t cur = M.kd_root;
    int sp = 0;
    for (;;) {
        bool near_skipped = false;
        while (!(cur & KD_LEAF)) {
            const float4* pn = M.kd_pn + 4 * (size_t)cur;
            const float4 bA = pn[0], bB = pn[1], bC = pn[2], nf = pn[3];
            const uint32_t ndz = __float_as_uint(nf.z), ndw = __float_as_uint(nf.w);
            const float lbc1 = box_lb(qx, qy, qz, bA.x, bA.y, bA.z, bA.w, bB.x, bB.y);
            const float lbc2 = box_lb(qx, qy, qz, bB.z, bB.w, bC.x, bC.y, bC.z, bC.w);
            const float divlow = nf.x, divhigh = nf.y;
            const uint32_t feat = (ndz >> sdfk::KD_FEAT_SHIFT) & 3u;
            const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
            const float diff1 = val - divlow, diff2 = val - divhigh;
            const bool lowside = (diff1 + diff2) < 0;
            const uint32_t r1 = ndz & ~KD_FEATM, r2 = ndw;
            const uint32_t farc = lowside ? r2 : r1;
            const float pl = lowside ? divhigh : divlow;
            const float cut = (val - pl) * (val - pl);
            const float dst = feat == 0 ? d0 : (feat == 1 ? d1 : d2);
            const float m2 = (mind + cut) - dst;  // nanoflann.hpp:1397
            if ((lowside ? lbc2 : lbc1) < best) {  // far subtree could still matter
                stk[(sp * 3 + 0) * stride] = farc | (feat << sdfk::KD_FEAT_SHIFT);
                stk[(sp * 3 + 1) * stride] = __float_as_uint(m2);
                stk[(sp * 3 + 2) * stride] = __float_as_uint(cut);
                ++sp;
            }
            if (!((lowside ? lbc1 : lbc2) < best)) {  // near subtree is a no-op
                near_skipped = true;
                break;
            }
            cur = lowside ? r1 : r2;
        }
        if (!near_skipped) {  // leaf (nanoflann.hpp:1350-1367); metric ((0+dx^2)+dy^2)+dz^2 (432-440)
            const uint32_t left = (cur & sdfk::KD_REF_MASK) >> 4, cnt = cur & 15u;
            const float4* p = M.kd_pts4 + left;
            for (uint32_t k = 0; k < cnt; ++k) {
                const float4 pk = p[k];
                const float dx = qx - pk.x, dy = qy - pk.y, dz = qz - pk.z;
                const float dist = (dx * dx + dy * dy) + dz * dz;
                if (dist < best) {  // strict: first visited wins ties (nanoflann.hpp:184)
                    best = dist;
                    bestpos = (int)(left + k);
                }
            }
        }
        for (;;) {
            if (sp == 0) {
                best_out = best;
                return bestpos;
            }
            --sp;
            const uint32_t w0 = stk[(sp * 3 + 0) * stride];
            const uint32_t feat = (w0 >> sdfk::KD_FEAT_SHIFT) & 3u;
            const uint32_t ref = w0 & ~KD_FEATM;
            const float w1 = __uint_as_float(stk[(sp * 3 + 1) * stride]);
            if (ref == KD_MARK) {  // dists[idx] = dst (nanoflann.hpp:1407)
                if (feat == 0)
                    d0 = w1;
                else if (feat == 1)
                    d1 = w1;
                else
                    d2 = w1;
                continue;
            }
            if (w1 <= best) {  // mindistsq * epsError <= worstDist(), epsError == 1
                const float cut = __uint_as_float(stk[(sp * 3 + 2) * stride]);
                float dst;
                if (feat == 0) {
                    dst = d0;
                    d0 = cut;
                } else if (feat == 1) {
                    dst = d1;
                    d1 = cut;
                } else {
                    dst = d2;
                    d2 = cut;
                }
                stk[(sp * 3 + 0) * stride] = KD_MARK | (feat << sdfk::KD_FEAT_SHIFT);
                stk[(sp * 3 + 1) * stride] = __float_as_uint(dst);
                ++sp;
                mind = w1;
                cur = ref;
                break;
            }
        }
    }
}

// =========================================================================================
// ray parity (Impl::_raycast, sdf.cpp:276-328).
// Result-relevant per face: closed-interval Overlap of the degenerate query
// box with the face's rotated AABB (RTree.h:1427-1438) && facing test && bary2d >= 0.
// The z+ and z- rays share (x,y), the facing dot product and bary2d, so they are one pass over
// the column grid's candidates; the x+ tie-break ray walks the threaded LBVH.
// =========================================================================================
__device__ __forceinline__ int ray_x_pos(const MeshDev& M, V3 q) {
    int c = 0;
    uint32_t i = 0;
    const uint32_t n = M.n_rnodes;
    while (i < n) {
        const float4 a = M.rnodes[2 * (size_t)i];
        const float4 b = M.rnodes[2 * (size_t)i + 1];
        // box: x in [q.x, FLT_MAX], y and z pinned
        const bool hit = !(q.y > a.w || a.y > q.y) && !(q.z > b.y || b.x > q.z) &&
                         !(q.x > a.z || a.x > FLT_MAX);
        const uint32_t skip = __float_as_uint(b.z), face = __float_as_uint(b.w);
        if (!hit) {
            i = skip;
            continue;
        }
        if (face != NOFACE) {
            const float4 f0 = M.rface[3 * (size_t)face];
            const float4 f1 = M.rface[3 * (size_t)face + 1];
            const float4 f2 = M.rface[3 * (size_t)face + 2];
            const V3 nr = {f2.y, f2.z, f2.w};
            const V3 dv = sdfk::sub(q, V3{f0.x, f0.y, f0.z});
            const float s = sdfk::dot(nr, dv) * nr.x;  // ax_idx = 0, ax_inv = false
            if (!(s > 0.f)) {
                // segment<2>(1): (y, z)
                if (sdfk::bary2d_inside(q.y, q.z, f0.y, f0.z, f1.x, f1.y, f1.w, f2.x)) c ^= 1;
            }
        }
        i = i + 1;
    }
    return c;
}

// Column grid.  The z+ / z- rays of a query are the column (q.x, q.y): the only faces that can
// pass the Overlap test are those whose rotated AABB contains (q.x, q.y) in x and y, and every
// such face is registered in the cell of (q.x, q.y): grid_cell is monotone non-decreasing in v
// (subtract, multiply by a positive constant, floor, clamp), so mn <= q <= mx implies
// cell(mn) <= cell(q) <= cell(mx).  Each lane walks its own cell's list — O(1) to find, a
// handful of faces long — and applies the reference's exact per-face predicate; XOR parity is
// order independent, so the result equals the R-tree search's (sdf.cpp:292-315).
__device__ __forceinline__ int grid_cell(float v, float o, float inv, int g) {
    const float f = floorf((v - o) * inv);
    if (!(f > 0.f)) return 0;  // also NaN
    if (f >= (float)(g - 1)) return g - 1;
    return (int)f;
}
__device__ __forceinline__ void ray_face_z(const MeshDev& M, V3 q, uint32_t face, int& zp, int& zm) {
    const float* bx = M.fbox + 6 * (size_t)face;
    const float mnx = bx[0], mny = bx[1], mnz = bx[2], mxx = bx[3], mxy = bx[4], mxz = bx[5];
    if ((q.x > mxx || mnx > q.x) || (q.y > mxy || mny > q.y)) return;  // Overlap, x and y
    const bool okp = !(q.z > mxz || mnz > FLT_MAX);   // box [q.z, FLT_MAX]
    const bool okm = !(-FLT_MAX > mxz || mnz > q.z);  // box [-FLT_MAX, q.z]
    const float4 f0 = M.rface[3 * (size_t)face];      // a'.xyz b'.x
    const float4 f1 = M.rface[3 * (size_t)face + 1];  // b'.yz c'.xy
    const float4 f2 = M.rface[3 * (size_t)face + 2];  // c'.z n'.xyz
    const V3 nr = {f2.y, f2.z, f2.w};
    const V3 dv = sdfk::sub(q, V3{f0.x, f0.y, f0.z});
    const float s = sdfk::dot(nr, dv) * nr.z;  // sdf.cpp:301-302, ax_idx = 2
    const bool pos = s > 0.f;                  // z- needs pos, z+ needs !pos
    if (pos ? okm : okp) {
        if (sdfk::bary2d_inside(q.x, q.y, f0.x, f0.y, f0.w, f1.x, f1.z, f1.w)) {
            if (pos)
                zm ^= 1;
            else
                zp ^= 1;
        }
    }
}
__device__ __forceinline__ void ray_z_pair_grid(const MeshDev& M, V3 q, int& zp, int& zm) {
    zp = 0;
    zm = 0;
    const int cx = grid_cell(q.x, M.gr_o0, M.gr_inv0, M.gr_g), cy = grid_cell(q.y, M.gr_o1, M.gr_inv1, M.gr_g);
    const uint32_t c = (uint32_t)cy * (uint32_t)M.gr_g + (uint32_t)cx;
    const uint32_t e = M.gr_start[c + 1];
    for (uint32_t k = M.gr_start[c]; k < e; ++k) ray_face_z(M, q, M.gr_items[k], zp, zm);
    for (uint32_t k = 0; k < M.gr_nbig; ++k) ray_face_z(M, q, M.gr_big[k], zp, zm);
}

// Wave-cooperative walk of the threaded LBVH for 64 Morton-adjacent queries (used instead of
// the column grid when faces are large against the grid, i.e. the lists are long): the walk is
// wave-uniform (a node is entered when ANY lane's point is inside its xy box; nodes and faces
// come in by scalar loads) and each lane applies the exact per-face predicate to the leaves
// whose box it is inside.  XOR parity is order independent and lanes only count faces that pass
// their own Overlap test, so the result per lane equals ray_z_pair's.
__device__ __forceinline__ void ray_z_pair_packet(const MeshDev& M, V3 q, bool active, int& zp, int& zm) {
    zp = 0;
    zm = 0;
    uint32_t i = 0;
    const uint32_t n = M.n_rnodes;
    while (i < n) {
        i = __builtin_amdgcn_readfirstlane(i);
        const float4 a = M.rnodes[2 * (size_t)i];
        const float4 b = M.rnodes[2 * (size_t)i + 1];
        const bool hit = active && !(q.x > a.z || a.x > q.x) && !(q.y > a.w || a.y > q.y);
        const uint32_t skip = __float_as_uint(b.z), face = __float_as_uint(b.w);
        if (__ballot(hit) == 0ull) {
            i = skip;
            continue;
        }
        if (face != NOFACE) {
            const float4 f0 = M.rface[3 * (size_t)face];      // a'.xyz b'.x
            const float4 f1 = M.rface[3 * (size_t)face + 1];  // b'.yz c'.xy
            const float4 f2 = M.rface[3 * (size_t)face + 2];  // c'.z n'.xyz
            if (hit) {
                const bool okp = !(q.z > b.y || b.x > FLT_MAX);
                const bool okm = !(-FLT_MAX > b.y || b.x > q.z);
                const V3 nr = {f2.y, f2.z, f2.w};
                const V3 dv = sdfk::sub(q, V3{f0.x, f0.y, f0.z});
                const float s = sdfk::dot(nr, dv) * nr.z;
                const bool pos = s > 0.f;
                if (pos ? okm : okp) {
                    if (sdfk::bary2d_inside(q.x, q.y, f0.x, f0.y, f0.w, f1.x, f1.z, f1.w)) {
                        if (pos)
                            zm ^= 1;
                        else
                            zp ^= 1;
                    }
                }
            }
        }
        i = i + 1;
    }
}

// _raycast for a whole wave; lanes with valid == false take no part
__device__ __forceinline__ float raycast_packet(const MeshDev& M, V3 p, bool valid) {
    const bool inbox = valid && !(p.x < M.bb[0] || p.x > M.bb[3] || p.y < M.bb[1] || p.y > M.bb[4] ||
                                  p.z < M.bb[2] || p.z > M.bb[5]);  // sdf.cpp:278-283
    if (__ballot(inbox) == 0ull) return -1.f;
    const V3 q = sdfk::mulR(p, M.R);
    int zp = 0, zm = 0;
    if (M.ray_lbvh)
        ray_z_pair_packet(M, q, inbox, zp, zm);
    else if (inbox)
        ray_z_pair_grid(M, q, zp, zm);
    int result = zp + zm;
    if (inbox && result == 1) result += ray_x_pos(M, q);  // tiebreaker (sdf.cpp:326), rare
    return (inbox && result > 1) ? 1.0f : -1.0f;
}

// returns +1 / -1 (sdf.cpp:327); detail (optional) = {z+, z-, x+|-1}
__device__ __forceinline__ float raycast(const MeshDev& M, V3 p, int* detail) {
    if (detail) detail[0] = detail[1] = detail[2] = -1;
    // closed mesh AABB (sdf.cpp:278-283)
    if (p.x < M.bb[0] || p.x > M.bb[3] || p.y < M.bb[1] || p.y > M.bb[4] || p.z < M.bb[2] ||
        p.z > M.bb[5])
        return -1.f;
    const V3 q = sdfk::mulR(p, M.R);
    int zp, zm;
    ray_z_pair_grid(M, q, zp, zm);
    int result = zp + zm;
    if (detail) detail[0] = zp, detail[1] = zm;
    if (result == 1) {  // tiebreaker (sdf.cpp:326)
        const int xp = ray_x_pos(M, q);
        if (detail) detail[2] = xp;
        result += xp;
    }
    return result > 1 ? 1.0f : -1.0f;
}

// =========================================================================================
// query kernels
// =========================================================================================
enum { Q_NN = 0, Q_CONTAINS = 1, Q_SDF = 2, Q_DETAIL = 3 };

// Impl::calc per-point body after the sign (sdf.cpp:112-153)
// `pos` = position of the nearest vertex in tree order (-1: none)
template <bool ROBUST>
__device__ __forceinline__ float calc_tail(const MeshDev& M, V3 p, bool trunc_aabb, float sign,
                                           int pos) {
    const float DIST_EPS = ROBUST ? 0.f : 1e-5f;
    float min_dist = FLT_MAX;
    if (trunc_aabb) {
        // the `continue` at sdf.cpp:120 only continues the t-loop: execution falls through
        if (p.x < M.bb[0] || p.x > M.bb[3] || p.y < M.bb[1] || p.y > M.bb[4] || p.z < M.bb[2] ||
            p.z > M.bb[5])
            min_dist = -FLT_MAX;
    }
    V3 avg = {0.f, 0.f, 0.f};
    V3 vnn = {0.f, 0.f, 0.f};
    if (pos >= 0) {
        const int vi = M.kd_vind[pos];
        const float4 pv = M.kd_pts4[pos];
        vnn = {pv.x, pv.y, pv.z};
        const uint32_t fb = M.adj_off[vi], fe = M.adj_off[vi + 1];
        for (uint32_t k = fb; k < fe; ++k) {
            const int f = M.adj_faces[k];
            const float4 g0 = M.geo[4 * (size_t)f];      // a.xyz b.x
            const float4 g1 = M.geo[4 * (size_t)f + 1];  // b.yz c.xy
            const float4 g2 = M.geo[4 * (size_t)f + 2];  // c.z n.xyz
            const float4 g3 = M.geo[4 * (size_t)f + 3];  // area
            const V3 n = {g2.y, g2.z, g2.w};
            const float d = sdfk::dist_point2tri(p, V3{g0.x, g0.y, g0.z}, V3{g0.w, g1.x, g1.y},
                                                 V3{g1.z, g1.w, g2.x}, n, g3.x);
            if (d < min_dist - DIST_EPS) {  // sdf.cpp:139-144
                min_dist = d;
                if (!ROBUST) avg = n;
            } else if (!ROBUST && d < min_dist + DIST_EPS) {
                avg = {avg.x + n.x, avg.y + n.y, avg.z + n.z};
            }
        }
    }
    min_dist = __builtin_sqrtf(min_dist);
    if (ROBUST) {
        min_dist *= sign;
    } else if (pos >= 0 && sdfk::dot(avg, sdfk::sub(p, vnn)) > 0) {
        min_dist = -min_dist;
    }
    return min_dist;
}

template <bool ROBUST, class PTR>
__device__ __forceinline__ float calc_point(const MeshDev& M, V3 p, bool trunc_aabb, float sign,
                                            PTR stk, uint32_t stride) {
    float bestd;
    const int pos = kd_nearest(M, p.x, p.y, p.z, stk, stride, bestd);
    return calc_tail<ROBUST>(M, p, trunc_aabb, sign, pos);
}

template <int KIND, bool ROBUST, bool GSTACK>
__global__ __launch_bounds__(QBLOCK) void k_query(MeshDev M, const float* __restrict__ pts,
                                                  const uint32_t* __restrict__ perm, int64_t n,
                                                  int trunc_aabb, void* __restrict__ out,
                                                  uint32_t* gstack, uint32_t depth_cap,
                                                  const uint32_t* __restrict__ n_dev) {
    extern __shared__ uint32_t lds[];
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t first = tid, step = nthreads;
    if (n_dev) {
        // deferred near-tie queries of the packet kernel: the count lives on the device.  These
        // queries are sparse (incoherent) and some are very long (symmetric spots with hundreds
        // of near-equidistant vertices), so only every 4th lane takes one: less divergence
        // serialisation per wave, and the launch is tail-latency bound anyway.
        n = (int64_t)*n_dev;
        if (tid & 3) return;
        first = tid >> 2;
        step = nthreads >> 2;
    }
    for (int64_t i = first; i < n; i += step) {
        const int64_t src = perm ? (int64_t)perm[i] : i;
        const V3 p = {pts[3 * src], pts[3 * src + 1], pts[3 * src + 2]};
        if (KIND == Q_NN) {
            float bd;
            int pos;
            if (GSTACK)
                pos = kd_nearest(M, p.x, p.y, p.z, gstack + tid, (uint32_t)nthreads, bd);
            else
                pos = kd_nearest(M, p.x, p.y, p.z,
                                 lds + (threadIdx.x >> 6) * depth_cap * 192 + (threadIdx.x & 63),
                                 64u, bd);
            ((int32_t*)out)[src] = pos >= 0 ? M.kd_vind[pos] : -1;
        } else if (KIND == Q_CONTAINS && ROBUST) {
            ((uint8_t*)out)[src] = raycast(M, p, nullptr) >= 0.0f;  // sdf.cpp:166
        } else if (KIND == Q_DETAIL) {
            int d[3];
            raycast(M, p, d);
            int32_t* o = (int32_t*)out + 3 * src;
            o[0] = d[0], o[1] = d[1], o[2] = d[2];
        } else {
            float sign = 0.f;
            if (ROBUST) sign = raycast(M, p, nullptr);  // sdf.cpp:110
            const bool trunc = (KIND == Q_CONTAINS) ? true : (trunc_aabb != 0);
            float v;
            if (GSTACK)
                v = calc_point<ROBUST>(M, p, trunc, sign, gstack + tid, (uint32_t)nthreads);
            else
                v = calc_point<ROBUST>(
                    M, p, trunc, sign,
                    lds + (threadIdx.x >> 6) * depth_cap * 192 + (threadIdx.x & 63), 64u);
            if (KIND == Q_CONTAINS)
                ((uint8_t*)out)[src] = v >= 0;  // sdf.cpp:171-172 (NaN -> false)
            else
                ((float*)out)[src] = v;
        }
    }
}


// =========================================================================================
// Packet nearest-vertex search: one WAVE walks the kd-tree for 64 Morton-adjacent queries.
//
// Far-field queries make nanoflann's per-query search visit ~100 leaves / ~500 vertices, and
// 64 independent lanes doing that diverge badly (29 % VALU lane utilisation measured).  Here
// the traversal is wave-uniform: the node stream is read with scalar loads, the stack is one
// small LDS array per wave, every lane evaluates every visited vertex (no divergence).  A
// subtree is skipped only when, for every lane, the tight box of its vertices is farther
// than the lane's current best * c:  LB = ((gx^2+gy^2)+gz^2) with g = per-axis gap to the
// box is a rigorous lower bound of every computed vertex distance in that box (rounding is
// monotone and the association equals the metric's, nanoflann.hpp:432-440).
//
// What this pass knows exactly per lane when it ends: the global fp32 minimum `best`, its
// (first found) vertex `holder`, and `second` = the smallest distance found that is >= best
// among all OTHER vertices whose distance is <= best * c.
//
// Equivalence with nanoflann (DESIGN.md "nn: packet pass + strict fallback"): nanoflann prunes
// a far subtree S iff mindistsq(S) > worst (nanoflann.hpp:1399).  mindistsq(S) is an fp32
// chain that never exceeds dist(p) * (1 + w_th) for any vertex p in S, with
// w_th = 2^-24 * (3.02 * depth + 11.2).  With c >= (1 + w_th) / (1 - 2^-24): if no other
// vertex lies within best * c (second > best * c), the unique minimiser cannot be pruned
// before it is visited and no other vertex can tie or beat it, so nanoflann returns `holder`.
// Otherwise (exact ties, near ties within ~5e-6 relative, underflow-range distances, NaNs)
// the query is "ambiguous" and is appended to a fallback list that the strict per-lane
// kernel (kd_nearest: exact emulation) processes afterwards.  Parity is by construction.
// =========================================================================================
#ifndef SDFK_PBLOCK
#define SDFK_PBLOCK 64
#endif
constexpr int PBLOCK = SDFK_PBLOCK;


// result of the packet pass for one lane
constexpr uint32_t H_NONE = 0x7FFFFFFFu;
struct PacketBest {
    float b;     // global fp32 minimum over all vertices
    float s;     // second smallest distance as a multiset (an exact tie shows as s == b); exact
                 // whenever it is <= b * c^2
    bool bailed;  // the wave exceeded its work budget: nothing is valid, see k_query_heavy
    uint32_t h;  // among the vertices at distance exactly b, the one nanoflann's near-first DFS
                 // (for THIS lane) meets first: its tree position, or (bit 31) the ref of the
                 // packet group in which it is the only vertex at that distance; H_NONE: none
};

// Which of two tree positions does nanoflann's DFS for query q meet first?  Walk down to the
// node where they part: nanoflann takes its near child first there (nanoflann.hpp:1379-1390);
// inside one leaf the lower position comes first (1354).
__device__ __forceinline__ int kd_dfs_first(const MeshDev& M, float qx, float qy, float qz, int pa,
                                            int pb) {
    uint32_t cur = M.kd_root;
    while (!(cur & KD_LEAF)) {
        const float4 nf = M.kd_pn[4 * (size_t)cur + 3];
        const uint32_t split = M.kd_split[cur];
        const bool a1 = (uint32_t)pa < split, b1 = (uint32_t)pb < split;
        const uint32_t ndz = __float_as_uint(nf.z);
        if (a1 != b1) {
            const uint32_t feat = (ndz >> sdfk::KD_FEAT_SHIFT) & 3u;
            const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
            const bool lowside = ((val - nf.x) + (val - nf.y)) < 0;  // near child is child 1
            return (a1 == lowside) ? pa : pb;
        }
        cur = a1 ? (ndz & ~KD_FEATM) : __float_as_uint(nf.w);
    }
    return pa < pb ? pa : pb;
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// One slot of a group.  Distances are non-negative floats (or +inf / NaN), whose bit patterns
// order like unsigned integers, so the multiset insert into (b <= s2) runs on the integer pipe
// without NaN canonicalisation: new s2 = median(b, s2, d), new b = min(b, d).  Sentinel slots
// have distance +inf and change nothing.  Which slot holds the minimum is not tracked here.
#define SDFK_PACKET_SLOT(C_, O_)                                                   \
    {                                                                              \
        const float dx_ = qx - __uint_as_float(C_[O_]);                            \
        const float dy_ = qy - __uint_as_float(C_[O_ + 1]);                        \
        const float dz_ = qz - __uint_as_float(C_[O_ + 2]);                        \
        const uint32_t du_ = __float_as_uint((dx_ * dx_ + dy_ * dy_) + dz_ * dz_); \
        s2 = umed3(b, s2, du_);                                                    \
        b = min(b, du_);                                                           \
    }
#define SDFK_PACKET_CHUNK(C_) \
    SDFK_PACKET_SLOT(C_, 0) SDFK_PACKET_SLOT(C_, 4) SDFK_PACKET_SLOT(C_, 8) SDFK_PACKET_SLOT(C_, 12)

// One chunk = four padded vertices = 64 B = one s_load_dwordx16.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// Deferred children of the wave live on a stack in LDS.  Per level: the uniform child ref and,
// per lane, the upper 16 bits of that lane's box lower bound (truncation keeps it a lower
// bound), so that a pop is one LDS read and a compare against the lane's current threshold —
// no node reload, no box arithmetic.
struct PacketStack {
    uint16_t* lb;   // [level][64]
    uint32_t* ref;  // [level]
};

#ifdef SDFK_STATS
// tuning build only (make stats): per-wave work counters of the packet pass
__device__ unsigned long long g_pk_stats[8];  // waves, quad nodes, groups, chunks, tie redos, pops, pushes
#define SDFK_COUNT(K, V) if ((threadIdx.x & 63) == 0) atomicAdd(&g_pk_stats[K], (unsigned long long)(V))
#else
#define SDFK_COUNT(K, V)
#endif

__device__ __forceinline__ PacketBest kd_nearest_packet(const MeshDev& M, float qx, float qy, float qz,
                                                        const PacketStack stk) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t b = 0x7F7FFFFFu, s2 = 0x7F7FFFFFu;  // FLT_MAX bits (KNNResultSet::init)
    uint32_t h = H_NONE;
    // A lane whose minimum is an exact tie that has been settled keeps s2 = b + 1 (so that one
    // more vertex at that distance shows up again as s2 == b) and remembers the tied value here.
    uint32_t tieb = 0xFFFFFFFFu;
    const float c = M.amb_c2;  // everything within best * c^2 is evaluated
    uint32_t cur = M.kd_qroot;
    uint32_t sp = 0;
    uint32_t work = 0;  // chunks evaluated so far (wave-uniform)
    SDFK_COUNT(0, 1);
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        bool need_pop = false;
        if (!(cur & KD_LEAF)) {
            SDFK_COUNT(1, 1);
            // 4-wide node: boxes and child refs by scalar loads; slots 0,1 / 2,3 are the two
            // binary sides
            const float* Q = (const float*)(M.kd_qn + 8 * (size_t)cur);
            const uint32_t r0 = __float_as_uint(Q[24]), r1 = __float_as_uint(Q[25]),
                           r2 = __float_as_uint(Q[26]), r3 = __float_as_uint(Q[27]);
            const float thr = __uint_as_float(b) * c;
            const float lb0 = box_lb(qx, qy, qz, Q[0], Q[1], Q[2], Q[3], Q[4], Q[5]);
            const float lb1 = box_lb(qx, qy, qz, Q[6], Q[7], Q[8], Q[9], Q[10], Q[11]);
            const float lb2 = box_lb(qx, qy, qz, Q[12], Q[13], Q[14], Q[15], Q[16], Q[17]);
            const float lb3 = box_lb(qx, qy, qz, Q[18], Q[19], Q[20], Q[21], Q[22], Q[23]);
            const uint32_t act = (__ballot(lb0 <= thr) != 0ull ? 1u : 0u) |
                                 (__ballot(lb1 <= thr) != 0ull ? 2u : 0u) |
                                 (__ballot(lb2 <= thr) != 0ull ? 4u : 0u) |
                                 (__ballot(lb3 <= thr) != 0ull ? 8u : 0u);
            if (act == 0u) {
                need_pop = true;
            } else if ((act & (act - 1u)) == 0u) {
                cur = act == 1u ? r0 : (act == 2u ? r1 : (act == 4u ? r2 : r3));
            } else {
                // visiting order by box distance, majority vote per comparison (any order is
                // exact; nearer first tightens `b` sooner).  Every active child is pushed,
                // farthest first, then the nearest is taken straight back.
                const uint32_t u0 = __float_as_uint(lb0), u1 = __float_as_uint(lb1),
                               u2 = __float_as_uint(lb2), u3 = __float_as_uint(lb3);
                const bool a_first = (int)__popcll(__ballot(min(u0, u1) <= min(u2, u3))) >= 32;
                const bool s01 = (int)__popcll(__ballot(u0 <= u1)) >= 32;
                const bool s23 = (int)__popcll(__ballot(u2 <= u3)) >= 32;
                uint32_t last = 0;
#define SDFK_PK_PUSH(K)                                              \
    if (act & (1u << K)) {                                           \
        stk.lb[sp * 64u + lane] = (uint16_t)(u##K >> 16);            \
        stk.ref[sp] = r##K;                                          \
        last = r##K;                                                 \
        ++sp;                                                        \
    }
#define SDFK_PK_ORDER(A, B, C, D) { SDFK_PK_PUSH(D) SDFK_PK_PUSH(C) SDFK_PK_PUSH(B) SDFK_PK_PUSH(A) }
                if (a_first) {
                    if (s01) {
                        if (s23) SDFK_PK_ORDER(0, 1, 2, 3) else SDFK_PK_ORDER(0, 1, 3, 2)
                    } else {
                        if (s23) SDFK_PK_ORDER(1, 0, 2, 3) else SDFK_PK_ORDER(1, 0, 3, 2)
                    }
                } else {
                    if (s01) {
                        if (s23) SDFK_PK_ORDER(2, 3, 0, 1) else SDFK_PK_ORDER(3, 2, 0, 1)
                    } else {
                        if (s23) SDFK_PK_ORDER(2, 3, 1, 0) else SDFK_PK_ORDER(3, 2, 1, 0)
                    }
                }
#undef SDFK_PK_ORDER
#undef SDFK_PK_PUSH
                --sp;
                cur = last;
            }
        } else {
            // packet group: chunks of four slots, 64 B each by one scalar load
            const uint32_t nch = cur & 15u;
            const uint32_t first = ((cur & sdfk::KD_REF_MASK) >> 4) << 2;
            const float4* P = M.kd_pts4 + first;
            const uint32_t b0 = b, s0 = s2, h0 = h;
            // Tail limiter.  A wave whose queries sit near a centre of symmetry of the mesh (the
            // middle of a sphere) needs most of the tree, one dependent scalar load after the
            // other: ~0.3 us per chunk, 12 ms for 180k vertices, and the launch cannot end before
            // it does.  Past the budget the wave hands its queries to k_query_heavy (one wave
            // per query, lanes over the vertices).
            work += nch;
            SDFK_COUNT(2, 1);
            SDFK_COUNT(3, nch);
            if (nch) {
                const u32x16* C = (const u32x16*)__builtin_assume_aligned(P, 64);
#ifdef SDFK_PREFETCH
                // one chunk ahead: the load of chunk ci + 1 is in flight while chunk ci is evaluated
                u32x16 A = C[0];
                for (uint32_t ci = 0; ci < nch; ++ci) {
                    const u32x16 N = C[ci + 1 < nch ? ci + 1 : ci];
                    SDFK_PACKET_CHUNK(A)
                    A = N;
                }
#else
                for (uint32_t ci = 0; ci < nch; ++ci) {
                    const u32x16 A = C[ci];
                    SDFK_PACKET_CHUNK(A)
                }
#endif
                if (b != b0) h = cur;  // strictly improved here: the holder is in this group
                if (__ballot(s2 == b) != 0ull) {
                    // rare: some lane's two smallest distances are equal.  Redo the group for
                    // the wave in ascending position and settle each exact tie by DFS order.
                    b = b0, s2 = s0, h = h0;
                    SDFK_COUNT(4, 1);
                    for (uint32_t k = 0; k < 4u * nch; ++k) {
                        const float4 Qp = P[k];
                        const float dx = qx - Qp.x, dy = qy - Qp.y, dz = qz - Qp.z;
                        const uint32_t du = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                        const uint32_t idx = first + k;
                        if (du == b && h != H_NONE) {
                            // any position of the holder's group stands for the holder: the
                            // two part above that group
                            const uint32_t hp = (h & KD_LEAF) ? (((h & sdfk::KD_REF_MASK) >> 4) << 2) : h;
                            // (a group ref in h always names an earlier group: h0 predates this one)
                            const bool idx_first =
                                M.fast ? idx < hp : (uint32_t)kd_dfs_first(M, qx, qy, qz, (int)hp, (int)idx) == idx;
                            if (idx_first) h = idx;
                        }
                        s2 = min(s2, max(du, b));
                        if (du < b) {
                            b = du;
                            h = idx;
                        }
                    }
                    if (s2 == b) tieb = b, s2 = b + 1u;
                }
            }
            need_pop = true;
        }
        if (need_pop) {
            const uint32_t thr16 = __float_as_uint(__uint_as_float(b) * c) >> 16;
            const bool over = work > M.heavy_budget;  // give up: drop what is still on the stack
            if (over) sp = 0;
            for (;;) {
                if (sp == 0) {
                    if (b == tieb) s2 = b;
                    return PacketBest{__uint_as_float(b), __uint_as_float(s2), over, h};
                }
                --sp;
                SDFK_COUNT(5, 1);
                const uint32_t l16 = stk.lb[sp * 64u + lane];
                const uint32_t r = stk.ref[sp];
                if (__ballot(l16 <= thr16) != 0ull) {
                    cur = r;
                    break;
                }
            }
        }
    }
}

// The exact tree position behind PacketBest::h: a group ref is resolved by finding the group's
// one vertex at distance b (per lane, once per query).
__device__ __forceinline__ int packet_resolve(const MeshDev& M, float qx, float qy, float qz, float bf,
                                              uint32_t h) {
    if (h == H_NONE) return -1;
    if (!(h & KD_LEAF)) return (int)h;
    const uint32_t first = ((h & sdfk::KD_REF_MASK) >> 4) << 2, n = 4u * (h & 15u);
    const uint32_t b = __float_as_uint(bf);
    for (uint32_t k = 0; k < n; ++k) {
        const float4 P = M.kd_pts4[first + k];
        const float dx = qx - P.x, dy = qy - P.y, dz = qz - P.z;
        if (__float_as_uint((dx * dx + dy * dy) + dz * dz) == b) return (int)(first + k);
    }
    return -1;
}

// Largest mindistsq' that nanoflann tests on its way to the leaf holding tree position `pos`
// (one value per far descent on that path, nanoflann.hpp:1396-1399), computed with the same
// mindistsq / dists[] arithmetic as kd_nearest.  If every one of them is <= the smallest
// distance any OTHER vertex has, none of those tests can fail whatever was visited before
// (worst >= that distance until `pos` itself is evaluated), so `pos` is certainly visited.
__device__ __forceinline__ float kd_path_max_bound(const MeshDev& M, float qx, float qy, float qz,
                                                   int pos) {
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, mind = 0.f;
    if (qx < M.bb[0]) { d0 = (qx - M.bb[0]) * (qx - M.bb[0]); mind += d0; }
    if (qx > M.bb[3]) { d0 = (qx - M.bb[3]) * (qx - M.bb[3]); mind += d0; }
    if (qy < M.bb[1]) { d1 = (qy - M.bb[1]) * (qy - M.bb[1]); mind += d1; }
    if (qy > M.bb[4]) { d1 = (qy - M.bb[4]) * (qy - M.bb[4]); mind += d1; }
    if (qz < M.bb[2]) { d2 = (qz - M.bb[2]) * (qz - M.bb[2]); mind += d2; }
    if (qz > M.bb[5]) { d2 = (qz - M.bb[5]) * (qz - M.bb[5]); mind += d2; }
    float worst_bound = 0.f;
    uint32_t cur = M.kd_root;
    while (!(cur & KD_LEAF)) {
        const float4 nf = M.kd_pn[4 * (size_t)cur + 3];
        const uint32_t split = M.kd_split[cur];
        const float divlow = nf.x, divhigh = nf.y;
        const uint32_t ndz = __float_as_uint(nf.z);
        const uint32_t feat = (ndz >> sdfk::KD_FEAT_SHIFT) & 3u;
        const float val = feat == 0 ? qx : (feat == 1 ? qy : qz);
        const bool lowside = ((val - divlow) + (val - divhigh)) < 0;  // near child is child 1
        const bool in1 = (uint32_t)pos < split;
        if (in1 != lowside) {  // far descent
            const float pl = lowside ? divhigh : divlow;
            const float cut = (val - pl) * (val - pl);
            const float dst = feat == 0 ? d0 : (feat == 1 ? d1 : d2);
            const float m2 = (mind + cut) - dst;
            if (!(m2 <= worst_bound)) worst_bound = m2;  // NaN propagates as "too large"
            mind = m2;
            if (feat == 0) d0 = cut; else if (feat == 1) d1 = cut; else d2 = cut;
        }
        cur = in1 ? (ndz & ~KD_FEATM) : __float_as_uint(nf.w);
    }
    return worst_bound;
}

template <int KIND, bool ROBUST>
__global__ __launch_bounds__(PBLOCK) void k_query_packet(MeshDev M, const float* __restrict__ pts,
                                                         const uint32_t* __restrict__ perm, int64_t n,
                                                         int trunc_aabb, void* __restrict__ out,
                                                         unsigned long long* __restrict__ amb_mask,
                                                         uint32_t* __restrict__ amb_cnt,
                                                         uint32_t depth_cap, uint32_t off,
                                                         uint32_t* __restrict__ heavy_list,
                                                         uint32_t* __restrict__ heavy_cnt) {
    extern __shared__ uint32_t lds[];
    // per wave: depth_cap rows of 64 x u16 lower bounds, then depth_cap refs
    PacketStack stk;
    stk.lb = (uint16_t*)(lds + (threadIdx.x >> 6) * depth_cap * 33u);
    stk.ref = lds + (threadIdx.x >> 6) * depth_cap * 33u + depth_cap * 32u;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    // lanes past the end of the batch repeat its last query (and write nothing): they then ask
    // for no subtree a real lane does not
    const int64_t ic = valid ? i : n - 1;
    const int64_t src = perm ? (int64_t)perm[ic] : ic;
    const V3 p = {pts[3 * src], pts[3 * src + 1], pts[3 * src + 2]};
    float sign = 0.f;
    if (ROBUST && KIND != Q_NN) sign = raycast_packet(M, p, valid);  // sdf.cpp:110
    if (ROBUST && KIND == Q_CONTAINS) {  // sdf.cpp:163-169: no nearest-vertex search at all
        if (valid) ((uint8_t*)out)[src] = sign >= 0.0f;
        if ((threadIdx.x & 63) == 0) amb_mask[i >> 6] = 0ull, amb_cnt[i >> 6] = 0u;
        return;
    }
    const PacketBest pb = kd_nearest_packet(M, p.x, p.y, p.z, stk);
    if (pb.bailed) {  // wave-uniform: every query of this wave goes to the heavy list (absolute indices)
        const unsigned long long vm = __ballot(valid);
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == 0) {
            base = atomicAdd(heavy_cnt, (uint32_t)__popcll(vm));
            amb_mask[i >> 6] = 0ull, amb_cnt[i >> 6] = 0u;
        }
        base = __builtin_amdgcn_readfirstlane(base);
        const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        if (valid) heavy_list[base + (uint32_t)__popcll(vm & lt)] = off + (uint32_t)src;
        return;
    }
    // pb.h is nanoflann's answer iff nanoflann does not prune it before visiting it:
    //  (1) no other vertex within b * c (c - 1 ~ 4.5e-6): impossible, see k_query_packet header;
    //  (2) otherwise measure it: every bound tested on the way to pb.h is <= the distance of
    //      any other vertex (>= pb.s; == b for an exact tie, where pb.h is the first met);
    //  (3) else defer to the strict kernel (exact emulation).
    const int pos = packet_resolve(M, p.x, p.y, p.z, pb.b, pb.h);
    bool amb = false;
    if (!M.fast) {
        if (valid && (!(pb.s > pb.b * M.amb_c) || !(pb.b >= 1e-30f))) {
            amb = true;
#ifdef SDFK_STATS
            atomicAdd(&g_pk_stats[6], 1ull);
#endif
            if (pos >= 0 && pb.b < FLT_MAX) {
                const float mb = kd_path_max_bound(M, p.x, p.y, p.z, pos);
                amb = !(mb <= pb.s);
            }
        }
        if (pos < 0) amb = true;
    }
    {   // ambiguous lanes of this wave, in sorted order: compacted (stably) by k_fb_compact so
        // that the strict fallback kernel still sees Morton-coherent waves
        const unsigned long long m = __ballot(valid && amb);
        SDFK_COUNT(7, __popcll(m));
        if ((threadIdx.x & 63) == 0) {
            amb_mask[i >> 6] = m;
            amb_cnt[i >> 6] = (uint32_t)__popcll(m);
        }
    }
    if (!valid || amb) return;
    if (KIND == Q_NN) {
        ((int32_t*)out)[src] = pos >= 0 ? M.kd_vind[pos] : -1;  // (-1: fast mode, no finite distance)
    } else {
        const bool trunc = (KIND == Q_CONTAINS) ? true : (trunc_aabb != 0);
        const float v = calc_tail<ROBUST>(M, p, trunc, sign, pos);
        if (KIND == Q_CONTAINS)
            ((uint8_t*)out)[src] = v >= 0;
        else
            ((float*)out)[src] = v;
    }
}


// amb_cnt has been exclusive-scanned: wave w's ambiguous queries go to fb_list[amb_cnt[w] ...]
// The list accumulates over the chunks of one call (entries are absolute point indices):
// *base_in = entries already in the list, *total_out (a different word) = new total.
__global__ void k_fb_compact(const unsigned long long* __restrict__ amb_mask,
                             const uint32_t* __restrict__ amb_scan, const uint32_t* __restrict__ perm,
                             int64_t n, uint32_t off, uint32_t* __restrict__ fb_list,
                             const uint32_t* __restrict__ base_in, uint32_t* __restrict__ total_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t w = i >> 6;
    const int lane = (int)(i & 63);
    const unsigned long long m = amb_mask[w];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t base = *base_in + amb_scan[w];
    if ((m >> lane) & 1ull) fb_list[base + (uint32_t)__popcll(m & lt)] = off + perm[i];
    if (i == n - 1) *total_out = base + (uint32_t)__popcll(m);
}

// The queries of the waves that gave up in the packet pass: ONE WAVE PER QUERY, the 64 lanes
// stride over all (padded) tree positions — no pruning, no dependent loads, V / 64 steps.  The
// (four queries per pass share each vertex load).  The same three things come out as from the packet pass: the exact fp32 minimum b, the second
// smallest distance s as a multiset (always exact here), and h = among the vertices at distance
// exactly b the one nanoflann's DFS meets first (folded pairwise with kd_dfs_first, a total order
// for a fixed query; fast_nn: the lowest position).  Then the packet kernel's own decision:
// window test, path check, else defer to the strict kernel's list.
constexpr int HEAVY_Q = 4;  // queries per wave pass: one vertex load serves four distance evaluations
template <int KIND, bool ROBUST>
__global__ __launch_bounds__(256) void k_query_heavy(MeshDev M, const float* __restrict__ pts,
                                                     const uint32_t* __restrict__ heavy_list,
                                                     const uint32_t* __restrict__ heavy_cnt, int trunc_aabb,
                                                     void* __restrict__ out, uint32_t* __restrict__ fb_list,
                                                     uint32_t* __restrict__ fb_cnt) {
    const uint32_t cnt = *heavy_cnt;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q0 = HEAVY_Q * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6); q0 < cnt; q0 += HEAVY_Q * nw) {
        uint32_t src[HEAVY_Q], b[HEAVY_Q], s2[HEAVY_Q], h[HEAVY_Q];
        V3 p[HEAVY_Q];
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            const uint32_t q = q0 + j < cnt ? q0 + j : cnt - 1;  // the tail repeats the last query
            src[j] = heavy_list[q];
            p[j] = {pts[3 * (size_t)src[j]], pts[3 * (size_t)src[j] + 1], pts[3 * (size_t)src[j] + 2]};
            b[j] = 0x7F7FFFFFu, s2[j] = 0x7F7FFFFFu, h[j] = H_NONE;
        }
        // four positions per step, their loads issued together (the loop is latency bound
        // otherwise).  Uniform trip count and no branch in the body: a position past the end reads
        // the last slot and gets the distance 0xFFFFFFFF, which changes nothing.
        for (uint32_t k0 = 0; k0 < M.kd_npos; k0 += 256u) {
            float4 P[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + 64u * u + lane;
                P[u] = M.kd_pts4[k < M.kd_npos ? k : M.kd_npos - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + 64u * u + lane;
#pragma unroll
                for (int j = 0; j < HEAVY_Q; ++j) {
                    const float dx = p[j].x - P[u].x, dy = p[j].y - P[u].y, dz = p[j].z - P[u].z;
                    uint32_t du = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                    du = k < M.kd_npos ? du : 0xFFFFFFFFu;
                    // branch-free: on du == b the earlier (lower) position stays, s2 = b flags the tie
                    const bool lt = du < b[j];
                    s2[j] = lt ? b[j] : min(s2[j], du);
                    h[j] = lt ? k : h[j];
                    b[j] = min(b[j], du);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t ob = __shfl_xor(b[j], o), os = __shfl_xor(s2[j], o), oh = __shfl_xor(h[j], o);
                const uint32_t nh = ob < b[j] ? oh : (b[j] < ob ? h[j] : min(h[j], oh));  // H_NONE is the largest
                s2[j] = min(min(s2[j], os), max(b[j], ob));
                b[j] = min(b[j], ob);
                h[j] = nh;
            }
            // Exact tie (wave-uniform after the reduction): h is the lowest tied position, which is
            // the fast_nn rule; nanoflann's rule wants the tied vertex its DFS meets first: second
            // scan over the vertices at distance exactly b, folded with kd_dfs_first (a total
            // order for a fixed query).  Kept out of the main loop: one tree walk per tied vertex.
            if (s2[j] == b[j] && h[j] != H_NONE && !M.fast) {
                uint32_t hh = H_NONE;
                for (uint32_t k0 = 0; k0 < M.kd_npos; k0 += 256u) {
                    uint32_t du[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k = k0 + 64u * u + lane;
                        const float4 P = M.kd_pts4[k < M.kd_npos ? k : M.kd_npos - 1];
                        const float dx = p[j].x - P.x, dy = p[j].y - P.y, dz = p[j].z - P.z;
                        du[u] = k < M.kd_npos ? __float_as_uint((dx * dx + dy * dy) + dz * dz) : 0xFFFFFFFFu;
                    }
                    if (__ballot(du[0] == b[j] || du[1] == b[j] || du[2] == b[j] || du[3] == b[j]) == 0ull) continue;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t k = k0 + 64u * u + lane;
                        if (du[u] == b[j])
                            hh = hh == H_NONE ? k : (uint32_t)kd_dfs_first(M, p[j].x, p[j].y, p[j].z, (int)hh, (int)k);
                    }
                }
                for (int o = 32; o >= 1; o >>= 1) {
                    const uint32_t oh = __shfl_xor(hh, o);
                    if (hh == H_NONE)
                        hh = oh;
                    else if (oh != H_NONE)
                        hh = (uint32_t)kd_dfs_first(M, p[j].x, p[j].y, p[j].z, (int)hh, (int)oh);
                }
                h[j] = hh;
            }
        }
        // lane j finishes query j
        if (lane >= (uint32_t)HEAVY_Q || q0 + lane >= cnt) continue;
        uint32_t mb = b[0], ms = s2[0], mh = h[0], msrc = src[0];
        V3 mp = p[0];
#pragma unroll
        for (int j = 1; j < HEAVY_Q; ++j)
            if (lane == (uint32_t)j) mb = b[j], ms = s2[j], mh = h[j], msrc = src[j], mp = p[j];
        const int pos = mh == H_NONE ? -1 : (int)mh;
        const float bf = __uint_as_float(mb), sf = __uint_as_float(ms);
        bool amb = false;
        if (!M.fast) {
            if (!(sf > bf * M.amb_c) || !(bf >= 1e-30f)) {
                amb = true;
                if (pos >= 0 && bf < FLT_MAX) amb = !(kd_path_max_bound(M, mp.x, mp.y, mp.z, pos) <= sf);
            }
            if (pos < 0) amb = true;
        }
        if (amb) {
            fb_list[atomicAdd(fb_cnt, 1u)] = msrc;
        } else if (KIND == Q_NN) {
            ((int32_t*)out)[msrc] = pos >= 0 ? M.kd_vind[pos] : -1;
        } else {
            float sign = 0.f;
            if (ROBUST) sign = raycast(M, mp, nullptr);
            const bool trunc = (KIND == Q_CONTAINS) ? true : (trunc_aabb != 0);
            const float v = calc_tail<ROBUST>(M, mp, trunc, sign, pos);
            if (KIND == Q_CONTAINS)
                ((uint8_t*)out)[msrc] = v >= 0;
            else
                ((float*)out)[msrc] = v;
        }
    }
}

// =========================================================================================
// sample_surface (sdf.cpp:176-234): face by inverse-CDF over the running sum of face areas,
// then a uniform point of the parallelogram (a, ab, ac) reflected over bc.  The reference draws
// from a time-seeded mt19937 (util.cpp:32-40); here sample j takes its three uniforms from
// Philox4x32-10(key = seed, counter = j), so a call is reproducible and order independent.
// =========================================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0, c1 = l1, c2 = n2, c3 = l0;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) {  // [0, 1 - eps], sdf.cpp:182-183
    return ((float)(x >> 8) * 5.9604644775390625e-8f) * (1.0f - FLT_EPSILON);
}
__global__ void k_sample_surface(const float4* __restrict__ geo, const float* __restrict__ cum,
                                 uint32_t nf, float total_area, uint64_t seed, int64_t n,
                                 float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t x[4];
    philox4x32_10((uint32_t)j, (uint32_t)((uint64_t)j >> 32), 0u, 0u, (uint32_t)seed,
                  (uint32_t)(seed >> 32), x);
    const float r = u01(x[0]) * total_area;
    // std::lower_bound: first index whose running sum is not less than r (sdf.cpp:205-209)
    uint32_t lo = 0, cnt = nf;
    while (cnt > 0) {
        const uint32_t half = cnt >> 1;
        if (cum[lo + half] < r)
            lo += half + 1, cnt -= half + 1;
        else
            cnt = half;
    }
    const uint32_t i = lo < nf - 1 ? lo : nf - 1;
    const float4 g0 = geo[4 * (size_t)i], g1 = geo[4 * (size_t)i + 1], g2 = geo[4 * (size_t)i + 2];
    const V3 a = {g0.x, g0.y, g0.z}, b = {g0.w, g1.x, g1.y}, c = {g1.z, g1.w, g2.x};
    const V3 nrm = {g2.y, g2.z, g2.w};
    const V3 ab = sdfk::sub(b, a), ac = sdfk::sub(c, a), bc = sdfk::sub(c, b);
    V3 perp = sdfk::cross(bc, nrm);
    const float pn = __builtin_sqrtf(sdfk::sqnorm(perp));  // normalized(): v / norm()
    if (pn > 0.f) perp = {perp.x / pn, perp.y / pn, perp.z / pn};
    const bool a_dir = sdfk::dot(ab, perp) < 0.0f;
    const float u = u01(x[1]), w = u01(x[2]);
    V3 p = {(a.x + u * ab.x) + w * ac.x, (a.y + u * ab.y) + w * ac.y, (a.z + u * ab.z) + w * ac.z};
    const float bp = sdfk::dot(sdfk::sub(p, b), perp);
    if ((bp >= 0.0f) != a_dir)
        p = {p.x - (bp * perp.x) * 2.f, p.y - (bp * perp.y) * 2.f, p.z - (bp * perp.z) * 2.f};
    out[3 * j] = p.x, out[3 * j + 1] = p.y, out[3 * j + 2] = p.z;
}


// =========================================================================================
// update(): per-face precompute (sdf.cpp:50-70)
// =========================================================================================
struct RMat {
    float m[9];
};

__global__ void k_face_precompute(const float* __restrict__ verts, const uint32_t* __restrict__ faces,
                                  uint32_t nf, RMat R, int robust, float4* __restrict__ geo,
                                  float4* __restrict__ rface, float* __restrict__ fbox,
                                  float* __restrict__ normals, float* __restrict__ areas) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const uint32_t ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
    const V3 a = {verts[3 * (size_t)ia], verts[3 * (size_t)ia + 1], verts[3 * (size_t)ia + 2]};
    const V3 b = {verts[3 * (size_t)ib], verts[3 * (size_t)ib + 1], verts[3 * (size_t)ib + 2]};
    const V3 c = {verts[3 * (size_t)ic], verts[3 * (size_t)ic + 1], verts[3 * (size_t)ic + 2]};
    V3 n = sdfk::cross(sdfk::sub(b, a), sdfk::sub(c, a));  // sdf.hpp:75-81
    const float area = __builtin_sqrtf(sdfk::sqnorm(n));     // sdf.cpp:68
    n = {n.x / area, n.y / area, n.z / area};                 // sdf.cpp:69
    areas[f] = area;
    normals[3 * f] = n.x, normals[3 * f + 1] = n.y, normals[3 * f + 2] = n.z;
    geo[4 * (size_t)f] = make_float4(a.x, a.y, a.z, b.x);
    geo[4 * (size_t)f + 1] = make_float4(b.y, b.z, c.x, c.y);
    geo[4 * (size_t)f + 2] = make_float4(c.z, n.x, n.y, n.z);
    geo[4 * (size_t)f + 3] = make_float4(area, 0.f, 0.f, 0.f);
    if (robust) {
        const V3 ra = sdfk::mulR(a, R.m), rb = sdfk::mulR(b, R.m), rc = sdfk::mulR(c, R.m);
        const V3 rn = sdfk::mulR(n, R.m);  // sdf.cpp:299-300
        V3 mn = ra, mx = ra;               // sdf.cpp:54-63
        mn = {sdfk::stdmin(mn.x, rb.x), sdfk::stdmin(mn.y, rb.y), sdfk::stdmin(mn.z, rb.z)};
        mn = {sdfk::stdmin(mn.x, rc.x), sdfk::stdmin(mn.y, rc.y), sdfk::stdmin(mn.z, rc.z)};
        mx = {sdfk::stdmax(mx.x, rb.x), sdfk::stdmax(mx.y, rb.y), sdfk::stdmax(mx.z, rb.z)};
        mx = {sdfk::stdmax(mx.x, rc.x), sdfk::stdmax(mx.y, rc.y), sdfk::stdmax(mx.z, rc.z)};
        rface[3 * (size_t)f] = make_float4(ra.x, ra.y, ra.z, rb.x);
        rface[3 * (size_t)f + 1] = make_float4(rb.y, rb.z, rc.x, rc.y);
        rface[3 * (size_t)f + 2] = make_float4(rc.z, rn.x, rn.y, rn.z);
        fbox[6 * (size_t)f] = mn.x, fbox[6 * (size_t)f + 1] = mn.y, fbox[6 * (size_t)f + 2] = mn.z;
        fbox[6 * (size_t)f + 3] = mx.x, fbox[6 * (size_t)f + 4] = mx.y, fbox[6 * (size_t)f + 5] = mx.z;
    }
}

// =========================================================================================
// LDS radix sort of (u32 key, u32 value) pairs, 8 bits per pass, stable.
//   k_rs_hist   : per-tile digit histogram (LDS atomics) -> hist[digit][tile]
//   scan        : exclusive scan of hist (digit-major), hierarchical
//   k_rs_scatter: per 64-key wave chunk, ballot "match" masks give each key its stable rank
//                 among equal digits; chunk bases are scanned in LDS per digit
// =========================================================================================
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 12;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 3072
constexpr int RS_CHUNKS = RS_TILE / 64;         // 48

// element (k, t) of a tile = index base + k * 256 + t, so a wave reads 64 consecutive keys
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n,
                                                        int shift, uint32_t* __restrict__ hist,
                                                        uint32_t ntiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// block-level exclusive scan of 2048 entries per block; block totals to sums[]
__global__ __launch_bounds__(256) void k_scan_block(uint32_t* __restrict__ data, uint32_t n,
                                                    uint32_t* __restrict__ sums) {
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * 2048u + threadIdx.x * 8u;
    uint32_t v[8], t = 0;
    for (int k = 0; k < 8; ++k) {
        v[k] = (base + k < n) ? data[base + k] : 0u;
        t += v[k];
    }
    // inclusive scan of t across the block
    uint32_t x = t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    uint32_t run = woff + x - t;
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255 && sums) sums[blockIdx.x] = woff + x;
}
__global__ void k_scan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ sums) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += sums[i / 2048u];
}

// in-place exclusive scan of n u32 (n <= 2048^3); s1/s2 hold ceil(n/2048) and ceil(n/2048^2) sums
static int exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* s1, uint32_t* s2, hipStream_t s) {
    if (n == 0) return 0;
    const uint32_t n1 = (n + 2047) / 2048, n2 = (n1 + 2047) / 2048;
    k_scan_block<<<n1, 256, 0, s>>>(data, n, s1);
    if (n1 > 1) {
        k_scan_block<<<n2, 256, 0, s>>>(s1, n1, s2);
        if (n2 > 1) {
            k_scan_block<<<1, 256, 0, s>>>(s2, n2, nullptr);
            k_scan_add<<<(n1 + 255) / 256, 256, 0, s>>>(s1, n1, s2);
        }
        k_scan_add<<<(n + 255) / 256, 256, 0, s>>>(data, n, s1);
    }
    return 0;
}

// Stable scatter of one tile.  Ranks come from per-64-key-chunk ballot "match" masks; the tile
// is first permuted into digit order inside LDS so that consecutive lanes then write
// consecutive global addresses (runs of tile/256 elements per digit instead of single keys).
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n, int shift,
    const uint32_t* __restrict__ hist, uint32_t ntiles, int vals_iota) {
    __shared__ uint16_t cnt[RS_CHUNKS * 256];  // per (chunk, digit) counts -> offsets   24 KB
    __shared__ uint32_t skey[RS_TILE];         // tile in digit order                   12 KB
    __shared__ uint32_t sval[RS_TILE];         //                                       12 KB
    __shared__ uint32_t bstart[256];           // first local slot of each digit
    __shared__ uint32_t gbase[256];            // global base of each digit for this tile
    __shared__ uint32_t wsum[4];
    for (int k = threadIdx.x; k < RS_CHUNKS * 256; k += RS_THREADS) cnt[k] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * RS_TILE;
    uint32_t key[RS_ITEMS], val[RS_ITEMS];
    uint16_t rank[RS_ITEMS];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int chunk = k * 4 + w;  // ascending key index
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[k] = valid ? (vals_iota ? i : vals_in[i]) : 0u;
        const uint32_t dg = (key[k] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bl = __ballot((dg >> b) & 1u);
            m &= ((dg >> b) & 1u) ? bl : ~bl;
        }
        rank[k] = (uint16_t)__popcll(m & lt);
        if (valid && rank[k] == 0) cnt[chunk * 256 + dg] = (uint16_t)__popcll(m);
    }
    __syncthreads();
    {   // thread = digit: offsets of its chunks, then a block scan of the digit totals
        const uint32_t dg = threadIdx.x;
        uint32_t run = 0;
        for (int c = 0; c < RS_CHUNKS; ++c) {
            const uint32_t t = cnt[c * 256 + dg];
            cnt[c * 256 + dg] = (uint16_t)run;
            run += t;
        }
        uint32_t x = run;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        bstart[dg] = woff + x - run;
        gbase[dg] = hist[dg * ntiles + blockIdx.x];
    }
    __syncthreads();
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int chunk = k * 4 + w;
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) {
            const uint32_t dg = (key[k] >> shift) & 255u;
            const uint32_t slot = bstart[dg] + cnt[chunk * 256 + dg] + rank[k];
            skey[slot] = key[k];
            sval[slot] = val[k];
        }
    }
    __syncthreads();
    const uint32_t tile_n = (n - base) < (uint32_t)RS_TILE ? (n - base) : (uint32_t)RS_TILE;
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t slot = k * RS_THREADS + threadIdx.x;
        if (slot < tile_n) {
            const uint32_t kk = skey[slot];
            const uint32_t dg = (kk >> shift) & 255u;
            const uint32_t dst = gbase[dg] + (slot - bstart[dg]);
            keys_out[dst] = kk;
            vals_out[dst] = sval[slot];
        }
    }
}

struct RadixSorter {
    uint32_t* keys_alt = nullptr;
    uint32_t* vals_alt = nullptr;
    uint32_t* hist = nullptr;
    uint32_t* sums1 = nullptr;
    uint32_t* sums2 = nullptr;
    uint32_t cap = 0;

    int reserve(uint32_t n) {
        if (n <= cap) return 0;
        release();
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        const uint32_t nh = 256u * ntiles;
        const uint32_t n1 = (nh + 2047) / 2048, n2 = (n1 + 2047) / 2048;
        HIPC(hipMalloc(&keys_alt, sizeof(uint32_t) * (size_t)n));
        HIPC(hipMalloc(&vals_alt, sizeof(uint32_t) * (size_t)n));
        HIPC(hipMalloc(&hist, sizeof(uint32_t) * (size_t)nh));
        HIPC(hipMalloc(&sums1, sizeof(uint32_t) * (size_t)(n1 + 1)));
        HIPC(hipMalloc(&sums2, sizeof(uint32_t) * (size_t)(n2 + 1)));
        cap = n;
        return 0;
    }
    void release() {
        void* ps[] = {keys_alt, vals_alt, hist, sums1, sums2};
        for (void* p : ps)
            if (p) (void)hipFree(p);
        keys_alt = vals_alt = hist = sums1 = sums2 = nullptr;
        cap = 0;
    }
    // sorts (keys, vals) ascending by the low `bits` bits of keys (8 bits per pass).  The result
    // is in *keys_out / *vals_out: the caller's buffers after an even number of passes, the
    // sorter's alternates after an odd number.  iota = 1: input values are taken as 0..n-1.
    int sort(uint32_t* keys, uint32_t* vals, uint32_t n, int bits, int iota, hipStream_t s,
             uint32_t** keys_out, uint32_t** vals_out) {
        if (keys_out) *keys_out = keys;
        if (vals_out) *vals_out = vals;
        if (n == 0) return 0;
        if (int rc = reserve(n)) return rc;
        const int passes = (bits + 7) / 8;
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        const uint32_t nh = 256u * ntiles;
        const uint32_t n1 = (nh + 2047) / 2048, n2 = (n1 + 2047) / 2048;
        if (n2 > 2048) return fail(SDFHIP_E_INVALID, "radix sort: too many keys (%u)", n);
        uint32_t *ki = keys, *vi = vals, *ko = keys_alt, *vo = vals_alt;
        for (int p = 0; p < passes; ++p) {
            const int shift = 8 * p;
            k_rs_hist<<<ntiles, RS_THREADS, 0, s>>>(ki, n, shift, hist, ntiles);
            exclusive_scan_u32(hist, nh, sums1, sums2, s);
            k_rs_scatter<<<ntiles, RS_THREADS, 0, s>>>(ki, vi, ko, vo, n, shift, hist, ntiles,
                                                       (p == 0 && iota) ? 1 : 0);
            uint32_t* t = ki;
            ki = ko, ko = t;
            t = vi;
            vi = vo, vo = t;
        }
        if (keys_out) *keys_out = ki;
        if (vals_out) *vals_out = vi;
        HIPC(hipGetLastError());
        return 0;
    }
};

// =========================================================================================
// Morton codes
// =========================================================================================
__device__ __forceinline__ uint32_t expand10(uint32_t v) {
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton30(float x, float y, float z, const float* lo, const float* inv) {
    float fx = (x - lo[0]) * inv[0], fy = (y - lo[1]) * inv[1], fz = (z - lo[2]) * inv[2];
    fx = fminf(fmaxf(fx, 0.f), 1023.f);
    fy = fminf(fmaxf(fy, 0.f), 1023.f);
    fz = fminf(fmaxf(fz, 0.f), 1023.f);
    return (expand10((uint32_t)fx) << 2) | (expand10((uint32_t)fy) << 1) | expand10((uint32_t)fz);
}

struct Bounds {
    float lo[3], inv[3];
};

__device__ __forceinline__ uint32_t expand16(uint32_t v) {  // 16 bits -> every other bit
    v &= 0xFFFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// Faces are ordered by the 2-D Morton code of their centroid in the (x', y') plane of the
// ray-cast frame: the parity rays are columns along z', so faces that one column can meet —
// whatever their z' — are neighbours in this order and share low subtrees.  (The x+ tie-break
// ray walks the same tree less efficiently; it is cast only when z+ and z- disagree.)
__global__ void k_face_morton(const float4* __restrict__ geo, uint32_t nf, Bounds B, RMat R,
                              uint32_t* __restrict__ keys) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const float4 g0 = geo[4 * (size_t)f], g1 = geo[4 * (size_t)f + 1], g2 = geo[4 * (size_t)f + 2];
    const float cx = (g0.x + g0.w + g1.z) * (1.f / 3.f), cy = (g0.y + g1.x + g1.w) * (1.f / 3.f),
                cz = (g0.z + g1.y + g2.x) * (1.f / 3.f);
    const float rx = cx * R.m[0] + cy * R.m[3] + cz * R.m[6];
    const float ry = cx * R.m[1] + cy * R.m[4] + cz * R.m[7];
    float fx = (rx - B.lo[0]) * B.inv[0], fy = (ry - B.lo[1]) * B.inv[1];
    fx = fminf(fmaxf(fx, 0.f), 65535.f);
    fy = fminf(fmaxf(fy, 0.f), 65535.f);
    keys[f] = (expand16((uint32_t)fx) << 1) | expand16((uint32_t)fy);
}

// query chunk bounds: block min/max -> atomics on an order-preserving uint encoding
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
static inline float ord2f_host(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    float f;
    std::memcpy(&f, &u, sizeof(f));
    return f;
}
__global__ void k_bounds_init(uint32_t* b) {
    if (threadIdx.x < 3) b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) b[threadIdx.x] = 0u;
}
// flat float view of the [n,3] points: thread-contiguous float4 loads (coordinate = index % 3);
// base pointers from hipMalloc / torch are 16-byte aligned, the tail is read scalar
__global__ __launch_bounds__(256) void k_bounds(const float* __restrict__ pts, int64_t n, uint32_t* b) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int64_t nf = 3 * n;
    const bool aligned = (((uintptr_t)pts) & 15) == 0;
    const int64_t ngroups = aligned ? nf / 12 : 0;  // 12 floats = 4 points = 3 float4
    const float4* p4 = (const float4*)pts;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
         g += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = p4[3 * g], c = p4[3 * g + 1], d = p4[3 * g + 2];
        const float v[12] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float x = v[k];
            if (fabsf(x) <= FLT_MAX) {  // skips NaN and inf
                lo[k % 3] = fminf(lo[k % 3], x);
                hi[k % 3] = fmaxf(hi[k % 3], x);
            }
        }
    }
    for (int64_t i = 12 * ngroups + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float x = pts[i];
        const int d = (int)(i % 3);
        if (fabsf(x) <= FLT_MAX) {
            if (d == 0) lo[0] = fminf(lo[0], x), hi[0] = fmaxf(hi[0], x);
            else if (d == 1) lo[1] = fminf(lo[1], x), hi[1] = fmaxf(hi[1], x);
            else lo[2] = fminf(lo[2], x), hi[2] = fmaxf(hi[2], x);
        }
    }
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o >= 1; o >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
    }
    // one set of atomics per block (they all hit the same six words)
    __shared__ float red[4][6];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; ++d) red[w][d] = lo[d], red[w][3 + d] = hi[d];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float v = red[0][d];
        for (int k = 1; k < 4; ++k) v = d < 3 ? fminf(v, red[k][d]) : fmaxf(v, red[k][d]);
        if (d < 3)
            atomicMin(&b[d], f2ord(v));
        else
            atomicMax(&b[d], f2ord(v));
    }
}
__global__ void k_query_morton(const float* __restrict__ pts, uint32_t n, const uint32_t* __restrict__ b,
                               uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float lo[3], inv[3];
    for (int d = 0; d < 3; ++d) {
        lo[d] = ord2f(b[d]);
        const float ext = ord2f(b[3 + d]) - lo[d];
        inv[d] = ext > 0.f ? 1024.f / ext : 0.f;
    }
    // 8 bits per axis are plenty for wave coherence: 3 radix passes instead of 4
    keys[i] = morton30(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], lo, inv) >> 6;
}

// =========================================================================================
// LBVH topology (Karras 2012) over Morton-sorted faces, then a threaded pre-order layout.
// Internal node i in [0, n-2], leaf p in [0, n-1] (sorted position).
//   pre-order index of a node = 2 * lo + (#ancestors entered through their LEFT child)
//   skip = index + 2 * (#leaves below) - 1
// =========================================================================================
struct LbvhTopo {
    uint32_t* lo;       // [n-1] first leaf of internal node
    uint32_t* hi;       // [n-1] last leaf
    uint32_t* split;    // [n-1] gamma: left child covers [lo, gamma]
    int32_t* parentI;   // [n-1] parent of internal node (-1 root)
    int32_t* parentL;   // [n]   parent of leaf
    uint32_t* posI;     // [n-1] pre-order index
    uint32_t* posL;     // [n]
    uint32_t* depthI;   // [n-1] #ancestors
};

__device__ __forceinline__ int lbvh_delta(const uint32_t* __restrict__ codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = ((unsigned long long)codes[i] << 32) | (uint32_t)i;
    const unsigned long long b = ((unsigned long long)codes[j] << 32) | (uint32_t)j;
    return __clzll((long long)(a ^ b));
}

__global__ void k_lbvh_karras(const uint32_t* __restrict__ codes, int n, LbvhTopo T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (lbvh_delta(codes, n, i, i + 1) - lbvh_delta(codes, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(codes, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(codes, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvh_delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvh_delta(codes, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(codes, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    T.lo[i] = lo, T.hi[i] = hi, T.split[i] = gamma;
    if (lo == gamma)
        T.parentL[gamma] = i;
    else
        T.parentI[gamma] = i;
    if (hi == gamma + 1)
        T.parentL[gamma + 1] = i;
    else
        T.parentI[gamma + 1] = i;
    if (i == 0) T.parentI[0] = -1;
}

// one thread per node (internal first, then leaves): walk to the root
__global__ void k_lbvh_layout(int n, LbvhTopo T, const uint32_t* __restrict__ sorted_face,
                              float4* __restrict__ rnodes, uint32_t* __restrict__ maxdepth) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n - 1) return;
    const bool leaf = t >= n - 1;
    const int id = leaf ? t - (n - 1) : t;
    uint32_t mylo = leaf ? (uint32_t)id : T.lo[id];
    const uint32_t lo0 = mylo, nleaves = leaf ? 1u : (T.hi[id] - T.lo[id] + 1u);
    int par = leaf ? T.parentL[id] : T.parentI[id];
    if (n == 1) par = -1;
    uint32_t lefts = 0, depth = 0;
    while (par >= 0) {
        const uint32_t plo = T.lo[par];
        lefts += (mylo == plo) ? 1u : 0u;
        ++depth;
        mylo = plo;
        par = T.parentI[par];
    }
    const uint32_t pos = 2u * lo0 + lefts;
    const uint32_t skip = pos + 2u * nleaves - 1u;
    if (leaf) {
        T.posL[id] = pos;
        // box is filled by k_lbvh_leaf_boxes
        rnodes[2 * (size_t)pos + 1] = make_float4(0.f, 0.f, __uint_as_float(skip),
                                                  __uint_as_float(sorted_face[id]));
    } else {
        T.posI[id] = pos;
        T.depthI[id] = depth;
        rnodes[2 * (size_t)pos + 1] = make_float4(0.f, 0.f, __uint_as_float(skip), __uint_as_float(NOFACE));
        atomicMax(maxdepth, depth);
    }
}

__global__ void k_lbvh_leaf_boxes(int n, const uint32_t* __restrict__ posL,
                                  const uint32_t* __restrict__ sorted_face,
                                  const float* __restrict__ fbox, float4* __restrict__ rnodes) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t f = sorted_face[p], pos = posL[p];
    const float* b = fbox + 6 * (size_t)f;
    rnodes[2 * (size_t)pos] = make_float4(b[0], b[1], b[3], b[4]);
    float4 w = rnodes[2 * (size_t)pos + 1];
    w.x = b[2], w.y = b[5];
    rnodes[2 * (size_t)pos + 1] = w;
}

// =========================================================================================
// "fast_nn" mode: the vertex tree of the packet pass built entirely on the GPU (no host kd-tree).
// Vertices are sorted by 30-bit Morton code; 16 consecutive ones form a packet group; an implicit
// 4-ary tree of quad nodes is reduced bottom-up over the groups.  Distances and the minimum are
// exact; what differs from the reference is the winner of an exact fp32 distance tie: the lowest
// tree position (Morton order, equal codes by vertex index) instead of nanoflann's first-visited.
// =========================================================================================
__global__ void k_vtx_morton(const float* __restrict__ verts, uint32_t nv, Bounds B,
                             uint32_t* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    keys[i] = morton30(verts[3 * (size_t)i], verts[3 * (size_t)i + 1], verts[3 * (size_t)i + 2], B.lo, B.inv);
}
__global__ void k_fast_positions(const float* __restrict__ verts, const uint32_t* __restrict__ sorted_idx,
                                 uint32_t nv, uint32_t nalloc, float4* __restrict__ pts4,
                                 int32_t* __restrict__ vind) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nalloc) return;
    if (i < nv) {
        const uint32_t v = sorted_idx[i];
        pts4[i] = make_float4(verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2], 3.0e38f);
        vind[i] = (int32_t)v;
    } else {
        pts4[i] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f);
        vind[i] = -1;
    }
}
constexpr uint32_t FAST_GROUP = 16;  // vertices per packet group (4 chunks)
__device__ __forceinline__ void quad_store(float4* q, int k, const float* lo, const float* hi, uint32_t ref,
                                           float* flat) {
    (void)q;
    for (int d = 0; d < 3; ++d) flat[6 * k + d] = lo[d], flat[6 * k + 3 + d] = hi[d];
    flat[24 + k] = __uint_as_float(ref);
    flat[28 + k] = 0.f;
}
__global__ void k_fast_quads_leaf(const float4* __restrict__ pts4, uint32_t nv, uint32_t ngroups,
                                  uint32_t nq, float4* __restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float flat[32];
    for (int k = 0; k < 4; ++k) {
        const uint32_t g = 4 * q + k;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t ref = KD_LEAF;  // unused slot: empty group (see kd_host.h quad_of)
        if (g < ngroups) {
            const uint32_t first = FAST_GROUP * g;
            const uint32_t cnt = nv - first < FAST_GROUP ? nv - first : FAST_GROUP;
            for (uint32_t j = 0; j < cnt; ++j) {
                const float4 p = pts4[first + j];
                lo[0] = fminf(lo[0], p.x), lo[1] = fminf(lo[1], p.y), lo[2] = fminf(lo[2], p.z);
                hi[0] = fmaxf(hi[0], p.x), hi[1] = fmaxf(hi[1], p.y), hi[2] = fmaxf(hi[2], p.z);
            }
            ref = KD_LEAF | ((first >> 2) << 4) | ((cnt + 3u) >> 2);
        }
        quad_store(nullptr, k, lo, hi, ref, flat);
    }
    for (int j = 0; j < 8; ++j)
        out[8 * (size_t)q + j] = make_float4(flat[4 * j], flat[4 * j + 1], flat[4 * j + 2], flat[4 * j + 3]);
}
__global__ void k_fast_quads_inner(const float4* __restrict__ child, uint32_t nchild, uint32_t child_off,
                                   uint32_t nq, float4* __restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float flat[32];
    for (int k = 0; k < 4; ++k) {
        const uint32_t c = 4 * q + k;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t ref = KD_LEAF;
        if (c < nchild) {
            const float* cf = (const float*)(child + 8 * (size_t)c);
            for (int j = 0; j < 4; ++j)
                for (int d = 0; d < 3; ++d) {
                    lo[d] = fminf(lo[d], cf[6 * j + d]);
                    hi[d] = fmaxf(hi[d], cf[6 * j + 3 + d]);
                }
            ref = child_off + c;
        }
        quad_store(nullptr, k, lo, hi, ref, flat);
    }
    for (int j = 0; j < 8; ++j)
        out[8 * (size_t)q + j] = make_float4(flat[4 * j], flat[4 * j + 1], flat[4 * j + 2], flat[4 * j + 3]);
}

// column grid build: count (or fill) the cells each face's rotated AABB touches
struct GridParams {
    int g;
    float o0, o1, inv0, inv1;
};
constexpr uint32_t GRID_BIG = 256;  // faces touching more cells go to the "big" list
template <bool FILL>
__global__ void k_grid_faces(const float* __restrict__ fbox, uint32_t nf, GridParams G,
                             uint32_t* __restrict__ cells /* counts | cursors */,
                             uint32_t* __restrict__ items, uint32_t* __restrict__ big,
                             uint32_t* __restrict__ nbig, int skip_nan) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const float* b = fbox + 6 * (size_t)f;
    // Renderer: a face whose image-space box has a NaN is never inserted (renderer.cpp:52-56)
    if (skip_nan && (b[0] != b[0] || b[1] != b[1] || b[3] != b[3] || b[4] != b[4])) return;
    const int x0 = grid_cell(b[0], G.o0, G.inv0, G.g), x1 = grid_cell(b[3], G.o0, G.inv0, G.g);
    const int y0 = grid_cell(b[1], G.o1, G.inv1, G.g), y1 = grid_cell(b[4], G.o1, G.inv1, G.g);
    if (x1 < x0 || y1 < y0) return;  // NaN box: the face can never pass Overlap
    const uint32_t nc = (uint32_t)(x1 - x0 + 1) * (uint32_t)(y1 - y0 + 1);
    if (nc > GRID_BIG) {
        if (!FILL) big[atomicAdd(nbig, 1u)] = f;
        return;
    }
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const uint32_t c = (uint32_t)y * (uint32_t)G.g + (uint32_t)x;
            if (FILL)
                items[atomicAdd(&cells[c], 1u)] = f;
            else
                atomicAdd(&cells[c], 1u);
        }
}

// one launch per depth, deepest first: children are complete (previous launch / leaves)
__global__ void k_lbvh_fit(int n, LbvhTopo T, uint32_t depth, float4* __restrict__ rnodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || T.depthI[i] != depth) return;
    const uint32_t pos = T.posI[i];
    const uint32_t c1 = pos + 1, c2 = pos + 2u * (T.split[i] - T.lo[i] + 1u);
    const float4 a1 = rnodes[2 * (size_t)c1], b1 = rnodes[2 * (size_t)c1 + 1];
    const float4 a2 = rnodes[2 * (size_t)c2], b2 = rnodes[2 * (size_t)c2 + 1];
    rnodes[2 * (size_t)pos] =
        make_float4(fminf(a1.x, a2.x), fminf(a1.y, a2.y), fmaxf(a1.z, a2.z), fmaxf(a1.w, a2.w));
    float4 w = rnodes[2 * (size_t)pos + 1];
    w.x = fminf(b1.x, b2.x), w.y = fmaxf(b1.y, b2.y);
    rnodes[2 * (size_t)pos + 1] = w;
}

// =========================================================================================
// handle
// =========================================================================================
// A column grid: cell -> ids of the faces whose 2-D box touches it (CSR), plus the "big" list of
// faces spanning more than GRID_BIG cells.  Used over (x', y') of the ray-cast frame by SDF and
// over the image plane by Renderer.
struct GridBuf {
    uint32_t* start = nullptr;   // [g*g + 1]
    uint32_t* cursor = nullptr;
    uint32_t* items = nullptr;
    uint32_t* big = nullptr;
    uint32_t* nbig_dev = nullptr;
    uint32_t* s1 = nullptr;
    uint32_t* s2 = nullptr;
    size_t cells_cap = 0, items_cap = 0, big_cap = 0;
    uint32_t nbig = 0, entries = 0;
    bool coarsened = false;  // faces are large against the cells at the natural resolution
    GridParams gr = {1, 0.f, 0.f, 0.f, 0.f};
    void release() {
        void* ps[] = {start, cursor, items, big, nbig_dev, s1, s2};
        for (void* p : ps)
            if (p) (void)hipFree(p);
        start = cursor = items = big = nbig_dev = s1 = s2 = nullptr;
        cells_cap = items_cap = big_cap = 0;
    }
};

struct sdfhip_ctx {
    int device = 0;
    bool robust = true, own = false;
    uint32_t nv = 0, nf = 0;
    const float* h_verts = nullptr;
    const uint32_t* h_faces = nullptr;
    std::vector<float> owned_verts;
    std::vector<uint32_t> owned_faces;
    // host-side products
    std::vector<uint32_t> adj_off;
    std::vector<int32_t> adj_faces;
    std::vector<float> h_normals, h_areas;
    bool h_face_valid = false;
    float aabb[6] = {0, 0, 0, 0, 0, 0};
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool R_pinned = false;
    sdfk::KdHost kd;
    // device
    float* d_verts = nullptr;
    uint32_t* d_faces = nullptr;
    float4* d_kd_pn = nullptr;
    float4* d_kd_qn = nullptr;
    size_t kd_quads_cap = 0;
    uint32_t* d_kd_split = nullptr;
    std::vector<float> perm_pts4;  // host staging of kd_pts4, kept across updates
    int32_t* d_kd_vind = nullptr;
    size_t kd_nodes_cap = 0, kd_pos_cap = 0;
    float4* d_kd_pts4 = nullptr;
    float amb_c = 1.f, amb_c2 = 1.f;
    uint32_t* d_fb_list = nullptr;
    uint32_t* d_fb_count = nullptr;  // two words, ping-pong (see k_fb_compact)
    int64_t fb_cap = 0;
    uint32_t last_fb = 0;  // length of the last deferred list (read back only while profiling)
    uint32_t last_heavy = 0;  // queries of the last call's heavy waves (same)
    int64_t opt_heavy_budget = 0;  // chunks per wave before it gives up; 0 = automatic
    unsigned long long* d_amb_mask = nullptr;
    uint32_t* d_amb_cnt = nullptr;
    uint32_t* d_amb_s1 = nullptr;
    uint32_t* d_amb_s2 = nullptr;
    int64_t opt_packet = 1;
    int64_t opt_build_threads = 1;
    uint32_t* d_adj_off = nullptr;
    int32_t* d_adj_faces = nullptr;
    float4* d_geo = nullptr;
    float4* d_rface = nullptr;
    float* d_area_cum = nullptr;  // running sum of face areas (sample_surface), valid iff cum_valid
    bool cum_valid = false;
    float total_area = 0.f;
    float* d_fbox = nullptr;
    // column grid (rebuilt by every update: face boxes move)
    GridBuf grid;
    // fast_nn mode (GPU-built vertex tree)
    int64_t opt_fast_nn = 0;   // applies at the next update
    bool fast_active = false;  // what the structures on the device currently are
    uint32_t fast_qroot = KD_LEAF, fast_levels = 1, fast_quads = 0;
    uint32_t* d_vkeys = nullptr;
    uint32_t* d_vidx = nullptr;
    size_t vkeys_cap = 0;
    int64_t opt_ray_mode = 0;  // 0 automatic, 1 column grid, 2 cooperative LBVH (packet kernel)
    float* d_normals = nullptr;
    float* d_areas = nullptr;
    float4* d_rnodes = nullptr;
    uint32_t n_rnodes = 0;
    uint32_t* d_sorted_face = nullptr;
    uint32_t* d_codes = nullptr;
    uint32_t* d_maxdepth = nullptr;
    LbvhTopo topo{};
    uint32_t bvh_depth = 0;
    bool topo_built = false;
    float topo_R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // rotation the LBVH topology was built for
    RadixSorter sorter;
    // query workspace
    uint32_t* d_qkeys = nullptr;
    uint32_t* d_qperm = nullptr;
    uint32_t* d_qbounds = nullptr;
    uint32_t* d_gstack = nullptr;
    size_t gstack_words = 0;
    int64_t q_cap = 0;
    // options
    int64_t opt_sort = 1, opt_sort_min = 8192, opt_chunk = 1 << 27;
    int64_t opt_sort_min_contains = 12000000;  // robust contains(): see run_query_t
    hipStream_t stream = nullptr;  // internal stream for create/update and host entry points
    // host entry points: double-buffered device staging, copies on their own streams
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    float* d_stage_in[2] = {nullptr, nullptr};
    void* d_stage_out[2] = {nullptr, nullptr};
    int64_t stage_cap = 0;     // points per staging slot
    int64_t opt_host_chunk = 0;  // 0 = automatic
    std::mutex mu;
    std::mt19937 rng;
    float last_update_ms = 0.f, last_kd_ms = 0.f;
    // optional live profiling of the query kernels (hipEvents on the launch stream)
    int64_t opt_profile = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending, prof_free;
    double prof_ms = 0.0;
    int64_t prof_launches = 0, prof_points = 0;
};

namespace {

struct DevGuard {
    int prev = -1;
    bool ok = true;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <class T>
int dalloc(T** p, size_t count) {
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    if (count == 0) count = 1;
    HIPC(hipMalloc((void**)p, sizeof(T) * count));
    return 0;
}

void draw_rotation(sdfhip_ctx* h) {
    // sdf.cpp:43-48: unit quaternion from 4 N(0,1) draws, Eigen toRotationMatrix term order
    std::normal_distribution<float> g(0.0f, 1.0f);
    float w = g(h->rng), x = g(h->rng), y = g(h->rng), z = g(h->rng);
    float nrm = std::sqrt(w * w + x * x + y * y + z * z);
    if (!(nrm > 0.f)) w = 1.f, x = y = z = 0.f, nrm = 1.f;
    w /= nrm, x /= nrm, y /= nrm, z /= nrm;
    const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    float* R = h->R;
    R[0] = 1 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
    R[3] = txy + twz, R[4] = 1 - (txx + tzz), R[5] = tyz - twx;
    R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1 - (txx + tyy);
}

// (Re)builds `B` over n face boxes (6 floats each: min xyz, max xyz; x and y are used) whose
// x / y range is [lo, hi]: count -> scan -> fill, three small launches (+ one retry per halving
// when the faces are large against the cells).
int grid_build(GridBuf& B, const float* d_fbox, uint32_t n, const float lo[2], const float hi[2],
               int skip_nan, hipStream_t s) {
    if (n == 0) {  // no faces: one empty cell (the query kernels always look their cell up)
        if (B.cells_cap < 2) {
            if (int rc = dalloc(&B.start, 2)) return rc;
            B.cells_cap = 2;
        }
        HIPC(hipMemsetAsync(B.start, 0, 2 * sizeof(uint32_t), s));
        B.gr = {1, 0.f, 0.f, 0.f, 0.f};
        B.nbig = 0, B.entries = 0, B.coarsened = false;
        return 0;
    }
    int g = (int)std::sqrt((double)n);
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    const int g_first = g;
    bool long_lists = false;  // > 10 cells per face at the natural resolution
    if (!B.nbig_dev)
        if (int rc = dalloc(&B.nbig_dev, 1)) return rc;
    if ((size_t)n > B.big_cap) {
        if (int rc = dalloc(&B.big, n)) return rc;
        B.big_cap = n;
    }
    for (;;) {
        GridParams G;
        G.g = g;
        G.o0 = lo[0], G.o1 = lo[1];
        const float e0 = hi[0] - lo[0], e1 = hi[1] - lo[1];
        G.inv0 = (e0 > 0.f && e0 < FLT_MAX) ? (float)g / e0 : 0.f;
        G.inv1 = (e1 > 0.f && e1 < FLT_MAX) ? (float)g / e1 : 0.f;
        const size_t cells = (size_t)g * g + 1;
        if (cells > B.cells_cap || !B.cursor) {
            if (int rc = dalloc(&B.start, cells)) return rc;
            if (int rc = dalloc(&B.cursor, cells)) return rc;
            if (int rc = dalloc(&B.s1, cells / 2048 + 4)) return rc;
            if (int rc = dalloc(&B.s2, cells / 2048 / 2048 + 4)) return rc;
            B.cells_cap = cells;
        }
        HIPC(hipMemsetAsync(B.start, 0, sizeof(uint32_t) * cells, s));
        HIPC(hipMemsetAsync(B.nbig_dev, 0, sizeof(uint32_t), s));
        k_grid_faces<false><<<(n + 255) / 256, 256, 0, s>>>(d_fbox, n, G, B.start, nullptr, B.big,
                                                            B.nbig_dev, skip_nan);
        exclusive_scan_u32(B.start, (uint32_t)cells, B.s1, B.s2, s);
        uint32_t total = 0, nbig = 0;
        HIPC(hipMemcpyAsync(&total, B.start + (cells - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPC(hipMemcpyAsync(&nbig, B.nbig_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        if (g == g_first) long_lists = (size_t)total + 256 * (size_t)nbig > 10 * (size_t)n;
        // many faces much larger than a cell: a coarser grid keeps the lists (and memory) short
        if (((size_t)total > 16 * (size_t)n + 4096 || (size_t)nbig > (size_t)n / 64 + 16) && g > 1) {
            g = g / 2;
            continue;
        }
        if ((size_t)total + 1 > B.items_cap) {
            const size_t cap = (size_t)total + (size_t)total / 4 + 1024;
            if (int rc = dalloc(&B.items, cap)) return rc;
            B.items_cap = cap;
        }
        HIPC(hipMemcpyAsync(B.cursor, B.start, sizeof(uint32_t) * cells, hipMemcpyDeviceToDevice, s));
        k_grid_faces<true><<<(n + 255) / 256, 256, 0, s>>>(d_fbox, n, G, B.cursor, B.items, B.big,
                                                           B.nbig_dev, skip_nan);
        HIPC(hipGetLastError());
        B.gr = G;
        B.coarsened = long_lists;
        B.nbig = nbig;
        B.entries = total;
        if (std::getenv("SDFHIP_DEBUG"))
            std::fprintf(stderr, "[sdfhip] column grid g=%d (first %d) entries=%u big=%u faces=%u\n", g, g_first, total, nbig, n);
        return 0;
    }
}

int build_topology(sdfhip_ctx* h) {
    const uint32_t n = h->nf;
    hipStream_t s = h->stream;
    if (n == 0) {
        h->n_rnodes = 0;
        h->topo_built = true;
        return 0;
    }
    // range of x', y' over the mesh: from the rotated corners of the world AABB
    Bounds B;
    RMat R;
    std::memcpy(R.m, h->R, sizeof(R.m));
    float lo[2] = {FLT_MAX, FLT_MAX}, hi[2] = {-FLT_MAX, -FLT_MAX};
    for (int c = 0; c < 8; ++c) {
        const float x = h->aabb[(c & 1) ? 3 : 0], y = h->aabb[(c & 2) ? 4 : 1], z = h->aabb[(c & 4) ? 5 : 2];
        for (int d = 0; d < 2; ++d) {
            const float r = x * R.m[d] + y * R.m[3 + d] + z * R.m[6 + d];
            lo[d] = r < lo[d] ? r : lo[d];
            hi[d] = r > hi[d] ? r : hi[d];
        }
    }
    for (int d = 0; d < 2; ++d) {
        B.lo[d] = lo[d];
        const float ext = hi[d] - lo[d];
        B.inv[d] = ext > 0.f && ext < FLT_MAX ? 65536.f / ext : 0.f;
    }
    B.lo[2] = 0.f, B.inv[2] = 0.f;
    k_face_morton<<<(n + 255) / 256, 256, 0, s>>>(h->d_geo, n, B, R, h->d_codes);
    if (int rc = h->sorter.sort(h->d_codes, h->d_sorted_face, n, 32, 1, s, nullptr, nullptr)) return rc;
    HIPC(hipMemsetAsync(h->d_maxdepth, 0, sizeof(uint32_t), s));
    if (n > 1) k_lbvh_karras<<<(n - 1 + 255) / 256, 256, 0, s>>>(h->d_codes, (int)n, h->topo);
    k_lbvh_layout<<<(2 * n - 1 + 255) / 256, 256, 0, s>>>((int)n, h->topo, h->d_sorted_face,
                                                          h->d_rnodes, h->d_maxdepth);
    HIPC(hipMemcpyAsync(&h->bvh_depth, h->d_maxdepth, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    h->n_rnodes = 2 * n - 1;
    h->topo_built = true;
    std::memcpy(h->topo_R, h->R, sizeof(h->topo_R));
    return 0;
}

int refit(sdfhip_ctx* h) {
    const uint32_t n = h->nf;
    if (n == 0) return 0;
    hipStream_t s = h->stream;
    k_lbvh_leaf_boxes<<<(n + 255) / 256, 256, 0, s>>>((int)n, h->topo.posL, h->d_sorted_face,
                                                      h->d_fbox, h->d_rnodes);
    if (n > 1)
        for (int d = (int)h->bvh_depth; d >= 0; --d)
            k_lbvh_fit<<<(n - 1 + 255) / 256, 256, 0, s>>>((int)n, h->topo, (uint32_t)d, h->d_rnodes);
    HIPC(hipGetLastError());
    return 0;
}

// column grid over the rotated face boxes (d_fbox must be current on the stream)
int build_grid(sdfhip_ctx* h) {
    // range of x', y': rotated corners of the world AABB (covers every face box up to rounding;
    // anything outside lands in the clamped border cells, on both the face and the query side)
    float lo[2] = {FLT_MAX, FLT_MAX}, hi[2] = {-FLT_MAX, -FLT_MAX};
    for (int c = 0; c < 8; ++c) {
        const float x = h->aabb[(c & 1) ? 3 : 0], y = h->aabb[(c & 2) ? 4 : 1], z = h->aabb[(c & 4) ? 5 : 2];
        for (int d = 0; d < 2; ++d) {
            const float r = x * h->R[d] + y * h->R[3 + d] + z * h->R[6 + d];
            lo[d] = r < lo[d] ? r : lo[d];
            hi[d] = r > hi[d] ? r : hi[d];
        }
    }
    return grid_build(h->grid, h->d_fbox, h->nf, lo, hi, 0, h->stream);
}

// rotation-dependent + vertex-dependent precompute; verts must already be in d_verts
int update_device_side(sdfhip_ctx* h) {
    hipStream_t s = h->stream;
    RMat R;
    std::memcpy(R.m, h->R, sizeof(R.m));
    if (h->nf)
        k_face_precompute<<<(h->nf + 255) / 256, 256, 0, s>>>(h->d_verts, h->d_faces, h->nf, R,
                                                               h->robust ? 1 : 0, h->d_geo, h->d_rface,
                                                               h->d_fbox, h->d_normals, h->d_areas);
    HIPC(hipGetLastError());
    h->h_face_valid = false;
    h->cum_valid = false;
    if (h->robust) {
        // the face order follows the ray-cast frame: rebuilt when R changed, refit otherwise
        if (!h->topo_built || std::memcmp(h->topo_R, h->R, sizeof(h->topo_R)) != 0)
            if (int rc = build_topology(h)) return rc;
        if (int rc = refit(h)) return rc;
        if (int rc = build_grid(h)) return rc;
    }
    return 0;
}

// fast_nn: vertex tree of the packet pass built on the device (d_verts and aabb are current)
int build_fast_tree(sdfhip_ctx* h) {
    hipStream_t s = h->stream;
    const uint32_t nv = h->nv;
    const uint32_t ngroups = (nv + FAST_GROUP - 1) / FAST_GROUP;
    const size_t nalloc = (size_t)FAST_GROUP * ngroups + 16;
    // level sizes of the implicit 4-ary tree over the groups, leaves first; the root is last
    uint32_t level_n[20];
    uint32_t levels = 0, total = 0;
    for (uint32_t c = ngroups; levels == 0 || c > 1;) {
        c = (c + 3) / 4;
        if (c == 0) c = 1;
        level_n[levels++] = c;
        total += c;
        if (c == 1) break;
    }
    if (total > h->kd_quads_cap) {
        if (int rc = dalloc(&h->d_kd_qn, 8 * (size_t)total)) return rc;
        h->kd_quads_cap = total;
    }
    if (nalloc > h->kd_pos_cap) {
        if (int rc = dalloc(&h->d_kd_pts4, nalloc)) return rc;
        if (int rc = dalloc(&h->d_kd_vind, nalloc)) return rc;
        h->kd_pos_cap = nalloc;
    }
    if ((size_t)nv > h->vkeys_cap) {
        if (int rc = dalloc(&h->d_vkeys, (size_t)nv)) return rc;
        if (int rc = dalloc(&h->d_vidx, (size_t)nv)) return rc;
        h->vkeys_cap = nv;
    }
    Bounds B;
    for (int d = 0; d < 3; ++d) {
        B.lo[d] = h->aabb[d];
        const float ext = h->aabb[3 + d] - h->aabb[d];
        B.inv[d] = (ext > 0.f && ext < FLT_MAX) ? 1024.f / ext : 0.f;
    }
    uint32_t* sorted_idx = h->d_vidx;
    if (nv) {
        k_vtx_morton<<<(nv + 255) / 256, 256, 0, s>>>(h->d_verts, nv, B, h->d_vkeys);
        if (int rc = h->sorter.sort(h->d_vkeys, h->d_vidx, nv, 30, 1, s, nullptr, &sorted_idx)) return rc;
    }
    k_fast_positions<<<(uint32_t)((nalloc + 255) / 256), 256, 0, s>>>(h->d_verts, sorted_idx, nv,
                                                                      (uint32_t)nalloc, h->d_kd_pts4,
                                                                      h->d_kd_vind);
    k_fast_quads_leaf<<<(level_n[0] + 63) / 64, 64, 0, s>>>(h->d_kd_pts4, nv, ngroups, level_n[0], h->d_kd_qn);
    uint32_t child_off = 0;
    for (uint32_t l = 1; l < levels; ++l) {
        const uint32_t off = child_off + level_n[l - 1];
        k_fast_quads_inner<<<(level_n[l] + 63) / 64, 64, 0, s>>>(h->d_kd_qn + 8 * (size_t)child_off,
                                                                 level_n[l - 1], child_off, level_n[l],
                                                                 h->d_kd_qn + 8 * (size_t)off);
        child_off = off;
    }
    HIPC(hipGetLastError());
    h->fast_qroot = total - 1;
    h->fast_levels = levels;
    h->fast_quads = total;
    return 0;
}

int do_update(sdfhip_ctx* h, bool draw_R) {
    auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = h->stream;
    // mesh AABB (sdf.cpp:36-37; == the kd root box): needed by the device side before the tree
    for (int d = 0; d < 6; ++d) h->aabb[d] = 0.f;
    if (h->nv) {
        float lo[3] = {h->h_verts[0], h->h_verts[1], h->h_verts[2]}, hi[3] = {lo[0], lo[1], lo[2]};
        for (size_t i = 1; i < h->nv; ++i)
            for (int d = 0; d < 3; ++d) {
                const float v = h->h_verts[3 * i + d];
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
        for (int d = 0; d < 3; ++d) h->aabb[d] = lo[d], h->aabb[d + 3] = hi[d];
    }
    if (h->robust && draw_R && !h->R_pinned) draw_rotation(h);
    // Device side (vertex upload, per-face precompute, LBVH order / refit) does not depend on
    // the kd-tree: it runs from a helper thread while this thread builds the tree.
    if (h->nv)
        HIPC(hipMemcpyAsync(h->d_verts, h->h_verts, sizeof(float) * 3 * (size_t)h->nv,
                            hipMemcpyHostToDevice, s));
    if (h->opt_fast_nn) {
        // no host kd-tree: everything is built on the device, on one stream
        h->last_kd_ms = 0.f;
        if (int rc = update_device_side(h)) return rc;
        if (int rc = build_fast_tree(h)) return rc;
        h->amb_c = 1.f, h->amb_c2 = 1.f;  // exact minimum; nothing is deferred in this mode
        HIPC(hipStreamSynchronize(s));
        h->fast_active = true;
        h->last_update_ms =
            std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    h->fast_active = false;
    std::string dev_err;
    auto dev_side = std::async(std::launch::async, [h, &dev_err]() -> int {
        if (hipSetDevice(h->device) != hipSuccess) return SDFHIP_E_HIP;
        const int rc = update_device_side(h);
        if (rc) dev_err = g_err;  // g_err is thread local
        return rc;
    });
    // strict kd-tree on the host (sdf.cpp:32 kd_tree.rebuild())
    auto tk = std::chrono::steady_clock::now();
    h->kd.build(h->h_verts, h->nv, (int)h->opt_build_threads);
    h->last_kd_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tk).count();
    // vertices in (padded) tree order; sentinel slots are far away: their distance is +inf
    const size_t npos = h->kd.pos_vert.size();
    const size_t nalloc = npos + 16;
    auto fill = std::async(std::launch::async, [h, npos, nalloc]() {
        h->perm_pts4.resize(4 * nalloc);
        float* dst = h->perm_pts4.data();
        for (size_t i = 0; i < nalloc; ++i) {
            const int32_t vi = i < npos ? h->kd.pos_vert[i] : -1;
            if (vi < 0) {
                dst[4 * i] = dst[4 * i + 1] = dst[4 * i + 2] = dst[4 * i + 3] = 3.0e38f;
            } else {
                const float* src = h->h_verts + 3 * (size_t)vi;
                dst[4 * i] = src[0], dst[4 * i + 1] = src[1], dst[4 * i + 2] = src[2], dst[4 * i + 3] = 3.0e38f;
            }
        }
    });
    auto tq0 = std::chrono::steady_clock::now();
    h->kd.build_quads();
    fill.get();
    auto tq1 = std::chrono::steady_clock::now();
    const int rc_dev = dev_side.get();
    auto tq2 = std::chrono::steady_clock::now();
    if (rc_dev) return fail(rc_dev, "%s", dev_err.empty() ? "device-side update failed" : dev_err.c_str());
    if (h->kd.nodes.size() >= (size_t)KD_MARK) return fail(SDFHIP_E_INVALID, "kd-tree too large");
    if (h->kd.quads.size() > h->kd_quads_cap) {
        if (int rc = dalloc(&h->d_kd_qn, 8 * h->kd.quads.size())) return rc;
        h->kd_quads_cap = h->kd.quads.size();
    }
    if (h->kd.nodes.size() > h->kd_nodes_cap) {
        if (int rc = dalloc(&h->d_kd_pn, 4 * h->kd.nodes.size())) return rc;
        if (int rc = dalloc(&h->d_kd_split, h->kd.nodes.size())) return rc;
        h->kd_nodes_cap = h->kd.nodes.size();
    }
    if (nalloc > h->kd_pos_cap) {
        if (int rc = dalloc(&h->d_kd_pts4, nalloc)) return rc;
        if (int rc = dalloc(&h->d_kd_vind, nalloc)) return rc;
        h->kd_pos_cap = nalloc;
    }
    {   // ambiguity factor of the packet pass: c = (1 + w_th + 4u) rounded up, w_th = u(3.02 D + 11.2)
        const double u = 5.9604644775390625e-8;
        const double w = u * (3.02 * (double)(h->kd.depth > 0 ? h->kd.depth : 1) + 11.2) + 4.0 * u;
        float c = (float)(1.0 + w);
        if ((double)c < 1.0 + w) c = std::nextafter(c, 2.0f);
        h->amb_c = c;
        float c2 = (float)((double)c * (double)c * (1.0 + 4.0 * u));
        if ((double)c2 < (double)c * (double)c * (1.0 + 4.0 * u)) c2 = std::nextafter(c2, 2.0f);
        h->amb_c2 = c2;
    }
    if (npos)
        HIPC(hipMemcpyAsync(h->d_kd_vind, h->kd.pos_vert.data(), sizeof(int32_t) * npos,
                            hipMemcpyHostToDevice, s));
    HIPC(hipMemcpyAsync(h->d_kd_pts4, h->perm_pts4.data(), sizeof(float) * 4 * nalloc,
                        hipMemcpyHostToDevice, s));
    if (!h->kd.nodes.empty()) {
        HIPC(hipMemcpyAsync(h->d_kd_pn, h->kd.nodes.data(), sizeof(sdfk::KdNodeH) * h->kd.nodes.size(),
                            hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(h->d_kd_split, h->kd.split.data(), sizeof(uint32_t) * h->kd.split.size(),
                            hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(h->d_kd_qn, h->kd.quads.data(), sizeof(sdfk::KdQuadH) * h->kd.quads.size(),
                            hipMemcpyHostToDevice, s));
    }
    HIPC(hipStreamSynchronize(s));  // host arrays are pageable and rebuilt by the next update
    h->last_update_ms =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("SDFHIP_DEBUG")) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[sdfhip] update %.2f ms: pre %.2f kd %.2f quads+fill %.2f wait-device %.2f upload %.2f\n",
                     h->last_update_ms, ms(t0, tk), h->last_kd_ms, ms(tq0, tq1), ms(tq1, tq2),
                     ms(tq2, std::chrono::steady_clock::now()));
    }
    return 0;
}

MeshDev mesh_view(const sdfhip_ctx* h) {
    MeshDev M;
    M.kd_pn = h->d_kd_pn;
    M.kd_split = h->d_kd_split;
    M.kd_qn = h->d_kd_qn;
    M.kd_qroot = h->fast_active ? h->fast_qroot : h->kd.quad_root;
    M.fast = h->fast_active ? 1 : 0;
    M.kd_vind = h->d_kd_vind;
    M.kd_root = h->fast_active ? KD_LEAF : h->kd.root_ref;
    for (int d = 0; d < 6; ++d) M.bb[d] = h->aabb[d];
    M.kd_pts4 = h->d_kd_pts4;
    M.kd_npos = h->fast_active ? (uint32_t)(FAST_GROUP * ((h->nv + FAST_GROUP - 1) / FAST_GROUP))
                               : (uint32_t)h->kd.pos_vert.size();
    // a wave may walk 1/8 of the tree (4 positions per chunk) before it hands its queries over
    M.heavy_budget = h->opt_heavy_budget > 0 ? (uint32_t)h->opt_heavy_budget
                                             : (M.kd_npos / 32u > 4096u ? M.kd_npos / 32u : 4096u);
    M.amb_c = h->amb_c;
    M.amb_c2 = h->amb_c2;
    M.adj_off = h->d_adj_off;
    M.adj_faces = h->d_adj_faces;
    M.geo = h->d_geo;
    M.rnodes = h->d_rnodes;
    M.n_rnodes = h->robust ? h->n_rnodes : 0;
    M.rface = h->d_rface;
    M.fbox = h->d_fbox;
    M.gr_start = h->grid.start;
    M.gr_items = h->grid.items;
    M.gr_big = h->grid.big;
    M.gr_nbig = h->robust ? h->grid.nbig : 0;
    M.gr_g = h->grid.gr.g;
    // automatic: faces that touch > 10 cells each at the natural resolution are large against
    // the cells and the per-lane lists are long; the wave-cooperative LBVH walk shares them
    // among 64 lanes
    M.ray_lbvh = h->opt_ray_mode == 2 || (h->opt_ray_mode == 0 && h->grid.coarsened) ? 1 : 0;
    M.gr_o0 = h->grid.gr.o0, M.gr_o1 = h->grid.gr.o1, M.gr_inv0 = h->grid.gr.inv0, M.gr_inv1 = h->grid.gr.inv1;
    for (int d = 0; d < 9; ++d) M.R[d] = h->R[d];
    return M;
}

// strict per-lane kernel (exact nanoflann emulation).  n_dev != nullptr: the point count is read
// from device memory (fallback list of the packet kernel) and `n` only sizes the grid.
template <int KIND, bool ROBUST>
int launch_strict(sdfhip_ctx* h, const MeshDev& M, const float* d_pts, const uint32_t* perm, int64_t n,
                  int trunc, void* d_out, const uint32_t* n_dev, hipStream_t s) {
    const bool needs_kd = !(KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST));
    const uint32_t depth = (uint32_t)(h->kd.depth > 0 ? h->kd.depth : 1);
    const int64_t blocks_full = (n + QBLOCK - 1) / QBLOCK;
    if (needs_kd && depth > (uint32_t)LDS_DEPTH_MAX) {
        const int64_t blocks = blocks_full < 4096 ? blocks_full : 4096;
        const size_t words = (size_t)blocks * QBLOCK * depth * 3;
        if (words > h->gstack_words) {
            if (int rc = dalloc(&h->d_gstack, words)) return rc;
            h->gstack_words = words;
        }
        k_query<KIND, ROBUST, true><<<(uint32_t)blocks, QBLOCK, 0, s>>>(M, d_pts, perm, n, trunc, d_out,
                                                                        h->d_gstack, depth, n_dev);
    } else {
        const size_t lds = needs_kd ? (size_t)(QBLOCK / 64) * depth * 192 * sizeof(uint32_t) : 0;
        const int64_t blocks = blocks_full < 0x7FFFFFFF ? blocks_full : 0x7FFFFFFF;
        k_query<KIND, ROBUST, false><<<(uint32_t)blocks, QBLOCK, lds, s>>>(M, d_pts, perm, n, trunc,
                                                                           d_out, nullptr, depth, n_dev);
    }
    HIPC(hipGetLastError());
    return 0;
}

struct ProfScope {  // optional hipEvent pair around everything one query call launches
    sdfhip_ctx* h;
    hipStream_t s;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    int begin() {
        if (h->opt_profile && h->prof_pending.size() < 8192) {
            if (!h->prof_free.empty()) {
                ev = h->prof_free.back();
                h->prof_free.pop_back();
            } else {
                HIPC(hipEventCreate(&ev.first));
                HIPC(hipEventCreate(&ev.second));
            }
            HIPC(hipEventRecord(ev.first, s));
        }
        return 0;
    }
    int end(int64_t n) {
        if (ev.first) {
            HIPC(hipEventRecord(ev.second, s));
            h->prof_pending.push_back(ev);
            h->prof_points += n;
        }
        return 0;
    }
};

int reserve_query_ws(sdfhip_ctx* h, int64_t n, int64_t fb_cap) {
    if (n > h->q_cap) {
        if (int rc = dalloc(&h->d_qkeys, (size_t)n)) return rc;
        if (int rc = dalloc(&h->d_qperm, (size_t)n)) return rc;
        const size_t nw = ((size_t)n + 63) / 64 + 4;
        if (int rc = dalloc(&h->d_amb_mask, nw)) return rc;
        if (int rc = dalloc(&h->d_amb_cnt, nw)) return rc;
        if (int rc = dalloc(&h->d_amb_s1, nw / 2048 + 4)) return rc;
        if (int rc = dalloc(&h->d_amb_s2, nw / 2048 / 2048 + 4)) return rc;
        if (!h->d_qbounds)
            if (int rc = dalloc(&h->d_qbounds, 8)) return rc;
        h->q_cap = n;
    }
    if (fb_cap > h->fb_cap) {  // [0, cap): deferred near-tie queries; [cap, 2 cap): heavy-wave queries
        if (int rc = dalloc(&h->d_fb_list, 2 * (size_t)fb_cap)) return rc;
        h->fb_cap = fb_cap;
    }
    return 0;
}

// The whole query path on device pointers, one (KIND, ROBUST) instantiation.
//   per chunk:  [bounds -> Morton keys -> radix sort]  ->  packet kernel (nn for 64 coherent
//   queries per wave + per-lane ray parity / face distances)  ->  scan + stable compaction of
//   the deferred near-tie queries into one list for the whole call
//   at the end (or when the list could overflow): strict kernel over the deferred list.
// Small / unsorted batches go straight to the strict kernel.
template <int KIND, bool ROBUST>
int run_query_t(sdfhip_ctx* h, const float* d_pts, int64_t n, int trunc, void* d_out, hipStream_t s) {
    const MeshDev M = mesh_view(h);
    const size_t osz = KIND == Q_CONTAINS ? 1 : (KIND == Q_DETAIL ? 12 : 4);
    const int64_t chunk = h->opt_chunk > 0 ? h->opt_chunk : (1 << 24);
    // Robust contains() is the ray walk only (per-lane O(1) grid lookups): sorting buys it cache
    // locality, not wave coherence, and pays off from ~12M points on (measured crossover 11M:
    // 4M points 0.84 ms unsorted vs 1.06 ms sorted, 64M points 9.1 vs 7.3 ms).
    // (Not when the cooperative LBVH walk serves the rays: that one lives on coherent waves.)
    const int64_t sort_min = (KIND == Q_CONTAINS && ROBUST && !M.ray_lbvh &&
                              h->opt_sort_min_contains > h->opt_sort_min)
                                 ? h->opt_sort_min_contains
                                 : h->opt_sort_min;
    const bool sort_any = h->opt_sort && n >= sort_min;
    // packet stack: <= 3 deferred slots per 4-wide level (+ the transient 4th); 132 B per slot
    const bool fast = h->fast_active;  // GPU-built tree: the packet kernel is the only nn search
    const int qdepth = fast ? (int)h->fast_levels : h->kd.quad_depth;
    const uint32_t pstack = 3u * (uint32_t)(qdepth > 0 ? qdepth : 1) + 4u;
    const bool packet_any =
        KIND != Q_DETAIL && (fast || (sort_any && h->opt_packet && pstack * 132u <= 48u * 1024u));
    const int64_t fb_cap = packet_any ? (n < (1ll << 27) ? n : (1ll << 27)) : 0;
    if (sort_any || fast)
        if (int rc = reserve_query_ws(h, n < chunk ? n : chunk, fb_cap)) return rc;
    int64_t deferred_pts = 0;  // points whose near ties may sit in the deferred list
    int flip = 0;              // d_fb_count[flip] holds the current list length
    // d_fb_count: [0], [1] length of the deferred list (ping-pong), [2] length of the heavy list
    if (packet_any) HIPC(hipMemsetAsync(h->d_fb_count, 0, 3 * sizeof(uint32_t), s));
    auto flush = [&]() -> int {
        if (deferred_pts == 0) return 0;
        constexpr int HK = (KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST)) ? Q_NN : KIND;
        if (!(KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST))) {
            // queries of the waves that gave up (tail limiter): one wave per query; what is
            // still ambiguous afterwards joins the deferred list
            const int64_t hw = deferred_pts < 8192 ? deferred_pts : 8192;
            k_query_heavy<HK, ROBUST><<<(uint32_t)((hw + 3) / 4), 256, 0, s>>>(
                M, d_pts, h->d_fb_list + h->fb_cap, h->d_fb_count + 2, trunc, d_out, h->d_fb_list,
                h->d_fb_count + flip);
            HIPC(hipGetLastError());
        }
        if (!fast) {
            const int64_t grid_pts = deferred_pts < (int64_t)QBLOCK * 4096 ? deferred_pts : (int64_t)QBLOCK * 4096;
            if (int rc = launch_strict<KIND, ROBUST>(h, M, d_pts, h->d_fb_list, grid_pts, trunc, d_out,
                                                     h->d_fb_count + flip, s))
                return rc;
        }
        if (h->opt_profile) {
            HIPC(hipMemcpyAsync(&h->last_fb, h->d_fb_count + flip, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPC(hipMemcpyAsync(&h->last_heavy, h->d_fb_count + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        HIPC(hipMemsetAsync(h->d_fb_count, 0, 3 * sizeof(uint32_t), s));
        flip = 0;
        deferred_pts = 0;
        return 0;
    };
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = (n - off) < chunk ? (n - off) : chunk;
        const float* p = d_pts + 3 * off;
        void* o = (char*)d_out + osz * (size_t)off;
        const uint32_t* perm = nullptr;
        if (h->opt_sort && m >= sort_min) {
            k_bounds_init<<<1, 64, 0, s>>>(h->d_qbounds);
            const int nb = (int)((m / 4 + 255) / 256 < 1024 ? (m / 4 + 255) / 256 + 1 : 1024);
            k_bounds<<<nb, 256, 0, s>>>(p, m, h->d_qbounds);
            k_query_morton<<<(uint32_t)((m + 255) / 256), 256, 0, s>>>(p, (uint32_t)m, h->d_qbounds,
                                                                       h->d_qkeys);
            uint32_t* sorted_perm = nullptr;
            if (int rc = h->sorter.sort(h->d_qkeys, h->d_qperm, (uint32_t)m, 24, 1, s, nullptr, &sorted_perm))
                return rc;
            perm = sorted_perm;
        }
        if (packet_any && (perm || fast)) {
            if (deferred_pts + m > h->fb_cap)
                if (int rc = flush()) return rc;
            const size_t lds = (size_t)(PBLOCK / 64) * pstack * 33u * sizeof(uint32_t);
            constexpr int PK = KIND == Q_DETAIL ? Q_NN : KIND;
            ProfScope prof{h, s};  // the dominant kernel, one event pair per launch
            if (int rc = prof.begin()) return rc;
            k_query_packet<PK, ROBUST><<<(uint32_t)((m + PBLOCK - 1) / PBLOCK), PBLOCK, lds, s>>>(
                M, p, perm, m, trunc, o, h->d_amb_mask, h->d_amb_cnt, pstack, (uint32_t)off,
                h->d_fb_list + h->fb_cap, h->d_fb_count + 2);
            if (int rc = prof.end(m)) return rc;
            if (fast) {  // no near-tie deferral in this mode, only the heavy list
                HIPC(hipGetLastError());
                deferred_pts += m;
                continue;
            }
            exclusive_scan_u32(h->d_amb_cnt, (uint32_t)((m + 63) / 64), h->d_amb_s1, h->d_amb_s2, s);
            k_fb_compact<<<(uint32_t)((m + 255) / 256), 256, 0, s>>>(
                h->d_amb_mask, h->d_amb_cnt, perm, m, (uint32_t)off, h->d_fb_list, h->d_fb_count + flip,
                h->d_fb_count + (flip ^ 1));
            HIPC(hipGetLastError());
            flip ^= 1;
            deferred_pts += m;
        } else {
            ProfScope prof{h, s};
            if (int rc = prof.begin()) return rc;
            if (int rc = launch_strict<KIND, ROBUST>(h, M, p, perm, m, trunc, o, nullptr, s)) return rc;
            if (int rc = prof.end(m)) return rc;
        }
    }
    return flush();
}

int run_query(sdfhip_ctx* h, int kind, const float* d_pts, int64_t n, int trunc, void* d_out,
              hipStream_t s) {
    if (n < 0) return fail(SDFHIP_E_INVALID, "negative point count");
    if (n == 0) return 0;
    if (!d_pts || !d_out) return fail(SDFHIP_E_INVALID, "null points/output pointer");
    if (n > 0xFFFF0000ll) {  // deferred-list entries are 32-bit absolute indices
        const size_t osz = kind == Q_CONTAINS ? 1 : (kind == Q_DETAIL ? 12 : 4);
        const int64_t half = 0x80000000ll;
        if (int rc = run_query(h, kind, d_pts, half, trunc, d_out, s)) return rc;
        return run_query(h, kind, d_pts + 3 * half, n - half, trunc, (char*)d_out + osz * (size_t)half, s);
    }
    if (h->robust) {
        switch (kind) {
            case Q_NN: return run_query_t<Q_NN, true>(h, d_pts, n, trunc, d_out, s);
            case Q_CONTAINS: return run_query_t<Q_CONTAINS, true>(h, d_pts, n, trunc, d_out, s);
            case Q_SDF: return run_query_t<Q_SDF, true>(h, d_pts, n, trunc, d_out, s);
            case Q_DETAIL: return run_query_t<Q_DETAIL, true>(h, d_pts, n, trunc, d_out, s);
        }
    } else {
        switch (kind) {
            case Q_NN: return run_query_t<Q_NN, false>(h, d_pts, n, trunc, d_out, s);
            case Q_CONTAINS: return run_query_t<Q_CONTAINS, false>(h, d_pts, n, trunc, d_out, s);
            case Q_SDF: return run_query_t<Q_SDF, false>(h, d_pts, n, trunc, d_out, s);
            default: return fail(SDFHIP_E_STATE, "raycast detail needs a robust handle");
        }
    }
    return fail(SDFHIP_E_INVALID, "unknown query kind");
}

// host-pointer entry: the batch is cut into chunks that flow through two device staging slots,
//   H2D(c+1) on s_in  ||  query(c) on the handle's stream  ||  D2H(c-1) on s_out,
// so that the PCIe transfers of pageable caller memory (blocking for the host, asynchronous
// for the GPU) overlap the kernels.  Chunks stay large (>= n/4, up to 2^25 points): the packet
// pass is faster on dense Morton-sorted batches.
int ensure_host_staging(sdfhip_ctx* h, int64_t cap) {
    if (!h->s_in) HIPC(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    if (!h->s_out) HIPC(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        if (!h->ev_in[k]) HIPC(hipEventCreateWithFlags(&h->ev_in[k], hipEventDisableTiming));
        if (!h->ev_done[k]) HIPC(hipEventCreateWithFlags(&h->ev_done[k], hipEventDisableTiming));
    }
    if (cap <= h->stage_cap) return 0;
    for (int k = 0; k < 2; ++k) {
        if (h->d_stage_in[k]) (void)hipFree(h->d_stage_in[k]);
        if (h->d_stage_out[k]) (void)hipFree(h->d_stage_out[k]);
        h->d_stage_in[k] = nullptr, h->d_stage_out[k] = nullptr;
    }
    h->stage_cap = 0;
    for (int k = 0; k < 2; ++k) {
        if (hipMalloc((void**)&h->d_stage_in[k], sizeof(float) * 3 * (size_t)cap) != hipSuccess ||
            hipMalloc(&h->d_stage_out[k], 12 * (size_t)cap) != hipSuccess)
            return fail(SDFHIP_E_NOMEM, "hipMalloc failed for the host staging buffers (%lld points)",
                        (long long)cap);
    }
    h->stage_cap = cap;
    return 0;
}

int run_query_host(sdfhip_ctx* h, int kind, const float* pts, int64_t n, int trunc, void* out) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    if (n < 0) return fail(SDFHIP_E_INVALID, "negative point count");
    if (n == 0) return 0;
    if (!pts || !out) return fail(SDFHIP_E_INVALID, "null points/output pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    const size_t osz = kind == Q_CONTAINS ? 1 : (kind == Q_DETAIL ? 12 : 4);
    int64_t cap = h->opt_host_chunk;
    if (cap <= 0) {
        cap = (n + 3) / 4;
        if (cap < (1 << 21)) cap = 1 << 21;
        if (cap > (1 << 25)) cap = 1 << 25;
    }
    if (cap > n) cap = n;
    if (int rc = ensure_host_staging(h, cap)) return rc;
    const int64_t nchunks = (n + cap - 1) / cap;
    auto span = [&](int64_t c) { return (n - c * cap) < cap ? (n - c * cap) : cap; };
    int rc = 0;
    hipError_t e = hipSuccess;
    auto upload = [&](int64_t c) {
        const int k = (int)(c & 1);
        e = hipMemcpyAsync(h->d_stage_in[k], pts + 3 * c * cap, sizeof(float) * 3 * (size_t)span(c),
                           hipMemcpyHostToDevice, h->s_in);
        if (e == hipSuccess) e = hipEventRecord(h->ev_in[k], h->s_in);
    };
    auto launch = [&](int64_t c) {
        const int k = (int)(c & 1);
        e = hipStreamWaitEvent(h->stream, h->ev_in[k], 0);
        if (e != hipSuccess) return;
        rc = run_query(h, kind, h->d_stage_in[k], span(c), trunc, h->d_stage_out[k], h->stream);
        if (!rc) e = hipEventRecord(h->ev_done[k], h->stream);
    };
    // Downloads run on a helper thread: a D2H copy into pageable memory blocks its caller just
    // like an H2D copy does, and the two directions are independent.
    std::mutex mx;
    std::condition_variable cv;
    int64_t launched = 0, downloaded = 0;  // chunks whose query is enqueued / whose results are home
    bool stop = false;
    hipError_t e_dl = hipSuccess;
    auto download_loop = [&]() {
        if (hipSetDevice(h->device) != hipSuccess) {
            std::lock_guard<std::mutex> l(mx);
            e_dl = hipErrorInvalidDevice;
            cv.notify_all();
            return;
        }
        for (int64_t c = 0; c < nchunks; ++c) {
            {
                std::unique_lock<std::mutex> l(mx);
                cv.wait(l, [&] { return launched > c || stop; });
                if (launched <= c) return;
            }
            const int k = (int)(c & 1);
            hipError_t ed = hipStreamWaitEvent(h->s_out, h->ev_done[k], 0);
            if (ed == hipSuccess)
                ed = hipMemcpyAsync((char*)out + osz * (size_t)(c * cap), h->d_stage_out[k],
                                    osz * (size_t)span(c), hipMemcpyDeviceToHost, h->s_out);
            if (ed == hipSuccess) ed = hipStreamSynchronize(h->s_out);
            std::lock_guard<std::mutex> l(mx);
            if (ed != hipSuccess) e_dl = ed;
            downloaded = c + 1;  // slot k is free again
            cv.notify_all();
            if (ed != hipSuccess) return;
        }
    };
    std::thread dl(download_loop);
    for (int64_t c = 0; c < nchunks && !rc && e == hipSuccess; ++c) {
        {   // slot c & 1 was last used by chunk c - 2
            std::unique_lock<std::mutex> l(mx);
            cv.wait(l, [&] { return downloaded >= c - 1 || e_dl != hipSuccess; });
            if (e_dl != hipSuccess) break;
        }
        upload(c);  // the host blocks here while earlier chunks run and download
        if (e == hipSuccess) launch(c);
        if (!rc && e == hipSuccess) {
            std::lock_guard<std::mutex> l(mx);
            launched = c + 1;
            cv.notify_all();
        }
    }
    {
        std::lock_guard<std::mutex> l(mx);
        if (rc || e != hipSuccess) stop = true;
        cv.notify_all();
    }
    dl.join();
    if (!rc && e == hipSuccess && e_dl != hipSuccess) e = e_dl;
    if (rc || e != hipSuccess) {
        (void)hipStreamSynchronize(h->s_in);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->s_out);
        if (!rc) rc = fail(SDFHIP_E_HIP, "HIP error in query: %s", hipGetErrorString(e));
    }
    return rc;
}

int run_query_dev(sdfhip_ctx* h, int kind, const float* d_pts, int64_t n, int trunc, void* d_out,
                  void* stream) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    return run_query(h, kind, d_pts, n, trunc, d_out, (hipStream_t)stream);
}

int fetch_face_host(sdfhip_ctx* h) {
    if (h->h_face_valid) return 0;
    DevGuard g(h->device);
    h->h_normals.resize(3 * (size_t)h->nf);
    h->h_areas.resize(h->nf);
    if (h->nf) {
        HIPC(hipMemcpy(h->h_normals.data(), h->d_normals, sizeof(float) * 3 * (size_t)h->nf,
                       hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(h->h_areas.data(), h->d_areas, sizeof(float) * (size_t)h->nf, hipMemcpyDeviceToHost));
    }
    h->h_face_valid = true;
    return 0;
}

// sample_surface: running sum exactly as the reference builds it (sequential fp32,
// sdf.cpp:196-201).  Unlike the reference, which computes face_area_cum once and keeps it across
// update() (stale after a deformation), it is rebuilt after every update.
int run_sample_surface(sdfhip_ctx* h, int64_t n, uint64_t seed, float* d_out, hipStream_t s) {
    if (n < 0) return fail(SDFHIP_E_INVALID, "negative sample count");
    if (h->nf == 0) return fail(SDFHIP_E_STATE, "No faces, can't sample surface.");
    if (n == 0) return 0;
    if (!d_out) return fail(SDFHIP_E_INVALID, "null output pointer");
    if (!h->cum_valid) {
        if (int rc = fetch_face_host(h)) return rc;
        std::vector<float> cum(h->nf);
        float run = 0.f, tot = 0.f;
        for (uint32_t i = 0; i < h->nf; ++i) {
            run = i ? run + h->h_areas[i] : h->h_areas[0];
            cum[i] = run;
            tot += h->h_areas[i];
        }
        h->total_area = tot;  // == sdfhip_surface_area
        if (!h->d_area_cum)
            if (int rc = dalloc(&h->d_area_cum, h->nf)) return rc;
        HIPC(hipMemcpy(h->d_area_cum, cum.data(), sizeof(float) * h->nf, hipMemcpyHostToDevice));
        h->cum_valid = true;
    }
    k_sample_surface<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(h->d_geo, h->d_area_cum, h->nf,
                                                                 h->total_area, seed, n, d_out);
    HIPC(hipGetLastError());
    return 0;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* sdfhip_last_error(void) { return g_err.c_str(); }
const char* sdfhip_version(void) { return "sdfhip 0.1 (gfx950)"; }

int sdfhip_create(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int robust,
                  int copy, int device, sdfhip_t** out) {
    if (!out) return fail(SDFHIP_E_INVALID, "out is null");
    *out = nullptr;
    if ((nv && !verts) || (nf && !faces)) return fail(SDFHIP_E_INVALID, "null verts/faces");
    if (nv >= sdfk::KD_MAX_POINTS) return fail(SDFHIP_E_INVALID, "too many vertices (max %u)", sdfk::KD_MAX_POINTS);
    if (nf >= (1u << 30)) return fail(SDFHIP_E_INVALID, "too many faces");
    for (size_t i = 0; i < 3 * (size_t)nf; ++i)
        if (faces[i] >= nv) return fail(SDFHIP_E_INVALID, "face index %u out of range at %zu", faces[i], i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SDFHIP_E_NODEVICE, "no HIP device available (there is no CPU fallback)");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) return fail(SDFHIP_E_INVALID, "device %d out of range (%d devices)", device, ndev);
    DevGuard g(device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", device);

    sdfhip_ctx* h = new sdfhip_ctx();
    h->device = device;
    h->robust = robust != 0;
    h->own = copy != 0;
    h->nv = nv, h->nf = nf;
    {
        unsigned hw = std::thread::hardware_concurrency();
        if (const char* e = getenv("OMP_NUM_THREADS")) {
            const int v = atoi(e);
            if (v > 0 && (unsigned)v < hw) hw = (unsigned)v;
        }
        h->opt_build_threads = hw < 1 ? 1 : (hw > 16 ? 16 : hw);
    }
    if (copy) {
        h->owned_verts.assign(verts, verts + 3 * (size_t)nv);
        h->owned_faces.assign(faces, faces + 3 * (size_t)nf);
        h->h_verts = h->owned_verts.data();
        h->h_faces = h->owned_faces.data();
    } else {
        h->h_verts = verts, h->h_faces = faces;
    }
    {  // util.cpp:32-40
        std::random_device rd;
        h->rng.seed(rd() ^ (unsigned)std::chrono::high_resolution_clock::now().time_since_epoch().count());
    }
    // adjacency (sdf.cpp:22-26): CSR, ascending face id per vertex
    h->adj_off.assign((size_t)nv + 1, 0);
    for (size_t i = 0; i < 3 * (size_t)nf; ++i) h->adj_off[h->h_faces[i] + 1]++;
    for (uint32_t v = 0; v < nv; ++v) h->adj_off[v + 1] += h->adj_off[v];
    h->adj_faces.resize(3 * (size_t)nf);
    {
        std::vector<uint32_t> fill(h->adj_off.begin(), h->adj_off.end() - 1);
        for (uint32_t f = 0; f < nf; ++f)
            for (int j = 0; j < 3; ++j) h->adj_faces[fill[h->h_faces[3 * (size_t)f + j]]++] = (int32_t)f;
    }
    int rc = 0;
    auto A = [&](int r) { if (!rc) rc = r; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(SDFHIP_E_HIP, "hipStreamCreate failed");
    A(dalloc(&h->d_verts, 3 * (size_t)nv));
    A(dalloc(&h->d_faces, 3 * (size_t)nf));
    A(dalloc(&h->d_fb_count, 4));
    A(dalloc(&h->d_adj_off, (size_t)nv + 1));
    A(dalloc(&h->d_adj_faces, 3 * (size_t)nf));
    A(dalloc(&h->d_geo, 4 * (size_t)nf));
    A(dalloc(&h->d_normals, 3 * (size_t)nf));
    A(dalloc(&h->d_areas, (size_t)nf));
    A(dalloc(&h->d_rface, h->robust ? 3 * (size_t)nf : 1));
    A(dalloc(&h->d_fbox, h->robust ? 6 * (size_t)nf : 1));
    if (h->robust) {
        A(dalloc(&h->d_rnodes, 2 * (size_t)(2 * (size_t)nf + 1)));
        A(dalloc(&h->d_sorted_face, (size_t)nf));
        A(dalloc(&h->d_codes, (size_t)nf));
        A(dalloc(&h->d_maxdepth, 1));
        A(dalloc(&h->topo.lo, (size_t)nf));
        A(dalloc(&h->topo.hi, (size_t)nf));
        A(dalloc(&h->topo.split, (size_t)nf));
        A(dalloc(&h->topo.parentI, (size_t)nf));
        A(dalloc(&h->topo.parentL, (size_t)nf));
        A(dalloc(&h->topo.posI, (size_t)nf));
        A(dalloc(&h->topo.posL, (size_t)nf));
        A(dalloc(&h->topo.depthI, (size_t)nf));
    }
    if (!rc && nf) {
        if (hipMemcpy(h->d_faces, h->h_faces, sizeof(uint32_t) * 3 * (size_t)nf, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_adj_faces, h->adj_faces.data(), sizeof(int32_t) * 3 * (size_t)nf, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(SDFHIP_E_HIP, "upload of faces failed");
    }
    if (!rc && hipMemcpy(h->d_adj_off, h->adj_off.data(), sizeof(uint32_t) * ((size_t)nv + 1), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(SDFHIP_E_HIP, "upload of adjacency failed");
    if (!rc) rc = do_update(h, true);  // sdf.cpp:28 update(false): same work, kd built once
    if (rc) {
        std::string keep = g_err;
        sdfhip_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return 0;
}

void sdfhip_destroy(sdfhip_t* h) {
    if (!h) return;
    DevGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_verts, h->d_faces, h->d_kd_pn, h->d_kd_split, h->d_kd_qn, h->d_kd_vind, h->d_adj_off,
                    h->d_adj_faces, h->d_geo, h->d_rface, h->d_area_cum, h->d_fbox, h->d_normals, h->d_areas,
                    h->d_rnodes, h->d_sorted_face, h->d_codes, h->d_maxdepth, h->topo.lo, h->topo.hi,
                    h->topo.split, h->topo.parentI, h->topo.parentL, h->topo.posI, h->topo.posL,
                    h->topo.depthI, h->d_qkeys, h->d_qperm, h->d_qbounds, h->d_gstack, h->d_kd_pts4,
                    h->d_fb_list, h->d_fb_count, h->d_amb_mask, h->d_amb_cnt, h->d_amb_s1, h->d_amb_s2,
                    h->d_vkeys, h->d_vidx};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->sorter.release();
    h->grid.release();
    for (int k = 0; k < 2; ++k) {
        if (h->d_stage_in[k]) (void)hipFree(h->d_stage_in[k]);
        if (h->d_stage_out[k]) (void)hipFree(h->d_stage_out[k]);
        if (h->ev_in[k]) (void)hipEventDestroy(h->ev_in[k]);
        if (h->ev_done[k]) (void)hipEventDestroy(h->ev_done[k]);
    }
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    for (auto& ev : h->prof_pending) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
    for (auto& ev : h->prof_free) (void)hipEventDestroy(ev.first), (void)hipEventDestroy(ev.second);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int sdfhip_update(sdfhip_t* h) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    return do_update(h, true);
}

int sdfhip_update_from_device(sdfhip_t* h, const float* d_verts, void* stream) {
    if (!h || !d_verts) return fail(SDFHIP_E_INVALID, "null handle/verts");
    if (!h->own)
        return fail(SDFHIP_E_STATE, "update_from_device needs an owning handle (copy=1): the host copy is refreshed");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    HIPC(hipMemcpyAsync(h->owned_verts.data(), d_verts, sizeof(float) * 3 * (size_t)h->nv,
                        hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPC(hipStreamSynchronize((hipStream_t)stream));
    return do_update(h, true);
}

int sdfhip_set_raycast_axes(sdfhip_t* h, const float R[9], int pin) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!pin) {
        h->R_pinned = false;
        return 0;
    }
    if (!R) return fail(SDFHIP_E_INVALID, "null R");
    std::memcpy(h->R, R, 9 * sizeof(float));
    h->R_pinned = true;
    if (!h->robust) return 0;
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    if (int rc = update_device_side(h)) return rc;
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}
int sdfhip_get_raycast_axes(const sdfhip_t* h, float R[9]) {
    if (!h || !R) return fail(SDFHIP_E_INVALID, "null handle/R");
    std::memcpy(R, h->R, 9 * sizeof(float));
    return 0;
}

int sdfhip_sdf(sdfhip_t* h, const float* p, int64_t n, int trunc, float* out) {
    return run_query_host(h, Q_SDF, p, n, trunc, out);
}
int sdfhip_sdf_device(sdfhip_t* h, const float* p, int64_t n, int trunc, float* out, void* s) {
    return run_query_dev(h, Q_SDF, p, n, trunc, out, s);
}
int sdfhip_contains(sdfhip_t* h, const float* p, int64_t n, uint8_t* out) {
    return run_query_host(h, Q_CONTAINS, p, n, 0, out);
}
int sdfhip_contains_device(sdfhip_t* h, const float* p, int64_t n, uint8_t* out, void* s) {
    return run_query_dev(h, Q_CONTAINS, p, n, 0, out, s);
}
int sdfhip_nn(sdfhip_t* h, const float* p, int64_t n, int32_t* out) {
    return run_query_host(h, Q_NN, p, n, 0, out);
}
int sdfhip_nn_device(sdfhip_t* h, const float* p, int64_t n, int32_t* out, void* s) {
    return run_query_dev(h, Q_NN, p, n, 0, out, s);
}
int sdfhip_raycast_detail(sdfhip_t* h, const float* p, int64_t n, int32_t* out3) {
    if (h && !h->robust) return fail(SDFHIP_E_STATE, "raycast detail needs a robust handle");
    return run_query_host(h, Q_DETAIL, p, n, 0, out3);
}

int sdfhip_robust(const sdfhip_t* h) { return h ? h->robust : 0; }
int sdfhip_own_data(const sdfhip_t* h) { return h ? h->own : 0; }
int sdfhip_device(const sdfhip_t* h) { return h ? h->device : -1; }
uint32_t sdfhip_num_verts(const sdfhip_t* h) { return h ? h->nv : 0; }
uint32_t sdfhip_num_faces(const sdfhip_t* h) { return h ? h->nf : 0; }
const float* sdfhip_verts(const sdfhip_t* h) { return h ? h->h_verts : nullptr; }
const uint32_t* sdfhip_faces(const sdfhip_t* h) { return h ? h->h_faces : nullptr; }
float* sdfhip_verts_mutable(sdfhip_t* h) {
    if (!h) return nullptr;
    if (!h->own) {
        fail(SDFHIP_E_STATE, "'verts' is non mutable, construct with copy=True");
        return nullptr;
    }
    return h->owned_verts.data();
}
uint32_t* sdfhip_faces_mutable(sdfhip_t* h) {
    if (!h) return nullptr;
    if (!h->own) {
        fail(SDFHIP_E_STATE, "'faces' is non mutable, construct with copy=True");
        return nullptr;
    }
    return h->owned_faces.data();
}
int sdfhip_surface_area(sdfhip_t* h, float* out) {
    if (!h || !out) return fail(SDFHIP_E_INVALID, "null handle/out");
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc = fetch_face_host(h)) return rc;
    float s = 0.f;
    for (float a : h->h_areas) s += a;
    *out = s;
    return 0;
}
int sdfhip_face_areas(sdfhip_t* h, float* out) {
    if (!h || !out) return fail(SDFHIP_E_INVALID, "null handle/out");
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc = fetch_face_host(h)) return rc;
    std::memcpy(out, h->h_areas.data(), sizeof(float) * h->h_areas.size());
    return 0;
}
int sdfhip_face_normals(sdfhip_t* h, float* out) {
    if (!h || !out) return fail(SDFHIP_E_INVALID, "null handle/out");
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc = fetch_face_host(h)) return rc;
    std::memcpy(out, h->h_normals.data(), sizeof(float) * h->h_normals.size());
    return 0;
}
int sdfhip_aabb(const sdfhip_t* h, float out[6]) {
    if (!h || !out) return fail(SDFHIP_E_INVALID, "null handle/out");
    std::memcpy(out, h->aabb, 6 * sizeof(float));
    return 0;
}
int sdfhip_adj_faces(const sdfhip_t* h, uint32_t v, const int32_t** ids) {
    if (!h || !ids) return fail(SDFHIP_E_INVALID, "null handle/ids");
    if (v >= h->nv) return fail(SDFHIP_E_INVALID, "vertex %u out of range", v);
    *ids = h->adj_faces.data() + h->adj_off[v];
    return (int)(h->adj_off[v + 1] - h->adj_off[v]);
}
int sdfhip_get_stats(const sdfhip_t* h, sdfhip_stats_t* o) {
    if (!h || !o) return fail(SDFHIP_E_INVALID, "null handle/out");
    o->kd_nodes = h->fast_active ? (int32_t)h->fast_quads : (int32_t)h->kd.nodes.size();
    o->kd_depth = h->fast_active ? (int32_t)h->fast_levels : h->kd.depth;
    o->bvh_nodes = h->robust ? (int32_t)h->n_rnodes : 0;
    o->bvh_depth = (int32_t)h->bvh_depth;
    o->last_update_ms = h->last_update_ms;
    o->last_kd_build_ms = h->last_kd_ms;
    sdfhip_ctx* hm = const_cast<sdfhip_ctx*>(h);
    {
        std::lock_guard<std::mutex> lk(hm->mu);
        DevGuard g(h->device);
        for (auto& ev : hm->prof_pending) {
            float ms = 0.f;
            if (hipEventSynchronize(ev.second) == hipSuccess &&
                hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                hm->prof_ms += ms;
                hm->prof_launches++;
            }
            hm->prof_free.push_back(ev);
        }
        hm->prof_pending.clear();
    }
    o->query_kernel_ms = h->prof_ms;
    o->query_kernel_launches = h->prof_launches;
    o->query_kernel_points = h->prof_points;
    o->last_fallback_points = 0;
    o->last_heavy_points = 0;
    if (h->d_fb_count) {
        DevGuard g(h->device);
        o->last_fallback_points = h->last_fb;
        o->last_heavy_points = h->last_heavy;
    }
    return 0;
}
int sdfhip_sample_surface_device(sdfhip_t* h, int64_t n, uint64_t seed, float* d_out, void* stream) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    return run_sample_surface(h, n, seed, d_out, (hipStream_t)stream);
}
int sdfhip_sample_surface(sdfhip_t* h, int64_t n, uint64_t seed, float* out) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    if (n > 0 && !out) return fail(SDFHIP_E_INVALID, "null output pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    if (n <= 0) return run_sample_surface(h, n, seed, nullptr, h->stream);
    float* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(float) * 3 * (size_t)n) != hipSuccess)
        return fail(SDFHIP_E_NOMEM, "hipMalloc failed for %lld samples", (long long)n);
    int rc = run_sample_surface(h, n, seed, d, h->stream);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, d, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(SDFHIP_E_HIP, "HIP error in sample_surface: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}
#ifdef SDFK_STATS
int sdfhip_debug_packet_stats(unsigned long long out[8], int reset) {
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_stats), 8 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPC(hipMemcpyToSymbol(HIP_SYMBOL(g_pk_stats), z, sizeof(z)));
    }
    return 0;
}
#endif
int sdfhip_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(SDFHIP_E_INVALID, "out is null");
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(SDFHIP_E_NOMEM, "hipHostMalloc of %zu bytes failed", bytes);
    }
    return 0;
}
int sdfhip_host_free(void* p) {
    if (!p) return 0;
    if (hipHostFree(p) != hipSuccess) return fail(SDFHIP_E_HIP, "hipHostFree failed");
    return 0;
}
int sdfhip_set_option(sdfhip_t* h, const char* key, int64_t v) {
    if (!h || !key) return fail(SDFHIP_E_INVALID, "null handle/key");
    std::lock_guard<std::mutex> lk(h->mu);
    if (!strcmp(key, "sort_queries"))
        h->opt_sort = v;
    else if (!strcmp(key, "profile")) {
        h->opt_profile = v;
        if (v == 2) h->prof_ms = 0.0, h->prof_launches = 0, h->prof_points = 0;  // reset
    }
    else if (!strcmp(key, "build_threads"))
        h->opt_build_threads = v < 1 ? 1 : v;
    else if (!strcmp(key, "packet"))
        h->opt_packet = v;
    else if (!strcmp(key, "heavy_budget")) {
        if (v < 0 || v > 0x7FFFFFFF) return fail(SDFHIP_E_INVALID, "heavy_budget must be >= 0 (0 = automatic)");
        h->opt_heavy_budget = v;
    } else if (!strcmp(key, "fast_nn"))
        h->opt_fast_nn = v ? 1 : 0;  // takes effect at the next sdfhip_update
    else if (!strcmp(key, "ray_mode")) {
        if (v < 0 || v > 2) return fail(SDFHIP_E_INVALID, "ray_mode must be 0 (auto), 1 (grid) or 2 (LBVH)");
        h->opt_ray_mode = v;
    } else if (!strcmp(key, "host_chunk")) {
        if (v != 0 && (v < 1024 || v > (1ll << 27)))
            return fail(SDFHIP_E_INVALID, "host_chunk must be 0 (automatic) or in [1024, 2^27]");
        h->opt_host_chunk = v;
    } else if (!strcmp(key, "packet_group")) {
        // vertices per packet group; takes effect at the next sdfhip_update
        if (v < 10 || v > (int64_t)sdfk::KD_GROUP_LIMIT)
            return fail(SDFHIP_E_INVALID, "packet_group must be in [10, %u]", sdfk::KD_GROUP_LIMIT);
        h->kd.group_max = (uint32_t)v;
    } else if (!strcmp(key, "sort_min"))
        h->opt_sort_min = v;
    else if (!strcmp(key, "sort_min_contains"))
        h->opt_sort_min_contains = v;
    else if (!strcmp(key, "chunk")) {
        if (v < 1024 || v > (1ll << 27)) return fail(SDFHIP_E_INVALID, "chunk must be in [1024, 2^27]");
        h->opt_chunk = v;
    } else
        return fail(SDFHIP_E_INVALID, "unknown option '%s'", key);
    return 0;
}

}  // extern "C"

#include "renderer.h"
