// kernels_query.h — device-side views of the mesh structures, the strict (nanoflann-exact) nearest-vertex
// search, the ray-parity walk and the one-lane-per-query kernel.  Part of sdfhip.hip (one translation unit).
#pragma once

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
    // wide packet walk: oriented boxes of the packet groups and of the runs of 64 groups (L1 nodes),
    // 64 B each, stored [node][4][64] float4 so that lane k reads child k with unit stride
    const float4* kd_gobb;   // [kd_nl1][4][64]
    const float4* kd_l1obb;  // [ceil(kd_nl1 / 64)][4][64]
    uint32_t kd_ngroups, kd_nl1;
    const float4* kd_gobb_aos;  // the group boxes once more, 64 B per group in group order (gathered through the cube lists)
    // cube lists of the running launch (k_cube_lists; cube_max == 0: none): the query domain is cut into cubes
    // (at most cube_max) and each cube lists the packet groups any query inside it can need
    uint32_t cube_max;
    const float* cube_par;      // {lo.xyz, cubes per unit length .xyz, cubes per axis .xyz}: a query's cube
    const float4* cube_hdr;     // [cube][3]: {centre.xyz, entries}, {half extents.xyz, -}, {trend direction g.xyz, -}
    const uint32_t* cube_list;  // [cube][CUBE_CAP] group indices; entry 0 is the group every wave evaluates first
    float obb_cm[3], obb_rm;  // centre of the mesh AABB, sum of its extents (rounding slack of the box tests)
    int fast;  // "fast_nn": GPU-built tree, exact distances, ties go to the lowest tree position
    const int32_t* kd_vind;
    uint32_t kd_root;
    float bb[6];  // mesh AABB == kd root bbox
    // packet traversal extras: vertices padded to 16 B, ambiguity factors c = 1 + W and ~c^2
    const float4* kd_pts4;
    const float4* kd_lbox;  // per tree position, 32 B: the tight box of its kd leaf {lo.xyz, hi.x}, {hi.yz, 0, 0} (kd_path_clear)
    uint32_t kd_npos;       // padded tree positions (sentinel slots included)
    uint32_t heavy_budget;  // packet pass: a wave that has evaluated more chunks than this gives up
    uint32_t coop_budget;   // ... and hands its packet to the cooperative kernel past this many (approximate walk)
    float amb_c, amb_c2;
    // approximate first pass, group-centred (k_apx_fill): per tree position -2 fl(v - cg), |fl(v - cg)|^2 (+inf for
    // sentinels); per 4 positions the centre cg of the group they belong to; the error model E(est) = apx_k1 est + apx_e0
    const float4* kd_apx4;
    const float4* kd_actr;
    float apx_k1, apx_e0;
    // far packets (kernels_packet.h, k_query_packet_far): queries farther than sqrt(far_r2) from far_c (the centre of
    // the mesh AABB) are "far"; far_r2 == 0: no far kernel behind this launch.  far_lo: first packet of the launch
    // the far kernel covers (it takes the packets from the end of the sorted order).
    float far_c[3], far_r2;
    uint32_t far_lo;
    const uint32_t* far_cnt;  // far queries of the launch (they hold the last sorted positions), counted by the key kernel
    // adjacency + face geometry
    const uint32_t* adj_off;
    const int32_t* adj_faces;
    const float4* geo;
    const float4* geo_adj;  // geo records in vertex-major order: entry k of adj_faces holds the record of face adj_faces[k]
    // ray structure
    const float4* rnodes;
    uint32_t n_rnodes;
    const float4* rface;
    // column grid over (x', y') of the ray-cast frame: cell -> faces whose rotated AABB touches it
    const float* fbox;          // per face: rotated AABB min xyz, max xyz
    const uint32_t* gr_start;   // [g*g + 1]
    const uint32_t* gr_items;   // face ids, cell by cell
    const float4* gr_ent;       // per entry of gr_items, 32 B: {a'.x, a'.y, b'.x, b'.y}, {c'.x, c'.y, face id, 0}
    const uint32_t* gr_big;     // faces spanning too many cells: tested by every query
    uint32_t gr_nbig;
    int ray_lbvh;  // packet kernel: walk the LBVH cooperatively instead of the grid (long lists)
    int gr_g;
    float gr_o0, gr_o1, gr_inv0, gr_inv1;
    // the same over (y', z') for the x+ tie-break ray (gx_start == nullptr: walk the LBVH instead)
    const uint32_t* gx_start;
    const float4* gx_ent;       // {a'.y, a'.z, b'.y, b'.z}, {c'.y, c'.z, face id, 0}
    const uint32_t* gx_big;
    uint32_t gx_nbig;
    int gx_g;
    float gx_o0, gx_o1, gx_inv0, gx_inv1;
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
#ifdef SDFK_STATS
// queries in the box, faces tested, faces passing Overlap in xy, records fetched; x+ rays, their nodes, the longest
// x+ walk (nodes), the longest cell list walked
__device__ unsigned long long g_ray_stats[8];
#define SDFK_RMAX(K, V) atomicMax(&g_ray_stats[K], (unsigned long long)(V))
#define SDFK_RADD(K, V) atomicAdd(&g_ray_stats[K], (unsigned long long)(V))
#define SDFK_RCOUNT(K) atomicAdd(&g_ray_stats[K], 1ull)
#else
#define SDFK_RCOUNT(K)
#define SDFK_RMAX(K, V)
#define SDFK_RADD(K, V)
#endif
__device__ __forceinline__ int ray_x_pos_lbvh(const MeshDev& M, V3 q) {
    int c = 0;
    uint32_t i = 0;
    const uint32_t n = M.n_rnodes;
#ifdef SDFK_STATS
    uint32_t visited = 0;
#endif
    while (i < n) {
#ifdef SDFK_STATS
        ++visited;
#endif
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
#ifdef SDFK_STATS
    SDFK_RCOUNT(4);
    SDFK_RADD(5, visited);
    SDFK_RMAX(6, visited);
#endif
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
    SDFK_RCOUNT(1);
    if ((q.x > mxx || mnx > q.x) || (q.y > mxy || mny > q.y)) return;  // Overlap, x and y
    SDFK_RCOUNT(2);
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
        SDFK_RCOUNT(3);
        if (sdfk::bary2d_inside(q.x, q.y, f0.x, f0.y, f0.w, f1.x, f1.z, f1.w)) {
            if (pos)
                zm ^= 1;
            else
                zp ^= 1;
        }
    }
}
// What is left of ray_face_z for a face whose Overlap in x, y and bary2d hold: the z side of Overlap and the
// facing rule decide which of the two rays (if any) it counts for.  The z range of the face's box is
// recomputed from its rotated corners exactly as k_face_precompute stored it.
__device__ __forceinline__ void ray_face_z_side(V3 q, const float4 f0, const float4 f1, const float4 f2, int& zp,
                                                int& zm) {
    SDFK_RCOUNT(3);
    const float mnz = sdfk::stdmin(sdfk::stdmin(f0.z, f1.y), f2.x), mxz = sdfk::stdmax(sdfk::stdmax(f0.z, f1.y), f2.x);
    const bool okp = !(q.z > mxz || mnz > FLT_MAX);   // box [q.z, FLT_MAX]
    const bool okm = !(-FLT_MAX > mxz || mnz > q.z);  // box [-FLT_MAX, q.z]
    const V3 nr = {f2.y, f2.z, f2.w};
    const V3 dv = sdfk::sub(q, V3{f0.x, f0.y, f0.z});
    const float s = sdfk::dot(nr, dv) * nr.z;  // sdf.cpp:301-302, ax_idx = 2
    const bool pos = s > 0.f;                  // z- needs pos, z+ needs !pos
    if (pos ? okm : okp) {
        if (pos)
            zm ^= 1;
        else
            zp ^= 1;
    }
}
// (a NaN operand is dropped, as by fminf / fmaxf)
__device__ __forceinline__ float fmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Overlap in the grid's two axes and bary2d of one list entry (the face's corners in those axes; (qx, qy):
// the query in them).  The box is rebuilt from the
// corners with v_min3 / v_max3: equal to the std::min / std::max chain of k_face_precompute for corners
// without a NaN (up to the sign of a zero, which no comparison sees); with a NaN corner the two may differ,
// and then bary2d is false (its denominator is NaN), which is all that a failed Overlap would have meant.
__device__ __forceinline__ bool ray_entry_inside(float qx, float qy, const float4 e0, const float4 e1) {
    const float mnx = fmin3(e0.x, e0.z, e1.x), mxx = fmax3(e0.x, e0.z, e1.x);
    const float mny = fmin3(e0.y, e0.w, e1.y), mxy = fmax3(e0.y, e0.w, e1.y);
    if ((qx > mxx || mnx > qx) || (qy > mxy || mny > qy)) return false;  // Overlap in the two pinned axes
    SDFK_RCOUNT(2);
    return sdfk::bary2d_inside(qx, qy, e0.x, e0.y, e0.z, e0.w, e1.x, e1.y);
}
// The cell's list carries each face's corners in x', y' next to its id (cell-major, read as a stream, 32 B per
// entry): Overlap in x, y and bary2d need nothing else, and only the faces the column really pierces (2-6 of
// the 50-70 listed on a rough surface) fetch their 48-byte record for the z side of Overlap and the facing
// rule.  The order of the tests is free: a face counts iff all of them hold (sdf.cpp:296-312).  Two entries
// per step, their loads in flight together.
__device__ __forceinline__ void ray_z_pair_grid(const MeshDev& M, V3 q, int& zp, int& zm) {
    zp = 0;
    zm = 0;
    SDFK_RCOUNT(0);
    const int cx = grid_cell(q.x, M.gr_o0, M.gr_inv0, M.gr_g), cy = grid_cell(q.y, M.gr_o1, M.gr_inv1, M.gr_g);
    const uint32_t c = (uint32_t)cy * (uint32_t)M.gr_g + (uint32_t)cx;
    const uint32_t e = M.gr_start[c + 1];
    SDFK_RMAX(7, e - M.gr_start[c]);
    for (uint32_t k = M.gr_start[c]; k < e; k += 2u) {
        const uint32_t k1 = k + 1u < e ? k + 1u : k;
        const float4 a0 = M.gr_ent[2 * (size_t)k], a1 = M.gr_ent[2 * (size_t)k + 1];
        const float4 b0 = M.gr_ent[2 * (size_t)k1], b1 = M.gr_ent[2 * (size_t)k1 + 1];
        SDFK_RCOUNT(1);
        if (k1 != k) SDFK_RCOUNT(1);
        const bool in0 = ray_entry_inside(q.x, q.y, a0, a1);
        const bool in1 = k1 != k && ray_entry_inside(q.x, q.y, b0, b1);
        const uint32_t i0 = __float_as_uint(a1.z), i1 = __float_as_uint(b1.z);
        float4 f00, f01, f02, f10, f11, f12;
        if (in0) f00 = M.rface[3 * (size_t)i0], f01 = M.rface[3 * (size_t)i0 + 1], f02 = M.rface[3 * (size_t)i0 + 2];
        if (in1) f10 = M.rface[3 * (size_t)i1], f11 = M.rface[3 * (size_t)i1 + 1], f12 = M.rface[3 * (size_t)i1 + 2];
        if (in0) ray_face_z_side(q, f00, f01, f02, zp, zm);
        if (in1) ray_face_z_side(q, f10, f11, f12, zp, zm);
    }
    for (uint32_t k = 0; k < M.gr_nbig; ++k) ray_face_z(M, q, M.gr_big[k], zp, zm);
}

// The x+ tie-break ray (sdf.cpp:326) on its own column grid over (y', z'): the same walk as the z rays.  The
// stackless LBVH walk it replaces met 1.5k (torus) to 21k (noisy icosphere-7) nodes per ray, one dependent
// load after the other in a single lane: rare (a handful of rays per million queries on a closed mesh) but
// each one kept its wave for up to 4 ms, the whole floor of a contains() call on a rough mesh.
__device__ __forceinline__ void ray_face_x(const MeshDev& M, V3 q, uint32_t face, int& c) {  // big list
    const float* bx = M.fbox + 6 * (size_t)face;
    if (!(!(q.y > bx[4] || bx[1] > q.y) && !(q.z > bx[5] || bx[2] > q.z) && !(q.x > bx[3] || bx[0] > FLT_MAX))) return;
    const float4 f0 = M.rface[3 * (size_t)face], f1 = M.rface[3 * (size_t)face + 1], f2 = M.rface[3 * (size_t)face + 2];
    const V3 nr = {f2.y, f2.z, f2.w};
    const V3 dv = sdfk::sub(q, V3{f0.x, f0.y, f0.z});
    const float s = sdfk::dot(nr, dv) * nr.x;  // ax_idx = 0, ax_inv = false
    if (!(s > 0.f))
        if (sdfk::bary2d_inside(q.y, q.z, f0.y, f0.z, f1.x, f1.y, f1.w, f2.x)) c ^= 1;
}
__device__ __forceinline__ int ray_x_pos(const MeshDev& M, V3 q) {
    if (!M.gx_start) return ray_x_pos_lbvh(M, q);
    int c = 0;
    const int cy = grid_cell(q.y, M.gx_o0, M.gx_inv0, M.gx_g), cz = grid_cell(q.z, M.gx_o1, M.gx_inv1, M.gx_g);
    const uint32_t cell = (uint32_t)cz * (uint32_t)M.gx_g + (uint32_t)cy;
    const uint32_t e = M.gx_start[cell + 1];
    SDFK_RCOUNT(4);
    SDFK_RADD(5, e - M.gx_start[cell]);
    SDFK_RMAX(6, e - M.gx_start[cell]);
    for (uint32_t k = M.gx_start[cell]; k < e; ++k) {
        const float4 e0 = M.gx_ent[2 * (size_t)k], e1 = M.gx_ent[2 * (size_t)k + 1];
        if (!ray_entry_inside(q.y, q.z, e0, e1)) continue;
        // Overlap in y, z and bary2d hold: the x side of Overlap (box [q.x, FLT_MAX]) and the facing rule
        const uint32_t face = __float_as_uint(e1.z);
        const float4 f0 = M.rface[3 * (size_t)face], f1 = M.rface[3 * (size_t)face + 1], f2 = M.rface[3 * (size_t)face + 2];
        const float mnx = sdfk::stdmin(sdfk::stdmin(f0.x, f0.w), f1.z), mxx = sdfk::stdmax(sdfk::stdmax(f0.x, f0.w), f1.z);
        if (q.x > mxx || mnx > FLT_MAX) continue;
        const V3 nr = {f2.y, f2.z, f2.w};
        const V3 dv = sdfk::sub(q, V3{f0.x, f0.y, f0.z});
        const float s = sdfk::dot(nr, dv) * nr.x;  // ax_idx = 0, ax_inv = false
        if (!(s > 0.f)) c ^= 1;
    }
    for (uint32_t k = 0; k < M.gx_nbig; ++k) ray_face_x(M, q, M.gx_big[k], c);
    return c;
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
        // the records of a vertex's incident faces, ascending face id (sdf.cpp:22-26), lie back to back in
        // geo_adj: no face-id load between the vertex and its faces, and one 384-byte run instead of six lines
        for (uint32_t k = fb; k < fe; ++k) {
            const float4 g0 = M.geo_adj[4 * (size_t)k];      // a.xyz b.x
            const float4 g1 = M.geo_adj[4 * (size_t)k + 1];  // b.yz c.xy
            const float4 g2 = M.geo_adj[4 * (size_t)k + 2];  // c.z n.xyz
            const float4 g3 = M.geo_adj[4 * (size_t)k + 3];  // area
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
                                                  const uint32_t* __restrict__ n_dev, uint32_t n_min = 0u) {
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
        if (n <= (int64_t)n_min) return;  // (a short heavy list is the exact scan's)
        // (a long list is not the sparse near-tie kind: whole packets as wide as the mesh were sent here, every lane works)
        if (n <= 262144) {
            if (tid & 3) return;
            first = tid >> 2;
            step = nthreads >> 2;
        }
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
