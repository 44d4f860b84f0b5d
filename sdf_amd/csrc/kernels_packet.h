// kernels_packet.h — the packet (one wave per 64 Morton-adjacent queries) nearest-vertex pass, the dominant
// kernel k_query_packet, the deferred-list compaction and the tail limiter's k_query_heavy.
// Part of sdfhip.hip (one translation unit).
#pragma once

// =========================================================================================
// Packet nearest-vertex search: one WAVE walks the kd-tree for 64 Morton-adjacent queries.
//
// Far-field queries make nanoflann's per-query search visit ~100 leaves / ~500 vertices, and
// 64 independent lanes doing that diverge badly (29 % VALU lane utilisation measured).  Here
// the traversal is wave-uniform (see "Wide packet walk" below): vertex groups come in by scalar
// loads and every lane evaluates every visited vertex (no divergence).  A group is skipped only
// when, for every lane, a rigorous lower bound of the distance to its box exceeds the lane's
// current best * c (with slack for every rounding on the way).
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
#ifdef SDFK_NO_NT
#define SDFK_NT_LOAD(P) (*(P))
#define SDFK_NT_STORE(V, P) (*(P) = (V))
#else
#define SDFK_NT_LOAD(P) __builtin_nontemporal_load(P)
#define SDFK_NT_STORE(V, P) __builtin_nontemporal_store((V), (P))
#endif
#ifndef SDFK_CHUNK_BATCH
#define SDFK_CHUNK_BATCH 3
#endif
#ifndef SDFK_PBLOCK
#define SDFK_PBLOCK 64
#endif
constexpr int PBLOCK = SDFK_PBLOCK;
// 80 SGPRs or fewer admit 8 waves per SIMD, 82-96 only 7 (MI355X_MICROARCH.md "Residency"); the
// compiler would take 90.  Measured: robust sdf kernel 16.3 -> 15.5 ms for 64M points, nn unchanged.
#ifndef SDFK_PK_NSGPR
#define SDFK_PK_NSGPR 96
#endif
#ifndef SDFK_PK_SGPR
#ifdef SDFK_PK_WAVES
#define SDFK_PK_SGPR __attribute__((amdgpu_num_sgpr(SDFK_PK_NSGPR), amdgpu_waves_per_eu(SDFK_PK_WAVES, SDFK_PK_WAVES)))
#elif defined(SDFK_PK_NVGPR)
#define SDFK_PK_SGPR __attribute__((amdgpu_num_sgpr(SDFK_PK_NSGPR), amdgpu_num_vgpr(SDFK_PK_NVGPR)))
#else
#define SDFK_PK_SGPR __attribute__((amdgpu_num_sgpr(SDFK_PK_NSGPR)))
#endif
#endif


// result of the packet pass for one lane
constexpr uint32_t H_NONE = 0x7FFFFFFFu;
struct PacketBest {
    float b;     // global fp32 minimum over all vertices
    float s;     // second smallest distance as a multiset (an exact tie shows as s == b); exact
                 // whenever it is <= b * c^2
    bool bailed;  // the wave exceeded its work budget: nothing is valid, see k_query_heavy
    bool lazy_ties = false;  // h names the holder GROUP even when several of its vertices are at distance b (group_exact_lazy)
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
// The chunk loads of the walk go through the CONSTANT address space: the tree arrays are never written while
// a query kernel runs, and saying so keeps these wave-uniform loads scalar (s_load_dwordx16) wherever they
// stand.  Through a generic pointer the compiler turns them into vector loads (2x slower walk) as soon as a
// store, an atomic or a barrier can precede them on any path — behind a loop's back edge, for one.
typedef const __attribute__((address_space(4))) u32x16* ChunkPtr;

#ifdef SDFK_STATS
// tuning build only (make stats): per-wave work counters of the packet pass
__device__ unsigned long long g_pk_stats[12];  // waves, wide tests, groups, chunks, tie redos, candidates, near ties, deferred,
                                               // leaf-box passes, of those failing the path walk (must be 0), path walks,
                                               // waves that took their cube's list
#define SDFK_COUNT(K, V) if ((threadIdx.x & 63) == 0) atomicAdd(&g_pk_stats[K], (unsigned long long)(V))
#else
#define SDFK_COUNT(K, V)
#endif

// ---- wave-wide reductions by DPP (all 64 lanes active, wave-uniform control flow) ----------------
#define SDFK_DPP(IDENT, V, CTRL, RMASK) \
    ((uint32_t)__builtin_amdgcn_update_dpp((int)(IDENT), (int)(V), (CTRL), (RMASK), 0xf, false))
// row_shr:1,2,4,8 fold each row of 16 into its last lane, row_bcast:15 / :31 fold the rows into lane 63
#define SDFK_WAVE_REDUCE(V, IDENT, OP)                          \
    V = OP(V, SDFK_DPP(IDENT, V, 0x111, 0xf));                  \
    V = OP(V, SDFK_DPP(IDENT, V, 0x112, 0xf));                  \
    V = OP(V, SDFK_DPP(IDENT, V, 0x114, 0xf));                  \
    V = OP(V, SDFK_DPP(IDENT, V, 0x118, 0xf));                  \
    V = OP(V, SDFK_DPP(IDENT, V, 0x142, 0xa));                  \
    V = OP(V, SDFK_DPP(IDENT, V, 0x143, 0xc));
__device__ __forceinline__ uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
    SDFK_WAVE_REDUCE(v, 0u, umax2)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
    SDFK_WAVE_REDUCE(v, 0xFFFFFFFFu, umin2)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// order-preserving map of finite / infinite floats onto unsigned integers (NaN sorts past +inf)
__device__ __forceinline__ uint32_t pk_f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float pk_ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ float wave_readf(float v, uint32_t l) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)l));
}

// =========================================================================================
// Wide packet walk.  The vertices' packet groups (maximal kd subtrees of <= group_max vertices, in
// position order) are bounded by ORIENTED boxes: the vertices lie on a surface, a patch of it is
// nearly flat, and an oriented box follows the patch where an axis-aligned one is as thick as the
// patch is wide (measured with a host-side model of the walk on the 424^2 torus: 2.6x fewer vertex slots per
// wave than the 4-wide tree of axis-aligned boxes this replaces).  Two levels, 64 children each:
// an L1 node is a run of 64 consecutive groups.  A node is tested against the PACKET by all lanes
// at once — lane k tests child k against the sphere around the wave's 64 queries — so a level
// costs ~40 VALU per wave where a per-lane box test of 4 children cost 56 per quad node; the
// children of the packet's candidates stay in the lanes' registers.
//
// Pruning is conservative at every step (a vertex v of child box B, query q_l of lane l, packet
// centre c, radius rho >= |q_l - c|, thresholds thr_l >= everything lane l must still see):
//   packet level   dist(c, B) > max_l sqrt(thr_l) + rho            =>  |q_l - v| > sqrt(thr_l) for all l
//   lane level     f0 + n.(q_l - c) > sqrt(thr_l)  for every lane  =>  the same, where the half space
//                  {x : n.(x - c) <= -f0}, |n| <= 1, contains B (n = direction from B to c, f0 from
//                  the support function of B): a plane bound that is nearly as sharp as the exact
//                  per-lane box distance (same model: 8.1 against 7.4 groups per wave) for 4 readlanes
//                  and 3 FMAs per candidate.
// Rounding: box records are fp32; the half extents are computed against the STORED centre and axes
// and rounded up (k_group_obb), the projections' absolute error (<= 8u |c - c_B|) is covered by Ew
// (64u x an upper bound of |c - c_B|), the relative ones by EPS = 2^-18 on the thresholds.  Every
// prune is of the form "x > y is true", so a NaN anywhere means "visit".
// What is exact is unchanged: every evaluated group runs the reference's arithmetic, and the walk
// ends with the same (b, s, h) / (B1..B4, H1..H3) as the tree walk it replaces.
// =========================================================================================
constexpr float PK_EPS = 3.8146973e-6f;  // 2^-18
#ifndef SDFK_CUBE_CAP
#define SDFK_CUBE_CAP 512
#endif
constexpr uint32_t CUBE_CAP = SDFK_CUBE_CAP;        // entries of one cube's list (eight wide tests: still half the walk's); longer: the cube is unusable
constexpr uint32_t CUBE_NONE = 0xFFFFFFFFu;
struct ObbLane {
    float dP;                 // squared distance from the packet centre to this lane's child box (lower bound)
    float nx, ny, nz, f0;     // half space containing the box: n.(x - c) <= -f0
    float pk;                 // f0 - |n - g| rho: packet-level bound against the thresholds' linear trend g
    uint32_t tag;             // group ref (group level)
};
// one lane tests one box record against c.  `recp`: the lane's record, its four float4 `stride` apart — stride 64
// for the lane-major node arrays ([4][64] float4, see k_group_obb: lane k reads child k with unit stride), 1 for
// the per-group copy the cube lists gather from (kd_gobb_aos)
template <bool PLANE>
__device__ __forceinline__ ObbLane obb_wide_at(const float4* __restrict__ recp, uint32_t stride, float cx, float cy,
                                               float cz, float Ew, float gx = 0.f, float gy = 0.f, float gz = 0.f,
                                               float hx = 0.f, float hy = 0.f, float hz = 0.f) {
    const float4 A = recp[0], B = recp[stride], C = recp[2u * stride], D = recp[3u * stride];
    const float dx = cx - A.x, dy = cy - A.y, dz = cz - A.z;
    const float p0 = __builtin_fmaf(B.x, dx, __builtin_fmaf(B.y, dy, B.z * dz));
    const float p1 = __builtin_fmaf(C.x, dx, __builtin_fmaf(C.y, dy, C.z * dz));
    const float p2 = __builtin_fmaf(D.x, dx, __builtin_fmaf(D.y, dy, D.z * dz));
    const float g0 = fmaxf(fabsf(p0) - (A.w + Ew), 0.f);
    const float g1 = fmaxf(fabsf(p1) - (B.w + Ew), 0.f);
    const float g2 = fmaxf(fabsf(p2) - (C.w + Ew), 0.f);
    ObbLane o;
    o.dP = __builtin_fmaf(g0, g0, __builtin_fmaf(g1, g1, g2 * g2));
    o.tag = __float_as_uint(D.w);
    o.nx = o.ny = o.nz = o.f0 = o.pk = 0.f;
    if (PLANE) {
        // n = sum_a sign(p_a) g_a R_a / |g|, scaled down so that |n| <= 1 whatever the rounding
        const float inv = o.dP > 0.f ? __builtin_amdgcn_rsqf(o.dP) * (1.f - PK_EPS) : 0.f;
        const float s0 = copysignf(g0, p0) * inv, s1 = copysignf(g1, p1) * inv, s2 = copysignf(g2, p2) * inv;
        o.nx = __builtin_fmaf(s0, B.x, __builtin_fmaf(s1, C.x, s2 * D.x));
        o.ny = __builtin_fmaf(s0, B.y, __builtin_fmaf(s1, C.y, s2 * D.y));
        o.nz = __builtin_fmaf(s0, B.z, __builtin_fmaf(s1, C.z, s2 * D.z));
        // f0 = n.(c - c_B) - max_{x in B} n.(x - c_B) = n.(c - c_B) - sum_a h_a |n.R_a|.  The stored axes are
        // orthonormal up to rounding (|R_a.R_b - delta_ab| <= 2.1u, n itself carries <= 3u |s|_1 per component),
        // so |n.R_a - s_a| <= 5.1u |s|_1 <= 9u: s_a stands for n.R_a (9 FMAs less per test) and the difference,
        // <= 27u x the largest half extent < Ew / 2, is one more Ew of slack.
        const float sup = __builtin_fmaf(A.w, fabsf(s0), __builtin_fmaf(B.w, fabsf(s1), C.w * fabsf(s2)));
        o.f0 = __builtin_fmaf(o.nx, dx, __builtin_fmaf(o.ny, dy, o.nz * dz)) - (sup + 5.f * Ew);
        // for every lane l:  f0 + n.e_l - rr_l  =  f0 + (n - g).e_l - (rr_l - g.e_l)  >=  pk - max_l (rr_l - g.e_l)
        // with |(n - g).e_l| <= sum_i |n_i - g_i| h_i, h = half extents of the queries' bounding box (rounded up)
        o.pk = o.f0 - __builtin_fmaf(fabsf(o.nx - gx), hx, __builtin_fmaf(fabsf(o.ny - gy), hy, fabsf(o.nz - gz) * hz));
    }
    return o;
}
// lane `lane` tests child `lane` of one node (rec: [4][64] float4) against c
template <bool PLANE>
__device__ __forceinline__ ObbLane obb_wide(const float4* __restrict__ rec, uint32_t lane, float cx, float cy,
                                            float cz, float Ew, float gx = 0.f, float gy = 0.f, float gz = 0.f,
                                            float hx = 0.f, float hy = 0.f, float hz = 0.f) {
    return obb_wide_at<PLANE>(rec + lane, 64u, cx, cy, cz, Ew, gx, gy, gz, hx, hy, hz);
}
// sort key of a box distance: its bit pattern (non-negative floats order like integers); NaN -> 0 (nearest)
__device__ __forceinline__ uint32_t obb_key(float dP, bool valid) {
    return valid ? (dP == dP ? __float_as_uint(dP) : 0u) : 0xFFFFFFFFu;
}

// per-lane state of the two flavours of the pass (what a group evaluation updates)
struct ExactState {
    uint32_t b, s2, h, tieb;  // see PacketBest; tieb: value of a settled exact tie (s2 is kept at b + 1)
};
struct ApproxState {
    float B1, B2, B3, B4;
    uint32_t H1, H2, H3;
};

// exact evaluation of group `cur` for the wave (reference arithmetic, 10 VALU per slot)
__device__ __forceinline__ void group_exact(const MeshDev& M, float qx, float qy, float qz, uint32_t cur,
                                            ExactState& S) {
    const uint32_t nch = cur & 15u;
    const uint32_t first = ((cur & sdfk::KD_REF_MASK) >> 4) << 2;
    const float4* P = M.kd_pts4 + first;
    uint32_t b = S.b, s2 = S.s2, h = S.h;
    const uint32_t b0 = b, s0 = s2, h0 = h;
    const ChunkPtr C = (ChunkPtr)__builtin_assume_aligned(P, 64);
    uint32_t ci = 0;
#if SDFK_CHUNK_BATCH >= 3
    for (; ci + 3u <= nch; ci += 3u) {
        const u32x16 A0 = C[ci], A1 = C[ci + 1u], A2 = C[ci + 2u];
        SDFK_PACKET_CHUNK(A0) SDFK_PACKET_CHUNK(A1) SDFK_PACKET_CHUNK(A2)
    }
#endif
#if SDFK_CHUNK_BATCH >= 2
    for (; ci + 2u <= nch; ci += 2u) {
        const u32x16 A0 = C[ci], A1 = C[ci + 1u];
        SDFK_PACKET_CHUNK(A0) SDFK_PACKET_CHUNK(A1)
    }
#endif
    for (; ci < nch; ++ci) {
        const u32x16 A0 = C[ci];
        SDFK_PACKET_CHUNK(A0)
    }
    if (b != b0) h = cur;  // strictly improved here: the holder is in this group
    if (__ballot(s2 == b) != 0ull) {
        // rare: some lane's two smallest distances are equal.  Redo the group for the wave in
        // ascending position and settle each exact tie by DFS order.
        b = b0, s2 = s0, h = h0;
        SDFK_COUNT(4, 1);
        for (uint32_t k = 0; k < 4u * nch; ++k) {
            const float4 Qp = P[k];
            const float dx = qx - Qp.x, dy = qy - Qp.y, dz = qz - Qp.z;
            const uint32_t du = __float_as_uint((dx * dx + dy * dy) + dz * dz);
            const uint32_t idx = first + k;
            if (du == b && h != H_NONE) {
                // any position of the holder's group stands for the holder: the two part above that group
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
        if (s2 == b) S.tieb = b, s2 = b + 1u;
    }
    S.b = b, S.s2 = s2, S.h = h;
}

// Exact evaluation with exact ties left for later (far packets: fp32 distances at 300 mesh sizes are quantised so
// coarsely that ties are the rule — group_exact's redo, a sequential pass over the group for the whole wave whenever
// some lane sees two equal smallest distances, ran 4.5 times per wave there, a third of the far kernel).  What the
// finish needs is (b, s2 with s2 == b flagging a tie, h = the group holding the tied vertex nanoflann's DFS meets
// first): s2 == b comes out of the fast path by itself; a strict improvement makes this group the holder; and a
// group that does not improve b but holds a vertex AT b (one more compare per slot, against the b before the group)
// ties with the holder across groups — groups are disjoint kd subtrees, so the DFS order of any two of their
// vertices is the order of the groups, decided by one kd_dfs_first on their first positions.  Ties inside the
// holder are folded by packet_resolve_ties at the end.
#define SDFK_PACKET_SLOT_EQ(C_, O_)                                                \
    {                                                                              \
        const float dx_ = qx - __uint_as_float(C_[O_]);                            \
        const float dy_ = qy - __uint_as_float(C_[O_ + 1]);                        \
        const float dz_ = qz - __uint_as_float(C_[O_ + 2]);                        \
        const uint32_t du_ = __float_as_uint((dx_ * dx_ + dy_ * dy_) + dz_ * dz_); \
        eq = eq || du_ == b0;                                                      \
        s2 = umed3(b, s2, du_);                                                    \
        b = min(b, du_);                                                           \
    }
#define SDFK_PACKET_CHUNK_EQ(C_) \
    SDFK_PACKET_SLOT_EQ(C_, 0) SDFK_PACKET_SLOT_EQ(C_, 4) SDFK_PACKET_SLOT_EQ(C_, 8) SDFK_PACKET_SLOT_EQ(C_, 12)
__device__ __forceinline__ void group_exact_lazy(const MeshDev& M, float qx, float qy, float qz, uint32_t cur,
                                                 ExactState& S) {
    const uint32_t nch = cur & 15u;
    const uint32_t first = ((cur & sdfk::KD_REF_MASK) >> 4) << 2;
    uint32_t b = S.b, s2 = S.s2, h = S.h;
    const uint32_t b0 = b, h0 = h;
    bool eq = false;
    const ChunkPtr C = (ChunkPtr)__builtin_assume_aligned(M.kd_pts4 + first, 64);
    uint32_t ci = 0;
    for (; ci + 2u <= nch; ci += 2u) {
        const u32x16 A0 = C[ci], A1 = C[ci + 1u];
        SDFK_PACKET_CHUNK_EQ(A0) SDFK_PACKET_CHUNK_EQ(A1)
    }
    for (; ci < nch; ++ci) {
        const u32x16 A0 = C[ci];
        SDFK_PACKET_CHUNK_EQ(A0)
    }
    if (b != b0) {
        h = cur;  // strictly improved here: the holder is in this group
    } else if (eq && h0 != H_NONE && h0 != cur) {  // a vertex at the standing minimum: which group comes first?
        const uint32_t hp = (h0 & KD_LEAF) ? (((h0 & sdfk::KD_REF_MASK) >> 4) << 2) : h0;
        const bool cur_first = M.fast ? first < hp : (uint32_t)kd_dfs_first(M, qx, qy, qz, (int)hp, (int)first) == first;
        if (cur_first) h = cur;
    }
    S.b = b, S.s2 = s2, S.h = h;
}
#undef SDFK_PACKET_CHUNK_EQ
#undef SDFK_PACKET_SLOT_EQ

// Error model of the approximate pass's estimates (group-centred form, k_apx_fill).  With u = 2^-24, R >= |w| for
// every stored w = fl(v - cg), t = fl(q - cg), d = |q - v|:  the estimate est = fl(a'' + Qg),
// a'' = fma chain of W - 2 t.w, Qg = fma chain of |t|^2, differs from d^2 by
//     2 d u (|t| + |w|)       roundings of t and w against the true q - cg, v - cg
//   + 3 u |t|^2 + u |w|^2     roundings of Qg, of W
//   + 3 u (2 |t||w| + |w|^2)  the three FMA roundings of a''
//   + u est                   the final add
// and with |t| <= d + R:  |est - d^2| <= u (6 d^2 + 16 d R + 13 R^2) <= u (14 d^2 + 21 R^2) (1 + small).  In terms of
// the estimate itself: E(est) = u (16 max(est, 0) + 24 R^2) >= that (M.apx_k1 = 16 u, M.apx_e0 = 24 u R^2, rounded
// up by the host).  The reference's own D(v) lies within 5.01 u of d^2, hence
//     lower(est) = (est - E)(1 - 8u)  <=  D(v)  <=  (est + E)(1 + 8u) = upper(est).
__device__ __forceinline__ float apx_err(const MeshDev& M, float est) {
    return __builtin_fmaf(fmaxf(est, 0.f), M.apx_k1, M.apx_e0);
}
__device__ __forceinline__ float apx_upper(const MeshDev& M, float est) {  // (NaN / inf pass through: prune nothing)
    const float x = est + apx_err(M, est);
    return fmaxf(fmaxf(x * (1.f + 8.f * 5.9604645e-8f), x), 0.f);
}
__device__ __forceinline__ float apx_lower(const MeshDev& M, float est) {
    // (est = +inf — a lane that has met fewer than four groups — must stay +inf: inf - inf would be a NaN, which every
    // window test reads as "inside": such lanes went to the exact scan as fourth-candidate queries)
    const float y = est - fminf(apx_err(M, est), FLT_MAX);
    return fminf(y * (1.f - 8.f * 5.9604645e-8f), y);
}

// approximate evaluation (see "Approximate first pass" below): group minimum of a(), sorted insert
// DEDUPE (continuation of a walk by the cooperative kernel): a lane that already lists `cur` among its first
// three groups keeps its list (the group was evaluated before the hand-over).
template <bool DEDUPE>
__device__ __forceinline__ void group_approx(const MeshDev& M, float qx, float qy, float qz, uint32_t cur,
                                             ApproxState& S) {
    const uint32_t nch = cur & 15u;
    const uint32_t first = ((cur & sdfk::KD_REF_MASK) >> 4) << 2;
    const ChunkPtr C = (ChunkPtr)__builtin_assume_aligned(M.kd_apx4 + first, 64);
    // the group's centre (a scalar load beside the chunk loads), the query relative to it, its squared norm
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 cg = ((const __attribute__((address_space(4))) f32x4*)M.kd_actr)[first >> 2];
    const float X2 = qx - cg.x, Y2 = qy - cg.y, Z2 = qz - cg.z;  // t = fl(q - cg); the slots hold -2 w
    const float Qg = __builtin_fmaf(X2, X2, __builtin_fmaf(Y2, Y2, Z2 * Z2));
    float gm = __builtin_inff();  // (a NaN never enters: fminf keeps the other operand)
#define SDFK_APX_SLOT(A, O_)                                                               \
    __builtin_fmaf(X2, __uint_as_float(A[O_]),                                             \
                   __builtin_fmaf(Y2, __uint_as_float(A[O_ + 1]),                          \
                                  __builtin_fmaf(Z2, __uint_as_float(A[O_ + 2]), __uint_as_float(A[O_ + 3]))))
    // 12 FMA + 2 v_min3 per chunk (a NaN operand is dropped by min3 as by min)
#define SDFK_APX_CHUNK(A)                                                                                  \
    {                                                                                                      \
        const float a0 = SDFK_APX_SLOT(A, 0), a1 = SDFK_APX_SLOT(A, 4), a2 = SDFK_APX_SLOT(A, 8),          \
                    a3 = SDFK_APX_SLOT(A, 12);                                                             \
        gm = fmin3(gm, a0, a1);                                                                            \
        gm = fmin3(gm, a2, a3);                                                                            \
    }
    // The walk is a chain of dependent loads: the chunk loads of a group go out SDFK_CHUNK_BATCH at a time
    uint32_t ci = 0;
#if SDFK_CHUNK_BATCH >= 3
    for (; ci + 3u <= nch; ci += 3u) {
        const u32x16 A0 = C[ci], A1 = C[ci + 1u], A2 = C[ci + 2u];
        SDFK_APX_CHUNK(A0) SDFK_APX_CHUNK(A1) SDFK_APX_CHUNK(A2)
    }
#endif
#if SDFK_CHUNK_BATCH >= 2
    for (; ci + 2u <= nch; ci += 2u) {
        const u32x16 A0 = C[ci], A1 = C[ci + 1u];
        SDFK_APX_CHUNK(A0) SDFK_APX_CHUNK(A1)
    }
#endif
    for (; ci < nch; ++ci) {
        const u32x16 A0 = C[ci];
        SDFK_APX_CHUNK(A0)
    }
#undef SDFK_APX_CHUNK
#undef SDFK_APX_SLOT
    gm += Qg;  // the estimate of the squared distance itself (+inf stays +inf; inf - inf = NaN never enters below)
    // sorted insert of this group's minimum into B1..B4 (+ the groups of the first three);
    // skipped when no lane's top four change (the insert is then the identity)
    if (DEDUPE && (cur == S.H1 || cur == S.H2 || cur == S.H3)) gm = __builtin_inff();
    if (__ballot(gm < S.B4) != 0ull) {
        const bool c1 = gm < S.B1, c2 = gm < S.B2, c3 = gm < S.B3;
        S.H3 = c2 ? S.H2 : (c3 ? cur : S.H3);
        S.H2 = c1 ? S.H1 : (c2 ? cur : S.H2);
        S.H1 = c1 ? cur : S.H1;
        float t = fmaxf(S.B1, gm);
        S.B1 = fminf(S.B1, gm);
        float t2 = fmaxf(S.B2, t);
        S.B2 = fminf(S.B2, t);
        t = fmaxf(S.B3, t2);
        S.B3 = fminf(S.B3, t2);
        S.B4 = fminf(S.B4, t);
    }
}

// =========================================================================================
// Approximate first pass (strict mode).  The exact evaluation spends 10 VALU per (query, vertex) on the
// reference's arithmetic although almost every vertex it meets is NOT the nearest one.
// Here a slot costs 3 FMA + half a min3: per packet group g with centre cg (k_apx_fill) the slots hold
// -2 w, |w|^2 for w = fl(v - cg), the lane forms t = fl(q - cg), |t|^2 once per group, and
//     est(v) = fma(tx, -2wx, fma(ty, -2wy, fma(tz, -2wz, |w|^2))) + |t|^2   ~   D(v)
// orders the vertices of one lane like nanoflann's fp32 distance D(v) up to the error E(est) derived at
// apx_err() above: u (16 est + 24 R^2), R the largest group radius — of the size of the distance and of a
// group, not of the scene (round 2 centred on the mesh: u (5 S + 4 Q) with S, Q the scene's squared size,
// which ruled the pass out for dense meshes).  With
//     D(v) = d^2(v) (1 + t), |t| <= 5.01 u               (nanoflann.hpp:432-440, five roundings)
// lower(est) <= D(v) <= upper(est).  Per lane the walk keeps the four smallest GROUP
// minima B1 <= B2 <= B3 <= B4 of est and the groups of the first three.  Pruning uses upper(B1)*c^2,
// an upper bound of what the exact walk would use, so every vertex the exact walk needs is met.
// Afterwards each lane evaluates EXACTLY (reference arithmetic, tie rules of the exact walk) the
// groups whose minimum could still be the nearest: H1, and Hk while lower(Bk) <= upper(B1).
// Every vertex outside them has D >= lower(B_next) > upper(B1) >= the exact minimum b found
// inside them, strictly: b is the global minimum, no outside vertex ties with it, and
// lower(B_next) is a lower bound of the distance of every outside vertex — which is all the
// decision logic needs of `s`.  A fourth candidate group defers the query to the strict kernel.
// =========================================================================================

// The walk itself.  APPROX: SA is updated (thresholds from upper(B1) * c^2, cm = c^2),
// else SE (thresholds b * c^2).  Returns true when the wave gave up (work budget, see k_query_heavy).
// CONT (cooperative kernel, APPROX only): SA holds the state of a walk that passed M.coop_budget in the main
// kernel; this wave continues it over the L1 nodes j with j % nparts == part (the walk is redone from the
// first L1 node with the saved, nearly final thresholds: groups met again are recognised per lane).
// budget: work units (chunks evaluated + candidates tested) after which the wave gives up.
// Returns 0 (done), 1 (gave up on the work budget) or 2: the packet is as wide as the mesh (queries spread over a
// region much larger than the mesh: every group is a candidate for some lane) — nothing was done and the
// queries go straight to the heavy list (no walk, no cooperative continuation); when that list is long the host
// has the strict kernel answer it, each lane walking the tree for itself, instead of the exact scan.
template <bool APPROX, bool CONT>
__device__ __forceinline__ uint32_t packet_walk(const MeshDev& M, float qx, float qy, float qz, float cm,
                                            ExactState& SE, ApproxState& SA,
                                            uint32_t part, uint32_t nparts, uint32_t budget) {
    const uint32_t lane = threadIdx.x & 63u;
    SDFK_COUNT(0, 1);
    if (M.kd_ngroups == 0u) return 0u;
    // sphere around the wave's queries: centre of their bounding box, radius = farthest query
    const float cx = 0.5f * (pk_ord2f(wave_umin(pk_f2ord(qx))) + pk_ord2f(wave_umax(pk_f2ord(qx))));
    const float cy = 0.5f * (pk_ord2f(wave_umin(pk_f2ord(qy))) + pk_ord2f(wave_umax(pk_f2ord(qy))));
    const float cz = 0.5f * (pk_ord2f(wave_umin(pk_f2ord(qz))) + pk_ord2f(wave_umax(pk_f2ord(qz))));
    const float ex = qx - cx, ey = qy - cy, ez = qz - cz;
    // half extents of the queries' box around c (>= |e| of every lane, with a margin for the dot products)
    const float hx = __uint_as_float(wave_umax(__float_as_uint(fabsf(ex)))) * (1.f + 4.f * PK_EPS);
    const float hy = __uint_as_float(wave_umax(__float_as_uint(fabsf(ey)))) * (1.f + 4.f * PK_EPS);
    const float hz = __uint_as_float(wave_umax(__float_as_uint(fabsf(ez)))) * (1.f + 4.f * PK_EPS);
    const float rho = __builtin_amdgcn_sqrtf(__builtin_fmaf(hx, hx, __builtin_fmaf(hy, hy, hz * hz))) * (1.f + PK_EPS);
    if (!CONT && !M.fast && rho > 0.5f * M.obb_rm) return 2u;  // (wave-uniform; obb_rm = sum of the mesh's extents)
    // absolute slack of the box tests: 64u x (an upper bound of |c - centre of any box|)
    const float reach = ((fabsf(cx - M.obb_cm[0]) + fabsf(cy - M.obb_cm[1])) + fabsf(cz - M.obb_cm[2])) + (M.obb_rm + rho);
    const float Ew = 3.8146973e-6f * reach;
    const float c2 = M.amb_c2;
    float rr = __builtin_inff();  // per lane: sqrt of the lane's threshold, with the slack of the plane test (inf: prunes nothing)
#define SDFK_PK_REFRESH()                                                                           \
    {                                                                                               \
        const float thr_ = APPROX ? apx_upper(M, SA.B1) * cm : __uint_as_float(SE.b) * c2;          \
        rr = __builtin_fmaf(__builtin_amdgcn_sqrtf(thr_), 1.f + PK_EPS, PK_EPS * rho);              \
    }
    // (the thresholds only move when some lane's minimum did: most groups past the first few change nothing)
#define SDFK_PK_EVAL(REF)                                                                 \
    {                                                                                     \
        SDFK_COUNT(2, 1);                                                                 \
        SDFK_COUNT(3, (REF) & 15u);                                                       \
        work += (REF) & 15u;                                                              \
        const uint32_t old_ = APPROX ? __float_as_uint(SA.B1) : SE.b;                     \
        if (APPROX)                                                                       \
            group_approx<CONT>(M, qx, qy, qz, (REF), SA);                                 \
        else                                                                              \
            group_exact(M, qx, qy, qz, (REF), SE);                                        \
        moved = __ballot(old_ != (APPROX ? __float_as_uint(SA.B1) : SE.b)) != 0ull;       \
        if (moved) SDFK_PK_REFRESH()                                                      \
    }
    uint32_t work = 0;  // chunks evaluated + candidates tested so far (wave-uniform)
    bool moved = false;
    float gx, gy, gz;
    // packet-level threshold: max over the lanes of rr_l - g.e_l (+ slack of the dot product)
#define SDFK_PK_RES() \
    (pk_ord2f(wave_umax(pk_f2ord(rr - __builtin_fmaf(gx, ex, __builtin_fmaf(gy, ey, gz * ez))))) + 2.f * PK_EPS * rho)
    // The candidates of one wide test (box record O in the lanes' registers, ALIVE: lanes whose child passed the
    // packet-level test): nearest first by the packet-level bound — the thresholds are final after the first few
    // groups, and once the nearest remaining child fails the packet-level test all do — then the lane-level
    // plane test, then the evaluation.
#define SDFK_PK_CANDS(O, ALIVE)                                                                                   \
    for (;;) {                                                                                                    \
        const uint32_t key_ = (ALIVE) ? pk_f2ord((O).pk) : 0xFFFFFFFFu;                                           \
        const uint32_t mk_ = wave_umin(key_);                                                                     \
        if (mk_ == 0xFFFFFFFFu || pk_ord2f(mk_) > res) break;                                                     \
        const uint32_t g_ = (uint32_t)__builtin_ctzll(__ballot(key_ == mk_));                                     \
        if (lane == g_) ALIVE = false;                                                                            \
        SDFK_COUNT(5, 1);                                                                                         \
        ++work;                                                                                                   \
        const float t_ = __builtin_fmaf(wave_readf((O).nx, g_), ex,                                               \
                                        __builtin_fmaf(wave_readf((O).ny, g_), ey,                                \
                                                       __builtin_fmaf(wave_readf((O).nz, g_), ez, wave_readf((O).f0, g_)))); \
        if (__ballot(!(t_ > rr)) == 0ull) continue;                                                               \
        const uint32_t ref_ = (uint32_t)__builtin_amdgcn_readlane((int)(O).tag, (int)g_);                         \
        SDFK_PK_EVAL(ref_)                                                                                        \
        /* give up (wave-uniform): the cooperative kernel / k_query_heavy takes over */                           \
        if (work > (CONT ? budget : M.heavy_budget)) return 1u;                                                   \
        if (moved) res = SDFK_PK_RES(); /* tightened */                                                           \
    }
    // ---- cube lists (k_cube_lists): the groups any query of this wave's cube can need were listed once per cube;
    // the wave tests that list — lane k its k-th entry — instead of walking the two levels of boxes from the top.
    if (!CONT && M.cube_max != 0u) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) float* CFloatPtr;
        typedef const __attribute__((address_space(4))) f32x4* CFloat4Ptr;
        typedef const __attribute__((address_space(4))) uint32_t* CUintPtr;
        const CFloatPtr par = (CFloatPtr)M.cube_par;  // {lo.xyz, cubes per unit length .xyz, cubes per axis .xyz (integers)}
        // the cube of lane 0's query (any cube will do: the wave checks that all its queries lie in the cube's box)
        const float q0x = wave_readf(qx, 0u), q0y = wave_readf(qy, 0u), q0z = wave_readf(qz, 0u);
        const uint32_t ncx = __float_as_uint(par[6]), ncy = __float_as_uint(par[7]), ncz = __float_as_uint(par[8]);
        const float fx = fminf(fmaxf((q0x - par[0]) * par[3], 0.f), (float)(ncx - 1u));  // (NaN -> 0)
        const float fy = fminf(fmaxf((q0y - par[1]) * par[4], 0.f), (float)(ncy - 1u));
        const float fz = fminf(fmaxf((q0z - par[2]) * par[5], 0.f), (float)(ncz - 1u));
        const uint32_t cube = __builtin_amdgcn_readfirstlane(((uint32_t)fx * ncy + (uint32_t)fy) * ncz + (uint32_t)fz);
        const CFloat4Ptr hd = (CFloat4Ptr)M.cube_hdr + 3u * (size_t)cube;
        const f32x4 H0 = hd[0], H1 = hd[1], H2 = hd[2];  // {c.xyz, count}, {h.xyz, -}, {g.xyz, -}
        const uint32_t cnt = __float_as_uint(H0.w);
        const bool inside = fabsf(qx - H0.x) <= H1.x && fabsf(qy - H0.y) <= H1.y && fabsf(qz - H0.z) <= H1.z;
        if (cnt - 1u < CUBE_CAP && __ballot(!inside) == 0ull) {  // (wave-uniform; an unusable cube has count 0xFFFFFFFF)
            SDFK_COUNT(11, 1);
            const CUintPtr lst = (CUintPtr)M.cube_list + (size_t)CUBE_CAP * cube;
            {   // entry 0 first and unconditionally: the cube-level exclusion assumes every wave has evaluated it
                const uint32_t id0 = lst[0];
                const uint32_t ref0 = __float_as_uint(((CFloat4Ptr)M.kd_gobb_aos)[4u * (size_t)id0 + 3u].w);
                SDFK_PK_EVAL(ref0)
            }
            gx = H2.x, gy = H2.y, gz = H2.z;
            float res = SDFK_PK_RES();
            // (Candidates in list order off a ballot mask, the list sorted by the cube-level bound — no wave reduction
            // per candidate — were measured: 11.6 instead of 10.8 evaluated groups per wave, kernel 11.5 against 11.4 ms
            // for nn and 15.2 against 15.0 ms for sdf at 125M points: the wave's own nearest-first order stays.)
            for (uint32_t base = 0; base < cnt; base += 64u) {
                const uint32_t idx = base + lane;
                const uint32_t id = M.cube_list[(size_t)CUBE_CAP * cube + (idx < cnt ? idx : 0u)];
                const ObbLane o = obb_wide_at<true>(M.kd_gobb_aos + 4u * (size_t)id, 1u, cx, cy, cz, Ew, gx, gy, gz, hx, hy, hz);
                SDFK_COUNT(1, 1);
                bool alive = idx < cnt && idx != 0u && !(o.pk > res);
                SDFK_PK_CANDS(o, alive)
            }
            return 0u;
        }
    }
    const uint32_t nrows = (M.kd_nl1 + 63u) >> 6;
    // ---- greedy descent: the nearest L1 node, its nearest group; evaluating it gives every lane a bound
    uint32_t j0, g0;
    {
        uint32_t bkey = 0xFFFFFFFFu, brow = 0u;
        for (uint32_t row = 0; row < nrows; ++row) {
            const ObbLane o = obb_wide<false>(M.kd_l1obb + 256u * (size_t)row, lane, cx, cy, cz, Ew);
            const uint32_t key = obb_key(o.dP, 64u * row + lane < M.kd_nl1);
            if (key < bkey) bkey = key, brow = row;
        }
        SDFK_COUNT(1, nrows);
        const uint32_t mkey = wave_umin(bkey);
        const uint32_t who = (uint32_t)__builtin_ctzll(__ballot(bkey == mkey));
        j0 = 64u * (uint32_t)__builtin_amdgcn_readlane((int)brow, (int)who) + who;
        const uint32_t cnt = min(64u, M.kd_ngroups - 64u * j0);
        const ObbLane o = obb_wide<true>(M.kd_gobb + 256u * (size_t)j0, lane, cx, cy, cz, Ew);
        SDFK_COUNT(1, 1);
        const uint32_t key = obb_key(o.dP, lane < cnt);
        const uint32_t mk = wave_umin(key);
        g0 = (uint32_t)__builtin_ctzll(__ballot(key == mk));
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)o.tag, (int)g0);
        if (CONT) {  // evaluated before the hand-over: only the thresholds are needed
            SDFK_PK_REFRESH()
        } else {
            SDFK_PK_EVAL(ref)
        }
        // Across the packet the thresholds follow the distance field, rr_l ~ r0 + g.e_l with g the
        // direction away from the nearest patch: the packet-level tests bound rr_l - g.e_l instead of
        // rr_l alone (any g is valid; this one makes the residual small in the far field).
        gx = wave_readf(o.nx, g0), gy = wave_readf(o.ny, g0), gz = wave_readf(o.nz, g0);
        if (!(fabsf(gx) <= 1.f && fabsf(gy) <= 1.f && fabsf(gz) <= 1.f)) gx = gy = gz = 0.f;
    }
    // ---- the walk: L1 rows -> candidate L1 nodes -> candidate groups -> plane test -> evaluation
    // Children are taken nearest first (by their packet-level bound), at both levels.
    for (uint32_t row = 0; row < nrows; ++row) {
        const ObbLane o1 = obb_wide<true>(M.kd_l1obb + 256u * (size_t)row, lane, cx, cy, cz, Ew, gx, gy, gz, hx, hy, hz);
        SDFK_COUNT(1, 1);
        float res = SDFK_PK_RES();
        bool alive1 = 64u * row + lane < M.kd_nl1 && !(o1.pk > res) && (!CONT || (64u * row + lane) % nparts == part);
        for (;;) {
            const uint32_t key = alive1 ? pk_f2ord(o1.pk) : 0xFFFFFFFFu;
            const uint32_t mk = wave_umin(key);
            if (mk == 0xFFFFFFFFu || pk_ord2f(mk) > res) break;
            const uint32_t k = (uint32_t)__builtin_ctzll(__ballot(key == mk));
            if (lane == k) alive1 = false;
            const uint32_t j = 64u * row + k;
            const uint32_t cnt = min(64u, M.kd_ngroups - 64u * j);
            const ObbLane o = obb_wide<true>(M.kd_gobb + 256u * (size_t)j, lane, cx, cy, cz, Ew, gx, gy, gz, hx, hy, hz);
            SDFK_COUNT(1, 1);
            // (g0: evaluated by the greedy descent.  Not so in a continuation: the main kernel may have taken its
            // cube's list, whose first group is the cube's, not this wave's — the group is evaluated again, which
            // the lists take as the identity: a lane that holds it among its first three skips it (DEDUPE), for the
            // others its minimum is >= their fourth)
            bool alive = lane < cnt && !(o.pk > res) && (CONT || !(j == j0 && lane == g0));
            SDFK_PK_CANDS(o, alive)
        }
    }
#undef SDFK_PK_CANDS
#undef SDFK_PK_RES
#undef SDFK_PK_EVAL
#undef SDFK_PK_REFRESH
    return 0u;
}

// =========================================================================================
// Cube lists.  A wave's walk spends most of its instructions finding its ~30 candidate groups: ~14 wide box
// tests over the two levels, one dependent 4 KB load each, for ~12 evaluated groups.  Neighbouring waves find
// nearly the same candidates.  So the query domain (the bounds of the running launch's points) is cut into
// nc^3 cubes holding a few thousand queries each, and ONE wave per cube lists, once, every group that any
// query inside the cube can need; the ~60 waves whose queries lie in that cube then test the list — lane k its
// k-th entry, one or two wide tests — instead of walking from the top.
//
// What a list must hold.  Every wave that uses it evaluates entry 0 (the group G0 nearest to the cube's centre
// c) first.  After that, lane l (query q = c + e, |e_i| <= h_i, |e| <= rho) has a threshold
//     thr(q) <= (D_q(v0) [exact walk]  |  upper(est_q(v0)) <= d^2(q, v0) (1 + 50u) + 80u R^2 [approximate walk]) * (1 + ~1e-5)
// for the vertex v0 of G0 nearest to c, and thresholds only shrink.  With d0 = |c - v0|, g = (c - v0) / d0:
//     |q - v0| = sqrt(d0^2 + 2 d0 g.e + |e|^2) <= d0 + g.e + rho^2 / (2 d0)          (sqrt(1 + x) <= 1 + x / 2)
// or, near the surface (d0 <= rho),  |q - v0| <= d0 + rho  with g = 0.  Hence for every query of the cube
//     rr(q) - g.e  <=  res := (d0 + min-form slack + 3e-3 R) (1 + 2^-15) + 2^-15 reach,
// and a group whose box lies in the half space {x : n.(x - c) <= -f0} can be left out when
//     f0 - sum_i |n_i - g_i| h_i > res      (then f0 + n.e > rr(q) for every q of the cube: the walk's own
//                                            lane-level test, at thresholds no smaller than the final ones)
// — the packet-level test of the walk with the cube in the packet's place.  Everything left out is farther from
// every query of the cube than that query's final threshold, exactly what the walk guarantees of the groups it
// prunes itself; the waves' results cannot depend on whether they took the list or walked.
// A wave uses a list only if all its queries lie inside the cube's stored box (checked in the wave, by
// coordinates; the cube index is only a guess at where to look) and the list did not overflow.  Waves that
// straddle two cubes, outliers beyond the sampled bounds, cubes as wide as the mesh: the ordinary walk.
// =========================================================================================
// cube_side: the cube edge the host asks for (about 1.3 x the size of a packet group: lists stay short while a
// cube still holds many waves); level > 0 forces 2^level cubes per axis instead.  The launch's bounds are known on
// the device only, so the grid of cubes is laid out here, by every block alike; at most `max_cubes` (the launch's
// grid) cubes.  Blocks past the last cube leave.
__global__ __launch_bounds__(64) void k_cube_lists(MeshDev M, const uint32_t* __restrict__ qbounds, float cube_side,
                                                   uint32_t level, uint32_t max_cubes, float* __restrict__ par,
                                                   float4* __restrict__ hdr, uint32_t* __restrict__ list) {
    const uint32_t lane = threadIdx.x, cube = blockIdx.x;
    float lo3[3], ext3[3];
    uint32_t nc3[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo3[d] = pk_ord2f(qbounds[d]);
        ext3[d] = pk_ord2f(qbounds[3 + d]) - lo3[d];
        if (!(ext3[d] > 0.f && ext3[d] <= FLT_MAX)) ext3[d] = 0.f;  // (no span, or no finite point at all)
    }
    // The queries are sorted by the Morton code of a 256^3 grid over these bounds and a wave is a run of that order:
    // it stays inside one cube only if the cubes are aligned nodes of that octree — 2^L cubes per axis, the same L
    // for every axis (a cube then has the aspect of the bounds).  L: the largest extent over the wanted edge,
    // rounded to a power of two, and no more cubes than the launch has blocks.
    if (level == 0u) {
        const float emax = fmaxf(ext3[0], fmaxf(ext3[1], ext3[2]));
        const float want = cube_side > 0.f ? emax / cube_side : 1.f;
        level = 1u;
        while (level < 8u && (float)(1u << level) * 1.41421356f < want) ++level;  // (nearest power of two, log scale)
    }
    while (level > 1u && (1ull << (3u * level)) > (unsigned long long)max_cubes) --level;
    nc3[0] = nc3[1] = nc3[2] = 1u << level;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (ext3[d] == 0.f) nc3[d] = 1u;
    if (cube == 0u && lane == 0u) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            par[d] = lo3[d];
            par[3 + d] = ext3[d] > 0.f ? (float)nc3[d] / ext3[d] : 0.f;
            par[6 + d] = __uint_as_float(nc3[d]);
        }
    }
    if (cube >= nc3[0] * nc3[1] * nc3[2]) return;  // block-uniform
    const uint32_t ci[3] = {cube / (nc3[1] * nc3[2]), (cube / nc3[2]) % nc3[1], cube % nc3[2]};
    float c[3], h[3];
    bool ok = M.kd_ngroups != 0u;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float w = ext3[d] / (float)nc3[d];
        c[d] = lo3[d] + ((float)ci[d] + 0.5f) * w;
        // the cell's half width, a little wider: a query the waves' index arithmetic puts in this cube lies in its box
        h[d] = 0.5f * w * 1.001f + 1e-6f * fabsf(c[d]);
        ok = ok && fabsf(c[d]) <= FLT_MAX && h[d] <= FLT_MAX;
    }
    float4* H = hdr + 3u * (size_t)cube;
    uint32_t* L = list + (size_t)CUBE_CAP * cube;
    const float rho = __builtin_amdgcn_sqrtf(__builtin_fmaf(h[0], h[0], __builtin_fmaf(h[1], h[1], h[2] * h[2]))) * (1.f + PK_EPS);
    ok = ok && rho <= 0.5f * M.obb_rm;  // (a cube as wide as the mesh: every group is a candidate)
    if (!ok) {  // block-uniform
        if (lane == 0u) H[0] = make_float4(0.f, 0.f, 0.f, __uint_as_float(CUBE_NONE));
        return;
    }
    const float cx = c[0], cy = c[1], cz = c[2];
    // half extents as the tests see them (margin for the dot products, as in packet_walk)
    const float hx = h[0] * (1.f + 8.f * PK_EPS), hy = h[1] * (1.f + 8.f * PK_EPS), hz = h[2] * (1.f + 8.f * PK_EPS);
    const float reach = ((fabsf(cx - M.obb_cm[0]) + fabsf(cy - M.obb_cm[1])) + fabsf(cz - M.obb_cm[2])) + (M.obb_rm + rho);
    const float Ew = 3.8146973e-6f * reach;
    const uint32_t nrows = (M.kd_nl1 + 63u) >> 6;
    // ---- G0: the nearest L1 node to c, its nearest group (the greedy descent of packet_walk)
    uint32_t j0, g0, ref0;
    {
        uint32_t bkey = 0xFFFFFFFFu, brow = 0u;
        for (uint32_t row = 0; row < nrows; ++row) {
            const ObbLane o = obb_wide<false>(M.kd_l1obb + 256u * (size_t)row, lane, cx, cy, cz, Ew);
            const uint32_t key = obb_key(o.dP, 64u * row + lane < M.kd_nl1);
            if (key < bkey) bkey = key, brow = row;
        }
        const uint32_t mkey = wave_umin(bkey);
        const uint32_t who = (uint32_t)__builtin_ctzll(__ballot(bkey == mkey));
        j0 = 64u * (uint32_t)__builtin_amdgcn_readlane((int)brow, (int)who) + who;
        const uint32_t cnt = min(64u, M.kd_ngroups - 64u * j0);
        const ObbLane o = obb_wide<false>(M.kd_gobb + 256u * (size_t)j0, lane, cx, cy, cz, Ew);
        const uint32_t key = obb_key(o.dP, lane < cnt);
        const uint32_t mk = wave_umin(key);
        g0 = (uint32_t)__builtin_ctzll(__ballot(key == mk));
        ref0 = (uint32_t)__builtin_amdgcn_readlane((int)o.tag, (int)g0);
    }
    // ---- v0: the vertex of G0 nearest to c (lane k looks at slot k: a group has at most 60 slots)
    float d0, gx, gy, gz;
    {
        const uint32_t first = ((ref0 & sdfk::KD_REF_MASK) >> 4) << 2, nslots = 4u * (ref0 & 15u);
        const float4 P = M.kd_pts4[first + (lane < nslots ? lane : 0u)];
        const float dx = cx - P.x, dy = cy - P.y, dz = cz - P.z;
        float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
        if (!(lane < nslots) || !(d2 <= FLT_MAX)) d2 = __builtin_inff();  // (sentinel slots lie at 3e38)
        const uint32_t mk = wave_umin(__float_as_uint(d2));  // (non-negative floats order like their bits)
        const uint32_t who = (uint32_t)__builtin_ctzll(__ballot(__float_as_uint(d2) == mk));
        const float d2min = __uint_as_float(mk);
        if (!(d2min <= FLT_MAX)) {  // wave-uniform: no finite vertex there
            if (lane == 0u) H[0] = make_float4(0.f, 0.f, 0.f, __uint_as_float(CUBE_NONE));
            return;
        }
        // an upper bound of |c - v0| (the fp32 d2 is within a few ulp: 2^-18 relative + a denormal-proof absolute term)
        d0 = __builtin_amdgcn_sqrtf(d2min) * (1.f + PK_EPS) + 1e-30f;
        const float wx = wave_readf(dx, who), wy = wave_readf(dy, who), wz = wave_readf(dz, who);
        const bool trend = d0 > rho;  // far enough for the linear trend to pay
        const float inv = trend ? 1.f / d0 : 0.f;
        gx = wx * inv, gy = wy * inv, gz = wz * inv;
        // The approximate walk's threshold after G0: (est + E(est)) cm with est <= d^2 + E_d, E_d <= u (14 d^2 + 21 R^2):
        // at most d^2 (1 + 50 u) + 80 u R^2, i.e. d (1 + 25 u) + sqrt(80 u) R in distance — the relative part is far
        // inside the (1 + 2^-15) below, sqrt(80 u) R = 2.2e-3 R is added (3e-3 sqrt(apx_e0 / (24 u)) = 3e-3 R).
        const float slackR = 3e-3f * __builtin_amdgcn_sqrtf(M.apx_e0 * (1.f / (24.f * 5.9604645e-8f)));
        const float slack = trend ? (rho * rho) * (0.5f * inv) : rho;
        d0 = ((d0 + slack) + slackR) * (1.f + 3.0517578e-5f) + 3.0517578e-5f * reach;
    }
    const float res = d0;  // >= rr(q) - g.e of every query of the cube once G0 is evaluated
    // ---- every group the cube-level test cannot exclude, in tree order (slot 0 is G0's)
    uint32_t cnt = 1u;
    const unsigned long long lt = lane == 0u ? 0ull : (~0ull >> (64u - lane));
    for (uint32_t row = 0; row < nrows; ++row) {
        const ObbLane o1 = obb_wide<true>(M.kd_l1obb + 256u * (size_t)row, lane, cx, cy, cz, Ew, gx, gy, gz, hx, hy, hz);
        unsigned long long m1 = __ballot(64u * row + lane < M.kd_nl1 && !(o1.pk > res));
        while (m1) {
            const uint32_t j = 64u * row + (uint32_t)__builtin_ctzll(m1);
            m1 &= m1 - 1ull;
            const uint32_t ng = min(64u, M.kd_ngroups - 64u * j);
            const ObbLane o = obb_wide<true>(M.kd_gobb + 256u * (size_t)j, lane, cx, cy, cz, Ew, gx, gy, gz, hx, hy, hz);
            const bool keep = lane < ng && !(o.pk > res) && !(j == j0 && lane == g0);
            const unsigned long long m = __ballot(keep);
            const uint32_t at = cnt + (uint32_t)__popcll(m & lt);
            if (keep && at < CUBE_CAP) L[at] = 64u * j + lane;
            cnt += (uint32_t)__popcll(m);
        }
    }
    if (lane == 0u) {
        L[0] = 64u * j0 + g0;
        H[0] = make_float4(cx, cy, cz, __uint_as_float(cnt <= CUBE_CAP ? cnt : CUBE_NONE));
        H[1] = make_float4(h[0], h[1], h[2], 0.f);
        H[2] = make_float4(gx, gy, gz, 0.f);
    }
}

// exact evaluation of one packet group for one lane (second phase of the approximate pass): the
// reference's metric, strict `<`, and among exact ties the vertex nanoflann's DFS meets first.
// TIES = false leaves the tie rule out (pos = the first vertex met at the minimum): valid whenever the
// result has s2 > b, i.e. the minimum is unique; the caller redoes the lane with TIES = true otherwise.
template <bool TIES>
__device__ __forceinline__ void exact_group(const MeshDev& M, float qx, float qy, float qz, uint32_t ref,
                                            uint32_t& b, uint32_t& s2, uint32_t& pos) {
    const uint32_t first = ((ref & sdfk::KD_REF_MASK) >> 4) << 2, nch = ref & 15u;
    for (uint32_t c = 0; c < nch; ++c) {  // a chunk at a time: its four loads are in flight together
        const float4* P = M.kd_pts4 + first + 4u * c;
        const float4 P0 = P[0], P1 = P[1], P2 = P[2], P3 = P[3];
        uint32_t du[4];
#define SDFK_EX_D(J, V)                                                        \
    {                                                                          \
        const float dx = qx - V.x, dy = qy - V.y, dz = qz - V.z;               \
        du[J] = __float_as_uint((dx * dx + dy * dy) + dz * dz);                \
    }
        SDFK_EX_D(0, P0) SDFK_EX_D(1, P1) SDFK_EX_D(2, P2) SDFK_EX_D(3, P3)
#undef SDFK_EX_D
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t idx = first + 4u * c + j;
            if (TIES && __builtin_expect(du[j] == b && pos != H_NONE, 0)) {
                if ((uint32_t)kd_dfs_first(M, qx, qy, qz, (int)pos, (int)idx) == idx) pos = idx;
            }
            s2 = min(s2, max(du[j], b));
            if (du[j] < b) {
                b = du[j];
                pos = idx;
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

// The path check proper: true when none of nanoflann's tests on the way to `pos` can fail while `worst`
// is >= s.  First a sufficient test without the walk.  Every plane on the path separates the query from
// the leaf of `pos` in its dimension (far descents only; lowside <=> val < divhigh, else val >= divlow, by
// monotone rounding), and so does the root box's side where the query lies outside it: with t_d = the
// distance from q_d to the leaf's tight interval, each dists[d] on the path is at most t_d^2 in exact
// arithmetic, exactly as for a vertex of the leaf, and the bound of the window test carries over with the
// box's nearest point in the vertex's place: m2 <= BD (1 + w_th), BD = sum t_d^2.  box_lb evaluates BD
// within 5.01u, the product rounds within u: bd * c^2 <= s (c^2 = amb_c2 >= 1 + w_th + 7u) implies every
// m2 <= s.  bd == 0: the query is inside the box, every cut on the path is exactly 0.  Tiny bd: relative
// bounds are void, walk.  The box is nearer than the vertex by about (leaf size) x (distance), the window
// is 1e-5 relative: the walk (20 dependent loads, ~450 VALU of the wave for 2 lanes in 64) is rare.
__device__ __forceinline__ bool kd_path_clear(const MeshDev& M, float qx, float qy, float qz, int pos, float s) {
    const float4 A = M.kd_lbox[2 * (size_t)pos], B = M.kd_lbox[2 * (size_t)pos + 1];
    const float bd = box_lb(qx, qy, qz, A.x, A.y, A.z, A.w, B.x, B.y);
    const bool quick = bd == 0.f || (bd >= 1e-30f && bd * M.amb_c2 <= s);
#ifdef SDFK_STATS
    const bool full = kd_path_max_bound(M, qx, qy, qz, pos) <= s;
    if (quick) atomicAdd(&g_pk_stats[8], 1ull);
    if (quick && !full) atomicAdd(&g_pk_stats[9], 1ull);
    if (!quick) atomicAdd(&g_pk_stats[10], 1ull);
    return full;
#else
    if (quick) return true;
    return kd_path_max_bound(M, qx, qy, qz, pos) <= s;
#endif
}

// every valid query of this wave goes to the heavy list (absolute indices); wave-uniform
__device__ __forceinline__ void packet_to_heavy(bool valid, int64_t i, int64_t src, uint32_t off,
                                                unsigned long long* __restrict__ amb_mask, uint32_t* __restrict__ amb_cnt,
                                                uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt) {
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
}

// packet_resolve for a walk that left exact ties alone (group_exact_lazy): among the vertices of the holder group at
// distance exactly b, the one nanoflann's DFS meets first (fast_nn: the lowest position, which packet_resolve gives)
__device__ __forceinline__ int packet_resolve_ties(const MeshDev& M, float qx, float qy, float qz, float bf, uint32_t h) {
    if (h == H_NONE) return -1;
    if (!(h & KD_LEAF)) return (int)h;
    const uint32_t first = ((h & sdfk::KD_REF_MASK) >> 4) << 2, n = 4u * (h & 15u);
    const uint32_t b = __float_as_uint(bf);
    int pos = -1;
    for (uint32_t k = 0; k < n; ++k) {
        const float4 P = M.kd_pts4[first + k];
        const float dx = qx - P.x, dy = qy - P.y, dz = qz - P.z;
        if (__float_as_uint((dx * dx + dy * dy) + dz * dz) == b)
            pos = pos < 0 ? (int)(first + k) : kd_dfs_first(M, qx, qy, qz, pos, (int)(first + k));
    }
    return pos;
}

// What follows the walk, per lane: the exact second phase (APPROX), the decision whether the walk's answer
// is nanoflann's, and the result (nearest vertex / incident-face distances / sign).
template <int KIND, bool ROBUST, bool APPROX>
__device__ __forceinline__ void packet_finish(const MeshDev& M, V3 p, bool valid, int64_t i, int64_t src, float sign,
                                              PacketBest pb, const ApproxState& A, int trunc_aabb,
                                              void* __restrict__ out, unsigned long long* __restrict__ amb_mask,
                                              uint32_t* __restrict__ amb_cnt, uint32_t off,
                                              uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt) {
    // pb.h is nanoflann's answer iff nanoflann does not prune it before visiting it:
    //  (1) no other vertex within b * c (c - 1 ~ 4.5e-6): impossible, see k_query_packet header;
    //  (2) otherwise measure it: every bound tested on the way to pb.h is <= the distance of
    //      any other vertex (>= pb.s; == b for an exact tie, where pb.h is the first met);
    //  (3) else defer to the strict kernel (exact emulation).
    int pos;
    bool amb = false;
    if (APPROX) {
        // second phase: the groups whose minimum could be the nearest vertex, exactly
        const float U = apx_upper(M, A.B1);  // >= D of B1's vertex >= the exact minimum
        const float L2 = apx_lower(M, A.B2), L3 = apx_lower(M, A.B3), L4 = apx_lower(M, A.B4);  // <= D outside
        // (Widening these tests to the window of the near-tie test below, so that a group's lower bound never
        // stands in for `s` there, was measured: near-tie lanes 2.28 -> 2.18 per wave, step 22.4 -> 23.0 ms.  The
        // near ties are real: far-field squared distances of neighbouring vertices differ by ~1e-5 relative.)
        const bool n2 = !(L2 > U), n3 = n2 && !(L3 > U), n4 = n3 && !(L4 > U);
#ifdef SDFK_STATS
        if (valid && n4) atomicAdd(&g_pk_stats[7], 1ull);  // (APPROX: a fourth candidate group)
#endif
        uint32_t b = 0x7F7FFFFFu, s2 = 0x7F7FFFFFu, hp = H_NONE;
        if (A.H1 != H_NONE && !n4) {
            exact_group<false>(M, p.x, p.y, p.z, A.H1, b, s2, hp);
            if (n2) exact_group<false>(M, p.x, p.y, p.z, A.H2, b, s2, hp);
            if (n3) exact_group<false>(M, p.x, p.y, p.z, A.H3, b, s2, hp);
            if (__builtin_expect(s2 == b && hp != H_NONE, 0)) {  // an exact tie at the minimum: again, with the tie rule
                b = 0x7F7FFFFFu, s2 = 0x7F7FFFFFu, hp = H_NONE;
                exact_group<true>(M, p.x, p.y, p.z, A.H1, b, s2, hp);
                if (n2) exact_group<true>(M, p.x, p.y, p.z, A.H2, b, s2, hp);
                if (n3) exact_group<true>(M, p.x, p.y, p.z, A.H3, b, s2, hp);
            }
        }
        const float s_out = n3 ? L4 : (n2 ? L3 : L2);  // every vertex outside the evaluated groups is farther
        pb.b = __uint_as_float(b);
        pb.s = fminf(__uint_as_float(s2), s_out);
        pos = hp == H_NONE ? -1 : (int)hp;  // (-1: nothing finite, or a fourth candidate group: deferred below)
    } else {
        pos = (pb.lazy_ties && pb.s == pb.b && !M.fast) ? packet_resolve_ties(M, p.x, p.y, p.z, pb.b, pb.h)
                                                         : packet_resolve(M, p.x, p.y, p.z, pb.b, pb.h);
    }
    if (!M.fast) {
        if (valid && (!(pb.s > pb.b * M.amb_c) || !(pb.b >= 1e-30f))) {
            amb = true;
#ifdef SDFK_STATS
            atomicAdd(&g_pk_stats[6], 1ull);
#endif
            if (pos >= 0 && pb.b < FLT_MAX) {
                amb = !kd_path_clear(M, p.x, p.y, p.z, pos, pb.s);
            }
        }
        if (pos < 0) amb = true;
    }
    if (APPROX) {
        // what the approximate pass cannot settle (a fourth candidate group, no finite distance, a
        // window the path check does not clear) goes to the exact scan of k_query_heavy, which
        // takes the same decision with exact (b, s, h) and defers to the strict kernel itself.
        // (Straight to the strict kernel's list instead, tried on a 1M-vertex mesh where the scan is a pass
        // over 1.3M positions per query: 18k queries took 35 ms there against 16 ms in the scan.)
        if (valid && amb) heavy_list[atomicAdd(heavy_cnt, 1u)] = off + (uint32_t)src;
        if ((threadIdx.x & 63) == 0) amb_mask[i >> 6] = 0ull, amb_cnt[i >> 6] = 0u;
    } else {   // ambiguous lanes of this wave, in sorted order: compacted (stably) by k_fb_compact so
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
        SDFK_NT_STORE(pos >= 0 ? M.kd_vind[pos] : -1, &((int32_t*)out)[src]);  // (-1: fast mode, no finite distance)
    } else {
        const bool trunc = (KIND == Q_CONTAINS) ? true : (trunc_aabb != 0);
        const float v = calc_tail<ROBUST>(M, p, trunc, sign, pos);
        if (KIND == Q_CONTAINS)
            ((uint8_t*)out)[src] = v >= 0;
        else
            SDFK_NT_STORE(v, &((float*)out)[src]);
    }
}

// Hand-over to the cooperative kernel: packet index + the per-lane lists of the walk so far.
// The main kernel must not pay for it in registers (two more VGPRs took it from 7 to 6 waves per SIMD and
// the 125M-point step from 23.5 to 24.5 ms): it gets no argument for it.  The switch is the low bit of
// M.heavy_budget (odd: hand over to the cooperative kernel at that budget; the host sets it per launch),
// the counter is the word after the heavy counter and the two pointers sit in device memory behind
// it (CoopDesc at heavy_cnt + 2, written once by the host), read only by the waves that hand over.
// ---- far packets (see k_query_packet_far below) ----
// distance shell of a query: 0 = not far (or not finite), else 1 + floor(log2(D / R_far)) capped at 7
__device__ __forceinline__ uint32_t far_shell(const float c[3], float r2, float x, float y, float z) {
    const float dx = x - c[0], dy = y - c[1], dz = z - c[2];
    const float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    if (!(d2 > r2) || !(d2 <= FLT_MAX) || !(r2 > 0.f)) return 0u;
    const uint32_t e = ((__float_as_uint(d2 / r2) >> 23) & 255u) - 127u;  // floor(log2(D^2 / R^2)) >= 0
    return 1u + min(e >> 1, 6u);
}
// wave-uniform: packet `pkt` of a launch of n queries lies in the far part of the sorted order as the key kernel
// counted it (none, when the key kernel saw no reason to sort far queries apart), within the far kernel's range, and
// every lane's query is far and of the same shell (lanes past the end repeat the last query)
__device__ __forceinline__ bool far_packet(const MeshDev& M, float x, float y, float z, int64_t pkt, int64_t n) {
    if (pkt < (int64_t)M.far_lo || 64 * pkt < n - (int64_t)*M.far_cnt) return false;
    const uint32_t sh = far_shell(M.far_c, __uint_as_float(M.far_cnt[1]), x, y, z);  // (the radius the key kernel chose)
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh);
    return s0 != 0u && __ballot(sh != s0) == 0ull;
}
#ifndef SDFK_COOP_WAVES
#define SDFK_COOP_WAVES 8
#endif
constexpr int COOP_WAVES = SDFK_COOP_WAVES;
constexpr uint32_t COOP_CAP = 32768;  // packets one launch can hand over (56 MiB of lists); the packets beyond go to the exact scan
struct CoopDesc {
    uint32_t* list;   // [COOP_CAP] packet (wave) index within the launch
    uint32_t* state;  // [COOP_CAP][7][64]: B1..B4 (float bits), H1..H3
};

template <int KIND, bool ROBUST, bool APPROX>
__global__ __launch_bounds__(PBLOCK) SDFK_PK_SGPR void k_query_packet(MeshDev M, const float* __restrict__ pts,
                                                         const uint32_t* __restrict__ perm, int64_t n,
                                                         int trunc_aabb, void* __restrict__ out,
                                                         unsigned long long* __restrict__ amb_mask,
                                                         uint32_t* __restrict__ amb_cnt,
                                                         uint32_t off,
                                                         uint32_t* __restrict__ heavy_list,
                                                         uint32_t* __restrict__ heavy_cnt) {
    // (One packet per wave.  A loop over consecutive packets — to fetch the next packet's points while this one
    // walks — was tried once the chunk loads were in the constant address space: the compiler then hoists every
    // kernel-argument load and every mesh-derived constant out of the loop and keeps them live, 91 VGPRs
    // instead of 72 with the scalar registers spilled into vector lanes.)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    // lanes past the end of the batch repeat its last query (and write nothing): they then ask
    // for no subtree a real lane does not
    const int64_t ic = valid ? i : n - 1;
    // the point stream (and the results) pass through once: non-temporal, so that L2 keeps the trees
    const int64_t src = perm ? (int64_t)SDFK_NT_LOAD(&perm[ic]) : ic;
    const V3 p = {SDFK_NT_LOAD(&pts[3 * src]), SDFK_NT_LOAD(&pts[3 * src + 1]), SDFK_NT_LOAD(&pts[3 * src + 2])};
    // (far packets of the covered range: k_query_packet_far walks them, launched behind this kernel; never set for
    // robust contains)
    if (M.far_r2 > 0.f && far_packet(M, p.x, p.y, p.z, i >> 6, n)) return;
    float sign = 0.f;
    if (ROBUST && KIND != Q_NN) sign = raycast_packet(M, p, valid);  // sdf.cpp:110
    if (ROBUST && KIND == Q_CONTAINS) {  // sdf.cpp:163-169: no nearest-vertex search at all
        if (valid) ((uint8_t*)out)[src] = sign >= 0.0f;
        if ((threadIdx.x & 63) == 0) amb_mask[i >> 6] = 0ull, amb_cnt[i >> 6] = 0u;
        return;
    }
    PacketBest pb;
    ExactState SE = {0x7F7FFFFFu, 0x7F7FFFFFu, H_NONE, 0xFFFFFFFFu};  // FLT_MAX bits (KNNResultSet::init)
    const float INF = __builtin_inff();
    ApproxState A = {INF, INF, INF, INF, H_NONE, H_NONE, H_NONE};
    uint32_t wr;
    if (APPROX) {
        wr = packet_walk<true, false>(M, p.x, p.y, p.z, M.amb_c2, SE, A, 0u, 1u, M.heavy_budget);
    } else {
        wr = packet_walk<false, false>(M, p.x, p.y, p.z, 1.f, SE, A, 0u, 1u, M.heavy_budget);
        if (SE.b == SE.tieb) SE.s2 = SE.b;  // a settled exact tie is a tie
        pb.b = __uint_as_float(SE.b), pb.s = __uint_as_float(SE.s2), pb.h = SE.h;
    }
    pb.bailed = wr == 1u;
    if (wr == 2u) {  // (wave-uniform) a packet as wide as the mesh
        packet_to_heavy(valid, i, src, off, amb_mask, amb_cnt, heavy_list, heavy_cnt);
        return;
    }
    if (pb.bailed) {  // wave-uniform
        uint32_t slot = 0xFFFFFFFFu;
        if (APPROX && (M.heavy_budget & 1u)) {
            if ((threadIdx.x & 63) == 0) slot = atomicAdd(heavy_cnt + 1, 1u);
            slot = __builtin_amdgcn_readfirstlane(slot);
        }
        if (slot < COOP_CAP) {  // the walk goes on in k_query_packet_coop, from these lists
            const CoopDesc* D = (const CoopDesc*)(heavy_cnt + 2);
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t* st = D->state + (size_t)slot * 448u + lane;
            st[0] = __float_as_uint(A.B1), st[64] = __float_as_uint(A.B2), st[128] = __float_as_uint(A.B3);
            st[192] = __float_as_uint(A.B4), st[256] = A.H1, st[320] = A.H2, st[384] = A.H3;
            if (lane == 0) {
                D->list[slot] = (uint32_t)(i >> 6);
                amb_mask[i >> 6] = 0ull, amb_cnt[i >> 6] = 0u;
            }
        } else {
            packet_to_heavy(valid, i, src, off, amb_mask, amb_cnt, heavy_list, heavy_cnt);
        }
        return;
    }
    packet_finish<KIND, ROBUST, APPROX>(M, p, valid, i, src, sign, pb, A, trunc_aabb, out, amb_mask, amb_cnt, off,
                                        heavy_list, heavy_cnt);
}

// =========================================================================================
// Far packets.  A batch spread over a region much larger than the mesh is sparse in space — 64 Morton-adjacent
// queries of 8M in a box 300 x the mesh span several mesh diameters, every group is a candidate for some lane and
// the walk above is useless (such packets went to the strict kernel, one tree walk per lane: 6 ns per query) — but
// dense in DIRECTION: the same 64 queries, sorted by the direction they see the mesh from, lie within 0.01 rad, and
// what a far query can have nearest is the cap of the mesh facing it.  The key kernel (k_query_morton) therefore
// sorts far queries (farther than sqrt(M.far_r2) from the mesh centre) behind the near ones, by distance shell
// (powers of two) and direction; waves whose queries are all far and of one shell are left alone by the main
// kernel and walked here, with bounds made for them:
//   for ANY origin o and the unit vector d_l from o to q_l:   |q_l - x|  >=  d_l.(q_l - x)  =  D_l - d_l.(x - o),
//   D_l = |q_l - o|, so a box B can be skipped for lane l when  D_l - sup_B(d_l) > rr_l  (rr_l: the lane's
//   threshold as a distance), with sup_B(d) = max_{x in B} d.(x - o) <= d.c' + sum_a |d.R_a| h_a  (c' = c_B - o).
// o is the centre of the first evaluated group's box (then rr_l - D_l is within the cap's depth of 0 for every
// lane, however far the queries are and however they are spread along their common direction), and with the
// packet's mean direction a, delta_l = |d_l - a| <= theta:
//   lane level     -(d_l.c' + supH + delta_l h1) > rr_l - D_l        supH = sum_a |a.R_a| h_a, h1 = sum_a h_a
//   packet level   -(a.c' + supH) - theta (|c'| + h1) > max_l (rr_l - D_l)
// — five readlanes and five VALU per candidate.  Everything else (approximate evaluation, the four group minima,
// second phase, near-tie decision) is the main kernel's.  Rounding: d_l is a unit vector within 4u, D_l is
// rounded down and rr_l up, projections carry <= 8u (|c'| + h1): slack 2^-19 (|c'| + h1) per test.
// =========================================================================================
struct FarLane {
    float cx, cy, cz;  // c' = c_B - o
    float supH, h1;    // sum_a |a.R_a| h_a, sum_a h_a
    float sl;          // slack of one test: 2^-19 (|c'| + h1)
    float pk;          // packet-level bound
    uint32_t tag;
};
__device__ __forceinline__ FarLane far_wide(const float4* __restrict__ rec, uint32_t lane, float ox, float oy, float oz,
                                            float ax, float ay, float az, float theta) {
    const float4 A = rec[lane], B = rec[64u + lane], C = rec[128u + lane], D = rec[192u + lane];
    FarLane o;
    o.cx = A.x - ox, o.cy = A.y - oy, o.cz = A.z - oz;
    const float p0 = fabsf(__builtin_fmaf(B.x, ax, __builtin_fmaf(B.y, ay, B.z * az)));
    const float p1 = fabsf(__builtin_fmaf(C.x, ax, __builtin_fmaf(C.y, ay, C.z * az)));
    const float p2 = fabsf(__builtin_fmaf(D.x, ax, __builtin_fmaf(D.y, ay, D.z * az)));
    o.h1 = (A.w + B.w) + C.w;
    o.supH = __builtin_fmaf(A.w, p0, __builtin_fmaf(B.w, p1, C.w * p2));
    const float cn = __builtin_amdgcn_sqrtf(__builtin_fmaf(o.cx, o.cx, __builtin_fmaf(o.cy, o.cy, o.cz * o.cz))) + o.h1;
    const float sa = __builtin_fmaf(ax, o.cx, __builtin_fmaf(ay, o.cy, az * o.cz));
    o.sl = 1.9073486e-6f * cn;
    o.pk = -(sa + o.supH) - __builtin_fmaf(theta, cn, o.sl);
    o.tag = __float_as_uint(D.w);
    return o;
}
// the packet's frame about the origin o: per lane D (rounded down), unit direction d, delta = |d - a| (rounded up);
// wave-uniform: the mean direction a and theta = max delta
struct FarFrame {
    float D, dx, dy, dz, delta;
    float ax, ay, az, theta;
};
__device__ __forceinline__ FarFrame far_frame(float qx, float qy, float qz, float ox, float oy, float oz) {
    FarFrame F;
    const float tx = qx - ox, ty = qy - oy, tz = qz - oz;
    const float D = __builtin_sqrtf(__builtin_fmaf(tx, tx, __builtin_fmaf(ty, ty, tz * tz)));
    const float inv = 1.f / D;
    F.dx = tx * inv, F.dy = ty * inv, F.dz = tz * inv;
    F.D = D - D * 2.3841858e-7f;
    const float mx = 0.5f * (pk_ord2f(wave_umin(pk_f2ord(F.dx))) + pk_ord2f(wave_umax(pk_f2ord(F.dx))));
    const float my = 0.5f * (pk_ord2f(wave_umin(pk_f2ord(F.dy))) + pk_ord2f(wave_umax(pk_f2ord(F.dy))));
    const float mz = 0.5f * (pk_ord2f(wave_umin(pk_f2ord(F.dz))) + pk_ord2f(wave_umax(pk_f2ord(F.dz))));
    const float mi = 1.f / __builtin_sqrtf(__builtin_fmaf(mx, mx, __builtin_fmaf(my, my, mz * mz)));
    F.ax = mx * mi, F.ay = my * mi, F.az = mz * mi;
    if (!(fabsf(F.ax) <= 2.f && fabsf(F.ay) <= 2.f && fabsf(F.az) <= 2.f)) F.ax = 1.f, F.ay = 0.f, F.az = 0.f;  // (opposite directions: any axis)
    const float ex = F.dx - F.ax, ey = F.dy - F.ay, ez = F.dz - F.az;
    F.delta = __builtin_sqrtf(__builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez))) * (1.f + 4.7683716e-7f) + 2.3841858e-7f;
    F.theta = __uint_as_float(wave_umax(__float_as_uint(F.delta)));  // (non-negative; NaN -> large)
    return F;
}

// the walk of a far packet (approximate pass); returns 1 when the wave gave up on its work budget
template <bool APPROX>
__device__ __forceinline__ uint32_t packet_walk_far(const MeshDev& M, float qx, float qy, float qz, float cm,
                                                ExactState& SE, ApproxState& SA, uint32_t budget) {
    const uint32_t lane = threadIdx.x & 63u;
    SDFK_COUNT(0, 1);
    if (M.kd_ngroups == 0u) return 0u;
    const uint32_t nrows = (M.kd_nl1 + 63u) >> 6;
    uint32_t work = 0;
    // ---- greedy: seen from the mesh centre, the L1 node and then the group reaching farthest towards the packet
    uint32_t j0, g0;
    float ox = M.far_c[0], oy = M.far_c[1], oz = M.far_c[2];
    {
        const FarFrame F0 = far_frame(qx, qy, qz, ox, oy, oz);
        uint32_t bkey = 0xFFFFFFFFu, brow = 0u;
        for (uint32_t row = 0; row < nrows; ++row) {
            const FarLane o = far_wide(M.kd_l1obb + 256u * (size_t)row, lane, ox, oy, oz, F0.ax, F0.ay, F0.az, 0.f);
            const uint32_t key = 64u * row + lane < M.kd_nl1 ? (o.pk == o.pk ? pk_f2ord(o.pk) : 0u) : 0xFFFFFFFFu;
            if (key < bkey) bkey = key, brow = row;
        }
        SDFK_COUNT(1, nrows);
        const uint32_t mkey = wave_umin(bkey);
        const uint32_t who = (uint32_t)__builtin_ctzll(__ballot(bkey == mkey));
        j0 = 64u * (uint32_t)__builtin_amdgcn_readlane((int)brow, (int)who) + who;
        const uint32_t cnt = min(64u, M.kd_ngroups - 64u * j0);
        const FarLane o = far_wide(M.kd_gobb + 256u * (size_t)j0, lane, ox, oy, oz, F0.ax, F0.ay, F0.az, 0.f);
        SDFK_COUNT(1, 1);
        const uint32_t key = lane < cnt ? (o.pk == o.pk ? pk_f2ord(o.pk) : 0u) : 0xFFFFFFFFu;
        const uint32_t mk = wave_umin(key);
        g0 = (uint32_t)__builtin_ctzll(__ballot(key == mk));
        const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)o.tag, (int)g0);
        SDFK_COUNT(2, 1);
        SDFK_COUNT(3, ref & 15u);
        work += ref & 15u;
        if (APPROX)
            group_approx<false>(M, qx, qy, qz, ref, SA);
        else
            group_exact_lazy(M, qx, qy, qz, ref, SE);
        // the origin of everything below: the centre of that group's box (a box that contains everything has centre 0)
        ox = wave_readf(o.cx, g0) + ox, oy = wave_readf(o.cy, g0) + oy, oz = wave_readf(o.cz, g0) + oz;
        if (!(fabsf(ox) <= FLT_MAX && fabsf(oy) <= FLT_MAX && fabsf(oz) <= FLT_MAX)) ox = M.far_c[0], oy = M.far_c[1], oz = M.far_c[2];
    }
    const FarFrame F = far_frame(qx, qy, qz, ox, oy, oz);
    float marg;  // per lane: rr_l - D_l (threshold distance, rounded up, minus the distance to o, rounded down)
#define SDFK_FAR_REFRESH() \
    marg = __builtin_amdgcn_sqrtf(APPROX ? apx_upper(M, SA.B1) * cm : __uint_as_float(SE.b) * cm) * (1.f + 4.7683716e-7f) - F.D;
#define SDFK_FAR_RES() pk_ord2f(wave_umax(pk_f2ord(marg)))
    SDFK_FAR_REFRESH()
    float res = SDFK_FAR_RES();
    // candidates of one wide test, deepest towards the packet first
#define SDFK_FAR_CANDS(O, ALIVE)                                                                                  \
    for (;;) {                                                                                                    \
        const uint32_t key_ = (ALIVE) ? ((O).pk == (O).pk ? pk_f2ord((O).pk) : 0u) : 0xFFFFFFFFu;                \
        const uint32_t mk_ = wave_umin(key_);                                                                     \
        if (mk_ == 0xFFFFFFFFu || (mk_ != 0u && pk_ord2f(mk_) > res)) break;                                      \
        const uint32_t g_ = (uint32_t)__builtin_ctzll(__ballot(key_ == mk_));                                     \
        if (lane == g_) ALIVE = false;                                                                            \
        SDFK_COUNT(5, 1);                                                                                         \
        ++work;                                                                                                   \
        const float cn_ = wave_readf((O).h1, g_);                                                                 \
        const float t_ = -(__builtin_fmaf(F.dx, wave_readf((O).cx, g_),                                           \
                                          __builtin_fmaf(F.dy, wave_readf((O).cy, g_), F.dz * wave_readf((O).cz, g_))) + \
                           __builtin_fmaf(F.delta, cn_, wave_readf((O).supH, g_)));                               \
        if (__ballot(!(t_ - wave_readf((O).sl, g_) > marg)) == 0ull) continue;                                    \
        const uint32_t ref_ = (uint32_t)__builtin_amdgcn_readlane((int)(O).tag, (int)g_);                         \
        SDFK_COUNT(2, 1);                                                                                         \
        SDFK_COUNT(3, ref_ & 15u);                                                                                \
        work += ref_ & 15u;                                                                                       \
        {                                                                                                         \
            const uint32_t old_ = APPROX ? __float_as_uint(SA.B1) : SE.b;                                         \
            if (APPROX)                                                                                           \
                group_approx<false>(M, qx, qy, qz, ref_, SA);                                                     \
            else                                                                                                  \
                group_exact_lazy(M, qx, qy, qz, ref_, SE);                                                        \
            if (__ballot((APPROX ? __float_as_uint(SA.B1) : SE.b) != old_) != 0ull) {                             \
                SDFK_FAR_REFRESH()                                                                                \
                res = SDFK_FAR_RES();                                                                             \
            }                                                                                                     \
        }                                                                                                         \
        if (work > budget) return 1u;                                                                             \
    }
    for (uint32_t row = 0; row < nrows; ++row) {
        const FarLane o1 = far_wide(M.kd_l1obb + 256u * (size_t)row, lane, ox, oy, oz, F.ax, F.ay, F.az, F.theta);
        SDFK_COUNT(1, 1);
        bool alive1 = 64u * row + lane < M.kd_nl1 && !(o1.pk > res);
        for (;;) {
            const uint32_t key = alive1 ? (o1.pk == o1.pk ? pk_f2ord(o1.pk) : 0u) : 0xFFFFFFFFu;
            const uint32_t mk = wave_umin(key);
            if (mk == 0xFFFFFFFFu || (mk != 0u && pk_ord2f(mk) > res)) break;
            const uint32_t k = (uint32_t)__builtin_ctzll(__ballot(key == mk));
            if (lane == k) alive1 = false;
            const uint32_t j = 64u * row + k;
            const uint32_t cnt = min(64u, M.kd_ngroups - 64u * j);
            const FarLane o = far_wide(M.kd_gobb + 256u * (size_t)j, lane, ox, oy, oz, F.ax, F.ay, F.az, F.theta);
            SDFK_COUNT(1, 1);
            ++work;
            bool alive = lane < cnt && !(o.pk > res) && !(j == j0 && lane == g0);
            SDFK_FAR_CANDS(o, alive)
        }
    }
#undef SDFK_FAR_CANDS
#undef SDFK_FAR_RES
#undef SDFK_FAR_REFRESH
    return 0u;
}

// One far packet per wave, taken from the END of the sorted order (the far queries sort last): block b walks
// packet npk - 1 - b if that packet is a far packet, and leaves otherwise — as the main kernel leaves every far
// packet from M.far_lo on to this one.  Launched with min(npk, FAR_CAP) blocks behind the main kernel.
// far_cnt: the far queries of the launch as the key kernel counted them (they hold the last sorted positions): blocks
// whose packets lie before them leave at once — a launch without far queries pays one empty block per four packets.
constexpr uint32_t FAR_CAP = 1u << 21;  // packets (2^27 queries: the largest chunk of a call)
#ifndef SDFK_FAR_BLOCK
#define SDFK_FAR_BLOCK 256
#endif
constexpr int FAR_BLOCK = SDFK_FAR_BLOCK;  // (four packets per block; 64 / 128 / 256 walk alike, 1024 is 20 % slower: a block holds
                                           // its slots until its slowest wave is through; the empty blocks of a launch without
                                           // far queries cost 0.1 ms per 500k, i.e. per 125M points)
template <int KIND, bool ROBUST, bool APPROX>
__global__ __launch_bounds__(FAR_BLOCK) SDFK_PK_SGPR void k_query_packet_far(MeshDev M, const float* __restrict__ pts,
                                                             const uint32_t* __restrict__ perm, int64_t n,
                                                             int trunc_aabb, void* __restrict__ out,
                                                             unsigned long long* __restrict__ amb_mask,
                                                             uint32_t* __restrict__ amb_cnt, uint32_t off,
                                                             uint32_t* __restrict__ heavy_list,
                                                             uint32_t* __restrict__ heavy_cnt) {
    const int64_t npk = (n + 63) >> 6;
    const int64_t nfar = (int64_t)*M.far_cnt;
    if ((int64_t)blockIdx.x * (FAR_BLOCK / 64) * 64 >= nfar + 64) return;  // block-uniform (+ the packet straddling the boundary)
    const int64_t pkt = npk - 1 - ((int64_t)blockIdx.x * (FAR_BLOCK / 64) + (int64_t)(threadIdx.x >> 6));
    if (pkt < 0) return;
    const int64_t i = pkt * 64 + (int64_t)(threadIdx.x & 63u);
    const bool valid = i < n;
    const int64_t ic = valid ? i : n - 1;
    const int64_t src = (int64_t)SDFK_NT_LOAD(&perm[ic]);
    const V3 p = {SDFK_NT_LOAD(&pts[3 * src]), SDFK_NT_LOAD(&pts[3 * src + 1]), SDFK_NT_LOAD(&pts[3 * src + 2])};
    if (!far_packet(M, p.x, p.y, p.z, pkt, n)) return;  // wave-uniform: the main kernel has it
    float sign = 0.f;
    if (ROBUST && KIND != Q_NN) sign = raycast_packet(M, p, valid);  // sdf.cpp:110
    PacketBest pb;
    ExactState SE = {0x7F7FFFFFu, 0x7F7FFFFFu, H_NONE, 0xFFFFFFFFu};
    const float INF = __builtin_inff();
    ApproxState A = {INF, INF, INF, INF, H_NONE, H_NONE, H_NONE};
    const uint32_t wr = packet_walk_far<APPROX>(M, p.x, p.y, p.z, M.amb_c2, SE, A, M.heavy_budget);
    if (!APPROX) {
        pb.b = __uint_as_float(SE.b), pb.s = __uint_as_float(SE.s2), pb.h = SE.h;
        pb.lazy_ties = true;
    }
    pb.bailed = wr == 1u;
    if (pb.bailed) {
        packet_to_heavy(valid, i, src, off, amb_mask, amb_cnt, heavy_list, heavy_cnt);
        return;
    }
    packet_finish<KIND, ROBUST, APPROX>(M, p, valid, i, src, sign, pb, A, trunc_aabb, out, amb_mask, amb_cnt, off,
                                        heavy_list, heavy_cnt);
}

// The packets whose walk passed M.coop_budget in the main kernel (queries near a centre of symmetry of the
// mesh need a large part of the tree; one wave doing that alone is the tail of the whole launch): COOP_WAVES
// waves share one packet.  Every wave holds the same 64 queries and the lists saved at the hand-over — so its
// thresholds are nearly final from the start — and redoes the walk over the L1 nodes j with
// j % COOP_WAVES == wave; a group met again is recognised per lane by its ref.  The lists of the waves are
// merged through LDS (sorted inserts; refs already listed are skipped; a wave's fourth minimum only bounds
// what lies outside the first three) and wave 0 finishes the queries.  A wave that passes a quarter of the
// hard budget (8 waves: twice the work the main kernel alone would spend) sends the whole packet to the
// exact scan (k_query_heavy), as the main kernel does without this one: the few packets at a centre of
// symmetry need every group, and each wave running to the full budget made this kernel 1.1 ms long.
template <int KIND, bool ROBUST>
__global__ __launch_bounds__(64 * COOP_WAVES) SDFK_PK_SGPR void k_query_packet_coop(
    MeshDev M, const float* __restrict__ pts, const uint32_t* __restrict__ perm, int64_t n, int trunc_aabb,
    void* __restrict__ out, unsigned long long* __restrict__ amb_mask, uint32_t* __restrict__ amb_cnt, uint32_t off,
    uint32_t* __restrict__ heavy_list, uint32_t* __restrict__ heavy_cnt, CoopDesc CB) {
    // One block per handed-over packet and NO loop over packets: with a barrier (a fence) or a store ahead of
    // the walk on any path — the back edge of such a loop — the compiler can no longer prove the tree arrays
    // unwritten and turns every scalar chunk load of the walk into a vector load (26 global_load_dwordx4 in
    // place of 9 s_load_dwordx16 in the ISA of the looped version).  The host launches COOP_CAP blocks at
    // most; those past the count leave at once.
    __shared__ uint32_t sS[COOP_WAVES][7][64];
    __shared__ uint32_t s_over[COOP_WAVES];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t cnt = min(heavy_cnt[1], COOP_CAP);
    const uint32_t e = blockIdx.x;
    if (e >= cnt) return;  // block-uniform
    {
        const int64_t i = (int64_t)CB.list[e] * 64 + lane;
        const bool valid = i < n;
        const int64_t ic = valid ? i : n - 1;
        const int64_t src = perm ? (int64_t)perm[ic] : ic;
        const V3 p = {pts[3 * src], pts[3 * src + 1], pts[3 * src + 2]};
        ExactState SE = {0x7F7FFFFFu, 0x7F7FFFFFu, H_NONE, 0xFFFFFFFFu};
        const uint32_t* st = CB.state + (size_t)e * 448u + lane;
        ApproxState A = {__uint_as_float(st[0]),   __uint_as_float(st[64]), __uint_as_float(st[128]),
                         __uint_as_float(st[192]), st[256],                 st[320],
                         st[384]};
        const bool over = packet_walk<true, true>(M, p.x, p.y, p.z, M.amb_c2, SE, A,
                                                  wave, (uint32_t)COOP_WAVES, max(M.coop_budget, M.heavy_budget / 4u)) != 0u;
        if (lane == 0) s_over[wave] = over ? 1u : 0u;
        if (wave != 0) {
            sS[wave][0][lane] = __float_as_uint(A.B1), sS[wave][1][lane] = __float_as_uint(A.B2);
            sS[wave][2][lane] = __float_as_uint(A.B3), sS[wave][3][lane] = __float_as_uint(A.B4);
            sS[wave][4][lane] = A.H1, sS[wave][5][lane] = A.H2, sS[wave][6][lane] = A.H3;
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t any_over = 0u;
            for (int w = 0; w < COOP_WAVES; ++w) any_over |= s_over[w];
            if (any_over) {
                packet_to_heavy(valid, i, src, off, amb_mask, amb_cnt, heavy_list, heavy_cnt);
            } else {
                for (int w = 1; w < COOP_WAVES; ++w) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float gm = __uint_as_float(sS[w][k][lane]);
                        const uint32_t ref = sS[w][4 + k][lane];
                        if (ref == H_NONE || ref == A.H1 || ref == A.H2 || ref == A.H3) continue;
                        const bool c1 = gm < A.B1, c2 = gm < A.B2, c3 = gm < A.B3;
                        A.H3 = c2 ? A.H2 : (c3 ? ref : A.H3);
                        A.H2 = c1 ? A.H1 : (c2 ? ref : A.H2);
                        A.H1 = c1 ? ref : A.H1;
                        float t = fmaxf(A.B1, gm);
                        A.B1 = fminf(A.B1, gm);
                        float t2 = fmaxf(A.B2, t);
                        A.B2 = fminf(A.B2, t);
                        t = fmaxf(A.B3, t2);
                        A.B3 = fminf(A.B3, t2);
                        A.B4 = fminf(A.B4, t);
                    }
                    // (>= the merged third minimum: the global third is <= any wave's third <= its fourth)
                    A.B4 = fminf(A.B4, __uint_as_float(sS[w][3][lane]));
                }
                float sign = 0.f;
                if (ROBUST && KIND != Q_NN) sign = raycast_packet(M, p, valid);
                PacketBest pb;
                pb.bailed = false;
                packet_finish<KIND, ROBUST, true>(M, p, valid, i, src, sign, pb, A, trunc_aabb, out, amb_mask,
                                                  amb_cnt, off, heavy_list, heavy_cnt);
            }
        }
    }
}

// amb_cnt has been exclusive-scanned: wave w's ambiguous queries go to fb_list[amb_cnt[w] ...]
// The list accumulates over the chunks of one call (entries are absolute point indices):
// *base_in = entries already in the list, *total_out (a different word) = new total.
__global__ void k_fb_compact(const unsigned long long* __restrict__ amb_mask,
                             const uint32_t* __restrict__ amb_scan, const uint32_t* __restrict__ perm,
                             int64_t n, uint32_t off, uint32_t* __restrict__ fb_list,
                             const uint32_t* __restrict__ base_in, uint32_t* __restrict__ total_out) {
    // one thread per wave word: ambiguous queries are rare, most threads read one zero and leave
    const int64_t nw = (n + 63) >> 6;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    unsigned long long m = amb_mask[w];
    const uint32_t base = *base_in + amb_scan[w];
    if (w == nw - 1) *total_out = base + (uint32_t)__popcll(m);
    uint32_t k = 0;
    while (m) {
        const int lane = __ffsll((long long)m) - 1;
        m &= m - 1;
        fb_list[base + k++] = off + perm[64 * w + lane];
    }
}

// The exact scan.  Answers (a) the queries of the waves that gave up in the packet pass (heavy
// list) and (b) whole small batches ("scan mode", heavy_cnt == nullptr): a BLOCK of WAVES waves
// takes HEAVY_Q queries at a time and its threads stride over all (padded) tree positions — no
// pruning, no dependent loads.  One vertex load serves HEAVY_Q distance evaluations; loads go out
// four at a time; the body is branch-free (the loop is latency bound otherwise).  The same three
// things come out as from the packet pass: the exact fp32 minimum b, the second smallest distance
// s as a multiset (always exact here), and h = among the vertices at distance exactly b the one
// nanoflann's DFS meets first (fast_nn: the lowest position).  Then the packet kernel's own
// decision: window test, path check, else defer to the strict kernel's list.
// pos_out != nullptr (scan mode, kinds with a tail): the tree position (-1 none, -2 deferred) is
// stored per query and k_scan_finish does ray parity + face distances one lane per query.
constexpr int HEAVY_Q_DEFAULT = 4;
constexpr int HEAVY_Q = HEAVY_Q_DEFAULT;  // (host side: queries per block of the default instantiation)
template <int KIND, bool ROBUST, int WAVES, int HEAVY_Q = HEAVY_Q_DEFAULT>
__global__ __launch_bounds__(64 * WAVES) void k_query_heavy(MeshDev M, const float* __restrict__ pts,
                                                            const uint32_t* __restrict__ heavy_list,
                                                            const uint32_t* __restrict__ heavy_cnt,
                                                            int trunc_aabb, void* __restrict__ out,
                                                            uint32_t* __restrict__ fb_list,
                                                            uint32_t* __restrict__ fb_cnt, uint32_t cnt_all,
                                                            int32_t* __restrict__ pos_out, uint32_t only_min,
                                                            uint32_t only_max) {
    __shared__ uint32_t sb[WAVES][HEAVY_Q], ss[WAVES][HEAVY_Q], sh[WAVES][HEAVY_Q];
    const uint32_t cnt = heavy_cnt ? *heavy_cnt : cnt_all;
    // The length of the heavy list is known on the device only: the host launches a 16-wave and a
    // 4-wave variant and each serves its own range of lengths (block-uniform exit, before any barrier).
    if (cnt < only_min || cnt > only_max) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t STEP = 64u * WAVES;  // positions per block per sub-step
    for (uint32_t q0 = HEAVY_Q * blockIdx.x; q0 < cnt; q0 += HEAVY_Q * gridDim.x) {  // block-uniform
        uint32_t src[HEAVY_Q], b[HEAVY_Q], s2[HEAVY_Q], h[HEAVY_Q];
        V3 p[HEAVY_Q];
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            const uint32_t q = q0 + j < cnt ? q0 + j : cnt - 1;  // the tail repeats the last query
            src[j] = heavy_cnt ? heavy_list[q] : q;
            p[j] = {pts[3 * (size_t)src[j]], pts[3 * (size_t)src[j] + 1], pts[3 * (size_t)src[j] + 2]};
            b[j] = 0x7F7FFFFFu, s2[j] = 0x7F7FFFFFu, h[j] = H_NONE;
        }
        // Uniform trip count, no branch in the body: a position past the end reads the last slot
        // and gets the distance 0xFFFFFFFF, which changes nothing.
        for (uint32_t k0 = 0; k0 < M.kd_npos; k0 += 4u * STEP) {
            float4 P[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + STEP * u + threadIdx.x;
                P[u] = M.kd_pts4[k < M.kd_npos ? k : M.kd_npos - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + STEP * u + threadIdx.x;
#pragma unroll
                for (int j = 0; j < HEAVY_Q; ++j) {
                    const float dx = p[j].x - P[u].x, dy = p[j].y - P[u].y, dz = p[j].z - P[u].z;
                    uint32_t du = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                    du = k < M.kd_npos ? du : 0xFFFFFFFFu;
                    // on du == b the earlier (lower) position stays and s2 = b flags the tie
                    const bool lt = du < b[j];
                    s2[j] = lt ? b[j] : min(s2[j], du);
                    h[j] = lt ? k : h[j];
                    b[j] = min(b[j], du);
                }
            }
        }
        // wave reduction, then the block's waves through LDS; every thread ends with the merged triple
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t ob = __shfl_xor(b[j], o), os = __shfl_xor(s2[j], o), oh = __shfl_xor(h[j], o);
                const uint32_t nh = ob < b[j] ? oh : (b[j] < ob ? h[j] : min(h[j], oh));  // H_NONE is the largest
                s2[j] = min(min(s2[j], os), max(b[j], ob));
                b[j] = min(b[j], ob);
                h[j] = nh;
            }
            if (lane == 0) sb[wave][j] = b[j], ss[wave][j] = s2[j], sh[wave][j] = h[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            b[j] = sb[0][j], s2[j] = ss[0][j], h[j] = sh[0][j];
            for (int w = 1; w < WAVES; ++w) {
                const uint32_t ob = sb[w][j], os = ss[w][j], oh = sh[w][j];
                const uint32_t nh = ob < b[j] ? oh : (b[j] < ob ? h[j] : min(h[j], oh));
                s2[j] = min(min(s2[j], os), max(b[j], ob));
                b[j] = min(b[j], ob);
                h[j] = nh;
            }
        }
        __syncthreads();  // LDS is reused below
        // Exact tie (block-uniform): h is the lowest tied position, which is the fast_nn rule;
        // nanoflann's rule wants the tied vertex its DFS meets first: second scan over the
        // vertices at distance exactly b, folded with kd_dfs_first (a total order for a fixed
        // query).  Kept out of the main loop: one tree walk per tied vertex.  One pass serves all the
        // tied queries of the block (next to the axis of a 1M-vertex torus most queries of a block are tied,
        // with hundreds of vertices each: a pass per query was 1.75 of the kernel's 3.0 ms there).
        bool tie[HEAVY_Q], any_tie = false;
        uint32_t hh[HEAVY_Q];
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            tie[j] = s2[j] == b[j] && h[j] != H_NONE && !M.fast;  // (block-uniform)
            any_tie = any_tie || tie[j];
            hh[j] = H_NONE;
        }
        if (any_tie) {
            for (uint32_t k0 = 0; k0 < M.kd_npos; k0 += 4u * STEP) {
                float4 P[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = k0 + STEP * u + threadIdx.x;
                    P[u] = M.kd_pts4[k < M.kd_npos ? k : M.kd_npos - 1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = k0 + STEP * u + threadIdx.x;
#pragma unroll
                    for (int j = 0; j < HEAVY_Q; ++j) {
                        if (!tie[j]) continue;
                        const float dx = p[j].x - P[u].x, dy = p[j].y - P[u].y, dz = p[j].z - P[u].z;
                        const uint32_t du = __float_as_uint((dx * dx + dy * dy) + dz * dz);
                        if (k < M.kd_npos && du == b[j])
                            hh[j] = hh[j] == H_NONE ? k : (uint32_t)kd_dfs_first(M, p[j].x, p[j].y, p[j].z, (int)hh[j], (int)k);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < HEAVY_Q; ++j) {
                if (!tie[j]) continue;
                uint32_t x = hh[j];
                for (int o = 32; o >= 1; o >>= 1) {
                    const uint32_t oh = __shfl_xor(x, o);
                    if (x == H_NONE)
                        x = oh;
                    else if (oh != H_NONE)
                        x = (uint32_t)kd_dfs_first(M, p[j].x, p[j].y, p[j].z, (int)x, (int)oh);
                }
                if (lane == 0) sh[wave][j] = x;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < HEAVY_Q; ++j) {
                if (!tie[j]) continue;
                uint32_t x = sh[0][j];
                for (int w = 1; w < WAVES; ++w) {
                    const uint32_t oh = sh[w][j];
                    if (x == H_NONE)
                        x = oh;
                    else if (oh != H_NONE)
                        x = (uint32_t)kd_dfs_first(M, p[j].x, p[j].y, p[j].z, (int)x, (int)oh);
                }
                h[j] = x;
            }
            __syncthreads();
        }
        // thread j of the block finishes query j
        if (threadIdx.x >= (uint32_t)HEAVY_Q || q0 + threadIdx.x >= cnt) continue;
        uint32_t mb = b[0], ms = s2[0], mh = h[0], msrc = src[0];
        V3 mp = p[0];
#pragma unroll
        for (int j = 1; j < HEAVY_Q; ++j)
            if (threadIdx.x == (uint32_t)j) mb = b[j], ms = s2[j], mh = h[j], msrc = src[j], mp = p[j];
        const int pos = mh == H_NONE ? -1 : (int)mh;
        const float bf = __uint_as_float(mb), sf = __uint_as_float(ms);
        bool amb = false;
        if (!M.fast) {
            if (!(sf > bf * M.amb_c) || !(bf >= 1e-30f)) {
                amb = true;
                if (pos >= 0 && bf < FLT_MAX) amb = !kd_path_clear(M, mp.x, mp.y, mp.z, pos, sf);
            }
            if (pos < 0) amb = true;
        }
        if (amb) fb_list[atomicAdd(fb_cnt, 1u)] = msrc;
        if (pos_out) {  // per slot of the list (scan mode: slot == query)
            pos_out[q0 + threadIdx.x] = amb ? -2 : pos;
        } else if (amb) {
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

// scan mode, kinds with a tail: ray parity + incident-face distances, one lane per query
// list != nullptr: slot s of pos_in belongs to query list[s], *cnt_dev slots (nothing to do beyond n of them:
// longer lists were finished inside k_query_heavy)
template <int KIND, bool ROBUST>
__global__ __launch_bounds__(256) void k_scan_finish(MeshDev M, const float* __restrict__ pts,
                                                     const int32_t* __restrict__ pos_in, uint32_t n,
                                                     int trunc_aabb, void* __restrict__ out,
                                                     const uint32_t* __restrict__ list = nullptr,
                                                     const uint32_t* __restrict__ cnt_dev = nullptr) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (cnt_dev && *cnt_dev > n) return;
    if (slot >= (cnt_dev ? *cnt_dev : n)) return;
    const int pos = pos_in[slot];
    if (pos == -2) return;  // deferred: the strict kernel answers it
    const uint32_t i = list ? list[slot] : slot;
    const V3 p = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
    if (KIND == Q_NN) {
        ((int32_t*)out)[i] = pos >= 0 ? M.kd_vind[pos] : -1;
        return;
    }
    float sign = 0.f;
    if (ROBUST) sign = raycast(M, p, nullptr);
    const bool trunc = (KIND == Q_CONTAINS) ? true : (trunc_aabb != 0);
    const float v = calc_tail<ROBUST>(M, p, trunc, sign, pos);
    if (KIND == Q_CONTAINS)
        ((uint8_t*)out)[i] = v >= 0;
    else
        ((float*)out)[i] = v;
}
