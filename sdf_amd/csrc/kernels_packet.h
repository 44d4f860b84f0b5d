// kernels_packet.h — the packet (one wave per 64 Morton-adjacent queries) nearest-vertex pass, the dominant
// kernel k_query_packet, the deferred-list compaction and the tail limiter's k_query_heavy.
// Part of sdfhip.hip (one translation unit).
#pragma once

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
// 80 SGPRs or fewer admit 8 waves per SIMD, 82-96 only 7 (MI355X_MICROARCH.md "Residency"); the
// compiler would take 90.  Measured: robust sdf kernel 16.3 -> 15.5 ms for 64M points, nn unchanged.
#ifndef SDFK_PK_SGPR
#define SDFK_PK_SGPR __attribute__((amdgpu_num_sgpr(80)))
#endif


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

__device__ __forceinline__ float fmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
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

// =========================================================================================
// Approximate first pass (strict mode).  The walk above spends 10 VALU per (query, vertex) on the
// reference's exact arithmetic although almost every vertex it meets is NOT the nearest one.
// Here a slot costs 4: with the vertex's squared norm in the padding word of kd_pts4,
//     a(v) = fma(-2qx, vx, fma(-2qy, vy, fma(-2qz, vz, |v|^2)))  ~  D(v) - |q|^2
// orders the vertices of one lane like nanoflann's fp32 distance D(v) up to a known error: with
// u = 2^-24, S = max|v|^2 + 2 sum_i |q_i| max|v_i| and Q = |q|^2 (fp32 Qf),
//     |a(v) + Qf - d^2(v)|  <=  u (4 S + 3 Q) =: Ea      (three roundings of partial sums <= S, the
//                                                        rounding of |v|^2, the error of Qf)
//     D(v) = d^2(v) (1 + t), |t| <= 5.01 u               (nanoflann.hpp:432-440, five roundings)
// so  lower(a) = (a + Qf - Ea)(1 - 8u)  <=  D(v)  <=  (a + Qf + Ea)(1 + 8u) = upper(a)   (constants
// rounded up; Ea is computed with 5 S + 4 Q).  Per lane the walk keeps the four smallest GROUP
// minima B1 <= B2 <= B3 <= B4 of a() and the groups of the first three.  Pruning uses upper(B1)*c^2,
// an upper bound of what the exact walk would use, so every vertex the exact walk needs is met.
// Afterwards each lane evaluates EXACTLY (reference arithmetic, tie rules of the exact walk) the
// groups whose minimum could still be the nearest: H1, and Hk while lower(Bk) <= upper(B1).
// Every vertex outside them has D >= lower(B_next) > upper(B1) >= the exact minimum b found
// inside them, strictly: b is the global minimum, no outside vertex ties with it, and
// lower(B_next) is a lower bound of the distance of every outside vertex — which is all the
// decision logic needs of `s`.  A fourth candidate group defers the query to the strict kernel.
// =========================================================================================
struct ApproxLane {
    float B1, B2, B3, B4;  // smallest group minima of a(), ascending
    uint32_t H1, H2, H3;   // refs of the groups of B1..B3 (H_NONE: none)
    bool bailed;
};

__device__ __forceinline__ ApproxLane kd_nearest_packet_approx(const MeshDev& M, float qx, float qy, float qz,
                                                               float Qe, float cm, const PacketStack stk) {
    const uint32_t lane = threadIdx.x & 63u;
    const float INF = __builtin_inff();
    const float X2 = -2.f * qx, Y2 = -2.f * qy, Z2 = -2.f * qz;
    float B1 = INF, B2 = INF, B3 = INF, B4 = INF;
    uint32_t H1 = H_NONE, H2 = H_NONE, H3 = H_NONE;
    uint32_t cur = M.kd_qroot;
    uint32_t sp = 0;
    uint32_t work = 0;
    SDFK_COUNT(0, 1);
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        bool need_pop = false;
        if (!(cur & KD_LEAF)) {
            SDFK_COUNT(1, 1);
            const float* Q = (const float*)(M.kd_qn + 8 * (size_t)cur);
            const uint32_t r0 = __float_as_uint(Q[24]), r1 = __float_as_uint(Q[25]),
                           r2 = __float_as_uint(Q[26]), r3 = __float_as_uint(Q[27]);
            const float thr = fmaxf(B1 + Qe, 0.f) * cm;  // >= (exact best so far) * c^2
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
            const uint32_t nch = cur & 15u;
            const uint32_t first = ((cur & sdfk::KD_REF_MASK) >> 4) << 2;
            work += nch;
            SDFK_COUNT(2, 1);
            SDFK_COUNT(3, nch);
            if (nch) {
                const u32x16* C = (const u32x16*)__builtin_assume_aligned(M.kd_pts4 + first, 64);
                float gm = INF;  // (a NaN never enters: fminf keeps the other operand)
                for (uint32_t ci = 0; ci < nch; ++ci) {
                    const u32x16 A = C[ci];
#define SDFK_APX_SLOT(O_)                                                                  \
    __builtin_fmaf(X2, __uint_as_float(A[O_]),                                             \
                   __builtin_fmaf(Y2, __uint_as_float(A[O_ + 1]),                          \
                                  __builtin_fmaf(Z2, __uint_as_float(A[O_ + 2]), __uint_as_float(A[O_ + 3]))))
                    // 12 FMA + 2 v_min3 per chunk (a NaN operand is dropped by min3 as by min)
                    const float a0 = SDFK_APX_SLOT(0), a1 = SDFK_APX_SLOT(4), a2 = SDFK_APX_SLOT(8),
                                a3 = SDFK_APX_SLOT(12);
                    gm = fmin3(gm, a0, a1);
                    gm = fmin3(gm, a2, a3);
#undef SDFK_APX_SLOT
                }
                // sorted insert of this group's minimum into B1..B4 (+ the groups of the first three);
                // skipped when no lane's top four change (the insert is then the identity)
                if (__ballot(gm < B4) != 0ull) {
                const bool c1 = gm < B1, c2 = gm < B2, c3 = gm < B3;
                H3 = c2 ? H2 : (c3 ? cur : H3);
                H2 = c1 ? H1 : (c2 ? cur : H2);
                H1 = c1 ? cur : H1;
                float t = fmaxf(B1, gm);
                B1 = fminf(B1, gm);
                float t2 = fmaxf(B2, t);
                B2 = fminf(B2, t);
                t = fmaxf(B3, t2);
                B3 = fminf(B3, t2);
                B4 = fminf(B4, t);
                }
            }
            need_pop = true;
        }
        if (need_pop) {
            const uint32_t thr16 = __float_as_uint(fmaxf(B1 + Qe, 0.f) * cm) >> 16;
            const bool over = work > M.heavy_budget;
            if (over) sp = 0;
            for (;;) {
                if (sp == 0) return ApproxLane{B1, B2, B3, B4, H1, H2, H3, over};
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

// exact evaluation of one packet group for one lane (second phase of the approximate pass): the
// reference's metric, strict `<`, and among exact ties the vertex nanoflann's DFS meets first
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
            if (__builtin_expect(du[j] == b && pos != H_NONE, 0)) {
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

template <int KIND, bool ROBUST, bool APPROX>
__global__ __launch_bounds__(PBLOCK) SDFK_PK_SGPR void k_query_packet(MeshDev M, const float* __restrict__ pts,
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
    PacketBest pb;
    ApproxLane A;
    float Qf = 0.f, Ea = 0.f;
    if (APPROX) {
        // error model of a() for this lane (see kd_nearest_packet_approx)
        const float u = 5.9604645e-8f;  // 2^-24
        const float Vx = fmaxf(fabsf(M.bb[0]), fabsf(M.bb[3])), Vy = fmaxf(fabsf(M.bb[1]), fabsf(M.bb[4])),
                    Vz = fmaxf(fabsf(M.bb[2]), fabsf(M.bb[5]));
        const float S = M.vmax2 + 2.f * ((fabsf(p.x) * Vx + fabsf(p.y) * Vy) + fabsf(p.z) * Vz);
        Qf = (p.x * p.x + p.y * p.y) + p.z * p.z;
        Ea = u * (5.f * S + 4.f * Qf);
        A = kd_nearest_packet_approx(M, p.x, p.y, p.z, Qf + Ea, (1.f + 8.f * u) * M.amb_c2, stk);
        pb.bailed = A.bailed;
    } else {
        pb = kd_nearest_packet(M, p.x, p.y, p.z, stk);
    }
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
    int pos;
    bool amb = false;
    if (APPROX) {
        // second phase: the groups whose minimum could be the nearest vertex, exactly
        const float u = 5.9604645e-8f;
        const float k1p = 1.f + 8.f * u, k1m = 1.f - 8.f * u;
        const float x1 = A.B1 + (Qf + Ea);
        const float U = fmaxf(fmaxf(x1 * k1p, x1), 0.f);  // >= D of B1's vertex >= the exact minimum
        const float y2 = (A.B2 + Qf) - Ea, y3 = (A.B3 + Qf) - Ea, y4 = (A.B4 + Qf) - Ea;
        const float L2 = fminf(y2 * k1m, y2), L3 = fminf(y3 * k1m, y3), L4 = fminf(y4 * k1m, y4);  // <= D outside
        const bool n2 = !(L2 > U), n3 = n2 && !(L3 > U), n4 = n3 && !(L4 > U);
        uint32_t b = 0x7F7FFFFFu, s2 = 0x7F7FFFFFu, hp = H_NONE;
        if (A.H1 != H_NONE && !n4) {
            exact_group(M, p.x, p.y, p.z, A.H1, b, s2, hp);
            if (n2) exact_group(M, p.x, p.y, p.z, A.H2, b, s2, hp);
            if (n3) exact_group(M, p.x, p.y, p.z, A.H3, b, s2, hp);
        }
        const float s_out = n3 ? L4 : (n2 ? L3 : L2);  // every vertex outside the evaluated groups is farther
        pb.b = __uint_as_float(b);
        pb.s = fminf(__uint_as_float(s2), s_out);
        pos = hp == H_NONE ? -1 : (int)hp;  // (-1: nothing finite, or a fourth candidate group: deferred below)
    } else {
        pos = packet_resolve(M, p.x, p.y, p.z, pb.b, pb.h);
    }
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
    if (APPROX) {
        // what the approximate pass cannot settle (a fourth candidate group, no finite distance, a
        // window the path check does not clear) goes to the exact scan of k_query_heavy, which
        // takes the same decision with exact (b, s, h) and defers to the strict kernel itself
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
constexpr int HEAVY_Q = 4;
template <int KIND, bool ROBUST, int WAVES>
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
        // query).  Kept out of the main loop: one tree walk per tied vertex.
#pragma unroll
        for (int j = 0; j < HEAVY_Q; ++j) {
            if (!(s2[j] == b[j] && h[j] != H_NONE && !M.fast)) continue;
            uint32_t hh = H_NONE;
            for (uint32_t k0 = 0; k0 < M.kd_npos; k0 += 4u * STEP) {
                uint32_t du[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = k0 + STEP * u + threadIdx.x;
                    const float4 P = M.kd_pts4[k < M.kd_npos ? k : M.kd_npos - 1];
                    const float dx = p[j].x - P.x, dy = p[j].y - P.y, dz = p[j].z - P.z;
                    du[u] = k < M.kd_npos ? __float_as_uint((dx * dx + dy * dy) + dz * dz) : 0xFFFFFFFFu;
                }
                if (__ballot(du[0] == b[j] || du[1] == b[j] || du[2] == b[j] || du[3] == b[j]) == 0ull) continue;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t k = k0 + STEP * u + threadIdx.x;
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
            if (lane == 0) sh[wave][j] = hh;
            __syncthreads();
            hh = sh[0][j];
            for (int w = 1; w < WAVES; ++w) {
                const uint32_t oh = sh[w][j];
                if (hh == H_NONE)
                    hh = oh;
                else if (oh != H_NONE)
                    hh = (uint32_t)kd_dfs_first(M, p[j].x, p[j].y, p[j].z, (int)hh, (int)oh);
            }
            h[j] = hh;
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
                if (pos >= 0 && bf < FLT_MAX) amb = !(kd_path_max_bound(M, mp.x, mp.y, mp.z, pos) <= sf);
            }
            if (pos < 0) amb = true;
        }
        if (amb) fb_list[atomicAdd(fb_cnt, 1u)] = msrc;
        if (pos_out) {
            pos_out[msrc] = amb ? -2 : pos;
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
template <int KIND, bool ROBUST>
__global__ __launch_bounds__(256) void k_scan_finish(MeshDev M, const float* __restrict__ pts,
                                                     const int32_t* __restrict__ pos_in, uint32_t n,
                                                     int trunc_aabb, void* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pos = pos_in[i];
    if (pos == -2) return;  // deferred: the strict kernel answers it
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
