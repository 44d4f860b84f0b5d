// kernels_update.h — sample_surface and the per-face precompute of update().  Part of sdfhip.hip.
#pragma once

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

// The approximate pass's own copy of the vertices (tree order), centred on the mesh: v' = fl(v - c) and |v'|^2
// rounded once from double (the host computes the same numbers for its maximum); sentinels: a() = +inf.
// kd_lbox: every tree position gets the tight box of its kd leaf, taken from the parent node's child box
// (what nanoflann's divideTree returns for a leaf: the extent of its points).  One thread per node; a tree
// that is one leaf (no nodes) takes the root box.
__global__ void k_leaf_boxes(const float4* __restrict__ pn, uint32_t nnodes, uint32_t root, float b0, float b1, float b2,
                             float b3, float b4, float b5, float4* __restrict__ lbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (nnodes == 0) {
        if (i == 0 && (root & sdfk::KD_LEAF_BIT)) {
            const uint32_t left = (root & sdfk::KD_REF_MASK) >> 4, cnt = root & 15u;
            for (uint32_t k = 0; k < cnt; ++k) {
                lbox[2 * (size_t)(left + k)] = make_float4(b0, b1, b2, b3);
                lbox[2 * (size_t)(left + k) + 1] = make_float4(b4, b5, 0.f, 0.f);
            }
        }
        return;
    }
    if (i >= nnodes) return;
    const float4 bA = pn[4 * (size_t)i], bB = pn[4 * (size_t)i + 1], bC = pn[4 * (size_t)i + 2], nf = pn[4 * (size_t)i + 3];
    const uint32_t r1 = __float_as_uint(nf.z) & ~(3u << sdfk::KD_FEAT_SHIFT), r2 = __float_as_uint(nf.w);
    if (r1 & sdfk::KD_LEAF_BIT) {
        const uint32_t left = (r1 & sdfk::KD_REF_MASK) >> 4, cnt = r1 & 15u;
        for (uint32_t k = 0; k < cnt; ++k) {
            lbox[2 * (size_t)(left + k)] = make_float4(bA.x, bA.y, bA.z, bA.w);
            lbox[2 * (size_t)(left + k) + 1] = make_float4(bB.x, bB.y, 0.f, 0.f);
        }
    }
    if (r2 & sdfk::KD_LEAF_BIT) {
        const uint32_t left = (r2 & sdfk::KD_REF_MASK) >> 4, cnt = r2 & 15u;
        for (uint32_t k = 0; k < cnt; ++k) {
            lbox[2 * (size_t)(left + k)] = make_float4(bB.z, bB.w, bC.x, bC.y);
            lbox[2 * (size_t)(left + k) + 1] = make_float4(bC.z, bC.w, 0.f, 0.f);
        }
    }
}

// The vertices in (padded) tree order: pts4[i] = (v, |v|^2 rounded once from double) for the vertex at tree position i,
// (3e38, 3e38, 3e38, +inf) for sentinel slots and the tail pad — their distance to anything is +inf.  (The host did
// this gather and uploaded the result until round 3: a helper thread and 16 B per position over PCIe per update.)
__global__ void k_tree_positions(const float* __restrict__ verts, const int32_t* __restrict__ vind, uint32_t npos,
                                 uint32_t nalloc, float4* __restrict__ pts4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nalloc) return;
    const int32_t vi = i < npos ? vind[i] : -1;
    if (vi < 0) {
        pts4[i] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __builtin_inff());
    } else {
        const float x = verts[3 * (size_t)vi], y = verts[3 * (size_t)vi + 1], z = verts[3 * (size_t)vi + 2];
        pts4[i] = make_float4(x, y, z, (float)(((double)x * x + (double)y * y) + (double)z * z));
    }
}

// The approximate pass's copy of the vertices, centred per packet group (round 3): group g with centre cg (its
// box centre; any fp32 point near the group will do) stores per slot  w2 = -2 fl(v - cg)  and  W = |fl(v - cg)|^2
// (rounded once from double; +inf for sentinel slots and non-finite vertices: their estimate is never a minimum),
// and cg itself at actr[first position / 4].  A lane then estimates d^2(q, v) as
//     fma(t.x, w2.x, fma(t.y, w2.y, fma(t.z, w2.z, W))) + |t|^2,   t = fl(q - cg),
// whose products are of the size of the group (|w| <= R ~ 0.03), not of the scene: the error is
// u (14 d^2 + 21 R^2) instead of u (5 S + 4 Q) with S, Q of the order of the scene's squared size (kernels_packet.h).
// One wave per group; the wave after the last group fills the tail pad.  rmax2_bits[0]: max W over the real
// slots, [1]: set when a finite vertex could not be stored, [2], [3]: density statistics (below).
__global__ __launch_bounds__(64) void k_apx_fill(const uint32_t* __restrict__ grefs, uint32_t ngroups,
                                                 const float4* __restrict__ gobb_aos, const float4* __restrict__ pts4,
                                                 const int32_t* __restrict__ vind, uint32_t npos, uint32_t nalloc,
                                                 float4* __restrict__ apx4, float4* __restrict__ actr,
                                                 uint32_t* __restrict__ rmax2_bits, float dens_k) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x;  // one wave per group
    const float inf = __builtin_inff();
    if (g == ngroups) {
        for (uint32_t k = npos + lane; k < nalloc; k += 64u) apx4[k] = make_float4(0.f, 0.f, 0.f, inf);
        for (uint32_t k = (npos >> 2) + lane; k < (nalloc + 3u) >> 2; k += 64u) actr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    if (g > ngroups) return;
    const uint32_t ref = grefs[g];
    const uint32_t first = ((ref & sdfk::KD_REF_MASK) >> 4) << 2, n4 = 4u * (ref & 15u);
    const float4 A = gobb_aos[4u * (size_t)g];
    const bool okc = fabsf(A.x) <= FLT_MAX && fabsf(A.y) <= FLT_MAX && fabsf(A.z) <= FLT_MAX;
    const float cx = okc ? A.x : 0.f, cy = okc ? A.y : 0.f, cz = okc ? A.z : 0.f;
    float rmax2 = 0.f;
    uint32_t wild = 0u, cnt = 0u;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t k = first + lane; k < first + n4; k += 64u) {
        const float4 P = pts4[k];
        if (vind[k] >= 0) {
            ++cnt;
            lo[0] = fminf(lo[0], P.x), lo[1] = fminf(lo[1], P.y), lo[2] = fminf(lo[2], P.z);
            hi[0] = fmaxf(hi[0], P.x), hi[1] = fmaxf(hi[1], P.y), hi[2] = fmaxf(hi[2], P.z);
        }
        const float x = P.x - cx, y = P.y - cy, z = P.z - cz;
        const float W = (float)(((double)x * x + (double)y * y) + (double)z * z);
        if (vind[k] < 0 || !(W <= FLT_MAX)) {  // sentinel slot, or a vertex no finite arithmetic covers
            apx4[k] = make_float4(0.f, 0.f, 0.f, inf);
            // A vertex with finite coordinates whose distance from its group's centre overflows (a group ~1e19 wide)
            // would be invisible to the pass although some query may have it nearest: the host turns the pass off
            // for such a mesh (rmax2_bits[1] != 0).  Non-finite vertices are never nearest to anything.
            const bool real = vind[k] >= 0 && fabsf(P.x) <= FLT_MAX && fabsf(P.y) <= FLT_MAX && fabsf(P.z) <= FLT_MAX;
            if (real) wild = 1u;
        } else {
            apx4[k] = make_float4(-2.f * x, -2.f * y, -2.f * z, W);
            rmax2 = fmaxf(rmax2, W);
        }
    }
    for (uint32_t k = (first >> 2) + lane; k < (first + n4) >> 2; k += 64u) actr[k] = make_float4(cx, cy, cz, 0.f);
    for (int o = 32; o >= 1; o >>= 1) {
        rmax2 = fmaxf(rmax2, __shfl_xor(rmax2, o));
        wild |= __shfl_xor(wild, o);
        cnt += __shfl_xor(cnt, o);
#pragma unroll
        for (int d = 0; d < 3; ++d) lo[d] = fminf(lo[d], __shfl_xor(lo[d], o)), hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
    }
    if (lane == 0) {
        atomicMax(rmax2_bits, __float_as_uint(rmax2));  // (non-negative floats order like their bits)
        if (wild) atomicOr(rmax2_bits + 1, 1u);
        // vertex density of the group against the pass's window (fill_apx, sdfhip.hip): its vertices over
        // the disc of its box's half diagonal; [2]: vertices seen (groups of >= 8), [3]: of those in dense groups
        if (cnt >= 8u) {
            const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
            const double r2 = 0.25 * (dx * dx + dy * dy + dz * dz);
            atomicAdd(rmax2_bits + 2, cnt);
            if (!((double)dens_k * cnt <= r2)) atomicAdd(rmax2_bits + 3, cnt);
        }
    }
}

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

// geo records once more in vertex-major order (calc_tail reads a vertex's incident faces as one run)
__global__ void k_geo_adj(const float4* __restrict__ geo, const int32_t* __restrict__ adj_faces, uint32_t n_adj,
                          float4* __restrict__ geo_adj) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per thread
    if (t >= 4u * n_adj) return;
    geo_adj[t] = geo[4u * (size_t)adj_faces[t >> 2] + (t & 3u)];
}
