// kernels_build.h — Morton codes, query bounds, the face LBVH (Karras topology, threaded layout, refit),
// the GPU-built vertex tree of fast_nn and the column-grid fill.  Part of sdfhip.hip (one translation unit).
#pragma once

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
    else if (threadIdx.x < 8) b[threadIdx.x] = 0u;  // ([6]: far queries of the launch, counted by k_query_morton)
}
// flat float view of the [n,3] points: thread-contiguous float4 loads (coordinate = index % 3).
// Base pointers from hipMalloc / torch are 16-byte aligned, but a slice (pts[1:], a shard that starts
// at an odd point) is not: the float4 view then starts at the first aligned float (`head` floats in),
// and the coordinates are rotated back at the end.  Head and tail are read scalar.
// sample > 1: only one run of 64 groups (256 points, 3 KB) in every `sample` runs is read.  The
// bounds only scale the Morton codes of the query sort (values outside are clamped into the border
// cells), so a sample costs at most a little coherence, never a result.
__global__ __launch_bounds__(256) void k_bounds(const float* __restrict__ pts, int64_t n, uint32_t* b,
                                                int sample) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int64_t nf = 3 * n;
    const int mis = (int)((((uintptr_t)pts) & 15) >> 2);  // floats past a 16-byte boundary
    const int64_t head = mis ? (4 - mis < nf ? 4 - mis : nf) : 0;
    const int64_t ngroups = (nf - head) / 12;  // 12 floats = 4 points = 3 float4
    const int64_t nsampled = sample > 1 ? (ngroups / (64 * (int64_t)sample)) * 64 : ngroups;
    const float4* p4 = (const float4*)(pts + head);
    for (int64_t gs = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gs < nsampled;
         gs += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = sample > 1 ? (gs >> 6) * 64 * sample + (gs & 63) : gs;
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
    if (head % 3 == 1) {  // float k of the aligned view is coordinate (k + head) % 3 of its point
        const float l2 = lo[2], h2 = hi[2];
        lo[2] = lo[1], hi[2] = hi[1], lo[1] = lo[0], hi[1] = hi[0], lo[0] = l2, hi[0] = h2;
    } else if (head % 3 == 2) {
        const float l0 = lo[0], h0 = hi[0];
        lo[0] = lo[1], hi[0] = hi[1], lo[1] = lo[2], hi[1] = hi[2], lo[2] = l0, hi[2] = h0;
    }
    // head and tail, scalar (a sampled pass may skip them: the bounds are a sample anyway — but a
    // batch too short for one sampled run is read completely)
    const bool edges = sample <= 1 || nsampled == 0;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; edges && j < nf - 12 * ngroups;
         j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = j < head ? j : 12 * ngroups + j;
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
// Sort key of robust contains(): the query's column of the ray-cast frame, a 12 + 12 bit Morton code of
// (x', y') over the column grid's range.  The z+ / z- rays of a query read the face list of its column, so
// the queries of one wave then share a handful of lists instead of reading 64 (the lists of a rough mesh
// are 2 KB each: the 3-D order moved 6 TB/s for them).  r0, r1: rows of R giving x', y'.
// far queries of a batch (kernels_packet.h, k_query_packet_far): centre and squared radius of the near region
struct FarKey {
    float c[3], r2;    // r2: squared far radius for a dense batch; 0: off
    float r2_sparse;   // ... for a sparse one
    float rho_dense, rho_sparse;  // packet radius (half the edge of a cube holding 64 points of the batch) up to which the
                                  // batch counts as dense / from which as sparse; in between the radius is interpolated
};
struct ColumnKey {
    int on;
    float r0[3], r1[3];
    float o0, o1, s0, s1;  // x' -> (x' - o0) * s0 in [0, 4096)
};
// One radix-sort tile per block (element (k, t) = base + k * 256 + t as in radix_sort.h): the keys, and the
// tile's digit histogram of the first sort pass (what k_rs_hist would read the keys again for).
__global__ __launch_bounds__(RS_THREADS) void k_query_morton(const float* __restrict__ pts, uint32_t n,
                                                             uint32_t* __restrict__ b, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ hist, uint32_t ntiles, ColumnKey CK, int key_shift,
                                                             FarKey FK) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    float lo[3] = {0.f, 0.f, 0.f}, inv[3] = {0.f, 0.f, 0.f};
    if (!CK.on)
        for (int d = 0; d < 3; ++d) {
            lo[d] = ord2f(b[d]);
            const float ext = ord2f(b[3 + d]) - lo[d];
            inv[d] = ext > 0.f ? 1024.f / ext : 0.f;
        }
    // Far queries (FarKey): when the batch reaches beyond the near box of the mesh, far queries sort behind the near
    // ones — top key bit — by distance shell (3 bits) and direction (Morton code of the unit vector), and the near
    // ones by their Morton code over the near box instead of the batch's bounds (which may be hundreds of meshes wide).
    bool far_on = false;
    uint32_t nfar = 0;
    float far_r2 = 0.f;
    if (!CK.on && FK.r2 > 0.f) {
        // The far radius follows the batch's density.  64 Morton neighbours of a dense batch stay a compact packet at
        // any distance and the spatial walk serves them well out to twelve half diagonals (closer in, the cap of the mesh
        // facing a query is large and the far walk evaluates too much: 8M points in a box +-4 around the unit torus,
        // far radius 2 / 6 half diagonals: 14.2 / 4.2 ms); a sparse batch's packets are wide wherever they lie, walk
        // badly or not at all, and the far walk takes over from two half diagonals on (8M points in +-30: 6.4 against
        // 7.1 ms, 1M points in +-16: 3.5 against 4.5).  Every block computes the same radius from the batch's bounds;
        // block 0 leaves it in b[7] for the packet kernels.
        float vol = 1.f;
        for (int d = 0; d < 3; ++d) vol *= fmaxf(ord2f(b[3 + d]) - ord2f(b[d]), 0.f);
        const float rho = 0.5f * cbrtf(64.f * vol / (float)n);
        float t = (rho - FK.rho_dense) / (FK.rho_sparse - FK.rho_dense);
        t = t >= 0.f ? (t <= 1.f ? t : 1.f) : 0.f;  // (NaN: dense)
        const float rf = __builtin_sqrtf(FK.r2) + t * (__builtin_sqrtf(FK.r2_sparse) - __builtin_sqrtf(FK.r2));
        far_r2 = rf * rf;
        if (blockIdx.x == 0 && threadIdx.x == 0) b[7] = __float_as_uint(far_r2);
        for (int d = 0; d < 3; ++d) far_on = far_on || !(ord2f(b[d]) >= FK.c[d] - rf) || !(ord2f(b[3 + d]) <= FK.c[d] + rf);
        if (far_on)
            for (int d = 0; d < 3; ++d) lo[d] = FK.c[d] - rf, inv[d] = 512.f / rf;
    }
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    float x[RS_ITEMS], y[RS_ITEMS], z[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        const size_t j = 3 * (size_t)(i < n ? i : n - 1);
        x[k] = pts[j], y[k] = pts[j + 1], z[k] = pts[j + 2];
    }
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t i = base + k * RS_THREADS + threadIdx.x;
        if (i < n) {
            // 8 bits per axis are plenty for wave coherence: 3 radix passes instead of 4
            uint32_t key;
            if (CK.on) {
                const float xr = __builtin_fmaf(CK.r0[0], x[k], __builtin_fmaf(CK.r0[1], y[k], CK.r0[2] * z[k]));
                const float yr = __builtin_fmaf(CK.r1[0], x[k], __builtin_fmaf(CK.r1[1], y[k], CK.r1[2] * z[k]));
                const float fx = fminf(fmaxf((xr - CK.o0) * CK.s0, 0.f), 4095.f);  // (NaN -> 0)
                const float fy = fminf(fmaxf((yr - CK.o1) * CK.s1, 0.f), 4095.f);
                key = expand16((uint32_t)fx) | (expand16((uint32_t)fy) << 1);
            } else {
                key = morton30(x[k], y[k], z[k], lo, inv) >> key_shift;
                if (far_on) {
                    const uint32_t sh = far_shell(FK.c, far_r2, x[k], y[k], z[k]);
                    key >>= 1;  // near: the top bit stays clear
                    if (sh) {
                        ++nfar;
                        const float dx = x[k] - FK.c[0], dy = y[k] - FK.c[1], dz = z[k] - FK.c[2];
                        const float iv = 1.f / __builtin_sqrtf(__builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz)));
                        const float l1[3] = {-1.f, -1.f, -1.f}, i1[3] = {511.5f, 511.5f, 511.5f};
                        const uint32_t dm = morton30(dx * iv, dy * iv, dz * iv, l1, i1);  // 30 bits
                        const int kb = 30 - key_shift;                                     // key bits
                        key = (1u << (kb - 1)) | ((sh & 7u) << (kb - 4)) | (dm >> (34 - kb));
                    }
                }
            }
            keys[i] = key;
            atomicAdd(&h[key & 255u], 1u);
        }
    }
    __syncthreads();
    hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
    if (far_on) {  // how many queries sort into the far part: the far kernel's blocks beyond it leave at once
        for (int o = 32; o >= 1; o >>= 1) nfar += __shfl_xor(nfar, o);
        if ((threadIdx.x & 63u) == 0u && nfar) atomicAdd((uint32_t*)b + 6, nfar);
    }
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
// fast_nn: the groups are the runs of FAST_GROUP consecutive positions
__global__ void k_fast_grefs(uint32_t nv, uint32_t ngroups, uint32_t* __restrict__ grefs) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const uint32_t first = FAST_GROUP * g;
    const uint32_t cnt = nv - first < FAST_GROUP ? nv - first : FAST_GROUP;
    grefs[g] = KD_LEAF | ((first >> 2) << 4) | ((cnt + 3u) >> 2);
}

// =========================================================================================
// Oriented boxes of the wide packet walk (kernels_packet.h).  A box record is 16 dwords
//     { c.xyz, h0 | R0.xyz, h1 | R1.xyz, h2 | R2.xyz, tag }
// (R_a: axes, rows; h_a: half extents; tag: the group's ref), stored per parent node as
// [4][64] float4 so that lane k of the walk reads child k with unit stride.  The frame is the
// PCA frame of the child's vertices (a surface patch is flat along its smallest axis); it only
// has to be roughly orthonormal — what must hold exactly is  |R_a.(v - c)| <= h_a  for every
// vertex v of the child and the STORED fp32 c, R_a: h_a is computed against those in double and
// rounded up.  Anything non-finite gives the box that contains everything (h = +inf).
// =========================================================================================
__device__ __forceinline__ void obb_jacobi(double a00, double a01, double a02, double a11, double a12, double a22,
                                           double V[3][3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 8; ++sweep) {
#define SDFK_JROT(P, Q)                                                                            \
    if (fabs(A[P][Q]) > 1e-300) {                                                                  \
        const double th = (A[Q][Q] - A[P][P]) / (2.0 * A[P][Q]);                                   \
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));                \
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;                                       \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                            \
            const double akp = A[k][P], akq = A[k][Q];                                             \
            A[k][P] = c * akp - s * akq, A[k][Q] = s * akp + c * akq;                              \
        }                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                            \
            const double apk = A[P][k], aqk = A[Q][k];                                             \
            A[P][k] = c * apk - s * aqk, A[Q][k] = s * apk + c * aqk;                              \
        }                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                            \
            const double vkp = V[k][P], vkq = V[k][Q];                                             \
            V[k][P] = c * vkp - s * vkq, V[k][Q] = s * vkp + c * vkq;                              \
        }                                                                                          \
    }
        SDFK_JROT(0, 1)
        SDFK_JROT(0, 2)
        SDFK_JROT(1, 2)
#undef SDFK_JROT
    }
}
struct ObbFrame {
    float R[3][3];  // rows = axes (fp32, as stored)
};
__device__ __forceinline__ ObbFrame obb_frame(double n, const double s[3], const double ss[6]) {
    // covariance from raw moments: ss = {xx, xy, xz, yy, yz, zz}
    ObbFrame F;
    const double inv = n > 0 ? 1.0 / n : 0.0;
    const double mx = s[0] * inv, my = s[1] * inv, mz = s[2] * inv;
    double V[3][3];
    obb_jacobi(ss[0] * inv - mx * mx, ss[1] * inv - mx * my, ss[2] * inv - mx * mz, ss[3] * inv - my * my,
               ss[4] * inv - my * mz, ss[5] * inv - mz * mz, V);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) F.R[a][d] = (float)V[d][a];
    return F;
}
__device__ __forceinline__ void obb_store(float4* __restrict__ rec, uint32_t k, const ObbFrame& F, const double lo[3],
                                          const double hi[3], const float c[3], const double hmax[3], uint32_t tag,
                                          bool have_h, float4* __restrict__ aos = nullptr /* the same record, contiguous */) {
    (void)lo, (void)hi;
    float h[3];
    bool ok = have_h;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // rounded up with a margin: |R_a.(v - c)| <= h_a must hold for the fp32 record
        h[a] = (float)(hmax[a] * (1.0 + 1e-6) + 1e-37);
        h[a] = h[a] * (1.f + 1.2e-7f);
        ok = ok && h[a] >= 0.f && h[a] < 3.0e38f && fabsf(c[a]) < 3.0e38f;
#pragma unroll
        for (int d = 0; d < 3; ++d) ok = ok && fabsf(F.R[a][d]) <= 2.f;
    }
    float4 r0, r1, r2, r3;
    if (ok) {
        r0 = make_float4(c[0], c[1], c[2], h[0]);
        r1 = make_float4(F.R[0][0], F.R[0][1], F.R[0][2], h[1]);
        r2 = make_float4(F.R[1][0], F.R[1][1], F.R[1][2], h[2]);
        r3 = make_float4(F.R[2][0], F.R[2][1], F.R[2][2], __uint_as_float(tag));
    } else {  // contains everything: always visited
        const float inf = __builtin_inff();
        r0 = make_float4(0.f, 0.f, 0.f, inf);
        r1 = make_float4(1.f, 0.f, 0.f, inf);
        r2 = make_float4(0.f, 1.f, 0.f, inf);
        r3 = make_float4(0.f, 0.f, 1.f, __uint_as_float(tag));
    }
    rec[k] = r0, rec[64u + k] = r1, rec[128u + k] = r2, rec[192u + k] = r3;
    if (aos) aos[0] = r0, aos[1] = r1, aos[2] = r2, aos[3] = r3;
}

// one thread per packet group
__global__ __launch_bounds__(64) void k_group_obb(const uint32_t* __restrict__ grefs, uint32_t ngroups,
                                                  const float4* __restrict__ pts4, const int32_t* __restrict__ vind,
                                                  float4* __restrict__ gobb, float4* __restrict__ gobb_aos) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const uint32_t ref = grefs[g];
    const uint32_t first = ((ref & sdfk::KD_REF_MASK) >> 4) << 2, n4 = 4u * (ref & 15u);
    double n = 0, s[3] = {0, 0, 0}, ss[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t k = first; k < first + n4; ++k) {
        if (vind[k] < 0) continue;
        const float4 P = pts4[k];
        const double x = P.x, y = P.y, z = P.z;
        n += 1.0, s[0] += x, s[1] += y, s[2] += z;
        ss[0] += x * x, ss[1] += x * y, ss[2] += x * z, ss[3] += y * y, ss[4] += y * z, ss[5] += z * z;
    }
    const ObbFrame F = obb_frame(n, s, ss);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t k = first; k < first + n4; ++k) {
        if (vind[k] < 0) continue;
        const float4 P = pts4[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double pr = (double)F.R[a][0] * P.x + (double)F.R[a][1] * P.y + (double)F.R[a][2] * P.z;
            lo[a] = fmin(lo[a], pr), hi[a] = fmax(hi[a], pr);
        }
    }
    float c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        c[d] = (float)(0.5 * ((lo[0] + hi[0]) * F.R[0][d] + (lo[1] + hi[1]) * F.R[1][d] + (lo[2] + hi[2]) * F.R[2][d]));
    double hmax[3] = {0, 0, 0};
    for (uint32_t k = first; k < first + n4; ++k) {
        if (vind[k] < 0) continue;
        const float4 P = pts4[k];
        const double x = (double)P.x - c[0], y = (double)P.y - c[1], z = (double)P.z - c[2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            hmax[a] = fmax(hmax[a], fabs((double)F.R[a][0] * x + (double)F.R[a][1] * y + (double)F.R[a][2] * z));
    }
    obb_store(gobb + 256u * (size_t)(g >> 6), g & 63u, F, lo, hi, c, hmax, ref, n > 0, gobb_aos + 4u * (size_t)g);
}

// one wave per L1 node (run of 64 consecutive groups): lane k handles the vertices of group k
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
    for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__global__ __launch_bounds__(64) void k_l1_obb(const uint32_t* __restrict__ grefs, uint32_t ngroups,
                                               const float4* __restrict__ pts4, const int32_t* __restrict__ vind,
                                               float4* __restrict__ l1obb) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    const uint32_t g = 64u * j + lane;
    uint32_t first = 0, n4 = 0;
    if (g < ngroups) {
        const uint32_t ref = grefs[g];
        first = ((ref & sdfk::KD_REF_MASK) >> 4) << 2, n4 = 4u * (ref & 15u);
    }
    double n = 0, s[3] = {0, 0, 0}, ss[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t k = first; k < first + n4; ++k) {
        if (vind[k] < 0) continue;
        const float4 P = pts4[k];
        const double x = P.x, y = P.y, z = P.z;
        n += 1.0, s[0] += x, s[1] += y, s[2] += z;
        ss[0] += x * x, ss[1] += x * y, ss[2] += x * z, ss[3] += y * y, ss[4] += y * z, ss[5] += z * z;
    }
    n = wave_sum_d(n);
#pragma unroll
    for (int d = 0; d < 3; ++d) s[d] = wave_sum_d(s[d]);
#pragma unroll
    for (int d = 0; d < 6; ++d) ss[d] = wave_sum_d(ss[d]);
    const ObbFrame F = obb_frame(n, s, ss);  // every lane, identically
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t k = first; k < first + n4; ++k) {
        if (vind[k] < 0) continue;
        const float4 P = pts4[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double pr = (double)F.R[a][0] * P.x + (double)F.R[a][1] * P.y + (double)F.R[a][2] * P.z;
            lo[a] = fmin(lo[a], pr), hi[a] = fmax(hi[a], pr);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = wave_min_d(lo[a]), hi[a] = wave_max_d(hi[a]);
    float c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
        c[d] = (float)(0.5 * ((lo[0] + hi[0]) * F.R[0][d] + (lo[1] + hi[1]) * F.R[1][d] + (lo[2] + hi[2]) * F.R[2][d]));
    double hmax[3] = {0, 0, 0};
    for (uint32_t k = first; k < first + n4; ++k) {
        if (vind[k] < 0) continue;
        const float4 P = pts4[k];
        const double x = (double)P.x - c[0], y = (double)P.y - c[1], z = (double)P.z - c[2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            hmax[a] = fmax(hmax[a], fabs((double)F.R[a][0] * x + (double)F.R[a][1] * y + (double)F.R[a][2] * z));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) hmax[a] = wave_max_d(hmax[a]);
    if (lane == 0) obb_store(l1obb + 256u * (size_t)(j >> 6), j & 63u, F, lo, hi, c, hmax, j, n > 0);
}

// column grid build: count (or fill) the cells each face's rotated AABB touches
struct GridParams {
    int g;
    float o0, o1, inv0, inv1;
};
constexpr uint32_t GRID_BIG = 256;  // faces touching more cells go to the "big" list
constexpr uint32_t GRID_BIG_TRI = 16384;  // ... unless their triangle can be tested against the cells

// A face is listed in the cells its rotated AABB touches: a rigorous superset of the faces that pass
// Overlap for a query of the cell (kernels_query.h grid_cell).  For slivers (steep faces of a rough
// surface: the AABB is 10-100 x the triangle) that superset is long, and so is the reference's own
// candidate set; but a candidate only COUNTS when bary2d >= 0 as well (sdf.cpp:310), so a cell may drop
// a face whose bary2d test is certainly false for every query point p of the cell that passes Overlap.
// With a the first corner, v0 = b - a, v1 = c - a, v2 = p - a, den = v0 x v1 (sdf.hpp:130-142), u = 2^-24:
//   b1 = (v2 x v1) / den.  The fp32 numerator differs from the exact one by <= 4.5u |v2| |v1| (two
//   rounded products, one difference, rounded inputs), so its sign is the exact one whenever p is
//   farther than 4.5u |v2| from the line (a, c); the same for b2 and the line (a, b).  The sign of the
//   fp32 den is the exact one while kappa = |v0||v1| / |den| < 1 / (4.5u).
//   b0 = 1 - (b1 + b2) = dist(p, line bc) / h_a.  |b0_fp32 - b0| <= 16u kappa |v2| (|v0| + |v1|) / |den| + 2u,
//   i.e. b0_fp32 < 0 whenever p is farther outside (b, c) than 16u kappa |v2| (|v0| + |v1|) / |bc| + 2u h_a.
// |v2| <= diagonal of the AABB for a p that passes Overlap.  The cell test below dilates the triangle by
// 4 x the largest of these bounds plus the rounding slack of the cell function, and excludes a cell only
// when one triangle edge separates the dilated triangle from the whole cell rectangle.  Faces that are
// nearly degenerate (kappa > 1e4), tiny / huge, or non-finite keep the AABB rule.
struct TriCell {
    double nx[3], ny[3], off[3];  // outward unit normals of the edges, n . (point on the edge)
    double m;                     // dilation
    bool on;
};
// cr: the corners in the grid's two axes {a0, a1, b0, b1, c0, c1}; bw, bh: extent of the face's box in them
__device__ __forceinline__ TriCell tri_cell_setup(double bw, double bh, const float cr[6], double slack) {
    TriCell T;
    T.on = false;
    T.m = 0;
    const double ax = cr[0], ay = cr[1], bxx = cr[2], by = cr[3], cx = cr[4], cy = cr[5];
    const double v0x = bxx - ax, v0y = by - ay, v1x = cx - ax, v1y = cy - ay;
    const double den = v0x * v1y - v1x * v0y;
    const double l0 = sqrt(v0x * v0x + v0y * v0y), l1 = sqrt(v1x * v1x + v1y * v1y);
    const double ex = cx - bxx, ey = cy - by, lbc = sqrt(ex * ex + ey * ey);
    const double D = sqrt(bw * bw + bh * bh);
    const double aden = fabs(den);
    if (!(aden > 1e-30 && aden < 1e30 && l0 > 0 && l1 > 0 && lbc > 0 && D < 1e15)) return T;
    const double kappa = l0 * l1 / aden;
    if (!(kappa < 1e4)) return T;
    const double u = 5.9604644775390625e-8;
    const double m0 = 16.0 * u * kappa * D * (l0 + l1) / lbc + 2.0 * u * aden / lbc;
    const double m1 = 4.5 * u * D;
    T.m = 4.0 * (m0 > m1 ? m0 : m1) + slack;
    if (!(T.m < 1e30)) return T;
    const double sgn = den > 0 ? 1.0 : -1.0;  // counter-clockwise: outward normal of p0 -> p1 is (dy, -dx)
    const double px[3] = {ax, bxx, cx}, py[3] = {ay, by, cy};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int e1 = e == 2 ? 0 : e + 1;
        const double dx = px[e1] - px[e], dy = py[e1] - py[e], len = sqrt(dx * dx + dy * dy);
        T.nx[e] = sgn * dy / len, T.ny[e] = -sgn * dx / len;
        T.off[e] = T.nx[e] * px[e] + T.ny[e] * py[e];
    }
    T.on = true;
    return T;
}
// true: no query point of the rectangle [x0, x1] x [y0, y1] can have bary2d >= 0 for this face
__device__ __forceinline__ bool tri_cell_excluded(const TriCell& T, double x0, double x1, double y0, double y1) {
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const double mn = T.nx[e] * (T.nx[e] > 0 ? x0 : x1) + T.ny[e] * (T.ny[e] > 0 ? y0 : y1);
        if (mn - T.off[e] > T.m) return true;
    }
    return false;
}
// the corners of rotated face f in axes (A0, A1) of the ray-cast frame
template <int A0, int A1>
__device__ __forceinline__ void face_corners2(const float4* __restrict__ rf, uint32_t f, float cr[6]) {
    const float4 f0 = rf[3 * (size_t)f], f1 = rf[3 * (size_t)f + 1];
    const float a[3] = {f0.x, f0.y, f0.z}, b[3] = {f0.w, f1.x, f1.y};
    float c[3] = {f1.z, f1.w, 0.f};
    if (A0 == 2 || A1 == 2) c[2] = rf[3 * (size_t)f + 2].x;
    cr[0] = a[A0], cr[1] = a[A1], cr[2] = b[A0], cr[3] = b[A1], cr[4] = c[A0], cr[5] = c[A1];
}
// Axes (A0, A1) of the boxes / corners span the grid: (0, 1) for the z rays (and the renderer's image plane),
// (1, 2) for the x+ ray.
template <bool FILL, int A0, int A1>
__global__ void k_grid_faces(const float* __restrict__ fbox, uint32_t nf, GridParams G,
                             uint32_t* __restrict__ cells /* counts | cursors */,
                             uint32_t* __restrict__ items, uint32_t* __restrict__ big,
                             uint32_t* __restrict__ nbig, int skip_nan,
                             const float4* __restrict__ rface /* a' b' c' n' per face, or nullptr: boxes only */,
                             const float4* __restrict__ ent_src /* FILL: a' b' c' n' per face for the entries */,
                             float4* __restrict__ ent /* FILL: two float4 per entry (see MeshDev::gr_ent), or nullptr */) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const float* bf = fbox + 6 * (size_t)f;
    const float b[5] = {bf[A0], bf[A1], 0.f, bf[3 + A0], bf[3 + A1]};  // (min at 0, 1; max at 3, 4)
    // Renderer: a face whose image-space box has a NaN is never inserted (renderer.cpp:52-56)
    if (skip_nan && (b[0] != b[0] || b[1] != b[1] || b[3] != b[3] || b[4] != b[4])) return;
    const int x0 = grid_cell(b[0], G.o0, G.inv0, G.g), x1 = grid_cell(b[3], G.o0, G.inv0, G.g);
    const int y0 = grid_cell(b[1], G.o1, G.inv1, G.g), y1 = grid_cell(b[4], G.o1, G.inv1, G.g);
    if (x1 < x0 || y1 < y0) return;  // NaN box: the face can never pass Overlap
    const uint32_t nc = (uint32_t)(x1 - x0 + 1) * (uint32_t)(y1 - y0 + 1);
    TriCell T;
    T.on = false;
    const double wx = G.inv0 > 0.f ? 1.0 / (double)G.inv0 : 0.0, wy = G.inv1 > 0.f ? 1.0 / (double)G.inv1 : 0.0;
    if (rface && nc > 1u && wx > 0.0 && wy > 0.0) {
        // slack of the cell function (subtract, multiply, floor in fp32): a few ulps of the grid's span
        const double span = fabs((double)G.o0) + fabs((double)G.o1) + (double)G.g * (wx + wy);
        float cr[6];
        face_corners2<A0, A1>(rface, f, cr);
        T = tri_cell_setup((double)b[3] - b[0], (double)b[4] - b[1], cr, 1e-6 * span);
    }
    if (nc > (T.on ? GRID_BIG_TRI : GRID_BIG)) {
        if (!FILL) big[atomicAdd(nbig, 1u)] = f;
        return;
    }
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            if (T.on) {  // border cells take everything beyond the grid's range
                const double rx0 = x == 0 ? -1e30 : (double)G.o0 + x * wx, rx1 = x == G.g - 1 ? 1e30 : (double)G.o0 + (x + 1) * wx;
                const double ry0 = y == 0 ? -1e30 : (double)G.o1 + y * wy, ry1 = y == G.g - 1 ? 1e30 : (double)G.o1 + (y + 1) * wy;
                if (tri_cell_excluded(T, rx0, rx1, ry0, ry1)) continue;
            }
            const uint32_t c = (uint32_t)y * (uint32_t)G.g + (uint32_t)x;
            if (FILL) {
                const uint32_t slot = atomicAdd(&cells[c], 1u);
                items[slot] = f;
                if (ent) {  // the triangle in the grid's axes and the face id: what a ray needs before any per-face fetch
                    float cr[6];
                    face_corners2<A0, A1>(ent_src, f, cr);
                    ent[2 * (size_t)slot] = make_float4(cr[0], cr[1], cr[2], cr[3]);
                    ent[2 * (size_t)slot + 1] = make_float4(cr[4], cr[5], __uint_as_float(f), 0.f);
                }
            } else {
                atomicAdd(&cells[c], 1u);
            }
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
