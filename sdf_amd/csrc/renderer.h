// renderer.h — sdf::Renderer (include/sdf/sdf.hpp:282-369, src/renderer.cpp) on the GPU.
// Included at the end of sdfhip.hip (one translation unit: it shares the column grid, the scan,
// the device helpers and — for fill_outside — the whole nearest-vertex query path).
//
// The reference projects the vertices to the image plane (renderer.cpp:37-40), inserts every
// face's 2-D box into an RTree<int, float, 2> (42-59) and answers a 2-D point by searching the
// degenerate box [p, p] and running bary2d on every candidate (158-174).  What a point sees is
// therefore the SET of faces passing  closed-interval Overlap(box) && bary2d >= 0  — the same
// structure as the z rays of SDF::_raycast, so the same column grid serves it: one O(1) cell
// lookup per lane, then the reference's exact per-face test.
//   depth  = min over hits of z0*b0 + z1*b1 + z2*b2, FLT_MAX -> 0        (renderer.cpp:63-70, 193)
//   mask   = any hit                                                      (72-76)
//   vertex = the hit face vertex with the largest barycentric weight, lowest z over the hits (78-87)
// depth and mask are order independent.  The vertex is not when two DIFFERENT vertices of exactly
// equal z compete (the reference keeps the one its R-tree meets first); here the rule is
// order independent: lowest z, then lowest face id (mismatch rate: tests/test_gpu_renderer.py).
// fill_outside (196-204): the 2-D nearest projected vertex by nanoflann, ties included — served
// by an internal nearest-vertex handle over (x', y', 0): a 3-D nanoflann tree of points in the
// plane z = 0 is node for node the 2-D tree (the z span is 0, never the split dimension; the
// metric adds an exact 0), checked against the real DIM = 2 index in tests/test_oracle_renderer.py.

struct RendDev {
    const float4* rf;  // 64 B / face: a'.xy b'.xy | c'.xy z0 z1 | z2 mn.xy mx.x | mx.y i0 i1 i2
    const uint32_t* gr_start;
    const uint32_t* gr_items;
    const uint32_t* gr_big;
    uint32_t gr_nbig;
    int gr_g;
    float o0, o1, inv0, inv1;
    int width;
};

enum { R_DEPTH = 0, R_MASK = 1, R_VERTEX = 2 };

// verts_cam = ((x * fx) / z) + cx, ((y * fy) / z) + cy   (renderer.cpp:37-40, elementwise)
__global__ void k_rend_project(const float* __restrict__ verts, uint32_t nv, float fx, float fy, float cx,
                               float cy, float* __restrict__ cam) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float z = verts[3 * (size_t)i + 2];
    cam[2 * (size_t)i] = (verts[3 * (size_t)i] * fx) / z + cx;
    cam[2 * (size_t)i + 1] = (verts[3 * (size_t)i + 1] * fy) / z + cy;
}

// per-face record + 2-D box (cwiseMin / cwiseMax, renderer.cpp:46-51) + range of the finite boxes
__global__ void k_rend_faces(const float* __restrict__ verts, const float* __restrict__ cam,
                             const uint32_t* __restrict__ faces, uint32_t nf, float4* __restrict__ rf,
                             float* __restrict__ fbox, uint32_t* __restrict__ bounds) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const uint32_t i0 = faces[3 * (size_t)f], i1 = faces[3 * (size_t)f + 1], i2 = faces[3 * (size_t)f + 2];
    const float ax = cam[2 * (size_t)i0], ay = cam[2 * (size_t)i0 + 1];
    const float bx = cam[2 * (size_t)i1], by = cam[2 * (size_t)i1 + 1];
    const float cx = cam[2 * (size_t)i2], cy = cam[2 * (size_t)i2 + 1];
    float mnx = ax, mny = ay, mxx = ax, mxy = ay;
    mnx = sdfk::stdmin(mnx, bx), mny = sdfk::stdmin(mny, by);
    mnx = sdfk::stdmin(mnx, cx), mny = sdfk::stdmin(mny, cy);
    mxx = sdfk::stdmax(mxx, bx), mxy = sdfk::stdmax(mxy, by);
    mxx = sdfk::stdmax(mxx, cx), mxy = sdfk::stdmax(mxy, cy);
    rf[4 * (size_t)f] = make_float4(ax, ay, bx, by);
    rf[4 * (size_t)f + 1] = make_float4(cx, cy, verts[3 * (size_t)i0 + 2], verts[3 * (size_t)i1 + 2]);
    rf[4 * (size_t)f + 2] = make_float4(verts[3 * (size_t)i2 + 2], mnx, mny, mxx);
    rf[4 * (size_t)f + 3] = make_float4(mxy, __uint_as_float(i0), __uint_as_float(i1), __uint_as_float(i2));
    float* b = fbox + 6 * (size_t)f;
    b[0] = mnx, b[1] = mny, b[2] = 0.f, b[3] = mxx, b[4] = mxy, b[5] = 0.f;
    if (fabsf(mnx) <= FLT_MAX) atomicMin(&bounds[0], f2ord(mnx));
    if (fabsf(mny) <= FLT_MAX) atomicMin(&bounds[1], f2ord(mny));
    if (fabsf(mxx) <= FLT_MAX) atomicMax(&bounds[2], f2ord(mxx));
    if (fabsf(mxy) <= FLT_MAX) atomicMax(&bounds[3], f2ord(mxy));
}
__global__ void k_rend_bounds_init(uint32_t* b) {
    if (threadIdx.x < 2) b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 4) b[threadIdx.x] = 0u;
}

struct RendPix {
    float depth;
    bool mask;
    int vertex;
    float vz;
    uint32_t vface;
};

// one candidate face against one image-space point: Overlap of the degenerate box with the face
// box (RTree.h:1427-1438, closed intervals), bary2d (sdf.hpp:130-142), handlers (renderer.cpp:63-87)
template <int KIND>
__device__ __forceinline__ void rend_face(const RendDev& D, uint32_t f, float px, float py, RendPix& P) {
    const float4 r2 = D.rf[4 * (size_t)f + 2], r3 = D.rf[4 * (size_t)f + 3];
    if (px > r2.w || r2.y > px || py > r3.x || r2.z > py) return;
    const float4 r0 = D.rf[4 * (size_t)f], r1 = D.rf[4 * (size_t)f + 1];
    const float v0x = r0.z - r0.x, v0y = r0.w - r0.y, v1x = r1.x - r0.x, v1y = r1.y - r0.y;
    const float v2x = px - r0.x, v2y = py - r0.y;
    const float invden = 1.f / (v0x * v1y - v1x * v0y);
    const float b1 = (v2x * v1y - v1x * v2y) * invden;
    const float b2 = (v0x * v2y - v2x * v0y) * invden;
    const float b0 = 1.0f - (b1 + b2);
    if (!(b0 >= 0.f && b1 >= 0.f && b2 >= 0.f)) return;
    P.mask = true;
    if (KIND == R_DEPTH) {
        const float nd = (r1.z * b0 + r1.w * b1) + r2.x * b2;
        P.depth = sdfk::stdmin(P.depth, nd);
    } else if (KIND == R_VERTEX) {
        int cv = 0;  // first largest weight (strict >, renderer.cpp:80-83)
        float bb = b0;
        if (b1 > bb) cv = 1, bb = b1;
        if (b2 > bb) cv = 2;
        const int cand = (int)__float_as_uint(cv == 0 ? r3.y : (cv == 1 ? r3.z : r3.w));
        const float zc = cv == 0 ? r1.z : (cv == 1 ? r1.w : r2.x);
        if (P.vertex == -1 || zc < P.vz || (zc == P.vz && f < P.vface)) P.vertex = cand, P.vz = zc, P.vface = f;
    }
}

// pts2 == nullptr: image pixels, point i0 + i = ((float)(idx % width), (float)(idx / width))
// (renderer.cpp:187-189)
template <int KIND>
__global__ __launch_bounds__(256) void k_rend_query(RendDev D, const float* __restrict__ pts2, int64_t i0,
                                                    int64_t n, void* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float px, py;
    if (pts2) {
        px = pts2[2 * i], py = pts2[2 * i + 1];
    } else {
        const int64_t idx = i0 + i;
        px = (float)(idx % D.width), py = (float)(idx / D.width);
    }
    RendPix P = {FLT_MAX, false, -1, 0.f, 0u};
    const int cxl = grid_cell(px, D.o0, D.inv0, D.gr_g), cyl = grid_cell(py, D.o1, D.inv1, D.gr_g);
    const uint32_t c = (uint32_t)cyl * (uint32_t)D.gr_g + (uint32_t)cxl;
    const uint32_t e = D.gr_start[c + 1];
    for (uint32_t k = D.gr_start[c]; k < e; ++k) {
        rend_face<KIND>(D, D.gr_items[k], px, py, P);
        if (KIND == R_MASK && P.mask) break;  // renderer.cpp:72-76 stops at the first hit
    }
    if (!(KIND == R_MASK && P.mask))
        for (uint32_t k = 0; k < D.gr_nbig; ++k) rend_face<KIND>(D, D.gr_big[k], px, py, P);
    if (KIND == R_DEPTH)
        ((float*)out)[i] = P.depth == FLT_MAX ? 0.0f : P.depth;  // renderer.cpp:193-194
    else if (KIND == R_MASK)
        ((uint8_t*)out)[i] = P.mask ? 1 : 0;
    else
        ((int32_t*)out)[i] = P.vertex;
}

// fill_outside: queries of the internal nearest-vertex handle, (x, y, 0)
__global__ void k_rend_pts3(const float* __restrict__ pts2, int64_t i0, int64_t n, int width,
                            float* __restrict__ pts3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float px, py;
    if (pts2) {
        px = pts2[2 * i], py = pts2[2 * i + 1];
    } else {
        const int64_t idx = i0 + i;
        px = (float)(idx % width), py = (float)(idx / width);
    }
    pts3[3 * i] = px, pts3[3 * i + 1] = py, pts3[3 * i + 2] = 0.f;
}
__global__ void k_rend_fill(int32_t* __restrict__ vertex, const int32_t* __restrict__ nn, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (vertex[i] < 0) vertex[i] = nn[i];  // renderer.cpp:196-204 (data < 0)
}

struct sdfhip_renderer {
    int device = 0;
    bool own = false;
    uint32_t nv = 0, nf = 0;
    int width = 0, height = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    const float* h_verts = nullptr;
    const uint32_t* h_faces = nullptr;
    std::vector<float> owned_verts;
    std::vector<uint32_t> owned_faces;
    float* d_verts = nullptr;
    uint32_t* d_faces = nullptr;
    float* d_cam = nullptr;
    float4* d_rf = nullptr;
    float* d_fbox = nullptr;
    uint32_t* d_bounds = nullptr;
    GridBuf grid;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // fill_outside: built at first use after an update (renderer.cpp:110-113)
    sdfhip_ctx* nn = nullptr;
    bool nn_ready = false;
    std::vector<float> cam3;
    // staging of the host entry points and scratch of fill_outside
    float* d_pts2 = nullptr;
    uint32_t* d_out = nullptr;
    float* d_pts3 = nullptr;
    int32_t* d_nn = nullptr;
    int64_t stage_cap = 0, fill_cap = 0;
};

namespace {

constexpr int64_t REND_CHUNK = 1 << 24;

int rend_update(sdfhip_renderer* r) {
    hipStream_t s = r->stream;
    r->nn_ready = false;
    // the reference re-reads the faces too (renderer.cpp:42-59 re-inserts every face)
    for (size_t i = 0; i < 3 * (size_t)r->nf; ++i)
        if (r->h_faces[i] >= r->nv)
            return fail(SDFHIP_E_INVALID, "face index %u out of range at %zu", r->h_faces[i], i);
    if (r->nf)
        HIPC(hipMemcpyAsync(r->d_faces, r->h_faces, sizeof(uint32_t) * 3 * (size_t)r->nf, hipMemcpyHostToDevice, s));
    if (r->nv)
        HIPC(hipMemcpyAsync(r->d_verts, r->h_verts, sizeof(float) * 3 * (size_t)r->nv, hipMemcpyHostToDevice, s));
    if (r->nv)
        k_rend_project<<<(r->nv + 255) / 256, 256, 0, s>>>(r->d_verts, r->nv, r->fx, r->fy, r->cx, r->cy, r->d_cam);
    k_rend_bounds_init<<<1, 64, 0, s>>>(r->d_bounds);
    if (r->nf)
        k_rend_faces<<<(r->nf + 255) / 256, 256, 0, s>>>(r->d_verts, r->d_cam, r->d_faces, r->nf, r->d_rf,
                                                         r->d_fbox, r->d_bounds);
    HIPC(hipGetLastError());
    uint32_t b[4];
    HIPC(hipMemcpyAsync(b, r->d_bounds, sizeof(b), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f};
    if (b[0] != 0xFFFFFFFFu && b[2] != 0u) lo[0] = ord2f_host(b[0]), hi[0] = ord2f_host(b[2]);
    if (b[1] != 0xFFFFFFFFu && b[3] != 0u) lo[1] = ord2f_host(b[1]), hi[1] = ord2f_host(b[3]);
    // A vertex close to the camera plane projects far away and would stretch the grid until the
    // whole image is a single cell: when the boxes reach more than 4 images beyond the image, the
    // grid covers the image and half of it on every side (cells clamp: the faces and the points
    // outside share the border cells, the candidate sets stay supersets).
    const float ext[2] = {(float)(r->width > 0 ? r->width : 1), (float)(r->height > 0 ? r->height : 1)};
    for (int d = 0; d < 2; ++d)
        if (hi[d] - lo[d] > 9.0f * ext[d]) {
            lo[d] = lo[d] > -0.5f * ext[d] ? lo[d] : -0.5f * ext[d];
            hi[d] = hi[d] < 1.5f * ext[d] ? hi[d] : 1.5f * ext[d];
            if (!(hi[d] > lo[d])) lo[d] = -0.5f * ext[d], hi[d] = 1.5f * ext[d];
        }
    r->grid.nbig = 0, r->grid.entries = 0;
    if (int rc = grid_build(r->grid, r->d_fbox, r->nf, lo, hi, 1, s)) return rc;
    HIPC(hipStreamSynchronize(s));
    return 0;
}

RendDev rend_view(const sdfhip_renderer* r) {
    RendDev D;
    D.rf = r->d_rf;
    D.gr_start = r->grid.start;
    D.gr_items = r->grid.items;
    D.gr_big = r->grid.big;
    D.gr_nbig = r->nf ? r->grid.nbig : 0;
    D.gr_g = r->grid.gr.g;
    D.o0 = r->grid.gr.o0, D.o1 = r->grid.gr.o1, D.inv0 = r->grid.gr.inv0, D.inv1 = r->grid.gr.inv1;
    D.width = r->width > 0 ? r->width : 1;
    return D;
}

// the 2-D nearest-vertex index (lazy, like kd_tree.rebuild() at renderer.cpp:110-113)
int rend_ensure_nn(sdfhip_renderer* r) {
    if (r->nn_ready) return 0;
    std::vector<float> cam(2 * (size_t)r->nv);
    if (r->nv) HIPC(hipMemcpy(cam.data(), r->d_cam, sizeof(float) * cam.size(), hipMemcpyDeviceToHost));
    r->cam3.resize(3 * (size_t)r->nv);
    for (size_t i = 0; i < r->nv; ++i) {
        const float x = cam[2 * i], y = cam[2 * i + 1];
        if (!(std::fabs(x) <= FLT_MAX) || !(std::fabs(y) <= FLT_MAX))
            return fail(SDFHIP_E_INVALID,
                        "fill_outside needs finite projected vertices (vertex %zu has z = 0 or is not finite; "
                        "sdf.hpp:273 assumes nothing at z <= 0)", i);
        r->cam3[3 * i] = x, r->cam3[3 * i + 1] = y, r->cam3[3 * i + 2] = 0.f;
    }
    if (!r->nn) {
        if (int rc = sdfhip_create(r->cam3.data(), r->nv, nullptr, 0, 0, 0, r->device, &r->nn)) return rc;
    } else {
        r->nn->h_verts = r->cam3.data();
        if (int rc = sdfhip_update(r->nn)) return rc;
    }
    r->nn_ready = true;
    return 0;
}

// everything one chunk needs, on stream s; d_pts2 may be null (image pixels from index i0)
int rend_query_chunk(sdfhip_renderer* r, int kind, const float* d_pts2, int64_t i0, int64_t m, int fill,
                     void* d_out, hipStream_t s) {
    const RendDev D = rend_view(r);
    const uint32_t blocks = (uint32_t)((m + 255) / 256);
    if (kind == R_DEPTH)
        k_rend_query<R_DEPTH><<<blocks, 256, 0, s>>>(D, d_pts2, i0, m, d_out);
    else if (kind == R_MASK)
        k_rend_query<R_MASK><<<blocks, 256, 0, s>>>(D, d_pts2, i0, m, d_out);
    else
        k_rend_query<R_VERTEX><<<blocks, 256, 0, s>>>(D, d_pts2, i0, m, d_out);
    HIPC(hipGetLastError());
    if (kind == R_VERTEX && fill && r->nv) {
        if (m > r->fill_cap) {
            if (int rc = dalloc(&r->d_pts3, 3 * (size_t)m)) return rc;
            if (int rc = dalloc(&r->d_nn, (size_t)m)) return rc;
            r->fill_cap = m;
        }
        k_rend_pts3<<<blocks, 256, 0, s>>>(d_pts2, i0, m, D.width, r->d_pts3);
        if (int rc = run_query_dev(r->nn, Q_NN, r->d_pts3, m, 0, r->d_nn, s)) return rc;
        k_rend_fill<<<blocks, 256, 0, s>>>((int32_t*)d_out, r->d_nn, m);
        HIPC(hipGetLastError());
    }
    return 0;
}

int rend_check(sdfhip_renderer* r, int kind, int64_t n, const void* out) {
    if (!r) return fail(SDFHIP_E_INVALID, "null renderer");
    if (kind < 0 || kind > 2) return fail(SDFHIP_E_INVALID, "unknown renderer query kind %d", kind);
    if (n < 0) return fail(SDFHIP_E_INVALID, "negative point count");
    if (n > 0 && !out) return fail(SDFHIP_E_INVALID, "null output pointer");
    return 0;
}

// host pointers; pts2 == nullptr renders the image (n must be width * height)
int rend_query_host(sdfhip_renderer* r, int kind, const float* pts2, int64_t n, int fill, void* out) {
    if (int rc = rend_check(r, kind, n, out)) return rc;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(r->mu);
    DevGuard g(r->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", r->device);
    if (kind == R_VERTEX && fill)
        if (int rc = rend_ensure_nn(r)) return rc;
    const int64_t cap = n < REND_CHUNK ? n : REND_CHUNK;
    if (cap > r->stage_cap) {
        if (int rc = dalloc(&r->d_pts2, 2 * (size_t)cap)) return rc;
        if (int rc = dalloc(&r->d_out, (size_t)cap)) return rc;
        r->stage_cap = cap;
    }
    const size_t osz = kind == R_MASK ? 1 : 4;
    hipStream_t s = r->stream;
    for (int64_t off = 0; off < n; off += cap) {
        const int64_t m = (n - off) < cap ? (n - off) : cap;
        if (pts2)
            HIPC(hipMemcpyAsync(r->d_pts2, pts2 + 2 * off, sizeof(float) * 2 * (size_t)m, hipMemcpyHostToDevice, s));
        if (int rc = rend_query_chunk(r, kind, pts2 ? r->d_pts2 : nullptr, off, m, fill, r->d_out, s)) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
        HIPC(hipMemcpyAsync((char*)out + osz * (size_t)off, r->d_out, osz * (size_t)m, hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
    }
    return 0;
}

}  // namespace

extern "C" {

static int renderer_create_impl(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int width,
                                int height, float fx, float fy, float cx, float cy, int copy, int device,
                                sdfhip_renderer_t** out) {
    if ((nv && !verts) || (nf && !faces)) return fail(SDFHIP_E_INVALID, "null verts/faces");
    if (width < 0 || height < 0) return fail(SDFHIP_E_INVALID, "negative image size");
    if (nv >= sdfk::KD_MAX_POINTS) return fail(SDFHIP_E_INVALID, "too many vertices (max %u)", sdfk::KD_MAX_POINTS);
    if (nf >= (1u << 30)) return fail(SDFHIP_E_INVALID, "too many faces");
    for (size_t i = 0; i < 3 * (size_t)nf; ++i)
        if (faces[i] >= nv) return fail(SDFHIP_E_INVALID, "face index %u out of range at %zu", faces[i], i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SDFHIP_E_NODEVICE, "no HIP device available (there is no CPU fallback)");
    if (device < 0)
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    if (device >= ndev) return fail(SDFHIP_E_INVALID, "device %d out of range (%d devices)", device, ndev);
    DevGuard g(device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", device);
    sdfhip_renderer* r = new sdfhip_renderer();
    *out = r;  // the caller destroys it if anything below fails or throws
    r->device = device, r->own = copy != 0, r->nv = nv, r->nf = nf;
    r->width = width, r->height = height, r->fx = fx, r->fy = fy, r->cx = cx, r->cy = cy;
    if (copy) {
        r->owned_verts.assign(verts, verts + 3 * (size_t)nv);
        r->owned_faces.assign(faces, faces + 3 * (size_t)nf);
        r->h_verts = r->owned_verts.data(), r->h_faces = r->owned_faces.data();
    } else {
        r->h_verts = verts, r->h_faces = faces;
    }
    int rc = 0;
    auto A = [&](int x) { if (!rc) rc = x; };
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(SDFHIP_E_HIP, "hipStreamCreate failed");
    A(dalloc(&r->d_verts, 3 * (size_t)nv));
    A(dalloc(&r->d_faces, 3 * (size_t)nf));
    A(dalloc(&r->d_cam, 2 * (size_t)nv));
    A(dalloc(&r->d_rf, 4 * (size_t)nf));
    A(dalloc(&r->d_fbox, 6 * (size_t)nf));
    A(dalloc(&r->d_bounds, 4));
    if (!rc) rc = rend_update(r);
    return rc;
}

int sdfhip_renderer_create(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int width,
                           int height, float fx, float fy, float cx, float cy, int copy, int device,
                           sdfhip_renderer_t** out) {
    if (!out) return fail(SDFHIP_E_INVALID, "out is null");
    *out = nullptr;
    const int rc = guarded([&]() -> int {
        return renderer_create_impl(verts, nv, faces, nf, width, height, fx, fy, cx, cy, copy, device, out);
    });
    if (rc && *out) {
        const std::string keep = g_err;
        sdfhip_renderer_destroy(*out);
        g_err = keep;
        *out = nullptr;
    }
    return rc;
}

void sdfhip_renderer_destroy(sdfhip_renderer_t* r) {
    if (!r) return;
    DevGuard g(r->device);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    if (r->nn) sdfhip_destroy(r->nn);
    void* ptrs[] = {r->d_verts, r->d_faces, r->d_cam, r->d_rf, r->d_fbox, r->d_bounds,
                    r->d_pts2,  r->d_out,   r->d_pts3, r->d_nn};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    r->grid.release();
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

int sdfhip_renderer_update(sdfhip_renderer_t* r) {
    if (!r) return fail(SDFHIP_E_INVALID, "null renderer");
    std::lock_guard<std::mutex> lk(r->mu);
    DevGuard g(r->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", r->device);
    return guarded([&]() -> int { return rend_update(r); });
}

int sdfhip_renderer_render_depth(sdfhip_renderer_t* r, float* out) {
    return guarded([&]() -> int { return rend_query_host(r, R_DEPTH, nullptr, r ? (int64_t)r->width * r->height : 0, 0, out); });
}
int sdfhip_renderer_render_mask(sdfhip_renderer_t* r, uint8_t* out) {
    return guarded([&]() -> int { return rend_query_host(r, R_MASK, nullptr, r ? (int64_t)r->width * r->height : 0, 0, out); });
}
int sdfhip_renderer_render_nn(sdfhip_renderer_t* r, int fill_outside, int32_t* out) {
    return guarded([&]() -> int { return rend_query_host(r, R_VERTEX, nullptr, r ? (int64_t)r->width * r->height : 0, fill_outside, out); });
}
int sdfhip_renderer_depth(sdfhip_renderer_t* r, const float* pts2, int64_t n, float* out) {
    if (n > 0 && !pts2) return fail(SDFHIP_E_INVALID, "null points pointer");
    return guarded([&]() -> int { return rend_query_host(r, R_DEPTH, pts2, n, 0, out); });
}
int sdfhip_renderer_contains(sdfhip_renderer_t* r, const float* pts2, int64_t n, uint8_t* out) {
    if (n > 0 && !pts2) return fail(SDFHIP_E_INVALID, "null points pointer");
    return guarded([&]() -> int { return rend_query_host(r, R_MASK, pts2, n, 0, out); });
}
int sdfhip_renderer_nn(sdfhip_renderer_t* r, const float* pts2, int64_t n, int fill_outside, int32_t* out) {
    if (n > 0 && !pts2) return fail(SDFHIP_E_INVALID, "null points pointer");
    return guarded([&]() -> int { return rend_query_host(r, R_VERTEX, pts2, n, fill_outside, out); });
}

int sdfhip_renderer_query_device(sdfhip_renderer_t* r, int kind, const float* d_pts2, int64_t n,
                                 int fill_outside, void* d_out, void* stream) {
    if (int rc = rend_check(r, kind, n, d_out)) return rc;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(r->mu);
    DevGuard g(r->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", r->device);
    if (kind == R_VERTEX && fill_outside)
        if (int rc = rend_ensure_nn(r)) return rc;
    const size_t osz = kind == R_MASK ? 1 : 4;
    for (int64_t off = 0; off < n; off += REND_CHUNK) {
        const int64_t m = (n - off) < REND_CHUNK ? (n - off) : REND_CHUNK;
        if (int rc = rend_query_chunk(r, kind, d_pts2 ? d_pts2 + 2 * off : nullptr, off, m, fill_outside,
                                      (char*)d_out + osz * (size_t)off, (hipStream_t)stream))
            return rc;
    }
    return 0;
}

int sdfhip_renderer_own_data(const sdfhip_renderer_t* r) { return r ? r->own : 0; }
int sdfhip_renderer_device(const sdfhip_renderer_t* r) { return r ? r->device : -1; }
int sdfhip_renderer_width(const sdfhip_renderer_t* r) { return r ? r->width : 0; }
int sdfhip_renderer_height(const sdfhip_renderer_t* r) { return r ? r->height : 0; }
uint32_t sdfhip_renderer_num_verts(const sdfhip_renderer_t* r) { return r ? r->nv : 0; }
uint32_t sdfhip_renderer_num_faces(const sdfhip_renderer_t* r) { return r ? r->nf : 0; }
const float* sdfhip_renderer_verts(const sdfhip_renderer_t* r) { return r ? r->h_verts : nullptr; }
const uint32_t* sdfhip_renderer_faces(const sdfhip_renderer_t* r) { return r ? r->h_faces : nullptr; }
float* sdfhip_renderer_verts_mutable(sdfhip_renderer_t* r) {
    if (!r) return nullptr;
    if (!r->own) {
        fail(SDFHIP_E_STATE, "'verts' is non mutable, construct with copy=True");
        return nullptr;
    }
    return r->owned_verts.data();
}
uint32_t* sdfhip_renderer_faces_mutable(sdfhip_renderer_t* r) {
    if (!r) return nullptr;
    if (!r->own) {
        fail(SDFHIP_E_STATE, "'faces' is non mutable, construct with copy=True");
        return nullptr;
    }
    return r->owned_faces.data();
}

}  // extern "C"
