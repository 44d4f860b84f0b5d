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
//   kd_gobb    64 B / packet group and kd_l1obb 64 B / run of 64 groups: oriented boxes of the wide
//              packet walk (lane k tests child k against the packet of 64 queries)
// One translation unit, in parts (included in this order):
//   kernels_query.h   device views, strict (nanoflann-exact) search, ray parity, k_query
//   kernels_packet.h  packet pass (k_query_packet, the dominant kernel), k_fb_compact, k_query_heavy
//   kernels_update.h  sample_surface, per-face precompute
//   radix_sort.h      LDS radix sort + scan
//   kernels_build.h   Morton codes, bounds, LBVH, fast_nn tree, column grid fill
//   (this file)       handle, update(), query pipeline, host entry points, C ABI
//   renderer.h        sdf::Renderer
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

// The C ABI never throws: host-side allocation (std::vector, std::thread) can.
template <class F>
static int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(SDFHIP_E_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(SDFHIP_E_STATE, "internal error: %s", e.what());
    } catch (...) {
        return fail(SDFHIP_E_STATE, "internal error");
    }
}

#include "kernels_query.h"

#include "kernels_packet.h"

#include "kernels_update.h"

#include "radix_sort.h"

#include "kernels_build.h"

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
    float4* ent = nullptr;       // per entry of items: the face's corners in x, y + its id, 32 B (only when want_ent)
    bool want_ent = false;
    uint32_t* big = nullptr;
    uint32_t* nbig_dev = nullptr;
    uint32_t* s1 = nullptr;
    uint32_t* s2 = nullptr;
    size_t cells_cap = 0, items_cap = 0, big_cap = 0;
    uint32_t nbig = 0, entries = 0;
    bool coarsened = false;  // faces are large against the cells at the natural resolution
    GridParams gr = {1, 0.f, 0.f, 0.f, 0.f};
    void release() {
        void* ps[] = {start, cursor, items, ent, big, nbig_dev, s1, s2};
        for (void* p : ps)
            if (p) (void)hipFree(p);
        start = cursor = items = big = nbig_dev = s1 = s2 = nullptr;
        ent = nullptr;
        cells_cap = items_cap = big_cap = 0;
    }
};

// Device workspace of one query call: Morton keys / permutation of the query sort (+ the sorter's
// alternates and histograms), the per-wave masks of deferred queries, the deferred / heavy lists.
constexpr uint32_t HEAVY_SHORT = 8192;  // heavy lists up to this length: the many-waves-per-query scan + k_scan_finish
struct QueryWs {
    uint32_t* qkeys = nullptr;
    uint32_t* qperm = nullptr;
    uint32_t* qbounds = nullptr;
    unsigned long long* amb_mask = nullptr;
    uint32_t* amb_cnt = nullptr;
    uint32_t* amb_s1 = nullptr;
    uint32_t* amb_s2 = nullptr;
    uint32_t* fb_list = nullptr;   // [0, cap): deferred near-tie queries; [cap, 2 cap): heavy-wave queries
    uint32_t* fb_count = nullptr;  // [0], [1] length of the deferred list (ping-pong, see k_fb_compact), [2] heavy list,
                                   // [3] packets handed to the cooperative kernel, [4..7] its CoopDesc
    int32_t* heavy_pos = nullptr;    // [HEAVY_SHORT] tree positions of a short heavy list (k_scan_finish)
    uint32_t* coop_list = nullptr;   // [COOP_CAP] (null in the static scratch: no cooperative kernel there)
    uint32_t* coop_state = nullptr;  // [COOP_CAP][7][64]
    // cube lists of the running launch (k_cube_lists); none in the static scratch
    float* cube_par = nullptr;
    float4* cube_hdr = nullptr;
    uint32_t* cube_list = nullptr;
    size_t cubes_cap = 0;
    int64_t q_cap = 0, fb_cap = 0;
    RadixSorter sorter;
    bool fixed = false;  // carved out of the static scratch array: never grown, never freed
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
    uint32_t* d_kd_gref = nullptr;  // packet groups in position order (refs)
    float4* d_kd_gobb = nullptr;    // their oriented boxes, [n_l1][4][64]
    float4* d_kd_gobb_aos = nullptr;  // the same records, 64 B per group in group order (cube lists gather from it)
    float4* d_kd_l1obb = nullptr;   // boxes of the runs of 64 groups, [rows][4][64]
    size_t kd_groups_cap = 0;
    uint32_t n_groups = 0, n_l1 = 0;
    uint32_t* d_kd_split = nullptr;
    int32_t* d_kd_vind = nullptr;
    size_t kd_nodes_cap = 0, kd_pos_cap = 0;
    float4* d_kd_pts4 = nullptr;
    float4* d_kd_apx4 = nullptr;  // the approximate pass's group-centred copy (k_apx_fill) ...
    float4* d_kd_actr = nullptr;  // ... the groups' centres, one per 4 positions ...
    uint32_t* d_apx_r2 = nullptr; // ... and [0] the largest squared group radius (bits of a float), [1] != 0: unusable
    uint32_t apx_r2_bits[4] = {0, 0, 0, 0};
    bool apx_usable = false;      // false: a vertex the pass cannot represent (k_apx_fill); the exact walk is taken
    float apx_k1 = 0.f, apx_e0 = 0.f;  // error model of the pass: E(est) = apx_k1 est + apx_e0 (fill_apx)
    float4* d_kd_lbox = nullptr;  // per position: its leaf's tight box (k_leaf_boxes, kd_path_clear)
    size_t kd_lbox_cap = 0, kd_apx_cap = 0;  // (capacities of their own: the GPU-built tree of fast_nn has neither array)
    // the strict tree the device arrays currently hold — built by this handle's own update, or by the leader of
    // its group and copied device to device (group.h): depth, root ref, padded positions, internal nodes
    int32_t kd_depth = 0;
    uint32_t kd_root = 0, kd_npos = 0, kd_nnodes = 0;
    int64_t kd_builds = 0;  // host kd-tree builds this handle has run (sdfhip_stats_t::kd_builds)
    float amb_c = 1.f, amb_c2 = 1.f;
    QueryWs ws;            // this handle's own query workspace, grown on demand
    QueryWs* W = &ws;      // the workspace of the running call (the static scratch in the no-alloc entry)
    uint32_t* h_counts = nullptr;  // page-locked: [0] length of the last deferred list, [1] queries of the
                                   // last call's heavy waves (read back only while profiling)
    int64_t opt_heavy_budget = 0;  // chunks per wave before it gives up; 0 = automatic
    int64_t opt_coop_budget = -1;  // work units after which a packet goes on in the cooperative kernel; 0 = never,
                                   // -1 = automatic: 1500, less for launches of < 1M points (see run_query_t)
    int64_t opt_coop_max = 1 << 27;  // launches of more points than this keep their long waves (default: none does)
    int64_t opt_scan_max = -1;  // n x positions up to which a batch is answered by the exact scan; -1 = automatic
    int64_t opt_packet = 1;
    int64_t opt_cube_lists = -1;   // packet pass, cube lists: -1 automatic (launches of >= opt_cube_min points), 0 off, 1 on
    int64_t opt_cube_min = 8 << 20;
    int64_t opt_cube_level = 0;    // cubes per axis = 2^level; 0 automatic (~4096 queries per cube)
    int64_t opt_far = 1;      // packet pass: far packets (k_query_packet_far); option "far_packets"
    int64_t opt_far_sparse_x10 = 20;  // ... from this many tenths on in a sparse batch; option "far_radius_sparse_x10"
    int64_t opt_far_x10 = 120; // ... from this many tenths of the mesh's half diagonal on; option "far_radius_x10"
    int64_t opt_approx = -1;  // packet pass, approximate first pass: -1 automatic (approx_ok), 0 off, 1 on
    bool approx_ok = false;   // decided by every update(): see approx_pays()
    bool approx_pre = false;  // ... its value when the tree was uploaded (fill_apx may still turn it off; read by followers)
    int64_t opt_packet_group = 0;  // vertices per packet group: 0 automatic (40 with the approximate pass, else 24)
    int64_t opt_build_threads = 1;
    uint32_t* d_adj_off = nullptr;
    int32_t* d_adj_faces = nullptr;
    float4* d_geo = nullptr;
    float4* d_geo_adj = nullptr;  // the same records per (vertex, incident face), 64 B x 3F
    float4* d_rface = nullptr;
    float* d_area_cum = nullptr;  // running sum of face areas (sample_surface), valid iff cum_valid
    bool cum_valid = false;
    float total_area = 0.f;
    float* d_fbox = nullptr;
    // column grid (rebuilt by every update: face boxes move)
    GridBuf grid;
    GridBuf grid_x;  // over (y', z'): the x+ tie-break ray (opt_x_grid)
    int64_t opt_x_grid = 1;
    // fast_nn mode (GPU-built vertex tree)
    int64_t opt_fast_nn = 0;   // applies at the next update
    bool fast_active = false;  // what the structures on the device currently are
    uint32_t* d_vkeys = nullptr;
    uint32_t* d_vidx = nullptr;
    size_t vkeys_cap = 0;
    int64_t opt_grid_per_face = 16;  // column grid: halve the resolution while the lists exceed this many entries per face
    int64_t opt_grid_tri = 1;  // column grid: cells drop faces whose triangle cannot contain their points
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
    bool lbvh_valid = false;   // rnodes hold the current face boxes (built only when something walks the LBVH)
    bool xgrid_built = false;  // grid_x is current (the x+ ray reads it)
    float topo_R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // rotation the LBVH topology was built for
    uint32_t* d_gstack = nullptr;  // strict kernel, trees deeper than LDS_DEPTH_MAX
    size_t gstack_words = 0;
    // options
    int64_t opt_sort = 1, opt_sort_min = 8192, opt_chunk = 1 << 27;
    int64_t opt_sort_bits = 24;  // Morton key bits of the query sort (24: three radix passes, 30: four)
    int64_t opt_sort_min_contains = -1;  // robust contains(): -1 automatic, see run_query_t
    int64_t opt_column_sort = 1;  // robust contains(): sort by column of the ray-cast frame (0: 3-D Morton order)
    hipStream_t stream = nullptr;  // internal stream for create/update and host entry points
    // host entry points: double-buffered device staging, copies on their own streams
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    float* d_stage_in[2] = {nullptr, nullptr};
    // page-locked bounce buffers for pageable caller memory (see BounceCopy): two per direction
    unsigned char* pin_up[2] = {nullptr, nullptr};
    unsigned char* pin_dn[2] = {nullptr, nullptr};
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_dn[2] = {nullptr, nullptr};
    std::unique_ptr<sdfk::KdTeam> team_up, team_dn;  // 3 helpers each: one thread copies 32 GB/s, four 62 GB/s
    void* d_stage_out[2] = {nullptr, nullptr};
    int64_t stage_cap = 0;     // points per staging slot
    int64_t opt_host_chunk = 0;  // 0 = automatic
    std::mutex mu;
    // Device-side ordering between calls.  Host-side calls are serialised by `mu`, but the
    // `_device` queries are asynchronous on the CALLER's stream and share the handle's workspace
    // and structures: the next call on another stream (a query, or update() on `stream`) first
    // waits for this event.
    hipEvent_t ev_query = nullptr;
    hipStream_t last_query_stream = nullptr;
    bool query_pending = false;
    std::mt19937 rng;
    float last_update_ms = 0.f, last_kd_ms = 0.f;
    int last_kernel = SDFHIP_KERNEL_NONE;  // dominant kernel of the last query call (sdfhip_stats_t::last_kernel)
    // optional live profiling of the query kernels (hipEvents on the launch stream)
    int64_t opt_profile = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending, prof_free;
    double prof_ms = 0.0;
    int64_t prof_launches = 0, prof_points = 0;
};

// =========================================================================================
// Static scratch of the allocation-free entry point (sdfhip_sdf_device_noalloc).
// Every GPU the library runs on holds its own copy of this array, at its own address, until the
// process ends.  In host code the array's NAME is only a stand-in, not a device address: the
// address is taken once per device with hipGetSymbolAddress (that device current) and host code
// uses that pointer everywhere — kernel arguments, hipMemsetAsync.  All calls on one GPU share
// the array and it keeps the last call's contents: the pipeline clears what must start at zero
// (the three list counters), and calls that overlap on one GPU race.
// =========================================================================================
constexpr int64_t NOALLOC_MAX_POINTS = SDFHIP_NOALLOC_MAX_POINTS;
constexpr size_t SCRATCH_BYTES = (size_t)26 << 20;  // 24 B per point + histograms, see scratch_carve
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
QueryWs* g_scratch_ws[MAX_DEVICES] = {};  // per device; built once, never freed (static storage behind it)

// lays the workspace of a query call of up to NOALLOC_MAX_POINTS points out in `base`
int scratch_carve(unsigned char* base, QueryWs* W) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base + off;
        off = (off + bytes + 255) & ~(size_t)255;
        return p;
    };
    const size_t n = (size_t)NOALLOC_MAX_POINTS, nw = n / 64 + 4;
    const size_t ntiles = (n + RS_TILE - 1) / RS_TILE, nh = 256 * ntiles;
    const size_t n1 = (nh + 2047) / 2048, n2 = (n1 + 2047) / 2048;
    W->qkeys = (uint32_t*)take(4 * n);
    W->qperm = (uint32_t*)take(4 * n);
    W->sorter.keys_alt = (uint32_t*)take(4 * n);
    W->sorter.vals_alt = (uint32_t*)take(4 * n);
    W->sorter.hist = (uint32_t*)take(4 * nh);
    W->sorter.sums1 = (uint32_t*)take(4 * (n1 + 1));
    W->sorter.sums2 = (uint32_t*)take(4 * (n2 + 1));
    W->sorter.cap = (uint32_t)n;
    W->fb_list = (uint32_t*)take(4 * 2 * n);
    W->amb_mask = (unsigned long long*)take(8 * nw);
    W->amb_cnt = (uint32_t*)take(4 * nw);
    W->amb_s1 = (uint32_t*)take(4 * (nw / 2048 + 4));
    W->amb_s2 = (uint32_t*)take(4 * (nw / 2048 / 2048 + 4));
    W->qbounds = (uint32_t*)take(4 * 8);
    W->fb_count = (uint32_t*)take(4 * 8);
    W->heavy_pos = (int32_t*)take(4 * (HEAVY_SHORT + 64));
    W->q_cap = W->fb_cap = (int64_t)n;
    W->fixed = true;
    if (off > SCRATCH_BYTES) return fail(SDFHIP_E_STATE, "static scratch too small (%zu > %zu bytes)", off, SCRATCH_BYTES);
    return 0;
}

// once per device, with that device current (called from sdfhip_create)
int scratch_setup(int device) {
    if (device < 0 || device >= MAX_DEVICES) return 0;  // such a device gets no no-alloc entry
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ws[device]) return 0;
    unsigned char* ptr = nullptr;
    HIPC(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(SDFHIP_E_HIP, "hipGetSymbolAddress(scratch) returned a null pointer");
    QueryWs* W = new QueryWs();
    if (int rc = scratch_carve(ptr, W)) {
        W->sorter = RadixSorter();  // (nothing of it is owned)
        delete W;
        return rc;
    }
    g_scratch_ws[device] = W;
    return 0;
}

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
               int skip_nan, hipStream_t s, const float4* d_rface = nullptr, int64_t max_per_face = 16,
               const float4* d_ent_src = nullptr /* the rotated faces, when B.want_ent */, bool yz = false) {
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
        if (yz)
            k_grid_faces<false, 1, 2><<<(n + 255) / 256, 256, 0, s>>>(d_fbox, n, G, B.start, nullptr, B.big,
                                                                      B.nbig_dev, skip_nan, d_rface, nullptr, nullptr);
        else
            k_grid_faces<false, 0, 1><<<(n + 255) / 256, 256, 0, s>>>(d_fbox, n, G, B.start, nullptr, B.big,
                                                                      B.nbig_dev, skip_nan, d_rface, nullptr, nullptr);
        exclusive_scan_u32(B.start, (uint32_t)cells, B.s1, B.s2, s);
        uint32_t total = 0, nbig = 0;
        HIPC(hipMemcpyAsync(&total, B.start + (cells - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPC(hipMemcpyAsync(&nbig, B.nbig_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        if (g == g_first) long_lists = (size_t)total + 256 * (size_t)nbig > 10 * (size_t)n;
        // many faces much larger than a cell: a coarser grid keeps the lists (and memory) short
        if (((size_t)total > (size_t)max_per_face * (size_t)n + 4096 || (size_t)nbig > (size_t)n / 64 + 16) && g > 1) {
            g = g / 2;
            continue;
        }
        if ((size_t)total + 1 > B.items_cap) {
            const size_t cap = (size_t)total + (size_t)total / 4 + 1024;
            if (int rc = dalloc(&B.items, cap)) return rc;
            if (B.want_ent)
                if (int rc = dalloc(&B.ent, 2 * cap)) return rc;
            B.items_cap = cap;
        }
        HIPC(hipMemcpyAsync(B.cursor, B.start, sizeof(uint32_t) * cells, hipMemcpyDeviceToDevice, s));
        if (yz)
            k_grid_faces<true, 1, 2><<<(n + 255) / 256, 256, 0, s>>>(d_fbox, n, G, B.cursor, B.items, B.big, B.nbig_dev,
                                                                     skip_nan, d_rface, d_ent_src, B.want_ent ? B.ent : nullptr);
        else
            k_grid_faces<true, 0, 1><<<(n + 255) / 256, 256, 0, s>>>(d_fbox, n, G, B.cursor, B.items, B.big, B.nbig_dev,
                                                                     skip_nan, d_rface, d_ent_src, B.want_ent ? B.ent : nullptr);
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
    if (int rc = h->ws.sorter.sort(h->d_codes, h->d_sorted_face, n, 32, 1, s, nullptr, nullptr)) return rc;
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
    h->grid.want_ent = true;
    h->xgrid_built = false;
    if (int rc = grid_build(h->grid, h->d_fbox, h->nf, lo, hi, 0, h->stream, h->opt_grid_tri ? h->d_rface : nullptr,
                            h->opt_grid_per_face, h->d_rface))
        return rc;
    if (!h->opt_x_grid) return 0;
    // the x+ tie-break ray's own grid over (y', z')
    float lo2[2] = {FLT_MAX, FLT_MAX}, hi2[2] = {-FLT_MAX, -FLT_MAX};
    for (int c = 0; c < 8; ++c) {
        const float x = h->aabb[(c & 1) ? 3 : 0], y = h->aabb[(c & 2) ? 4 : 1], z = h->aabb[(c & 4) ? 5 : 2];
        for (int d = 0; d < 2; ++d) {
            const float r = x * h->R[d + 1] + y * h->R[3 + d + 1] + z * h->R[6 + d + 1];
            lo2[d] = r < lo2[d] ? r : lo2[d];
            hi2[d] = r > hi2[d] ? r : hi2[d];
        }
    }
    h->grid_x.want_ent = true;
    if (int rc = grid_build(h->grid_x, h->d_fbox, h->nf, lo2, hi2, 0, h->stream, h->opt_grid_tri ? h->d_rface : nullptr,
                            h->opt_grid_per_face, h->d_rface, true))
        return rc;
    h->xgrid_built = true;
    return 0;
}

// Does anything walk the LBVH?  The z rays in ray mode 2 (or when the automatic rule picks it, see mesh_view),
// the x+ ray without its grid.  Otherwise the LBVH is neither ordered nor refit by update() (0.3 ms of 2.6).
bool lbvh_rule(const sdfhip_ctx* h) {
    return h->opt_ray_mode == 2 ||
           (h->opt_ray_mode == 0 && (h->grid.nbig > 64u || (h->grid.coarsened && !h->opt_grid_tri)));
}
bool wants_lbvh(const sdfhip_ctx* h) { return lbvh_rule(h) || !h->xgrid_built; }
int ensure_lbvh(sdfhip_ctx* h) {
    if (h->lbvh_valid) return 0;
    // the face order follows the ray-cast frame: rebuilt when R changed, refit otherwise
    if (!h->topo_built || std::memcmp(h->topo_R, h->R, sizeof(h->topo_R)) != 0)
        if (int rc = build_topology(h)) return rc;
    if (int rc = refit(h)) return rc;
    h->lbvh_valid = true;
    return 0;
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
    if (h->nf)
        k_geo_adj<<<(uint32_t)((12 * (size_t)h->nf + 255) / 256), 256, 0, s>>>(h->d_geo, h->d_adj_faces, 3u * h->nf, h->d_geo_adj);
    HIPC(hipGetLastError());
    h->h_face_valid = false;
    h->cum_valid = false;
    if (h->robust) {
        h->lbvh_valid = false;
        if (int rc = build_grid(h)) return rc;
        if (wants_lbvh(h))
            if (int rc = ensure_lbvh(h)) return rc;
    }
    return 0;
}

// (re)allocates the group arrays and builds the oriented boxes of the wide packet walk from the group
// refs in d_kd_gref (kd_pts4 / kd_vind must be current on the stream)
int reserve_groups(sdfhip_ctx* h, uint32_t ngroups) {
    if ((size_t)ngroups > h->kd_groups_cap) {
        const size_t cap = (size_t)ngroups + ngroups / 8 + 64;
        const size_t nl1 = (cap + 63) / 64, rows = (nl1 + 63) / 64;
        if (int rc = dalloc(&h->d_kd_gref, cap)) return rc;
        if (int rc = dalloc(&h->d_kd_gobb, 256 * nl1)) return rc;
        if (int rc = dalloc(&h->d_kd_gobb_aos, 256 * nl1)) return rc;
        if (int rc = dalloc(&h->d_kd_l1obb, 256 * rows)) return rc;
        // lanes past the last child read (and ignore) what lies there: keep it defined
        HIPC(hipMemsetAsync(h->d_kd_gobb, 0, sizeof(float4) * 256 * nl1, h->stream));
        HIPC(hipMemsetAsync(h->d_kd_l1obb, 0, sizeof(float4) * 256 * rows, h->stream));
        h->kd_groups_cap = cap;
    }
    return 0;
}
int build_obbs(sdfhip_ctx* h, uint32_t ngroups) {
    hipStream_t s = h->stream;
    h->n_groups = ngroups;
    h->n_l1 = (ngroups + 63u) / 64u;
    if (ngroups == 0) return 0;
    k_group_obb<<<(ngroups + 63) / 64, 64, 0, s>>>(h->d_kd_gref, ngroups, h->d_kd_pts4, h->d_kd_vind, h->d_kd_gobb,
                                                   h->d_kd_gobb_aos);
    k_l1_obb<<<h->n_l1, 64, 0, s>>>(h->d_kd_gref, ngroups, h->d_kd_pts4, h->d_kd_vind, h->d_kd_l1obb);
    HIPC(hipGetLastError());
    return 0;
}

// fast_nn: vertex tree of the packet pass built on the device (d_verts and aabb are current)
int build_fast_tree(sdfhip_ctx* h) {
    hipStream_t s = h->stream;
    const uint32_t nv = h->nv;
    const uint32_t ngroups = (nv + FAST_GROUP - 1) / FAST_GROUP;
    const size_t nalloc = (size_t)FAST_GROUP * ngroups + 16;
    if (int rc = reserve_groups(h, ngroups)) return rc;
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
        if (int rc = h->ws.sorter.sort(h->d_vkeys, h->d_vidx, nv, 30, 1, s, nullptr, &sorted_idx)) return rc;
    }
    k_fast_positions<<<(uint32_t)((nalloc + 255) / 256), 256, 0, s>>>(h->d_verts, sorted_idx, nv,
                                                                      (uint32_t)nalloc, h->d_kd_pts4,
                                                                      h->d_kd_vind);
    if (ngroups) k_fast_grefs<<<(ngroups + 255) / 256, 256, 0, s>>>(nv, ngroups, h->d_kd_gref);
    HIPC(hipGetLastError());
    return build_obbs(h, ngroups);
}

// The approximate first pass of the packet kernel orders the vertices of a lane by an estimate of the squared
// distance whose error E = u (16 d^2 + 24 R^2) (u = 2^-24, d the distance, R the radius of a packet group:
// kernels_packet.h, apx_err) is of the size of the distance, not of the scene.  Every group whose minimum lies
// within 2E of the best is re-evaluated exactly, a fourth one sends the query to the exact scan — results never
// depend on it, the time does.  Expected number of vertices in that window for a query at distance d from a
// surface of vertex density rho: lambda = rho * pi * 2E.  rho is taken from the surface of the AABB, d as a
// quarter of its diagonal (queries spread over the mesh's own box); both crude, the threshold is generous.
// E: the absolute window of the approximate pass, lambda: vertices per window
bool approx_window(const sdfhip_ctx* h, double& E, double& lambda) {
    double e[3];
    for (int d = 0; d < 3; ++d) e[d] = (double)h->aabb[d + 3] - (double)h->aabb[d];
    const double diag2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    E = 5.9604644775390625e-8 * 16.0 * diag2 / 16.0;
    const double area = 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    if (!(area > 0.0) || !std::isfinite(area) || !std::isfinite(diag2)) return false;
    lambda = (double)h->nv / area * 3.141592653589793 * 2.0 * E;
    return true;
}
constexpr double APPROX_LAMBDA_MAX = 0.5;
bool approx_pays(const sdfhip_ctx* h) {
    if (h->nv < 64) return false;
    double E, lambda;
    if (!approx_window(h, E, lambda)) return false;
    return lambda <= APPROX_LAMBDA_MAX;
}
// device arrays of the strict tree: `nnodes` internal nodes, `nalloc` padded positions.  d_kd_pts4 / d_kd_vind are
// shared with the GPU-built tree of fast_nn (build_fast_tree grows them under kd_pos_cap); the approximate pass's
// copy and the leaf boxes exist in strict mode only and keep capacities of their own — a fast_nn update may have
// grown kd_pos_cap past what they were allocated for.
int reserve_strict_tree(sdfhip_ctx* h, size_t nnodes, size_t nalloc) {
    if (nnodes > h->kd_nodes_cap) {
        if (int rc = dalloc(&h->d_kd_pn, 4 * nnodes)) return rc;
        if (int rc = dalloc(&h->d_kd_split, nnodes)) return rc;
        h->kd_nodes_cap = nnodes;
    }
    if (nalloc > h->kd_pos_cap) {
        if (int rc = dalloc(&h->d_kd_pts4, nalloc)) return rc;
        if (int rc = dalloc(&h->d_kd_vind, nalloc)) return rc;
        h->kd_pos_cap = nalloc;
    }
    if (nalloc > h->kd_apx_cap) {
        if (int rc = dalloc(&h->d_kd_apx4, nalloc)) return rc;
        if (int rc = dalloc(&h->d_kd_actr, nalloc / 4 + 4)) return rc;
        if (!h->d_apx_r2)
            if (int rc = dalloc(&h->d_apx_r2, 4)) return rc;
        h->kd_apx_cap = nalloc;
    }
    if (nalloc > h->kd_lbox_cap) {
        if (int rc = dalloc(&h->d_kd_lbox, 2 * nalloc)) return rc;
        h->kd_lbox_cap = nalloc;
    }
    return 0;
}

// The approximate pass's copy of the tree positions (k_apx_fill; needs the groups' boxes: after build_obbs) and the
// error model that goes with it, E(est) = apx_k1 est + apx_e0 = u (16 est + 24 R^2), rounded up; R^2 = the largest
// |fl(v - cg)|^2 the kernel met.  Synchronizes the stream.
int fill_apx(sdfhip_ctx* h, uint32_t ngroups, size_t nalloc) {
    hipStream_t s = h->stream;
    // The window of the pass held against the density of each packet group (its vertices over the disc of its box's half
    // diagonal): a mesh whose vertices crowd in one place (a fine part in a coarse scene) passes the average test
    // of approx_pays() while every query that has the fine part nearest meets four candidate groups and goes to the
    // exact scan — a pass over all vertices each (424^2 torus + a copy shrunk to 1 % beside it, 32M random points: nn
    // 195 ms with the pass, 39 ms without).  More than 2 % of the vertices in groups denser than four times the limit
    // on the average turn the pass off (a torus is twice as dense inside as outside, and the limit on the average
    // was tuned on tori).  Counted by k_apx_fill, one wave per group.
    double E = 0.0, lambda = 0.0;
    const bool have_window = approx_window(h, E, lambda);
    const float dens_k = have_window ? (float)(2.0 * E / (4.0 * APPROX_LAMBDA_MAX)) : 0.f;
    HIPC(hipMemsetAsync(h->d_apx_r2, 0, 4 * sizeof(uint32_t), s));
    k_apx_fill<<<ngroups + 1u, 64, 0, s>>>(h->d_kd_gref, ngroups, h->d_kd_gobb_aos, h->d_kd_pts4, h->d_kd_vind,
                                          h->kd_npos, (uint32_t)nalloc, h->d_kd_apx4, h->d_kd_actr, h->d_apx_r2, dens_k);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(h->apx_r2_bits, h->d_apx_r2, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    float r2;
    std::memcpy(&r2, &h->apx_r2_bits[0], sizeof r2);
    const double u = 5.9604644775390625e-8;
    auto up = [](double x) {
        float f = (float)x;
        return (double)f < x ? std::nextafter(f, FLT_MAX) : f;
    };
    h->apx_k1 = up(16.0 * u * 1.001);
    h->apx_e0 = up(24.0 * u * (double)r2 * 1.001);
    h->apx_usable = h->apx_r2_bits[1] == 0u && std::isfinite(h->apx_e0);
    // The absolute term alone as a window (approx_window): one stray group as wide as the scene makes every query
    // ambiguous.  (Followers of a group take the same decision from the same data.)
    if (h->approx_ok && have_window) {
        lambda *= (double)h->apx_e0 / E;
        if (!(lambda <= APPROX_LAMBDA_MAX)) h->approx_ok = false;
    }
    if (!have_window || (uint64_t)h->apx_r2_bits[3] * 50u > (uint64_t)h->apx_r2_bits[2]) h->approx_ok = false;  // (locally dense)
    if (!h->apx_usable) h->approx_ok = false;
    if (std::getenv("SDFHIP_DEBUG"))
        std::fprintf(stderr, "[sdfhip] approximate pass: largest group radius^2 %.4g, e0 %.4g, usable %d, on %d\n", (double)r2,
                     (double)h->apx_e0, (int)h->apx_usable, (int)h->approx_ok);
    return 0;
}

// A group update (group.h): the leader builds the tree once and tells the other members how far it is.
//   stage 1: the new vertices are enqueued on the leader's stream (ev_verts recorded behind them); the mesh
//            AABB, the rotation and the centre of the approximate pass are final
//   stage 2: every upload of the strict tree is enqueued (ev_tree recorded behind them); the scalars that
//            describe the tree (kd_depth .. kd_nnodes, approx_ok) are final
// fail(): the leader's update ended with an error; waiting members give up with it.
struct UpdateSync {
    std::mutex mu;
    std::condition_variable cv;
    int stage = 0, rc = 0;
    std::string err;
    hipEvent_t ev_verts = nullptr, ev_tree = nullptr;
    void post(int st) {
        std::lock_guard<std::mutex> l(mu);
        if (st > stage) stage = st;
        cv.notify_all();
    }
    void fail_with(int code, const std::string& msg) {
        std::lock_guard<std::mutex> l(mu);
        if (!rc) rc = code ? code : SDFHIP_E_STATE, err = msg;
        cv.notify_all();
    }
    int wait(int st) {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [&] { return stage >= st || rc != 0; });
        return rc;
    }
};

// order `s` after the last asynchronous query of this handle (no-op on the same stream)
int wait_for_queries(sdfhip_ctx* h, hipStream_t s) {
    if (h->query_pending && h->last_query_stream != s) HIPC(hipStreamWaitEvent(s, h->ev_query, 0));
    return 0;
}
int mark_query(sdfhip_ctx* h, hipStream_t s) {
    if (!h->ev_query) HIPC(hipEventCreateWithFlags(&h->ev_query, hipEventDisableTiming));
    HIPC(hipEventRecord(h->ev_query, s));
    h->last_query_stream = s;
    h->query_pending = true;
    return 0;
}

int do_update(sdfhip_ctx* h, bool draw_R, UpdateSync* sync = nullptr) {
    auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = h->stream;
    if (int rc = wait_for_queries(h, s)) return rc;  // queries still in flight read what update rewrites
    // mesh AABB (sdf.cpp:36-37; == the kd root box): needed by the device side before the tree
    h->kd.bounds(h->h_verts, h->nv, (int)h->opt_build_threads, h->aabb);
    h->approx_ok = approx_pays(h);
    // Packet groups: a slot of the approximate pass costs 3.5 VALU against 10 of the exact walk, so
    // larger groups (fewer box tests and sorted inserts per wave) pay there.  Measured on
    // the 424^2 torus, groups of 24 / 40 / 60: 64M random points sdf 16.8 / 16.3 / 16.9 ms, 256^3 grid
    // nn 5.1 / 4.6 / 4.8 ms, 4M points nn 2.8 / 2.6 / 2.6 ms.  The exact walk was tuned to 24 on the 4-wide tree;
    // on the wide walk (costs per group, not per vertex) 40 is better there too from ~50k vertices on
    // (24 / 40: 424^2 torus nn 64M 13.8 / 13.1 ms, icosphere-7 32M 9.3 / 8.5, 1M-vertex torus 32M 39.1 / 31.5;
    // 6.9k-vertex torus, 256^3 grid 1.85 / 2.03).
    h->kd.group_max = h->opt_packet_group > 0
                          ? (uint32_t)h->opt_packet_group
                          : (h->nv >= 600000u ? 60u  // (1M-vertex torus, 32M points, groups of 32 / 40 / 60: nn 33.8 / 30.7 / 28.9 ms)
                             : (((h->opt_approx < 0 ? h->approx_ok : h->opt_approx != 0) || h->nv >= 50000u) ? 40u : 24u));
    if (h->robust && draw_R && !h->R_pinned) draw_rotation(h);
    // Device side (vertex upload, per-face precompute, LBVH order / refit) does not depend on
    // the kd-tree: it runs from a helper thread while this thread builds the tree.
    if (h->nv)
        HIPC(hipMemcpyAsync(h->d_verts, h->h_verts, sizeof(float) * 3 * (size_t)h->nv,
                            hipMemcpyHostToDevice, s));
    if (sync) {
        HIPC(hipEventRecord(sync->ev_verts, s));
        sync->post(1);
    }
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
    h->kd.build(h->h_verts, h->nv, (int)h->opt_build_threads, h->aabb);
    h->kd_builds++;
    h->last_kd_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tk).count();
    // vertices in (padded) tree order; sentinel slots are far away: their distance is +inf
    const size_t npos = h->kd.pos_vert.size();
    const size_t nalloc = npos + 16;
    auto tq0 = std::chrono::steady_clock::now();
    h->kd.build_groups();
    auto tq1 = std::chrono::steady_clock::now();
    const int rc_dev = dev_side.get();
    auto tq2 = std::chrono::steady_clock::now();
    if (rc_dev) return fail(rc_dev, "%s", dev_err.empty() ? "device-side update failed" : dev_err.c_str());
    if (h->kd.nodes.size() >= (size_t)KD_MARK) return fail(SDFHIP_E_INVALID, "kd-tree too large");
    if (int rc = reserve_groups(h, (uint32_t)h->kd.groups.size())) return rc;
    if (int rc = reserve_strict_tree(h, h->kd.nodes.size(), nalloc)) return rc;
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
    k_tree_positions<<<(uint32_t)((nalloc + 255) / 256), 256, 0, s>>>(h->d_verts, h->d_kd_vind, (uint32_t)npos, (uint32_t)nalloc,
                                                                     h->d_kd_pts4);
    if (!h->kd.nodes.empty()) {
        HIPC(hipMemcpyAsync(h->d_kd_pn, h->kd.nodes.data(), sizeof(sdfk::KdNodeH) * h->kd.nodes.size(),
                            hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(h->d_kd_split, h->kd.split.data(), sizeof(uint32_t) * h->kd.split.size(),
                            hipMemcpyHostToDevice, s));
    }
    {
        const uint32_t nn = (uint32_t)h->kd.nodes.size();
        k_leaf_boxes<<<(nn + 255) / 256 + (nn == 0), 256, 0, s>>>(h->d_kd_pn, nn, h->kd.root_ref, h->aabb[0], h->aabb[1], h->aabb[2],
                                                               h->aabb[3], h->aabb[4], h->aabb[5], h->d_kd_lbox);
    }
    if (!h->kd.groups.empty())
        HIPC(hipMemcpyAsync(h->d_kd_gref, h->kd.groups.data(), sizeof(uint32_t) * h->kd.groups.size(),
                            hipMemcpyHostToDevice, s));
    h->kd_depth = h->kd.depth, h->kd_root = h->kd.root_ref;
    h->kd_npos = (uint32_t)npos, h->kd_nnodes = (uint32_t)h->kd.nodes.size();
    h->approx_pre = h->approx_ok;
    if (sync) {  // (before the boxes: the other members build their own from the copied tree)
        HIPC(hipEventRecord(sync->ev_tree, s));
        sync->post(2);
    }
    if (int rc = build_obbs(h, (uint32_t)h->kd.groups.size())) return rc;
    if (int rc = fill_apx(h, (uint32_t)h->kd.groups.size(), nalloc)) return rc;  // (ends with a stream synchronize:
                                                                                 // host arrays are pageable and rebuilt by the next update)
    h->last_update_ms =
        std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("SDFHIP_DEBUG")) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[sdfhip] update %.2f ms: pre %.2f kd %.2f groups+fill %.2f wait-device %.2f upload %.2f\n",
                     h->last_update_ms, ms(t0, tk), h->last_kd_ms, ms(tq0, tq1), ms(tq1, tq2),
                     ms(tq2, std::chrono::steady_clock::now()));
    }
    return 0;
}

// far packets (kernels_packet.h): squared radius of the near region of a DENSE batch — twelve half diagonals of the mesh
// AABB around its centre; the key kernel shortens it to two for a sparse batch (k_query_morton).  8M points in a box
// +-4 / +-8 / +-30 around the unit torus with one radius for all batches, nn: 2 half diagonals 14.2 / 10.6 / 6.3 ms,
// 4.5: 4.4 / 7.3 / 6.7, 6: 4.25 / 6.9 / 6.9, 9: 4.2 / 6.2 / 8.8 — closer in, the cap facing a query is too large for the
// far walk while a dense batch's spatial packets still do well; with the sparse radius apart, 6 / 12 for dense
// batches: +-8 8M points 7.3 / 6.1 ms, +-16 64M points 35.1 / 29.4.  0: no far packets
float far_radius2(const sdfhip_ctx* h) {
    double r2 = 0.0;
    for (int d = 0; d < 3; ++d) {
        const double e = 0.5 * ((double)h->aabb[d + 3] - (double)h->aabb[d]);
        r2 += e * e;
        if (!std::isfinite(0.5 * ((double)h->aabb[d + 3] + (double)h->aabb[d]))) return 0.f;
    }
    const double k = 0.1 * (double)h->opt_far_x10;
    const float f = (float)(k * k * r2);
    return h->opt_far && f > 0.f && f < 1e30f ? f : 0.f;
}

MeshDev mesh_view(const sdfhip_ctx* h) {
    MeshDev M;
    M.kd_pn = h->d_kd_pn;
    M.kd_split = h->d_kd_split;
    M.kd_gobb = h->d_kd_gobb;
    M.kd_gobb_aos = h->d_kd_gobb_aos;
    M.cube_max = 0u, M.cube_par = nullptr, M.cube_hdr = nullptr, M.cube_list = nullptr;  // (set per launch, run_query_t)
    M.kd_l1obb = h->d_kd_l1obb;
    M.kd_ngroups = h->n_groups;
    M.kd_nl1 = h->n_l1;
    M.obb_rm = 0.f;
    for (int d = 0; d < 3; ++d) {
        M.obb_cm[d] = 0.5f * (h->aabb[d] + h->aabb[d + 3]);
        M.obb_rm += h->aabb[d + 3] - h->aabb[d];
    }
    M.fast = h->fast_active ? 1 : 0;
    M.kd_vind = h->d_kd_vind;
    M.kd_root = h->fast_active ? KD_LEAF : h->kd_root;
    for (int d = 0; d < 6; ++d) M.bb[d] = h->aabb[d];
    M.kd_pts4 = h->d_kd_pts4;
    M.kd_apx4 = h->d_kd_apx4;
    M.kd_lbox = h->d_kd_lbox;
    M.kd_actr = h->d_kd_actr;
    M.apx_k1 = h->apx_k1, M.apx_e0 = h->apx_e0;
    for (int d = 0; d < 3; ++d) M.far_c[d] = M.obb_cm[d];
    M.far_r2 = 0.f, M.far_lo = 0u, M.far_cnt = nullptr;  // (far packets: set per launch, run_query_t)
    M.kd_npos = h->fast_active ? (uint32_t)(FAST_GROUP * ((h->nv + FAST_GROUP - 1) / FAST_GROUP))
                               : h->kd_npos;
    // a wave may walk 1/8 of the tree (4 positions per chunk) before it hands its queries over
    M.heavy_budget = h->opt_heavy_budget > 0 ? (uint32_t)h->opt_heavy_budget
                                             : (M.kd_npos / 32u > 4096u ? M.kd_npos / 32u : 4096u);
    // hand-over budget of the cooperative continuation: 1500 work units on the bench mesh (230k positions); a wave's
    // ordinary walk grows with the tree (groups of 60 from 600k vertices on), and every packet beyond COOP_CAP goes
    // to the exact scan: 1M-vertex torus, 32M points, budget 1500 / 3000 / 4500 / 6000 / 10000 / 20000:
    // nn 63.2 (164k queries in the scan) / 22.0 / 21.5 / 21.8 / 22.1 / 25.3 ms
    const int64_t coop_auto = std::min<int64_t>(std::max<int64_t>(1500, (int64_t)M.kd_npos / 280), 6000);
    M.coop_budget = (uint32_t)std::min<int64_t>(h->opt_coop_budget < 0 ? coop_auto : h->opt_coop_budget, M.heavy_budget);
    M.amb_c = h->amb_c;
    M.amb_c2 = h->amb_c2;
    M.adj_off = h->d_adj_off;
    M.adj_faces = h->d_adj_faces;
    M.geo = h->d_geo;
    M.geo_adj = h->d_geo_adj;
    M.rnodes = h->d_rnodes;
    M.n_rnodes = h->robust && h->lbvh_valid ? h->n_rnodes : 0;
    M.rface = h->d_rface;
    M.fbox = h->d_fbox;
    M.gr_start = h->grid.start;
    M.gr_items = h->grid.items;
    M.gr_ent = h->grid.ent;
    M.gr_big = h->grid.big;
    M.gr_nbig = h->robust ? h->grid.nbig : 0;
    M.gr_g = h->grid.gr.g;
    // automatic: the wave-cooperative LBVH walk shares long candidate lists among 64 lanes.  It pays when
    // many faces span too many cells to be listed (every query tests the "big" list), or — without the
    // triangle filter of the cells — when faces touch > 10 cells each at the natural resolution.  With the
    // filter the grid wins also on rough surfaces (noisy icosphere-7, 64M points: 28 against 55 ms).
    M.ray_lbvh = lbvh_rule(h) && h->lbvh_valid ? 1 : 0;  // (an option changed since the update: set_option builds it)
    M.gr_o0 = h->grid.gr.o0, M.gr_o1 = h->grid.gr.o1, M.gr_inv0 = h->grid.gr.inv0, M.gr_inv1 = h->grid.gr.inv1;
    const bool xg = h->robust && h->xgrid_built && h->grid_x.start;
    M.gx_start = xg ? h->grid_x.start : nullptr;
    M.gx_ent = h->grid_x.ent;
    M.gx_big = h->grid_x.big;
    M.gx_nbig = xg ? h->grid_x.nbig : 0;
    M.gx_g = h->grid_x.gr.g;
    M.gx_o0 = h->grid_x.gr.o0, M.gx_o1 = h->grid_x.gr.o1, M.gx_inv0 = h->grid_x.gr.inv0, M.gx_inv1 = h->grid_x.gr.inv1;
    for (int d = 0; d < 9; ++d) M.R[d] = h->R[d];
    return M;
}

// strict per-lane kernel (exact nanoflann emulation).  n_dev != nullptr: the point count is read
// from device memory (fallback list of the packet kernel) and `n` only sizes the grid.
template <int KIND, bool ROBUST>
int launch_strict(sdfhip_ctx* h, const MeshDev& M, const float* d_pts, const uint32_t* perm, int64_t n,
                  int trunc, void* d_out, const uint32_t* n_dev, hipStream_t s, uint32_t n_min = 0u) {
    const bool needs_kd = !(KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST));
    const uint32_t depth = (uint32_t)(h->kd_depth > 0 ? h->kd_depth : 1);
    const int64_t blocks_full = (n + QBLOCK - 1) / QBLOCK;
    if (needs_kd && depth > (uint32_t)LDS_DEPTH_MAX) {
        const int64_t blocks = blocks_full < 4096 ? blocks_full : 4096;
        const size_t words = (size_t)blocks * QBLOCK * depth * 3;
        if (words > h->gstack_words) {
            if (h->W->fixed) return fail(SDFHIP_E_STATE, "kd-tree deeper than %d: the no-alloc entry cannot serve this mesh", LDS_DEPTH_MAX);
            if (int rc = dalloc(&h->d_gstack, words)) return rc;
            h->gstack_words = words;
        }
        k_query<KIND, ROBUST, true><<<(uint32_t)blocks, QBLOCK, 0, s>>>(M, d_pts, perm, n, trunc, d_out,
                                                                        h->d_gstack, depth, n_dev, n_min);
    } else {
        const size_t lds = needs_kd ? (size_t)(QBLOCK / 64) * depth * 192 * sizeof(uint32_t) : 0;
        const int64_t blocks = blocks_full < 0x7FFFFFFF ? blocks_full : 0x7FFFFFFF;
        k_query<KIND, ROBUST, false><<<(uint32_t)blocks, QBLOCK, lds, s>>>(M, d_pts, perm, n, trunc,
                                                                           d_out, nullptr, depth, n_dev, n_min);
    }
    HIPC(hipGetLastError());
    return 0;
}

struct ProfScope {  // optional hipEvent pair around everything one query call launches
    sdfhip_ctx* h;
    hipStream_t s;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    int begin() {
        if (h->opt_profile && !h->W->fixed && h->prof_pending.size() < 8192) {
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
    QueryWs& W = *h->W;
    if (W.fixed) {  // the static scratch of the no-alloc entry: sized once, for SDFHIP_NOALLOC_MAX_POINTS
        if (n > W.q_cap || fb_cap > W.fb_cap)
            return fail(SDFHIP_E_INVALID, "batch exceeds the static scratch workspace (%lld points)", (long long)W.q_cap);
        return 0;
    }
    if (n > W.q_cap) {
        if (int rc = dalloc(&W.qkeys, (size_t)n)) return rc;
        if (int rc = dalloc(&W.qperm, (size_t)n)) return rc;
        const size_t nw = ((size_t)n + 63) / 64 + 4;
        if (int rc = dalloc(&W.amb_mask, nw)) return rc;
        if (int rc = dalloc(&W.amb_cnt, nw)) return rc;
        if (int rc = dalloc(&W.amb_s1, nw / 2048 + 4)) return rc;
        if (int rc = dalloc(&W.amb_s2, nw / 2048 / 2048 + 4)) return rc;
        if (!W.qbounds)
            if (int rc = dalloc(&W.qbounds, 8)) return rc;
        if (!W.coop_list) {
            if (int rc = dalloc(&W.coop_list, (size_t)COOP_CAP)) return rc;
            if (int rc = dalloc(&W.coop_state, (size_t)COOP_CAP * 448)) return rc;
            const CoopDesc D = {W.coop_list, W.coop_state};
            HIPC(hipMemcpy(W.fb_count + 4, &D, sizeof(D), hipMemcpyHostToDevice));
        }
        W.q_cap = n;
    }
    if (fb_cap > W.fb_cap) {
        if (int rc = dalloc(&W.fb_list, 2 * (size_t)fb_cap)) return rc;
        W.fb_cap = fb_cap;
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
    QueryWs& W = *h->W;
    const size_t osz = KIND == Q_CONTAINS ? 1 : (KIND == Q_DETAIL ? 12 : 4);
    const int64_t chunk = h->opt_chunk > 0 ? h->opt_chunk : (1 << 24);
    // Robust contains() is the ray walk only (per-lane O(1) grid lookups): sorting (by column, see ColumnKey)
    // buys it shared cell lists, nothing else, and costs ~0.025 ms per million points.  It pays when the lists
    // are long (noisy icosphere-7, 43 entries per cell: 8M points 1.66 against 4.92 ms unsorted, 32M 5.1
    // against 19.1) and never when they are short (424^2 torus, 9 per cell: 32M points 2.30 against 1.63 ms
    // unsorted) — automatic: from 1M points on if the cells hold more than 20 entries on average.
    // (Not when the cooperative LBVH walk serves the rays: that one lives on coherent waves.)
    const int64_t cells = (int64_t)h->grid.gr.g * h->grid.gr.g;
    const int64_t contains_min = h->opt_sort_min_contains >= 0 ? h->opt_sort_min_contains
                                 : ((int64_t)h->grid.entries > 20 * cells ? (1ll << 20) : (1ll << 62));
    const int64_t sort_min = (KIND == Q_CONTAINS && ROBUST && !M.ray_lbvh && contains_min > h->opt_sort_min)
                                 ? contains_min
                                 : h->opt_sort_min;
    const bool sort_any = h->opt_sort && n >= sort_min;
    const bool fast = h->fast_active;  // GPU-built tree: the packet kernel is the only nn search
    // Small batches.  A tree walk is a chain of dependent loads: ~2 ms for ONE far-field query on a
    // 180k-vertex tree, whether a lane walks alone (strict kernel) or a wave walks for 64 sparse
    // queries (their union is most of the tree).  While n x positions is small the exact scan of
    // k_query_heavy (one wave per four queries, lanes over all vertices, no dependent loads) is
    // both the shortest and the cheapest: 1000 queries on 180k vertices in ~0.1 ms instead of 2.3.
    constexpr bool needs_nn = !(KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST));
    // Threshold on n x positions: the scan costs ~n x positions / 2.7e12 s, the tree-walking path has
    // a floor that grows with the tree (0.25 ms at 4k positions, 2.2 ms at 190k: ~positions^0.57).
    const double scan_max = h->opt_scan_max >= 0
                                ? (double)h->opt_scan_max
                                : 5.0e8 * std::pow((double)(M.kd_npos > 4000u ? M.kd_npos : 4000u) / 4000.0, 0.57);
    if (needs_nn && M.kd_npos > 0 && n <= 0x7FFFFFFF && (double)n * (double)M.kd_npos <= scan_max) {
        if (int rc = reserve_query_ws(h, 0, n)) return rc;
        constexpr int HK = needs_nn ? KIND : Q_NN;
        HIPC(hipMemsetAsync(W.fb_count, 0, 4 * sizeof(uint32_t), s));
        const int64_t packs = (n + HEAVY_Q - 1) / HEAVY_Q;
        // kinds with a tail: positions first, then one lane per query for ray parity + face distances
        int32_t* pos = KIND == Q_NN ? nullptr : (int32_t*)(W.fb_list + W.fb_cap);
        h->last_kernel = SDFHIP_KERNEL_SCAN;
        ProfScope prof{h, s};
        if (int rc = prof.begin()) return rc;
        if (packs <= 1024)  // few queries: 16 waves share each scan (the chain of one wave is ~0.4 ms at 190k)
            k_query_heavy<HK, ROBUST, 16><<<(uint32_t)packs, 1024, 0, s>>>(
                M, d_pts, nullptr, nullptr, trunc, d_out, W.fb_list, W.fb_count, (uint32_t)n, pos, 0u, 0xFFFFFFFFu);
        else
            k_query_heavy<HK, ROBUST, 4><<<(uint32_t)(packs < 16384 ? packs : 16384), 256, 0, s>>>(
                M, d_pts, nullptr, nullptr, trunc, d_out, W.fb_list, W.fb_count, (uint32_t)n, pos, 0u, 0xFFFFFFFFu);
        if (pos)
            k_scan_finish<HK, ROBUST><<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(M, d_pts, pos, (uint32_t)n, trunc,
                                                                                 d_out);
        if (int rc = prof.end(n)) return rc;
        HIPC(hipGetLastError());
        if (!fast) {  // what the scan could not decide (near ties failing the path check): exact emulation
            const int64_t grid_pts = n < (int64_t)QBLOCK * 4096 ? n : (int64_t)QBLOCK * 4096;
            if (int rc = launch_strict<KIND, ROBUST>(h, M, d_pts, W.fb_list, grid_pts, trunc, d_out,
                                                     W.fb_count, s))
                return rc;
        }
        if (h->opt_profile && !W.fixed)
            HIPC(hipMemcpyAsync(&h->h_counts[0], W.fb_count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        h->h_counts[1] = (uint32_t)n;
        return 0;
    }
    const bool packet_any =
        KIND != Q_DETAIL && (fast || (sort_any && h->opt_packet));
    const int64_t fb_cap = packet_any ? (n < (1ll << 27) ? n : (1ll << 27)) : 0;
    if (sort_any || fast)
        if (int rc = reserve_query_ws(h, n < chunk ? n : chunk, fb_cap)) return rc;
    int64_t deferred_pts = 0;  // points whose near ties may sit in the deferred list
    int flip = 0;              // d_fb_count[flip] holds the current list length
    // d_fb_count: [0], [1] length of the deferred list (ping-pong), [2] length of the heavy list
    if (packet_any) HIPC(hipMemsetAsync(W.fb_count, 0, 4 * sizeof(uint32_t), s));
    auto flush = [&]() -> int {
        if (deferred_pts == 0) return 0;
        constexpr int HK = (KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST)) ? Q_NN : KIND;
        if (!(KIND == Q_DETAIL || (KIND == Q_CONTAINS && ROBUST))) {
            // queries of the waves that gave up (tail limiter): one wave per query; what is
            // still ambiguous afterwards joins the deferred list
            const int64_t packs = (deferred_pts + HEAVY_Q - 1) / HEAVY_Q;
            // How many queries the list holds is known on the device only.  A short list (2k of 125M on
            // the bench mesh) leaves the GPU empty with one 4-wave block per four queries, and one wave's
            // chain over 190k positions is ~0.4 ms: 8 waves share each scan then (16 waves, two blocks per CU:
            // 531 blocks for 2122 queries are two rounds of 512, the second nearly empty: nn 4M 2.05 -> 1.96 ms,
            // 125M sdf step 20.73 -> 20.5 ms with 8).  A long
            // one (41k of 64M on the smooth icosphere-7) is throughput bound and 4-wave blocks are faster
            // (sdf 22.3 against 24.5 ms).  Both variants are launched; each returns at once outside its range.
            // A very long list is not the near-symmetry kind: the batch is spread over a region much larger than
            // the mesh and whole packets, as wide as the mesh, were sent here unwalked (packet_walk).  The scan
            // costs a pass over every vertex per query (8M points in a box 300 x the mesh: 554 ms); the strict
            // kernel, every lane walking the tree for itself, answers such far queries in ~5 ns each (51 ms) but
            // has a floor of several ms.  Above heavy_strict queries the strict kernel takes the list (strict mode only).
            const double hs = 4.0e10 / (double)(M.kd_npos ? M.kd_npos : 1u);
            const uint32_t heavy_strict = fast ? 0xFFFFFFFFu : (uint32_t)(hs < 65536.0 ? 65536.0 : (hs > 4.0e9 ? 4.0e9 : hs));
            // short lists: SHORT_Q queries per block of SHORT_W waves, one round of blocks on the GPU (2.3k queries are
            // 289 blocks; with 4 queries per block of 8 waves they were 577: two rounds of 512).  Kinds with a tail
            // store the tree position per list slot and k_scan_finish does ray parity + face distances one lane per
            // query — inside the scan kernel four threads of a block did that, one dependent load after the other,
            // while the other 508 waited (sdf 125M: 0.42 ms against nn's 0.19).
#ifndef SDFK_SHORT_W
#define SDFK_SHORT_W 8
#endif
#ifndef SDFK_SHORT_Q
#define SDFK_SHORT_Q 4
#endif
            constexpr int SHORT_W = SDFK_SHORT_W, SHORT_Q = SDFK_SHORT_Q;
            const int64_t packs_s = (std::min<int64_t>(deferred_pts, HEAVY_SHORT) + SHORT_Q - 1) / SHORT_Q;
            int32_t* spos = (KIND == Q_NN) ? nullptr : W.heavy_pos;
            k_query_heavy<HK, ROBUST, SHORT_W, SHORT_Q><<<(uint32_t)packs_s, 64 * SHORT_W, 0, s>>>(
                M, d_pts, W.fb_list + W.fb_cap, W.fb_count + 2, trunc, d_out, W.fb_list,
                W.fb_count + flip, 0u, spos, 1u, HEAVY_SHORT);
            if (spos)
                k_scan_finish<HK, ROBUST><<<HEAVY_SHORT / 256, 256, 0, s>>>(M, d_pts, spos, HEAVY_SHORT, trunc, d_out,
                                                                            W.fb_list + W.fb_cap, W.fb_count + 2);
            if (deferred_pts > (int64_t)HEAVY_SHORT)
                k_query_heavy<HK, ROBUST, 4><<<(uint32_t)(packs < 8192 ? packs : 8192), 256, 0, s>>>(
                    M, d_pts, W.fb_list + W.fb_cap, W.fb_count + 2, trunc, d_out, W.fb_list,
                    W.fb_count + flip, 0u, nullptr, HEAVY_SHORT + 1u, heavy_strict);
            HIPC(hipGetLastError());
            if (!fast && deferred_pts > (int64_t)heavy_strict) {
                const int64_t gp = deferred_pts < (int64_t)QBLOCK * 65536 ? deferred_pts : (int64_t)QBLOCK * 65536;
                if (int rc = launch_strict<KIND, ROBUST>(h, M, d_pts, W.fb_list + W.fb_cap, gp, trunc, d_out, W.fb_count + 2,
                                                         s, heavy_strict))
                    return rc;
            }
        }
        if (!fast) {
            const int64_t grid_pts = deferred_pts < (int64_t)QBLOCK * 4096 ? deferred_pts : (int64_t)QBLOCK * 4096;
            if (int rc = launch_strict<KIND, ROBUST>(h, M, d_pts, W.fb_list, grid_pts, trunc, d_out,
                                                     W.fb_count + flip, s))
                return rc;
        }
        if (h->opt_profile && !W.fixed) {
            HIPC(hipMemcpyAsync(&h->h_counts[0], W.fb_count + flip, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPC(hipMemcpyAsync(&h->h_counts[1], W.fb_count + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        HIPC(hipMemsetAsync(W.fb_count, 0, 4 * sizeof(uint32_t), s));
        flip = 0;
        deferred_pts = 0;
        return 0;
    };
    // far packets: sorted launches of the packet pass (not the static scratch: its sizes are small and its callers near
    // the mesh)
    const float far_r2 = far_radius2(h);
    const bool far_ok = packet_any && !W.fixed && far_r2 > 0.f && !(KIND == Q_CONTAINS && ROBUST);
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = (n - off) < chunk ? (n - off) : chunk;
        const float* p = d_pts + 3 * off;
        void* o = (char*)d_out + osz * (size_t)off;
        const uint32_t* perm = nullptr;
        if (h->opt_sort && m >= sort_min) {
            // robust contains() is rays only: its queries are ordered by column of the ray-cast frame
            ColumnKey CK = {};
            if (KIND == Q_CONTAINS && ROBUST && h->opt_column_sort) {
                CK.on = 1;
                for (int d = 0; d < 3; ++d) CK.r0[d] = M.R[3 * d], CK.r1[d] = M.R[3 * d + 1];
                CK.o0 = M.gr_o0, CK.o1 = M.gr_o1;
                CK.s0 = M.gr_inv0 * (4096.f / (float)M.gr_g), CK.s1 = M.gr_inv1 * (4096.f / (float)M.gr_g);
            } else {
                k_bounds_init<<<1, 64, 0, s>>>(W.qbounds);
                // large batches: the bounds come from a strided sample (1/16 of the points, in 3 KB runs)
                const int sample = m >= (8 << 20) ? 16 : 1;
                const int64_t groups = m / 4 / sample;
                const int nb = (int)((groups + 255) / 256 < 1024 ? (groups + 255) / 256 + 1 : 1024);
                k_bounds<<<nb, 256, 0, s>>>(p, m, W.qbounds, sample);
            }
            if (int rc = W.sorter.reserve((uint32_t)m)) return rc;  // (the key kernel fills its first histogram)
            const uint32_t ntiles = (uint32_t)((m + RS_TILE - 1) / RS_TILE);
            const int key_bits = CK.on ? 24 : (int)h->opt_sort_bits;
            // far queries sort behind the near ones, by direction, when the packet pass will walk them as far packets
            FarKey FK = {{M.far_c[0], M.far_c[1], M.far_c[2]}, 0.f, 0.f, 0.f, 1.f};
            if (far_ok && !CK.on) {
                const double hd = std::sqrt((double)far_r2) / (0.1 * (double)h->opt_far_x10);  // half diagonal of the mesh AABB
                const double rs = 0.1 * (double)std::min(h->opt_far_sparse_x10, h->opt_far_x10) * hd;
                FK.r2 = far_r2, FK.r2_sparse = (float)(rs * rs);
                FK.rho_dense = (float)(0.10 * hd), FK.rho_sparse = (float)(0.16 * hd);
            }
            k_query_morton<<<ntiles, RS_THREADS, 0, s>>>(p, (uint32_t)m, W.qbounds, W.qkeys, W.sorter.hist, ntiles, CK, 30 - key_bits, FK);
            uint32_t* sorted_perm = nullptr;
            if (int rc = W.sorter.sort(W.qkeys, W.qperm, (uint32_t)m, key_bits, 1, s, nullptr, &sorted_perm, false, true))
                return rc;
            perm = sorted_perm;
        }
        if (packet_any && (perm || fast)) {
            if (deferred_pts + m > W.fb_cap)
                if (int rc = flush()) return rc;
            constexpr int PK = KIND == Q_DETAIL ? Q_NN : KIND;
            ProfScope prof{h, s};  // the dominant kernel, one event pair per launch
            if (int rc = prof.begin()) return rc;
            // strict mode: approximate first pass + exact second phase (kernels_packet.h); the GPU-built
            // tree of fast_nn has no strict kernel behind it and keeps the exact walk
            const bool approx = !fast && h->apx_usable && (h->opt_approx < 0 ? h->approx_ok : h->opt_approx != 0);
            h->last_kernel = approx ? SDFHIP_KERNEL_PACKET_APPROX : SDFHIP_KERNEL_PACKET_EXACT;
            // Long walks go on in the cooperative kernel (8 waves per packet); not for ray-only contains().
            // It is one more serial phase (~0.3 ms): it pays most while the launch is short enough for its few
            // long waves to be the tail (nn 4M 2.68 -> 2.09 ms), and since its chunk loads are scalar again
            // (no loop over packets) also on the longest launches, where fewer packets end in the exact scan
            // (125M sdf 22.22 -> 21.93 ms, icosphere-7 64M 14.0 -> 13.55 ms).
            const bool coop = approx && M.coop_budget != 0u && W.coop_list && !(ROBUST && KIND == Q_CONTAINS) &&
                              m <= h->opt_coop_max;
            // Hand-over budget.  The main launch cannot end before its longest wave does, and a wave's walk is a chain
            // of dependent loads (~0.35 us per work unit when the GPU is nearly empty): a small launch is as long as
            // its budget.  Automatic: 1500 units from 1M points on (tools/budget_sweep.sh), proportionally less below,
            // not under 300 (100k points: nn 1.16 -> 0.95 ms; the cooperative kernel takes the packets over earlier —
            // few of them in a small launch, far from its capacity COOP_CAP).
            uint32_t coop_budget = M.coop_budget;
            if (h->opt_coop_budget < 0 && m < (1ll << 20))
                coop_budget = std::min<uint32_t>(coop_budget, (uint32_t)std::max<int64_t>(300, 1500 * m >> 20));
            MeshDev Mm = M;  // the main kernel's view: its budget, and in the low bit who takes over beyond it
            Mm.heavy_budget = coop ? (coop_budget | 1u) : (M.heavy_budget & ~1u);
            const int64_t npk_all = (m + 63) / 64;
            const bool far_run = far_ok && perm != nullptr;
            if (far_run) Mm.far_r2 = far_r2, Mm.far_lo = (uint32_t)std::max<int64_t>(0, npk_all - (int64_t)FAR_CAP), Mm.far_cnt = W.qbounds + 6;
            // Cube lists (kernels_packet.h, k_cube_lists): the sorted launch's query domain is cut into nc^3 cubes of
            // ~4096 queries; one wave per cube lists the groups its queries can need and the packet waves test that
            // list instead of walking the boxes from the top.  It pays when a cube is small against the mesh (short
            // lists) and holds many waves: large launches.  Needs the launch's bounds (k_bounds), i.e. a sorted launch.
            const bool lists = perm && !W.fixed && M.kd_ngroups != 0u &&
                               (h->opt_cube_lists < 0 ? m >= h->opt_cube_min : h->opt_cube_lists != 0);
            if (lists) {
                // cube edge: ~1.3 x the size of a packet group, estimated from the surface of the mesh AABB (as
                // approx_window does); the device lays the grid out over the launch's bounds (k_cube_lists)
                double ex[3], area;
                for (int d = 0; d < 3; ++d) ex[d] = (double)h->aabb[d + 3] - (double)h->aabb[d];
                area = 2.0 * (ex[0] * ex[1] + ex[1] * ex[2] + ex[2] * ex[0]);
                double side = 1.3 * std::sqrt(area / (double)M.kd_ngroups);
                if (!(side > 0.0) || !std::isfinite(side)) side = 1.3 * std::fmax(ex[0], std::fmax(ex[1], ex[2])) / std::cbrt((double)M.kd_ngroups);
                if (!(side > 0.0) || !std::isfinite(side)) side = 1.0;
                const uint32_t level = (uint32_t)(h->opt_cube_level > 0 ? h->opt_cube_level : 0);
                // at most one cube per 256 queries (a cube must hold several waves to pay for its list)
                size_t ncubes = level ? (size_t)1 << (3 * level) : (size_t)std::min<int64_t>(std::max<int64_t>(m / 256, 64), 1 << 18);
                if (ncubes > W.cubes_cap) {
                    if (!W.cube_par)
                        if (int rc = dalloc(&W.cube_par, 16)) return rc;
                    if (int rc = dalloc(&W.cube_hdr, 3 * ncubes)) return rc;
                    if (int rc = dalloc(&W.cube_list, (size_t)CUBE_CAP * ncubes)) return rc;
                    W.cubes_cap = ncubes;
                }
                k_cube_lists<<<(uint32_t)ncubes, 64, 0, s>>>(M, W.qbounds, (float)side, level, (uint32_t)ncubes, W.cube_par,
                                                              W.cube_hdr, W.cube_list);
                Mm.cube_max = (uint32_t)ncubes, Mm.cube_par = W.cube_par, Mm.cube_hdr = W.cube_hdr, Mm.cube_list = W.cube_list;
            }
            if (approx)
                k_query_packet<PK, ROBUST, true><<<(uint32_t)((m + PBLOCK - 1) / PBLOCK), PBLOCK, 0, s>>>(
                    Mm, p, perm, m, trunc, o, W.amb_mask, W.amb_cnt, (uint32_t)off,
                    W.fb_list + W.fb_cap, W.fb_count + 2);
            else
                k_query_packet<PK, ROBUST, false><<<(uint32_t)((m + PBLOCK - 1) / PBLOCK), PBLOCK, 0, s>>>(
                    Mm, p, perm, m, trunc, o, W.amb_mask, W.amb_cnt, (uint32_t)off,
                    W.fb_list + W.fb_cap, W.fb_count + 2);
            if (int rc = prof.end(m)) return rc;  // (the dominant kernel alone: what rocprofv3 lists under its name)
            if (far_run) {  // the far packets the main kernel left alone (from the end of the sorted order)
                MeshDev Mf = Mm;
                Mf.heavy_budget = M.heavy_budget;
                const int64_t fpk = std::min<int64_t>(npk_all, (int64_t)FAR_CAP);
                k_query_packet_far<PK, ROBUST, false><<<(uint32_t)((fpk + FAR_BLOCK / 64 - 1) / (FAR_BLOCK / 64)), FAR_BLOCK, 0, s>>>(
                    Mf, p, perm, m, trunc, o, W.amb_mask, W.amb_cnt, (uint32_t)off, W.fb_list + W.fb_cap, W.fb_count + 2);
            }
            if (coop) {
                const int64_t npk = (m + 63) / 64;
                const CoopDesc CB = {W.coop_list, W.coop_state};
                k_query_packet_coop<PK, ROBUST><<<(uint32_t)(npk < (int64_t)COOP_CAP ? npk : (int64_t)COOP_CAP), 64 * COOP_WAVES, 0, s>>>(
                    M, p, perm, m, trunc, o, W.amb_mask, W.amb_cnt, (uint32_t)off,
                    W.fb_list + W.fb_cap, W.fb_count + 2, CB);
                HIPC(hipMemsetAsync(W.fb_count + 3, 0, sizeof(uint32_t), s));
            }
            if (fast) {  // no near-tie deferral in this mode, only the heavy list
                HIPC(hipGetLastError());
                deferred_pts += m;
                continue;
            }
            exclusive_scan_u32(W.amb_cnt, (uint32_t)((m + 63) / 64), W.amb_s1, W.amb_s2, s);
            k_fb_compact<<<(uint32_t)(((m + 63) / 64 + 255) / 256), 256, 0, s>>>(
                W.amb_mask, W.amb_cnt, perm, m, (uint32_t)off, W.fb_list, W.fb_count + flip,
                W.fb_count + (flip ^ 1));
            HIPC(hipGetLastError());
            flip ^= 1;
            deferred_pts += m;
        } else {
            h->last_kernel = SDFHIP_KERNEL_STRICT;
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
// for the GPU) overlap the kernels.  Chunks stay large (see the schedule in run_query_host): the
// packet pass is faster on dense Morton-sorted batches and every chunk pays the serial phases once.
// Pageable caller memory.  hipMemcpyAsync stages such copies itself, at 31 GB/s host-to-device and 18 GB/s
// device-to-host on the target box, blocking the caller; a plain memcpy into page-locked memory runs at
// 60+ GB/s on one thread and the DMA from there at 57 GB/s, so the two are pipelined by hand: pieces of
// BOUNCE_BYTES through two page-locked buffers per direction (copy piece i + 1 while piece i is on the bus).
constexpr size_t BOUNCE_BYTES = (size_t)8 << 20;
constexpr int BOUNCE_THREADS = 4;
bool is_pageable(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (an unregistered pointer is reported as an error by some versions)
        return true;
    }
    return a.type == hipMemoryTypeUnregistered;
}
int ensure_bounce(sdfhip_ctx* h) {
    for (int k = 0; k < 2; ++k) {
        if (!h->pin_up[k] && hipHostMalloc((void**)&h->pin_up[k], BOUNCE_BYTES, hipHostMallocDefault) != hipSuccess)
            return fail(SDFHIP_E_NOMEM, "hipHostMalloc failed for the bounce buffers");
        if (!h->pin_dn[k] && hipHostMalloc((void**)&h->pin_dn[k], BOUNCE_BYTES, hipHostMallocDefault) != hipSuccess)
            return fail(SDFHIP_E_NOMEM, "hipHostMalloc failed for the bounce buffers");
        if (!h->ev_up[k]) HIPC(hipEventCreateWithFlags(&h->ev_up[k], hipEventDisableTiming));
        if (!h->ev_dn[k]) HIPC(hipEventCreateWithFlags(&h->ev_dn[k], hipEventDisableTiming));
    }
    if (!h->team_up) h->team_up.reset(new sdfk::KdTeam(BOUNCE_THREADS - 1));
    if (!h->team_dn) h->team_dn.reset(new sdfk::KdTeam(BOUNCE_THREADS - 1));
    return 0;
}
// memcpy by the caller and the team's helpers, in 64-byte aligned slices
void team_memcpy(sdfk::KdTeam& team, void* dst, const void* src, size_t len) {
    const size_t per = ((len + BOUNCE_THREADS - 1) / BOUNCE_THREADS + 63) & ~(size_t)63;
    team.run(BOUNCE_THREADS, [&](int t) {
        const size_t lo = (size_t)t * per;
        if (lo < len) std::memcpy((char*)dst + lo, (const char*)src + lo, len - lo < per ? len - lo : per);
    });
}
// host -> device on `s`; returns once the last piece is enqueued (the source may be reused when it returns:
// every byte has left it).  One caller at a time per direction (the handle's mutex / the download thread).
hipError_t bounce_h2d(sdfhip_ctx* h, void* dst, const void* src, size_t bytes, hipStream_t s) {
    hipError_t e = hipSuccess;
    sdfk::KdTeam::Session awake(*h->team_up);  // (the helpers spin between the pieces of this copy only)
    for (size_t off = 0, i = 0; off < bytes && e == hipSuccess; off += BOUNCE_BYTES, ++i) {
        const int b = (int)(i & 1);
        const size_t len = bytes - off < BOUNCE_BYTES ? bytes - off : BOUNCE_BYTES;
        if (i >= 2) e = hipEventSynchronize(h->ev_up[b]);  // the DMA that last read this buffer
        if (e != hipSuccess) break;
        team_memcpy(*h->team_up, h->pin_up[b], (const char*)src + off, len);
        e = hipMemcpyAsync((char*)dst + off, h->pin_up[b], len, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(h->ev_up[b], s);
    }
    // the buffers are reused by the next call: wait for the last two DMAs (the bus time, not more)
    for (int b = 0; b < 2 && e == hipSuccess; ++b)
        if (bytes > (size_t)b * BOUNCE_BYTES) e = hipEventSynchronize(h->ev_up[b]);
    return e;
}
// device -> host on `s`, complete when it returns
hipError_t bounce_d2h(sdfhip_ctx* h, void* dst, const void* src, size_t bytes, hipStream_t s) {
    hipError_t e = hipSuccess;
    sdfk::KdTeam::Session awake(*h->team_dn);
    const size_t np = (bytes + BOUNCE_BYTES - 1) / BOUNCE_BYTES;
    auto len_of = [&](size_t i) { return bytes - i * BOUNCE_BYTES < BOUNCE_BYTES ? bytes - i * BOUNCE_BYTES : BOUNCE_BYTES; };
    for (size_t i = 0; i <= np && e == hipSuccess; ++i) {
        if (i < np) {  // piece i on the bus ...
            const int b = (int)(i & 1);
            e = hipMemcpyAsync(h->pin_dn[b], (const char*)src + i * BOUNCE_BYTES, len_of(i), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipEventRecord(h->ev_dn[b], s);
        }
        if (i >= 1 && e == hipSuccess) {  // ... while piece i - 1 goes home
            const int b = (int)((i - 1) & 1);
            e = hipEventSynchronize(h->ev_dn[b]);
            if (e == hipSuccess) team_memcpy(*h->team_dn, (char*)dst + (i - 1) * BOUNCE_BYTES, h->pin_dn[b], len_of(i - 1));
        }
    }
    return e;
}

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
    // Chunk schedule (start[c] .. start[c + 1]).  Every chunk pays ~1.6-2 ms of serial phases on the device
    // (sort passes, the tail of the long walks, the deferred lists) next to ~0.2 ms per million points, and
    // a page-locked upload moves a million points in ~0.22 ms: a batch of up to 8M points gains nothing from
    // being cut; a larger one starts with a quarter of the points (the only upload nothing overlaps) and
    // goes on in equal chunks of up to 12M (32M: 8 + 12 + 12, not 4 x 8).
    std::vector<int64_t> start{0};
    if (h->opt_host_chunk > 0) {
        for (int64_t o = h->opt_host_chunk; o < n; o += h->opt_host_chunk) start.push_back(o);
    } else if (h->robust && kind == Q_CONTAINS) {
        // the ray walk alone has next to no serial phase: quarters (2M .. 32M points) overlap best
        const int64_t per = std::min<int64_t>(std::max<int64_t>((n + 3) / 4, 1 << 21), 1 << 25);
        for (int64_t o = per; o < n; o += per) start.push_back(o);
    } else if (n > (8ll << 20)) {
        const int64_t first = std::min<int64_t>(n / 4, 8ll << 20), rest = n - first;
        const int64_t k = (rest + (12ll << 20) - 1) / (12ll << 20), per = (rest + k - 1) / k;
        for (int64_t o = first; o < n; o += per) start.push_back(o);
    }
    start.push_back(n);
    const int64_t nchunks = (int64_t)start.size() - 1;
    int64_t cap = 0;
    for (int64_t c = 0; c < nchunks; ++c) cap = std::max(cap, start[c + 1] - start[c]);
    if (int rc = ensure_host_staging(h, cap)) return rc;
    if (int rc = wait_for_queries(h, h->stream)) return rc;  // (this call ends synchronised: nothing to mark)
    // pageable arrays of a megabyte or more go through the page-locked bounce buffers (BounceCopy above)
    const bool pg_in = sizeof(float) * 3 * (size_t)n >= ((size_t)1 << 20) && is_pageable(pts);
    const bool pg_out = osz * (size_t)n >= ((size_t)1 << 20) && is_pageable(out);
    if (pg_in || pg_out)
        if (int rc = ensure_bounce(h)) return rc;
    if (nchunks == 1) {  // nothing to overlap: one stream, no helper thread (latency of small calls)
        hipStream_t s = h->stream;
        if (pg_in)
            HIPC(bounce_h2d(h, h->d_stage_in[0], pts, sizeof(float) * 3 * (size_t)n, s));
        else
            HIPC(hipMemcpyAsync(h->d_stage_in[0], pts, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
        if (int rc1 = run_query(h, kind, h->d_stage_in[0], n, trunc, h->d_stage_out[0], s)) {
            (void)hipStreamSynchronize(s);
            return rc1;
        }
        if (pg_out)
            HIPC(bounce_d2h(h, out, h->d_stage_out[0], osz * (size_t)n, s));
        else
            HIPC(hipMemcpyAsync(out, h->d_stage_out[0], osz * (size_t)n, hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        return 0;
    }
    auto span = [&](int64_t c) { return start[c + 1] - start[c]; };
    int rc = 0;
    hipError_t e = hipSuccess;
    static const bool dbg = std::getenv("SDFHIP_DEBUG_HOST") != nullptr;
    const auto T0 = std::chrono::steady_clock::now();
    auto now_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count(); };
    auto upload = [&](int64_t c) {
        const int k = (int)(c & 1);
        const double t0 = dbg ? now_ms() : 0.0;
        if (pg_in)
            e = bounce_h2d(h, h->d_stage_in[k], pts + 3 * start[c], sizeof(float) * 3 * (size_t)span(c), h->s_in);
        else
            e = hipMemcpyAsync(h->d_stage_in[k], pts + 3 * start[c], sizeof(float) * 3 * (size_t)span(c),
                               hipMemcpyHostToDevice, h->s_in);
        if (e == hipSuccess) e = hipEventRecord(h->ev_in[k], h->s_in);
        if (dbg) std::fprintf(stderr, "[host] upload %lld: %.2f .. %.2f\n", (long long)c, t0, now_ms());
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
            // The wait is on the host, not a hipStreamWaitEvent on s_out: copies queued behind a pending
            // stream wait made the runtime run the uploads on s_in at half speed while kernels were in flight
            // (32M pageable points: 21 ms instead of 14 ms per call; tools/numpy_call_probe.py).
            hipError_t ed = hipEventSynchronize(h->ev_done[k]);
            const double tdw = now_ms();
            if (ed == hipSuccess)
                ed = pg_out ? bounce_d2h(h, (char*)out + osz * (size_t)start[c], h->d_stage_out[k],
                                         osz * (size_t)span(c), h->s_out)
                            : hipMemcpyAsync((char*)out + osz * (size_t)start[c], h->d_stage_out[k],
                                             osz * (size_t)span(c), hipMemcpyDeviceToHost, h->s_out);
            const double td0 = now_ms();
            if (ed == hipSuccess) ed = hipStreamSynchronize(h->s_out);
            if (dbg) std::fprintf(stderr, "[host] download %lld: start %.2f issued %.2f done %.2f\n", (long long)c, tdw, td0, now_ms());
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
    if (int rc = wait_for_queries(h, (hipStream_t)stream)) return rc;
    const int rc = run_query(h, kind, d_pts, n, trunc, d_out, (hipStream_t)stream);
    if (n > 0)
        if (int rc2 = mark_query(h, (hipStream_t)stream)) return rc ? rc : rc2;
    return rc;
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

static int create_impl(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int robust,
                       int copy, int device, sdfhip_t** out, bool first_update) {
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
    *out = h;  // the caller destroys it if anything below fails or throws
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
    if (const char* e = getenv("SDFHIP_APPROX")) {  // default of option "approx" (tests run both packet walks)
        const long long v = atoll(e);
        h->opt_approx = v < 0 ? -1 : (v ? 1 : 0);
    }
    if (const char* e = getenv("SDFHIP_CUBE_LISTS")) {  // default of option "cube_lists" (tests run the suite with the lists forced)
        const long long v = atoll(e);
        h->opt_cube_lists = v < 0 ? -1 : (v ? 1 : 0);
    }
    if (const char* e = getenv("SDFHIP_CUBE_LEVEL")) {
        const long long v = atoll(e);
        h->opt_cube_level = v < 0 ? 0 : (v > 6 ? 6 : v);
    }
    if (const char* e = getenv("SDFHIP_COOP_BUDGET")) {  // default of option "coop_budget" (tuning: internal handles too)
        const long long v = atoll(e);
        h->opt_coop_budget = v < 0 ? -1 : v;
    }
    if (const char* e = getenv("SDFHIP_SCAN_MAX")) {  // default of option "scan_max" (tests pin the tree paths with 0)
        const long long v = atoll(e);
        h->opt_scan_max = v < 0 ? -1 : v;
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
    A(scratch_setup(device));  // this device's address of the static scratch (no-alloc entry)
    if (!rc && hipEventCreateWithFlags(&h->ev_query, hipEventDisableTiming) != hipSuccess)
        rc = fail(SDFHIP_E_HIP, "hipEventCreate failed");
    if (!rc && hipHostMalloc((void**)&h->h_counts, 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess)
        rc = fail(SDFHIP_E_NOMEM, "hipHostMalloc failed");
    if (!rc) h->h_counts[0] = h->h_counts[1] = 0;
    A(dalloc(&h->d_verts, 3 * (size_t)nv));
    A(dalloc(&h->d_faces, 3 * (size_t)nf));
    A(dalloc(&h->ws.fb_count, 8));
    A(dalloc(&h->ws.heavy_pos, HEAVY_SHORT + 64));
    A(dalloc(&h->d_adj_off, (size_t)nv + 1));
    A(dalloc(&h->d_adj_faces, 3 * (size_t)nf));
    A(dalloc(&h->d_geo, 4 * (size_t)nf));
    A(dalloc(&h->d_geo_adj, 12 * (size_t)nf));
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
    // sdf.cpp:28 update(false): same work, kd built once.  (Not for the non-leading members of a group: they
    // receive the leader's tree, group.h.)
    if (!rc && first_update) rc = do_update(h, true);
    return rc;
}

static int create_handle(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int robust,
                         int copy, int device, sdfhip_t** out, bool first_update) {
    if (!out) return fail(SDFHIP_E_INVALID, "out is null");
    *out = nullptr;
    const int rc = guarded([&]() -> int { return create_impl(verts, nv, faces, nf, robust, copy, device, out, first_update); });
    if (rc && *out) {
        const std::string keep = g_err;
        sdfhip_destroy(*out);
        g_err = keep;
        *out = nullptr;
    }
    return rc;
}

int sdfhip_create(const float* verts, uint32_t nv, const uint32_t* faces, uint32_t nf, int robust,
                  int copy, int device, sdfhip_t** out) {
    return create_handle(verts, nv, faces, nf, robust, copy, device, out, true);
}

void sdfhip_destroy(sdfhip_t* h) {
    if (!h) return;
    DevGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->ev_query) {
        if (h->query_pending) (void)hipEventSynchronize(h->ev_query);  // a caller-stream query may still run
        (void)hipEventDestroy(h->ev_query);
    }
    void* ptrs[] = {h->d_verts, h->d_faces, h->d_kd_pn, h->d_kd_split, h->d_kd_gref, h->d_kd_gobb, h->d_kd_gobb_aos, h->d_kd_l1obb,
                    h->ws.cube_par, h->ws.cube_hdr, h->ws.cube_list, h->d_kd_vind, h->d_adj_off,
                    h->d_adj_faces, h->d_geo, h->d_geo_adj, h->d_rface, h->d_area_cum, h->d_fbox, h->d_normals, h->d_areas,
                    h->d_rnodes, h->d_sorted_face, h->d_codes, h->d_maxdepth, h->topo.lo, h->topo.hi,
                    h->topo.split, h->topo.parentI, h->topo.parentL, h->topo.posI, h->topo.posL,
                    h->topo.depthI, h->ws.qkeys, h->ws.qperm, h->ws.qbounds, h->d_gstack, h->d_kd_pts4, h->d_kd_apx4, h->d_kd_actr, h->d_apx_r2, h->d_kd_lbox,
                    h->ws.fb_list, h->ws.fb_count, h->ws.heavy_pos, h->ws.coop_list, h->ws.coop_state, h->ws.amb_mask, h->ws.amb_cnt, h->ws.amb_s1, h->ws.amb_s2,
                    h->d_vkeys, h->d_vidx};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->ws.sorter.release();
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    for (int k = 0; k < 2; ++k) {
        if (h->pin_up[k]) (void)hipHostFree(h->pin_up[k]);
        if (h->pin_dn[k]) (void)hipHostFree(h->pin_dn[k]);
        if (h->ev_up[k]) (void)hipEventDestroy(h->ev_up[k]);
        if (h->ev_dn[k]) (void)hipEventDestroy(h->ev_dn[k]);
    }
    h->grid.release();
    h->grid_x.release();
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
    return guarded([&]() -> int { return do_update(h, true); });
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
    return guarded([&]() -> int { return do_update(h, true); });
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
    if (!h->robust || pin == 2) return 0;  // (2: stored now, applied by the next sdfhip_update)
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    if (int rc = wait_for_queries(h, h->stream)) return rc;
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
    return guarded([&]() -> int { return run_query_host(h, Q_SDF, p, n, trunc, out); });
}
int sdfhip_sdf_device(sdfhip_t* h, const float* p, int64_t n, int trunc, float* out, void* s) {
    return guarded([&]() -> int { return run_query_dev(h, Q_SDF, p, n, trunc, out, s); });
}
int sdfhip_sdf_device_noalloc(sdfhip_t* h, const float* p, int64_t n, int trunc, float* out, void* stream) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    if (n > NOALLOC_MAX_POINTS)
        return fail(SDFHIP_E_INVALID, "sdfhip_sdf_device_noalloc takes at most %lld points per call (got %lld)",
                    (long long)NOALLOC_MAX_POINTS, (long long)n);
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(h->mu);
        DevGuard g(h->device);
        if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
        QueryWs* W = nullptr;
        if (h->device >= 0 && h->device < MAX_DEVICES) {
            std::lock_guard<std::mutex> ls(g_scratch_mu);
            W = g_scratch_ws[h->device];
        }
        if (!W) return fail(SDFHIP_E_STATE, "no static scratch on device %d", h->device);
        // Under stream capture (hipStreamBeginCapture / torch.cuda.graph) the call only records its kernels and
        // memsets into the graph.  The cross-stream ordering events of the handle stay out of it: an event
        // recorded inside a capture belongs to the graph and could not be waited for by update() later.  The
        // caller orders graph launches against update() and against queries on other streams, and captures
        // again after an update (the graph holds the mesh's pointers and sizes as they were).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = hipStreamCaptureStatusNone;
        }
        const bool capturing = cap != hipStreamCaptureStatusNone;
        if (!capturing)
            if (int rc = wait_for_queries(h, (hipStream_t)stream)) return rc;
        // Every buffer of this call comes out of the static scratch; nothing is allocated — no hipMalloc, and no
        // event either: option "profile" times the other entry points only (ProfScope skips a fixed workspace).
        struct UseScratch {  // the handle's own workspace is back whatever run_query does, exceptions included
            sdfhip_ctx* h;
            UseScratch(sdfhip_ctx* h_, QueryWs* W_) : h(h_) { h->W = W_; }
            ~UseScratch() { h->W = &h->ws; }
        };
        int rc;
        {
            UseScratch use(h, W);
            rc = run_query(h, Q_SDF, p, n, trunc, out, (hipStream_t)stream);
        }
        if (n > 0 && !capturing)
            if (int rc2 = mark_query(h, (hipStream_t)stream)) return rc ? rc : rc2;
        return rc;
    });
}
int sdfhip_contains(sdfhip_t* h, const float* p, int64_t n, uint8_t* out) {
    return guarded([&]() -> int { return run_query_host(h, Q_CONTAINS, p, n, 0, out); });
}
int sdfhip_contains_device(sdfhip_t* h, const float* p, int64_t n, uint8_t* out, void* s) {
    return guarded([&]() -> int { return run_query_dev(h, Q_CONTAINS, p, n, 0, out, s); });
}
int sdfhip_nn(sdfhip_t* h, const float* p, int64_t n, int32_t* out) {
    return guarded([&]() -> int { return run_query_host(h, Q_NN, p, n, 0, out); });
}
int sdfhip_nn_device(sdfhip_t* h, const float* p, int64_t n, int32_t* out, void* s) {
    return guarded([&]() -> int { return run_query_dev(h, Q_NN, p, n, 0, out, s); });
}
int sdfhip_raycast_detail(sdfhip_t* h, const float* p, int64_t n, int32_t* out3) {
    if (h && !h->robust) return fail(SDFHIP_E_STATE, "raycast detail needs a robust handle");
    return guarded([&]() -> int { return run_query_host(h, Q_DETAIL, p, n, 0, out3); });
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
    if (int rc = guarded([&]() -> int { return fetch_face_host(h); })) return rc;
    float s = 0.f;
    for (float a : h->h_areas) s += a;
    *out = s;
    return 0;
}
int sdfhip_face_areas(sdfhip_t* h, float* out) {
    if (!h || !out) return fail(SDFHIP_E_INVALID, "null handle/out");
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc = guarded([&]() -> int { return fetch_face_host(h); })) return rc;
    std::memcpy(out, h->h_areas.data(), sizeof(float) * h->h_areas.size());
    return 0;
}
int sdfhip_face_normals(sdfhip_t* h, float* out) {
    if (!h || !out) return fail(SDFHIP_E_INVALID, "null handle/out");
    std::lock_guard<std::mutex> lk(h->mu);
    if (int rc = guarded([&]() -> int { return fetch_face_host(h); })) return rc;
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
    o->kd_nodes = h->fast_active ? (int32_t)h->n_groups : (int32_t)h->kd_nnodes;
    o->kd_depth = h->fast_active ? 2 : h->kd_depth;
    o->kd_builds = (int32_t)(h->kd_builds > 0x7FFFFFFF ? 0x7FFFFFFF : h->kd_builds);
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
    o->last_kernel = h->last_kernel;
    if (h->h_counts) {
        // the counters arrive by an asynchronous copy on the query's stream: wait for that query
        DevGuard g(h->device);
        if (h->query_pending && h->ev_query) (void)hipEventSynchronize(h->ev_query);
        o->last_fallback_points = h->h_counts[0];
        o->last_heavy_points = h->h_counts[1];
    }
    return 0;
}
int sdfhip_sample_surface_device(sdfhip_t* h, int64_t n, uint64_t seed, float* d_out, void* stream) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    return guarded([&]() -> int {
        if (int rc = wait_for_queries(h, (hipStream_t)stream)) return rc;
        if (int rc = run_sample_surface(h, n, seed, d_out, (hipStream_t)stream)) return rc;
        return n > 0 ? mark_query(h, (hipStream_t)stream) : 0;  // update() must not overtake it
    });
}
int sdfhip_sample_surface(sdfhip_t* h, int64_t n, uint64_t seed, float* out) {
    if (!h) return fail(SDFHIP_E_INVALID, "null handle");
    if (n > 0 && !out) return fail(SDFHIP_E_INVALID, "null output pointer");
    std::lock_guard<std::mutex> lk(h->mu);
    DevGuard g(h->device);
    if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
    if (n <= 0) return guarded([&]() -> int { return run_sample_surface(h, n, seed, nullptr, h->stream); });
    float* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(float) * 3 * (size_t)n) != hipSuccess)
        return fail(SDFHIP_E_NOMEM, "hipMalloc failed for %lld samples", (long long)n);
    int rc = guarded([&]() -> int { return run_sample_surface(h, n, seed, d, h->stream); });
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, d, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(SDFHIP_E_HIP, "HIP error in sample_surface: %s", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}
#ifdef SDFK_STATS
int sdfhip_debug_ray_stats(unsigned long long out[8], int reset) {
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_stats), 8 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIPC(hipMemcpyToSymbol(HIP_SYMBOL(g_ray_stats), z, sizeof(z)));
    }
    return 0;
}
// tuning build: the cube tables of the handle's last launch (device pointers; header = 3 float4 per cube)
int sdfhip_debug_cube_tables(sdfhip_t* h, void** hdr, void** list, unsigned long long* ncubes) {
    uint32_t par[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (h->ws.cube_par) HIPC(hipMemcpy(par, h->ws.cube_par, sizeof(par), hipMemcpyDeviceToHost));
    *hdr = h->ws.cube_hdr, *list = h->ws.cube_list, *ncubes = (unsigned long long)par[6] * par[7] * par[8];
    return 0;
}
int sdfhip_debug_packet_stats(unsigned long long out[12], int reset) {
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_stats), 12 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
    else if (!strcmp(key, "far_packets"))
        h->opt_far = v;
    else if (!strcmp(key, "far_radius_sparse_x10"))
        h->opt_far_sparse_x10 = v < 11 ? 11 : v;
    else if (!strcmp(key, "far_radius_x10"))
        h->opt_far_x10 = v < 11 ? 11 : v;
    else if (!strcmp(key, "profile")) {
        h->opt_profile = v;
        if (v == 2) h->prof_ms = 0.0, h->prof_launches = 0, h->prof_points = 0;  // reset
    }
    else if (!strcmp(key, "build_threads"))
        h->opt_build_threads = v < 1 ? 1 : v;
    else if (!strcmp(key, "packet"))
        h->opt_packet = v;
    else if (!strcmp(key, "approx"))
        h->opt_approx = v < 0 ? -1 : (v ? 1 : 0);
    else if (!strcmp(key, "cube_lists"))
        h->opt_cube_lists = v < 0 ? -1 : (v ? 1 : 0);
    else if (!strcmp(key, "cube_min")) {
        if (v < 0) return fail(SDFHIP_E_INVALID, "cube_min must be >= 0");
        h->opt_cube_min = v;
    } else if (!strcmp(key, "cube_level")) {
        if (v < 0 || v > 6) return fail(SDFHIP_E_INVALID, "cube_level must be 0 (automatic) or in [1, 6]");
        h->opt_cube_level = v;
    }
    else if (!strcmp(key, "scan_max"))
        h->opt_scan_max = v < 0 ? -1 : v;
    else if (!strcmp(key, "heavy_budget")) {
        if (v < 0 || v > 0x7FFFFFFF) return fail(SDFHIP_E_INVALID, "heavy_budget must be >= 0 (0 = automatic)");
        h->opt_heavy_budget = v;
    } else if (!strcmp(key, "coop_budget")) {
        if (v < -1 || v > 0x7FFFFFFF) return fail(SDFHIP_E_INVALID, "coop_budget must be >= -1 (0 = no cooperative kernel, -1 = automatic)");
        h->opt_coop_budget = v;
    } else if (!strcmp(key, "coop_max")) {
        if (v < 0) return fail(SDFHIP_E_INVALID, "coop_max must be >= 0");
        h->opt_coop_max = v;
    } else if (!strcmp(key, "fast_nn"))
        h->opt_fast_nn = v ? 1 : 0;  // takes effect at the next sdfhip_update
    else if (!strcmp(key, "ray_mode")) {
        if (v < 0 || v > 2) return fail(SDFHIP_E_INVALID, "ray_mode must be 0 (auto), 1 (grid) or 2 (LBVH)");
        h->opt_ray_mode = v;
        if (h->robust && h->nf > 0 && h->d_fbox && wants_lbvh(h)) {  // the LBVH is only kept current while something walks it
            DevGuard g(h->device);
            if (!g.ok) return fail(SDFHIP_E_HIP, "hipSetDevice(%d) failed", h->device);
            if (int rc = wait_for_queries(h, h->stream)) return rc;
            return guarded([&]() -> int { return ensure_lbvh(h); });
        }
    } else if (!strcmp(key, "grid_per_face")) {
        if (v < 1 || v > 4096) return fail(SDFHIP_E_INVALID, "grid_per_face must be in [1, 4096]");
        h->opt_grid_per_face = v;  // takes effect at the next update / set_raycast_axes
    } else if (!strcmp(key, "grid_tri")) {
        h->opt_grid_tri = v ? 1 : 0;  // takes effect at the next update / set_raycast_axes
    } else if (!strcmp(key, "host_chunk")) {
        if (v != 0 && (v < 1024 || v > (1ll << 27)))
            return fail(SDFHIP_E_INVALID, "host_chunk must be 0 (automatic) or in [1024, 2^27]");
        h->opt_host_chunk = v;
    } else if (!strcmp(key, "packet_group")) {
        // vertices per packet group; takes effect at the next sdfhip_update
        if (v != 0 && (v < 10 || v > (int64_t)sdfk::KD_GROUP_LIMIT))
            return fail(SDFHIP_E_INVALID, "packet_group must be 0 (automatic) or in [10, %u]", sdfk::KD_GROUP_LIMIT);
        h->opt_packet_group = v;
    } else if (!strcmp(key, "sort_min"))
        h->opt_sort_min = v;
    else if (!strcmp(key, "sort_bits")) {
        if (v != 24 && v != 30) return fail(SDFHIP_E_INVALID, "sort_bits must be 24 or 30");
        h->opt_sort_bits = v;
    } else if (!strcmp(key, "sort_min_contains"))
        h->opt_sort_min_contains = v;
    else if (!strcmp(key, "column_sort"))
        h->opt_column_sort = v ? 1 : 0;
    else if (!strcmp(key, "x_grid"))
        h->opt_x_grid = v ? 1 : 0;  // from the next update on (0: the x+ ray walks the LBVH)
    else if (!strcmp(key, "chunk")) {
        if (v < 1024 || v > (1ll << 27)) return fail(SDFHIP_E_INVALID, "chunk must be in [1024, 2^27]");
        h->opt_chunk = v;
    } else
        return fail(SDFHIP_E_INVALID, "unknown option '%s'", key);
    return 0;
}

}  // extern "C"

#include "renderer.h"

#include "group.h"
