/* sdfhip.h — C ABI of the MI355X-native batched point-query path of sxyu/sdf.
 *
 * This is the drop-in boundary: every entry point below replaces one member of the
 * reference's `sdf::SDF` (include/sdf/sdf.hpp:150-258, src/sdf.cpp:331-395) and is what the
 * reference's own C++ façade / pybind module (pybind.cpp:18-83) would bind.  Plain pointers
 * and sizes only; no C++ or torch types.  All "file:line" citations are into /root/reference.
 *
 * Conventions
 *   - return 0 on success, a negative SDFHIP_E_* code on failure; never throws.
 *     `sdfhip_last_error()` returns a thread-local message for the last failure.
 *   - points / verts are row-major float32 [N,3]; faces row-major uint32 [F,3]
 *     (sdf.hpp:43-46).  Results: float32 [N] (SDF, positive inside, sdf.hpp:174),
 *     uint8 [N] (contains, 0/1), int32 [N] (nn, index into the input verts).
 *   - the caller owns every buffer.  Host entry points take host pointers; the `_device`
 *     variants take device pointers on the handle's GPU plus a HIP stream (hipStream_t
 *     passed as void*, NULL = the null stream) and are asynchronous on that stream.
 *   - a handle is bound to one GPU.  Query calls on one handle are serialised by the
 *     library (they share the handle's query workspace); `sdfhip_update` is exclusive.
 *     This holds on the device too: a `_device` call on another stream than the previous
 *     one, and `sdfhip_update`, are ordered after the previous call's kernels by an event,
 *     so an asynchronous query is never overtaken by an update or by a query on a second
 *     stream.
 *   - `n_threads` of the reference API (sdf.hpp:187-204) has no meaning on the GPU and is
 *     accepted and ignored by the façades above this ABI.
 *   - there is NO CPU fallback: without a usable HIP device `sdfhip_create` fails.
 */
#ifndef SDFHIP_H_
#define SDFHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdfhip_ctx sdfhip_t;

enum {
    SDFHIP_OK = 0,
    SDFHIP_E_INVALID = -1, /* bad argument */
    SDFHIP_E_HIP = -2,     /* a HIP runtime call failed */
    SDFHIP_E_NODEVICE = -3,
    SDFHIP_E_NOMEM = -4,
    SDFHIP_E_STATE = -5 /* e.g. mutable access on a non-owning handle */
};

/* Replaces SDF::SDF(verts, faces, robust, copy) (sdf.hpp:168-169, sdf.cpp:331-341) and
 * Impl::Impl (sdf.cpp:17-29): builds vertex->incident-face lists, then runs update().
 * copy = 0: the handle keeps `verts`/`faces` pointers for its lifetime and re-reads `verts`
 *           on every sdfhip_update (Eigen::Ref semantics, sdf.cpp:339).
 * copy = 1: the handle owns host copies (sdf.cpp:334-337); see sdfhip_verts_mutable.
 * device  : HIP device ordinal, or -1 for the current device.
 * The ray-cast rotation (sdf.cpp:43-48) is drawn from a time-seeded RNG as in the
 * reference (util.cpp:32-40) unless pinned with sdfhip_set_raycast_axes. */
int sdfhip_create(const float* verts, uint32_t n_verts, const uint32_t* faces, uint32_t n_faces,
                  int robust, int copy, int device, sdfhip_t** out);

/* Replaces SDF::~SDF (sdf.cpp:343). */
void sdfhip_destroy(sdfhip_t* h);

/* Replaces SDF::update() -> Impl::update(true) (sdf.cpp:391, 31-73): re-reads the vertex
 * positions, rebuilds the vertex kd-tree, mesh AABB, per-face normals/areas, draws a new
 * rotation (unless pinned) and refits the face LBVH. */
int sdfhip_update(sdfhip_t* h);

/* Same as sdfhip_update but the new vertex positions are read from a device buffer
 * (float32 [n_verts,3]) on `stream`; the host copy / borrowed host pointer is refreshed
 * too when the handle owns its data.  No counterpart in the reference (its vertices live
 * in host memory); for deforming meshes produced on the GPU (BASELINE.json config 5). */
int sdfhip_update_from_device(sdfhip_t* h, const float* d_verts, void* stream);

/* The reference's rotation is private and non-deterministic (sdf.cpp:271, util.cpp:32-40),
 * so parity is only defined "given the same R".  set: pins R (row-major 3x3, used as
 * `p * R`, sdf.cpp:284-285) for this and all later updates and re-runs the rotation-
 * dependent part of update().  pin = 2 pins R without that re-run: it takes effect at the next
 * sdfhip_update.  pin = 0 un-pins (later updates draw a fresh R). */
int sdfhip_set_raycast_axes(sdfhip_t* h, const float R[9], int pin);
int sdfhip_get_raycast_axes(const sdfhip_t* h, float R[9]);

/* Replaces SDF::operator()(points, trunc_aabb, n_threads) -> Impl::calc
 * (sdf.cpp:377-379, 93-158). */
int sdfhip_sdf(sdfhip_t* h, const float* points, int64_t n, int trunc_aabb, float* out);
int sdfhip_sdf_device(sdfhip_t* h, const float* d_points, int64_t n, int trunc_aabb,
                      float* d_out, void* stream);

/* sdfhip_sdf_device for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments and the
 * same results as sdfhip_sdf_device, but the call allocates nothing (no hipMalloc / hipFree, no
 * event or stream creation): every intermediate buffer (query sort keys and permutation, radix-sort
 * histograms, deferred-query lists and their counters) lives in one static device array of 26 MiB
 * that each GPU the library runs on holds until the process ends.  Limit: at most
 * SDFHIP_NOALLOC_MAX_POINTS points per call (more: SDFHIP_E_INVALID); meshes whose kd-tree is
 * deeper than 40 levels are refused (SDFHIP_E_STATE).  All calls on one GPU — from any handle —
 * share that array: calls that overlap on one GPU (different handles on different streams) race;
 * calls on one handle are ordered as usual.  Option "profile" does not time this entry (it would create
 * events).  On a capturing stream (hipStreamBeginCapture, torch.cuda.graph) the call records its kernels and
 * memsets into the graph and leaves the handle's cross-stream ordering events out of it: the caller orders
 * graph launches against sdfhip_update and against queries on other streams, and captures again after an
 * update (a graph holds the mesh's device pointers and sizes as they were at capture). */
#define SDFHIP_NOALLOC_MAX_POINTS (1 << 20)
int sdfhip_sdf_device_noalloc(sdfhip_t* h, const float* d_points, int64_t n, int trunc_aabb,
                              float* d_out, void* stream);

/* Replaces SDF::contains(points, n_threads) -> Impl::contains (sdf.cpp:385-389, 160-174). */
int sdfhip_contains(sdfhip_t* h, const float* points, int64_t n, uint8_t* out);
int sdfhip_contains_device(sdfhip_t* h, const float* d_points, int64_t n, uint8_t* d_out,
                           void* stream);

/* Replaces SDF::nn(points, n_threads) -> Impl::nn (sdf.cpp:381-383, 75-91). */
int sdfhip_nn(sdfhip_t* h, const float* points, int64_t n, int32_t* out);
int sdfhip_nn_device(sdfhip_t* h, const float* d_points, int64_t n, int32_t* d_out, void* stream);

/* Debug/verification aid (no counterpart: the reference's per-ray parities are locals of
 * Impl::_raycast, sdf.cpp:292-327): out3[i] = {z+, z-, x+ or -1 if not cast}; all -1 when
 * the point is outside the closed mesh AABB.  Robust handles only. */
int sdfhip_raycast_detail(sdfhip_t* h, const float* points, int64_t n, int32_t* out3);

/* Data accessors (sdf.hpp:215-245, sdf.cpp:345-375). */
int sdfhip_robust(const sdfhip_t* h);
int sdfhip_own_data(const sdfhip_t* h);
int sdfhip_device(const sdfhip_t* h);
uint32_t sdfhip_num_verts(const sdfhip_t* h);
uint32_t sdfhip_num_faces(const sdfhip_t* h);
/* SDF::verts() / faces(): the host arrays the handle currently reads (sdf.cpp:359, 368). */
const float* sdfhip_verts(const sdfhip_t* h);
const uint32_t* sdfhip_faces(const sdfhip_t* h);
/* SDF::verts_mutable() (sdf.cpp:369-375): pointer to the owned copy; NULL and
 * SDFHIP_E_STATE (message "'verts' is non mutable, construct with copy=True") when the
 * handle does not own its data.  Call sdfhip_update after writing. */
float* sdfhip_verts_mutable(sdfhip_t* h);
/* SDF::faces_mutable() (sdf.cpp:360-366).  As in the reference, faces must not actually
 * change for the lifetime of the handle (sdf.hpp:162-163). */
uint32_t* sdfhip_faces_mutable(sdfhip_t* h);
/* SDF::surface_area() (sdf.cpp:349): sum of face_areas (each 2x the geometric area,
 * sdf.cpp:68). */
int sdfhip_surface_area(sdfhip_t* h, float* out);
/* SDF::face_areas() / face_normals() / aabb() (sdf.cpp:351-357): copied into caller
 * buffers of F, 3F and 6 floats. */
int sdfhip_face_areas(sdfhip_t* h, float* out);
int sdfhip_face_normals(sdfhip_t* h, float* out);
int sdfhip_aabb(const sdfhip_t* h, float out[6]);
/* SDF::adj_faces(pointid) (sdf.cpp:345-347): incident face ids, ascending.
 * Returns the count (>= 0) or a negative error; *ids points into handle-owned memory. */
int sdfhip_adj_faces(const sdfhip_t* h, uint32_t vert, const int32_t** ids);

/* Replaces SDF::sample_surface(num_points) (sdf.hpp:206-209, sdf.cpp:176-234): n points on the
 * surface, float32 [n,3]; face i with probability face_areas[i] / surface_area through the
 * running sum of face areas, then a uniform point of the parallelogram (a, b-a, c-a) reflected
 * over bc.  The reference draws from a time-seeded mt19937 (util.cpp:32-40); here sample j uses
 * Philox4x32-10(key = seed, counter = j), so equal seeds give equal samples and the
 * distribution, not the values, is what matches the reference.  No faces: SDFHIP_E_STATE with
 * the reference's message. */
int sdfhip_sample_surface(sdfhip_t* h, int64_t n, uint64_t seed, float* out);
int sdfhip_sample_surface_device(sdfhip_t* h, int64_t n, uint64_t seed, float* d_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One sdf::SDF on several GPUs of one process (no counterpart: the reference is CPU-only, its
 * parallelism is maybe_parallel_for over point indices, util.cpp:7-29).  A group holds one handle per
 * listed device (an ordinal may repeat), all holding the same mesh and the same ray-cast rotation.
 * The mesh is built ONCE per create / update, like the single update() of sdf.cpp:31-73: member 0 uploads
 * the vertices and builds the nanoflann-exact kd-tree on the host with all build threads; every other
 * member receives the vertices and the built tree arrays (kd nodes, permuted vertices, group refs: ~6 MB at
 * 180k vertices) device to device — hipMemcpyPeerAsync, i.e. over xGMI between two GPUs of a node — and runs
 * only the device-side part of the update (per-face precompute, column grids, oriented boxes) on its own
 * GPU, concurrently with member 0's host build.
 * The host-pointer queries below replace SDF::operator() / contains / nn like their
 * single-handle namesakes: the N points are cut into contiguous ranges [r * ceil(N/G), ...), member r
 * evaluates its range on its own GPU (own host thread, staging buffers, stream and PCIe link) and
 * writes into the same range of `out`.  There is no data-path collective.  copy = 1: member 0 owns
 * the host copy (sdfhip_verts_mutable(sdfhip_group_member(g, 0))), the others read it.
 * sdfhip_group_update replaces SDF::update(): member 0 draws the new rotation (unless pinned through
 * sdfhip_group_set_raycast_axes), every member then re-reads the vertices and rebuilds concurrently. */
typedef struct sdfhip_group sdfhip_group_t;
int sdfhip_group_create(const float* verts, uint32_t n_verts, const uint32_t* faces, uint32_t n_faces,
                        int robust, int copy, const int* devices, int n_devices, sdfhip_group_t** out);
void sdfhip_group_destroy(sdfhip_group_t* g);
int sdfhip_group_size(const sdfhip_group_t* g);
sdfhip_t* sdfhip_group_member(sdfhip_group_t* g, int k);
int sdfhip_group_update(sdfhip_group_t* g);
int sdfhip_group_set_raycast_axes(sdfhip_group_t* g, const float R[9], int pin);
int sdfhip_group_get_raycast_axes(const sdfhip_group_t* g, float R[9]);
int sdfhip_group_sdf(sdfhip_group_t* g, const float* points, int64_t n, int trunc_aabb, float* out);
int sdfhip_group_contains(sdfhip_group_t* g, const float* points, int64_t n, uint8_t* out);
int sdfhip_group_nn(sdfhip_group_t* g, const float* points, int64_t n, int32_t* out);
/* The same three queries on device-resident shards (no counterpart; the shape of BASELINE.json configs[3]:
 * every GPU holds its own range of the points): member k evaluates counts[k] points at d_points[k] into
 * d_out[k], both on member k's GPU, asynchronously on streams[k] (hipStream_t as void*; streams == NULL or
 * streams[k] == NULL: the null stream of that GPU).  counts[k] may be 0 (pointers ignored).  The launches of
 * the members are issued concurrently, one host thread per member; like the `_device` entry points of a single
 * handle the call returns once everything is enqueued. */
int sdfhip_group_sdf_device(sdfhip_group_t* g, const float* const* d_points, const int64_t* counts,
                            int trunc_aabb, float* const* d_out, void* const* streams);
int sdfhip_group_contains_device(sdfhip_group_t* g, const float* const* d_points, const int64_t* counts,
                                 uint8_t* const* d_out, void* const* streams);
int sdfhip_group_nn_device(sdfhip_group_t* g, const float* const* d_points, const int64_t* counts,
                           int32_t* const* d_out, void* const* streams);
/* sdfhip_group_update with the new vertex positions in a device buffer (float32 [n_verts,3]) on member 0's
 * GPU, read on `stream` (the group's counterpart of sdfhip_update_from_device; copy = 1 groups only). */
int sdfhip_group_update_from_device(sdfhip_group_t* g, const float* d_verts, void* stream);

/* Statistics of the structures behind the handle (no counterpart in the reference). */
typedef struct {
    int32_t kd_nodes;      /* internal nodes of the vertex kd-tree */
    int32_t kd_depth;      /* max number of internal nodes on a root-to-leaf path */
    int32_t bvh_nodes;     /* nodes of the threaded face LBVH (0 when non-robust) */
    int32_t bvh_depth;
    float last_update_ms;  /* wall time of the last create/update */
    float last_kd_build_ms;
    /* live hipEvent timing of the query kernels, accumulated while option "profile" != 0:
     * total kernel time, number of launches and points they processed */
    double query_kernel_ms;
    int64_t query_kernel_launches;
    int64_t query_kernel_points;
    /* queries of the last packet-kernel launch that were deferred to the strict kernel */
    int64_t last_fallback_points;
    /* queries of the last call whose wave exceeded its work budget in the packet pass and were
     * answered by the one-wave-per-query kernel (option "heavy_budget") */
    int64_t last_heavy_points;
    /* the kernel that answered (most of) the last query call: SDFHIP_KERNEL_* */
    int32_t last_kernel;
    /* host kd-tree builds this handle has run so far (create + updates).  In a group only member 0 builds:
     * the other members' count stays 0 and their last_kd_build_ms is 0. */
    int32_t kd_builds;
} sdfhip_stats_t;
enum {
    SDFHIP_KERNEL_NONE = 0,
    SDFHIP_KERNEL_STRICT = 1,        /* k_query: one lane per query, exact nanoflann emulation */
    SDFHIP_KERNEL_PACKET_EXACT = 2,  /* k_query_packet<kind, robust, false>: wide walk, exact evaluation */
    SDFHIP_KERNEL_PACKET_APPROX = 3, /* k_query_packet<kind, robust, true>: approximate first pass + exact second phase */
    SDFHIP_KERNEL_SCAN = 4           /* k_query_heavy in scan mode: small batches */
};
int sdfhip_get_stats(const sdfhip_t* h, sdfhip_stats_t* out);

/* ---------------------------------------------------------------------------------------------
 * sdf::Renderer (sdf.hpp:282-369, src/renderer.cpp): image-space raycaster over the mesh seen by a
 * pinhole camera at the origin facing +z.  Points are row-major float32 [N,2] image coordinates
 * (x, y); images are row-major [height, width], pixel (r, c) = the point ((float)c, (float)r)
 * (renderer.cpp:187-189).  Same conventions as above (0 / negative codes, caller-owned buffers).
 * Results equal the reference's except for one documented rule: when two different vertices of
 * exactly equal z compete for a pixel's vertex id, the reference keeps the one its R-tree meets
 * first; here the lowest face id wins (order independent). */
typedef struct sdfhip_renderer sdfhip_renderer_t;

/* Replaces Renderer::Renderer(verts, faces, width, height, fx, fy, cx, cy, copy) (sdf.hpp:298-301,
 * renderer.cpp:256-269) + Impl::Impl (renderer.cpp:20-31).  copy as in sdfhip_create. */
int sdfhip_renderer_create(const float* verts, uint32_t n_verts, const uint32_t* faces, uint32_t n_faces,
                           int width, int height, float fx, float fy, float cx, float cy, int copy,
                           int device, sdfhip_renderer_t** out);
/* Replaces Renderer::~Renderer (renderer.cpp:321). */
void sdfhip_renderer_destroy(sdfhip_renderer_t* r);
/* Replaces Renderer::update() (renderer.cpp:286, 33-60): re-reads verts and faces, re-projects,
 * rebuilds the face structure; the 2-D nearest-vertex index is rebuilt at its next use. */
int sdfhip_renderer_update(sdfhip_renderer_t* r);
/* Replace Renderer::render_depth / render_mask / render_nn (sdf.hpp:310-324, renderer.cpp:93-116):
 * out is [height * width]; depth 0 = nothing there; nn -1 = nothing there unless fill_outside
 * (then the 2-D nearest projected vertex, nanoflann's tie rule). */
int sdfhip_renderer_render_depth(sdfhip_renderer_t* r, float* out);
int sdfhip_renderer_render_mask(sdfhip_renderer_t* r, uint8_t* out);
int sdfhip_renderer_render_nn(sdfhip_renderer_t* r, int fill_outside, int32_t* out);
/* Replace Renderer::operator() / contains / nn on continuous image points (sdf.hpp:328-340,
 * renderer.cpp:118-140, 306-319). */
int sdfhip_renderer_depth(sdfhip_renderer_t* r, const float* points2, int64_t n, float* out);
int sdfhip_renderer_contains(sdfhip_renderer_t* r, const float* points2, int64_t n, uint8_t* out);
int sdfhip_renderer_nn(sdfhip_renderer_t* r, const float* points2, int64_t n, int fill_outside,
                       int32_t* out);
/* The same queries on device pointers, asynchronous on `stream` (no counterpart).  kind: 0 depth
 * (float32 out), 1 mask (uint8), 2 vertex id (int32).  d_points2 == NULL: the image pixels
 * (n = height * width). */
int sdfhip_renderer_query_device(sdfhip_renderer_t* r, int kind, const float* d_points2, int64_t n,
                                 int fill_outside, void* d_out, void* stream);
/* Accessors: Renderer::verts / faces / verts_mutable / faces_mutable / own_data
 * (sdf.hpp:348-356, renderer.cpp:288-304). */
int sdfhip_renderer_own_data(const sdfhip_renderer_t* r);
int sdfhip_renderer_device(const sdfhip_renderer_t* r);
int sdfhip_renderer_width(const sdfhip_renderer_t* r);
int sdfhip_renderer_height(const sdfhip_renderer_t* r);
uint32_t sdfhip_renderer_num_verts(const sdfhip_renderer_t* r);
uint32_t sdfhip_renderer_num_faces(const sdfhip_renderer_t* r);
const float* sdfhip_renderer_verts(const sdfhip_renderer_t* r);
const uint32_t* sdfhip_renderer_faces(const sdfhip_renderer_t* r);
float* sdfhip_renderer_verts_mutable(sdfhip_renderer_t* r);
uint32_t* sdfhip_renderer_faces_mutable(sdfhip_renderer_t* r);

/* Page-locked host memory for the host-pointer entry points (no counterpart: the reference's
 * Eigen / numpy arrays are pageable).  Points read from, and results written to, such memory move
 * by DMA at the full PCIe rate and without blocking the calling thread; pageable memory works
 * everywhere too.  (Page-locking the caller's own buffers per call, hipHostRegister, was tried and
 * is refused for numpy-owned memory on the target boxes.) */
int sdfhip_host_alloc(size_t bytes, void** out);
int sdfhip_host_free(void* p);

/* Tuning knobs (no counterpart). key: "sort_queries" (0/1, default 1: Morton-sort query
 * chunks of >= sort_min points), "sort_min", "sort_min_contains" (the same threshold for robust contains(),
 * default -1 = automatic: from 1M points on when the column grid's cells hold more than 20 faces on average, never otherwise; below it the ray walk runs on the caller's order), "chunk" (points per internal chunk), "packet" (0: strict kernel only),
 * "approx" (packet pass: approximate first pass + exact second phase; -1 automatic (default), 0 off, 1 on; results are identical either way),
 * "scan_max" (batches with points x padded vertices up to this value are answered by the exact one-wave-per-query
 * scan instead of a tree walk; -1 = automatic (the default: ~5e8 x (vertices / 4000)^0.57), 0 = never), "heavy_budget" (chunks of four vertices a wave of the packet pass may evaluate before its queries go
 * to the one-wave-per-query kernel; 0 = automatic: 1/8 of the tree), "fast_nn" (0/1, applies at the next update: the vertex tree is built on the GPU, no host kd-tree;
 * distances stay exact, an exact fp32 distance tie goes to the lowest (Morton code, index) vertex instead of the
 * reference's first-visited one), "packet_group" (vertices per leaf group of the packet pass, 10..60, 0 = automatic: 40 with the approximate first pass, else 24; applies at the next update), "host_chunk" (points per
 * staging slot of the host entry points, 0 = automatic),
 * "build_threads" (host threads of the kd-tree build; default min(cores, OMP_NUM_THREADS, 16)), "profile" (1: time every query kernel launch
 * with hipEvents on its stream, read back through sdfhip_get_stats; 2: same and reset),
 * "ray_mode" (0 automatic, 1 column grid, 2 cooperative LBVH walk for the z rays), "grid_tri" (0/1: triangle filter of the column
 * grid's cells), "grid_per_face" (list entries per face beyond which the grid coarsens), "x_grid" (0/1, applies at the next
 * update: the x+ tie-break ray reads its own column grid over (y', z') instead of walking the LBVH), "column_sort" (0/1: robust
 * contains() sorts its queries by column of the ray-cast frame instead of 3-D Morton order), "coop_budget" (work units after
 * which a packet's walk is continued by eight waves, 0 = never; default -1 = automatic: 1500 .. 6000 with the size of the tree,
 * less for launches of < 1M points), "coop_max" (largest launch, in points, that
 * uses it; default 2^27 = every launch), "cube_lists" (packet pass: the query domain of a sorted launch is cut into cubes, one
 * wave per cube lists the vertex groups its queries can need and the packet waves test that list instead of walking the box
 * hierarchy from the top; -1 automatic (default: launches of >= "cube_min" points, default 2^23), 0 off, 1 on for every sorted
 * launch), "cube_level" (2^level cubes per axis; 0 = automatic, default: cube edge ~1.3 x the size of a vertex group),
 * "sort_bits" (Morton key bits of the query sort: 24 (default, three radix passes) or 30 (four)),
 * "far_packets" (1, default: in a sorted launch that reaches beyond the near region of the mesh, queries farther than
 * "far_radius_x10" tenths (default 120; "far_radius_sparse_x10", default 20, in a sparse batch) of the half diagonal of the mesh's box from its centre are sorted by distance shell and
 * direction and walked by a kernel of their own; 0: the strict kernel answers them, one tree walk per query).
 * No option changes a result, except "fast_nn" as described. */
int sdfhip_set_option(sdfhip_t* h, const char* key, int64_t value);

const char* sdfhip_last_error(void);
const char* sdfhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SDFHIP_H_ */
