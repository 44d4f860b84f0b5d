// sdf_amd/sdf.hpp — C++ façade with the reference's `sdf::SDF` and `sdf::Renderer` surfaces
// (include/sdf/sdf.hpp:150-258, 282-369 of sxyu/sdf) over the C ABI of sdfhip.h.
//
// Two spellings of every method:
//   * plain-pointer / std::vector signatures: always available;
//   * the reference's Eigen signatures (Eigen::Ref<const Points> ...), compiled only when
//     <Eigen/Core> is on the include path.  Eigen is absent from the build image, so that
//     half is not compile-tested there; it is a thin wrapper over the pointer half.
//
// Constructor, update() and query signatures, defaults (`robust = true`, `copy = false`),
// the public const members `robust` / `own_data`, and the stderr messages of the mutable
// accessors follow the reference.  `n_threads` is accepted and ignored (GPU path).
// Errors that the reference leaves undefined (empty mesh, no device, ...) throw
// std::runtime_error with the C ABI's message.
//
// Header-only: link with -lsdfhip (sdf_amd/libsdfhip.so).
#pragma once
#ifndef SDF_AMD_SDF_HPP_
#define SDF_AMD_SDF_HPP_

#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <chrono>
#include <random>
#include <vector>

#include "../sdfhip.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define SDF_AMD_HAVE_EIGEN 1
#endif
#endif

namespace sdf {

using Index = uint32_t;  // sdf.hpp:43

#ifdef SDF_AMD_HAVE_EIGEN
using Points = Eigen::Matrix<float, Eigen::Dynamic, 3, Eigen::RowMajor>;       // sdf.hpp:44
using Triangles = Eigen::Matrix<Index, Eigen::Dynamic, 3, Eigen::RowMajor>;    // sdf.hpp:46
using Vector = Eigen::Matrix<float, Eigen::Dynamic, 1>;                         // sdf.hpp:49
#endif

struct SDF {
    // ---- plain-pointer construction: verts [n_verts,3] float32, faces [n_faces,3] uint32 ----
    SDF(const float* verts, uint32_t n_verts, const Index* faces, uint32_t n_faces, bool robust = true,
        bool copy = false, int device = -1)
        : robust(robust), own_data(copy) {
        check(sdfhip_create(verts, n_verts, faces, n_faces, robust ? 1 : 0, copy ? 1 : 0, device, &h_));
    }
#ifdef SDF_AMD_HAVE_EIGEN
    // sdf.hpp:168-169.  copy = false keeps referencing the caller's matrices (sdf.cpp:339).
    SDF(Eigen::Ref<const Points> verts, Eigen::Ref<const Triangles> faces, bool robust = true,
        bool copy = false)
        : SDF(verts.data(), (uint32_t)verts.rows(), faces.data(), (uint32_t)faces.rows(), robust, copy) {}
#endif
    // several GPUs of this process (no counterpart: the reference is one CPU process): the host-array
    // queries are cut into contiguous ranges, one per listed device (sdfhip_group_*)
    SDF(const float* verts, uint32_t n_verts, const Index* faces, uint32_t n_faces, bool robust, bool copy,
        const std::vector<int>& devices)
        : robust(robust), own_data(copy) {
        check(sdfhip_group_create(verts, n_verts, faces, n_faces, robust ? 1 : 0, copy ? 1 : 0, devices.data(),
                                  (int)devices.size(), &g_));
        h_ = sdfhip_group_member(g_, 0);  // accessors (verts_mutable, aabb, ...) go to member 0
    }
    ~SDF() {
        if (g_)
            sdfhip_group_destroy(g_);
        else
            sdfhip_destroy(h_);
    }
    static uint64_t fresh_seed() {
        std::random_device rd;
        return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^
               (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    }
    SDF(const SDF&) = delete;  // as the reference: const members + pImpl
    SDF& operator=(const SDF&) = delete;

    /*** PRIMARY INTERFACE (sdf.hpp:172-208) ***/
    std::vector<float> operator()(const float* points, int64_t n, bool trunc_aabb = false,
                                  int /*n_threads*/ = 0) const {
        std::vector<float> out((size_t)n);
        check(g_ ? sdfhip_group_sdf(g_, points, n, trunc_aabb ? 1 : 0, out.data())
                 : sdfhip_sdf(h_, points, n, trunc_aabb ? 1 : 0, out.data()));
        return out;
    }
    std::vector<int> nn(const float* points, int64_t n, int /*n_threads*/ = 0) const {
        std::vector<int> out((size_t)n);
        check(g_ ? sdfhip_group_nn(g_, points, n, out.data()) : sdfhip_nn(h_, points, n, out.data()));
        return out;
    }
    std::vector<uint8_t> contains(const float* points, int64_t n, int /*n_threads*/ = 0) const {
        std::vector<uint8_t> out((size_t)n);
        check(g_ ? sdfhip_group_contains(g_, points, n, out.data()) : sdfhip_contains(h_, points, n, out.data()));
        return out;
    }
    // device-resident points / results on `stream` (hipStream_t), asynchronous
    void sdf_device(const float* d_points, int64_t n, bool trunc_aabb, float* d_out, void* stream) const {
        check(sdfhip_sdf_device(h_, d_points, n, trunc_aabb ? 1 : 0, d_out, stream));
    }
    void contains_device(const float* d_points, int64_t n, uint8_t* d_out, void* stream) const {
        check(sdfhip_contains_device(h_, d_points, n, d_out, stream));
    }
    void nn_device(const float* d_points, int64_t n, int32_t* d_out, void* stream) const {
        check(sdfhip_nn_device(h_, d_points, n, d_out, stream));
    }
#ifdef SDF_AMD_HAVE_EIGEN
    Vector operator()(Eigen::Ref<const Points> points, bool trunc_aabb = false,
                      int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Vector out(points.rows());
        check(sdfhip_sdf(h_, points.data(), points.rows(), trunc_aabb ? 1 : 0, out.data()));
        return out;
    }
    Eigen::VectorXi nn(Eigen::Ref<const Points> points,
                       int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::VectorXi out(points.rows());
        check(sdfhip_nn(h_, points.data(), points.rows(), out.data()));
        return out;
    }
    Eigen::Matrix<bool, Eigen::Dynamic, 1> contains(
        Eigen::Ref<const Points> points, int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::Matrix<bool, Eigen::Dynamic, 1> out(points.rows());
        static_assert(sizeof(bool) == 1, "bool must be one byte");
        check(sdfhip_contains(h_, points.data(), points.rows(), reinterpret_cast<uint8_t*>(out.data())));
        return out;
    }
#endif
    // sdf.hpp:206-208 / sdf.cpp:391
    void update() { check(g_ ? sdfhip_group_update(g_) : sdfhip_update(h_)); }

    // sdf.hpp:210-213 / sdf.cpp:176-234.  `seed` (extension) makes the samples reproducible;
    // by default a fresh one is drawn, like the reference's time-seeded generator.
    std::vector<float> sample_surface_vec(int num_points, uint64_t seed = fresh_seed()) const {
        if (num_faces() == 0) {
            std::cerr << "ERROR: No faces, can't sample surface.\n";  // sdf.cpp:178
            return {};
        }
        std::vector<float> out(3 * (size_t)(num_points > 0 ? num_points : 0));
        check(sdfhip_sample_surface(h_, num_points, seed, out.data()));
        return out;
    }
#ifdef SDF_AMD_HAVE_EIGEN
    Points sample_surface(int num_points) const {
        if (num_faces() == 0) {
            std::cerr << "ERROR: No faces, can't sample surface.\n";
            return Points();
        }
        Points out(num_points, 3);
        check(sdfhip_sample_surface(h_, num_points, fresh_seed(), out.data()));
        return out;
    }
#endif

    // the reference's rotation is private and random (sdf.cpp:43-48); parity needs it injectable
    void set_raycast_axes(const float R[9], bool pin = true) { check(sdfhip_set_raycast_axes(h_, R, pin)); }
    void get_raycast_axes(float R[9]) const { check(sdfhip_get_raycast_axes(h_, R)); }

    /*** DATA ACCESSORS (sdf.hpp:215-245) ***/
    std::vector<int> adj_faces(int pointid) const {
        const int32_t* ids = nullptr;
        const int n = sdfhip_adj_faces(h_, (uint32_t)pointid, &ids);
        if (n < 0) check(n);
        return std::vector<int>(ids, ids + n);
    }
    float surface_area() const {
        float a = 0.f;
        check(sdfhip_surface_area(h_, &a));
        return a;
    }
    std::vector<float> face_areas_vec() const {
        std::vector<float> out(sdfhip_num_faces(h_));
        check(sdfhip_face_areas(h_, out.data()));
        return out;
    }
    std::vector<float> face_normals_vec() const {
        std::vector<float> out(3 * (size_t)sdfhip_num_faces(h_));
        check(sdfhip_face_normals(h_, out.data()));
        return out;
    }
    void aabb(float out[6]) const { check(sdfhip_aabb(h_, out)); }
    uint32_t num_verts() const { return sdfhip_num_verts(h_); }
    uint32_t num_faces() const { return sdfhip_num_faces(h_); }
    const float* verts_data() const { return sdfhip_verts(h_); }
    const Index* faces_data() const { return sdfhip_faces(h_); }
    float* verts_mutable_data() {  // sdf.cpp:369-375
        if (!own_data) std::cerr << "ERROR: 'verts' is non mutable, construct with copy=True\n";
        return sdfhip_verts_mutable(h_);
    }
    Index* faces_mutable_data() {  // sdf.cpp:360-366
        if (!own_data) std::cerr << "ERROR: 'faces' is non mutable, construct with copy=True\n";
        return sdfhip_faces_mutable(h_);
    }
#ifdef SDF_AMD_HAVE_EIGEN
    Vector face_areas() const {
        Vector out(num_faces());
        check(sdfhip_face_areas(h_, out.data()));
        return out;
    }
    Points face_normals() const {
        Points out(num_faces(), 3);
        check(sdfhip_face_normals(h_, out.data()));
        return out;
    }
    Eigen::Matrix<float, 6, 1> aabb() const {
        Eigen::Matrix<float, 6, 1> out;
        check(sdfhip_aabb(h_, out.data()));
        return out;
    }
    Eigen::Map<const Triangles> faces() const { return {faces_data(), (Eigen::Index)num_faces(), 3}; }
    Eigen::Map<const Points> verts() const { return {verts_data(), (Eigen::Index)num_verts(), 3}; }
    Eigen::Map<Points> verts_mutable() {
        float* p = verts_mutable_data();
        return {p, p ? (Eigen::Index)num_verts() : 0, 3};
    }
    Eigen::Map<Triangles> faces_mutable() {
        Index* p = faces_mutable_data();
        return {p, p ? (Eigen::Index)num_faces() : 0, 3};
    }
#endif

    const bool robust;    // sdf.hpp:242
    const bool own_data;  // sdf.hpp:245

    sdfhip_t* handle() const { return h_; }

   private:
    static void check(int rc) {
        if (rc != 0) throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error());
    }
    sdfhip_t* h_ = nullptr;
    sdfhip_group_t* g_ = nullptr;  // set when the object spans several GPUs (then h_ = member 0)
};

// sdf::Renderer (sdf.hpp:282-369 of the reference, src/renderer.cpp): image-space raycaster.
// Images are row-major [height, width]; image points are row-major [n, 2] (x, y).
struct Renderer {
    Renderer(const float* verts, uint32_t n_verts, const Index* faces, uint32_t n_faces, int width = 1080,
             int height = 1080, float fx = 2600.f, float fy = 2600.f, float cx = 540.f, float cy = 540.f,
             bool copy = false, int device = -1)
        : own_data(copy) {
        check(sdfhip_renderer_create(verts, n_verts, faces, n_faces, width, height, fx, fy, cx, cy,
                                     copy ? 1 : 0, device, &h_));
    }
#ifdef SDF_AMD_HAVE_EIGEN
    using Points2D = Eigen::Matrix<float, Eigen::Dynamic, 2, Eigen::RowMajor>;  // sdf.hpp:45
    // sdf.hpp:298-301
    Renderer(Eigen::Ref<const Points> verts, Eigen::Ref<const Triangles> faces, int width = 1080,
             int height = 1080, float fx = 2600.f, float fy = 2600.f, float cx = 540.f, float cy = 540.f,
             bool copy = false)
        : Renderer(verts.data(), (uint32_t)verts.rows(), faces.data(), (uint32_t)faces.rows(), width, height,
                   fx, fy, cx, cy, copy) {}
#endif
    ~Renderer() { sdfhip_renderer_destroy(h_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    /*** PRIMARY INTERFACE (sdf.hpp:306-344) ***/
    std::vector<float> render_depth_vec(int /*n_threads*/ = 0) const {
        std::vector<float> out((size_t)width() * height());
        check(sdfhip_renderer_render_depth(h_, out.data()));
        return out;
    }
    std::vector<uint8_t> render_mask_vec(int /*n_threads*/ = 0) const {
        std::vector<uint8_t> out((size_t)width() * height());
        check(sdfhip_renderer_render_mask(h_, out.data()));
        return out;
    }
    std::vector<int> render_nn_vec(bool fill_outside = false, int /*n_threads*/ = 0) const {
        std::vector<int> out((size_t)width() * height());
        check(sdfhip_renderer_render_nn(h_, fill_outside ? 1 : 0, out.data()));
        return out;
    }
    std::vector<float> operator()(const float* points2, int64_t n, int /*n_threads*/ = 0) const {
        std::vector<float> out((size_t)n);
        check(sdfhip_renderer_depth(h_, points2, n, out.data()));
        return out;
    }
    std::vector<uint8_t> contains(const float* points2, int64_t n, int /*n_threads*/ = 0) const {
        std::vector<uint8_t> out((size_t)n);
        check(sdfhip_renderer_contains(h_, points2, n, out.data()));
        return out;
    }
    std::vector<int> nn(const float* points2, int64_t n, bool fill_outside = false, int /*n_threads*/ = 0) const {
        std::vector<int> out((size_t)n);
        check(sdfhip_renderer_nn(h_, points2, n, fill_outside ? 1 : 0, out.data()));
        return out;
    }
    // device pointers on `stream` (hipStream_t); kind 0 depth, 1 mask, 2 vertex id; d_points2 may be
    // null (image pixels)
    void query_device(int kind, const float* d_points2, int64_t n, bool fill_outside, void* d_out,
                      void* stream) const {
        check(sdfhip_renderer_query_device(h_, kind, d_points2, n, fill_outside ? 1 : 0, d_out, stream));
    }
#ifdef SDF_AMD_HAVE_EIGEN
    Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> render_depth(
        int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> out(height(), width());
        check(sdfhip_renderer_render_depth(h_, out.data()));
        return out;
    }
    Eigen::Matrix<bool, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> render_mask(
        int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::Matrix<bool, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> out(height(), width());
        check(sdfhip_renderer_render_mask(h_, reinterpret_cast<uint8_t*>(out.data())));
        return out;
    }
    Eigen::Matrix<int, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> render_nn(
        bool fill_outside = false, int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::Matrix<int, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> out(height(), width());
        check(sdfhip_renderer_render_nn(h_, fill_outside ? 1 : 0, out.data()));
        return out;
    }
    Vector operator()(Eigen::Ref<const Points2D> points,
                      int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Vector out(points.rows());
        check(sdfhip_renderer_depth(h_, points.data(), points.rows(), out.data()));
        return out;
    }
    Eigen::Matrix<bool, Eigen::Dynamic, 1> contains(
        Eigen::Ref<const Points2D> points, int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::Matrix<bool, Eigen::Dynamic, 1> out(points.rows());
        check(sdfhip_renderer_contains(h_, points.data(), points.rows(), reinterpret_cast<uint8_t*>(out.data())));
        return out;
    }
    Eigen::VectorXi nn(Eigen::Ref<const Points2D> points, bool fill_outside = false,
                       int /*n_threads*/ = std::thread::hardware_concurrency()) const {
        Eigen::VectorXi out(points.rows());
        check(sdfhip_renderer_nn(h_, points.data(), points.rows(), fill_outside ? 1 : 0, out.data()));
        return out;
    }
#endif
    void update() { check(sdfhip_renderer_update(h_)); }  // sdf.hpp:344

    /*** DATA ACCESSORS (sdf.hpp:346-356) ***/
    int width() const { return sdfhip_renderer_width(h_); }
    int height() const { return sdfhip_renderer_height(h_); }
    uint32_t num_verts() const { return sdfhip_renderer_num_verts(h_); }
    uint32_t num_faces() const { return sdfhip_renderer_num_faces(h_); }
    const float* verts_data() const { return sdfhip_renderer_verts(h_); }
    const Index* faces_data() const { return sdfhip_renderer_faces(h_); }
    float* verts_mutable_data() {  // renderer.cpp:298-304
        if (!own_data) std::cerr << "ERROR: 'verts' is non mutable, construct with copy=True\n";
        return sdfhip_renderer_verts_mutable(h_);
    }
    Index* faces_mutable_data() {  // renderer.cpp:289-295
        if (!own_data) std::cerr << "ERROR: 'faces' is non mutable, construct with copy=True\n";
        return sdfhip_renderer_faces_mutable(h_);
    }
#ifdef SDF_AMD_HAVE_EIGEN
    Eigen::Map<const Triangles> faces() const { return {faces_data(), (Eigen::Index)num_faces(), 3}; }
    Eigen::Map<const Points> verts() const { return {verts_data(), (Eigen::Index)num_verts(), 3}; }
    Eigen::Map<Points> verts_mutable() {
        float* p = verts_mutable_data();
        return {p, p ? (Eigen::Index)num_verts() : 0, 3};
    }
    Eigen::Map<Triangles> faces_mutable() {
        Index* p = faces_mutable_data();
        return {p, p ? (Eigen::Index)num_faces() : 0, 3};
    }
#endif

    const bool own_data;  // sdf.hpp:356

    sdfhip_renderer_t* handle() const { return h_; }

   private:
    static void check(int rc) {
        if (rc != 0) throw std::runtime_error(std::string("sdfhip: ") + sdfhip_last_error());
    }
    sdfhip_renderer_t* h_ = nullptr;
};

}  // namespace sdf

#endif  // SDF_AMD_SDF_HPP_
