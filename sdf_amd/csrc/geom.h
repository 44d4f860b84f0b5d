// geom.h — fp32 geometry of the point-query path, device side.
//
// Every expression keeps the reference's association so that results are bit-identical to
// an x86-64 SSE2 build without FMA contraction (CMakeLists.txt:33); this translation unit
// MUST be compiled with -ffp-contract=off.  IEEE division and sqrt are the correctly
// rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt, denormals kept).
// 3-term Eigen reductions are c0 + (c1 + c2).
#pragma once
#include <hip/hip_runtime.h>

namespace sdfk {

struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float sum3(float c0, float c1, float c2) { return c0 + (c1 + c2); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float sqnorm(V3 a) { return dot(a, a); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// row vector * row-major 3x3 (sdf.cpp:54-64, 284-285, 299-309)
__device__ __forceinline__ V3 mulR(V3 v, const float* R) {
    return {sum3(v.x * R[0], v.y * R[3], v.z * R[6]), sum3(v.x * R[1], v.y * R[4], v.z * R[7]),
            sum3(v.x * R[2], v.y * R[5], v.z * R[8])};
}
// std::min / std::max semantics (Eigen cwiseMin/cwiseMax, std::max/min in sdf.hpp:71)
__device__ __forceinline__ float stdmin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float stdmax(float a, float b) { return (a < b) ? b : a; }

// sdf.hpp:64-73 (squared distance)
__device__ __forceinline__ float dist_point2lineseg(V3 p, V3 a, V3 b) {
    V3 ap = sub(p, a), ab = sub(b, a);
    float t = dot(ap, ab) / sqnorm(ab);
    t = stdmax(0.0f, stdmin(1.0f, t));
    V3 r = {ap.x - t * ab.x, ap.y - t * ab.y, ap.z - t * ab.z};
    return sqnorm(r);
}

// sdf.hpp:88-127: bary + first-negative-wins cascade (squared distance)
__device__ __forceinline__ float dist_point2tri(V3 p, V3 a, V3 b, V3 c, V3 n, float area) {
    float area_pbc = dot(n, cross(sub(b, p), sub(c, p)));
    float area_pca = dot(n, cross(sub(c, p), sub(a, p)));
    float u = area_pbc / area;
    float v = area_pca / area;
    float w = (1.0f - u) - v;
    // the three edge cases share one segment routine: pick the end points, call once
    V3 e0, e1;
    bool edge = true;
    if (u < 0) {
        e0 = b, e1 = c;
    } else if (v < 0) {
        e0 = a, e1 = c;
    } else if (w < 0) {
        e0 = a, e1 = b;
    } else {
        edge = false;
        e0 = a, e1 = b;
    }
    if (edge) return dist_point2lineseg(p, e0, e1);
    V3 r = {((u * a.x + v * b.x) + w * c.x) - p.x, ((u * a.y + v * b.y) + w * c.y) - p.y,
            ((u * a.z + v * b.z) + w * c.z) - p.z};
    return sqnorm(r);
}

// sdf.hpp:130-142 + the >= 0 test of sdf.cpp:310
__device__ __forceinline__ bool bary2d_inside(float px, float py, float ax, float ay, float bx,
                                              float by, float cx, float cy) {
    float v0x = bx - ax, v0y = by - ay, v1x = cx - ax, v1y = cy - ay, v2x = px - ax, v2y = py - ay;
    float invden = 1.f / (v0x * v1y - v1x * v0y);
    float b1 = (v2x * v1y - v1x * v2y) * invden;
    float b2 = (v0x * v2y - v2x * v0y) * invden;
    float b0 = 1.0f - (b1 + b2);
    return b0 >= 0.f && b1 >= 0.f && b2 >= 0.f;
}

}  // namespace sdfk
