"""`pysdf.util` (pybind.cpp:163-194): the scalar geometry helpers of sdf::util (sdf.hpp:51-143).

These are single-primitive CPU helpers in the reference as well (they are not part of the batched
query path; that path lives in libsdfhip.so).  Every operation is done in float32 in the
reference's order; 3-term reductions follow Eigen 3.3's c0 + (c1 + c2) like the rest of this
repository."""
import numpy as np

_f = np.float32


def _v3(x):
    v = np.asarray(x, dtype=np.float32).reshape(-1)
    if v.shape[0] != 3:
        raise TypeError("expected a 3-vector")
    return v


def _v2(x):
    v = np.asarray(x, dtype=np.float32).reshape(-1)
    if v.shape[0] != 2:
        raise TypeError("expected a 2-vector")
    return v


def _dot(a, b):
    return _f(a[0] * b[0]) + (_f(a[1] * b[1]) + _f(a[2] * b[2]))


def _cross(a, b):
    return np.array([_f(a[1] * b[2]) - _f(a[2] * b[1]), _f(a[2] * b[0]) - _f(a[0] * b[2]),
                     _f(a[0] * b[1]) - _f(a[1] * b[0])], dtype=np.float32)


def normal(a, b, c):
    """Triangle normal (each row is a point): (b - a) x (c - a), not normalised (sdf.hpp:75-81)"""
    a, b, c = _v3(a), _v3(b), _v3(c)
    return _cross(b - a, c - a)


def _unit_normal(a, b, c):
    n = normal(a, b, c)
    with np.errstate(all="ignore"):
        area = np.sqrt(_dot(n, n))
        return (n / area).astype(np.float32), _f(area)


def dist_point2line(p, a, b):
    """Compute 3d point-line squared distance (sdf.hpp:53-61)"""
    p, a, b = _v3(p), _v3(a), _v3(b)
    ap, ab = p - a, b - a
    with np.errstate(all="ignore"):
        r = ap - _f(_dot(ap, ab) / _dot(ab, ab)) * ab
    return float(_dot(r, r))


def dist_point2lineseg(p, a, b):
    """Compute 3d point-line segment squared distance (sdf.hpp:64-73)"""
    p, a, b = _v3(p), _v3(a), _v3(b)
    ap, ab = p - a, b - a
    with np.errstate(all="ignore"):
        t = _f(_dot(ap, ab) / _dot(ab, ab))
    t1 = t if t < _f(1.0) else _f(1.0)   # std::min(1, t)
    t = t1 if _f(0.0) < t1 else _f(0.0)  # std::max(0, .)
    r = ap - t * ab
    return float(_dot(r, r))


def _bary(p, a, b, c, n, area):
    with np.errstate(all="ignore"):
        u = _f(_dot(n, _cross(b - p, c - p)) / area)
        v = _f(_dot(n, _cross(c - p, a - p)) / area)
    return np.array([u, v, (_f(1.0) - u) - v], dtype=np.float32)


def bary(p, a, b, c):
    """3D point to barycentric (pybind.cpp:168-176; sdf.hpp:88-104)"""
    p, a, b, c = _v3(p), _v3(a), _v3(b), _v3(c)
    n, area = _unit_normal(a, b, c)
    return _bary(p, a, b, c, n, area)


def dist_point2tri(p, a, b, c):
    """Compute 3d point-triangle squared distance (pybind.cpp:183-192; sdf.hpp:110-127)"""
    p, a, b, c = _v3(p), _v3(a), _v3(b), _v3(c)
    n, area = _unit_normal(a, b, c)
    uvw = _bary(p, a, b, c, n, area)
    if uvw[0] < 0:
        return dist_point2lineseg(p, b, c)
    if uvw[1] < 0:
        return dist_point2lineseg(p, a, c)
    if uvw[2] < 0:
        return dist_point2lineseg(p, a, b)
    r = ((uvw[0] * a + uvw[1] * b) + uvw[2] * c) - p
    return float(_dot(r, r))


def bary2d(p, a, b, c):
    """Test if point is in triangle (2d): barycentric coordinates of p (sdf.hpp:130-142)"""
    p, a, b, c = _v2(p), _v2(a), _v2(b), _v2(c)
    v0, v1, v2 = b - a, c - a, p - a
    with np.errstate(all="ignore"):
        invden = _f(1.0) / (_f(v0[0] * v1[1]) - _f(v1[0] * v0[1]))
        b1 = (_f(v2[0] * v1[1]) - _f(v1[0] * v2[1])) * invden
        b2 = (_f(v0[0] * v2[1]) - _f(v2[0] * v0[1])) * invden
    return np.array([_f(1.0) - (b1 + b2), b1, b2], dtype=np.float32)
