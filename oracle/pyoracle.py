"""ctypes wrapper around oracle/liboracle.so and oracle/_ref/libref_indices.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by sdf_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_indices.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)


def build(force: bool = False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "sdf_oracle.cpp")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src)
    need_ref = os.path.isdir("/root/reference/include/sdf/internal") and (
        not os.path.exists(_REF)
        or os.path.getmtime(_REF) < os.path.getmtime(os.path.join(_HERE, "ref_shim.cpp")))
    if force or stale or need_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def _f(a):
    return a.ctypes.data_as(_fp)


def _load():
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [_fp, C.c_int, _up, C.c_int, C.c_int, _fp]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_update.argtypes = [C.c_void_p, _fp]
    L.orc_set_brute.argtypes = [C.c_void_p, C.c_int]
    L.orc_nn.argtypes = [C.c_void_p, _fp, C.c_int, _ip, C.c_int]
    L.orc_calc.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _fp, C.c_int]
    L.orc_contains.argtypes = [C.c_void_p, _fp, C.c_int, _bp, C.c_int]
    L.orc_raycast_detail.argtypes = [C.c_void_p, _fp, C.c_int, _ip, C.c_int]
    L.orc_ray_parity_candidates.restype = C.c_int
    L.orc_ray_parity_candidates.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _ip, C.c_int]
    L.orc_rotate_point.argtypes = [C.c_void_p, _fp, _fp]
    L.orc_get_face_normals.argtypes = [C.c_void_p, _fp]
    L.orc_get_face_areas.argtypes = [C.c_void_p, _fp]
    L.orc_get_aabb.argtypes = [C.c_void_p, _fp]
    L.orc_get_face_boxes.argtypes = [C.c_void_p, _fp, _fp]
    L.orc_adj_count.restype = C.c_int
    L.orc_adj_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_adj_get.argtypes = [C.c_void_p, C.c_int, _ip]
    L.orc_kd_num_nodes.restype = C.c_int
    L.orc_kd_num_nodes.argtypes = [C.c_void_p]
    L.orc_kd_max_depth.restype = C.c_int
    L.orc_kd_max_depth.argtypes = [C.c_void_p]
    L.orc_kd_export.argtypes = [C.c_void_p, _ip, _ip, _fp]
    L.orc_kd_stats.argtypes = [C.c_void_p, _fp, C.c_int, C.POINTER(C.c_double)]
    L.orc_quat_to_R.argtypes = [_fp, _fp]
    L.orc_sample_surface.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_float, _fp]
    L.orc_dist_point2tri.restype = C.c_float
    L.orc_dist_point2tri.argtypes = [_fp, _fp, _fp, _fp]
    L.orc_dist_point2lineseg.restype = C.c_float
    L.orc_dist_point2lineseg.argtypes = [_fp, _fp, _fp]
    L.orc_bary2d_inside.restype = C.c_int
    L.orc_bary2d_inside.argtypes = [_fp, _fp, _fp, _fp]
    L.orc_rend_create.restype = C.c_void_p
    L.orc_rend_create.argtypes = [_fp, C.c_int, _up, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                  C.c_float, C.c_float]
    L.orc_rend_destroy.argtypes = [C.c_void_p]
    L.orc_rend_update.argtypes = [C.c_void_p]
    L.orc_rend_get_cam.argtypes = [C.c_void_p, _fp]
    L.orc_rend_get_face_boxes.argtypes = [C.c_void_p, _fp, _fp, _bp]
    L.orc_rend_query.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.orc_rend_eval_candidates.argtypes = [C.c_void_p, C.c_float, C.c_float, _ip, C.c_int, _fp, _bp, _ip]
    return L


_L = None


def lib():
    global _L
    if _L is None:
        _L = _load()
    return _L


def default_threads():
    return max(1, min(os.cpu_count() or 1, 32))  # pybind.cpp:12


def _pts(points):
    p = np.ascontiguousarray(points, dtype=np.float32)
    if p.ndim == 1:
        p = p.reshape(1, 3)
    assert p.ndim == 2 and p.shape[1] == 3
    return p


class OracleSDF:
    """CPU restatement with the pysdf.SDF call surface (pybind.cpp:18-83) plus an injectable
    rotation.  Holds references to the numpy arrays (no copy), like Eigen::Ref."""

    def __init__(self, verts, faces, robust=True, R=None):
        self.verts = np.ascontiguousarray(verts, dtype=np.float32)
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32)
        self.robust = bool(robust)
        if R is None:
            R = np.eye(3, dtype=np.float32)
        self.R = np.ascontiguousarray(R, dtype=np.float32).reshape(3, 3)
        self._h = lib().orc_create(_f(self.verts), self.verts.shape[0],
                                   self.faces.ctypes.data_as(_up), self.faces.shape[0],
                                   int(self.robust), _f(self.R))

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def update(self, R=None):
        if R is not None:
            self.R = np.ascontiguousarray(R, dtype=np.float32).reshape(3, 3)
        lib().orc_update(self._h, _f(self.R))

    def set_brute(self, b=True):
        lib().orc_set_brute(self._h, int(b))

    def nn(self, points, n_threads=None):
        p = _pts(points)
        out = np.empty(p.shape[0], dtype=np.int32)
        lib().orc_nn(self._h, _f(p), p.shape[0], out.ctypes.data_as(_ip),
                     n_threads or default_threads())
        return out

    def __call__(self, points, trunc_aabb=False, n_threads=None):
        p = _pts(points)
        out = np.empty(p.shape[0], dtype=np.float32)
        lib().orc_calc(self._h, _f(p), p.shape[0], int(trunc_aabb), _f(out),
                       n_threads or default_threads())
        return out

    calc = __call__

    def contains(self, points, n_threads=None):
        p = _pts(points)
        out = np.empty(p.shape[0], dtype=np.uint8)
        lib().orc_contains(self._h, _f(p), p.shape[0], out.ctypes.data_as(_bp),
                           n_threads or default_threads())
        return out.astype(bool)

    def raycast_detail(self, points, n_threads=None):
        p = _pts(points)
        out = np.empty((p.shape[0], 3), dtype=np.int32)
        lib().orc_raycast_detail(self._h, _f(p), p.shape[0], out.ctypes.data_as(_ip),
                                 n_threads or default_threads())
        return out

    def ray_parity_candidates(self, q_rot, ax, inv, cand):
        q = np.ascontiguousarray(q_rot, dtype=np.float32)
        c = np.ascontiguousarray(cand, dtype=np.int32)
        return lib().orc_ray_parity_candidates(self._h, _f(q), ax, int(inv),
                                               c.ctypes.data_as(_ip), c.shape[0])

    def rotate_point(self, p):
        p = np.ascontiguousarray(p, dtype=np.float32)
        q = np.empty(3, dtype=np.float32)
        lib().orc_rotate_point(self._h, _f(p), _f(q))
        return q

    @property
    def face_normals(self):
        out = np.empty((self.faces.shape[0], 3), dtype=np.float32)
        lib().orc_get_face_normals(self._h, _f(out))
        return out

    @property
    def face_areas(self):
        out = np.empty(self.faces.shape[0], dtype=np.float32)
        lib().orc_get_face_areas(self._h, _f(out))
        return out

    @property
    def aabb(self):
        out = np.empty(6, dtype=np.float32)
        lib().orc_get_aabb(self._h, _f(out))
        return out

    def face_boxes(self):
        mn = np.empty((self.faces.shape[0], 3), dtype=np.float32)
        mx = np.empty((self.faces.shape[0], 3), dtype=np.float32)
        lib().orc_get_face_boxes(self._h, _f(mn), _f(mx))
        return mn, mx

    def adj_faces(self, v):
        n = lib().orc_adj_count(self._h, int(v))
        out = np.empty(n, dtype=np.int32)
        if n:
            lib().orc_adj_get(self._h, int(v), out.ctypes.data_as(_ip))
        return out

    def sample_surface(self, n, seed, total_area):
        out = np.empty((n, 3), dtype=np.float32)
        lib().orc_sample_surface(self._h, n, seed, np.float32(total_area), _f(out))
        return out

    def kd_export(self):
        n = lib().orc_kd_num_nodes(self._h)
        vind = np.empty(self.verts.shape[0], dtype=np.int32)
        ni = np.empty((n, 5), dtype=np.int32)
        nf = np.empty((n, 2), dtype=np.float32)
        lib().orc_kd_export(self._h, vind.ctypes.data_as(_ip), ni.ctypes.data_as(_ip), _f(nf))
        return vind, ni, nf

    def kd_max_depth(self):
        return lib().orc_kd_max_depth(self._h)

    def kd_stats(self, points):
        p = _pts(points)
        out = (C.c_double * 3)()
        lib().orc_kd_stats(self._h, _f(p), p.shape[0], out)
        return tuple(out)


def _pts2(points):
    p = np.ascontiguousarray(points, dtype=np.float32)
    if p.ndim == 1:
        p = p.reshape(1, 2)
    assert p.ndim == 2 and p.shape[1] == 2
    return p


class OracleRenderer:
    """CPU restatement of sdf::Renderer (src/renderer.cpp) with the pysdf.Renderer call surface
    (pybind.cpp:85-161).  Holds references to the numpy arrays (no copy), like Eigen::Ref."""

    def __init__(self, verts, faces, width=1080, height=1080, fx=2600.0, fy=2600.0, cx=540.0, cy=540.0):
        self.verts = np.ascontiguousarray(verts, dtype=np.float32)
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32)
        self.width, self.height = int(width), int(height)
        self._h = lib().orc_rend_create(_f(self.verts), self.verts.shape[0], self.faces.ctypes.data_as(_up),
                                        self.faces.shape[0], self.width, self.height, fx, fy, cx, cy)

    def __del__(self):
        try:
            if self._h:
                lib().orc_rend_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def update(self):
        lib().orc_rend_update(self._h)

    def _q(self, kind, points, fill=False, n_threads=None):
        dt = (np.float32, np.uint8, np.int32)[kind]
        if points is None:
            n, p = self.width * self.height, None
        else:
            pp = _pts2(points)
            n, p = pp.shape[0], _f(pp)
        out = np.empty(n, dtype=dt)
        lib().orc_rend_query(self._h, kind, p, n, int(fill), out.ctypes.data_as(C.c_void_p),
                             n_threads or default_threads())
        if kind == 1:
            out = out.view(np.bool_)
        return out if points is not None else out.reshape(self.height, self.width)

    def __call__(self, points, n_threads=None):
        return self._q(0, points, n_threads=n_threads)

    def contains(self, points, n_threads=None):
        return self._q(1, points, n_threads=n_threads)

    def nn(self, points, fill_outside=True, n_threads=None):
        return self._q(2, points, fill_outside, n_threads)

    def render_depth(self, n_threads=None):
        return self._q(0, None, n_threads=n_threads)

    def render_mask(self, n_threads=None):
        return self._q(1, None, n_threads=n_threads)

    def render_nn(self, fill_outside=False, n_threads=None):
        return self._q(2, None, fill_outside, n_threads)

    @property
    def verts_cam(self):
        out = np.empty((self.verts.shape[0], 2), dtype=np.float32)
        lib().orc_rend_get_cam(self._h, _f(out))
        return out

    def face_boxes(self):
        nf = self.faces.shape[0]
        mn, mx = np.empty((nf, 2), np.float32), np.empty((nf, 2), np.float32)
        valid = np.empty(nf, np.uint8)
        lib().orc_rend_get_face_boxes(self._h, _f(mn), _f(mx), valid.ctypes.data_as(_bp))
        return mn, mx, valid

    def eval_candidates(self, px, py, cand):
        """the reference's handlers (renderer.cpp:63-87) over `cand` in the given order"""
        c = np.ascontiguousarray(cand, dtype=np.int32)
        d = np.zeros(1, np.float32)
        m = np.zeros(1, np.uint8)
        v = np.zeros(1, np.int32)
        lib().orc_rend_eval_candidates(self._h, float(px), float(py), c.ctypes.data_as(_ip), c.shape[0],
                                       _f(d), m.ctypes.data_as(_bp), v.ctypes.data_as(_ip))
        return d[0], bool(m[0]), int(v[0])


def quat_to_R(q4):
    q = np.ascontiguousarray(q4, dtype=np.float32)
    R = np.empty((3, 3), dtype=np.float32)
    lib().orc_quat_to_R(_f(q), _f(R))
    return R


# ----------------------------------------------------------------------------------------
# real vendored indices (oracle/_ref) — available when the .so has been built
# ----------------------------------------------------------------------------------------
_R = None


def ref_available():
    return os.path.exists(_REF)


def ref_lib():
    global _R
    if _R is None:
        R = C.CDLL(_REF)
        R.ref_kd_create.restype = C.c_void_p
        R.ref_kd_create.argtypes = [_fp, C.c_int]
        R.ref_kd_destroy.argtypes = [C.c_void_p]
        R.ref_kd_nn.argtypes = [C.c_void_p, _fp, C.c_int, _ip, _fp]
        R.ref_kd_vind.argtypes = [C.c_void_p, _ip]
        R.ref_kd_export.restype = C.c_int
        R.ref_kd_export.argtypes = [C.c_void_p, _ip, _fp]
        R.ref_rtree_create.restype = C.c_void_p
        R.ref_rtree_create.argtypes = [_fp, _fp, C.c_int]
        R.ref_rtree_destroy.argtypes = [C.c_void_p]
        R.ref_rtree_ray.restype = C.c_int
        R.ref_rtree_ray.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _ip, C.c_int]
        R.ref_kd2_create.restype = C.c_void_p
        R.ref_kd2_create.argtypes = [_fp, C.c_int]
        R.ref_kd2_destroy.argtypes = [C.c_void_p]
        R.ref_kd2_nn.argtypes = [C.c_void_p, _fp, C.c_int, _ip]
        R.ref_rtree2_create.restype = C.c_void_p
        R.ref_rtree2_create.argtypes = [_fp, _fp, _bp, C.c_int]
        R.ref_rtree2_destroy.argtypes = [C.c_void_p]
        R.ref_rtree2_point.restype = C.c_int
        R.ref_rtree2_point.argtypes = [C.c_void_p, _fp, _ip, C.c_int]
        _R = R
    return _R


class RefKd:
    """The reference's real nanoflann index (nn incl. tie-breaks)."""

    def __init__(self, verts):
        self.verts = np.ascontiguousarray(verts, dtype=np.float32)
        self._h = ref_lib().ref_kd_create(_f(self.verts), self.verts.shape[0])

    def __del__(self):
        try:
            ref_lib().ref_kd_destroy(self._h)
        except Exception:
            pass

    def nn(self, points):
        p = _pts(points)
        out = np.empty(p.shape[0], dtype=np.int32)
        d = np.empty(p.shape[0], dtype=np.float32)
        ref_lib().ref_kd_nn(self._h, _f(p), p.shape[0], out.ctypes.data_as(_ip), _f(d))
        return out, d

    def vind(self):
        out = np.empty(self.verts.shape[0], dtype=np.int32)
        ref_lib().ref_kd_vind(self._h, out.ctypes.data_as(_ip))
        return out

    def export(self):
        n = ref_lib().ref_kd_export(self._h, None, None)
        ni = np.empty((n, 3), dtype=np.int32)
        nf = np.empty((n, 2), dtype=np.float32)
        ref_lib().ref_kd_export(self._h, ni.ctypes.data_as(_ip), _f(nf))
        return ni, nf


class RefRTree:
    """The reference's real R-tree over per-face boxes (candidate sets per ray)."""

    def __init__(self, bmin, bmax):
        self.bmin = np.ascontiguousarray(bmin, dtype=np.float32)
        self.bmax = np.ascontiguousarray(bmax, dtype=np.float32)
        self._h = ref_lib().ref_rtree_create(_f(self.bmin), _f(self.bmax), self.bmin.shape[0])
        self._buf = np.empty(1 << 16, dtype=np.int32)

    def __del__(self):
        try:
            ref_lib().ref_rtree_destroy(self._h)
        except Exception:
            pass

    def ray(self, q_rot, ax, inv):
        q = np.ascontiguousarray(q_rot, dtype=np.float32)
        n = ref_lib().ref_rtree_ray(self._h, _f(q), ax, int(inv),
                                    self._buf.ctypes.data_as(_ip), self._buf.shape[0])
        assert n <= self._buf.shape[0]
        return self._buf[:n].copy()


class RefKd2:
    """The reference's real nanoflann index with DIM = 2 (renderer.cpp:248-250)."""

    def __init__(self, pts2):
        self.pts = np.ascontiguousarray(pts2, dtype=np.float32)
        self._h = ref_lib().ref_kd2_create(_f(self.pts), self.pts.shape[0])

    def __del__(self):
        try:
            ref_lib().ref_kd2_destroy(self._h)
        except Exception:
            pass

    def nn(self, points2):
        p = _pts2(points2)
        out = np.empty(p.shape[0], dtype=np.int32)
        ref_lib().ref_kd2_nn(self._h, _f(p), p.shape[0], out.ctypes.data_as(_ip))
        return out


class RefRTree2:
    """The reference's real RTree<int, float, 2> over per-face image-space boxes (renderer.cpp:42-59)."""

    def __init__(self, bmin, bmax, valid):
        self.bmin = np.ascontiguousarray(bmin, dtype=np.float32)
        self.bmax = np.ascontiguousarray(bmax, dtype=np.float32)
        self.valid = np.ascontiguousarray(valid, dtype=np.uint8)
        self._h = ref_lib().ref_rtree2_create(_f(self.bmin), _f(self.bmax), self.valid.ctypes.data_as(_bp),
                                              self.bmin.shape[0])
        self._buf = np.empty(1 << 16, dtype=np.int32)

    def __del__(self):
        try:
            ref_lib().ref_rtree2_destroy(self._h)
        except Exception:
            pass

    def point(self, q2):
        """candidate face ids of the degenerate box [q, q], in the R-tree's search order"""
        q = np.ascontiguousarray(q2, dtype=np.float32)
        n = ref_lib().ref_rtree2_point(self._h, _f(q), self._buf.ctypes.data_as(_ip), self._buf.shape[0])
        assert n <= self._buf.shape[0]
        return self._buf[:n].copy()
