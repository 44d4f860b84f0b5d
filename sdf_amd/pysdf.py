"""`pysdf`-compatible Python surface over the C ABI (include/sdfhip.h).

Mirrors the reference's pybind module (pybind.cpp:14-83, 195): class `SDF` with the same
constructor, `__call__`/`calc`, `contains`, `nn`, `update`, `sample_surface`, properties and
`__repr__`, and the module attribute `num_threads`.  numpy inputs are converted like
pybind11/eigen.h does (float64 -> float32, int64 -> uint32, a length-3 vector is a 1x3
point; README.md:24-25, 103-106) and numpy results are returned.  torch tensors that live on
the handle's GPU are passed to the `_device` entry points without a host round trip and a
tensor on the same device is returned.

There is no CPU fallback: importing works anywhere, constructing an SDF needs the built
`libsdfhip.so` and a HIP device and raises otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("SDFHIP_LIB") or os.path.join(_HERE, "libsdfhip.so")  # SDFHIP_LIB: A/B builds while tuning

# pybind.cpp:12 / 195
num_threads = min(os.cpu_count() or 1, 32)

_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


class SdfHipError(RuntimeError):
    pass


class _Stats(C.Structure):
    _fields_ = [("kd_nodes", C.c_int32), ("kd_depth", C.c_int32), ("bvh_nodes", C.c_int32),
                ("bvh_depth", C.c_int32), ("last_update_ms", C.c_float),
                ("last_kd_build_ms", C.c_float), ("query_kernel_ms", C.c_double),
                ("query_kernel_launches", C.c_int64), ("query_kernel_points", C.c_int64),
                ("last_fallback_points", C.c_int64), ("last_heavy_points", C.c_int64),
                ("last_kernel", C.c_int32), ("kd_builds", C.c_int32)]


KERNEL_NAMES = {0: "none", 1: "k_query (strict)", 2: "k_query_packet<exact walk>", 3: "k_query_packet<approx walk>",
                4: "k_query_heavy (scan)"}


_lib = None

# every symbol include/sdfhip.h declares (tests check that the library exports them all)
ABI_SYMBOLS = [
    "sdfhip_create", "sdfhip_destroy", "sdfhip_update", "sdfhip_update_from_device",
    "sdfhip_set_raycast_axes", "sdfhip_get_raycast_axes", "sdfhip_sdf", "sdfhip_sdf_device", "sdfhip_sdf_device_noalloc",
    "sdfhip_contains", "sdfhip_contains_device", "sdfhip_nn", "sdfhip_nn_device",
    "sdfhip_raycast_detail", "sdfhip_robust", "sdfhip_own_data", "sdfhip_device",
    "sdfhip_num_verts", "sdfhip_num_faces", "sdfhip_verts", "sdfhip_faces",
    "sdfhip_verts_mutable", "sdfhip_faces_mutable", "sdfhip_surface_area", "sdfhip_face_areas",
    "sdfhip_face_normals", "sdfhip_aabb", "sdfhip_adj_faces", "sdfhip_get_stats",
    "sdfhip_set_option", "sdfhip_last_error", "sdfhip_version",
    "sdfhip_sample_surface", "sdfhip_sample_surface_device",
    "sdfhip_host_alloc", "sdfhip_host_free",
    "sdfhip_group_create", "sdfhip_group_destroy", "sdfhip_group_size", "sdfhip_group_member",
    "sdfhip_group_update", "sdfhip_group_set_raycast_axes", "sdfhip_group_get_raycast_axes",
    "sdfhip_group_sdf", "sdfhip_group_contains", "sdfhip_group_nn",
    "sdfhip_group_sdf_device", "sdfhip_group_contains_device", "sdfhip_group_nn_device",
    "sdfhip_group_update_from_device",
    "sdfhip_renderer_create", "sdfhip_renderer_destroy", "sdfhip_renderer_update",
    "sdfhip_renderer_render_depth", "sdfhip_renderer_render_mask", "sdfhip_renderer_render_nn",
    "sdfhip_renderer_depth", "sdfhip_renderer_contains", "sdfhip_renderer_nn",
    "sdfhip_renderer_query_device", "sdfhip_renderer_own_data", "sdfhip_renderer_device",
    "sdfhip_renderer_width", "sdfhip_renderer_height", "sdfhip_renderer_num_verts",
    "sdfhip_renderer_num_faces", "sdfhip_renderer_verts", "sdfhip_renderer_faces",
    "sdfhip_renderer_verts_mutable", "sdfhip_renderer_faces_mutable",
]


def load_library():
    """Load libsdfhip.so (built in-tree by `__graft_entry__.build()` / sdf_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise SdfHipError(
            f"{_LIBPATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950).  There is no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as
    # /opt/rocm's).  Loading torch first makes this library bind to that copy; the other
    # order would put two runtimes in the process and the second finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIBPATH)
    vp = C.c_void_p
    L.sdfhip_create.restype = C.c_int
    L.sdfhip_create.argtypes = [_fp, C.c_uint32, _up, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                C.POINTER(vp)]
    L.sdfhip_destroy.restype = None
    L.sdfhip_destroy.argtypes = [vp]
    L.sdfhip_update.argtypes = [vp]
    L.sdfhip_update_from_device.argtypes = [vp, vp, vp]
    L.sdfhip_set_raycast_axes.argtypes = [vp, _fp, C.c_int]
    L.sdfhip_get_raycast_axes.argtypes = [vp, _fp]
    L.sdfhip_sdf.argtypes = [vp, _fp, C.c_int64, C.c_int, _fp]
    L.sdfhip_sdf_device.argtypes = [vp, vp, C.c_int64, C.c_int, vp, vp]
    L.sdfhip_sdf_device_noalloc.argtypes = [vp, vp, C.c_int64, C.c_int, vp, vp]
    L.sdfhip_contains.argtypes = [vp, _fp, C.c_int64, _bp]
    L.sdfhip_contains_device.argtypes = [vp, vp, C.c_int64, vp, vp]
    L.sdfhip_nn.argtypes = [vp, _fp, C.c_int64, _ip]
    L.sdfhip_nn_device.argtypes = [vp, vp, C.c_int64, vp, vp]
    L.sdfhip_raycast_detail.argtypes = [vp, _fp, C.c_int64, _ip]
    for name in ("sdfhip_robust", "sdfhip_own_data", "sdfhip_device"):
        getattr(L, name).argtypes = [vp]
    L.sdfhip_num_verts.restype = C.c_uint32
    L.sdfhip_num_verts.argtypes = [vp]
    L.sdfhip_num_faces.restype = C.c_uint32
    L.sdfhip_num_faces.argtypes = [vp]
    L.sdfhip_verts.restype = _fp
    L.sdfhip_verts.argtypes = [vp]
    L.sdfhip_faces.restype = _up
    L.sdfhip_faces.argtypes = [vp]
    L.sdfhip_verts_mutable.restype = _fp
    L.sdfhip_verts_mutable.argtypes = [vp]
    L.sdfhip_faces_mutable.restype = _up
    L.sdfhip_faces_mutable.argtypes = [vp]
    L.sdfhip_surface_area.argtypes = [vp, _fp]
    L.sdfhip_face_areas.argtypes = [vp, _fp]
    L.sdfhip_face_normals.argtypes = [vp, _fp]
    L.sdfhip_aabb.argtypes = [vp, _fp]
    L.sdfhip_adj_faces.argtypes = [vp, C.c_uint32, C.POINTER(_ip)]
    L.sdfhip_sample_surface.argtypes = [vp, C.c_int64, C.c_uint64, _fp]
    L.sdfhip_sample_surface_device.argtypes = [vp, C.c_int64, C.c_uint64, vp, vp]
    L.sdfhip_get_stats.argtypes = [vp, C.POINTER(_Stats)]
    L.sdfhip_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.sdfhip_group_create.argtypes = [_fp, C.c_uint32, _up, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                      C.POINTER(C.c_void_p)]
    L.sdfhip_group_destroy.argtypes = [vp]
    L.sdfhip_group_destroy.restype = None
    L.sdfhip_group_size.argtypes = [vp]
    L.sdfhip_group_member.argtypes = [vp, C.c_int]
    L.sdfhip_group_member.restype = C.c_void_p
    L.sdfhip_group_update.argtypes = [vp]
    L.sdfhip_group_set_raycast_axes.argtypes = [vp, _fp, C.c_int]
    L.sdfhip_group_get_raycast_axes.argtypes = [vp, _fp]
    L.sdfhip_group_sdf.argtypes = [vp, _fp, C.c_int64, C.c_int, _fp]
    L.sdfhip_group_contains.argtypes = [vp, _fp, C.c_int64, _bp]
    L.sdfhip_group_nn.argtypes = [vp, _fp, C.c_int64, _ip]
    pp, i64p = C.POINTER(C.c_void_p), C.POINTER(C.c_int64)
    L.sdfhip_group_sdf_device.argtypes = [vp, pp, i64p, C.c_int, pp, pp]
    L.sdfhip_group_contains_device.argtypes = [vp, pp, i64p, pp, pp]
    L.sdfhip_group_nn_device.argtypes = [vp, pp, i64p, pp, pp]
    L.sdfhip_group_update_from_device.argtypes = [vp, vp, vp]
    L.sdfhip_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.sdfhip_host_free.argtypes = [C.c_void_p]
    L.sdfhip_renderer_create.argtypes = [_fp, C.c_uint32, _up, C.c_uint32, C.c_int, C.c_int, C.c_float,
                                         C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                         C.POINTER(vp)]
    L.sdfhip_renderer_destroy.restype = None
    L.sdfhip_renderer_destroy.argtypes = [vp]
    L.sdfhip_renderer_update.argtypes = [vp]
    L.sdfhip_renderer_render_depth.argtypes = [vp, _fp]
    L.sdfhip_renderer_render_mask.argtypes = [vp, _bp]
    L.sdfhip_renderer_render_nn.argtypes = [vp, C.c_int, _ip]
    L.sdfhip_renderer_depth.argtypes = [vp, _fp, C.c_int64, _fp]
    L.sdfhip_renderer_contains.argtypes = [vp, _fp, C.c_int64, _bp]
    L.sdfhip_renderer_nn.argtypes = [vp, _fp, C.c_int64, C.c_int, _ip]
    L.sdfhip_renderer_query_device.argtypes = [vp, C.c_int, vp, C.c_int64, C.c_int, vp, vp]
    for name in ("own_data", "device", "width", "height"):
        getattr(L, "sdfhip_renderer_" + name).argtypes = [vp]
    L.sdfhip_renderer_num_verts.restype = C.c_uint32
    L.sdfhip_renderer_num_verts.argtypes = [vp]
    L.sdfhip_renderer_num_faces.restype = C.c_uint32
    L.sdfhip_renderer_num_faces.argtypes = [vp]
    L.sdfhip_renderer_verts.restype = _fp
    L.sdfhip_renderer_verts.argtypes = [vp]
    L.sdfhip_renderer_faces.restype = _up
    L.sdfhip_renderer_faces.argtypes = [vp]
    L.sdfhip_renderer_verts_mutable.restype = _fp
    L.sdfhip_renderer_verts_mutable.argtypes = [vp]
    L.sdfhip_renderer_faces_mutable.restype = _up
    L.sdfhip_renderer_faces_mutable.argtypes = [vp]
    L.sdfhip_last_error.restype = C.c_char_p
    L.sdfhip_version.restype = C.c_char_p
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise SdfHipError(f"sdfhip error {rc}: {load_library().sdfhip_last_error().decode()}")


class _PinnedBlock:
    """owner of one sdfhip_host_alloc block; numpy arrays made from it keep it alive"""

    def __init__(self, nbytes):
        self._L = load_library()
        self.ptr = C.c_void_p()
        _check(self._L.sdfhip_host_alloc(max(1, int(nbytes)), C.byref(self.ptr)))
        self.nbytes = int(nbytes)

    def __del__(self):
        p = getattr(self, "ptr", None)
        if p:
            self._L.sdfhip_host_free(p)
            self.ptr = None


def pinned_empty(shape, dtype=np.float32):
    """numpy array in page-locked host memory (sdfhip_host_alloc): points kept in such arrays, and
    results written into them through `out=`, cross PCIe by DMA at the full rate (no counterpart
    in pysdf, whose arrays are pageable; plain numpy arrays work everywhere too)."""
    dt = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    blk = _PinnedBlock(n)
    buf = (C.c_char * max(1, n)).from_address(blk.ptr.value)
    buf._sdfhip_owner = blk  # the ctypes buffer is the array's base: the block lives as long
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


def _is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _as_points(points):
    p = np.asarray(points)
    if p.ndim == 1 and p.shape[0] == 3:
        p = p.reshape(1, 3)
    if p.ndim != 2 or p.shape[1] != 3:
        raise TypeError("points must have shape (N, 3)")
    return np.ascontiguousarray(p, dtype=np.float32)


class GroupSDF:
    """One mesh on several GPUs of this process (sdfhip_group_*): the host-array queries of `SDF`, the
    points cut into contiguous ranges, one per listed device (an ordinal may repeat).  No counterpart in
    pysdf; same query signatures and results as `SDF`."""

    def __init__(self, verts, faces, robust=True, copy=True, devices=(0,)):
        L = load_library()
        v32 = np.ascontiguousarray(np.asarray(verts), dtype=np.float32)
        f32 = np.ascontiguousarray(np.asarray(faces), dtype=np.uint32)
        if v32.ndim != 2 or v32.shape[1] != 3 or f32.ndim != 2 or f32.shape[1] != 3:
            raise TypeError("verts must be (V, 3) and faces (F, 3)")
        if not copy and (not np.shares_memory(v32, verts) or not np.shares_memory(f32, faces)):
            raise TypeError("copy=False needs C-contiguous float32 verts and uint32 faces")
        self._keep = None if copy else (v32, f32)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        g = C.c_void_p()
        _check(L.sdfhip_group_create(v32.ctypes.data_as(_fp), v32.shape[0], f32.ctypes.data_as(_up), f32.shape[0],
                                     int(bool(robust)), int(bool(copy)), devs, len(devices), C.byref(g)))
        self._g, self._L, self._nv, self._copy = g, L, v32.shape[0], bool(copy)

    def __del__(self):
        g = getattr(self, "_g", None)
        if g:
            self._L.sdfhip_group_destroy(g)
            self._g = None

    def __len__(self):
        return self._L.sdfhip_group_size(self._g)

    def __call__(self, points, trunc_aabb=False, n_threads=None):
        p = _as_points(points)
        out = np.empty(p.shape[0], dtype=np.float32)
        _check(self._L.sdfhip_group_sdf(self._g, p.ctypes.data_as(_fp), p.shape[0], int(trunc_aabb),
                                        out.ctypes.data_as(_fp)))
        return out

    calc = __call__

    def contains(self, points, n_threads=None):
        p = _as_points(points)
        out = np.empty(p.shape[0], dtype=np.uint8)
        _check(self._L.sdfhip_group_contains(self._g, p.ctypes.data_as(_fp), p.shape[0], out.ctypes.data_as(_bp)))
        return out.astype(bool)

    def nn(self, points, n_threads=None):
        p = _as_points(points)
        out = np.empty(p.shape[0], dtype=np.int32)
        _check(self._L.sdfhip_group_nn(self._g, p.ctypes.data_as(_fp), p.shape[0], out.ctypes.data_as(_ip)))
        return out

    def update(self):
        _check(self._L.sdfhip_group_update(self._g))

    def update_from_device(self, verts_tensor):
        """update() with the new vertex positions in a (V, 3) float32 tensor on member 0's GPU"""
        import torch
        t = verts_tensor.contiguous().float()
        if tuple(t.shape) != (self._nv, 3):
            raise TypeError("verts tensor must have the shape of the mesh vertices")
        s = torch.cuda.current_stream(t.device).cuda_stream
        _check(self._L.sdfhip_group_update_from_device(self._g, t.data_ptr(), s))

    @property
    def devices(self):
        return [int(self._L.sdfhip_device(self._L.sdfhip_group_member(self._g, k))) for k in range(len(self))]

    def member_stats(self, k):
        s = _Stats()
        _check(self._L.sdfhip_get_stats(self._L.sdfhip_group_member(self._g, k), C.byref(s)))
        return {name: getattr(s, name) for name, _ in _Stats._fields_}

    def _query_device(self, kind, shards, trunc_aabb=False):
        """`shards`: one (n_k, 3) float32 torch tensor per member, on that member's GPU (None or empty: that
        member idles).  Returns one result tensor per member on the same GPU; asynchronous on each GPU's
        current torch stream."""
        import torch
        G = len(self)
        if len(shards) != G:
            raise TypeError(f"one shard per member expected ({G}), got {len(shards)}")
        devs = self.devices
        dt = {"sdf": torch.float32, "contains": torch.uint8, "nn": torch.int32}[kind]
        pts, outs = [], []
        P, O, S = (C.c_void_p * G)(), (C.c_void_p * G)(), (C.c_void_p * G)()
        N = (C.c_int64 * G)()
        for k, t in enumerate(shards):
            dev = torch.device("cuda", devs[k])
            if t is None:
                t = torch.empty((0, 3), dtype=torch.float32, device=dev)
            if not (_is_torch(t) and t.is_cuda and t.device.index == devs[k]):
                raise SdfHipError(f"shard {k} must be a torch tensor on cuda:{devs[k]}")
            if t.dim() != 2 or t.shape[1] != 3:
                raise TypeError("every shard must have shape (n, 3)")
            t = t.contiguous().float()
            o = torch.empty(t.shape[0], dtype=dt, device=dev)
            pts.append(t), outs.append(o)
            P[k], O[k], N[k] = t.data_ptr(), o.data_ptr(), t.shape[0]
            S[k] = torch.cuda.current_stream(dev).cuda_stream
        if kind == "sdf":
            _check(self._L.sdfhip_group_sdf_device(self._g, P, N, int(trunc_aabb), O, S))
        elif kind == "contains":
            _check(self._L.sdfhip_group_contains_device(self._g, P, N, O, S))
            outs = [o.bool() for o in outs]
        else:
            _check(self._L.sdfhip_group_nn_device(self._g, P, N, O, S))
        return outs

    def sdf_device(self, shards, trunc_aabb=False):
        return self._query_device("sdf", shards, trunc_aabb)

    def contains_device(self, shards):
        return self._query_device("contains", shards)

    def nn_device(self, shards):
        return self._query_device("nn", shards)

    def set_raycast_axes(self, R, pin=True):
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        _check(self._L.sdfhip_group_set_raycast_axes(self._g, R.ctypes.data_as(_fp), int(pin)))

    @property
    def raycast_axes(self):
        R = np.empty(9, dtype=np.float32)
        _check(self._L.sdfhip_group_get_raycast_axes(self._g, R.ctypes.data_as(_fp)))
        return R.reshape(3, 3)

    @property
    def vertices_mutable(self):
        """the owned host copy (member 0's; every member re-reads it on update())"""
        if not self._copy:
            raise SdfHipError("'verts' is non mutable, construct with copy=True")
        h0 = self._L.sdfhip_group_member(self._g, 0)
        ptr = self._L.sdfhip_verts_mutable(h0)
        return np.ctypeslib.as_array(ptr, shape=(self._nv, 3))

    verts_mutable = vertices_mutable


class SDF:
    """Signed-distance / containment / nearest-vertex queries on a triangle mesh, on one
    MI355X.  Same call surface as `pysdf.SDF` (pybind.cpp:18-83)."""

    def __init__(self, verts, faces, robust=True, copy=True, device=None):
        L = load_library()
        v = np.asarray(verts)
        f = np.asarray(faces)
        if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
            raise TypeError("verts must be (V, 3) and faces (F, 3)")
        if f.size and (f.min() < 0):
            raise TypeError("faces must be non-negative")
        v32 = np.ascontiguousarray(v, dtype=np.float32)
        f32 = np.ascontiguousarray(f, dtype=np.uint32)
        if not copy:
            # Eigen::Ref semantics (sdf.cpp:339): the handle keeps reading this memory.  A dtype
            # or layout conversion would silently detach it from the caller's array.
            if v32 is not v and not np.shares_memory(v32, v):
                raise TypeError("copy=False needs C-contiguous float32 verts")
            if f32 is not f and not np.shares_memory(f32, f):
                raise TypeError("copy=False needs C-contiguous uint32 faces")
        self._keep = (v32, f32)  # keep borrowed memory alive
        h = C.c_void_p()
        dev = -1 if device is None else int(device)
        _check(L.sdfhip_create(v32.ctypes.data_as(_fp), v32.shape[0], f32.ctypes.data_as(_up),
                               f32.shape[0], int(bool(robust)), int(bool(copy)), dev, C.byref(h)))
        self._h = h
        self._L = L
        if copy:
            self._keep = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.sdfhip_destroy(h)
            self._h = None

    # ---- queries -----------------------------------------------------------------------
    def _query(self, kind, points, trunc_aabb=False, out=None):
        L = self._L
        if not _is_torch(points) and not isinstance(points, np.ndarray) and hasattr(points, "__dlpack__") \
                and hasattr(points, "__dlpack_device__"):
            # DLPack producers other than torch (CuPy, JAX, another framework's tensors): device arrays
            # (kDLCUDA = 2, kDLROCM = 10) reach the `_device` entry points without a copy, and the result
            # is a torch tensor on that device (itself a DLPack producer: `xp.from_dlpack(result)`);
            # host arrays take the numpy path.  pybind.cpp:23-31 accepts host arrays only.
            import torch
            if int(points.__dlpack_device__()[0]) in (2, 10):
                points = torch.from_dlpack(points)
            else:
                points = np.from_dlpack(points)
        if _is_torch(points):
            import torch
            t = points
            if t.dim() == 1 and t.shape[0] == 3:
                t = t.reshape(1, 3)
            if t.dim() != 2 or t.shape[1] != 3:
                raise TypeError("points must have shape (N, 3)")
            if t.is_cuda:
                if t.device.index != self.device:
                    raise SdfHipError("points tensor is on another GPU than the SDF handle")
                t = t.contiguous().float()
                n = t.shape[0]
                dt = {"sdf": torch.float32, "contains": torch.uint8, "nn": torch.int32}[kind]
                out = torch.empty(n, dtype=dt, device=t.device)
                s = torch.cuda.current_stream(t.device).cuda_stream
                if kind == "sdf":
                    _check(L.sdfhip_sdf_device(self._h, t.data_ptr(), n, int(trunc_aabb),
                                               out.data_ptr(), s))
                elif kind == "contains":
                    _check(L.sdfhip_contains_device(self._h, t.data_ptr(), n, out.data_ptr(), s))
                    out = out.bool()
                else:
                    _check(L.sdfhip_nn_device(self._h, t.data_ptr(), n, out.data_ptr(), s))
                return out
            points = t.detach().cpu().numpy()
        p = _as_points(points)
        n = p.shape[0]
        odt = {"sdf": np.float32, "contains": np.bool_, "nn": np.int32}[kind]
        if out is None:
            out = np.empty(n, dtype=odt)
        elif not (isinstance(out, np.ndarray) and out.dtype == odt and out.shape == (n,)
                  and out.flags.c_contiguous and out.flags.writeable):
            raise TypeError(f"out must be a writable C-contiguous numpy array of shape ({n},) and dtype {np.dtype(odt)}")
        if kind == "sdf":
            _check(L.sdfhip_sdf(self._h, p.ctypes.data_as(_fp), n, int(trunc_aabb),
                                out.ctypes.data_as(_fp)))
        elif kind == "contains":
            _check(L.sdfhip_contains(self._h, p.ctypes.data_as(_fp), n, out.ctypes.data_as(_bp)))
        else:
            _check(L.sdfhip_nn(self._h, p.ctypes.data_as(_fp), n, out.ctypes.data_as(_ip)))
        return out

    def __call__(self, points, trunc_aabb=False, n_threads=num_threads, *, out=None):
        """Compute SDF on points (positive iff inside).  `out` (numpy inputs only, no counterpart
        in pysdf): result array to fill instead of allocating one."""
        return self._query("sdf", points, trunc_aabb, out)

    def calc(self, points, trunc_aabb=False, n_threads=num_threads, *, out=None):
        """Compute SDF on points (positive iff inside), alias of __call__."""
        return self._query("sdf", points, trunc_aabb, out)

    def contains(self, points, n_threads=num_threads, *, out=None):
        """Test if points are in mesh."""
        return self._query("contains", points, out=out)

    def nn(self, points, n_threads=num_threads, *, out=None):
        """Find nearest neighbor indices."""
        return self._query("nn", points, out=out)

    def update(self):
        """Update the SDF to reflect any changes in verts."""
        _check(self._L.sdfhip_update(self._h))

    def update_from_device(self, verts_tensor):
        """update() with the new vertex positions taken from a GPU tensor (V, 3) float32."""
        import torch
        t = verts_tensor.contiguous().float()
        if tuple(t.shape) != (self._nv, 3):
            raise TypeError("verts tensor must have the shape of the mesh vertices")
        s = torch.cuda.current_stream(t.device).cuda_stream
        _check(self._L.sdfhip_update_from_device(self._h, t.data_ptr(), s))

    def raycast_detail(self, points):
        p = _as_points(points)
        out = np.empty((p.shape[0], 3), dtype=np.int32)
        _check(self._L.sdfhip_raycast_detail(self._h, p.ctypes.data_as(_fp), p.shape[0],
                                             out.ctypes.data_as(_ip)))
        return out

    def sample_surface(self, num_points, seed=None, device=False):
        """Sample num_points uniformly random points on the mesh surface (sdf.cpp:176-234,
        pybind.cpp:31): a face with probability proportional to its area, then a uniform point
        in it (parallelogram sample reflected over bc).  Runs on the GPU (`k_sample_surface`).
        `seed` (extension; default: fresh entropy, like the reference's time-seeded generator)
        makes the call reproducible; `device=True` returns a torch tensor on the handle's GPU."""
        n = int(num_points)
        if self.faces.shape[0] == 0:
            import sys
            print("ERROR: No faces, can't sample surface.", file=sys.stderr)  # sdf.cpp:178
            return np.zeros((0, 3), dtype=np.float32)
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if device:
            import torch
            dev = torch.device("cuda", self._L.sdfhip_device(self._h))
            out = torch.empty((max(n, 0), 3), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _check(self._L.sdfhip_sample_surface_device(self._h, n, seed, out.data_ptr(), stream))
            return out
        out = np.empty((max(n, 0), 3), dtype=np.float32)
        _check(self._L.sdfhip_sample_surface(self._h, n, seed, out.ctypes.data_as(_fp)))
        return out

    # ---- rotation injection (no counterpart: the reference's R is private + random) -----
    def set_raycast_axes(self, R, pin=True):
        R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
        _check(self._L.sdfhip_set_raycast_axes(self._h, R.ctypes.data_as(_fp), int(pin)))

    @property
    def raycast_axes(self):
        R = np.empty(9, dtype=np.float32)
        _check(self._L.sdfhip_get_raycast_axes(self._h, R.ctypes.data_as(_fp)))
        return R.reshape(3, 3)

    def set_option(self, key, value):
        _check(self._L.sdfhip_set_option(self._h, key.encode(), int(value)))

    @property
    def stats(self):
        s = _Stats()
        _check(self._L.sdfhip_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _Stats._fields_}

    # ---- data accessors (pybind.cpp:39-77) ------------------------------------------------
    @property
    def _nv(self):
        return int(self._L.sdfhip_num_verts(self._h))

    @property
    def _nf(self):
        return int(self._L.sdfhip_num_faces(self._h))

    @property
    def device(self):
        return int(self._L.sdfhip_device(self._h))

    @property
    def surface_area(self):
        out = C.c_float()
        _check(self._L.sdfhip_surface_area(self._h, C.byref(out)))
        return float(out.value)

    @property
    def face_areas(self):
        out = np.empty(self._nf, dtype=np.float32)
        _check(self._L.sdfhip_face_areas(self._h, out.ctypes.data_as(_fp)))
        return out

    @property
    def face_normals(self):
        out = np.empty((self._nf, 3), dtype=np.float32)
        _check(self._L.sdfhip_face_normals(self._h, out.ctypes.data_as(_fp)))
        return out

    @property
    def aabb(self):
        out = np.empty(6, dtype=np.float32)
        _check(self._L.sdfhip_aabb(self._h, out.ctypes.data_as(_fp)))
        return out

    def _view(self, ptr, shape, dtype, writable):
        if shape[0] == 0:
            return np.zeros(shape, dtype=dtype)
        n = int(np.prod(shape))
        arr = np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape)
        arr.flags.writeable = writable
        return arr

    @property
    def faces(self):
        return self._view(self._L.sdfhip_faces(self._h), (self._nf, 3), np.uint32, False)

    @property
    def verts(self):
        return self._view(self._L.sdfhip_verts(self._h), (self._nv, 3), np.float32, False)

    vertices = verts

    def _mutable(self, which):
        import sys
        if not self.own_data:
            # sdf.cpp:360-375: prints and returns the (empty) owned matrix
            print(f"ERROR: '{which}' is non mutable, construct with copy=True", file=sys.stderr)
            return np.zeros((0, 3), dtype=np.float32 if which == "verts" else np.uint32)
        if which == "verts":
            return self._view(self._L.sdfhip_verts_mutable(self._h), (self._nv, 3), np.float32, True)
        return self._view(self._L.sdfhip_faces_mutable(self._h), (self._nf, 3), np.uint32, True)

    @property
    def verts_mutable(self):
        return self._mutable("verts")

    @verts_mutable.setter
    def verts_mutable(self, val):
        dst = self._mutable("verts")
        dst[...] = np.asarray(val, dtype=np.float32).reshape(dst.shape)  # pybind.cpp:60-62

    vertices_mutable = verts_mutable

    @property
    def faces_mutable(self):
        return self._mutable("faces")

    @faces_mutable.setter
    def faces_mutable(self, val):
        dst = self._mutable("faces")
        dst[...] = np.asarray(val, dtype=np.uint32).reshape(dst.shape)

    @property
    def robust(self):
        return bool(self._L.sdfhip_robust(self._h))

    @property
    def own_data(self):
        return bool(self._L.sdfhip_own_data(self._h))

    def adj_faces(self, pointid):
        ids = _ip()
        n = self._L.sdfhip_adj_faces(self._h, int(pointid), C.byref(ids))
        if n < 0:
            _check(n)
        return [int(ids[i]) for i in range(n)]

    def __repr__(self):
        return (f"<sdf.SDF(verts.shape=({self._nv}, 3), faces.shape=({self._nf}, 3), "
                f"robust={'True' if self.robust else 'False'})>")


def _as_points2(points):
    p = np.asarray(points)
    if p.ndim == 1 and p.shape[0] == 2:
        p = p.reshape(1, 2)
    if p.ndim != 2 or p.shape[1] != 2:
        raise TypeError("points must have shape (N, 2)")
    return np.ascontiguousarray(p, dtype=np.float32)


class Renderer:
    """Image-space raycast renderer (depth maps, masks, vertex ids) on one MI355X.  Same call
    surface as `pysdf.Renderer` (pybind.cpp:85-161; sdf.hpp:282-369): camera at the origin facing
    +z, right = +x, up = -y; nothing may lie at z <= 0."""

    _KIND = {"depth": (0, np.float32), "mask": (1, np.bool_), "nn": (2, np.int32)}

    def __init__(self, verts, faces, width=1080, height=1080, fx=2600.0, fy=2600.0, cx=540.0,
                 cy=540.0, copy=True, device=None):
        L = load_library()
        v = np.asarray(verts)
        f = np.asarray(faces)
        if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
            raise TypeError("verts must be (V, 3) and faces (F, 3)")
        if f.size and (f.min() < 0):
            raise TypeError("faces must be non-negative")
        v32 = np.ascontiguousarray(v, dtype=np.float32)
        f32 = np.ascontiguousarray(f, dtype=np.uint32)
        if not copy:
            if v32 is not v and not np.shares_memory(v32, v):
                raise TypeError("copy=False needs C-contiguous float32 verts")
            if f32 is not f and not np.shares_memory(f32, f):
                raise TypeError("copy=False needs C-contiguous uint32 faces")
        self._keep = None if copy else (v32, f32)
        h = C.c_void_p()
        dev = -1 if device is None else int(device)
        _check(L.sdfhip_renderer_create(v32.ctypes.data_as(_fp), v32.shape[0], f32.ctypes.data_as(_up),
                                        f32.shape[0], int(width), int(height), float(fx), float(fy),
                                        float(cx), float(cy), int(bool(copy)), dev, C.byref(h)))
        self._h = h
        self._L = L

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.sdfhip_renderer_destroy(h)
            self._h = None

    def _query(self, kind, points, fill_outside=False):
        L = self._L
        k, dt = self._KIND[kind]
        if points is not None and _is_torch(points) and points.is_cuda:
            import torch
            t = points
            if t.dim() == 1 and t.shape[0] == 2:
                t = t.reshape(1, 2)
            if t.dim() != 2 or t.shape[1] != 2:
                raise TypeError("points must have shape (N, 2)")
            if t.device.index != self.device:
                raise SdfHipError("points tensor is on another GPU than the Renderer handle")
            t = t.contiguous().float()
            n = t.shape[0]
            out = torch.empty(n, dtype={0: torch.float32, 1: torch.uint8, 2: torch.int32}[k], device=t.device)
            s = torch.cuda.current_stream(t.device).cuda_stream
            _check(L.sdfhip_renderer_query_device(self._h, k, t.data_ptr(), n, int(bool(fill_outside)),
                                                  out.data_ptr(), s))
            return out.bool() if k == 1 else out
        if points is not None and _is_torch(points):
            points = points.detach().cpu().numpy()
        if points is None:
            hh, ww = self.height, self.width
            out = np.empty(hh * ww, dtype=dt)
            if k == 0:
                _check(L.sdfhip_renderer_render_depth(self._h, out.ctypes.data_as(_fp)))
            elif k == 1:
                _check(L.sdfhip_renderer_render_mask(self._h, out.ctypes.data_as(_bp)))
            else:
                _check(L.sdfhip_renderer_render_nn(self._h, int(bool(fill_outside)), out.ctypes.data_as(_ip)))
            return out.reshape(hh, ww)
        p = _as_points2(points)
        n = p.shape[0]
        out = np.empty(n, dtype=dt)
        if k == 0:
            _check(L.sdfhip_renderer_depth(self._h, p.ctypes.data_as(_fp), n, out.ctypes.data_as(_fp)))
        elif k == 1:
            _check(L.sdfhip_renderer_contains(self._h, p.ctypes.data_as(_fp), n, out.ctypes.data_as(_bp)))
        else:
            _check(L.sdfhip_renderer_nn(self._h, p.ctypes.data_as(_fp), n, int(bool(fill_outside)),
                                        out.ctypes.data_as(_ip)))
        return out

    def __call__(self, points, n_threads=num_threads):
        """Compute depth on 2D image-space points (0 if nothing there)"""
        return self._query("depth", points)

    def calc(self, points, n_threads=num_threads):
        """Compute depth on 2D image-space points (0 if nothing there), alias of __call__"""
        return self._query("depth", points)

    def contains(self, points, n_threads=num_threads):
        """True where the ray cast from (x, y, 0) in +z direction in image space hits the mesh."""
        return self._query("mask", points)

    def nn(self, points, fill_outside=True, n_threads=num_threads):
        """Index of the closest vertex hit by a ray cast from (x, y, 0); in empty space the 2D
        nearest neighbour if fill_outside else -1."""
        return self._query("nn", points, fill_outside)

    def render_depth(self, n_threads=num_threads):
        """(height, width) float32 depth image, 0 = no object."""
        return self._query("depth", None)

    def render_mask(self, n_threads=num_threads):
        """(height, width) bool silhouette."""
        return self._query("mask", None)

    def render_nn(self, fill_outside=False, n_threads=num_threads):
        """(height, width) int32 vertex ids."""
        return self._query("nn", None, fill_outside)

    def update(self):
        """Update the Renderer to reflect any changes in verts"""
        _check(self._L.sdfhip_renderer_update(self._h))

    @property
    def device(self):
        return int(self._L.sdfhip_renderer_device(self._h))

    @property
    def width(self):
        return int(self._L.sdfhip_renderer_width(self._h))

    @property
    def height(self):
        return int(self._L.sdfhip_renderer_height(self._h))

    @property
    def _nv(self):
        return int(self._L.sdfhip_renderer_num_verts(self._h))

    @property
    def _nf(self):
        return int(self._L.sdfhip_renderer_num_faces(self._h))

    _view = SDF._view

    @property
    def faces(self):
        return self._view(self._L.sdfhip_renderer_faces(self._h), (self._nf, 3), np.uint32, False)

    @property
    def verts(self):
        return self._view(self._L.sdfhip_renderer_verts(self._h), (self._nv, 3), np.float32, False)

    vertices = verts

    @property
    def own_data(self):
        return bool(self._L.sdfhip_renderer_own_data(self._h))

    def _mutable(self, which):
        import sys
        if not self.own_data:  # renderer.cpp:289-304: prints and returns the (empty) owned matrix
            print(f"ERROR: '{which}' is non mutable, construct with copy=True", file=sys.stderr)
            return np.zeros((0, 3), dtype=np.float32 if which == "verts" else np.uint32)
        if which == "verts":
            return self._view(self._L.sdfhip_renderer_verts_mutable(self._h), (self._nv, 3), np.float32, True)
        return self._view(self._L.sdfhip_renderer_faces_mutable(self._h), (self._nf, 3), np.uint32, True)

    @property
    def verts_mutable(self):
        return self._mutable("verts")

    @verts_mutable.setter
    def verts_mutable(self, val):
        dst = self._mutable("verts")
        dst[...] = np.asarray(val, dtype=np.float32).reshape(dst.shape)

    vertices_mutable = verts_mutable

    @property
    def faces_mutable(self):
        return self._mutable("faces")

    @faces_mutable.setter
    def faces_mutable(self, val):
        dst = self._mutable("faces")
        dst[...] = np.asarray(val, dtype=np.uint32).reshape(dst.shape)

    def __repr__(self):  # pybind.cpp:157-161 (the reference's string has this unbalanced bracket)
        return f"<sdf.Renderer(verts.shape=({self._nv}, 3), faces.shape=({self._nf}, 3)>"
