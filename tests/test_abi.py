"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/sdfhip.h declares; host-side argument checking works without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sdf_amd
from sdf_amd import pysdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sdfhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdfhip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = pysdf.load_library()
    decl = _declared()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(L, name), f"{name} declared in sdfhip.h but not exported"
    assert sorted(pysdf.ABI_SYMBOLS) == decl


def test_version_and_error_string():
    L = pysdf.load_library()
    assert b"gfx950" in L.sdfhip_version()
    assert isinstance(L.sdfhip_last_error(), bytes)


def test_create_rejects_bad_arguments_before_touching_the_gpu():
    L = pysdf.load_library()
    v = np.zeros((3, 3), np.float32)
    f = np.array([[0, 1, 7]], np.uint32)  # index out of range
    h = C.c_void_p()
    rc = L.sdfhip_create(v.ctypes.data_as(pysdf._fp), 3, f.ctypes.data_as(pysdf._up), 1, 1, 1, -1,
                         C.byref(h))
    assert rc == -1 and b"out of range" in L.sdfhip_last_error() and not h.value
    assert L.sdfhip_create(None, 3, None, 1, 1, 1, -1, C.byref(h)) == -1


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    v, f = sdf_amd.meshes.cube()
    with pytest.raises(sdf_amd.SdfHipError) as e:
        sdf_amd.SDF(v, f)
    assert "no HIP device" in str(e.value) or "error -3" in str(e.value) or "error -2" in str(e.value)


def test_python_surface_matches_pysdf():
    """names and defaults of pybind.cpp:18-83, 195"""
    import inspect
    S = sdf_amd.SDF
    sig = inspect.signature(S.__init__)
    assert list(sig.parameters)[1:5] == ["verts", "faces", "robust", "copy"]
    assert sig.parameters["robust"].default is True and sig.parameters["copy"].default is True
    def positional(fn):  # keyword-only extras (out=) are additions, not part of pysdf's surface
        return [k for k, q in inspect.signature(fn).parameters.items() if q.kind == q.POSITIONAL_OR_KEYWORD][1:]
    for name in ("__call__", "calc"):
        p = inspect.signature(getattr(S, name)).parameters
        assert positional(getattr(S, name)) == ["points", "trunc_aabb", "n_threads"] and p["trunc_aabb"].default is False
    for name in ("contains", "nn"):
        assert positional(getattr(S, name)) == ["points", "n_threads"]
    for name in ("update", "sample_surface", "surface_area", "face_areas", "face_normals", "aabb",
                 "faces", "verts", "vertices", "faces_mutable", "verts_mutable", "vertices_mutable",
                 "robust", "own_data", "__repr__"):
        assert hasattr(S, name)
    assert sdf_amd.num_threads == min(os.cpu_count(), 32)


def test_cpp_facade_compiles_and_fails_loudly_without_gpu(tmp_path):
    """include/sdf_amd/sdf.hpp mirrors sdf::SDF (sdf.hpp:150-258); tests/cpp/facade_smoke.cpp uses it."""
    import subprocess
    import torch
    exe = str(tmp_path / "facade_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "facade_smoke.cpp"),
                           "-L" + os.path.join(ROOT, "sdf_amd"), "-lsdfhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "sdf_amd"), "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "no HIP device" in r.stdout


def test_parallel_kd_build_is_identical_to_serial(tmp_path):
    import subprocess
    exe = str(tmp_path / "kdchk")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread",
                           os.path.join(ROOT, "tests", "cpp", "kd_build_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_packet_kernel_keeps_seven_waves_per_simd(tmp_path):
    """The dominant kernel runs 7 waves per SIMD only with <= 72 VGPRs and <= 96 SGPRs and no scratch
    ("registers"): two more VGPRs cost 4 % of the step.  Read from the compiler's own metadata."""
    import subprocess
    asm = str(tmp_path / "sdfhip.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                           "-fno-slp-vectorize", "-Wno-unused-result", "--cuda-device-only", "-S", "-o", asm,
                           os.path.join(ROOT, "sdf_amd", "csrc", "sdfhip.hip")])
    text = open(asm).read()
    seen = 0
    for m in re.finditer(r"\.name:\s+(_Z14k_query_packetILi([02])ELb([01])ELb1EE\S*)\n(.*?)\.wavefront_size", text, re.S):
        body = m.group(4)
        vg = int(re.search(r"\.vgpr_count:\s+(\d+)", body).group(1))
        sg = int(re.search(r"\.sgpr_count:\s+(\d+)", body).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
        assert vg <= 72 and sg <= 96 and scratch == 0, (m.group(1), vg, sg, scratch)
        seen += 1
    assert seen == 4  # nn / sdf x robust / non-robust, approximate walk
