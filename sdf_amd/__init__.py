"""sdf_amd — MI355X-native batched point queries of sxyu/sdf (`SDF.__call__`, `contains`, `nn`).

`from sdf_amd import SDF` is the drop-in for `from pysdf import SDF`.
"""
from .pysdf import SDF, GroupSDF, Renderer, SdfHipError, load_library, num_threads, pinned_empty  # noqa: F401
from . import meshes  # noqa: F401
from . import util  # noqa: F401  (pysdf.util, pybind.cpp:163-194)

__all__ = ["SDF", "GroupSDF", "Renderer", "SdfHipError", "load_library", "num_threads", "pinned_empty", "meshes", "util"]
