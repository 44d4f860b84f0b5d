import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sdf_amd
from sdf_amd import meshes
v, f = meshes.torus(424, 424)
s = sdf_amd.SDF(v, f); s.set_raycast_axes(meshes.random_rotation(1234))
n = 32_000_000
pts = np.random.default_rng(7).uniform(-1, 1, (n, 3)).astype(np.float32)
out = np.zeros(n, np.float32)
for name, fn in (("sdf out=", lambda: s(pts, out=out)), ("sdf fresh result", lambda: s(pts)), ("contains fresh", lambda: s.contains(pts)), ("nn fresh", lambda: s.nn(pts))):
    fn(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    print(f"{name:18s} {best*1e3:6.1f} ms  {n/best/1e6:7.1f} Mevals/s")
