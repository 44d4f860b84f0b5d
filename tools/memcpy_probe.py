#!/usr/bin/env python3
"""Host memcpy rates on the box (pageable -> pageable / page-locked), single thread and threaded: what a bounce-buffer
copy of caller memory can reach."""
import time, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd
n = 96_000_000
src = np.random.default_rng(0).random(n, dtype=np.float32)
dst = np.empty_like(src)
pin = sdf_amd.pinned_empty(n, np.float32)
def best(fn, reps=4):
    fn(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return src.nbytes / b / 1e9
print("numpy copyto pageable->pageable (1 thread): %.1f GB/s" % best(lambda: np.copyto(dst, src)))
print("numpy copyto pageable->pinned   (1 thread): %.1f GB/s" % best(lambda: np.copyto(pin, src)))
ts, tp = torch.from_numpy(src), torch.from_numpy(pin)
for th in (1, 2, 4, 8):
    torch.set_num_threads(th)
    print("torch copy_ pageable->pinned (%d threads): %.1f GB/s" % (th, best(lambda: tp.copy_(ts))))
