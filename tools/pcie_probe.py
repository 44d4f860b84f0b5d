#!/usr/bin/env python3
"""Host<->device copy rates on this box: pageable vs pinned, and host memcpy into pinned staging."""
import time, torch, numpy as np
dev = torch.device("cuda", 0)
n = 256 << 20  # bytes
d = torch.empty(n, dtype=torch.uint8, device=dev)
pg = torch.empty(n, dtype=torch.uint8)
pg.fill_(1)
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n / best / 1e9
print("H2D pageable  %.1f GB/s" % t(lambda: d.copy_(pg)))
print("H2D pinned    %.1f GB/s" % t(lambda: d.copy_(pin, non_blocking=True)))
print("D2H pageable  %.1f GB/s" % t(lambda: pg.copy_(d)))
print("D2H pinned    %.1f GB/s" % t(lambda: pin.copy_(d, non_blocking=True)))
print("host memcpy pageable->pinned (1 thread) %.1f GB/s" % t(lambda: pin.copy_(pg)))
torch.set_num_threads(8)
print("host memcpy pageable->pinned (torch 8 threads) %.1f GB/s" % t(lambda: pin.copy_(pg)))
# bidirectional pinned
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
pin2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
def bidir():
    with torch.cuda.stream(s1): d.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2): pin2.copy_(d2, non_blocking=True)
print("bidirectional pinned (each way) %.1f GB/s" % t(bidir))
