#!/usr/bin/env python3
"""tools/fuzz_long.py [n_seeds] -- the seeded random-shape parity sweep of tests/test_gpu_fuzz.py with many more seeds
(one-off soak on the GPU box; not part of the test suite)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import oracle_lib as O
import test_gpu_fuzz as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
for seed in range(100, 100 + n):
    T.test_fuzz_ligero.__wrapped__(O, seed) if hasattr(T.test_fuzz_ligero, "__wrapped__") else T.test_fuzz_ligero(O, seed)
    if seed % 4 == 0:
        T.test_fuzz_brakedown(O, seed)
    if seed % 20 == 0:
        print("seed", seed, "ok, %.0f s" % (time.time() - t0), flush=True)
print("all", n, "seeds ok")
