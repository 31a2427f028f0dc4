#!/usr/bin/env python3
"""tools/bench_misc.py -- numbers quoted in DESIGN.md that are not the driver's metric: encoder construction time
(twiddle table / matgen, outside the timed commit as in the reference's rough_bench) and the PCIe-inclusive commit
rate of the host-pointer entry point lcpc_commit."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import lcpc_amd
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding

for name, mk in (("LigeroEncoding::new(2^26) ft255", lambda: LigeroEncoding.new(3, 1 << 26)),
                 ("LigeroEncoding::new(2^28) ft255", lambda: LigeroEncoding.new(3, 1 << 28)),
                 ("SdigEncoding::new(2^24, seed 0) ft255", lambda: SdigEncoding.new(3, 1 << 24, 0)),
                 ("SdigEncoding::new(2^26, seed 0) ft255", lambda: SdigEncoding.new(3, 1 << 26, 0))):
    t0 = time.perf_counter()
    e = mk()
    print(json.dumps({"construct": name, "seconds": round(time.perf_counter() - t0, 3), "dims": e.get_dims(1)[1:]}), flush=True)
    del e

n = 1 << 26
enc = LigeroEncoding.new(3, n)
rng = np.random.default_rng(1)
host = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
pinned = torch.from_numpy(host.view(np.int64)).pin_memory()
for label, arr in (("pageable numpy", host), ("pinned (torch.pin_memory)", pinned.numpy().view(np.uint64))):
    for _ in range(2):
        t0 = time.perf_counter()
        c = LcCommit.commit(arr, enc)
        c.get_root()                      # synchronises: the commit itself is enqueued asynchronously
        dt = time.perf_counter() - t0
    print(json.dumps({"lcpc_commit host pointer, 2^26 ft255": label, "ms": round(dt * 1e3, 2), "elems_per_s": n / dt,
                      "h2d_GBps_equiv": round(n * 32 / dt / 1e9, 1)}), flush=True)
