#!/usr/bin/env python3
"""tools/bench_shard_rank.py -- per-rank cost of the row-sharded 2^26 commit for N = 1, 2, 4, 8 on ONE GPU: the
local encode + chunk CVs + subtree pre-merge and the post-exchange finish, with the all-gather replaced by a
buffer of the right size (so: everything except the RCCL time).  Gives the strong-scaling ceiling quoted in DESIGN.md."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import lcpc_amd
from lcpc_amd import LigeroEncoding
from lcpc_amd.distributed import HipShardEngine, slots_per_rank

BASE = {}

n_rows, npr, nc = lcpc_amd.static_get_dims(3, 0, 1 << 26)
for world in (1, 2, 4, 8):
    rank = world - 1          # the last rank owns the extra tail chunk: the slowest one
    enc = LigeroEncoding.new_from_dims(3, npr, nc, shard=(rank, world))
    eng = HipShardEngine(enc)
    rb, re, cb, ce, nch = eng.layout(n_rows)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    coeffs = torch.randint(-(1 << 63), (1 << 63) - 1, ((re - rb) * npr, 4), dtype=torch.int64, device="cuda", generator=g)
    coeffs[:, 3] &= (1 << 62) - 1
    slots = slots_per_rank(nch, world)
    gathered = torch.zeros((world * slots, nc, 32), dtype=torch.uint8, device="cuda")

    def step():
        nodes = eng.commit_shard(coeffs, n_rows)
        gathered[rank * slots:rank * slots + nodes.shape[0]] = nodes      # stands in for the all-gather
        eng.commit_finish(gathered, n_rows, slots, want_root=False)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(json.dumps({"world": world, "rank": rank, "rows": re - rb, "chunks": ce - cb, "slots_per_rank": slots,
                      "gather_MB": round(world * slots * nc * 32 / 1e6, 1), "ms_per_step_without_exchange": round(dt * 1e3, 3),
                      "ceiling_speedup": round(BASE.setdefault("ms", dt * 1e3) / (dt * 1e3), 2)}), flush=True)   # vs this run's world = 1
    del enc, eng, coeffs, gathered
