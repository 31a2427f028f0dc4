#!/usr/bin/env python3
"""tools/bench_host_first_call.py -- what an encoder's FIRST lcpc_commit from pageable memory costs beside the steady state (the pinned
bounce ring is allocated as its buffers are first used), with LCPC_HOST_STAGE=1 (ring) and =0 (the runtime's own path).  profiles/r05_host_path.jsonl."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lcpc_amd import LcCommit, LigeroEncoding
n = 1 << 26
torch.zeros(1).cuda()
host = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
for stage in ("1", "0"):
    os.environ["LCPC_HOST_STAGE"] = stage
    enc = LigeroEncoding.new(3, n)
    obj = LcCommit(enc)
    LcCommit.commit_device(torch.from_numpy(host[:1 << 20].view(np.int64)).cuda().data_ptr(), 1 << 20, LigeroEncoding.new(3, 1 << 20))  # warm the runtime
    ts = []
    for i in range(3):
        t0 = time.perf_counter(); LcCommit.commit(host, enc, into=obj).get_root(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(json.dumps({"LCPC_HOST_STAGE": stage, "first_second_third_call_ms": ts, "note": "fresh encoder and LcCommit: the first call allocates 6 GiB of HBM and (stage=1) the 4 x 64 MiB pinned ring"}))
    del obj, enc
