#!/usr/bin/env python3
"""tools/bench_host_path.py -- the host-pointer entry lcpc_commit (== LcCommit::commit(&coeffs, &enc) for a Rust caller,
lcpc-2d/src/lib.rs:299-301,636-645) from PAGEABLE memory (a plain Vec / numpy / malloc buffer) against the same call from
pinned memory and against the device-resident commit.  One JSON line per leg; kept under profiles/rNN_host_path.jsonl.

  python tools/bench_host_path.py [--log2 26] [--reps 5] [--tag before|after]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import lcpc_amd
from lcpc_amd import LcCommit, LigeroEncoding

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=26)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tag", default="")
ap.add_argument("--field", type=int, default=3)
args = ap.parse_args()

n = 1 << args.log2
L = (1, 2, 3, 4)[args.field]
enc = LigeroEncoding.new(args.field, n)
rng = np.random.default_rng(1)
host = rng.integers(0, 1 << 62, size=(n, L), dtype=np.uint64)          # pageable: numpy's allocator (malloc / mmap)
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [ctypes.c_size_t]
libc.free.argtypes = [ctypes.c_void_p]
raw = libc.malloc(host.nbytes)                                          # pageable: plain malloc, touched once
mal = np.ctypeslib.as_array(ctypes.cast(raw, ctypes.POINTER(ctypes.c_uint64)), shape=(n, L))
mal[:] = host
pinned = torch.from_numpy(host.view(np.int64)).pin_memory()
dev = pinned.cuda()


def emit(d):
    d["tag"] = args.tag
    d["LCPC_HOST_STAGE"] = os.environ.get("LCPC_HOST_STAGE", "")
    d["log2"] = args.log2
    d["field"] = args.field
    d["cores"] = os.cpu_count()
    print(json.dumps(d), flush=True)


roots = {}
obj = LcCommit(enc)                                                     # one commitment refilled: steady state, no allocation
for label, arr in (("pinned", pinned.numpy().view(np.uint64)), ("pageable_numpy", host), ("pageable_malloc", mal)):
    ts = []
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        cc = LcCommit.commit(arr, enc, into=obj)
        r = cc.get_root()
        ts.append(time.perf_counter() - t0)
    roots[label] = bytes(r)
    ts = ts[1:]
    emit({"leg": "lcpc_commit(host ptr)", "source": label, "ms_min": round(min(ts) * 1e3, 2), "ms_mean": round(sum(ts) / len(ts) * 1e3, 2),
          "GBps_equiv": round(n * 8 * L / min(ts) / 1e9, 1)})
# a caller that builds a fresh Vec for every commit: pages the runtime has never seen (no lock / registration to reuse)
ts = []
for _ in range(args.reps + 1):
    fresh = np.empty_like(host)
    np.copyto(fresh, host)
    t0 = time.perf_counter()
    cc = LcCommit.commit(fresh, enc, into=obj)
    r = cc.get_root()
    ts.append(time.perf_counter() - t0)
    del fresh
roots["pageable_fresh"] = bytes(r)
emit({"leg": "lcpc_commit(host ptr)", "source": "pageable_fresh_each_call", "ms_min": round(min(ts[1:]) * 1e3, 2), "ms_mean": round(sum(ts[1:]) / args.reps * 1e3, 2),
      "GBps_equiv": round(n * 8 * L / min(ts[1:]) / 1e9, 1)})
# device-resident figure on the same box, for the ratio
ts = []
for _ in range(args.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cc = LcCommit.commit_device(dev.data_ptr(), n, enc, into=obj)
    r = cc.get_root()
    ts.append(time.perf_counter() - t0)
roots["device"] = bytes(r)
emit({"leg": "lcpc_commit_device", "source": "hbm", "ms_min": round(min(ts[1:]) * 1e3, 2)})
# what the bus and the host memory system can do: plain copies of the same 2 GiB
t = torch.empty_like(dev)
for label, src in (("pinned", pinned), ("pageable", torch.from_numpy(host.view(np.int64)))):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    emit({"leg": "plain H2D copy", "source": label, "ms_min": round(min(ts) * 1e3, 2), "GBps": round(host.nbytes / min(ts) / 1e9, 1)})
ts = []
dst = np.empty_like(host)
for _ in range(3):
    t0 = time.perf_counter()
    np.copyto(dst, host)
    ts.append(time.perf_counter() - t0)
emit({"leg": "host memcpy, 1 thread", "ms_min": round(min(ts) * 1e3, 2), "GBps": round(host.nbytes / min(ts) / 1e9, 1)})
emit({"leg": "roots", "all_equal": len(set(roots.values())) == 1})
libc.free(raw)
