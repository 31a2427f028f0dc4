#!/usr/bin/env python3
"""tools/ab_shard_slices.py -- what the column-sliced exchange of the row-sharded commit costs and what it hides, on ONE GPU.

 A. world = 1, the library's native path (lcpc_commit_sharded_device on a world-1 RCCL communicator), headline shape
    (2^26 Ft255 coefficients): LCPC_SHARD_SLICES=1 (everything in sequence on the caller's stream) against the default 4
    slices (hash of slice s+1 on the caller's stream while slice s is exchanged and finished on the commitment's second
    stream), interleaved in one process.  The sliced form must cost <= 1 %.
 B. the slowest rank (the last one) of world = 2, 4, 8 with the four-step API and a MODEL of the wire on a second stream:
    the bytes the rank receives per slice are moved by a device-to-device copy (real HBM traffic next to the hashing
    kernels) and the stream is then held (torch.cuda._sleep, one idle wave) until bytes / B seconds have passed, for a few
    assumed in-bound rates B.  Reported: the step with the wire in sequence (one slice) and pipelined (four slices), and
    the same with the whole tail (wire, leaf digests, tree) on the side stream and two commitments filled alternately (async2:
    what LCPC_COMMIT_ASYNC_TAIL does natively), and the strong-scaling ceiling each gives against this run's unsharded step.  Nobody has measured RCCL's all-gather rate on
    an 8-GPU MI355X node for this size from here; B is a parameter, not a claim.
One JSON line per measurement (profiles/r04_shard_slices.jsonl)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import lcpc_amd
from lcpc_amd import LcCommit, LigeroEncoding
from lcpc_amd.distributed import HipShardEngine, aligned_nodes, chunk_split, slice_bounds, slots_per_rank

LOG = int(os.environ.get("AB_LOG_LEN", "26"))
n_rows, npr, nc = lcpc_amd.static_get_dims(3, 0, 1 << LOG)
dev = torch.device("cuda", 0)


def rand_coeffs(rows, seed=1):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (rows * npr, 4), dtype=torch.int64, device="cuda", generator=g)
    t[:, 3] &= (1 << 62) - 1
    return t


def timeit(step, n=10, reps=5):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    return out


# ---- A: native world-1: one slice (default) | 4 slices | async tail on two alternating commitments | plain commit, interleaved ----
coeffs = rand_coeffs(n_rows)
engs = {}
for S in ("1", "4"):
    os.environ["LCPC_SHARD_SLICES"] = S
    enc = LigeroEncoding.new_from_dims(3, npr, nc, shard=(0, 1))
    os.environ.pop("LCPC_SHARD_SLICES")
    e = HipShardEngine(enc)
    e.comm_init()
    engs[S] = e
pair = [engs["1"], HipShardEngine(engs["1"].enc)]          # two commitments of the one-slice encoder, filled alternately
plain_enc = LigeroEncoding.new_from_dims(3, npr, nc)
plain = LcCommit(plain_enc)
st = torch.cuda.current_stream().cuda_stream
roots = {S: e.commit_native(coeffs, n_rows) for S, e in engs.items()}
roots["async"] = pair[1].commit_native(coeffs, n_rows, async_tail=True)
assert roots["1"] == roots["4"] == roots["async"] == LcCommit.commit_device(coeffs.data_ptr(), n_rows * npr, plain_enc, st, into=plain).get_root()
res = {"1": [], "4": [], "async2": [], "plain": []}
turn = [0]


def async_step():
    turn[0] ^= 1
    pair[turn[0]].commit_native(coeffs, n_rows, want_root=False, async_tail=True)


for rnd in range(6):
    for S in ("1", "4", "async2", "plain"):
        if S == "plain":
            f = lambda: LcCommit.commit_device(coeffs.data_ptr(), n_rows * npr, plain_enc, st, sync=False, into=plain)
        elif S == "async2":
            f = async_step
        else:
            f = (lambda e: (lambda: e.commit_native(coeffs, n_rows, want_root=False)))(engs[S])
        res[S] += timeit(f, n=10, reps=2)
mean = {k: sum(v) / len(v) for k, v in res.items()}
base_ms = mean["plain"]
print(json.dumps({"part": "A", "what": "native sharded commit, world 1, 2^%d Ft255: one slice | 4 slices | async tail on two alternating "
                  "commitments | lcpc_commit_device, interleaved" % LOG,
                  "ms_mean": {k: round(v, 3) for k, v in mean.items()}, "ms_min": {k: round(min(v), 3) for k, v in res.items()},
                  "sliced_over_unsliced": round(mean["4"] / mean["1"], 4), "unsliced_over_plain": round(mean["1"] / mean["plain"], 4),
                  "async2_over_plain": round(mean["async2"] / mean["plain"], 4)}), flush=True)
del engs, pair, plain, coeffs

# ---- B: the last rank of world = 2, 4, 8 with a modelled wire -----------------------------------------------------------------------
# _sleep calibration: cycles per millisecond of an idle spinning wave
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(50_000_000); e1.record(); torch.cuda.synchronize()
CYC_PER_MS = 50_000_000 / e0.elapsed_time(e1)

side = torch.cuda.Stream()
for world in (2, 4, 8):
    rank = world - 1
    enc = LigeroEncoding.new_from_dims(3, npr, nc, shard=(rank, world))
    eng = HipShardEngine(enc)
    rb, re, cb, ce, nch = eng.layout(n_rows)
    coeffs = rand_coeffs(re - rb)
    slots = slots_per_rank(nch, world)
    # what crosses the wire into this rank (compact layout of the native exchange): node 0 of every other rank + their extra nodes
    n_in = sum(len(aligned_nodes(b, e)) for g, (b, e) in enumerate(chunk_split(nch, world)) if g != rank)
    gathered = torch.zeros((world * slots, nc, 32), dtype=torch.uint8, device="cuda")
    staging = torch.zeros((n_in, nc, 32), dtype=torch.uint8, device="cuda")        # "the other ranks' nodes"
    main = torch.cuda.current_stream()

    def wire(c0, c1, gbps):
        """on the current stream: the slice's in-bound bytes by device copy, then idle until bytes / gbps have passed"""
        w = c1 - c0
        nbytes = n_in * w * 32
        dst = gathered.view(-1)[:nbytes]
        dst.copy_(staging.view(-1)[:nbytes], non_blocking=True)
        if gbps:
            torch.cuda._sleep(int(nbytes / (gbps * 1e9) * 1e3 * CYC_PER_MS))

    def step(S, gbps):
        eng.commit_encode(coeffs, n_rows)
        b = slice_bounds(nc, S)
        for c0, c1 in zip(b[:-1], b[1:]):
            nodes = eng.commit_hash_cols(c0, c1)
            if S == 1:
                wire(c0, c1, gbps)
                sl = gathered.view(-1)[:world * slots * (c1 - c0) * 32].view(world * slots, c1 - c0, 32)
                sl[rank * slots:rank * slots + nodes.shape[0]] = nodes
                eng.commit_finish_cols(sl, slots, c0, c1)
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    off = world * slots * c0 * 32
                    sl = gathered.view(-1)[off:off + world * slots * (c1 - c0) * 32].view(world * slots, c1 - c0, 32)
                    nbytes = n_in * (c1 - c0) * 32
                    sl.view(-1)[:nbytes].copy_(staging.view(-1)[:nbytes], non_blocking=True)
                    if gbps:
                        torch.cuda._sleep(int(nbytes / (gbps * 1e9) * 1e3 * CYC_PER_MS))
                    sl[rank * slots:rank * slots + nodes.shape[0]] = nodes
                    eng.commit_finish_cols(sl, slots, c0, c1)
                nodes.record_stream(side)
        if S > 1:
            main.wait_stream(side)
        eng.commit_merkle(want_root=False)

    # the async tail: two commitments of the encoder filled alternately; wire + leaf digests + tree on the side stream, which the
    # main stream does not wait for (a refill of the same commitment does, through the commitment's event)
    eng2 = HipShardEngine(enc)
    gathered2 = torch.zeros_like(gathered)
    flip = [0]

    def step_async(gbps):
        flip[0] ^= 1
        e, gbuf = (eng, gathered) if flip[0] else (eng2, gathered2)
        e.commit_encode(coeffs, n_rows)
        nodes = e.commit_hash_cols(0, nc)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            nbytes = n_in * nc * 32
            gbuf.view(-1)[:nbytes].copy_(staging.view(-1)[:nbytes], non_blocking=True)
            if gbps:
                torch.cuda._sleep(int(nbytes / (gbps * 1e9) * 1e3 * CYC_PER_MS))
            gbuf[rank * slots:rank * slots + nodes.shape[0]] = nodes
            e.commit_finish_cols(gbuf, slots, 0, nc)
            e.commit_merkle(want_root=False)
        nodes.record_stream(side)

    row = {"part": "B", "world": world, "rank": rank, "rows": re - rb, "chunks": ce - cb, "in_MB_per_commit": round(n_in * nc * 32 / 1e6, 1),
           "unsharded_ms": round(base_ms, 3)}
    for gbps in (0, 300, 150, 75):
        for S in (1, 4, "async2"):
            f = (lambda: step_async(gbps)) if S == "async2" else (lambda: step(S, gbps))
            t = timeit(f, n=10, reps=3)
            m = sum(t) / len(t)
            key = ("copy_only" if gbps == 0 else "%dGBps" % gbps) + {1: "_seq", 4: "_sliced4", "async2": "_async2"}[S]
            row[key + "_ms"] = round(m, 3)
            row[key + "_ceiling"] = round(base_ms / m, 2)
    print(json.dumps(row), flush=True)
    del enc, eng, eng2, coeffs, gathered, gathered2, staging
