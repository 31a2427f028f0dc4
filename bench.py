#!/usr/bin/env python3
"""bench.py -- headline benchmark: field-elements committed per second, Ligero commit of 2^26 Ft255
coefficients (rho = 1/2, BLAKE3), BASELINE.json's metric, on N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1 without WORLD_SIZE: bench.py starts its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1, what the driver does)

A "step" is one full commit (pad -> row NTTs -> column BLAKE3 -> Merkle tree, lcpc-2d/src/lib.rs:622-671) of
one synthetic coefficient vector that is already resident in HBM when the timed region starts.  For N > 1 the
512 rows of the SAME 2^26 commitment are sharded by BLAKE3-chunk-aligned row blocks across the ranks, with one
RCCL exchange of subtree chaining values inside the library (strong scaling; `--scaling weak` keeps 512 rows per GPU).
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.

Inputs (SURVEY.md 8d "Synthetic inputs"): the job's coefficient vector is Field::random over ChaCha20Rng::from_seed([0; 32]), stream
0, drawn ON THE DEVICE (lcpc_random_coeffs_device); the cpu_baseline leg draws the same vector on the host with the oracle's
generator, checks the two element for element, and commits it: `root_equals_hip_root` speaks about the timed data itself.

Timing protocol (mirroring rough_bench, lcpc-ligero-pc/src/tests.rs:78-98): the encoder is built outside
the timed region; W warm-up steps; K steps between barrier + synchronize brackets give `value` / `ms_per_step` (mean);
HIP events between the same steps give `min_ms_per_step`; afterwards, untimed: one instrumented step (kernel-group
events), the end-to-end figure from host memory (`e2e_host`, PCIe-inclusive, never `value`) and the CPU baseline.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def kernel_sources():
    """everything that decides WHAT a commit launches and HOW: every device source and header of the library, plus the host
    units that plan the passes, order the tiles and sequence the launches (ctx.cpp, commit.cpp, shard.cpp)"""
    d = os.path.join(ROOT, "lcpc_amd", "csrc")
    return sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h")) or n in ("ctx.cpp", "commit.cpp", "shard.cpp"))


def kernel_stamp():
    """sha256 over kernel_sources(): a committed PMC file is only quoted while it describes this build."""
    h = hashlib.sha256()
    for n in kernel_sources():
        h.update(n.encode())
        with open(os.path.join(ROOT, "lcpc_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


INPUT_SEED = 0              # ChaCha20Rng::from_seed([INPUT_SEED; 32]), stream 0


def job_coeffs(enc, n):
    """the job's coefficient vector on the encoder's device: n x Field::random from the fixed generator (module docstring)"""
    return enc.random_coeffs_device(n, seed=INPUT_SEED)


def sample_power(step, torch, seconds=1.6):
    """rocm-smi shader clock (MHz) and socket power (W), sampled while `step` loops on another thread; None if rocm-smi
    is unavailable or prints something else."""
    import re
    import subprocess
    import threading
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            for _ in range(8):
                step()
            torch.cuda.synchronize()

    th = threading.Thread(target=loop)
    th.start()
    clk, pw = [], []
    try:
        time.sleep(0.5)
        t_end = time.time() + seconds
        while time.time() < t_end:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", txt)
            w = re.search(r"Power \(W\):\s*([0-9.]+)", txt)
            if m and w:
                clk.append(int(m.group(1)))
                pw.append(float(w.group(1)))
    except Exception:
        pass
    finally:
        stop.set()
        th.join()
    if not clk:
        return None
    return {"sclk_MHz": round(sum(clk) / len(clk)), "socket_W": round(sum(pw) / len(pw)), "samples": len(clk),
            "source": "rocm-smi while the commit loops (after the timed region)"}


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU boxes show all
    256 hardware threads but grant 16 CPUs of time; oversubscribing 256 OpenMP threads into that quota is slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]              # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def host_memory_available():
    """bytes this process may still allocate: MemAvailable capped by the cgroup limit (a box that dies of OOM is a strike)."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, int(lim) - cur) if avail is not None else int(lim) - cur
    except Exception:
        pass
    return avail


def cpu_commit(O, coeffs, n_per_row, n_cols, threads, keep=False):
    """one oracle commit (C port of the reference algorithm, OpenMP over rows / 32-column blocks) of `coeffs` (Ft255 Montgomery
    limbs on the host) with the headline row shape; returns (rate, seconds, root[, the oracle's encoding and commitment])."""
    enc = O.Encoding.ligero_from_dims(3, n_per_row, n_cols)
    t0 = time.perf_counter()
    c = O.Commit.commit(coeffs, enc, n_threads=threads)
    dt = time.perf_counter() - t0
    root = c.get_root()
    if keep:
        return len(coeffs) / dt, dt, root, enc, c
    del c
    return len(coeffs) / dt, dt, root


C5_X = 0x123456789abcdef        # the evaluation point of the C5 leg (outer = powers of x^n_per_row, inner = powers of x: ligero tests.rs:120-128)


FT255_P = 0x663c799b6e4d2900fda9df04b9575969ef73c79086595f3002a4f20000000001      # lcpc-test-fields/src/lib.rs:50-58


def powers_mont(x, n, step=1):
    """[x^(k step)] for k < n in Ft255 as ff_derive's Montgomery limbs (value * 2^256 mod p, 4 little-endian u64), shape (n, 4)"""
    import numpy as np
    base, cur, out = pow(x, step, FT255_P), 1, np.zeros((n, 4), np.uint64)
    for k in range(n):
        m = (cur << 256) % FT255_P
        out[k] = [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
        cur = cur * base % FT255_P
    return out


def mk_transcript(T, root, n_col_opens):
    """the reference's test transcript (lcpc-ligero-pc/src/tests.rs:243-245)"""
    tr = T(b"test transcript")
    tr.append_message(b"polycommit", bytes(root))
    tr.append_message(b"ncols", int(n_col_opens).to_bytes(8, "big"))
    return tr


def commit_bytes_8d(F, n_rows, n_per_row, n_cols):
    """SURVEY.md 8(d) B_commit: read coeffs + write comm + read comm for the column hash + write / re-read the digests"""
    np2 = 1
    while np2 < n_cols:
        np2 *= 2
    return F * n_rows * n_per_row + 2 * F * n_rows * n_cols + 32 * (2 * np2 - 1) + 32 * (2 * np2 - 2)


def time_commits(torch, LcCommit, enc, dev_coeffs, n, stream, reps=12):
    """>= 10 back-to-back commits of a device-resident vector into ONE LcCommit object, a HIP event after each on the launch stream:
    (mean ms, min ms, reps, the object -- holding the last commit)"""
    c = LcCommit(enc)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:           # clocks up, buffers allocated
        for _ in range(3):
            LcCommit.commit_device(dev_coeffs.data_ptr(), n, enc, stream, sync=False, into=c)
        torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        LcCommit.commit_device(dev_coeffs.data_ptr(), n, enc, stream, sync=False, into=c)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
    return sum(ms) / reps, min(ms), reps, c


def hip_configs(torch, lcpc_amd, coeffs, n_total, enc, cm, step, stream, device):
    """BASELINE.json's other configs on this GPU, untimed region of an N = 1 run: C1, C2, C3 commits and C5's prove /
    verify, each >= 10 repetitions with mean and min, on inputs that the cpu_baseline leg re-commits / re-proves with the oracle
    (`checked` stays false until it has).  C2 / C3 commit the first 2^24 elements of the job's own vector, C5 proves the job's commitment.
    Returns (public dict, private dict of the HIP results the oracle leg compares)."""
    import numpy as np
    from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding, Transcript
    pub, priv = {}, {}
    # ---- C1: lcpc-ligero-pc commit over ft63, 2^16 coeffs
    n1 = 1 << 16
    e1 = LigeroEncoding.new(lcpc_amd.FT63, n1, device=device)
    x1 = e1.random_coeffs_device(n1, seed=INPUT_SEED)
    mean, mn, reps, c1 = time_commits(torch, LcCommit, e1, x1, n1, stream, reps=40)
    b = commit_bytes_8d(8, c1.n_rows, c1.n_per_row, c1.n_cols)
    pub["C1"] = {"workload": "lcpc-ligero-pc commit, Ft63, 2^16 coeffs, rho=1/2, BLAKE3", "dims": [c1.n_rows, c1.n_per_row, c1.n_cols], "ms": round(mean, 4),
                 "min_ms": round(mn, 4), "reps": reps, "algorithmic_GBps": round(b / (mean * 1e-3) / 1e9, 1), "checked": False}
    priv["C1"] = (c1.get_root(), x1.cpu().numpy())
    del c1, e1, x1
    # ---- C2: lcpc-ligero-pc commit, 2^24 coeffs, Ft255
    n2 = 1 << 24
    if n_total >= n2:
        e2 = LigeroEncoding.new(lcpc_amd.FT255, n2, device=device)
        mean, mn, reps, c2 = time_commits(torch, LcCommit, e2, coeffs, n2, stream)
        b = commit_bytes_8d(32, c2.n_rows, c2.n_per_row, c2.n_cols)
        pub["C2"] = {"workload": "lcpc-ligero-pc commit, Ft255, 2^24 coeffs, rho=1/2, BLAKE3", "dims": [c2.n_rows, c2.n_per_row, c2.n_cols], "ms": round(mean, 4),
                     "min_ms": round(mn, 4), "reps": reps, "algorithmic_GBps": round(b / (mean * 1e-3) / 1e9, 1), "checked": False,
                     "input": "the first 2^24 elements of the job's vector"}
        priv["C2"] = (c2.get_root(), (c2.n_per_row, c2.n_cols))
        del c2, e2
        # ---- C3: lcpc-brakedown-pc commit, 2^24 coeffs, Ft255 (SdigCode3, seed 0); the encoder is built outside the timed commits
        t0 = time.perf_counter()
        e3 = SdigEncoding.new(lcpc_amd.FT255, n2, 0, device=device)
        t_build = time.perf_counter() - t0
        mean, mn, reps, c3 = time_commits(torch, LcCommit, e3, coeffs, n2, stream)
        b = commit_bytes_8d(32, c3.n_rows, c3.n_per_row, c3.n_cols)
        pub["C3"] = {"workload": "lcpc-brakedown-pc commit, Ft255, 2^24 coeffs, SdigCode3 seed 0, BLAKE3", "dims": [c3.n_rows, c3.n_per_row, c3.n_cols],
                     "ms": round(mean, 4), "min_ms": round(mn, 4), "reps": reps, "algorithmic_GBps": round(b / (mean * 1e-3) / 1e9, 1), "checked": False,
                     "encoder_build_s": round(t_build, 3), "input": "the first 2^24 elements of the job's vector"}
        priv["C3"] = (c3.get_root(), (c3.n_per_row, c3.n_cols))
        del c3, e3
    torch.cuda.empty_cache()
    # ---- C5: prove + verify on the job's own commitment (the headline's 2^26): eval_outer random linear combination + column openings
    root = step(sync=True).get_root()
    nr, npr = cm.n_rows, cm.n_per_row
    inner = powers_mont(C5_X, npr)
    outer = powers_mont(C5_X, nr, npr)
    n_open = enc.get_n_col_opens()
    tp, tv, pf = [], [], None
    for rep in range(11):                        # rep 0 = first use (allocations, helper threads): not counted
        step(sync=True)                          # prove follows commit in the reference's flow (tests.rs:243-262): the GPU is at working clocks
        t0 = time.perf_counter()
        pf = cm.prove(outer, enc, mk_transcript(Transcript, root, n_open))
        t1 = time.perf_counter()
        ev = pf.verify(root, outer, inner, enc, mk_transcript(Transcript, root, n_open))
        t2 = time.perf_counter()
        if rep:
            tp.append((t1 - t0) * 1e3)
            tv.append((t2 - t1) * 1e3)
    b = 32 * nr * npr + n_open * nr * 32
    data = pf.to_bytes()
    pub["C5"] = {"workload": "lcpc-2d prove + verify, Ft255, 2^%d coeffs (the job's commitment), %d column openings" % (n_total.bit_length() - 1, n_open),
                 "prove": {"ms": round(sum(tp) / len(tp), 3), "min_ms": round(min(tp), 3), "reps": len(tp), "algorithmic_GBps": round(b / (sum(tp) / len(tp) * 1e-3) / 1e9, 1),
                           "note": "wall time of LcCommit::prove incl. the host transcript (2 x n_per_row serial STROBE absorbs, lcpc-2d/src/lib.rs:1045-1047)"},
                 "verify": {"ms": round(sum(tv) / len(tv), 3), "min_ms": round(min(tv), 3), "reps": len(tv)},
                 "proof_bytes": len(data), "checked": False}
    priv["C5"] = (data, root, outer, inner, np.asarray(ev))
    return pub, priv


def check_configs(O, pub, priv, cpu_coeffs, threads, oenc26, oc26):
    """the oracle side of hip_configs: roots of C1-C3 on the same inputs, C5's proof bytes == the oracle PROVER's on the oracle's
    commitment of the timed vector and the oracle VERIFIER accepts the HIP proof with the same evaluation.  Any difference ends the run."""
    import numpy as np
    if "C1" in priv:
        root, x = priv["C1"]
        want = O.random_elems(0, 1 << 16, INPUT_SEED)
        if not (x.view(np.uint64).reshape(want.shape) == want).all():
            raise SystemExit("bench.py configs: C1's device-drawn Ft63 coefficients differ from the oracle's stream")
        if O.Commit.commit(want, O.Encoding.ligero(0, 1 << 16)).get_root() != root:
            raise SystemExit("bench.py configs: C1 HIP root != oracle root")
        pub["C1"]["checked"] = True
    if "C2" in priv and len(cpu_coeffs) >= (1 << 24):
        root, (npr, nc) = priv["C2"]
        if O.Commit.commit(cpu_coeffs[:1 << 24], O.Encoding.ligero_from_dims(3, npr, nc), n_threads=threads).get_root() != root:
            raise SystemExit("bench.py configs: C2 HIP root != oracle root")
        pub["C2"]["checked"] = True
    if "C3" in priv and len(cpu_coeffs) >= (1 << 24):
        root, (npr, nc) = priv["C3"]
        if O.Commit.commit(cpu_coeffs[:1 << 24], O.Encoding.sdig_from_dims(3, npr, nc, 0, 3), n_threads=threads).get_root() != root:
            raise SystemExit("bench.py configs: C3 HIP root != oracle root")
        pub["C3"]["checked"] = True
    if "C5" in priv and oc26 is not None:
        data, root, outer, inner, ev = priv["C5"]
        if oc26.get_root() != root:
            raise SystemExit("bench.py configs: C5 the oracle's commitment has another root")
        n_open = oenc26.get_n_col_opens()
        opf, _ = oc26.prove(outer, oenc26, mk_transcript(O.Transcript, root, n_open))
        if opf != data:
            raise SystemExit("bench.py configs: C5 HIP proof bytes != the oracle prover's")
        rc, oev = O.verify(oenc26, root, outer, inner, data, mk_transcript(O.Transcript, root, n_open))
        if rc != 0 or not (np.asarray(oev) == ev).all():
            raise SystemExit("bench.py configs: C5 the oracle verifier rejects the HIP proof (rc %d) or returns another evaluation" % rc)
        pub["C5"]["checked"] = True
    for k, v in pub.items():
        v["check"] = ("root == the oracle's root of the same coefficients" if k != "C5" else
                      "proof bytes == the oracle prover's on its own commitment of the timed vector; oracle verifier accepts with the same evaluation") \
            if v["checked"] else "NOT checked in this run (no oracle leg, or the CPU sample was shorter than the input)"


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) started without a launcher: become the launcher.  One rank per GPU under
    torch.distributed.run on 127.0.0.1, the same command line; rank 0's JSON line goes to this process's stdout.  Fewer
    than N visible devices is an error (never a silent 1-GPU run), unless the debug switches that let ranks share a
    device (--dist-backend gloo / --force-device) are given."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared_ok = args.dist_backend != "nccl" or args.force_device is not None
    if have < args.gpus and not (shared_ok and have >= 1):
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible -- refusing to run a smaller job under that label" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench.py] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank job failed (exit code %d)" % (args.gpus, rc))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-len", type=int, default=26)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--borrow-coeffs", action="store_true", help="LcCommit.coeffs aliases the caller's HBM buffer (LCPC_COMMIT_BORROW_COEFFS) "
                    "instead of the private copy LcCommit::commit makes (lcpc-2d/src/lib.rs:636-645), which is the default and what `value` times")
    ap.add_argument("--copy-coeffs", action="store_true", help="(the default since round 3; kept so that older command lines still parse)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-sample", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the untimed `configs` legs (C1, C2, C3, C5 of BASELINE.json, N = 1 only)")
    ap.add_argument("--lean", action="store_true", help="the timed loop and the instrumented step only (what tools/profile_round.sh wraps in "
                    "rocprofv3, so that per-kernel averages are not diluted by the secondary legs)")
    ap.add_argument("--cpu-sample-log-len", type=int, default=None, help="default: the full 2^log-len if host memory allows, else one less")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native",
                    help="N > 1: native = RCCL inside the library (lcpc_commit_sharded_device); torch = torch.distributed all-gather "
                         "between the two library phases")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) | gloo (debug: ranks may share a GPU; implies --exchange torch)")
    ap.add_argument("--force-device", type=int, default=None, help="debug: every rank uses this HIP device")
    ap.add_argument("--no-async-tail", action="store_true", help="N > 1, native exchange: one LcCommit object, exchange + leaf digests + tree in "
                    "sequence on the launch stream (default: two objects filled alternately with LCPC_COMMIT_ASYNC_TAIL, so that commit k's "
                    "exchange overlaps commit k + 1's encode)")
    ap.add_argument("--check", action="store_true", help="(always on for N > 1 since round 3; kept so that older command lines still parse)")
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the untimed comparison of the sharded root with an unsharded commit of the same data")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args)

    if args.lean:
        args.no_cpu_baseline = args.no_power_sample = args.no_e2e = args.no_configs = True
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    import torch
    import torch.distributed as dist
    import lcpc_amd
    from lcpc_amd import _lib as _lcpc_lib
    if not os.path.exists(_lcpc_lib.LIB_PATH) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        _lcpc_lib.build()                       # a checkout without the prebuilt HIP library: build it (there is no other path)
    from lcpc_amd import LcCommit, LigeroEncoding
    from lcpc_amd.distributed import HipShardEngine, sharded_commit

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:                         # never report a job of another size than the one asked for
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the lcpc HIP path has no CPU fallback")
    if args.force_device is not None:
        local_rank = args.force_device
    elif args.dist_backend != "nccl":              # debug runs: more ranks than GPUs share the devices
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py --gpus %d: rank %d has no device of its own (%d visible); one GPU per rank, or the debug switches "
                         "--dist-backend gloo / --force-device" % (args.gpus, rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if args.dist_backend != "nccl" and not (args.exchange == "native" and os.environ.get("LCPC_RCCL_LIB")):
        args.exchange = "torch"          # (debug runs over gloo keep the native exchange only with an explicit stand-in library)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

    fid, L, F = lcpc_amd.FT255, 4, 32
    n_total = 1 << args.log_len
    n_rows1, n_per_row, n_cols = lcpc_amd.static_get_dims(fid, lcpc_amd.ENC_LIGERO, n_total)
    n_rows_total = n_rows1 * world if args.scaling == "weak" else n_rows1
    n_coeffs_job = n_rows_total * n_per_row
    borrow = args.borrow_coeffs
    stream = torch.cuda.current_stream().cuda_stream

    if not distributed:
        enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, device=local_rank)
        coeffs = job_coeffs(enc, n_coeffs_job)
        cm = LcCommit(enc)                     # ONE LcCommit object refilled every step (buffers reused; no hipMalloc in the loop)

        def step(sync=False, borrow_=borrow):
            return LcCommit.commit_device(coeffs.data_ptr(), n_coeffs_job, enc, stream, sync=sync, borrow=borrow_, into=cm)
    else:
        enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, device=local_rank, shard=(rank, world))
        engine = HipShardEngine(enc)
        cm = engine.cm
        rb, re, cb, ce, n_chunks = engine.layout(n_rows_total)
        # every rank draws the job's whole vector (one serial stream: element i is the i-th accepted candidate) and keeps its rows
        full = job_coeffs(enc, n_coeffs_job)
        coeffs = full[rb * n_per_row:max(re, rb + 1) * n_per_row].clone() if re > rb else full[:n_per_row].clone()
        if args.no_check:
            del full
        # bring the communicators up outside the measured region (RCCL connects lazily on the first collective)
        t_init = torch.ones(1, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t_init)
        torch.cuda.synchronize()
        ranks_seen = int(t_init.item())
        if ranks_seen != args.gpus:
            raise SystemExit("bench.py --gpus %d: the all-reduce saw %d ranks" % (args.gpus, ranks_seen))
        exchange_note = None
        if args.exchange == "native":
            # the library's own RCCL exchange; cross-checked once against the torch.distributed exchange of the same
            # shard phases (outside the timed region) -- any disagreement or failure falls back to the latter, and says so
            ok = 1
            try:
                engine.comm_init()
                r_native = engine.commit_native(coeffs, n_rows_total, want_root=True, borrow=borrow)
                r_torch = sharded_commit(engine, coeffs, n_rows_total, want_root=True, borrow=borrow)
                if r_native != r_torch:
                    ok, exchange_note = 0, "native root != torch.distributed root"
            except Exception as ex:
                ok, exchange_note = 0, "native exchange failed: %r" % (ex,)
            t_ok = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            if int(t_ok.item()) == 0:
                args.exchange = "torch"
                exchange_note = exchange_note or "another rank reported a native-exchange problem"
                print("[rank %d] falling back to the torch.distributed exchange: %s" % (rank, exchange_note), file=sys.stderr)
        if args.exchange == "native" and not args.no_async_tail:
            # LCPC_COMMIT_ASYNC_TAIL on two LcCommit objects of the encoder, filled alternately: commit k's exchange, leaf digests and
            # tree run on its own stream while commit k + 1 encodes on the launch stream -- how a prover committing a batch of
            # polynomials drives the library.  Every commit of the timed region is complete at its closing synchronize.
            engines = [engine, HipShardEngine(enc)]
            turn = [0]

            def step(sync=False, borrow_=borrow):
                if sync:                        # the checked / instrumented steps: the first object, in sequence on the launch stream
                    return engine.commit_native(coeffs, n_rows_total, want_root=True, borrow=borrow_)
                turn[0] ^= 1
                return engines[turn[0]].commit_native(coeffs, n_rows_total, want_root=False, borrow=borrow_, async_tail=True)
        elif args.exchange == "native":
            def step(sync=False, borrow_=borrow):
                return engine.commit_native(coeffs, n_rows_total, want_root=sync, borrow=borrow_)
        else:
            def step(sync=False, borrow_=borrow):
                return sharded_commit(engine, coeffs, n_rows_total, want_root=sync, borrow=borrow_)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    check = None
    if distributed and not args.no_check:
        # first step of every N > 1 run (untimed): every rank compares the sharded root with a plain single-context commit of
        # the job's whole vector
        root_sharded = step(sync=True)
        ref = LcCommit.commit_device(full.data_ptr(), n_coeffs_job, LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, device=local_rank), stream)
        ok = ref.get_root() == root_sharded
        print("[rank %d] sharded root %s unsharded root: %s" % (rank, "==" if ok else "!=", root_sharded.hex()), file=sys.stderr)
        t_ok = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if int(t_ok.item()) == 0:
            raise SystemExit("sharded commit root mismatch")
        check = {"sharded_root_equals_unsharded_root": True, "on": "all %d ranks, the job's own coefficients, before the timed region" % world,
                 "root": root_sharded.hex()}
        del full, ref
        torch.cuda.empty_cache()

    # setup, not warm-up steps: a device that sat idle while the inputs were generated needs tens of milliseconds to
    # come back to working clocks (measured: a 0.6 ms kernel takes 8-20 ms right after an idle period)
    for _ in range(4):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]    # on the launch stream (torch's current stream)
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_coeffs_job * args.steps / dt
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    min_ms = min(step_ms)
    async_loop = distributed and args.exchange == "native" and not args.no_async_tail
    if async_loop:
        min_ms = None          # the launch-stream events bracket only a part of an async-tail step: no per-step minimum is claimed
    elif distributed:
        t = torch.tensor([min_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        min_ms = float(t.item())

    # N > 1, native exchange, untimed region: what ONE run must tell about scaling
    red_dev = dev if args.dist_backend == "nccl" else "cpu"

    def over_ranks(vals, op):
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return [float(x) for x in t]

    scaling_extra = None
    if distributed and args.exchange == "native":
        scaling_extra = {}
        if async_loop:
            # (a) the roots the async-tail commits of the timed loop left in BOTH objects (they were never looked at)
            want = check["root"] if check else engines[0].cm.get_root().hex()
            roots_ok = all(e.cm.get_root().hex() == want for e in engines)
            if over_ranks([1.0 if roots_ok else 0.0], dist.ReduceOp.MIN)[0] == 0.0:
                raise SystemExit("bench.py: an async-tail commit of the timed loop left a wrong root")
            scaling_extra["async_tail_roots_checked"] = "both LcCommit objects of every rank after the timed loop == " + \
                                                       ("the checked sharded root" if check else "each other")
        # (b) the serial figure: ONE object, exchange + leaf digests + tree in sequence on the launch stream (what --no-async-tail times;
        #     comparable with the per-commit numbers of rounds 1-3)
        def serial_step():
            return engine.commit_native(coeffs, n_rows_total, want_root=False, borrow=borrow)
        for _ in range(2):
            serial_step()
        fence()
        sev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t1 = time.perf_counter()
        sev[0].record()
        for i in range(args.steps):
            serial_step()
            sev[i + 1].record()
        fence()
        sdt = time.perf_counter() - t1
        s_ms = [sev[i].elapsed_time(sev[i + 1]) for i in range(args.steps)]
        sdt, s_min = over_ranks([sdt, min(s_ms)], dist.ReduceOp.MAX)
        scaling_extra["serial_ms_per_step"] = round(sdt / args.steps * 1e3, 4)
        scaling_extra["serial_min_ms_per_step"] = round(s_min, 4)
        scaling_extra["serial_value"] = n_coeffs_job * args.steps / sdt
        # (c) the wire alone: 20 x the library's own exchange on the real payload (lcpc_shard_exchange_probe)
        for _ in range(3):
            engine.exchange_probe()
        fence()
        pe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        pe[0].record()
        for _ in range(20):
            b_in = engine.exchange_probe()
        pe[1].record()
        fence()
        p_ms, b_max = over_ranks([pe[0].elapsed_time(pe[1]) / 20, float(b_in)], dist.ReduceOp.MAX)
        scaling_extra["exchange_probe"] = {"bytes_in": int(b_max), "ms": round(p_ms, 4), "GBps_in": round(b_max / (p_ms * 1e-3) / 1e9, 2) if p_ms > 0 else None,
                                           "reps": 20, "what": "ncclAllGather of node 0 of every rank + grouped ncclBroadcasts of the extra nodes, exactly as the "
                                                                "commit enqueues them, back to back on the launch stream; bytes_in = received from other ranks "
                                                                "per exchange, MAX over ranks; ms = MAX over ranks"}
        scaling_extra["rccl_version"] = HipShardEngine.rccl_version()

    # kernel-group timing with HIP events on the launch stream (one extra, untimed, instrumented step)
    cm.set_timing(True)
    step(sync=True)
    tm = cm.timings()
    cm.set_timing(False)
    rows_local = n_rows_total if not distributed else (re - rb)
    enc_bytes = F * rows_local * (n_per_row + n_cols)                 # read coeffs + write comm
    np2 = n_cols
    commit_bytes = F * rows_local * n_per_row + 2 * F * rows_local * n_cols + 32 * (4 * np2 - 3)
    ntt_launches = max(1, tm.encode_launches)
    ntt_ms = tm.encode_ms / ntt_launches
    achieved = (enc_bytes / ntt_launches) / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
    traffic, traffic_src, valu_issue = None, None, None
    stamp = kernel_stamp()
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path) and world == 1 and args.log_len == 26:
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
        # (tools/profile_round.sh); quoted only while the file was taken from exactly these kernel sources
        pmc = json.load(open(pmc_path))
        if pmc.get("kernel_stamp") == stamp and pmc.get("borrow_coeffs", False) == borrow:
            per = [(2.0 * v["FETCH_KB"] + v["WRITE_KB"]) * 1024 / 1e9 for kname, v in pmc["kernels"].items() if kname.startswith("ntt_pass")]
            insts = [v["SQ_INSTS_VALU"] for kname, v in pmc["kernels"].items() if kname.startswith("ntt_pass") and "SQ_INSTS_VALU" in v]
            if insts and ntt_ms > 0:
                # what actually binds the kernel: vector-ALU instruction issue.  Peak = one mad-class (VOP3 / 64-bit)
                # wave instruction per SIMD every 4 cycles, 256 CUs x 4 SIMDs at the 2.4 GHz peak clock; measured
                # (profiles/r02_ubench_valu.txt): v_mad_i64_i32 4.7 cycles, plain VOP2 integer ops 2.4; the board
                # sustains ~2.1 GHz at its power limit while this kernel runs
                per_s = sum(insts) / len(insts) / (ntt_ms * 1e-3)
                valu_issue = {"wave_insts_per_launch": round(sum(insts) / len(insts)), "achieved_Ginst_per_s": round(per_s / 1e9, 1),
                              "peak_Ginst_per_s": 614.4, "frac": round(per_s / 614.4e9, 3),
                              "source": "SQ_INSTS_VALU, profiles/pmc_latest.json (same stamp)"}
                # the ceiling of THIS instruction mix (tools/isa_mix.py: disassembly classes x their measured issue cost), at the
                # clock the board holds while the commit loops (filled in below, once rocm-smi has been sampled)
                mix_path = os.path.join(ROOT, "profiles", "valu_mix_latest.json")
                if os.path.exists(mix_path):
                    mix = json.load(open(mix_path))
                    if mix.get("kernel_stamp") == stamp:
                        valu_issue["mix"] = {"mean_cost_cycles": mix["mean_cost_cycles"], "ubench_clock_GHz": mix["ubench_clock_GHz"],
                                             "by_class": {k: v["by_class"] for k, v in mix["kernels"].items()},
                                             "source": "profiles/valu_mix_latest.json (tools/isa_mix.py, same stamp): static class counts x profiles/r06_ubench_mix.txt"}
                        valu_issue["weighted_peak_Ginst_per_s"] = mix["weighted_peak_Ginst_per_s_at_ubench_clock"]
                        valu_issue["frac_weighted"] = round(per_s / 1e9 / mix["weighted_peak_Ginst_per_s_at_ubench_clock"], 3)
                        valu_issue["weighted_at"] = "the microbenchmark's clock (%.2f GHz)" % mix["ubench_clock_GHz"]
            if per:         # the NTT passes are separate kernel instantiations, one launch each per commit: mean per launch
                traffic = round(sum(per) / len(per), 3)
                traffic_src = "profiles/pmc_latest.json (rocprofv3 --pmc, kernel stamp %s; GB per launch = 2*FETCH_SIZE + WRITE_SIZE, mean over the %d NTT passes)" % (stamp, len(per))
        else:
            traffic_src = "profiles/pmc_latest.json is stale for these kernels (stamp %s != %s): not quoted" % (pmc.get("kernel_stamp"), stamp)
    roofline = {"bound": "valu_issue", "bound_note": "what limits the kernel is vector-ALU instruction issue (`valu_issue` below, DESIGN.md section 6); "
                "achieved / peak / frac are the HBM figures the contract asks for (algorithmic bytes per launch / launch time against 8 TB/s)",
                "kernel": "ntt_pass_l9s_kernel (row NTT, Ft255 signed lazy-limb arithmetic, shape-specialised; %d pass launches per commit)" % ntt_launches,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_unit": "GB/launch", "traffic_source": traffic_src,
                "algorithmic_GB_per_launch": round(enc_bytes / ntt_launches / 1e9, 3),
                "avg_launch_ms": round(ntt_ms, 4), "valu_issue": valu_issue,
                "note": "255-bit modular multiply: integer-VALU bound (DESIGN.md section 6), not HBM-bound",
                "commit_GBps": round(commit_bytes / (tm.total_ms * 1e-3) / 1e9, 1) if tm.total_ms > 0 else None,
                "power": None,
                "group_ms": {"encode": round(tm.encode_ms, 3), "hash": round(tm.hash_ms, 3), "merkle": round(tm.merkle_ms, 3),
                             "total": round(tm.total_ms, 3)}}

    # shader clock and socket power while the same step loops (N = 1 only; ~1.5 s, outside the timed region)
    power = None
    if not distributed and not args.no_power_sample:
        power = sample_power(step, torch)

    # secondary figures, N = 1, untimed region: the other LcCommit.coeffs mode and the end-to-end commit from host memory
    other_mode, e2e = None, None
    if not distributed and not args.lean:
        for _ in range(2):
            step(borrow_=not borrow)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            step(borrow_=not borrow)
        torch.cuda.synchronize()
        other_mode = {"coeffs": "borrowed (LCPC_COMMIT_BORROW_COEFFS: LcCommit.coeffs aliases the caller's HBM buffer)" if not borrow
                                else "copied (the reference's LcCommit owns its coeffs)",
                      "ms_per_step": round((time.perf_counter() - t1) / 5 * 1e3, 3)}
        step()
        torch.cuda.synchronize()
        if not args.no_e2e:
            try:
                import numpy as np
                host = torch.empty((n_coeffs_job, L), dtype=torch.int64).pin_memory()
                host.copy_(coeffs)
                pageable = np.empty((n_coeffs_job, L), dtype=np.int64)      # plain malloc / mmap memory: what a Rust Vec<F> is
                np.copyto(pageable, host.numpy())
                cm2 = LcCommit(enc)
                root_buf = (C.c_uint8 * 32)()
                lib = _lcpc_lib.lib()
                root_dev = step(sync=True).get_root()

                def host_leg(ptr):
                    ts = []
                    for _ in range(4):
                        t1 = time.perf_counter()
                        cm2._check(lib.lcpc_commit(cm2._h, C.c_void_p(ptr), n_coeffs_job, root_buf))
                        ts.append(time.perf_counter() - t1)
                    ts = ts[1:]
                    return {"ms_mean": round(sum(ts) / len(ts) * 1e3, 2), "ms_min": round(min(ts) * 1e3, 2), "value": n_coeffs_job / (sum(ts) / len(ts)),
                            "root_matches_device_commit": bytes(root_buf) == root_dev, "staged_slices": cm2.timings().staged_slices}
                pin = host_leg(host.data_ptr())
                pag = host_leg(pageable.ctypes.data)
                pag["vs_pinned"] = round(pag["ms_mean"] / pin["ms_mean"], 3)
                e2e = dict(pin)
                e2e.update({"unit": "field-elements/s", "pageable": pag,
                            "what": "lcpc_commit (== LcCommit::commit(&coeffs, &enc), lcpc-2d/src/lib.rs:299-301) from host memory: H2D of the "
                                    "coefficients (row batches overlapped with the NTTs) + commit + root D2H; PCIe-inclusive, reported beside "
                                    "`value`, never as it.  Top level: from pinned memory (direct async copies).  `pageable`: from a plain "
                                    "malloc'ed buffer -- what a Rust caller's Vec is -- staged through the library's pinned bounce ring by the "
                                    "host pool (staged_slices copies); profiles/r05_host_path.jsonl has the A/B against the runtime's own path"})
                del cm2, host, pageable
            except Exception as ex:          # e.g. not enough pinnable host memory: the headline does not depend on it
                e2e = {"error": repr(ex)}

    shard_ms = None
    if distributed and args.exchange == "torch":
        # where a sharded step spends its time on this rank (one extra, untimed, phase-synchronised step): local encode +
        # node CVs, the all-gather, leaf finish + Merkle tree; MAX over ranks
        from lcpc_amd.distributed import exchange_nodes
        fence()
        t_a = time.perf_counter()
        nodes = engine.commit_shard(coeffs, n_rows_total)
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        gathered, slots = exchange_nodes(nodes, n_chunks)
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        engine.commit_finish(gathered, n_rows_total, slots, want_root=False)
        torch.cuda.synchronize()
        t_d = time.perf_counter()
        ph = torch.tensor([t_b - t_a, t_c - t_b, t_d - t_c], dtype=torch.float64, device=dev)
        dist.all_reduce(ph, op=dist.ReduceOp.MAX)
        shard_ms = {"local_encode_hash": round(float(ph[0]) * 1e3, 3), "all_gather": round(float(ph[1]) * 1e3, 3),
                    "finish": round(float(ph[2]) * 1e3, 3), "gathered_MB": round(gathered.numel() / 1e6, 1)}
    elif distributed:
        # native exchange: the library's own event brackets (encode | hash | exchange + finish), MAX over ranks
        # exchange_exposed_ms: from the end of the local column hash to the arrival of the leaf digests (wire + leaf-digest time)
        ph = torch.tensor([tm.encode_ms, tm.hash_ms, tm.merkle_ms, tm.exchange_exposed_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(ph, op=dist.ReduceOp.MAX)
        # (d) per rank: min / max over ranks of each phase of the instrumented commit
        mine = [tm.encode_ms, tm.hash_ms, tm.exchange_wire_ms, max(0.0, tm.merkle_ms - tm.exchange_wire_ms)]
        lo, hi = over_ranks(mine, dist.ReduceOp.MIN), over_ranks(mine, dist.ReduceOp.MAX)
        scaling_extra["per_rank_ms"] = {k: [round(lo[i], 3), round(hi[i], 3)] for i, k in enumerate(("encode", "hash", "exchange_exposed", "finish_tree"))}
        scaling_extra["per_rank_ms"]["what"] = "[min, max] over ranks, one instrumented commit in sequence: exchange_exposed = end of the local hash -> " \
                                               "end of the collectives on the stream; finish_tree = leaf digests + Merkle tree"
        shard_ms = {"local_encode": round(float(ph[0]), 3), "local_hash": round(float(ph[1]), 3),
                    "exchange_tail_plus_merkle": round(float(ph[2]), 3), "exchange_exposed_ms": round(float(ph[3]), 3),
                    "async_tail": not args.no_async_tail,
                    "note": "one instrumented commit in sequence on the launch stream (MAX over ranks): exchange_exposed_ms is the time between "
                            "the end of the local hash and the arrival of the leaf digests, i.e. the wire + leaf-digest time a lone commit "
                            "pays; in the timed loop (async_tail) it overlaps the next commit's encode"}

    out = {"metric": "field-elements committed/sec (whole node), Ligero 2^%d coeffs" % args.log_len,
           "value": value, "unit": "field-elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "min_ms_per_step": None if min_ms is None else round(min_ms, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
           "dtype": "u32 limbs (255-bit prime field Ft255, Montgomery form)", "data": "synthetic",
           "config": {"workload": "lcpc-ligero-pc commit, Ft255, 2^%d coeffs, rho=1/2, BLAKE3" % args.log_len,
                      "n_rows": n_rows_total, "n_per_row": n_per_row, "n_cols": n_cols,
                      "sharding": ("rows x%d (BLAKE3-chunk aligned), 1 exchange of subtree CVs (%s)" %
                                   (world, ("RCCL inside the library" + ("" if args.no_async_tail else "; two LcCommit objects filled alternately, "
                                            "commit k's exchange + leaf digests + tree on its own stream while commit k + 1 encodes "
                                            "(LCPC_COMMIT_ASYNC_TAIL)")) if args.exchange == "native" else "torch.distributed all-gather")) if distributed else "none",
                      "input": "device-resident (HBM); Field::random over ChaCha20Rng::from_seed([%d; 32]), stream 0, drawn on the device "
                               "(lcpc_random_coeffs_device; SURVEY.md 8d)" % INPUT_SEED,
                      "coeffs": "borrowed: LcCommit.coeffs aliases the caller's HBM buffer (LCPC_COMMIT_BORROW_COEFFS)" if borrow
                                else "copied into the LcCommit (as LcCommit::commit does, lcpc-2d/src/lib.rs:636-645)"},
           "roofline": roofline}
    roofline["power"] = power
    if power and valu_issue and "mix" in valu_issue:
        # one number for the remaining headroom: the mix-weighted issue peak at the clock sampled while this very commit loops
        ghz = power["sclk_MHz"] / 1e3
        wp = valu_issue["weighted_peak_Ginst_per_s"] * ghz / valu_issue["mix"]["ubench_clock_GHz"]
        valu_issue["weighted_peak_Ginst_per_s"] = round(wp, 1)
        valu_issue["frac_weighted"] = round(valu_issue["achieved_Ginst_per_s"] / wp, 3)
        valu_issue["weighted_at"] = "the sampled clock (%.3f GHz)" % ghz
    if other_mode is not None:
        out["other_coeffs_mode"] = other_mode
    if e2e is not None:
        out["e2e_host"] = e2e
    if shard_ms is not None:
        out["shard_ms"] = shard_ms
    if scaling_extra:
        out.update(scaling_extra)
    if distributed:
        out["ranks_seen"] = ranks_seen
        out["check"] = check
        out["devices"] = "one HIP device per rank" if (args.dist_backend == "nccl" and args.force_device is None) else \
                         "DEBUG: ranks share devices (--dist-backend %s%s)" % (args.dist_backend, "" if args.force_device is None else ", --force-device %d" % args.force_device)
    if distributed and exchange_note:
        out["exchange_fallback"] = exchange_note

    cfg_pub, cfg_priv = None, {}
    if world == 1 and not args.no_configs:
        cfg_pub, cfg_priv = hip_configs(torch, lcpc_amd, coeffs, n_coeffs_job, enc, cm, step, stream, local_rank)
        out["configs"] = cfg_pub

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import numpy as np
        import oracle_lib as O
        threads = usable_cores()
        lg = args.cpu_sample_log_len
        if lg is None:
            avail = host_memory_available()
            need = lambda l: 10 * (32 << l)            # coeffs + the oracle's coeffs copy + comm (2x) + slack
            lg = args.log_len if (avail is None or avail > need(args.log_len) + (8 << 30)) else args.log_len - 1
        # THE timed vector, drawn again on the host by the oracle's serial generator (the first 2^lg elements of the same stream)
        t_gen = time.perf_counter()
        cpu_coeffs = O.random_elems(3, 1 << lg, INPUT_SEED)
        t_gen = time.perf_counter() - t_gen
        same = bool((torch.from_numpy(cpu_coeffs.view(np.int64)) == coeffs[:1 << lg].cpu()).all())
        if not same:
            raise SystemExit("bench.py: the device-drawn coefficients differ from the oracle's Field::random stream")
        v, secs, root_cpu, oenc26, oc26 = cpu_commit(O, cpu_coeffs, n_per_row, n_cols, threads, keep=True)
        # the HIP root of the same data: at the full length it is the root the timed loop itself produced
        root_gpu = LcCommit.commit_device(coeffs.data_ptr(), 1 << lg, enc, stream, sync=True, into=cm).get_root()
        if root_gpu != root_cpu:
            raise SystemExit("bench.py: HIP root != oracle root on the timed coefficients (2^%d): %s vs %s" % (lg, root_gpu.hex(), root_cpu.hex()))
        if cfg_pub is not None:
            check_configs(O, cfg_pub, cfg_priv, cpu_coeffs, threads, oenc26, oc26 if lg == args.log_len else None)
        del oc26, oenc26
        lg1 = max(lg - 4, 17)
        v1, secs1, _ = cpu_commit(O, cpu_coeffs[:1 << lg1], n_per_row, n_cols, 1)
        del cpu_coeffs
        out["cpu_baseline"] = {"value": v, "unit": "field-elements/s", "cores": threads, "kind": "port",
                               "host_hw_threads": os.cpu_count(),
                               "root_equals_hip_root": True, "coeffs_equal_timed_coeffs": True,
                               "root": root_cpu.hex(),
                               "one_thread": {"value": v1, "cores": 1, "sample": "2^%d coeffs (%d rows), %.1f s wall" % (lg1, (1 << lg1) // n_per_row, secs1)},
                               "sample": "oracle C port (OpenMP, one thread per usable core: affinity mask capped by the cgroup CPU "
                                         "quota), Ligero Ft255 commit of 2^%d coeffs with the headline row "
                                         "shape (%d x %d -> %d), %.1f s wall (+ %.1f s drawing them); the coefficients are the TIMED vector%s, "
                                         "checked element for element against the device copy, and the HIP commit of it gives the same root"
                                         % (lg, (1 << lg) // n_per_row, n_per_row, n_cols, secs, t_gen, "" if lg == args.log_len else "'s first 2^%d elements" % lg)}
    if cfg_pub is not None and args.no_cpu_baseline:
        check_configs(None, cfg_pub, {}, [], 0, None, None)
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
