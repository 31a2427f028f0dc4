import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R]
os.environ["LCPC_DEBUG_TIMING"] = "1"
import numpy as np, torch
import bench
import lcpc_amd
from lcpc_amd import LcCommit, LigeroEncoding, Transcript
n = 1 << 26
enc = LigeroEncoding.new(3, n)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit.commit_device(dev.data_ptr(), n, enc, st)
root = c.get_root()
inner = bench.powers_mont(bench.C5_X, c.n_per_row); outer = bench.powers_mont(bench.C5_X, c.n_rows, c.n_per_row)
for rep in range(4):
    LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=True, into=c)
    t0 = time.perf_counter(); pf = c.prove(outer, enc, bench.mk_transcript(Transcript, root, enc.get_n_col_opens())); t1 = time.perf_counter()
    ev = pf.verify(root, outer, inner, enc, bench.mk_transcript(Transcript, root, enc.get_n_col_opens())); t2 = time.perf_counter()
    print("prove %.2f verify %.2f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
