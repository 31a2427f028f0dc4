"""tools/lab/pv_dbg_small.py [log_len] -- prove / verify phase times (LCPC_DEBUG_TIMING) at a mid size, ten repetitions: where does the spread sit?"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R]
os.environ["LCPC_DEBUG_TIMING"] = "1"
import torch, bench
from lcpc_amd import LcCommit, LigeroEncoding, Transcript
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n = 1 << lg
kind = sys.argv[2] if len(sys.argv) > 2 else "ligero"
if kind == "sdig":
    from lcpc_amd import SdigEncoding
    enc = SdigEncoding.new(3, n, 0)
else:
    enc = LigeroEncoding.new(3, n)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit.commit_device(dev.data_ptr(), n, enc, st)
root = c.get_root()
inner = bench.powers_mont(bench.C5_X, c.n_per_row); outer = bench.powers_mont(bench.C5_X, c.n_rows, c.n_per_row)
for rep in range(10):
    t0 = time.perf_counter(); pf = c.prove(outer, enc, bench.mk_transcript(Transcript, root, enc.get_n_col_opens())); t1 = time.perf_counter()
    ev = pf.verify(root, outer, inner, enc, bench.mk_transcript(Transcript, root, enc.get_n_col_opens())); t2 = time.perf_counter()
    print("prove %.3f verify %.3f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
