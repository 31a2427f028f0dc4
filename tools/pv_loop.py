import os, sys, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tools")
import torch, numpy as np
import bench_configs as B
import oracle_lib as O
from common import mk_transcript, powers
import pyref as P
from lcpc_amd import LcCommit, LigeroEncoding, Transcript
n = 1 << 26
enc = LigeroEncoding.new(3, n)
coeffs = B.rand_coeffs(n, 4, 1)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, borrow=True)
x = 0x123456789abcdef % P.FT255.p
inner = powers(O, 3, x, c.n_per_row); outer = powers(O, 3, x, c.n_rows, c.n_per_row)
root = c.get_root(); nco = enc.get_n_col_opens()
for rep in range(int(os.environ.get("PV_REPS", "4"))):
    LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, borrow=True, into=c)
    t0 = time.perf_counter(); pf = c.prove(outer, enc, mk_transcript(Transcript, root, nco)); tp = time.perf_counter() - t0
    t0 = time.perf_counter(); pf.verify(root, outer, inner, enc, mk_transcript(Transcript, root, nco)); tv = time.perf_counter() - t0
    print(json.dumps({"prove_ms": round(tp * 1e3, 2), "verify_ms": round(tv * 1e3, 2)}), flush=True)
