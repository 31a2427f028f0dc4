#!/usr/bin/env python3
"""tools/leak_soak.py [iters] -- resource soak: encoder contexts and commitments are created and destroyed, commitments
refilled, proofs made, verified and freed, over and over; device memory in use (hipMemGetInfo through torch) and the host
RSS must not grow once the first iterations have warmed the pools up.  Exits non-zero on growth."""
import gc
import os
import resource
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import oracle_lib as O
from common import mk_transcript, powers
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding, Transcript


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * resource.getpagesize() / 1e6


def dev_used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e6


iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fid = 3
shapes = [("ligero", 1 << 14), ("ligero", 50000), ("sdig", 30000), ("ligero", 1 << 17)]
data = {n: O.random_elems(fid, n, 7 + n % 97) for _, n in shapes}
dev = {n: torch.from_numpy(d.view(np.int64)).cuda() for n, d in data.items()}
marks = []
for it in range(iters):
    kind, n = shapes[it % len(shapes)]
    enc = LigeroEncoding.new(fid, n) if kind == "ligero" else SdigEncoding.new(fid, n, it % 5)
    c = LcCommit(enc)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):                                   # refill the same object
        LcCommit.commit_device(dev[n].data_ptr(), n, enc, st, sync=True, into=c, borrow=bool(rep))
    c2 = LcCommit.commit(data[n], enc)                     # a second commitment under the same encoder, host entry point
    assert c2.get_root() == c.get_root()
    x = 12345 + it
    inner, outer = powers(O, fid, x, c.n_per_row), powers(O, fid, x, c.n_rows, c.n_per_row)
    root, nco = c.get_root(), enc.get_n_col_opens()
    pf = c.prove(outer, enc, mk_transcript(Transcript, root, nco))
    pf.verify(root, outer, inner, enc, mk_transcript(Transcript, root, nco))
    c.open_columns([0, 1, c.n_cols - 1])
    del pf, c, c2, enc
    if it % 25 == 24:
        gc.collect()
        torch.cuda.synchronize()
        marks.append((it + 1, dev_used_mb(), rss_mb()))
        print("iter %d: device %.1f MB in use, host RSS %.1f MB" % marks[-1], flush=True)
warm = [m for m in marks if m[0] >= iters // 3]
d_dev = warm[-1][1] - warm[0][1]
d_rss = warm[-1][2] - warm[0][2]
print("growth after warm-up over %d iterations: device %+.1f MB, host %+.1f MB" % (warm[-1][0] - warm[0][0], d_dev, d_rss))
if d_dev > 64 or d_rss > 64:
    sys.exit("resource growth")
print("no growth")
