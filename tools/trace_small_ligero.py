#!/usr/bin/env python3
"""tools/trace_small_ligero.py [log_len] -- five Ligero commits of 2^log_len Ft255 coefficients (default 13), to be run under
`rocprofv3 --kernel-trace` and read with tools/rocpd_dispatches.py: which launches a small commitment consists of."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
import bench_configs as B
from lcpc_amd import LcCommit, LigeroEncoding
lgl = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = 1 << lgl
enc = LigeroEncoding.new(3, n)
coeffs = B.rand_coeffs(n, 4, 1)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
for _ in range(5):
    LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, into=c)
