#!/usr/bin/env python3
"""tools/trace_small_any.py FIELD L LOG_LEN -- five Ligero commits of 2^LOG_LEN coefficients in field FIELD (L u64 limbs), to be run
under `rocprofv3 --kernel-trace` and read with tools/rocpd_dispatches.py (tools/trace_small.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
import bench_configs as B
from lcpc_amd import LcCommit, LigeroEncoding
fid, L, n = int(sys.argv[1]), int(sys.argv[2]), 1 << int(sys.argv[3])
enc = LigeroEncoding.new(fid, n)
coeffs = B.rand_coeffs(n, L, 1)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
for _ in range(5):
    LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, into=c)
