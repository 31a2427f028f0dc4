"""the child process of ab_shapes.py / ab_quick.py: commits of log_cols x rows on the library given as argv[1]"""
CHILD = r'''
import os, sys, time
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, LigeroEncoding
log_cols, rows = int(sys.argv[2]), int(sys.argv[3])
npr = 1 << (log_cols - 1)
n = rows * npr
enc = LigeroEncoding.new_from_dims(3, npr, 2 * npr)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
root = LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c).get_root()
for _ in range(5): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print(root.hex()[:16], round(ms, 3))
'''
