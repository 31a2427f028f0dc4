"""tools/lab/ab_hash.py -- hash_ms (column hash: leaf chunks + fold) and total of the headline commit on two builds, interleaved"""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, time
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, LigeroEncoding
n = 1 << 26
enc = LigeroEncoding.new(3, n)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
root = LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c).get_root()
for _ in range(10): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize()
c.set_timing(True)
hs, ts = [], []
for _ in range(10):
    LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c); t = c.timings(); hs.append(t.hash_ms); ts.append(t.total_ms)
print(root.hex()[:16], "hash %%.4f total %%.3f" %% (sum(hs) / 10, sum(ts) / 10))
''' % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for rep in range(3):
    for name, path in libs.items():
        out = subprocess.run([sys.executable, "-c", child, path], capture_output=True, text=True)
        print("2^26", name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
