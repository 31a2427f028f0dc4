import os, sys, subprocess, json
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, time
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, LigeroEncoding
lg = int(sys.argv[2]); rows = 512 if lg >= 26 else 256
n = 1 << lg
enc = LigeroEncoding.new(3, n)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
root = LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c).get_root()
for _ in range(5): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
c.set_timing(True); LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c); t = c.timings()
print(root.hex()[:16], round(ms, 3), round(t.encode_ms, 3))
''' % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for lg in (26, 24):
    for rep in range(3):
        for name, path in libs.items():
            out = subprocess.run([sys.executable, "-c", child, path, str(lg)], capture_output=True, text=True)
            print("2^%d" % lg, name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
