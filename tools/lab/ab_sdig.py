"""tools/lab/ab_sdig.py -- two builds of the library on the Brakedown commit of 2^24 Ft255 coefficients (C3), one box, interleaved"""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, time
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, SdigEncoding
fid, lg = int(sys.argv[2]), int(sys.argv[3])
n = 1 << lg
enc = SdigEncoding.new(fid, n, 0)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit(enc)
root = LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c).get_root()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    for _ in range(5): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
    torch.cuda.synchronize()
reps = 0; t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3 or reps < 20:
    for _ in range(10): LcCommit.commit_device(dev.data_ptr(), n, enc, st, sync=False, into=c)
    torch.cuda.synchronize(); reps += 10
ms = (time.perf_counter() - t0) / reps * 1e3
c.set_timing(True); LcCommit.commit_device(dev.data_ptr(), n, enc, st, into=c); t = c.timings()
print(root.hex()[:16], round(ms, 3), round(t.encode_ms, 3))
''' % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for fid, lg in ((3, 24), (2, 24)):
    for rep in range(3):
        for name, path in libs.items():
            out = subprocess.run([sys.executable, "-c", child, path, str(fid), str(lg)], capture_output=True, text=True)
            print("sdig field %d 2^%d" % (fid, lg), name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
