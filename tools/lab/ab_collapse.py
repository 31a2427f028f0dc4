"""tools/lab/ab_collapse.py -- eval_outer (collapse_columns of 1 and 2 tensors, lcpc-2d/src/lib.rs:1095-1123) at the headline shape on two
builds of the library, interleaved: device-side timing with HIP events through lcpc_collapse_device"""
import os, sys, subprocess
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
child = r'''
import os, sys, time, ctypes as C
sys.path[:0] = [%r]
import lcpc_amd._lib as L
L.LIB_PATH = sys.argv[1]
import torch
from lcpc_amd import LcCommit, LigeroEncoding
n = 1 << 26
enc = LigeroEncoding.new(3, n)
dev = enc.random_coeffs_device(n, seed=0)
st = torch.cuda.current_stream().cuda_stream
c = LcCommit.commit_device(dev.data_ptr(), n, enc, st)
lib = L.lib()
for nt in (1, 2):
    t = enc.random_coeffs_device(nt * c.n_rows, seed=3)
    out = torch.empty((nt * c.n_per_row, 4), dtype=torch.int64, device="cuda")
    for _ in range(3): c._check(lib.lcpc_collapse_device(c._h, C.c_void_p(t.data_ptr()), nt, C.c_void_p(st), C.c_void_p(out.data_ptr())))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c._check(lib.lcpc_collapse_device(c._h, C.c_void_p(t.data_ptr()), nt, C.c_void_p(st), C.c_void_p(out.data_ptr())))
    e1.record(); torch.cuda.synchronize()
    print("nt=%%d %%.4f ms" %% (nt, e0.elapsed_time(e1) / 20), int(out.sum().item()) & 0xffff, end="  ")
print()
''' % R
libs = {"base": os.path.join(R, "tools/lab/base/liblcpc_hip.so"), "new ": os.path.join(R, "lcpc_amd/lib/liblcpc_hip.so")}
for rep in range(3):
    for name, path in libs.items():
        out = subprocess.run([sys.executable, "-c", child, path], capture_output=True, text=True)
        print("collapse 2^26", name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
