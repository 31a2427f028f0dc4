"""The library's NATIVE sharded commit / prove (lcpc_comm_init, lcpc_commit_sharded_device incl. LCPC_COMMIT_ASYNC_TAIL,
lcpc_prove_sharded_rccl) with 2 ... 8 ranks on the single-GPU test box.

RCCL itself refuses two ranks on one device, so until round 4 these entry points had only ever run with world = 1 (where every
rank-dependent branch -- the compact node layout, the broadcasts of second and third nodes, ranks that own nothing, the order of
collectives of several commitments -- is trivial).  tests/native/fake_rccl.cpp is an in-process stand-in for librccl (ranks = host
threads sharing the GPU; stream-ordered copies between the ranks' buffers) that the library loads through LCPC_RCCL_LIB; the
driver tests/native/fake_rccl_worlds.py runs 14 cases against the oracle.  What this does NOT test is RCCL's transport; what it
does test is every line of this repository that runs at N > 1."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_native_exchange_many_ranks_one_gpu(tmp_path):
    so = str(tmp_path / "libfake_rccl.so")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cc = subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + rocm + "/include",
                         os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"), "-o", so, "-L" + rocm + "/lib", "-lamdhip64", "-lpthread"],
                        capture_output=True, text=True, timeout=300)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, LCPC_RCCL_LIB=so)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native", "fake_rccl_worlds.py")], capture_output=True, text=True,
                       timeout=1500, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("all ok"), (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") == 14
