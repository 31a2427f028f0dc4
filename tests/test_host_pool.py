"""The fork-join pool behind the host-side parallel_for (lcpc_amd/csrc/host_par.h; the reference uses rayon's global pool at the
same places, lcpc-2d/src/lib.rs:923-944): concurrent callers, nested regions, exception propagation -- a native stress test
(tests/native/host_par_stress.cpp) compiled against the library's own sources.  No GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_pool_stress(tmp_path):
    exe = str(tmp_path / "host_par_stress")
    src = [os.path.join(ROOT, "tests", "native", "host_par_stress.cpp"), os.path.join(ROOT, "lcpc_amd", "csrc", "encoding.cpp"),
           os.path.join(ROOT, "lcpc_amd", "csrc", "host_crypto.cpp")]
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "lcpc_amd", "csrc"), *src, "-o", exe],
                        capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-3000:]
    for _ in range(3):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
