"""The C-ABI used from plain C (examples/abi_demo.c, built by __graft_entry__.build()): no Python, no torch in
the process.  Its commitment root must equal the golden fixture and its verified evaluation the true one."""
import os
import subprocess

import pytest

from common import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_demo_matches_golden():
    exe = os.path.join(ROOT, "examples", "abi_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_demo.c"),
                               "-L" + os.path.join(ROOT, "lcpc_amd", "lib"), "-llcpc_hip", "-Wl,-rpath,$ORIGIN/../lcpc_amd/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    case = [c for c in load_golden("commit_cases.json") if c["name"] == "ligero_ft63_2e10_iota"][0]
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["root"] == case["root"]
    assert int(lines["proof_bytes"]) == case["proof_len"]
    assert lines["eval"].split()[0] == "%016x" % int(case["eval"], 16) and lines["eval"].endswith("OK")
    # the commitment's own serde (bincode of WrappedLcCommit): 4 rows x 512 + 4 x 256 elements of 8 bytes, dims, 1023 digests
    n_rows, n_per_row, n_cols = 4, 256, 512
    size = 8 + n_rows * n_cols * 8 + 8 + n_rows * n_per_row * 8 + 24 + 8 + (2 * n_cols - 1) * 40
    assert lines["commit_bincode_bytes"] == "%d OK" % size
