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


def test_host_code_under_sanitizers(tmp_path):
    """the same stress, and the transcript's batched absorb with every Keccak-f[1600] implementation (the generated asm
    statements included), built with AddressSanitizer + UndefinedBehaviorSanitizer."""
    inc = "-I" + os.path.join(ROOT, "lcpc_amd", "csrc")
    san = ["-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"]
    csrc = os.path.join(ROOT, "lcpc_amd", "csrc")
    builds = {
        "pool": [os.path.join(ROOT, "tests", "native", "host_par_stress.cpp"), os.path.join(csrc, "encoding.cpp"), os.path.join(csrc, "host_crypto.cpp")],
        "transcript": [os.path.join(ROOT, "tools", "bench_transcript.cpp"), os.path.join(csrc, "host_crypto.cpp")],
    }
    for name, src in builds.items():
        exe = str(tmp_path / name)
        cc = subprocess.run(["g++", *san, inc, *src, "-o", exe], capture_output=True, text=True, timeout=900)
        assert cc.returncode == 0, cc.stderr[-3000:]
        for mix in ([None] if name == "pool" else ["portable", "tern", "xor"]):
            env = dict(os.environ, **({"LCPC_KECCAK": mix} if mix else {}))
            r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, (name, mix, r.stdout[-500:], r.stderr[-3000:])
            if name == "transcript":
                assert "chk 0be85deb" in r.stdout, r.stdout      # the same transcript whatever the permutation's implementation
