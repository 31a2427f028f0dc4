#!/usr/bin/env python3
"""tools/bench_rs.py -- the matrix of the reference's `cargo bench` functions (lcpc-ligero-pc/src/bench.rs:23-212,
lcpc-brakedown-pc/src/bench.rs:22-155; feature `bench`): commit / prove / verify for Ft127 and Ft255 at 2^16, 2^20 and 2^24
coefficients, Ligero (its default rate there: rho = 1/2 alias) and Brakedown, on the MI355X path; one JSON line per cell
(tools/bench_pvs.run: mean of 10 iterations, encoder construction outside, root / proof on the host every iteration)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import bench_pvs

for kind in ("ligero_hlf", "sdig"):
    for fid in (1, 3):
        for lgl in (16, 20, 24):
            bench_pvs.run(kind, lgl, fid)
