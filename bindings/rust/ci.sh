#!/usr/bin/env bash
# bindings/rust/ci.sh -- the three commands a maintainer runs on a machine that has BOTH a Rust toolchain (with crates.io access) and
# an MI355X (gfx950).  Nothing in this repository's own image can run them (no cargo / rustc, no network), so neither the two crates
# here nor the CPU oracle have ever been checked against the real reference: DESIGN.md carries that as "parity: partial / unpinned".
# The first green run of this script is what turns it green:
#   1. rustc sees the crates for the first time (trait bounds, lifetimes, the serde round trips through the reference's types);
#   2. `commit_prove_verify_equals_reference` shows the GPU path == the reference's own commit / prove, root and proof bytes;
#   3. oracle/repin shows the CPU oracle (and with it the 500-odd GPU parity tests) == the real crates on the nine golden cases.
#
#   usage: bindings/rust/ci.sh <checkout of conroi/lcpc>          (this repository is found from the script's own location)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
REF="${1:?usage: ci.sh <checkout of conroi/lcpc>}"
REF="$(cd "$REF" && pwd)"

# the product library (hipcc --offload-arch=gfx950) and the place cargo finds it
make -C "$REPO/lcpc_amd/csrc" -s -j4
export LCPC_HIP_LIB_DIR="$REPO/lcpc_amd/lib"
export LD_LIBRARY_PATH="$LCPC_HIP_LIB_DIR:${LD_LIBRARY_PATH:-}"

# the two crates sit inside the reference's workspace as hip/ (the `path =` dependencies of lcpc-hip/Cargo.toml assume that)
rm -rf "$REF/hip"
cp -r "$HERE" "$REF/hip"
grep -q '"hip/lcpc-hip"' "$REF/Cargo.toml" || sed -i 's|members = \[|members = [\n    "hip/lcpc-hip-sys",\n    "hip/lcpc-hip",|' "$REF/Cargo.toml"

cd "$REF"
echo "== 1/3  cargo test -p lcpc-hip   (the crates compile; same root and proof bytes as the reference's own commit / prove)"
cargo test -p lcpc-hip-sys --release
cargo test -p lcpc-hip --release

echo "== 2/3  the end-to-end example at 2^24 (GPU commit + prove, the reference's verify accepts the proof)"
cargo run -p lcpc-hip --release --example commit_prove -- 24

echo "== 3/3  oracle/repin: the nine golden cases from the real crates against tests/golden/commit_cases.json"
[ -e "$REPO/../lcpc" ] || ln -s "$REF" "$REPO/../lcpc"      # oracle/repin/Cargo.toml looks for the reference beside this repository
( cd "$REPO/oracle/repin" && cargo +nightly run --release | python3 compare.py )

echo "all three green: the parity status in DESIGN.md section 2 may be changed from 'partial (unpinned)' to 'pinned', citing this run"
