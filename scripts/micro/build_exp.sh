#!/usr/bin/env bash
# Build an EXPERIMENTAL copy of libmdno.so with extra -D flags, beside the real one (measurement aid only):
#   scripts/micro/build_exp.sh out-name.so -DMDNO_EXP_SOMETHING ...
# The result goes to scripts/micro/exp/ (git-ignored, shipped to the GPU box); run with MDNO_LIB=<path> (its build id
# reads "experimental", which is what lets the loader accept it: such a library is by construction not built from the tree's sources).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
src="$root/molecular_dynamics_neural_operator_amd/csrc"
out="$root/scripts/micro/exp/$1"; shift
tmp="$(mktemp -d)"; trap 'rm -rf "$tmp"' EXIT
pids=()
for f in "$src"/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function \
    -DMDNO_BUILD_ID='"experimental"' "$@" -c "$f" -o "$tmp/$(basename "${f%.hip}").o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$tmp"/*.o
echo "built $out"
