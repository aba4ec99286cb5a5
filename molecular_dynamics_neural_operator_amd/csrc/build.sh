#!/usr/bin/env bash
# Build libmdno.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libmdno.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
srcs=(engine.hip graph.hip edge_mlp.hip edge_mlp_split.hip factored.hip moment.hip nnconv.hip node_ops.hip train.hip train_bf16.hip train_nodes.hip collate.hip gemm_bf16.hip)
objs=()
pids=()
mkdir -p "$here/build"
for f in "${srcs[@]}"; do
  o="$here/build/${f%.hip}.o"
  objs+=("$o")
  if [[ ! -f "$o" || "$here/$f" -nt "$o" || "$here/common.h" -nt "$o" || "$here/kernels.h" -nt "$o" || "$here/split_layout.h" -nt "$o" || "$here/graph_small.h" -nt "$o" || "$here/mfma_f32.h" -nt "$o" || "$here/reduce.h" -nt "$o" || "$here/../../include/mdno.h" -nt "$o" ]]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
      -c "$here/$f" -o "$o" "$@" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
