#!/usr/bin/env bash
# Build libmdno.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# Usage: csrc/build.sh [extra hipcc flags]
#
# The library is tied to its sources by CONTENT, not by timestamps: the build id is the first 16 hex digits of
# sha256 over the bytes of csrc/*.{hip,h,sh} and include/mdno.h in C-locale name order (the Python side,
# _lib.source_build_id(), computes the same thing).  It is compiled into the library (mdno_build_id()) and kept in
# build/BUILD_ID; if the sources' id, the flags or the library differ from what build/ holds, EVERYTHING is rebuilt
# from scratch (all 13 files in parallel: ~10 s) — there is no per-file incrementality to go stale.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libmdno.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
export LC_ALL=C
mapfile -t id_files < <( { ls "$here"/*.hip "$here"/*.h "$here"/*.sh | sort; echo "$here/../../include/mdno.h"; } )
id="$(cat "${id_files[@]}" | sha256sum | cut -c1-16)"
stamp="$id $*"
if [[ -f "$out" && -f "$here/build/BUILD_ID" && "$(cat "$here/build/BUILD_ID")" == "$stamp" ]]; then
  echo "up to date: $out (build id $id)"
  exit 0
fi
rm -rf "$here/build"
mkdir -p "$here/build"
objs=()
pids=()
for src in "$here"/*.hip; do
  o="$here/build/$(basename "${src%.hip}").o"
  objs+=("$o")
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
    -DMDNO_BUILD_ID="\"$id\"" -c "$src" -o "$o" "$@" &
  pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[[ $fail == 0 ]] || { echo "build.sh: a compile failed" >&2; exit 1; }
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "$stamp" > "$here/build/BUILD_ID"
echo "built $out (build id $id)"
