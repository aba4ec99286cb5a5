#!/usr/bin/env bash
# Per-kernel register / LDS / scratch usage of one HIP source, compiled device-only for gfx950 (no GPU needed).
# Usage: scripts/kernel_resources.sh csrc-file.hip [name-filter-regex]
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/molecular_dynamics_neural_operator_amd/csrc/$1"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMDNO_BUILD_ID=\"x\" --cuda-device-only -c "$src" -o "$tmp/dev.bundle"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$tmp/dev.bundle" \
    --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$tmp/dev.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.co" | awk -v f="${2:-}" '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2}
  /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.vgpr_spill_count:/ {sp=$2; if (name ~ f) printf "%s vgpr %d agpr %d sgpr %d lds %d scratch %d spill %d\n", name, v, a, s, l, p, sp}' | c++filt
