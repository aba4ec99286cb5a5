#!/bin/bash
# Run on the GPU box (via gpurun): a few rocprofv3 --pmc passes over a short eager bench run.
# usage: scripts/pmc_passes.sh "<counters pass 1>" "<counters pass 2>" ...   -> gpurun_out/pmc/passN
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for counters in "$@"; do
  i=$((i+1))
  rm -rf $ROOT/gpurun_out/pmc/pass$i
  rocprofv3 --pmc $counters --output-format csv -d $ROOT/gpurun_out/pmc/pass$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --skip-cpu-baseline --skip-roofline --single-mode > $ROOT/gpurun_out/pmc/pass$i.json 2> $ROOT/gpurun_out/pmc/pass$i.err
  echo "pass $i done: $counters"
done
