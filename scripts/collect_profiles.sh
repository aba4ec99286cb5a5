#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + the PMC passes the roofline numbers come from, for the
# 1-member (BASELINE configs[1]) and the 8-member (one GPU's share of configs[2] at 8 GPUs) rollout, and
# a kernel trace of the training script.  Output under gpurun_out/prof; then, back in the build container,
#   python scripts/summarize_profile.py gpurun_out/prof rNN
# writes the summaries under profiles/.  Counters are collected in their own runs (no trace options).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --skip-cpu-baseline --skip-ensemble-leg"
for M in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_m$M -- $B --steps 50 --warmup 5 --total-members $M > $OUT/bench_m$M.json 2> $OUT/trace_m$M.err
  echo "trace m$M done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_m$M -- $B --steps 3 --warmup 1 --no-graph --total-members $M > $OUT/pmc_fetch_m$M.json 2> $OUT/pmc_fetch_m$M.err
  echo "fetch m$M done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_m$M -- $B --steps 3 --warmup 1 --no-graph --total-members $M > $OUT/pmc_write_m$M.json 2> $OUT/pmc_write_m$M.err
  echo "write m$M done"
done
# training step (cfg4 stand-in): kernel trace only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -- python3 $ROOT/scripts/train_synthetic.py > $OUT/train.json 2> $OUT/train.err
echo "train trace done"
