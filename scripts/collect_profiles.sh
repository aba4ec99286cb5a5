#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + the two PMC passes the roofline numbers come from.
# Output: gpurun_out/prof/{trace,pmc_fetch,pmc_write}; then `python scripts/summarize_profile.py gpurun_out/prof rNN`.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 50 --warmup 5 --skip-cpu-baseline > $OUT/bench_line.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-graph --skip-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-graph --skip-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
# training step (cfg4 stand-in): kernel trace only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -- python3 $ROOT/scripts/train_synthetic.py > $OUT/train.json 2> $OUT/train.err
echo "train trace done"
