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
B="python3 $ROOT/bench.py --skip-cpu-baseline --skip-ensemble-leg --skip-config-legs"
for M in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_m$M -- $B --steps 50 --warmup 5 --total-members $M > $OUT/bench_m$M.json 2> $OUT/trace_m$M.err
  echo "trace m$M done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_m$M -- $B --steps 3 --warmup 1 --no-graph --total-members $M > $OUT/pmc_fetch_m$M.json 2> $OUT/pmc_fetch_m$M.err
  echo "fetch m$M done"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_m$M -- $B --steps 3 --warmup 1 --no-graph --total-members $M > $OUT/pmc_write_m$M.json 2> $OUT/pmc_write_m$M.err
  echo "write m$M done"
done
# matrix-pipe occupancy of the MFMA kernels of the timed path (1 member): its own counter pass
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_m1 -- $B --single-mode --steps 3 --warmup 1 --no-graph --total-members 1 > $OUT/pmc_mfma_m1.json 2> $OUT/pmc_mfma_m1.err
echo "mfma m1 done"
# training (cfg4: N=28, batch 128, k=1024, depth 6, bf16): kernel trace, then the same counters for its GEMMs
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -- python3 $ROOT/scripts/train_synthetic.py --frames 4000 > $OUT/train.json 2> $OUT/train.err
echo "train trace done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_train -- python3 $ROOT/scripts/train_synthetic.py --frames 600 > $OUT/pmc_mfma_train.json 2> $OUT/pmc_mfma_train.err
echo "mfma train done"
# the other BASELINE configurations, kernel traces only: training in fp32, shape A (N = 28, one member), shape C (N = 50,000)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_fp32_trace -- python3 $ROOT/scripts/train_synthetic.py --frames 2000 --precision fp32 > $OUT/train_fp32.json 2> $OUT/train_fp32.err
echo "train fp32 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shape_a_trace -- python3 $ROOT/scripts/shape_a_breakdown.py --steps 200 > $OUT/shape_a.json 2> $OUT/shape_a.err
echo "shape A trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shape_c_trace -- python3 $ROOT/scripts/run_shape_c.py --steps 2 > $OUT/shape_c.json 2> $OUT/shape_c.err
echo "shape C trace done"
# shape C counters (HBM bytes of K1, csrc/moment.hip, and of the materialised conv kernel on the 2.0M-edge slice): own passes
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_shape_c -- python3 $ROOT/scripts/run_shape_c.py --steps 1 > $OUT/pmc_fetch_shape_c.json 2> $OUT/pmc_fetch_shape_c.err
echo "shape C fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_shape_c -- python3 $ROOT/scripts/run_shape_c.py --steps 1 > $OUT/pmc_write_shape_c.json 2> $OUT/pmc_write_shape_c.err
echo "shape C write done"
