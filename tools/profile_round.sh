#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box into gpurun_out/prof/ (copy what should be judged into profiles/ afterwards):
#   kernel-trace --stats of the bf16 / split / CNN train steps, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (summarised with the
#   kernel-source hashes of THIS tree), SQ counters in two passes.  Usage (from the repo root, on the box):  bash tools/profile_round.sh r02
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
BENCH="python3 bench.py --no-cpu-baseline --no-fp32-leg --no-parity-leg --no-inference-leg --no-cnn-leg --profile-steps 0"
for P in bf16 split; do
  echo "== $P kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${P}_stats -- $BENCH --precision $P --steps 40 --warmup 10 > $OUT/${P}_stats.json 2> $OUT/${P}_stats.err
  cp $(find $OUT/${P}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${P}_kernel_stats.csv
  cp $OUT/${P}_stats.json $OUT/${TAG}_${P}_bench_under_rocprof.json
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== $P pmc $C"; rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${P}_$C -- $BENCH --precision $P --steps 5 --warmup 2 > /dev/null 2> $OUT/${P}_$C.err
    cp $(find $OUT/${P}_$C -name "*counter_collection.csv" | head -1) $OUT/${TAG}_${P}_pmc_$C.csv
  done
  python3 profiles/summarize_pmc.py $OUT/${TAG}_${P}_pmc_FETCH_SIZE.csv $OUT/${TAG}_${P}_pmc_WRITE_SIZE.csv $OUT/${TAG}_${P}_pmc_traffic.json "Command: rocprofv3 --kernel-trace --pmc <counter> -- $BENCH --precision $P --steps 5 --warmup 2 (round ${TAG})."
  echo "== $P sq"; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${P}_sq1 -- $BENCH --precision $P --steps 5 --warmup 2 > /dev/null 2> $OUT/${P}_sq1.err
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${P}_sq2 -- $BENCH --precision $P --steps 5 --warmup 2 > /dev/null 2> $OUT/${P}_sq2.err
  python3 profiles/summarize_sq.py $OUT/${TAG}_${P}_pmc_sq.json $(find $OUT/${P}_sq1 $OUT/${P}_sq2 -name "*counter_collection.csv")
done
echo "== cnn kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cnn_stats -- python3 bench.py --workload cnn --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/${TAG}_cnn_bench_under_rocprof.json 2> $OUT/cnn_stats.err
cp $(find $OUT/cnn_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cnn_kernel_stats.csv
CNN="python3 bench.py --workload cnn --no-cpu-baseline --profile-steps 0"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== cnn pmc $C"; rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cnn_$C -- $CNN --steps 5 --warmup 2 > /dev/null 2> $OUT/cnn_$C.err
  cp $(find $OUT/cnn_$C -name "*counter_collection.csv" | head -1) $OUT/${TAG}_cnn_pmc_$C.csv
done
python3 profiles/summarize_pmc.py $OUT/${TAG}_cnn_pmc_FETCH_SIZE.csv $OUT/${TAG}_cnn_pmc_WRITE_SIZE.csv $OUT/${TAG}_cnn_pmc_traffic.json "Command: rocprofv3 --kernel-trace --pmc <counter> -- $CNN --steps 5 --warmup 2 (round ${TAG})."
echo "== cnn sq"; rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/cnn_sq1 -- $CNN --steps 5 --warmup 2 > /dev/null 2> $OUT/cnn_sq1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/cnn_sq2 -- $CNN --steps 5 --warmup 2 > /dev/null 2> $OUT/cnn_sq2.err
python3 profiles/summarize_sq.py $OUT/${TAG}_cnn_pmc_sq.json $(find $OUT/cnn_sq1 $OUT/cnn_sq2 -name "*counter_collection.csv")
ls -la $OUT/${TAG}_*
# the raw per-dispatch traces are large: keep only the summaries
rm -rf $OUT/*_stats $OUT/bf16_FETCH_SIZE $OUT/bf16_WRITE_SIZE $OUT/split_FETCH_SIZE $OUT/split_WRITE_SIZE $OUT/cnn_FETCH_SIZE $OUT/cnn_WRITE_SIZE $OUT/*_sq1 $OUT/*_sq2
