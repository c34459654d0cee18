set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r3_fullsuite.log
rc=$?
cat gpurun_out/r3_fullsuite.log
exit $rc
