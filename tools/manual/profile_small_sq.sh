# --pmc SQ_* pass over the single-CU kernels (MFMA utilisation, wait / issue split of wave time): gpurun_out/<tag>/pmc_sq_small.txt
TAG=${1:-r4sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $OUT/pmc_sq_small.txt
for W in ekf64 ukf64; do
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq_$W -o p -- python3 bench.py --workload $W --no-legs --cpu-sample 0 --steps 2 --warmup 1 > /dev/null 2> $OUT/sq_$W.err
  echo "# bench.py --workload $W" >> $OUT/pmc_sq_small.txt
  python3 tools/pmc_summary.py $OUT/sq_$W >> $OUT/pmc_sq_small.txt 2>&1
  rm -rf $OUT/sq_$W
done
cat $OUT/pmc_sq_small.txt
