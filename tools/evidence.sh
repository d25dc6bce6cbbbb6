# The round's evidence set (run on the GPU box through gpurun; outputs under gpurun_out/ev, copied to
# profiles/ by hand):  tools/evidence.sh [part]   part = tests | bench | prof | pmc | all (default)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev; mkdir -p $O
PART=${1:-all}
cd $R
if [ $PART = tests ] || [ $PART = all ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
fi
if [ $PART = bench ] || [ $PART = all ]; then
  timeout -k 10 500 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_c3.json; cut -c1-200 $O/bench_c3.json
  timeout -k 10 300 python bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c4.log 2>&1; grep '^{"metric"' $O/bench_c4.log > $O/bench_c4.json; cut -c1-160 $O/bench_c4.json
  timeout -k 10 300 python bench.py --workload c5 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_c5.log 2>&1; grep '^{"metric"' $O/bench_c5.log > $O/bench_c5.json; cut -c1-160 $O/bench_c5.json
  timeout -k 10 300 python bench.py --n 4096 --steps 24 --warmup 4 --no-cpu-baseline --no-c4 > $O/bench_c2.log 2>&1; grep '^{"metric"' $O/bench_c2.log > $O/bench_c2_n4096.json; cut -c1-160 $O/bench_c2_n4096.json
  timeout -k 10 300 python bench.py --gpus 2 --rehearse --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_rehearse2.log 2>&1; grep '^{"metric"' $O/bench_rehearse2.log > $O/bench_c3_rehearse_2ranks_1gpu.json; cut -c1-160 $O/bench_c3_rehearse_2ranks_1gpu.json
fi
if [ $PART = prof ] || [ $PART = all ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-c4 --no-c1 > $O/prof_default.log 2>&1
  grep '^{"metric"' $O/prof_default.log > $O/prof_default.json
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seq -- python3 $R/bench.py --grid-lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-c1 > $O/prof_seq.log 2>&1
  grep '^{"metric"' $O/prof_seq.log > $O/prof_seq.json
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seq_c4 -- python3 $R/bench.py --workload c4 --grid-lanes 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_seq_c4.log 2>&1
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seq_c5 -- python3 $R/bench.py --workload c5 --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_seq_c5.log 2>&1
  python3 $R/tools/prof_summary.py $O/prof_default $O/bench_c3_kernel_stats.txt "bench.py --no-cpu-baseline --no-c4 --no-c1 (default: 4 lanes; overlapping launches inflate durations)" > /dev/null
  python3 $R/tools/prof_summary.py $O/prof_seq $O/bench_c3_sequential_kernel_stats.txt "bench.py --grid-lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-c4 --no-c1" > /dev/null
  python3 $R/tools/prof_summary.py $O/prof_seq_c4 $O/bench_c4_sequential_kernel_stats.txt "bench.py --workload c4 --grid-lanes 1 --steps 1 --warmup 1 (N = 8192, one evaluation at a time)" > /dev/null
  python3 $R/tools/prof_summary.py $O/prof_seq_c5 $O/bench_c5_kernel_stats.txt "bench.py --workload c5 --steps 4 --warmup 1 (joint [y, y'] covariance, order 16384)" > /dev/null
  rm -rf $O/prof_default $O/prof_seq $O/prof_seq_c4 $O/prof_seq_c5
  cd $R
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  bash $R/tools/pmc_bench.sh c3 16384 --grid-lanes 1   # the configuration of bench.py's roofline pass: one evaluation at a time
  cp $R/gpurun_out/pmc_bench_c3_n16384.json $R/gpurun_out/pmc_bench_c3_n16384_sequential.json
  bash $R/tools/pmc_bench.sh c3 16384                  # default: 4 lanes + the sequential passes
  mv $R/gpurun_out/pmc_bench_c3_n16384.json $R/gpurun_out/pmc_bench_c3_n16384_lanes4_mix.json
  bash $R/tools/pmc_bench.sh c4 8192 --steps 1   # (two back-to-back grid calls under --pmc hang in the profiler; one step is 64 evaluations anyway)
  bash $R/tools/pmc_bench.sh c5 16384
fi
echo done
