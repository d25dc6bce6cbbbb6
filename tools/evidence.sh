set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 500 python bench.py > $O/bench_default.log 2>&1; grep '^{"metric"' $O/bench_default.log > $O/bench_default.json; cut -c1-200 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline > $O/prof_default.log 2>&1
grep '^{"metric"' $O/prof_default.log > $O/prof_default.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seq -- python3 $R/bench.py --grid-lanes 1 --lookahead 0 --steps 4 --warmup 1 --no-cpu-baseline > $O/prof_seq.log 2>&1
grep '^{"metric"' $O/prof_seq.log > $O/prof_seq.json
python3 $R/tools/prof_summary.py $O/prof_default $O/prof_default_kernel_stats.txt "bench.py (default) c3" > /dev/null
python3 $R/tools/prof_summary.py $O/prof_seq $O/prof_seq_kernel_stats.txt "bench.py --grid-lanes 1 --lookahead 0 --steps 4 --warmup 1" > /dev/null
rm -rf $O/prof_default $O/prof_seq
bash $R/tools/pmc_bench.sh c3
rm -rf $R/gpurun_out/pmcb_c3
echo done
