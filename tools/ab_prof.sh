# A/B of two library builds under rocprofv3 --stats at one size, one evaluation at a time:
#   tools/ab_prof.sh N   (expects gp_amd/csrc/libgpmi_old.so next to libgpmi.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abp; mkdir -p $O; N=${1:-4096}
cp $R/gp_amd/csrc/libgpmi.so $R/gp_amd/csrc/new.so
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then cp $R/gp_amd/csrc/libgpmi_old.so $R/gp_amd/csrc/libgpmi.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -- python3 $R/bench.py --n $N --grid-lanes 1 --steps 8 --warmup 1 --no-cpu-baseline --no-c4 --no-c1 > $O/p_$v.log 2>&1
  python3 $R/tools/prof_summary.py $O/p_$v $O/stats_${v}_n$N.txt "$v n=$N" > /dev/null
  head -8 $O/stats_${v}_n$N.txt | cut -c1-150
  rm -rf $O/p_$v
done
cp $R/gp_amd/csrc/new.so $R/gp_amd/csrc/libgpmi.so
