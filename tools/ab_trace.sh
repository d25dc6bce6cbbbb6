# launch-by-launch durations of the last evaluation for two library builds:  tools/ab_trace.sh N
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abt; mkdir -p $O; N=${1:-4096}
cp $R/gp_amd/csrc/libgpmi.so $R/gp_amd/csrc/new.so
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then cp $R/gp_amd/csrc/libgpmi_old.so $R/gp_amd/csrc/libgpmi.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_$v -- python3 $R/bench.py --n $N --grid-lanes 1 --steps 2 --warmup 1 --no-cpu-baseline --no-c4 --no-c1 > $O/t_$v.log 2>&1
  python3 - $O/t_$v > $O/launches_${v}_n$N.txt <<'PY'
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(nm): return re.sub(r"\(.*", "", nm.replace("(anonymous namespace)::", "").replace("void ", ""))[:16]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))) for r in rows]
st = max(i for i, e in enumerate(ev) if e[2].startswith("k_se_cov"))
for e in ev[st:]:
    print("%-16s %5d wg %8.2f us" % (e[2], e[3], (e[1] - e[0]) / 1e3))
    if e[2].startswith("k_logml_fin"): break
PY
  rm -rf $O/t_$v
done
cp $R/gp_amd/csrc/new.so $R/gp_amd/csrc/libgpmi.so
paste $O/launches_new_n$N.txt $O/launches_old_n$N.txt | awk '{d=$4-$9; printf "%s  %s\n", $0, (d>1.0||d<-1.0)? sprintf("%+.1f",d):""}' > $O/cmp_n$N.txt
