set -e
cd gp_amd/csrc && cp libgpmi.so new.so && cd ../..
run() {
  python3 bench.py --no-cpu-baseline --no-c4 --no-c1 > gpurun_out/ab_c3_$1.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --workload c4 --steps 3 --warmup 1 --no-c1 > gpurun_out/ab_c4_$1.json 2>/dev/null
  python3 bench.py --no-cpu-baseline --n 4096 --no-c4 --no-c1 > gpurun_out/ab_c2_$1.json 2>/dev/null
  python3 - <<PY
import json
for w in ("c3","c4","c2"):
    d=json.load(open("gpurun_out/ab_%s_$1.json"%w))
    print("$1", w, "value %.2f  ms/step %.3f  seq %.3f  syrk %.2f TF" % (d["value"], d["ms_per_step"], d["ms_per_eval_sequential"], d["roofline"]["achieved"]))
PY
}
for r in 1 2; do
  cp gp_amd/csrc/new.so gp_amd/csrc/libgpmi.so; run new$r
  cp gp_amd/csrc/libgpmi_old.so gp_amd/csrc/libgpmi.so; run old$r
done
cp gp_amd/csrc/new.so gp_amd/csrc/libgpmi.so
