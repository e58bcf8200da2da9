# round 4, session f: per-layer kernel records of the small configurations (eager, one stream)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_f; mkdir -p $O; cd $R
for A in mnist dcgan32 cyclegan256; do
  python bench.py --arch $A --by-layer --streams 1 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/$A.log 2>$O/$A.err
  python - $A <<'PY'
import json, sys
a = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r04_f/{a}.log") if l.startswith("{")][-1])
print(a, d["value"], d["ms_per_step"])
rows = sorted(d["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])
for k, v in rows[:28]:
    print("  %-92s n=%4d %7.1f us %6.1f TF (nominal %6.1f) share %.3f" % (k[:92], v["launches"], v["avg_us"], v["tflops"], v["nominal_tflops"], v["share_of_step"]))
PY
done
