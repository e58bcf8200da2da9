# round 4, session m: config 5 per layer with / without the 128x64 rule for under-filled launches that leave statistics
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_m; mkdir -p $O; cd $R
for v in hip nonarrow; do
  CGS_LIB=$R/collaborative-gan-sampling_amd/libcgs_$v.so python bench.py --arch cyclegan256 --by-layer --streams 1 --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/$v.log 2>$O/$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r04_m/{v}.log") if l.startswith("{")][-1])
print(v, d["value"], d["ms_per_step"])
for k, x in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:14]:
    print("  %-86s n=%4d %7.1f us %6.1f TF share %.3f" % (k[:86], x["launches"], x["avg_us"], x["tflops"], x["share_of_step"]))
PY
  for i in 1 2; do CGS_LIB=$R/collaborative-gan-sampling_amd/libcgs_$v.so python bench.py --arch cyclegan256 --no-cpu-baseline --no-other-configs 2>>$O/stderr.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   default run', d['value'])"; done
done
