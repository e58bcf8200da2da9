# round 4, session a: the GPU suite (new class-surface tests first), then the driver's default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_a; mkdir -p $O; cd $R
( time python -m pytest tests/test_gpu_refine.py -m gpu -x -q ) > $O/pytest_refine.log 2>&1; tail -3 $O/pytest_refine.log
( time python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_refine.py ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( time python bench.py ) > $O/bench.log 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_a/bench.log") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_executed_frac"])
for k, v in d["other_configs"].items():
    print(k, v["samples_per_s"], v.get("cpu_baseline", {}).get("value"), (v.get("roofline") or {}).get("step_executed_frac"))
print(json.dumps(d["class_surface"], indent=1))
PY
