# round 4, session c: the split-bf16 mode through the goldens, the fuzz shapes and the full-size oracle cases; then the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_c; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests/test_gpu_refine.py tests/test_gpu_fuzz.py tests/test_gpu_ops.py -m gpu -q -x ) > $O/pytest_a.log 2>&1; tail -n 6 $O/pytest_a.log
( time timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "dcgan64 or cyclegan or bx6" -s ) > $O/pytest_full.log 2>&1; grep -E "passed|failed|agreement" $O/pytest_full.log | tail -n 30
( time python bench.py --no-cpu-baseline ) > $O/bench.log 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_c/bench.log") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_executed_frac"])
b = d["bx6"]; print("bx6", b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["frac"], b["roofline"]["step_executed_frac"])
for k, v in b["kernels"].items(): print("   ", k, v)
PY
