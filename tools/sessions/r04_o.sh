# round 4, session o: norm apply passes with the per-channel parameters hoisted out of the stream loop: parity, then the configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_o; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_sync_bn.py tests/test_gpu_cyclegan.py -m gpu -q -x -k "bn or norm or statistics or cyclegan or sync" ) > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get("hbm",{}); print(d["value"], d["ms_per_step"], {k: (v["avg_us"], v["frac"]) for k, v in h.items() if "bn" in k or "norm" in k})'
for a in dcgan64 mnist dcgan32 cyclegan256; do echo -n "$a: "; python bench.py --arch $a --no-cpu-baseline --no-other-configs 2>>$O/stderr.log | python -c "$J"; done | tee $O/bench.log
