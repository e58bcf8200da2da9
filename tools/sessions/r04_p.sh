# round 4, session p: sign-mask output of the split-bf16 epilogue: parity, engine wiring, the bx6 line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_p; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py -m gpu -q -x -k "sign or golden" ) > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
for i in 1 2; do python bench.py --contraction bx6 --no-cpu-baseline --no-other-configs 2>>$O/stderr.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bx6', d['value'], d['ms_per_step'], [ (k, v['avg_us']) for k, v in d['hbm'].items() if 'patch2' in k])"; done | tee $O/bench.log
