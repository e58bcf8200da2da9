# session start: GPU suite, cyclegan256 rows-per-launch sweep, headline (one gpurun call)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sa; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d.get("algorithmic_tflops"))'
for G in 1 2 4 8; do for S in 1 2 4; do echo -n "cyclegan256 fuse $G streams $S: "; python bench.py --arch cyclegan256 --fuse $G --streams $S --steps 8 --no-cpu-baseline --no-other-configs 2>$O/err.log | python -c "$J"; done; done 2>&1 | tee $O/cg_sweep.log
python bench.py --no-cpu-baseline --no-other-configs > $O/head.log 2>&1; python -c "$J" < $O/head.log
