# round 4, session l: mnist logical batches per launch chosen so that the conv grids fill whole rounds of block slots (49 G blocks on 1280 slots)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_l; mkdir -p $O; cd $R
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["step_executed_frac"])'
for G in 13 20 24 25 26 27 28 32 39 52; do echo -n "mnist fuse $G streams 4: "; python bench.py --arch mnist --fuse $G --streams 4 --steps 10 --no-cpu-baseline --no-other-configs 2>$O/err.log | python -c "$J"; done 2>&1 | tee $O/sweep.log
