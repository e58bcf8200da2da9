# round 4, session g: logical batches per launch x engine calls in flight for the small configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_g; mkdir -p $O; cd $R
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["step_executed_frac"])'
for G in 32 48 64 96; do for S in 2 4; do echo -n "mnist fuse $G streams $S: "; python bench.py --arch mnist --fuse $G --streams $S --steps 8 --no-cpu-baseline --no-other-configs 2>$O/err.log | python -c "$J"; done; done 2>&1 | tee $O/sweep.log
for G in 8 12 16 24; do for S in 2 4; do echo -n "dcgan32 fuse $G streams $S: "; python bench.py --arch dcgan32 --fuse $G --streams $S --steps 8 --no-cpu-baseline --no-other-configs 2>$O/err.log | python -c "$J"; done; done 2>&1 | tee -a $O/sweep.log
