# round 4, session h: the whole GPU suite after the split-bf16 work, then config 5 in both contraction modes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_h; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["step_executed_frac"])'
