# round 4, session e: LDS-DMA weights (default build) vs register-staged (v0), the 256x64 config; operator parity first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -m gpu -q -x -k "bx6" ) > $O/pytest_ops.log 2>&1; tail -n 4 $O/pytest_ops.log
for v in hip; do CGS_LIB=$R/collaborative-gan-sampling_amd/libcgs_$v.so python tools/bx6_bench.py 10 2>&1 | grep -v amdgpu.ids; done | tee $O/bench.log
for st in 2; do python bench.py --contraction bx6 --streams $st --no-cpu-baseline --no-other-configs 2>>$O/stderr.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bx6 streams $st', d['value'], d['ms_per_step'], d['roofline']['step_executed_frac'])"; done | tee -a $O/bench.log
