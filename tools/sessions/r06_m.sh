#!/bin/bash
# round 6, session m: igemm_kernel / igemm_ns_kernel twins (product) against the variant without the sums branch (= the kernel of the rounds before)
mkdir -p gpurun_out/r06_m
V=collaborative-gan-sampling_amd/libcgs_nons.so
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "norm_backward or leaves_the_norm or conv2d_fwd or bwd_data" 2>&1 | tail -2
for rep in 1 2 3; do
  for lib in product variant; do
    if [ $lib = variant ]; then export CGS_LIB=$V CGS_NSTAT_MAX_MB=0; else unset CGS_LIB; export CGS_NSTAT_MAX_MB=48; fi
    python bench.py --no-graph --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --detail "" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'], r['step_executed_frac'])" >> gpurun_out/r06_m/ab.txt
  done
done
cat gpurun_out/r06_m/ab.txt
unset CGS_LIB CGS_NSTAT_MAX_MB
AB="CGS_NSTAT_MAX_MB=0;CGS_NSTAT_MAX_MB=48;CGS_NSTAT_MAX_MB=100000"
for cfg in "dcgan64 1024 1" "cyclegan256 8 1" "dcgan64 64 1" "dcgan32 256 1" "mnist 64 32" "dcgan32 256 8"; do
  set -- $cfg
  LB_AB="$AB" LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_m/step_ab.txt
done
cat gpurun_out/r06_m/step_ab.txt
