#!/bin/bash
# round 6, session k: the fused backward sums with the epilogue's x loads issued a row group ahead -- threshold sweep again + the dominant kernel's duration
mkdir -p gpurun_out/r06_k
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "norm_backward or leaves_the_norm" 2>&1 | tail -2
AB="CGS_NSTAT_MAX_MB=0;CGS_NSTAT_MAX_MB=48;CGS_NSTAT_MAX_MB=100000"
for cfg in "dcgan64 1024 1" "cyclegan256 8 1" "dcgan32 256 8" "dcgan64 64 1" "dcgan32 256 1"; do
  set -- $cfg
  LB_AB="$AB" LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_k/step_ab.txt
done
cat gpurun_out/r06_k/step_ab.txt
for mb in 0 100000; do
  CGS_NSTAT_MAX_MB=$mb python bench.py --no-graph --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --detail gpurun_out/r06_k/detail_$mb.json | cut -c1-1700
done
