#!/bin/bash
# round 6, session w: 16-byte row copies in the best-sample select (elementwise.hip): previous library against the new one, alternating processes
mkdir -p gpurun_out/r06_w
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py -x -q -m gpu -k "select or golden or policy" 2>&1 | tail -2
P=collaborative-gan-sampling_amd/libcgs_prev.so
for rep in 1 2; do
for cfg in "cyclegan256 8 1" "mnist 64 32" "dcgan64 1024 1" "mnist 64 1"; do
  set -- $cfg
  echo "--- previous library" >> gpurun_out/r06_w/step_ab.txt
  CGS_LIB=$P LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_w/step_ab.txt
  echo "--- new library" >> gpurun_out/r06_w/step_ab.txt
  LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_w/step_ab.txt
done
done
cat gpurun_out/r06_w/step_ab.txt
