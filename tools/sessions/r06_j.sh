#!/bin/bash
# round 6, session j: split-K / tail reduce passes with their slice loads in flight (igemm.hip) -- previous library against the new one, alternating processes
mkdir -p gpurun_out/r06_j
P=collaborative-gan-sampling_amd/libcgs_prev.so
for rep in 1 2; do
for cfg in "dcgan32 64 1" "dcgan64 64 1" "mnist 64 1" "dcgan32 256 1" "mnist 64 32"; do
  set -- $cfg
  echo "--- previous library" >> gpurun_out/r06_j/step_ab.txt
  CGS_LIB=$P LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_j/step_ab.txt
  echo "--- new library" >> gpurun_out/r06_j/step_ab.txt
  LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_j/step_ab.txt
done
done
cat gpurun_out/r06_j/step_ab.txt
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
