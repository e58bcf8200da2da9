#!/bin/bash
# round 6, session z: finalize + apply of a norm in ONE launch where its group owns <= 256 partial rows (norm_fa_kernel): parity, then same-process A/B
mkdir -p gpurun_out/r06_z
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -x -q -m gpu -k "statistics or norm or bn_" 2>&1 | tail -2
python -m pytest tests/test_gpu_refine.py tests/test_gpu_cyclegan.py tests/test_gpu_class_fused.py tests/test_gpu_shaping.py tests/test_gpu_sync_bn.py -x -q -m gpu 2>&1 | tail -2
E=collaborative-gan-sampling_amd/libcgs_exp.so
for cfg in "dcgan32 64 1" "dcgan64 64 1" "mnist 64 1" "dcgan32 256 1" "mnist 64 32" "dcgan32 256 8" "cyclegan256 8 1" "dcgan64 1024 1"; do
  set -- $cfg
  CGS_LIB=$E LB_AB="CGS_NORM_FA=0;CGS_NORM_FA=1" LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_z/step_ab.txt
done
cat gpurun_out/r06_z/step_ab.txt
