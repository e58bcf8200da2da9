#!/bin/bash
# round 6, session g: the norm backward's column sums in the producing backward-data epilogue (cgs_*_bwd_data_nstats): parity, then whole-step A/B
mkdir -p gpurun_out/r06_g
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "norm_backward or leaves_the_norm or fused_statistics or bn_train or group_statistics" > gpurun_out/r06_g/ops.log 2>&1; tail -5 gpurun_out/r06_g/ops.log
python -m pytest tests/test_gpu_refine.py tests/test_gpu_cyclegan.py tests/test_gpu_class_fused.py tests/test_gpu_fullsize.py tests/test_gpu_shaping.py -x -q -m gpu > gpurun_out/r06_g/engine.log 2>&1; tail -5 gpurun_out/r06_g/engine.log
for a in cyclegan256 mnist dcgan32 dcgan64; do LB_AB="CGS_NO_FUSED_BN_BWD_STATS=1;CGS_NO_FUSED_BN_BWD_STATS=-" LB_REPS=5 python tools/step_ab.py $a >> gpurun_out/r06_g/step_ab.txt 2>&1; done
LB_AB="CGS_NO_FUSED_BN_BWD_STATS=1;CGS_NO_FUSED_BN_BWD_STATS=-" LB_REPS=5 python tools/step_ab.py dcgan64 64 1 >> gpurun_out/r06_g/step_ab.txt 2>&1
LB_AB="CGS_NO_FUSED_BN_BWD_STATS=1;CGS_NO_FUSED_BN_BWD_STATS=-" LB_REPS=5 python tools/step_ab.py mnist 64 1 >> gpurun_out/r06_g/step_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06_g/step_ab.txt
