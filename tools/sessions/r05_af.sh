# round 5, session af: config 5 lost 5 % to the ticket atomics of the best-row select (2048 blocks per sample on one address): capped at 64 blocks per sample
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_cyclegan.py tests/test_gpu_refine.py -q 2>&1 | tail -2 > gpurun_out/r05_af_tests.log
python tools/step_ab.py cyclegan256 > gpurun_out/r05_af_step.log 2>&1
python tools/step_ab.py dcgan64 >> gpurun_out/r05_af_step.log 2>&1
python tools/step_ab.py mnist >> gpurun_out/r05_af_step.log 2>&1
python tools/step_ab.py dcgan32 >> gpurun_out/r05_af_step.log 2>&1
