# round 5, session v: which test file of the GPU suite dies (session u: a fatal error in the middle of the run)
cd $GRAFT_REPO_ROOT
for f in test_gpu_abi_host test_gpu_fuzz test_gpu_fuzz_archs test_gpu_regressions test_gpu_ops_param_grads test_gpu_synthetic test_gpu_distribution test_gpu_bench_contract; do
  python -m pytest tests/$f.py -q -m gpu -x 2>&1 | grep -v "^  File\|Extension modules" | tail -25 > gpurun_out/r05_v_$f.log
done
