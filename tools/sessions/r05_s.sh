# round 5, session s: kernel tables of the batch-64 calls (one 64-image batch per call, hipGraph replays)
cd /tmp && export TMPDIR=/tmp
for A in mnist dcgan32; do
  LB_ITERS=5 LB_REPS=2 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r05_s_prof -o ${A}_b64 -- python3 $GRAFT_REPO_ROOT/tools/step_ab.py $A 64 1 > $GRAFT_REPO_ROOT/gpurun_out/r05_s_${A}.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/kernel_table.py $GRAFT_REPO_ROOT/gpurun_out/r05_s_prof/${A}_b64_results.db > $GRAFT_REPO_ROOT/gpurun_out/r05_s_table_${A}.txt 2>&1
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05_s_prof
