# round 5, session w: the whole GPU suite again with its complete output kept (session u died with a fatal error somewhere in the middle)
cd $GRAFT_REPO_ROOT
python -m pytest tests -v -m gpu -x > gpurun_out/r05_w_fullsuite_full.log 2>&1
echo "rc=$?" >> gpurun_out/r05_w_fullsuite_full.log
dmesg 2>/dev/null | tail -20 > gpurun_out/r05_w_dmesg.log
