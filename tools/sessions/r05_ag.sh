# round 5, session ag: the frozen profiles of the round's last sources (tools/freeze_profiles.sh r05_ag)
cd $GRAFT_REPO_ROOT
bash tools/freeze_profiles.sh r05_ag > gpurun_out/r05_ag_freeze_inner.log 2>&1
