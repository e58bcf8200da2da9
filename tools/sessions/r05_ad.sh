# round 5, session ad: the whole GPU suite on the round's last sources, then the frozen profiles (tools/freeze_profiles.sh r05_ad)
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05_ad_fullsuite.log
bash tools/freeze_profiles.sh r05_ad > gpurun_out/r05_ad_freeze_inner.log 2>&1
