# round 5, session ap: the whole GPU suite on the round's final sources
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r05_ap_fullsuite.log
