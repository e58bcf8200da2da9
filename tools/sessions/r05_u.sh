# round 5, session u: the whole GPU suite + the default bench line on the round's current sources
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r05_u_fullsuite.log
python bench.py > gpurun_out/r05_u_bench.log 2> gpurun_out/r05_u_bench.err
