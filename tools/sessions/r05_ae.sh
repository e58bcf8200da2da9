# round 5, session ae: the plain default bench line once more, now that profiles/traffic.json belongs to these sources (roofline.traffic filled in)
cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/r05_ae_bench.log 2> gpurun_out/r05_ae_bench.err
