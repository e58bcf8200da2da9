# round 5, session x: config 5 (cyclegan256, 8 per GPU): several 8-sample batches per launch (instance norm: per-sample statistics, so a launch of G x 8 is G reference batches exactly)
cd $GRAFT_REPO_ROOT
for cfg in "1 4" "2 2" "4 1" "4 2" "2 4"; do
  set -- $cfg
  echo "== fuse $1 streams $2" >> gpurun_out/r05_x_cyclegan_fuse.log
  python bench.py --arch cyclegan256 --fuse $1 --streams $2 --steps 8 --no-cpu-baseline --no-other-configs 2>&1 | grep "^{\|Error\|error" | cut -c1-1200 >> gpurun_out/r05_x_cyclegan_fuse.log
done
