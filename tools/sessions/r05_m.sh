# round 5, session m: logical batches per launch for mnist / dcgan32 with the round's tile plans (one box, back to back)
cd $GRAFT_REPO_ROOT
for G in 16 24 32 40 48 64; do
  python bench.py --arch mnist --fuse $G --no-cpu-baseline --no-other-configs --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mnist fuse', $G, d['value'], d['roofline']['step_executed_frac'])"
done > gpurun_out/r05_m_fuse.log
for G in 4 8 12 16; do
  python bench.py --arch dcgan32 --fuse $G --no-cpu-baseline --no-other-configs --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dcgan32 fuse', $G, d['value'], d['roofline']['step_executed_frac'])"
done >> gpurun_out/r05_m_fuse.log
