# cyclegan256 per-layer kernel records at 8 and 32 rows per launch (eager, one stream) + rocprof kernel stats at 32
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sb; mkdir -p $O; cd $R
for G in 1 4; do python bench.py --arch cyclegan256 --fuse $G --streams 1 --no-graph --by-layer --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/cg_f$G.log 2>$O/err_$G.log; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4 -o p4 -- python3 $R/bench.py --arch cyclegan256 --fuse 4 --no-graph --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $O/p4_bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete
ls -R $O | head -30
