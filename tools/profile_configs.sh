# rocprofv3 kernel tables of the non-headline configurations (run on the GPU box through gpurun; results under gpurun_out/$1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-cfg}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for A in mnist dcgan32 cyclegan256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$A -o $A -- python3 $R/bench.py --arch $A --no-graph --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > $O/${A}_bench.log 2>&1
  cp $O/$A/*/${A}_kernel_stats.csv $O/${A}_kernel_stats.csv 2>/dev/null || find $O/$A -name "*kernel_stats.csv" -exec cp {} $O/${A}_kernel_stats.csv \;
  find $O/$A -name "*kernel_trace.csv" -delete
done
cd $R
for A in mnist dcgan32 cyclegan256; do python bench.py --arch $A --no-cpu-baseline --no-other-configs > $O/${A}_default.log 2>&1; done
ls -la $O
