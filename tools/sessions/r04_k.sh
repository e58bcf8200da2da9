# round 4, session k: do the kernels of the four mnist / dcgan32 engine calls in flight overlap?  (kernel trace of the default runs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for A in mnist; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$A -o t -- python3 $R/bench.py --arch $A --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > $O/$A.log 2>&1
  f=$(find $O/$A -name "t_kernel_trace.csv" | head -1)
  echo "== $A"; python3 $R/tools/trace_overlap.py $f
  rm -f $f
done
