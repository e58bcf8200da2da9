set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o s1 -- python3 $R/bench.py --no-graph --streams 1 --no-cpu-baseline --no-other-configs > $O/s1_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s2 -o s2 -- python3 $R/bench.py --no-graph --no-cpu-baseline --no-other-configs > $O/s2_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d32 -o d32 -- python3 $R/tools/layer_bench.py dcgan32 256 > $O/d32_layer.log 2>&1
cd $R
python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-300
ls -la $O $O/*/ | head -50
# keep only what is needed (size limit)
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
