# End-of-round evidence, one gpurun call:  bash tools/freeze_profiles.sh <tag>   (e.g. r03_e)  -> gpurun_out/<tag>/
# kernel stats (one and two batches in flight, eager launches: the profiled runs change no kernel), the two PMC passes behind
# profiles/traffic.json, the SQ counters of the layer sweep, the split-bf16 mode, the small configurations, and the plain default bench line.
set -x
T=${1:-r04_a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/bench.py --detail $O/fetch_detail.json --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/bench.py --detail $O/write_detail.json --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -o s1 -- python3 $R/bench.py --detail $O/s1_bench_detail.json --no-graph --streams 1 --no-cpu-baseline --no-other-configs > $O/s1_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s2 -o s2 -- python3 $R/bench.py --detail $O/s2_bench_detail.json --no-graph --no-cpu-baseline --no-other-configs > $O/s2_bench.log 2>&1
LB_ITERS=3 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o sq -- python3 $R/tools/layer_bench.py dcgan64 1024 > $O/sq_layer.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_bx6 -o f -- python3 $R/bench.py --detail $O/fetch_bx6_detail.json --contraction bx6 --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/fetch_bx6.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_bx6 -o w -- python3 $R/bench.py --detail $O/write_bx6_detail.json --contraction bx6 --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/write_bx6.log 2>&1
# the opt-in split-bf16 contraction: kernel table (one batch in flight, eager) and the SQ counters of its kernels on the layer sweep
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bx6 -o bx6 -- python3 $R/bench.py --detail $O/bx6_bench_detail.json --contraction bx6 --no-graph --streams 1 --no-cpu-baseline --no-other-configs > $O/bx6_bench.log 2>&1
LB_ITERS=3 CGS_CONTRACTION=bx6 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq_bx6 -o sq -- python3 $R/tools/layer_bench.py dcgan64 1024 > $O/sq_bx6_layer.log 2>&1
for A in mnist dcgan32 cyclegan256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$A -o $A -- python3 $R/bench.py --detail $O/${A}_bench_detail.json --arch $A --no-graph --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > $O/${A}_bench.log 2>&1
  # counter traffic of the configuration at its bench.py launch sizes (-> traffic.json `_by_arch`, tools/make_traffic_json.py --arch)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$A -o f -- python3 $R/bench.py --detail $O/fetch_${A}_detail.json --arch $A --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/fetch_$A.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$A -o w -- python3 $R/bench.py --detail $O/write_${A}_detail.json --arch $A --no-graph --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $O/write_$A.log 2>&1
done
# row f2: the kernel table of the D-shaping loop (weight gradients, Adam) at the reference's batch 64
rocprofv3 --kernel-trace --stats --output-format csv -d $O/shaping -o shaping -- python3 $R/tools/shaping_bench.py > $O/shaping_bench.log 2>&1
cd $R
( time python bench.py --detail $O/bench_default_detail.json ) > $O/bench_default.log 2>$O/bench_default.err
tail -1 $O/bench_default.log | cut -c1-300
# keep only what is needed (size limit): the per-dispatch traces of the stats runs are large and not used
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
ls $O $O/*/ | head -60
