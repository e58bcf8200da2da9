#!/bin/bash
# round 6, session l: does the PRESENCE of the norm-backward-sums branch in the igemm epilogue cost the launches that do not take it?
# (the product library against a variant with the branch compiled out, the fusion switched off in both: alternating processes on one box)
mkdir -p gpurun_out/r06_l
V=collaborative-gan-sampling_amd/libcgs_nons.so
for rep in 1 2 3; do
  for lib in product variant; do
    if [ $lib = variant ]; then export CGS_LIB=$V; else unset CGS_LIB; fi
    CGS_NSTAT_MAX_MB=0 python bench.py --no-graph --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs --detail "" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'])" >> gpurun_out/r06_l/ab.txt
  done
done
cat gpurun_out/r06_l/ab.txt
