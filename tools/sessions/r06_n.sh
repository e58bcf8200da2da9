#!/bin/bash
# round 6, session n: how much does the dominant kernel pay for epilogue branches it does not take?  scratch variants of igemm.hip (built outside the
# tree): xp1 = no LRELU / TANH / TANH_BWD cases; xp2 = xp1 without the sign-mask branch; xp3 = xp2 with the NONE case only (results of the variants are
# wrong where a launch needs a removed case: timing only)
mkdir -p gpurun_out/r06_n
for rep in 1 2 3; do
  for lib in product xp1 xp2 xp3; do
    if [ $lib = product ]; then unset CGS_LIB; else export CGS_LIB=collaborative-gan-sampling_amd/libcgs_$lib.so; fi
    python bench.py --no-graph --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --detail gpurun_out/r06_n/d.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=json.load(open('gpurun_out/r06_n/d.json'))['kernels']
print('$lib', d['ms_per_step'], ' | '.join(f\"{n.replace('igemm_kernel','ig').replace('igemm_ns_kernel','ig-ns')} {v['avg_us']}\" for n,v in k.items() if n.startswith('igemm')))" >> gpurun_out/r06_n/ab.txt
  done
done
cat gpurun_out/r06_n/ab.txt
