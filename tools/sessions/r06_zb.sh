#!/bin/bash
# round 6, session zb: igemm.hip built with other LLVM scheduling strategies (variant libraries: tools/build_variant.sh f<tag> igemm.hip "-mllvm ..."):
# average duration of every igemm instantiation in one eager single-stream run, alternating processes
mkdir -p gpurun_out/r06_zb
for rep in 1 2; do
  for lib in product filp fmemclause fnopost fiterilp; do
    if [ $lib = product ]; then unset CGS_LIB; else export CGS_LIB=collaborative-gan-sampling_amd/libcgs_$lib.so; fi
    python bench.py --no-graph --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --detail gpurun_out/r06_zb/d.json 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=json.load(open('gpurun_out/r06_zb/d.json'))['kernels']
print('$lib', d['ms_per_step'], ' | '.join(f\"{n.replace('igemm_kernel','ig')} {v['avg_us']}\" for n,v in sorted(k.items()) if n.startswith('igemm') and v['launches'] > 5))" >> gpurun_out/r06_zb/ab.txt
  done
done
cat gpurun_out/r06_zb/ab.txt
