# round 5, session a: per-layer tables of every configuration (one eager single-stream step, HIP events per launch, keyed by kernel + layer)
cd $GRAFT_REPO_ROOT
for A in mnist dcgan32 cyclegan256 dcgan64; do
  python bench.py --arch $A --by-layer --no-cpu-baseline --no-other-configs --steps 4 --warmup 1 > gpurun_out/r05_layers_$A.log 2> gpurun_out/r05_layers_$A.err
done
python tools/by_layer.py gpurun_out/r05_layers_*.log > gpurun_out/r05_layers.txt
