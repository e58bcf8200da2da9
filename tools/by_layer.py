#!/usr/bin/env python3
"""Print the per-(kernel, layer) table of a `python bench.py --by-layer ...` JSON line: launches, avg us, issued TFLOP/s, share of the step.
    python tools/by_layer.py gpurun_out/<log> [...]"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads([l for l in open(path) if l.startswith("{")][0])
    print(f"== {path}: {d['value']} samples/s, {d['ms_per_step']} ms/step, step_executed_frac {d['roofline']['step_executed_frac']}")
    rows = [(v["share_of_step"], k, v) for k, v in d["kernels"].items()]
    tot = 0.0
    for sh, k, v in sorted(rows, reverse=True):
        peak = 2500.0 / 6 if k.startswith("igemm_bx6") else 157.3
        print(f"  {sh:6.3f}  {v['launches']:5d} x {v['avg_us']:8.1f} us  {v['tflops']:7.1f} TF ({v['tflops'] / peak:5.3f})  nominal {v['nominal_tflops']:7.1f}  {k}")
        tot += sh
    hb = sum(v["share_of_step"] for v in d.get("hbm", {}).values())
    print(f"  contraction share {tot:.3f}; hbm-bound share {hb:.3f}")
    for k, v in sorted(d.get("hbm", {}).items(), key=lambda kv: -kv[1]["share_of_step"]):
        print(f"     hbm {v['share_of_step']:6.3f} {v['launches']:5d} x {v['avg_us']:8.1f} us  {v['achieved']:7.1f} GB/s ({v['frac']:.3f})  {k}")
