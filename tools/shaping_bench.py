#!/usr/bin/env python3
"""Row f2 stand-alone (what bench.py's `shaping` object holds): the reference's D-shaping iteration at batch 64 -- probabilistic refine + one Adam
step of D + refresh -- for mnist and dcgan64; one JSON line.  Under `rocprofv3 --kernel-trace --stats` it gives the kernel table of the
shaping loop (wgrad_kernel, wgrad_reduce_kernel, adam_kernel, ... next to the refinement kernels): profiles/r05_*_shaping_kernel_stats.csv."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

print(json.dumps(bench.shaping_record(torch.device("cuda:0"))))
