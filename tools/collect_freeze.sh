#!/bin/bash
# After `bash tools/freeze_profiles.sh <tag>` ran on the GPU box (results merged into gpurun_out/<tag>/): copy what is judged into profiles/
# under <tag>_* names and rebuild profiles/traffic.json (headline table + one table per configuration + the split-bf16 kernels).
#   bash tools/collect_freeze.sh r05_j
set -e
T=${1:?tag}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$T
cd $R
python tools/make_traffic_json.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv ${T}_final $O/fetch_bx6/f_counter_collection.csv $O/write_bx6/w_counter_collection.csv \
  --arch mnist $O/fetch_mnist/f_counter_collection.csv $O/write_mnist/w_counter_collection.csv \
  --arch dcgan32 $O/fetch_dcgan32/f_counter_collection.csv $O/write_dcgan32/w_counter_collection.csv \
  --arch cyclegan256 $O/fetch_cyclegan256/f_counter_collection.csv $O/write_cyclegan256/w_counter_collection.csv > $O/traffic_summary.txt
cp $O/s1/s1_kernel_stats.csv profiles/${T}_final_streams1_kernel_stats.csv; grep "^{" $O/s1_bench.log > profiles/${T}_final_streams1_bench.log
cp $O/s2/s2_kernel_stats.csv profiles/${T}_final_default_kernel_stats.csv; grep "^{" $O/s2_bench.log > profiles/${T}_final_default_bench.log
cp $O/bx6/bx6_kernel_stats.csv profiles/${T}_final_bx6_streams1_kernel_stats.csv; grep "^{" $O/bx6_bench.log > profiles/${T}_final_bx6_streams1_bench.log
for A in mnist dcgan32 cyclegan256; do
  cp $O/$A/${A}_kernel_stats.csv profiles/${T}_final_${A}_streams1_kernel_stats.csv; grep "^{" $O/${A}_bench.log > profiles/${T}_final_${A}_streams1_bench.log
done
cp $O/shaping/shaping_kernel_stats.csv profiles/${T}_shaping_kernel_stats.csv; grep "^{" $O/shaping_bench.log > profiles/${T}_shaping_bench.log
grep "^{" $O/bench_default.log > profiles/${T}_final_default_hipgraph_bench.log
# the full records behind the compact lines (round 6: the line names its sidecar; kept per run)
cp $O/bench_default_detail.json profiles/${T}_final_default_hipgraph_bench_detail.json
cp $O/s1_bench_detail.json profiles/${T}_final_streams1_bench_detail.json
cp $O/bx6_bench_detail.json profiles/${T}_final_bx6_streams1_bench_detail.json
for A in mnist dcgan32 cyclegan256; do cp $O/${A}_bench_detail.json profiles/${T}_final_${A}_streams1_bench_detail.json; done
python - "$T" <<'PY'
import csv, sys
T = sys.argv[1]
for src, dst in ((f"gpurun_out/{T}/sq/sq_counter_collection.csv", f"profiles/{T}_layerbench_pmc_sq.csv"), (f"gpurun_out/{T}/sq_bx6/sq_counter_collection.csv", f"profiles/{T}_layerbench_bx6_pmc_sq.csv")):
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("igemm", "conv_patch", "convt_rows", "convt_quad"))]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(keep)
PY
ls profiles | grep "^${T}_"
