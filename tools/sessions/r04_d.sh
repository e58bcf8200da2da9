# round 4, session d: SQ counters of the split-bf16 kernel on the headline's layers (counters in their own pass)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LB_ITERS=3 CGS_CONTRACTION=bx6 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o sq -- python3 $R/tools/layer_bench.py dcgan64 1024 > $O/sq.log 2>&1
LB_ITERS=3 CGS_CONTRACTION=bx6 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -o sq2 -- python3 $R/tools/layer_bench.py dcgan64 1024 > $O/sq2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("sq", "sq2"):
    f = glob.glob(f"gpurun_out/r04_d/{tag}/**/*counter_collection.csv", recursive=True)
    if not f: print("no file", tag); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        if "igemm" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        n = sum(1 for r in rows if r["Kernel_Name"].split("(")[0] == k and r["Counter_Name"] == next(iter(v)))
        print(tag, k, n, {c: round(x / n) for c, x in v.items()})
PY
