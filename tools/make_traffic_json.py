#!/usr/bin/env python3
"""rocprofv3 PMC passes -> profiles/traffic.json (what bench.py reports as roofline.traffic).

    python tools/make_traffic_json.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <tag> [<FETCH csv 2> <WRITE csv 2>]
                                      [--arch <name> <FETCH csv> <WRITE csv>] ...
(the optional second pair: the same passes of `bench.py --contraction bx6`, whose igemm_bx6 kernels are added to the table;
 every --arch triple: the same two passes of `bench.py --arch <name>` -> the table `_by_arch[<name>]`, which bench.py's `other_configs.<name>`
 reads: round 5, no benched configuration without counter traffic)

Per kernel name: launches, average FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 prints them) and the HBM-side bytes per launch
= 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, HBM section:
the loads are 16 B per lane).  Also writes trimmed copies of the two CSVs (conv-family, batch-norm, momentum-update and logit-head kernels) under profiles/."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("igemm_kernel", "igemm_ns_kernel", "igemm_bx6", "convt_quad", "convt_rows", "conv_patch", "conv_smalln", "convt_smalln", "bn_", "refine_update", "linear_out1")


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def load(path, counter):
    acc, rows = collections.defaultdict(list), []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        if any(k in r["Kernel_Name"] for k in KEEP):
            rows.append(r)
    return acc, rows


def table(F, W):
    out = {}
    for k in F:
        if not any(s in k for s in KEEP) and "pack" not in k:
            continue
        f, w = sum(F[k]) / len(F[k]), sum(W.get(k, [0.0])) / max(1, len(W.get(k, [])))
        out[k] = {"launches": len(F[k]), "FETCH_SIZE_KB_avg": round(f, 1), "WRITE_SIZE_KB_avg": round(w, 1),
                  "hbm_bytes_per_launch_corrected": int(round((2 * f + w) * 1024))}
    return out


def main():
    by_arch = {}
    while "--arch" in sys.argv:
        i = sys.argv.index("--arch")
        name, fp, wp = sys.argv[i + 1:i + 4]
        del sys.argv[i:i + 4]
        Fa, _ = load(fp, "FETCH_SIZE")
        Wa, _ = load(wp, "WRITE_SIZE")
        by_arch[name] = table(Fa, Wa)
    fpath, wpath, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    F, frows = load(fpath, "FETCH_SIZE")
    W, wrows = load(wpath, "WRITE_SIZE")
    if len(sys.argv) > 5:            # the split-bf16 run: only its own kernels are taken from it
        F2, frows2 = load(sys.argv[4], "FETCH_SIZE")
        W2, wrows2 = load(sys.argv[5], "WRITE_SIZE")
        for k in F2:
            if "igemm_bx6" in k:
                F[k], W[k] = F2[k], W2.get(k, [0.0])
        frows += [r for r in frows2 if "igemm_bx6" in r["Kernel_Name"]]
        wrows += [r for r in wrows2 if "igemm_bx6" in r["Kernel_Name"]]
    out = table(F, W)
    sys.path.insert(0, ROOT)
    from cgs_amd.lib import source_hash  # (== the stamp embedded in the library built from this tree: lib.built_from())
    out["_source_sha256"] = source_hash()        # bench.py prints roofline.traffic = null once the kernel sources move on
    out["_tag"] = tag
    out["_by_arch"] = by_arch
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    out.pop("_source_sha256"); out.pop("_tag"); out.pop("_by_arch")
    for rows, name in ((frows, "fetch_size"), (wrows, "write_size")):
        with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{name}.csv"), "w", newline="") as fh:
            wr = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            wr.writeheader(); wr.writerows(rows)
    for k, v in out.items():
        print(f"{k:50s} x{v['launches']:4d}  {v['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB / launch")
    for a, t in by_arch.items():
        for k, v in t.items():
            print(f"[{a}] {k:50s} x{v['launches']:4d}  {v['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB / launch")


if __name__ == "__main__":
    main()
