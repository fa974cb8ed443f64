#!/bin/bash
# MFMA counters of the MFMA-bound kernels (exponential integrators, Hessians): gpurun -- 'bash profiles/collect_mfma_exp.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_mfma_exp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- \
    python3 $R/profiles/time_variants.py > $OUT/variants.txt 2> $OUT/log.txt
python3 - <<PY
import csv, glob, json, collections, re
res = {}
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        mm = re.search(r"(qc_mfma\w+(<[^>]*>)?)", k)
        if mm:
            acc[mm.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["launches"] = len(next(iter(cs.values())))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            d["MfmaUtil_percent"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
        res[k] = d
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
