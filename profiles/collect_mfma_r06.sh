#!/bin/bash
# MFMA utilisation counters per kernel for BASELINE config 5 (the large-n case north_star asks it for): the sparse-drive kernels that
# serve it (qc_mfma32_ell.hip) and, with QC_NO_ELL=1, the dense-image kernels they replaced; config 3 beside them.
#   gpurun -- 'bash profiles/collect_mfma_r06.sh'
# Counters in their own pass (no trace domains).  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_mfma_util_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in cfg3 cfg5 cfg5dense; do
  cfg=${tag:3:1}
  if [ $tag = cfg5dense ]; then export QC_NO_ELL=1; else unset QC_NO_ELL; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$tag -- \
      python3 $R/bench.py --config $cfg --steps 100 --warmup 10 --cpu-seconds 0 --prewarm-seconds 0 --no-host-visible --no-config5 > $OUT/bench_$tag.json 2> $OUT/$tag.log
done
unset QC_NO_ELL
# round 6: the exponential integrator's launches (F + dF, mu_d2F) at config 3's size
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/exp3 -- \
    python3 $R/profiles/exp_bench.py 3 > $OUT/bench_exp3.txt 2> $OUT/exp3.log
python3 - <<PY
import csv, glob, json, collections
res = {}
for tag, label in (("cfg3", "config3"), ("cfg5", "config5"), ("cfg5dense", "config5 (QC_NO_ELL=1)"), ("exp3", "config3 exponential")):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "qc_mfma" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:r["Kernel_Name"].rfind(">(") + 1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            d = {c: sum(v) / len(v) for c, v in cs.items()}
            d["launches"] = len(next(iter(cs.values())))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
                d["MfmaUtil_percent"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
            res["%s %s" % (label, k.replace("void (anonymous namespace)::", ""))] = d
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
