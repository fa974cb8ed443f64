#!/bin/bash
# MFMA utilisation counters for the large-n case (config 5, MFMA32 kernel) and the metric workload (config 3):
#   gpurun -- 'bash profiles/collect_mfma.sh'
# Counters in their own pass (no trace domains besides the implicit kernel dispatch records).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_mfma_util
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in 3 5; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cfg$cfg -- \
      python3 $R/bench.py --config $cfg --steps 100 --warmup 10 --cpu-seconds 0 > $OUT/bench_cfg$cfg.json 2> $OUT/cfg$cfg.log
done
python3 - <<PY
import csv, glob, json, collections
res = {}
for cfg in (3, 5):
    for f in glob.glob("$OUT/cfg%d/**/*counter_collection.csv" % cfg, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "qc_mfma" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        d = {k: sum(v) / len(v) for k, v in acc.items()}
        # MfmaUtil (rocprofv3 derived-metric formula): MFMA busy cycles / (GRBM_GUI_ACTIVE * SIMDs), SIMD_NUM = 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            d["MfmaUtil_percent"] = 100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
        res["config%d" % cfg] = d
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
