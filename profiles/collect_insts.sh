#!/bin/bash
# Dynamic instruction counts per kernel (rocprofv3 --pmc, counters only): is a launch bound by instruction issue?
#   gpurun -- 'bash profiles/collect_insts.sh <tag> <script and args>'      e.g.  bash profiles/collect_insts.sh c5 profiles/c5_times.py 500
set -u
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_insts_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/a -- python3 $R/$* > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/b -- python3 $R/$* > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $R/$* > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qc_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:r["Kernel_Name"].rfind(">(") + 1].replace("void (anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, d in res.items():
    w = d.get("SQ_WAVES", 0) or 1
    print(k[:90])
    print("   per wave: " + "  ".join(f"{c[8:] if c.startswith('SQ_INSTS_') else c} {d[c] / w:8.1f}" for c in sorted(d) if c != "SQ_WAVES") + f"   waves {w:.0f}")
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
PY
