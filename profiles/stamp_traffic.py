#!/usr/bin/env python3
"""Turns a profiles/collect.sh summary (gpurun_out/prof_<tag>/summary.json) into profiles/pmc_traffic.json, stamped with the commit and
with the sha256 of the metric kernel's sources (bench.py's kernel_source_hash): bench.py drops `roofline.traffic` when the sources have
changed since.  Run in the repository (needs git), right after the GPU call, BEFORE touching the kernel sources again:
    python profiles/stamp_traffic.py gpurun_out/prof_r05a/summary.json r05a"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for rel in ("quantumcollocation.jl_amd/csrc/qc_mfma_kernels.hip", "quantumcollocation.jl_amd/csrc/qc_mfma_common.h",
                "quantumcollocation.jl_amd/csrc/qc_internal.h"):      # = bench.py KERNEL_SOURCES
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


summary, tag = sys.argv[1], sys.argv[2]
res = json.load(open(summary))
hits = [(k, v) for k, v in res.items() if k.startswith("void (anonymous namespace)::qc_mfma16_pade4_kernel<true") or "qc_mfma16_pade4_kernel<true" in k]
hits = [(k, v) for k, v in hits if "hbm_bytes_per_launch_corrected" in v]
assert hits, "no traced qc_mfma16_pade4_kernel<true, ...> with both counters in " + summary
k, v = max(hits, key=lambda kv: kv[1].get("calls", 0))
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "quantumcollocation.jl_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
rec = {"kernel": "mfma", "config": 3, "T": 1000, "hbm_bytes_per_launch": v["hbm_bytes_per_launch_corrected"], "hbm_write_bytes": v["hbm_write_bytes"],
       "hbm_read_bytes_corrected": v["hbm_read_bytes_corrected"], "kernel_avg_us_under_rocprofv3": v.get("avg_us"), "traced_kernel": k,
       "commit": commit + ("+uncommitted csrc changes" if dirty else ""), "kernel_source_sha256": kernel_source_hash(),
       "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python bench.py --steps 300 --warmup 30 "
                 "--prewarm-seconds 0 --no-host-visible`, profiles/collect.sh; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)"}
json.dump(rec, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
