#!/bin/bash
# A/B of the priority-stagger variants of qc_mfma32_ell.hip (profiles/build_variant.sh prioN ... -DQC_ELL_PRIO_MODE=N) at config 5
cd ${GRAFT_REPO_ROOT:-.}
./tests/hip/wg_placement 499 > gpurun_out/r05_wg_placement.txt 2>&1
for r in 1 2 3; do for v in "" prio1 prio2 prio3 prio4 prio5; do echo -n "variant=[$v] "; QCOLLOC_HIP_VARIANT=$v python profiles/c5_times.py 500 2>/dev/null | tail -1; done; done
