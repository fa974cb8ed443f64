cd ${GRAFT_REPO_ROOT:-.}
for fl in 0 4 8 12 16 0 8; do echo "QC_ELL_FLAGS=$fl"; QC_ELL_FLAGS=$fl python profiles/c5_times.py 500 2>&1 | grep "T="; done
