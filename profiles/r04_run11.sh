cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
for fl in 0 1 0 1; do echo "QC_ELL_FLAGS=$fl"; QC_ELL_FLAGS=$fl python profiles/c5_times.py 500 1000 2>&1 | grep "T="; done
QC_ELL_FLAGS=0 python profiles/stamps_ell32.py fused 500 2>&1 | grep -v amdgpu
QC_ELL_FLAGS=1 python profiles/stamps_ell32.py fused 500 2>&1 | grep -v amdgpu
