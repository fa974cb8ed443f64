cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu --maxfail=40 -x > gpurun_out/r04/run4_tests.txt 2>&1
tail -25 gpurun_out/r04/run4_tests.txt
for k in fused jac hess; do python profiles/stamps_ell32.py $k 500 > gpurun_out/r04/ell32_timeline_$k.txt 2>&1; cat gpurun_out/r04/ell32_timeline_$k.txt; done
