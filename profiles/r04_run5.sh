cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse_drive or dense_drives_keep or mfma32 or config_hessian_parity or full_size or config5 or kernel_names or fused_launch or config_parity or compact_host or new_x" > gpurun_out/r04/run5_tests.txt 2>&1
tail -5 gpurun_out/r04/run5_tests.txt
python profiles/c5_times.py 500 1000 250 > gpurun_out/r04/c5_times_ell.txt 2>&1
cat gpurun_out/r04/c5_times_ell.txt
for k in fused jac hess; do python profiles/stamps_ell32.py $k 500 > gpurun_out/r04/ell32_timeline_$k.txt 2>&1; cat gpurun_out/r04/ell32_timeline_$k.txt; done
