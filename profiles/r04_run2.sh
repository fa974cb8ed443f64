cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse_drive or dense_drives_keep or mfma32 or config_hessian_parity or full_size or config5 or kernel_names" > gpurun_out/r04/run2_tests.txt 2>&1
tail -15 gpurun_out/r04/run2_tests.txt
python profiles/c5_times.py 500 1000 > gpurun_out/r04/c5_times_ell.txt 2>&1
QC_NO_ELL=1 python profiles/c5_times.py 500 1000 > gpurun_out/r04/c5_times_dense.txt 2>&1
cat gpurun_out/r04/c5_times_ell.txt gpurun_out/r04/c5_times_dense.txt
python profiles/stamps_hess32_ell.py 500 > gpurun_out/r04/hess32_ell_timeline.txt 2>&1
cat gpurun_out/r04/hess32_ell_timeline.txt
