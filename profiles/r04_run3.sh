cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse_drive or dense_drives_keep or mfma32 or config_hessian_parity or full_size or config5 or kernel_names or fused_launch or config_parity or compact_host or new_x" > gpurun_out/r04/run3_tests.txt 2>&1
tail -15 gpurun_out/r04/run3_tests.txt
python profiles/c5_times.py 500 1000 > gpurun_out/r04/c5_times_ell.txt 2>&1
QC_ELL_JAC=0 python profiles/c5_times.py 500 > gpurun_out/r04/c5_times_elljac0.txt 2>&1
cat gpurun_out/r04/c5_times_ell.txt gpurun_out/r04/c5_times_elljac0.txt
