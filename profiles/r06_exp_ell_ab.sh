# The row-gather forms of the exponential integrator's MFMA kernels against their dense-image forms (QC_EXP_ELL=0), one gpurun call
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_density.py -m gpu -x -q -k "exponential or kernel_names or density" 2>&1 | grep -E "passed|failed|rror|assert|Mismatch|Max" | head -20
for v in 1 0; do echo "QC_EXP_ELL=$v"; QC_EXP_ELL=$v python profiles/exp_bench.py 3 2>&1 | tail -1; QC_EXP_ELL=$v python profiles/exp_bench.py 5 2>&1 | tail -2; QC_EXP_ELL=$v python profiles/exp_bench.py 5 50 2>&1 | tail -1; done
