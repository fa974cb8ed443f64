cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_reference_golden.py tests/test_gpu_parity.py -q -m gpu -k "mock_records or compact_host or sentinel or stress" > gpurun_out/r04/run7_tests.txt 2>&1
tail -5 gpurun_out/r04/run7_tests.txt
bash profiles/collect.sh r04a > gpurun_out/r04/collect_r04a.log 2>&1
tail -3 gpurun_out/r04/collect_r04a.log
bash profiles/collect_mfma_r04.sh > gpurun_out/r04/collect_mfma_r04.log 2>&1
tail -60 gpurun_out/r04/collect_mfma_r04.log | head -80
