cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
python -m pytest tests/test_closures.py tests/test_evaluator.py tests/test_end_to_end.py -x -q -m gpu > gpurun_out/r04/run9_tests.txt 2>&1; tail -4 gpurun_out/r04/run9_tests.txt
for seed in 3 11 12 13; do python tests/stress_host_gpu.py 300 $seed > gpurun_out/r04/stress_host_$seed.txt 2>&1; tail -1 gpurun_out/r04/stress_host_$seed.txt | cut -c1-200; done
