cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
for s in 41 42 43; do python tests/stress_gpu.py 700 $s 2>&1 | tail -1 | cut -c1-330; done > gpurun_out/r04/stress_kernels.txt
cat gpurun_out/r04/stress_kernels.txt
for s in 51 52; do python tests/stress_host_gpu.py 700 $s 2>&1 | tail -1 | cut -c1-200; done > gpurun_out/r04/stress_host2.txt
cat gpurun_out/r04/stress_host2.txt
