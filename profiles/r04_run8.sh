cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
./tests/hip/fill_floor > gpurun_out/r04/fill_floor.txt 2>&1; cat gpurun_out/r04/fill_floor.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke.txt 2>&1; tail -3 gpurun_out/r04/smoke.txt
python tests/stress_host_gpu.py 150 > gpurun_out/r04/stress_host.txt 2>&1; tail -3 gpurun_out/r04/stress_host.txt
python tests/stress_gpu.py 150 > gpurun_out/r04/stress_gpu.txt 2>&1; tail -3 gpurun_out/r04/stress_gpu.txt
