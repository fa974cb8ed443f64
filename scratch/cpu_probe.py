import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import __graft_entry__ as g
qc = g.load_package()
from oracle_bridge import problem_from_inputs
import oracle.qc_oracle_c as oc
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
inp = qc.config_inputs(3, T=1000)
prob = problem_from_inputs(inp)
Z = inp.traj.datavec
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    co = oc.COracle(prob, threads=th)
    co.F_dF(Z)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.5:
        co.F_dF(Z); n += 1
    print(th, 'threads:', (time.perf_counter() - t0) / n * 1e3, 'ms/eval')
