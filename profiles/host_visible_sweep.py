import os, sys, json, time
import numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3, T=1000)
Z = inp.traj.datavec
def run(env):
    for k in ("QC_HOST_COMPACT","QC_HOST_THREADS","QC_HOST_CHUNKS","QC_HOST_PIECES","QC_HOST_NT"): os.environ.pop(k, None)
    os.environ.update(env)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    F, J = np.empty(int(dyn.dims.F_len)), np.empty(int(dyn.dims.jac_nnz))
    for _ in range(5): dyn.F_dF(Z, out=(F, J))
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); dyn.F_dF(Z, out=(F, J)); ts.append(time.perf_counter() - t0)
    dyn.close()
    ts = np.array(ts) * 1e3
    print(json.dumps(env), "median %.3f min %.3f max %.3f ms" % (np.median(ts), ts.min(), ts.max()), flush=True)
# NOTE: QC_HOST_PIECES / QC_HOST_THREADS are read once per process (static), so every setting runs in a fresh process
if __name__ == "__main__":
    run(json.loads(sys.argv[1]) if len(sys.argv) > 1 else {})
