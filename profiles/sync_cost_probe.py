"""What the closing `torch.cuda.synchronize()` of bench.py's timed region costs when the device is ALREADY idle (the launching thread has
seen the last step's event), under the runtime's wait settings.  One child process per setting (the variables are read at HIP start-up;
the parent never touches the GPU).    gpurun -- 'python profiles/sync_cost_probe.py'"""
import json
import os
import subprocess
import sys
import time

CHILD = r'''
import os, sys, time, json
import torch
sys.path.insert(0, os.environ["QC_ROOT"])
import __graft_entry__ as g
qc = g.load_package()
dev = torch.device("cuda", 0)
inp = qc.config_inputs(3, T=1000)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).to(dev)
F = torch.empty(int(dyn.dims.F_len), dtype=torch.float64, device=dev)
J = torch.empty(int(dyn.dims.jac_nnz), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)
launch = dyn.bind_F_dF_device(Z, F, J, st)
for _ in range(2000): launch()
torch.cuda.synchronize()
ev = torch.cuda.Event(enable_timing=True); ev.record(st); torch.cuda.synchronize()
idle, after, blocking, total = [], [], [], []
for rep in range(40):
    t0 = time.perf_counter(); torch.cuda.synchronize(); idle.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(20): launch()
    ev.record(st)
    while not ev.query(): pass
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    after.append(t2 - t1); total.append(t2 - t0)
    t0 = time.perf_counter()
    for _ in range(20): launch()
    torch.cuda.synchronize(); blocking.append(time.perf_counter() - t0)
med = lambda v: sorted(v)[len(v) // 2] * 1e6
print(json.dumps({"sync_idle_us": med(idle), "sync_after_seen_done_us": med(after), "20_steps_poll_then_sync_us": med(total), "20_steps_blocking_sync_us": med(blocking)}))
'''

def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for name, env in (("default", {}), ("ROC_ACTIVE_WAIT_TIMEOUT=1000", {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
                      ("ROC_CPU_WAIT_FOR_SIGNAL=0", {"ROC_CPU_WAIT_FOR_SIGNAL": "0"}), ("ROC_CPU_WAIT_FOR_SIGNAL=1", {"ROC_CPU_WAIT_FOR_SIGNAL": "1"}),
                      ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}), ("AMD_DIRECT_DISPATCH=0", {"AMD_DIRECT_DISPATCH": "0"}),
                      ("DEBUG_HIP_BLOCK_SYNC=0", {"DEBUG_HIP_BLOCK_SYNC": "0"}), ("ROC_SYSTEM_SCOPE_SIGNAL=0", {"ROC_SYSTEM_SCOPE_SIGNAL": "0"}),
                      ("HSA_ENABLE_INTERRUPT=0+ROC_ACTIVE_WAIT_TIMEOUT=1000", {"HSA_ENABLE_INTERRUPT": "0", "ROC_ACTIVE_WAIT_TIMEOUT": "1000"})):
        e = dict(os.environ, QC_ROOT=root, **env)
        try:
            r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=90)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[name] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
        except subprocess.TimeoutExpired:
            out[name] = {"error": "no result within 90 s (the setting hangs the process)"}
        print(name, out[name], flush=True)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "gpurun_out", "sync_cost_probe.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
