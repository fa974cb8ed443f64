import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3, T=1000)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
for k in range(4):
    try:
        dyn.F_dF(inp.traj.datavec)
        print('NO ERROR')
    except qc._lib.QCollocError as e:
        print('ERROR:', e)
dyn.close()
print('closed')
