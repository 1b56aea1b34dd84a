import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from robot_mpcs_amd.scenarios import make_scenario
from robot_mpcs_amd._lib import Solver
from oracle.oracle import Oracle
np.set_printoptions(precision=4, suppress=False, linewidth=200)
for name, B in [('cfg1',1),('cfg2',256),('cfg3',256),('cfg4',128),('boxer',64)]:
    sc = make_scenario(name, B=B, seed=0)
    o = Oracle(sc.desc)
    cpu = o.solve_batch(sc.xinit, sc.x0, sc.params)
    s = Solver(sc.desc, max_batch=B)
    # sweep-level check
    dbg = s.debug_sweep(sc.xinit, sc.x0, sc.params)
    t=time.time(); gpu = s.solve(sc.xinit, sc.x0, sc.params); dt=time.time()-t
    nxs = sc.desc['nx']+sc.desc['ns']
    du = np.abs(gpu['z'][:,0,nxs:]-cpu['z'][:,0,nxs:]).max(axis=1)
    dz = np.abs(gpu['z']-cpu['z']).reshape(B,-1).max(axis=1)
    print(name, 'B',B,'gpu time %.3fs passes %d'%(dt, s.last_passes()), 'exit gpu', dict(zip(*np.unique(gpu['exitflag'],return_counts=True))), 'cpu', dict(zip(*np.unique(cpu['exitflag'],return_counts=True))))
    print('   iters gpu mean %.2f max %d | cpu mean %.2f max %d | iters equal frac %.3f'%(gpu['iters'].mean(), gpu['iters'].max(), cpu['iters'].mean(), cpu['iters'].max(), (gpu['iters']==cpu['iters']).mean()))
    print('   max du0 %.3e  max dz %.3e  obj diff %.3e kkt max %.2e'%(du.max(), dz.max(), np.abs(gpu['obj']-cpu['obj']).max(), gpu['kkt'].max()))
    s.close()
