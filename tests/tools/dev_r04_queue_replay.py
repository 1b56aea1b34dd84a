#!/usr/bin/env python3
"""Development aid (CPU): replay of the fused kernel's instance queue with the oracle's pass counts -- makespan of a lone launch of 4096
instances on 2048 half-wavefronts in index order, random order and perfect longest-first order."""
import sys, os, numpy as np, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robot_mpcs_amd.scenarios import make_scenario
from oracle.oracle import Oracle
def makespan(passes, order, slots=2048):
    # slots = half-wavefronts; each instance occupies a slot for `passes` pass-times (lockstep halves: same pass time)
    h=[0.0]*slots; heapq.heapify(h)
    end=0
    for i in order:
        t=heapq.heappop(h); t2=t+passes[i]; end=max(end,t2); heapq.heappush(h,t2)
    return end
for cfg,B,seed in (("cfg2",4096,1000),("cfg2",4096,1017),("cfg2",4096,1068),("cfg3",4096,2000)):
    sc=make_scenario(cfg,B=B,seed=seed)
    r=Oracle(sc.desc).solve_batch(sc.xinit,sc.x0,sc.params)
    p=r['iters'].astype(float)+1
    rng=np.random.default_rng(0)
    print(cfg,seed,'mean %.1f max %d | ideal (work/slots) %.1f | index order %.0f random %.0f perfect LPT %.0f | top-half-first (perfect halves, random inside) %.0f'%(
        p.mean(),p.max(),p.sum()/2048,makespan(p,range(B)),makespan(p,rng.permutation(B)),makespan(p,np.argsort(-p)),
        makespan(p,np.concatenate([rng.permutation(np.argsort(-p)[:2048]),rng.permutation(np.argsort(-p)[2048:])]))))
