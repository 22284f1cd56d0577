import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import altro_amd_loader
import altro_mpc_icra2021_amd as altro
B,S=8192,20
pb = altro.problems.gen_random_linear_batch(B, steps=S + 5)
mp = altro.mpc.BatchMPC(pb)
mp.initial_solve()
for i in range(5): mp.step(i)
outer=[]; its=[]
for i in range(5, 25):
    mp.step(i); st=altro.stats(mp.solver); outer.append(st.iterations_outer.copy()); its.append(st.iterations.copy())
outer=np.array(outer); its=np.array(its)
print("outer iterations per solve: hist", np.bincount(outer.ravel()))
print("iterations per solve hist", np.bincount(its.ravel())[:12])
two = (outer==2)
print("solves with 2 outer iterations: iterations hist", np.bincount(its[two])[:10])
print("per instance: fraction of its 20 solves with outer>=2: hist", np.bincount((outer>=2).sum(0)))
