import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from cg_mrslam_amd import synth, Context
from oracle import oracle as O
ctx=Context(0)
for (V,E,it) in [(2,1,2),(50,120,5),(2000,8000,8),(10000,40000,10)]:
    g=synth.make_pose_graph(V,E,seed=5)
    a=(g['poses'],g['fixed'],g['edge_from'],g['edge_to'],g['meas'],g['info'])
    t=time.time(); rc,p,chi=ctx.gn_optimize(*a,it); tg=time.time()-t
    t=time.time(); st,p2,chi2,tm=O.gn_optimize(*a,it); tc=time.time()-t
    print(V,E,'rc',rc,'gpu %.4fs cpu %.4fs'%(tg,tc), 'chi rel', np.max(np.abs(chi-chi2)/np.maximum(chi2,1e-30)), 'pose', np.abs(p-p2).max())
    print('  ', chi[-3:], chi2[-3:]); print('  ', ctx.gn_last_timing())
g=synth.make_pose_graph(10000,40000,seed=12345)
a=(g['poses'],g['fixed'],g['edge_from'],g['edge_to'],g['meas'],g['info'])
for r in range(3):
    t=time.time(); rc,p,chi=ctx.gn_optimize(*a,10); print('run',r,time.time()-t, ctx.gn_last_timing())
ctx.set_profiling(True); ctx.gn_optimize(*a,10); print(ctx.gn_kernel_times())
