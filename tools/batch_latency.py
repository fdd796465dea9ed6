import sys,time,warnings,os
warnings.simplefilter("ignore")
sys.path[:0]=[".","osqp-python_amd"]
import numpy as np, torch, osqp_amd, problems
P,q,A,L,U=problems.mpc_batch(4096)
for variant in ("auto","direct256"):
    if variant!="auto": os.environ["OSQP_HIP_BATCH_VARIANT"]=variant
    s=osqp_amd.OSQP(algebra='hip'); s.setup(P,q,A,L[0],U[0],eps_abs=1e-6,eps_rel=1e-6,verbose=False,max_iter=4000)
    for B in (256,512,4096):
        x,y,rec=s._solver.hip_batch_solve(l=L[:B],u=U[:B])
        x,y,rec=s._solver.hip_batch_solve(l=L[:B],u=U[:B])
        k=s._solver.hip_stats()['gpu_solve_ms']
        it=rec[:,1]
        print(variant,B,"kernel ms %.3f"%k,"iters max %d mean %.1f sum %d"%(it.max(),it.mean(),it.sum()),"us/iter(max chain) %.2f"%(1e3*k/it.max()),"rho upd mean %.2f"%rec[:,6].mean(), "solved",int((rec[:,0]==1).sum()))
