import sys,os,warnings
warnings.simplefilter("ignore")
sys.path[:0]=[".","osqp-python_amd"]
os.environ['OSQP_HIP_SETUP_TIMING']='1'
import numpy as np, osqp_amd, problems
n=int(sys.argv[1]); w=int(sys.argv[2]); fr=float(sys.argv[3])
P,q,A,l,u=problems.banded_qp(n,window=w,long_range=fr)
m=osqp_amd.OSQP(); m.setup(P,q,A,l,u,eps_abs=1e-6,eps_rel=1e-6,verbose=False,max_iter=20000,adaptive_rho_interval=50,check_termination=25)
r=m.solve(); s=m._solver.hip_stats()
print(r.info.status, r.info.iter, s['pcg_fused'], s['f1_replicas'], s['f1_far_columns'], s['kernel_launches'], s['pcg_iters_total']/r.info.iter, "gpu ms", s['gpu_solve_ms'])
