import sys,warnings; sys.path[:0]=["osqp-python_amd","."]; warnings.simplefilter("ignore")
import osqp_amd, problems
P,q,A,l,u=problems.banded_qp(100000)
m=osqp_amd.OSQP(); m.setup(P,q,A,l,u,verbose=False); m.update_settings(max_iter=60); m.solve()
for w in (10,11,12,13,3,4,0,2): print(w, round(m._solver.hip_time_kernel(w,300)*1e3,2), "us")
