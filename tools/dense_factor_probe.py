"""How fast are the library pieces a large-rank Woodbury correction needs on one MI355X (fp64)?  syrk-like GEMM, Cholesky, inverse."""
import time, torch
dev = 'cuda'
for r, k in ((4096, 8192), (10000, 15000)):
    W = torch.randn(r, k, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for name, f in (('gemm W W^T', lambda: W @ W.T),):
        f(); torch.cuda.synchronize(); t = time.time(); S = f(); torch.cuda.synchronize(); dt = time.time() - t
        print('r=%d k=%d %s: %.1f ms, %.1f TFLOP/s' % (r, k, name, dt * 1e3, 2.0 * r * r * k / dt / 1e12), flush=True)
    S = S + r * torch.eye(r, dtype=torch.float64, device=dev)
    torch.linalg.cholesky(S[:512, :512]); torch.cuda.synchronize()
    t = time.time(); L = torch.linalg.cholesky(S); torch.cuda.synchronize(); dt = time.time() - t
    print('   cholesky: %.1f ms (%.1f TFLOP/s)' % (dt * 1e3, r ** 3 / 3 / dt / 1e12), flush=True)
    t = time.time(); Si = torch.cholesky_inverse(L); torch.cuda.synchronize(); dt = time.time() - t
    print('   cholesky_inverse: %.1f ms' % (dt * 1e3), flush=True)
    g = torch.randn(r, dtype=torch.float64, device=dev)
    Si @ g; torch.cuda.synchronize(); t = time.time()
    for _ in range(20): h = Si @ g
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print('   gemv: %.3f ms (%.0f GB/s)' % (dt * 1e3, 8.0 * r * r / dt / 1e9), flush=True)
    t = time.time(); X = torch.cholesky_solve(g[:, None], L); torch.cuda.synchronize(); dt = time.time() - t
    print('   cholesky_solve (2 trsv): %.2f ms' % (dt * 1e3), flush=True)
    print('   |S Si - I| max', float((S @ Si - torch.eye(r, dtype=torch.float64, device=dev)).abs().max()), flush=True)
