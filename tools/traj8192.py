import sys, time
sys.path.insert(0,'algebraic-multigrid_amd')
import amg_ctypes as amg
n=8192
cp,ri,v=amg.laplacian(n); b=amg.rhs(n)
for sm,name,kw in ((amg.SM_MULTICOLOR_GS,"mc",dict(smoother_iters=1)),(amg.SM_JACOBI,"jac",dict(smoother_iters=2,omega=0.6))):
    for L in (9,12,18):
        t=time.time()
        mg=amg.Multigrid(cp,ri,v,b,L,smoother=sm,**kw)
        tr=[mg.rss()]
        for c in range(12):
            mg.vcycle(); tr.append(mg.rss())
        mg.sync(); t1=time.perf_counter(); mg.vcycle(10); mg.sync(); dt=(time.perf_counter()-t1)/10
        print(name,L,mg.coarse_solve_kind(),f"setup {time.time()-t:.1f}s cycle {dt*1e3:.2f} ms"," ".join(f"{x:.3e}" for x in tr),flush=True)
        mg.close()
