import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd")); sys.path.insert(0, ROOT)
import amg_ctypes as amg
from oracle import oracle
oracle.build()
n, L, iters = 512, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 2
A, b = oracle.laplacian(n), oracle.rhs(n)
amg.set_patch_min_rows(0)
mg = amg.Multigrid(A.colptr, A.rowind, A.val, b, L, smoother=amg.SM_MULTICOLOR_GS, smoother_iters=iters, exact_coarse_solve=True, keep_residual=True)
ref = oracle.Multigrid(A, b, L, smoother=oracle.SM_MULTICOLOR, smoother_iters=iters)
for l in range(L):
    col, nc = mg.get_colors(l); ref.set_colors(l, col, nc); print("level", l, "colours", nc, "n", len(col))
ref.vcycle(); mg.vcycle()
for l in range(L):
    for nm, a, r in (("f", mg.get_rhs(l), ref.get_vec(l, "f")), ("r", mg.get_residual(l) if l < L - 1 else None, ref.get_vec(l, "r")), ("u", mg.get_soln(l), ref.get_vec(l, "u"))):
        if a is None: continue
        bad = np.flatnonzero(a != r)
        m = n >> l
        print(l, nm, "mismatch", len(bad), "first", [(int(i) // m, int(i) % m) for i in bad[:6]], "maxrel", float(np.max(np.abs(a - r)) / (np.max(np.abs(r)) + 1e-300)))
