import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))
import amg_ctypes as amg
n = int(sys.argv[1]); L = int(sys.argv[2])
cp, ri, v = amg.laplacian(n); b = amg.rhs(n)
mg = amg.Multigrid(cp, ri, v, b, L, smoother=amg.SM_MULTICOLOR_GS, host_only=True)
for l in range(L - 1):
    col, nc = mg.get_colors(l)
    m = n >> l
    lines = (len(col) + m - 1) // m
    pad = np.full(lines * m, -1, dtype=np.int64); pad[:len(col)] = col
    g = pad.reshape(lines, m)
    print("level", l, "n", len(col), "m", m, "colours", nc, "counts", np.bincount(col).tolist())
    for j in list(range(0, 6)) + [lines - 2, lines - 1]:
        print("  line", j, "first", g[j, :6].tolist(), "last", g[j, -6:].tolist())
