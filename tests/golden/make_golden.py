"""Generates the golden fixtures under tests/golden/ from the CPU oracle
(oracle/amg_oracle.cpp), which is itself pinned to the reference's published
known-answer values by tests/test_oracle_kat.py.  The reference cannot be built
in this image (Eigen 3.4 / Catch2 absent), so these vectors are oracle outputs,
cross-checked against SURVEY.md KAT-4 (independent SciPy restatement).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def config1():
    n, L = 128, 3
    A, b = O.laplacian(n), O.rhs(n)
    mg = O.Multigrid(A, b, L)
    rss = []
    for _ in range(12):
        mg.vcycle()
        rss.append(mg.rss())
    u = mg.get_vec(0, "u")
    return {"config": "2D 5-point Poisson 128x128, 3-level V-cycle, SparseGaussSeidel() nu1=nu2=2",
            "n": n, "n_levels": L, "rss": rss, "u_norm2": float(np.linalg.norm(u)),
            "u_8255": float(u[8255]),
            "level_sizes": [mg.n_dofs(l) for l in range(L)],
            "level_nnz": [mg.level_matrix(l).nnz for l in range(L)]}


def kat35():
    A, b = O.laplacian(35), O.rhs(35)
    mg = O.Multigrid(A, b, 8)
    it, conv, last, traj = mg.solve(1e-9, 5, 100)
    return {"config": "testlib.cpp:147-206: 35x35, 8 levels, tol 1e-9, every 5, max 100",
            "iters": it, "converged": conv, "rss_checks": traj.tolist(),
            "level_sizes": [mg.n_dofs(l) for l in range(8)]}


if __name__ == "__main__":
    json.dump(config1(), open(os.path.join(HERE, "kat_config1.json"), "w"), indent=1)
    json.dump(kat35(), open(os.path.join(HERE, "kat_35.json"), "w"), indent=1)
    print("wrote golden fixtures")
