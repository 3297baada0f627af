#!/usr/bin/env python3
"""DIAGNOSTIC (GPU box): phase shares, residency and dispatch gaps of the level-0 K-Patch down-leg
from the stamps of a -DAMG_PATCH_STAMPS build (kernels.hip: patch_stamp).  Never a timing source.
usage: AMG_HIP_LIBRARY=<stamps build> python tools/patch_stamps.py [n]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "algebraic-multigrid_amd"))
import amg_ctypes as amg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = amg.lib()
L = 16 if n == 4096 else 12
mg = amg.Multigrid.poisson(n, L, smoother=amg.SM_JACOBI, smoother_iters=2, omega=0.6)
mg.vcycle(3)
mg.sync()
tiles = ((n + 41) // 42) * (n // 64)
buf = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
fn = C.CDLL(os.environ.get("AMG_HIP_LIBRARY", amg.LIB_PATH)).amg_hip_debug_patch_stamps
fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(buf.data_ptr())) == 0
mg.profile_fine_sweep(6)          # back-to-back launches of the level-0 down-leg: the stamps of the last one stay
mg.sync()
fn(C.c_void_p(0))
s = buf.cpu().numpy().reshape(tiles, 8)
t = s[:, :6].astype(np.float64) * 0.01      # us (100 MHz)
t0 = t[:, 0].min()
t -= t0
dur = t[:, 5] - t[:, 0]
names = ["load (entry -> tile in LDS)", "sweep 1", "sweep 2", "copy-out + residual", "restriction + coarse sweep"]
print(f"{tiles} workgroups, kernel span {t[:, 5].max():.1f} us (diagnostic build), workgroup life avg {dur.mean():.2f} us "
      f"(min {dur.min():.2f}, max {dur.max():.2f})")
for k in range(5):
    d = t[:, k + 1] - t[:, k]
    print(f"  {names[k]:32s} avg {d.mean():6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  share {100 * d.mean() / dur.mean():5.1f} %")
hw = s[:, 7]
xcc = (hw >> 32) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
cus = np.unique(key)
print(f"distinct CUs seen: {cus.size}")
# residency: time-average number of workgroups alive per CU; gaps between an end and the next start on that CU
span = t[:, 5].max()
alive = dur.sum() / (cus.size * span)
print(f"time-average workgroups resident per CU: {alive:.2f}")
gaps, conc = [], []
for c in cus:
    idx = np.nonzero(key == c)[0]
    ev = sorted([(t[i, 0], 1) for i in idx] + [(t[i, 5], -1) for i in idx])
    cur, mx = 0, 0
    for _, d in ev:
        cur += d
        mx = max(mx, cur)
    conc.append(mx)
print(f"max concurrent workgroups on a CU: min {min(conc)}, median {int(np.median(conc))}, max {max(conc)}; workgroups per CU: "
      f"min {min(np.sum(key == c) for c in cus)}, max {max(np.sum(key == c) for c in cus)}")
# slots idle while work was still waiting: per CU, integrate (4 - resident) until the CU's last start
idle, busy = 0.0, 0.0
for c in cus:
    idx = np.nonzero(key == c)[0]
    last_start = t[idx, 0].max()
    ev = sorted([(t[i, 0], 1) for i in idx] + [(t[i, 5], -1) for i in idx])
    cur, prev = 0, 0.0
    for tm, d in ev:
        if tm > last_start:
            tm = last_start
        idle += (4 - cur) * max(0.0, tm - prev)
        busy += cur * max(0.0, tm - prev)
        prev = max(prev, tm)
        cur += d
print(f"before a CU's last workgroup starts: {100 * idle / (idle + busy):.1f} % of its four slots stand empty")
# gap between a workgroup's end and the next start on the same CU
g = []
for c in cus:
    idx = np.nonzero(key == c)[0]
    st = np.sort(t[idx, 0])
    en = np.sort(t[idx, 5])
    for e in en:
        nxt = st[st >= e]
        if nxt.size:
            g.append(nxt[0] - e)
g = np.array(g)
print(f"end -> next start on the same CU: median {np.median(g):.2f} us, p90 {np.percentile(g, 90):.2f} us")
starts = np.sort(t[:, 0])
print("start times (us) percentiles 0/25/50/75/100:", [round(float(np.percentile(starts, q)), 1) for q in (0, 25, 50, 75, 100)])
ends = np.sort(t[:, 5])
print("end times   (us) percentiles 0/25/50/75/100:", [round(float(np.percentile(ends, q)), 1) for q in (0, 25, 50, 75, 100)])
# per-XCD finish
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCD {x}: {int(m.sum())} workgroups, last end {t[m, 5].max():.1f} us, CUs {np.unique(key[m]).size}")
