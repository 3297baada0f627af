"""Row-block sharded V-cycle: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Scope: the 5-point Poisson hierarchy is banded in the flat dof
index, so every level is cut into contiguous row blocks, one per rank:

  level 0   rank g owns rows [g*n0/G, (g+1)*n0/G)
  level l+1 rank g owns the coarse dofs j whose C-point 2j+1 it owns on level l,
            i.e. [floor(s/2), floor(e/2))

Exchanges (the only collectives on the data path):
  * halo: before every Jacobi sweep and before the residual each rank swaps the
    boundary unknowns its neighbours' rows reference (width = the level's half
    bandwidth m_l+1) with rank-1 / rank+1 -- grouped isend/irecv, KiB-sized;
  * restriction / prolongation need one remote entry per side (width-1 halo);
  * when a level gets small (fewer than `dist_min_rows` rows in total: the halo
    exchanges then cost more than the sweeps they split) its right-hand
    side is all-gathered and every rank runs the rest of the V-cycle redundantly
    as an ordinary single-GPU solver on the sub-hierarchy (the C ABI solver with
    its hipGraph), then keeps its slice of the correction;
  * rss: all_reduce(SUM) of one fp64.

The smoother is the true (two-buffer) Jacobi -- the lexicographic reference
smoother is a cross-rank sequential recurrence and does not shard.
Per-row arithmetic is the single-GPU kernels' (same CSR kernels, same column
order), so the G-rank result equals the 1-rank result bit for bit; the CPU
tests (tests/test_dist_gloo.py) check exactly that with a numpy backend.

All compute goes through a `backend` object: HipBackend (torch device tensors +
the C ABI's device-pointer launchers) in production; tests pass their own.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist


# ----------------------------------------------------------------- partition ---
def row_bounds(n, world):
    """Level-0 row blocks: rank g owns [bounds[g], bounds[g+1])."""
    return [n * g // world for g in range(world + 1)]


def coarse_bounds(bounds, n_H):
    """Coarse dof j lives with the owner of fine dof 2j+1 (its C-point)."""
    out = [min(b // 2, n_H) for b in bounds]
    out[0] = 0
    out[-1] = n_H
    return out


class LocalMatrix:
    """Rows [s, e) of a global CSR matrix with columns renumbered into the
    rank's halo-extended vector [c_lo .. c_hi]."""

    def __init__(self, rowptr, col, val, s, e, c_lo=None, c_hi=None):
        p0, p1 = int(rowptr[s]), int(rowptr[e])
        self.n_rows = e - s
        self.rowptr = (rowptr[s:e + 1] - p0).astype(np.int32)
        gcol = col[p0:p1]
        self.val = np.ascontiguousarray(val[p0:p1], dtype=np.float64)
        if c_lo is None:
            c_lo = int(gcol.min()) if gcol.size else s
            c_hi = int(gcol.max()) if gcol.size else e - 1
            c_lo, c_hi = min(c_lo, s), max(c_hi, e - 1)
        self.c_lo, self.c_hi = c_lo, c_hi
        self.col = (gcol - c_lo).astype(np.int32)
        self.nnz = int(self.col.size)
        self.halo_lo = s - c_lo          # entries needed from lower ranks
        self.halo_hi = c_hi - (e - 1)    # entries needed from higher ranks
        self.diag_shift = s - c_lo       # local row i has its diagonal at column i + shift


def linear_R_rows(cs, ce, n_h, s_fine):
    """CSR rows [cs, ce) of R = P^T (interpolator.hpp:106-134): row j =
    {0.5, 1.0, 0.5} at fine columns {2j, 2j+1, 2j+2} (< n_h).  Columns are local
    to the fine residual vector extended by ONE entry each side: index = c - (s_fine-1)."""
    j = np.arange(cs, ce, dtype=np.int64)
    cols = np.stack([2 * j, 2 * j + 1, 2 * j + 2], 1)
    vals = np.tile(np.array([0.5, 1.0, 0.5]), (j.size, 1))
    keep = cols < n_h
    cnt = keep.sum(1)
    rowptr = np.zeros(j.size + 1, np.int32)
    np.cumsum(cnt, out=rowptr[1:])
    return rowptr, (cols[keep] - (s_fine - 1)).astype(np.int32), vals[keep].astype(np.float64)


def linear_P_rows(s, e, n_H, cs):
    """CSR rows [s, e) of P: odd i=2j+1 -> {1.0 @ j}; even i=2j -> {0.5 @ j-1,
    0.5 @ j} (ascending column, entries outside [0, n_H) dropped).  Columns are
    local to the coarse vector extended by ONE entry each side: index = j - (cs-1)."""
    i = np.arange(s, e, dtype=np.int64)
    j = i // 2
    odd = (i & 1) == 1
    c0 = np.where(odd, j, j - 1)
    c1 = j
    v0 = np.where(odd, 1.0, 0.5)
    k0 = (c0 >= 0) & (c0 < n_H)
    k1 = (~odd) & (c1 < n_H)
    cols = np.stack([c0, c1], 1)
    vals = np.stack([v0, np.full(i.size, 0.5)], 1)
    keep = np.stack([k0, k1], 1)
    cnt = keep.sum(1)
    rowptr = np.zeros(i.size + 1, np.int32)
    np.cumsum(cnt, out=rowptr[1:])
    return rowptr, (cols[keep] - (cs - 1)).astype(np.int32), vals[keep].astype(np.float64)


# ------------------------------------------------------------------- backend ---
class HipBackend:
    """torch device tensors + libamg_hip.so device-pointer launchers."""

    def __init__(self, device):
        import amg_ctypes as amg
        self.amg = amg
        self.lib = amg.lib()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # a dedicated (non-null) stream carries everything this rank does: kernels,
        # the agglomerated solver's graph, and the waits of the collectives
        self._stream = torch.cuda.Stream(self.device)
        torch.cuda.set_stream(self._stream)

    def stream(self):
        return self._stream.cuda_stream

    def vec(self, n):
        return torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)[:n]

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def to_numpy(self, t):
        return t.detach().cpu().numpy()

    def matrix(self, rowptr, col, val, ncols=None, diag_shift=0):
        """Upload a local CSR block in the library's device layout (dictionary-coded when
        the block qualifies, else SELL-64 panels)."""
        import ctypes as C
        rowptr = np.ascontiguousarray(rowptr, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        if ncols is None:
            ncols = int(col.max()) + 1 if col.size else 1
        h = C.c_void_p()
        i32, f64 = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        self._chk(self.lib.amg_hip_devmat_create(
            rowptr.size - 1, ncols, rowptr.ctypes.data_as(i32), col.ctypes.data_as(i32),
            val.ctypes.data_as(f64), self.amg.LAYOUT_AUTO, diag_shift, self.device.index,
            C.byref(h)))
        self._mats = getattr(self, "_mats", [])
        self._mats.append(h)
        return h

    def matrix_layout(self, m):
        """(layout name, matrix stream bytes) of an uploaded block."""
        import ctypes as C
        lay, nb = C.c_int32(0), C.c_int64(0)
        self._chk(self.lib.amg_hip_devmat_layout(m, C.byref(lay), C.byref(nb)))
        return {1: "csr", 2: "sell", 3: "dict"}.get(lay.value, "?"), int(nb.value)

    def _chk(self, st):
        if st != 0:
            raise self.amg.AmgHipError(st, self.lib.amg_hip_last_error().decode())

    def residual(self, m, u_ext, f, r):
        self._chk(self.lib.amg_hip_devmat_apply(m, 0, u_ext.data_ptr(), f.data_ptr(), r.data_ptr(),
                                                1.0, 0, self.stream()))

    def jacobi_from_zero(self, diag, b, u_out, omega):
        self._chk(self.lib.amg_hip_dev_jacobi_from_zero(u_out.numel(), diag.data_ptr(),
                  b.data_ptr(), u_out.data_ptr(), omega, self.stream()))

    def jacobi(self, m, u_ext, b, u_out, omega, diag_shift):
        self._chk(self.lib.amg_hip_devmat_apply(m, 1, u_ext.data_ptr(), b.data_ptr(),
                                                u_out.data_ptr(), omega, diag_shift, self.stream()))

    def spmv(self, m, v_ext, out):
        self._chk(self.lib.amg_hip_devmat_apply(m, 2, v_ext.data_ptr(), None, out.data_ptr(),
                                                1.0, 0, self.stream()))

    def add_(self, y, x):
        self._chk(self.lib.amg_hip_dev_axpy1(y.numel(), x.data_ptr(), y.data_ptr(), self.stream()))

    def sumsq(self, r):
        if not hasattr(self, "_scratch"):
            self._scratch = self.vec(1100)
        out = self._scratch[1024:1025]
        self._chk(self.lib.amg_hip_dev_sumsq(r.numel(), r.data_ptr(), out.data_ptr(),
                                             self._scratch.data_ptr(), self.stream()))
        return out.clone()

    def sync(self):
        self._stream.synchronize()

    def tail(self, colptr, rowind, val, n_levels, omega, sweeps, use_graph=True):
        """The agglomerated coarse part: an ordinary single-GPU solver."""
        return _HipTail(self, colptr, rowind, val, n_levels, omega, sweeps, use_graph)


class _HipTail:
    def __init__(self, be, colptr, rowind, val, n_levels, omega, sweeps, use_graph=True):
        amg = be.amg
        n = colptr.size - 1
        self.be = be
        # same stream as the rest of the rank's work: no host synchronisation needed
        self.mg = amg.Multigrid(colptr, rowind, val, np.zeros(n), n_levels,
                                smoother=amg.SM_JACOBI, smoother_iters=sweeps, omega=omega,
                                device=be.device.index, stream=be.stream(), use_graph=use_graph)
        self.n = n

    def cycle(self, f_full, u_full, zero_guess=True):
        """u_full = one V-cycle on the sub-hierarchy, rhs f_full; from a zero guess
        (multigrid.hpp:278) unless this is the whole hierarchy."""
        self.mg.copy_vec_dev(0, "f", f_full.data_ptr(), True)
        if zero_guess:
            self.mg.zero_vec(0, "u")
        self.mg.vcycle(1)
        self.mg.copy_vec_dev(0, "u", u_full.data_ptr(), False)

    def rss(self):
        return self.mg.rss()


class IpcUnavailable(RuntimeError):
    """Raised on EVERY rank when the hipIpc halo path cannot be set up on some rank."""


class _DevArray:
    """__cuda_array_interface__ view of raw device memory (arena slices)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


def _align(x, a=256):
    return (x + a - 1) // a * a


# ------------------------------------------------------------------- driver ----
class DistLevel:
    pass


class DistributedVcycle:
    """multigrid.hpp:263-305 over row-block shards (true Jacobi smoother)."""

    def __init__(self, hierarchy, b, backend, rank, world, omega=0.6, sweeps=2,
                 dist_min_rows=6000000, group=None, host_staged=False, comm="p2p"):
        """hierarchy: object with n_levels, get_n_dofs(l), get_coefficient_matrix(l)
        -> CSC (colptr, rowind, val) of the (symmetric) level matrix.
        host_staged: exchange through host buffers (a process group whose backend
        cannot move device tensors, e.g. gloo with a GPU compute backend)."""
        self.be, self.rank, self.world, self.group = backend, rank, world, group
        self.host_staged = bool(host_staged)
        # comm: "p2p"   = torch.distributed isend/irecv + all_gather (RCCL);
        #       "ipc"   = direct pushes into the neighbours' hipIpc-mapped halo slots,
        #                 stream-ordered epoch flags (host-driven stream memory ops);
        #       "graph" = the same pushes and the all-gather as kernels with in-kernel
        #                 flags, the whole sharded V-cycle captured in ONE hipGraph
        self.comm = comm
        self.graph_exec = None
        self._cycles_run = 0
        self.arena = None
        self._peer_bases = {}
        self.omega, self.sweeps = float(omega), int(sweeps)
        L = hierarchy.n_levels
        self.n_levels = L
        sizes = [hierarchy.get_n_dofs(l) for l in range(L)]
        # ---- which levels stay distributed ----
        bounds = [row_bounds(sizes[0], world)]
        for l in range(1, L):
            bounds.append(coarse_bounds(bounds[-1], sizes[l]))
        self.n_dist = 0
        mats = []
        for l in range(L):
            if world == 1:
                break
            cp, ri, v = hierarchy.get_coefficient_matrix(l)
            owned = [bounds[l][g + 1] - bounds[l][g] for g in range(world)]
            # half bandwidth of the level (every rank derives the same number)
            rows_of = np.repeat(np.arange(sizes[l], dtype=np.int64), np.diff(cp))
            hb = int(np.abs(ri.astype(np.int64) - rows_of).max()) if ri.size else 0
            del rows_of
            small = sizes[l] < max(dist_min_rows, 1) or min(owned) < hb + 2
            last_possible = (l == L - 1)   # the coarsest level is always solved redundantly
            if small or last_possible:
                break
            mats.append((cp, ri, v))
            self.n_dist = l + 1
        self.bounds = bounds
        self.sizes = sizes
        # ---- distributed levels ----
        self.lv = []
        for l in range(self.n_dist):
            cp, ri, v = mats[l]
            s, e = bounds[l][rank], bounds[l][rank + 1]
            D = DistLevel()
            D.n, D.s, D.e = sizes[l], s, e
            D.A = LocalMatrix(cp, ri, v, s, e)     # symmetric: CSC arrays == CSR arrays
            D.mat = backend.matrix(D.A.rowptr, D.A.col, D.A.val, diag_shift=D.A.diag_shift)
            dg = np.zeros(e - s)                    # a_ii of the owned rows
            rows_l = np.repeat(np.arange(e - s), np.diff(D.A.rowptr))
            on_d = D.A.col == rows_l + D.A.diag_shift
            dg[rows_l[on_d]] = D.A.val[on_d]
            D.diag = backend.from_numpy(dg)
            n_ext = D.A.halo_lo + (e - s) + D.A.halo_hi
            D.n_ext = n_ext
            D.f = backend.vec(e - s)
            D.tmp = backend.vec(e - s)
            cs, ce = bounds[l + 1][rank], bounds[l + 1][rank + 1]
            D.cs, D.ce = cs, ce
            rp, c, vv = linear_R_rows(cs, ce, sizes[l], s)
            D.R = backend.matrix(rp, c, vv)
            rp, c, vv = linear_P_rows(s, e, sizes[l + 1], cs)
            D.P = backend.matrix(rp, c, vv)
            self.lv.append(D)
        # exchangeable vectors: u, u2 (halo-extended), r and uH (width-1 halos)
        if self.comm in ("ipc", "graph") and self.n_dist:
            self._setup_ipc()
        else:
            for D in self.lv:
                D.u = backend.vec(D.n_ext)
                D.u2 = backend.vec(D.n_ext)
                D.r = backend.vec(D.e - D.s + 2)
                D.uH = backend.vec(D.ce - D.cs + 2)
        # halo widths of every rank, so senders know how much a neighbour wants
        if self.n_dist:
            mine = torch.tensor([[D.A.halo_lo, D.A.halo_hi] for D in self.lv], dtype=torch.int64,
                                device="cpu" if self.host_staged else backend.device)
            allw = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allw, mine, group=group)
            allw = [t.cpu() for t in allw]
            self.halo = [[(int(allw[g][l, 0]), int(allw[g][l, 1])) for g in range(world)]
                         for l in range(self.n_dist)]
        # ---- the redundant (agglomerated) part: levels n_dist .. L-1 ----
        la = self.n_dist
        cp, ri, v = hierarchy.get_coefficient_matrix(la) if la < L else (None, None, None)
        if self.comm == "graph" and self.n_dist:
            self.tail = backend.tail(cp, ri, v, L - la, self.omega, self.sweeps, use_graph=False)
        else:
            self.tail = backend.tail(cp, ri, v, L - la, self.omega, self.sweeps)
        self.tail_n = sizes[la]
        if not (self.comm == "graph" and self.n_dist):
            self.tail_f = backend.vec(self.tail_n)   # graph mode: lives in the arena
        self.tail_u = backend.vec(self.tail_n)
        if la > 0:
            counts = [bounds[la][g + 1] - bounds[la][g] for g in range(world)]
            self.tail_counts = counts
            self.tail_max = max(counts)
            self.gather_in = backend.vec(self.tail_max)
            self.gather_out = backend.vec(self.tail_max * world)
        # rhs
        if self.n_dist:
            D0 = self.lv[0]
            D0.f.copy_(backend.from_numpy(np.asarray(b[D0.s:D0.e], dtype=np.float64)))
        else:  # nothing is big enough to shard: every rank runs the whole cycle
            self.tail_f.copy_(backend.from_numpy(np.asarray(b, dtype=np.float64)))

    # ---- hipIpc mode: arena, peers, descriptors ----
    def _setup_ipc(self):
        amg, lib = self.be.amg, self.be.lib
        import ctypes as C
        dev = self.be.device
        la = self.n_dist
        tail_n = self.sizes[la]
        # layout of my arena: per level [u | u2 | r | uH | flags(256 B)], then
        # [tail_f | gather DATA(64 B) | gather FREE(64 B) | timeout]
        table, off = [], 0
        for D in self.lv:
            n, nH = D.e - D.s, D.ce - D.cs
            row = []
            for cnt in (D.n_ext, D.n_ext, n + 2, nH + 2):
                row.append(off)
                off = _align(off + 8 * max(cnt, 1))
            row.append(off)                      # flags: [0,64) epoch words, [64,128) 0/1 words
            off = _align(off + 256)
            row += [D.A.halo_lo, D.A.halo_hi, n, nH]
            table.append(row)
        misc = [off]                             # tail_f
        off = _align(off + 8 * max(tail_n, 1))
        misc.append(off)                         # gather DATA words (one per source rank)
        off = _align(off + 64)
        misc.append(off)                         # gather FREE words (one per destination rank)
        off = _align(off + 64)
        misc.append(off)                         # timeout word
        off = _align(off + 64)
        cdev = "cpu" if self.host_staged else dev

        def all_ok(ok, what):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            if int(t.item()) == 0:
                self._ipc_cleanup()
                raise IpcUnavailable(what)

        h = C.c_void_p()
        hb = C.create_string_buffer(64)
        ok = (lib.amg_hip_arena_create(max(off, 256), dev.index, C.byref(h)) == 0)
        if ok:
            self.arena = h
            ok = (lib.amg_hip_arena_export(h, hb) == 0)
        base = lib.amg_hip_arena_base(h) if ok else 0
        self._base = base
        if ok:  # FREE words start at 1 (slots are free), everything else at 0
            for row in table:
                for chan in range(4):
                    ok = ok and lib.amg_hip_fill_u32(base + row[4] + 64 + 16 * chan + 8, 2, 1, None) == 0
            ok = ok and lib.amg_hip_fill_u32(base + misc[2], 16, 1, None) == 0
            torch.cuda.synchronize(dev)
        all_ok(ok, "hipIpc arena could not be created / exported on some rank")
        self._misc = misc
        for D, row in zip(self.lv, table):
            n, nH = D.e - D.s, D.ce - D.cs
            mk = lambda o, cnt: torch.as_tensor(_DevArray(base + o, cnt), device=dev)
            D.u, D.u2 = mk(row[0], D.n_ext), mk(row[1], D.n_ext)
            D.r, D.uH = mk(row[2], n + 2), mk(row[3], nH + 2)
            D.chan = {D.u.data_ptr(): 0, D.u2.data_ptr(): 1, D.r.data_ptr(): 2, D.uH.data_ptr(): 3}
            D.epoch = [0, 0, 0, 0]
            D.desc = [None] * 4
            D.kdesc = [None] * 4
        if self.comm == "graph":
            self.tail_f = torch.as_tensor(_DevArray(base + misc[0], tail_n), device=dev)
        # exchange handles and layout tables
        mine = torch.tensor(list(hb.raw), dtype=torch.uint8, device=cdev)
        handles = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(handles, mine, group=self.group)
        tab = torch.tensor([r + [0] * 0 for r in table] , dtype=torch.int64, device=cdev)
        tabs = [torch.zeros_like(tab) for _ in range(self.world)]
        dist.all_gather(tabs, tab, group=self.group)
        self._tabs = [t.cpu().tolist() for t in tabs]
        mt = torch.tensor(misc, dtype=torch.int64, device=cdev)
        mts = [torch.zeros_like(mt) for _ in range(self.world)]
        dist.all_gather(mts, mt, group=self.group)
        self._miscs = [t.cpu().tolist() for t in mts]
        ok = True
        peers = range(self.world) if self.comm == "graph" else (self.rank - 1, self.rank + 1)
        for g in peers:
            if 0 <= g < self.world and g != self.rank:
                pb = C.c_void_p()
                raw = bytes(handles[g].cpu().tolist())
                if lib.amg_hip_arena_open_peer(raw, C.byref(pb)) == 0:
                    self._peer_bases[g] = pb.value
                else:
                    ok = False
        if self.comm == "graph" and self.world > 16:
            ok = False
        all_ok(ok, "a neighbour's hipIpc arena could not be mapped on some rank")

    def _ipc_cleanup(self):
        for pb in self._peer_bases.values():
            self.be.lib.amg_hip_arena_close_peer(pb)
        self._peer_bases = {}
        for D in self.lv:
            D.u = D.u2 = D.r = D.uH = None
        if self.arena is not None:
            self.be.lib.amg_hip_arena_destroy(self.arena)
            self.arena = None

    def _ipc_desc(self, l, chan, n_owned, lo, hi, send_prev, send_next):
        """Static part of the descriptor of channel `chan` on level l."""
        amg = self.be.amg
        D = self.lv[l]
        r, w = self.rank, self.world
        me = self._tabs[r][l]
        d = amg.HaloDesc()
        fl = self._base + me[4] + 16 * chan          # my 4 flag words of this channel
        d.my_data_from_prev, d.my_data_from_next = fl, fl + 4
        d.my_ack_from_prev, d.my_ack_from_next = fl + 8, fl + 12
        mybuf = self._base + me[chan]
        d.recv_from_prev = int(r > 0 and lo > 0)
        d.recv_from_next = int(r < w - 1 and hi > 0)
        if r > 0:
            pt = self._tabs[r - 1][l]
            pb = self._peer_bases[r - 1]
            pfl = pb + pt[4] + 16 * chan
            if chan < 2:
                p_lo, p_n = pt[5], pt[7]
            else:
                p_lo, p_n = 1, (pt[7] if chan == 2 else pt[8])
            if send_prev > 0:
                d.src_prev = mybuf + 8 * lo
                d.dst_prev = pb + pt[chan] + 8 * (p_lo + p_n)   # its upper halo
                d.bytes_prev = 8 * send_prev
            d.data_flag_at_prev = pfl + 4        # its "data from next"
            d.ack_flag_at_prev = pfl + 12        # its "ack from next"
        if r < w - 1:
            nt = self._tabs[r + 1][l]
            nb = self._peer_bases[r + 1]
            nfl = nb + nt[4] + 16 * chan
            if send_next > 0:
                d.src_next = mybuf + 8 * (lo + n_owned - send_next)
                d.dst_next = nb + nt[chan]                       # its lower halo starts the vector
                d.bytes_next = 8 * send_next
            d.data_flag_at_next = nfl            # its "data from prev"
            d.ack_flag_at_next = nfl + 8         # its "ack from prev"
        return d

    def _exchange_ipc(self, l, vec, n_owned, lo, hi, send_prev, send_next):
        import ctypes as C
        D = self.lv[l]
        chan = D.chan[vec.data_ptr()]
        if D.desc[chan] is None:
            D.desc[chan] = self._ipc_desc(l, chan, n_owned, lo, hi, send_prev, send_next)
        d = D.desc[chan]
        D.epoch[chan] += 1
        d.epoch = D.epoch[chan]
        self.be._chk(self.be.lib.amg_hip_halo_push_wait(C.byref(d), self.be.stream()))

    def _consumed(self, l, vec):
        """The kernel that read vec's halos has been enqueued: acknowledge."""
        if self.comm == "graph":
            D = self.lv[l]
            k = D.kdesc[D.chan[vec.data_ptr()]]
            if k is not None and (k[1] or k[2]):
                self.be._chk(self.be.lib.amg_hip_halo_ack_kernel(k[1], k[2], self.be.stream()))
            return
        if self.comm != "ipc":
            return
        import ctypes as C
        D = self.lv[l]
        d = D.desc[D.chan[vec.data_ptr()]]
        if d is not None:
            self.be._chk(self.be.lib.amg_hip_halo_ack(C.byref(d), self.be.stream()))

    # ---- "graph" mode: exchange / ack / gather as kernels with 0/1 flags ----
    def _exchange_k(self, l, vec, n_owned, lo, hi, send_prev, send_next):
        import ctypes as C
        amg = self.be.amg
        D = self.lv[l]
        chan = D.chan[vec.data_ptr()]
        if D.kdesc[chan] is None:
            r, w = self.rank, self.world
            me = self._tabs[r][l]
            d = amg.HaloKDesc()
            fl = self._base + me[4] + 64 + 16 * chan      # my words: DATA_p, DATA_n, FREE_p, FREE_n
            d.my_data_from_prev, d.my_data_from_next = fl, fl + 4
            d.my_free_from_prev, d.my_free_from_next = fl + 8, fl + 12
            d.timeout = self._base + self._misc[3]
            mybuf = self._base + me[chan]
            d.recv_prev = int(r > 0 and lo > 0)
            d.recv_next = int(r < w - 1 and hi > 0)
            ack_prev = ack_next = None
            if r > 0:
                pt, pb = self._tabs[r - 1][l], self._peer_bases[r - 1]
                pfl = pb + pt[4] + 64 + 16 * chan
                p_lo, p_n = (pt[5], pt[7]) if chan < 2 else (1, pt[7] if chan == 2 else pt[8])
                if send_prev > 0:
                    d.src_prev = mybuf + 8 * lo
                    d.dst_prev = pb + pt[chan] + 8 * (p_lo + p_n)
                    d.cnt_prev = send_prev
                d.data_at_prev = pfl + 4                  # its DATA_from_next
                if d.recv_prev:
                    ack_prev = pfl + 12                   # its FREE_from_next
            if r < w - 1:
                nt, nb = self._tabs[r + 1][l], self._peer_bases[r + 1]
                nfl = nb + nt[4] + 64 + 16 * chan
                if send_next > 0:
                    d.src_next = mybuf + 8 * (lo + n_owned - send_next)
                    d.dst_next = nb + nt[chan]
                    d.cnt_next = send_next
                d.data_at_next = nfl                      # its DATA_from_prev
                if d.recv_next:
                    ack_next = nfl + 8                    # its FREE_from_prev
            D.kdesc[chan] = (d, ack_prev, ack_next)
        d = D.kdesc[chan][0]
        self.be._chk(self.be.lib.amg_hip_halo_exchange_kernel(C.byref(d), self.be.stream()))

    def _gather_desc(self):
        amg = self.be.amg
        r, w = self.rank, self.world
        d = amg.GatherKDesc()
        d.rank, d.world = r, w
        d.src = self.gather_in.data_ptr()
        d.cnt = self.tail_counts[r]
        d.off = sum(self.tail_counts[:r])
        for g in range(w):
            base = self._base if g == r else self._peer_bases[g]
            m = self._miscs[g]
            d.dst[g] = base + m[0]
            d.data_at[g] = base + m[1] + 4 * r
            d.free_at[g] = base + m[2] + 4 * r
        d.my_data_from = self._base + self._misc[1]
        d.my_free_from = self._base + self._misc[2]
        d.timeout = self._base + self._misc[3]
        return d

    def timed_out(self):
        """True when a bounded spin of the in-kernel protocol gave up on this rank."""
        if self.arena is None or self.comm != "graph":
            return False
        t = torch.as_tensor(_DevArray(self._base + self._misc[3], 1, "<u4"), device=self.be.device)
        return bool(int(t.cpu()[0]) != 0)

    def close(self):
        if self.graph_exec is not None:
            self.be.sync()
            self.be.lib.amg_hip_graph_destroy(self.graph_exec)
            self.graph_exec = None
        if self.arena is not None:
            self.be.sync()
            dist.barrier(group=self.group)   # nobody is still pushing into anybody's arena
            self._ipc_cleanup()
            dist.barrier(group=self.group)

    # ---- halo exchange of an extended vector [lo | owned | hi] ----
    def _exchange(self, vec, n_owned, lo, hi, send_prev, send_next):
        ops = []
        r, w = self.rank, self.world
        if r > 0:
            if lo > 0:
                ops.append(dist.P2POp(dist.irecv, vec[0:lo], r - 1, self.group))
            if send_prev > 0:
                ops.append(dist.P2POp(dist.isend, vec[lo:lo + send_prev], r - 1, self.group))
        if r < w - 1:
            if hi > 0:
                ops.append(dist.P2POp(dist.irecv, vec[lo + n_owned:lo + n_owned + hi], r + 1, self.group))
            if send_next > 0:
                ops.append(dist.P2POp(dist.isend, vec[lo + n_owned - send_next:lo + n_owned], r + 1, self.group))
        if not ops:
            return
        if self.host_staged:
            self.be.sync()
            staged, back = [], []
            for op in ops:
                h = op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                staged.append(dist.P2POp(op.op, h, op.peer, self.group))
                if op.op is dist.irecv:
                    back.append((op.tensor, h))
            for req in dist.batch_isend_irecv(staged):
                req.wait()
            for dst, h in back:
                dst.copy_(h)
            return
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def _all_gather(self, out, inp):
        if self.host_staged:
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, inp.cpu(), group=self.group)
            out.copy_(ho)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def _all_reduce_sum(self, t):
        if self.host_staged:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            return h
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _exchange_u(self, l, vec):
        D = self.lv[l]
        h = self.halo[l]
        r = self.rank
        send_prev = h[r - 1][1] if r > 0 else 0               # what rank-1 wants above its block
        send_next = h[r + 1][0] if r < self.world - 1 else 0  # what rank+1 wants below its block
        if self.comm == "ipc":
            self._exchange_ipc(l, vec, D.e - D.s, D.A.halo_lo, D.A.halo_hi, send_prev, send_next)
        elif self.comm == "graph":
            self._exchange_k(l, vec, D.e - D.s, D.A.halo_lo, D.A.halo_hi, send_prev, send_next)
        else:
            self._exchange(vec, D.e - D.s, D.A.halo_lo, D.A.halo_hi, send_prev, send_next)

    def _exchange_1(self, l, vec, n_owned):
        if self.comm == "ipc":
            self._exchange_ipc(l, vec, n_owned, 1, 1, 1, 1)
        elif self.comm == "graph":
            self._exchange_k(l, vec, n_owned, 1, 1, 1, 1)
        else:
            self._exchange(vec, n_owned, 1, 1, 1, 1)

    # ---- smoother: `sweeps` two-buffer Jacobi passes on level l ----
    def _smooth(self, l, from_zero=False):
        D = self.lv[l]
        lo, n = D.A.halo_lo, D.e - D.s
        for k in range(self.sweeps):
            if k == 0 and from_zero:
                # u_l == 0 (multigrid.hpp:278): the sweep needs f and the diagonal only,
                # and no halo (same bits as the full sweep on zeros)
                self.be.jacobi_from_zero(D.diag, D.f, D.u2[lo:lo + n], self.omega)
                D.u, D.u2 = D.u2, D.u
                continue
            self._exchange_u(l, D.u)
            self.be.jacobi(D.mat, D.u, D.f, D.u2[lo:lo + n], self.omega, D.A.diag_shift)
            self._consumed(l, D.u)
            D.u, D.u2 = D.u2, D.u

    def vcycle(self):
        """One V-cycle.  "graph" mode: the first call runs eagerly (loads every kernel,
        exercises the protocol), then the body is captured once and replayed."""
        if self.comm == "graph" and self.n_dist:
            import ctypes as C
            lib, st = self.be.lib, self.be.stream()
            if self._cycles_run == 0:
                self._vcycle_body()
            else:
                if self.graph_exec is None:
                    self.be._chk(lib.amg_hip_capture_begin(st))
                    try:
                        self._vcycle_body()
                    finally:
                        ex = C.c_void_p()
                        self.be._chk(lib.amg_hip_capture_end(st, C.byref(ex)))
                    self.graph_exec = ex
                self.be._chk(lib.amg_hip_graph_launch(self.graph_exec, st))
            self._cycles_run += 1
            return
        self._vcycle_body()
        self._cycles_run += 1

    def _vcycle_body(self):
        be = self.be
        nd = self.n_dist
        for l in range(nd):                                   # multigrid.hpp:265
            D = self.lv[l]
            lo, n = D.A.halo_lo, D.e - D.s
            self._smooth(l, from_zero=(l >= 1 and self.sweeps >= 1))  # :268
            self._exchange_u(l, D.u)
            be.residual(D.mat, D.u, D.f, D.r[1:1 + n])       # :272-274
            self._consumed(l, D.u)
            self._exchange_1(l, D.r, n)
            nH = D.ce - D.cs
            if l + 1 < nd:                                    # :278, :281-282
                C = self.lv[l + 1]
                if self.sweeps < 1:
                    C.u.zero_()
                be.spmv(D.R, D.r, C.f)
            else:
                be.spmv(D.R, D.r, self.gather_in[:nH])
            self._consumed(l, D.r)
        # ---- agglomerated levels (coarse solve included), redundantly on every rank ----
        if nd == 0:
            self.tail.cycle(self.tail_f, self.tail_u, zero_guess=False)
            return
        elif self.comm == "graph":
            import ctypes as C
            if not hasattr(self, "_gdesc"):
                self._gdesc = self._gather_desc()
            be._chk(be.lib.amg_hip_gather_kernel(C.byref(self._gdesc), be.stream()))
            self.tail.cycle(self.tail_f, self.tail_u)
            be._chk(be.lib.amg_hip_gather_ack_kernel(C.byref(self._gdesc), be.stream()))
        else:
            self._all_gather(self.gather_out, self.gather_in)
            off = 0
            for g, c in enumerate(self.tail_counts):
                self.tail_f[off:off + c].copy_(self.gather_out[g * self.tail_max:g * self.tail_max + c])
                off += c
            self.tail.cycle(self.tail_f, self.tail_u)
        for l in range(nd - 1, -1, -1):                       # :291
            D = self.lv[l]
            lo, n = D.A.halo_lo, D.e - D.s
            nH = D.ce - D.cs
            if l + 1 < nd:
                C = self.lv[l + 1]
                clo = C.A.halo_lo
                D.uH[1:1 + nH].copy_(C.u[clo:clo + nH])
                self._exchange_1(l, D.uH, nH)
                exchanged = True
            else:  # every rank holds the whole level: halo entries are local reads
                g0 = max(D.cs - 1, 0)
                g1 = min(D.ce + 1, self.tail_n)
                D.uH[(g0 - (D.cs - 1)):(g1 - (D.cs - 1))].copy_(self.tail_u[g0:g1])
                exchanged = False
            be.spmv(D.P, D.uH, D.tmp)                         # :294-296
            if exchanged:
                self._consumed(l, D.uH)
            be.add_(D.u[lo:lo + n], D.tmp)
            self._smooth(l)                                   # :300

    # ---- diagnostics ----
    def rss(self):
        """AMG::rss(A_0, u_0, b) (common.hpp:17-27), all_reduce over ranks."""
        if self.n_dist == 0:
            return self.tail.rss()
        D = self.lv[0]
        n = D.e - D.s
        self._exchange_u(0, D.u)
        self.be.residual(D.mat, D.u, D.f, D.r[1:1 + n])
        self._consumed(0, D.u)
        part = self._all_reduce_sum(self.be.sumsq(D.r[1:1 + n]).to(torch.float64))
        return float(part.item())

    def solution_checksum(self):
        """Partition-independent fingerprint of the level-0 solution: the sum of the 64-bit
        patterns of all entries modulo 2^64 (owned rows, all-reduced when sharded).  Two
        configurations that hold the same bits have the same fingerprint however the rows are
        cut; rss does not have that property (its partial sums follow the partition)."""
        if self.n_dist == 0:
            return int(self.be.to_numpy(self.tail_u).view(np.int64).sum(dtype=np.int64))
        D = self.lv[0]
        lo, n = D.A.halo_lo, D.e - D.s
        part = D.u[lo:lo + n].contiguous().view(torch.int64).sum().reshape(1)
        return int(self._all_reduce_sum(part).item())

    def gather_solution(self):
        """Level-0 solution on every rank (tests only)."""
        if self.n_dist == 0:
            return self.be.to_numpy(self.tail_u).copy()
        D = self.lv[0]
        lo, n = D.A.halo_lo, D.e - D.s
        counts = [self.bounds[0][g + 1] - self.bounds[0][g] for g in range(self.world)]
        mx = max(counts)
        buf = self.be.vec(mx)
        buf[:n].copy_(D.u[lo:lo + n])
        out = self.be.vec(mx * self.world)
        self._all_gather(out, buf)
        out = self.be.to_numpy(out)
        return np.concatenate([out[g * mx:g * mx + c] for g, c in enumerate(counts)])


# --------------------------------------------------------------------- bench ---
class TransportWatchdog:
    """Bounds the time an alternative halo transport (hipIpc stream operations, in-graph
    spin-flag kernels) may take.  They have never run on real xGMI links here, so a hang
    must not be reported as success: on expiry the rank says which transport hung and
    where (stderr), rank 0 may still print the RCCL (p2p) line it already has, and the
    process leaves with status 3."""

    EXIT_STATUS = 3

    def __init__(self, rank, timeout_s, stash):
        self.rank, self.timeout_s, self.stash = rank, float(timeout_s), stash
        self._timer, self.mode = None, None

    def arm(self, mode):
        import threading
        self.disarm()
        self.mode = mode
        self._timer = threading.Timer(self.timeout_s, self._expired)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def _expired(self):
        st = self.stash
        flag = None
        try:
            flag = st["dv"].timed_out() if st.get("dv") is not None else None
        except Exception:
            pass
        sys.stderr.write(f"[dist_vcycle] rank {self.rank}: halo transport '{self.mode}' did not finish "
                         f"within {self.timeout_s:.0f} s (device-side spin timeout flag: {flag}; "
                         f"distributed levels: {st.get('n_dist')}); leaving with status "
                         f"{self.EXIT_STATUS}\n")
        sys.stderr.flush()
        if self.rank == 0 and st.get("json") is not None:
            print(st["json"], flush=True)
        os._exit(self.EXIT_STATUS)


def _result_line(args, world, L, dv, results, best, notes, rehearsal, avg_ms, sweep_bytes, t0,
                 layout=None):
    from bench import HBM_PEAK_GBS, metric_string
    dt, rss0, rss = results[best][:3]
    roof = None
    if avg_ms:
        lay, mat_bytes, rows = layout if layout else ("?", None, None)
        kern = {"dict": "dict_kernel<CSR_JACOBI> (dictionary-coded rows)",
                "sell": "sell_kernel<CSR_JACOBI> (SELL-64 panels)",
                "csr": "csr_stage_kernel<CSR_JACOBI> (LDS-staged CSR)"}.get(lay, "Jacobi sweep")
        # bytes the timed kernel has to move (bench.fine_sweep_roofline): the dictionary-coded
        # layout streams its matrix bytes + f + x + out, the CSR layouts the CSR formula
        must = (mat_bytes + 24 * rows) if (lay == "dict" and mat_bytes is not None) else sweep_bytes
        achieved = must / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": f"{kern}: rank 0's row block of the level-0 Jacobi sweep",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": must, "avg_launch_ms": avg_ms,
                "layout": lay, "csr_formula_bytes_per_sweep": sweep_bytes,
                "note": "per rank: rank 0's row block; PMC traffic is collected on the N=1 line"}
    n_dist = notes.get(best + "_distributed_levels", dv.n_dist)   # of the reported transport
    dim = getattr(args, "dim", 2)
    multicolor = getattr(args, "smoother", "jacobi") == "multicolor"
    problem = (f"2D 5-point Poisson {args.n}x{args.n} (Grid::laplacian/rhs)" if dim == 2 else
               f"3D 7-point Poisson {args.n}^3 (the 7-point analogue of Grid::laplacian/rhs)")
    smoother = ("multicolour symmetric GS 1+1 passes" if multicolor else
                f"true Jacobi omega={args.omega} {args.sweeps}+{args.sweeps} sweeps")
    return {
        "metric": metric_string(args.n, dim),
        "value": args.steps / dt, "unit": "V-cycles/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": (f"{problem}, {smoother}, {L}-level V-cycle, "
                         f"row-block shards over {world} GPUs ({n_dist} distributed levels, "
                         f"rest agglomerated), fp64"),
            "n": args.n, "dim": dim, "smoother": "multicolor" if multicolor else "jacobi",
            "levels": L, "distributed_levels": n_dist, "rehearsal": rehearsal,
            "dist_min_rows": args.dist_min_rows, "setup_seconds": time.time() - t0,
            "halo_exchange": best, "exchange_modes": notes,
            "vcycles_per_sec_by_exchange": {k: args.steps / v[0] for k, v in results.items()},
            "rss_after_warmup": rss0, "rss_after_steps": rss,
        },
        "roofline": roof,
    }


def bench(args):
    """bench.py --gpus N (N > 1): launched by torch.distributed.run, one rank per GPU."""
    import amg_ctypes as amg
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # AMG_DIST_REHEARSAL=1: every rank on GPU 0, gloo + host-staged exchange.  Only for
    # rehearsing this code path on a one-GPU box (RCCL refuses two ranks on one device).
    rehearsal = os.environ.get("AMG_DIST_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    be = HipBackend(local)
    from bench import n_levels_for, HBM_PEAK_GBS
    import window_vcycle
    t0 = time.time()
    dim = getattr(args, "dim", 2)
    multicolor = getattr(args, "smoother", "jacobi") == "multicolor"
    # BASELINE configs 4 / 5 (multicolour GS, 3-D): replicated and window-sharded only -- the
    # per-sweep and slab designs are true-Jacobi 2-D and start from a host copy of the hierarchy
    general = dim == 3 or multicolor
    w_smoother = window_vcycle.SM_MULTICOLOR if multicolor else window_vcycle.SM_JACOBI
    w_iters = 1 if multicolor else args.sweeps
    L = args.levels or n_levels_for(args.n, dim=dim)
    hier = None
    if not general:
        colptr, rowind, val = amg.laplacian(args.n)
        b = amg.rhs(args.n)
        hier = amg.Multigrid(colptr, rowind, val, b, L, smoother=amg.SM_JACOBI,
                             smoother_iters=args.sweeps, omega=args.omega, host_only=True)
        del colptr, rowind, val
    def timed(dv):
        for _ in range(args.warmup):
            dv.vcycle()
        r0 = dv.rss()
        be.sync()
        dist.barrier()
        be.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dv.vcycle()
        be.sync()
        dist.barrier()
        be.sync()
        el = torch.tensor([time.perf_counter() - t1], dtype=torch.float64,
                          device="cpu" if rehearsal else be.device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), r0, dv.rss(), dv.solution_checksum()

    def same_result(a, b):
        # same bits of the level-0 solution; rss only to rounding (its summation order follows
        # the partition, so two shardings of the same vector may differ in the last digit)
        return a[3] == b[3] and abs(a[2] - b[2]) <= 1e-12 * abs(b[2]) and abs(a[1] - b[1]) <= 1e-12 * abs(b[1])

    import json as _json
    results, notes = {}, {}
    stash = {"json": None}
    dog = TransportWatchdog(rank, args.comm_timeout, stash)

    def agree(name):
        """Every rank must agree that candidate `name` stands (a rank that failed half way leaves
        the others inside a collective: the watchdog, still armed, ends that with status 3)."""
        flag = torch.tensor([1 if name in results else 0], dtype=torch.int32,
                            device="cpu" if rehearsal else be.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and name in results:
            del results[name]
            notes[name] = "discarded: failed or was refused on another rank"

    def stash_line(note):
        best_so_far = min(results, key=lambda k: results[k][0])
        stash["json"] = _json.dumps(_result_line(args, world, L, dvr, results, best_so_far,
                                                 dict(notes, note=note), rehearsal, None, None, t0)) if rank == 0 else None
        stash["dv"], stash["n_dist"] = None, None

    # (1) Replicated: nothing is distributed, every rank runs the whole (fused, K-Patch) single-GPU
    # cycle.  No data-path exchange at all, so it cannot hang; it is the single-GPU result every
    # sharded candidate has to reproduce bit for bit, and the line that is printed if one hangs.
    if general:
        dvr = window_vcycle.ReplicatedVcycle(amg, be.device, be._stream, dim, args.n, L, w_smoother, w_iters,
                                             args.omega)
    else:
        dvr = DistributedVcycle(hier, b, be, rank, world, omega=args.omega, sweeps=args.sweeps,
                                dist_min_rows=1 << 62, host_staged=rehearsal, comm="p2p")
    results["replicated"] = timed(dvr)
    notes["replicated"] = "ok"
    notes["replicated_distributed_levels"] = dvr.n_dist
    ref = results["replicated"]

    # (2) Per-sweep halo exchange through torch.distributed isend/irecv + all_gather (RCCL): one
    # exchange before every sweep, residual and transfer of the distributed levels.
    dv = dvr
    if world > 1 and not general:
        stash_line("p2p exchange hung")
        dog.arm("p2p")
        try:
            if os.environ.get("AMG_DIST_FORCE_HANG") == "p2p":
                time.sleep(args.comm_timeout + 30)
            dv = DistributedVcycle(hier, b, be, rank, world, omega=args.omega, sweeps=args.sweeps,
                                   dist_min_rows=args.dist_min_rows, host_staged=rehearsal, comm="p2p")
            notes["p2p_distributed_levels"] = dv.n_dist
            res = timed(dv)
            if same_result(res, ref):
                results["p2p"] = res
                notes["p2p"] = "ok"
            else:
                notes["p2p"] = "ran but did not reproduce the single-GPU result; discarded"
        except Exception as ex:   # noqa: BLE001 -- a candidate that fails must not cost the line we already have
            notes["p2p"] = f"failed: {type(ex).__name__}: {ex}"
            dv = dvr
        try:
            agree("p2p")
        finally:
            dog.disarm()

    # Slab sharding (slab_vcycle.py): the K-Patch levels over this rank's grid lines + redundant
    # halo, two exchanges per cycle (grouped send/recv of the level-0 halo lines, one all-gather),
    # the rest replicated.  Same RCCL primitives as "p2p"; kept only if it reproduces its result.
    dvs = None
    if world > 1 and args.comm in ("safe", "auto", "slab", "library") and args.sweeps == 2 and not general:
        import slab_vcycle
        stash_line("slab exchange hung")
        dog.arm("slab")
        try:
            if os.environ.get("AMG_DIST_FORCE_HANG") == "slab":
                time.sleep(args.comm_timeout + 30)
            eng = slab_vcycle.HipSlabEngine(amg, be.device, be._stream, args.n, L, args.omega, args.sweeps,
                                            rank, world, args.slab_levels,
                                            patch_min_rows=args.slab_patch_min_rows)
            dvs = slab_vcycle.SlabVcycle(eng, rank, world, host_staged=rehearsal)
            notes["slab_distributed_levels"] = dvs.n_dist
            notes["slab_halo_lines"] = int(eng.info.halo_lines)
            res = timed(dvs)
            if same_result(res, ref):
                results["slab"] = res
                notes["slab"] = "ok"
            else:
                notes["slab"] = "ran but did not reproduce the single-GPU result; discarded"
        except amg.AmgHipError as ex:
            kind = "unavailable" if ex.status == amg.EUNSUPPORTED else "failed"   # EUNSUPPORTED: host arithmetic, every rank alike
            notes["slab"] = f"{kind}: {ex.message}"
        except Exception as ex:   # noqa: BLE001 -- a candidate that fails must not cost the line we already have
            notes["slab"] = f"failed: {type(ex).__name__}: {ex}"
        try:
            agree("slab")
        finally:
            dog.disarm()

    # Window sharding (window_vcycle.py): every rank sets up and stores only its window of the
    # distributed levels; general kernels, so multicolour GS and 3-D shard too.  Two exchanges per
    # cycle with the same RCCL primitives; kept only if it reproduces the replicated result.
    dvw = None
    if world > 1 and args.comm in ("safe", "auto", "window", "library"):
        stash_line("window exchange hung")
        dog.arm("window")
        try:
            if os.environ.get("AMG_DIST_FORCE_HANG") == "window":
                time.sleep(args.comm_timeout + 30)
            k = args.window_levels
            if k < 0:
                k = window_vcycle.auto_levels(dim, args.n, world, L, w_smoother, w_iters, args.window_min_rows)
            if k < 1:
                raise ValueError("no level is large enough to distribute (or the halo does not fit a block)")
            plan = window_vcycle.WindowPlan(dim, args.n, rank, world, k, w_smoother, w_iters)
            eng = window_vcycle.HipWindowEngine(amg, be.device, be._stream, plan, args.omega,
                                                patch_min_rows=args.slab_patch_min_rows)
            comm = window_vcycle.TorchComm(rank, world, host_staged=rehearsal, sync=be.sync,
                                           device=None if rehearsal else be.device)
            dvw = window_vcycle.WindowVcycle(eng, plan, comm, L)
            notes["window_distributed_levels"] = k
            notes["window_halo_units"] = plan.halo
            notes["window_rows_level0_per_rank"] = plan.window_rows(0)
            res = timed(dvw)
            if same_result(res, ref):
                results["window"] = res
                notes["window"] = "ok"
                notes["window_must_move_bytes_per_rank"] = (eng.mg.cycle_must_move(1) + eng.mg.cycle_must_move(3)
                                                            + eng.tail.cycle_must_move(0))
            else:
                notes["window"] = "ran but did not reproduce the single-GPU result; discarded"
        except (amg.AmgHipError, ValueError, window_vcycle.WindowUnsupported) as ex:
            notes["window"] = f"unavailable: {ex}"
        except Exception as ex:   # noqa: BLE001 -- a candidate that fails must not cost the line we already have
            notes["window"] = f"failed: {type(ex).__name__}: {ex}"
        try:
            agree("window")
        finally:
            dog.disarm()

    # The same two sharded cycles with the exchanges issued by the LIBRARY (amg_hip_slab_cycle /
    # amg_hip_window_cycle over its own RCCL communicator: one C call per cycle, no Python and no
    # torch.distributed on the data path).  The unique id travels through torch.distributed.
    # Not part of the default ("safe") candidate set: it has run with a world of 1 only
    # (tests/test_gpu_comm.py), and a fault inside a second communicator would cost the line the
    # candidates above already earned.  --comm auto / slab / window / library time it.
    if world > 1 and not rehearsal and args.comm in ("auto", "slab", "window", "library") and \
            (("slab" in results and dvs is not None) or ("window" in results and dvw is not None)):
        stash_line("in-library RCCL exchange hung")
        dog.arm("library RCCL")
        libc = None
        try:
            if os.environ.get("AMG_DIST_FORCE_HANG") == "rccl":
                time.sleep(args.comm_timeout + 30)
            idt = torch.zeros(128, dtype=torch.uint8, device=be.device)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(amg.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            libc = amg.Comm(bytes(idt.cpu().numpy().tobytes()), rank, world, local)
            for name, d in (("slab", dvs), ("window", dvw)):
                if name not in results or d is None:
                    continue
                d.use_library_comm(libc)
                res = timed(d)
                if same_result(res, ref):
                    results[name + "_rccl"] = res
                    notes[name + "_rccl"] = "ok"
                    notes[name + "_rccl_distributed_levels"] = notes.get(name + "_distributed_levels")
                else:
                    notes[name + "_rccl"] = "ran but did not reproduce the single-GPU result; discarded"
                    d.libcomm = None
        except amg.AmgHipError as ex:
            notes["library_rccl"] = f"unavailable: {ex.message}"
        except Exception as ex:   # noqa: BLE001
            notes["library_rccl"] = f"failed: {type(ex).__name__}: {ex}"
        try:
            agree("slab_rccl")
            agree("window_rccl")
        finally:
            dog.disarm()

    if dv.n_dist and "p2p" in results and args.comm not in ("p2p", "safe", "slab", "window", "library"):
        stash_line("alternative exchange hung")
        # cheaper exchanges pay off on smaller levels (results do not depend on the
        # threshold); the in-graph exchange is timed with two thresholds because the
        # cost of a flag round trip over xGMI cannot be rehearsed on one GPU
        cand = {"ipc": [("ipc", args.dist_min_rows_ipc)],
                "graph": [("graph", args.dist_min_rows_graph), ("graph@8x", 8 * args.dist_min_rows_graph)]}
        modes = cand["ipc"] + cand["graph"] if args.comm == "auto" else cand[args.comm]
        for mode, min_rows in modes:
            stash["dv"], stash["n_dist"] = None, None
            dog.arm(mode)
            try:
                if os.environ.get("AMG_DIST_FORCE_HANG") == mode:   # test hook for the watchdog
                    time.sleep(args.comm_timeout + 30)
                dvx = DistributedVcycle(hier, b, be, rank, world, omega=args.omega,
                                        sweeps=args.sweeps, dist_min_rows=min_rows,
                                        host_staged=rehearsal, comm=mode.split("@")[0])
                notes[mode + "_distributed_levels"] = dvx.n_dist
                stash["dv"], stash["n_dist"] = dvx, dvx.n_dist
                res = timed(dvx)
                same = same_result(res, ref) and not dvx.timed_out()
                flag = torch.tensor([1 if same else 0], dtype=torch.int32,
                                    device="cpu" if rehearsal else be.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    results[mode] = res
                    notes[mode] = "ok"
                else:
                    notes[mode] = "ran but did not reproduce the single-GPU result; discarded"
                dvx.close()
            except IpcUnavailable as ex:
                notes[mode] = f"unavailable: {ex}"
            finally:
                dog.disarm()
    if hier is not None:
        hier.close()
    best = min(results, key=lambda k: results[k][0])
    dt, rss0, rss = results[best][:3]
    # dominant kernel: this rank's level-0 Jacobi sweep (HIP events on the rank's stream)
    D = dv.lv[0] if (dv.n_dist and best in ("p2p", "ipc", "graph", "graph@8x")) else None
    avg_ms, sweep_bytes = None, None
    if D is not None:
        lo, n = D.A.halo_lo, D.e - D.s
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.profile_launches)]
        for a_, c_ in ev:
            a_.record(be._stream)
            be.jacobi(D.mat, D.u, D.f, D.u2[lo:lo + n], dv.omega, D.A.diag_shift)
            c_.record(be._stream)
        be.sync()
        ms = [a_.elapsed_time(c_) for a_, c_ in ev]
        avg_ms = sum(ms) / len(ms)
        sweep_bytes = 12.0 * D.A.nnz + 28.0 * n
    # nothing distributed (replicated, or the threshold left no level to shard): every rank runs
    # the single-GPU solver, and the dominant kernel is that solver's (same object as at N = 1)
    best_nd = notes.get(best + "_distributed_levels")
    whole = dvr if best_nd == 0 else None
    roof_whole = None
    if general and best == "replicated":
        whole = dvr
    best0 = best.replace("_rccl", "")
    if best0 in ("slab", "window") or (whole is not None and (general or hasattr(whole.tail, "mg"))):
        from bench import fine_sweep_roofline
        mgw = dvs.eng.mg if best0 == "slab" else (dvw.eng.mg if best0 == "window" else
                                                  (whole.mg if general else whole.tail.mg))
        lay_id, mat_b = mgw.level_layout(0)
        lay_nm = {amg.LAYOUT_CSR: "csr", amg.LAYOUT_SELL: "sell", amg.LAYOUT_DICT: "dict"}[lay_id]
        roof_whole = fine_sweep_roofline(amg, mgw, args, lay_nm, mat_b, mgw.get_n_dofs(0), mgw.cycle_bytes()[1],
                                         launches=max(8, args.profile_launches // 2))
        roof_whole["note"] = ("per rank: rank 0's lines + halo of the level-0 down-leg (slab sharding)" if best0 == "slab"
                              else "per rank: the level-0 launch over rank 0's window (owned units + halo)" if best0 == "window"
                              else "per rank: every rank runs the whole cycle (nothing is distributed)")
    out = None
    if rank == 0:
        if args.warmup >= 1 and not (rss < rss0):
            raise SystemExit(f"V-cycle iteration is not converging (rss {rss0:.3e} -> {rss:.3e})")
        lay = None
        if D is not None:
            name, mat_bytes = be.matrix_layout(D.mat)
            lay = (name, mat_bytes, D.e - D.s)
        out = _result_line(args, world, L, dv, results, best, notes, rehearsal, avg_ms, sweep_bytes, t0,
                           layout=lay)
        if roof_whole is not None:
            out["roofline"] = roof_whole
    if dv is not dvr:
        dv.close()
    dvr.close()
    if dvs is not None:
        dvs.close()
    if dvw is not None:
        dvw.close()
    dist.barrier()
    dist.destroy_process_group()
    return out
