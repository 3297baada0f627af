"""Window-sharded V-cycle: row blocks with sharded STORAGE and SETUP.

The reference's flat-index coarsening (multigrid.hpp:127-130, interpolator.hpp:116-129) halves the
fast axis only, so the "units" of the slowest axis -- grid lines in 2-D, x-y planes in 3-D -- are
the same on every level and a block of whole units cuts the whole hierarchy consistently.  Rank g
owns units [g*chunk, (g+1)*chunk) and builds an ORDINARY solver (include/amg_hip.h:
amg_hip_create_poisson_window) on its WINDOW = owned units + `halo` units either side: the
principal submatrix of Grid::laplacian(n) and the slice of Grid::rhs(n).  The Galerkin rows of the
window's hierarchy equal the global ones bit for bit away from the window's edges (same entries,
same summation order), so nothing global is ever assembled above the gathered level.

One V-cycle (multigrid.hpp:263-305) with k distributed levels:

  1. halo: `halo` units of the level-0 solution from rank-1 / rank+1 (grouped send/recv);
  2. engine.run(1): down-legs of levels 0..k-1 over the whole window with the single-GPU kernels,
     whatever the smoother (true Jacobi, multicolour GS), whatever the dimension.  Each sweep /
     colour stage / residual / transfer makes one more unit at either end of the window stale;
     WindowPlan chooses the halo so that what is needed stays valid;
  3. all-gather of the owned units of f_k; levels >= k run replicated on every rank (the TAIL
     solver, built once from the all-gathered rows of A_k), coarse solve included;
  4. the window of u_k goes back into level k of the window solver; engine.run(3): up-legs.

Two exchanges per cycle, like slab_vcycle.py, but every rank allocates and sets up only its
window of the distributed levels (slab_vcycle.py replicates everything), and the legs are the
general kernels, so BASELINE configs 4 (multicolour GS, 8192^2) and 5 (3-D 7-point, 512^3) shard
too.  Per-row arithmetic is the single-GPU kernels': the owned units of the result equal the
single-GPU cycle bit for bit (tests/test_window_gloo.py with an oracle engine,
tests/test_gpu_window.py on the device).

All compute sits behind an `engine` (HipWindowEngine below; tests/window_engine.py is the CPU
stand-in), all communication behind a `comm` (TorchComm: torch.distributed, "nccl" = RCCL over
xGMI; tests/window_engine.py: ThreadComm runs several ranks in one process).
"""
import numpy as np
import torch
import torch.distributed as dist

SM_JACOBI, SM_MULTICOLOR = 3, 4   # amg_ctypes.SM_JACOBI / SM_MULTICOLOR_GS


def stage_losses(smoother, iters, n_colors, level):
    """Units of validity one leg of `level` costs at either end of the window:
    (S, D, U) = pre-smoothing alone, whole down-leg (pre-smoothing + residual + restriction),
    whole up-leg (prolongation + post-smoothing).  A sweep, a colour stage or a residual reaches
    one unit and one entry (the corner couplings of the Galerkin operators) beyond a row, the
    transfers one or two entries: the entries of a whole leg stay below one unit, which each of
    D and U carries as its "+ 1"."""
    if smoother == SM_JACOBI:
        # on a coarse level the first sweep starts from u == 0 everywhere (multigrid.hpp:278):
        # its result depends on f alone, no validity is lost
        S = iters if level == 0 else max(iters - 1, 0)
        P = iters
    elif smoother == SM_MULTICOLOR:
        S = P = 2 * n_colors * iters      # colours 0..nc-1, then nc-1..0: 2 nc dependent stages
    else:
        raise ValueError("the lexicographic smoothers do not shard (SURVEY F9)")
    return S, S + 2, P + 1


class WindowPlan:
    """Pure arithmetic (no device, no communication): which units a rank owns, how deep its
    halo has to be, where its window starts.  `colors[l]`: colours assumed on level l
    (multicolour GS; checked against the solver after setup)."""

    def __init__(self, dim, n, rank, world, k, smoother=SM_JACOBI, iters=2, colors=None, even_start=None,
                 tamper=0):
        if dim not in (2, 3) or n < 1 or world < 1 or not (0 <= rank < world) or k < 1:
            raise ValueError("WindowPlan: bad argument")
        self.dim, self.n, self.rank, self.world, self.k = dim, n, rank, world, k
        self.smoother, self.iters = smoother, iters
        self.units = n
        self.unit_rows = n * n if dim == 3 else n            # level-0 rows per unit
        if self.unit_rows % (1 << k):
            raise ValueError(f"WindowPlan: {self.unit_rows} rows per unit do not halve {k} times")
        if colors is None:
            colors = [2] + [4 if dim == 2 else 8] * (k - 1)
        self.colors = list(colors)
        S, D, U = zip(*[stage_losses(smoother, iters, self.colors[l], l) for l in range(k)])
        self.S, self.D, self.U = S, D, U
        up_need = [0] * (k + 1)                               # units of the final u_l beyond the owned block
        for l in range(k):
            up_need[l + 1] = up_need[l] + U[l]
        need = [0] * (k + 1)                                  # units of the down-leg input of level l
        for l in range(k - 1, -1, -1):
            need[l] = max(up_need[l] + U[l] + S[l], need[l + 1] + D[l])
        self.up_need, self.need = up_need, need
        self.halo = need[0] - tamper                          # tamper: tests only (a halo that is too shallow)
        self.tamper = tamper
        self.chunk = -(-n // world)
        if world > 1 and (n - self.chunk * (world - 1) < max(self.halo + 1, 1) or self.chunk <= self.halo):
            raise ValueError(f"WindowPlan: {self.chunk} units per rank cannot supply a halo of {self.halo}")
        # multicolour: the greedy colouring of the window has to be the window of the global one;
        # the pattern repeats every two units, so windows start at even units
        self.even_start = (smoother == SM_MULTICOLOR) if even_start is None else bool(even_start)
        self.own0, self.own1 = self._own(rank)
        self.w0, self.w1 = self._window(rank)

    def _own(self, r):
        return min(self.n, self.chunk * r), min(self.n, self.chunk * (r + 1))

    def _window(self, r):
        o0, o1 = self._own(r)
        if self.world == 1:
            return 0, self.n
        w0, w1 = max(0, o0 - self.halo), min(self.n, o1 + self.halo)
        if w0 > 0 and ((w0 * self.unit_rows) & 1 or (self.even_start and (w0 & 1))):
            w0 -= 1                                           # coarse dof j <-> fine dof 2j+1: even start
        return w0, w1

    def pitch(self, l):
        return self.unit_rows >> l

    def global_rows(self, l):
        """n_l of the whole problem: n_H = (n_h + 1) / 2 - 1 (multigrid.hpp:127-130)"""
        n = self.units * self.unit_rows
        for _ in range(l):
            n = (n + 1) // 2 - 1
        return n

    def window_rows(self, l):
        n = (self.w1 - self.w0) * self.unit_rows
        for _ in range(l):
            n = (n + 1) // 2 - 1
        return n

    def owned_local(self, l):
        """rows [a, b) of the window's level-l vectors that this rank owns"""
        p = self.pitch(l)
        return (self.own0 - self.w0) * p, min((self.own1 - self.w0) * p, self.window_rows(l))

    def halo_sizes(self):
        """(units received from rank-1, units received from rank+1, units sent to rank-1,
        units sent to rank+1)"""
        r, w = self.rank, self.world
        rp = self.own0 - self.w0
        rn = self.w1 - self.own1
        sp = (self._window(r - 1)[1] - self._own(r - 1)[1]) if r > 0 else 0
        sn = (self._own(r + 1)[0] - self._window(r + 1)[0]) if r < w - 1 else 0
        return rp, rn, sp, sn

    def patch_ranges(self):
        """window-local unit ranges (2-D: grid lines) the K-Patch legs have to cover, per level"""
        lo = lambda h: max(self.w0, self.own0 - h) - self.w0 if self.world > 1 else 0
        hi = lambda h: min(self.w1, self.own1 + h) - self.w0 if self.world > 1 else self.n
        k = self.k
        return ([lo(self.need[l]) for l in range(k)], [hi(self.need[l]) for l in range(k)],
                [lo(self.up_need[l]) for l in range(k)], [hi(self.up_need[l]) for l in range(k)])


# ------------------------------------------------------------------ comm ---------
class TorchComm:
    """torch.distributed: "nccl" (= RCCL over xGMI) with device tensors, "gloo" with CPU tensors,
    or gloo next to device tensors (host_staged: every exchange goes through host copies).
    device: where a CPU tensor (setup-time flags, colours) is staged for a backend that only moves
    device memory (nccl)."""

    def __init__(self, rank, world, group=None, host_staged=False, sync=None, device=None):
        self.rank, self.world, self.group = rank, world, group
        self.host_staged, self._sync, self.device = bool(host_staged), sync, device

    def _wire(self, t):
        """the tensor the backend can move for t: a host copy (host_staged), a device copy of a CPU
        tensor (nccl), or t itself"""
        if self.host_staged:
            return t if t.device.type == "cpu" else t.cpu()
        if self.device is not None and t.device.type == "cpu":
            return t.to(self.device)
        return t

    def neighbor_exchange(self, send_prev, recv_prev, send_next, recv_next):
        """grouped send/recv with rank-1 / rank+1; tensors or None; received in place"""
        r, w, ops, back = self.rank, self.world, [], []
        if self.host_staged and self._sync:
            self._sync()

        def post(kind, t, peer):
            if t is None or t.numel() == 0:
                return
            if kind is dist.irecv:
                wt = self._wire(t)
                if wt is not t:
                    wt = torch.empty(t.shape, dtype=t.dtype, device=wt.device)
                    back.append((t, wt))
            else:
                wt = self._wire(t)
            ops.append(dist.P2POp(kind, wt, peer, self.group))

        if r > 0:
            post(dist.irecv, recv_prev, r - 1)
            post(dist.isend, send_prev, r - 1)
        if r < w - 1:
            post(dist.irecv, recv_next, r + 1)
            post(dist.isend, send_next, r + 1)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for dst, wt in back:
            dst.copy_(wt)

    def all_gather_blocks(self, inp, out):
        """out = [inp of rank 0 | inp of rank 1 | ...] (equal blocks)"""
        if self.world == 1:
            out.copy_(inp)
            return
        if self.host_staged and self._sync:
            self._sync()
        wi = self._wire(inp)
        if wi is inp:
            dist.all_gather_into_tensor(out, inp, group=self.group)
        else:
            wo = torch.empty(out.shape, dtype=out.dtype, device=wi.device)
            dist.all_gather_into_tensor(wo, wi, group=self.group)
            out.copy_(wo)

    def all_reduce_sum(self, t):
        if self.world == 1:
            return t
        if self.host_staged and self._sync:
            self._sync()
        wt = self._wire(t)
        dist.all_reduce(wt, op=dist.ReduceOp.SUM, group=self.group)
        return wt

    def all_gather_numpy(self, arr, device=None):
        """setup-time gather of one numpy array per rank (different lengths)"""
        arr = np.ascontiguousarray(arr)
        if self.world == 1:
            return [arr]
        dev = "cpu" if (self.host_staged or self.device is None) else self.device
        cnt = torch.tensor([arr.size], dtype=torch.int64, device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        cnts = [int(c.item()) for c in cnts]
        mx = max(max(cnts), 1)
        buf = torch.zeros(mx, dtype=torch.from_numpy(arr[:0]).dtype, device=dev)
        buf[:arr.size] = torch.from_numpy(arr).to(dev)
        out = torch.empty(mx * self.world, dtype=buf.dtype, device=dev)
        dist.all_gather_into_tensor(out, buf, group=self.group)
        out = out.cpu().numpy()
        return [out[g * mx:g * mx + c].copy() for g, c in enumerate(cnts)]


# ---------------------------------------------------------------- driver ---------
class WindowUnsupported(RuntimeError):
    """Raised on every rank alike when the windows do not add up to the global hierarchy
    (colourings that disagree on an overlap)."""


class WindowVcycle:
    """multigrid.hpp:263-305 over window-sharded row blocks."""

    def __init__(self, engine, plan, comm, total_levels):
        self.eng, self.plan, self.comm = engine, plan, comm
        self.rank, self.world = plan.rank, plan.world
        self.L = int(total_levels)
        k = plan.k
        if self.L < k + 1:
            raise ValueError("WindowVcycle: the hierarchy needs at least one replicated level")
        self.n_dist = k
        self.nk = plan.global_rows(k)
        self.pk = plan.pitch(k)
        self.block = plan.chunk * self.pk                    # all-gather block of level k
        self._check_colors()
        self._build_tail()
        dev = engine.u0.device
        self._stage_in = torch.zeros(self.block, dtype=torch.float64, device=dev)
        self._gathered = torch.zeros(self.block * self.world, dtype=torch.float64, device=dev)
        self._cycles_run = 0
        self.libcomm = None

    def use_library_comm(self, libcomm):
        """From now on one V-cycle = ONE C call (amg_hip_window_cycle): the exchanges are the
        library's own RCCL calls on the solver's stream (amg_ctypes.Comm), nothing returns to
        Python in between.  Needs the C-ABI engine (HipWindowEngine)."""
        import amg_ctypes as amg
        p, k = self.plan, self.plan.k
        rp, rn, sp, sn = p.halo_sizes()
        m = p.pitch(0)
        c = amg.WindowPlanC()
        c.own0_off, c.own0_end = p.owned_local(0)
        c.send_prev_cnt, c.recv_prev_cnt, c.send_next_cnt, c.recv_next_cnt = sp * m, rp * m, sn * m, rn * m
        a, b = p.owned_local(k)
        c.own_k_off, c.own_k_cnt, c.block_k, c.uk_off = a, b - a, self.block, p.w0 * self.pk
        self._plan_c, self.libcomm = c, libcomm

    # ---- setup: the replicated levels k..L-1 from the all-gathered rows of A_k ----
    def _build_tail(self):
        p, k = self.plan, self.plan.k
        colptr, rowind, val = self.eng.level_matrix(k)      # CSC of the window's level k
        a, b = p.owned_local(k)
        q0, q1 = int(colptr[a]), int(colptr[b])
        cnt = np.diff(colptr[a:b + 1]).astype(np.int32)
        shift = p.w0 * self.pk                              # window row -> global row
        rows = (rowind[q0:q1].astype(np.int64) + shift).astype(np.int32)
        cnts = self.comm.all_gather_numpy(cnt)
        rws = self.comm.all_gather_numpy(rows)
        vls = self.comm.all_gather_numpy(val[q0:q1])
        cnt_all = np.concatenate(cnts)
        if cnt_all.size != self.nk:
            raise WindowUnsupported(f"gathered level {k} has {cnt_all.size} columns, expected {self.nk}")
        cp = np.zeros(self.nk + 1, np.int64)
        np.cumsum(cnt_all, out=cp[1:])
        if cp[-1] >= 2 ** 31 - 1:
            raise WindowUnsupported("gathered level exceeds int32 indexing")
        self.eng.build_tail(cp.astype(np.int32), np.concatenate(rws), np.concatenate(vls), self.L - k)

    # ---- setup: neighbouring windows must colour their overlap alike ----
    def _plan_of(self, r):
        p = self.plan
        return WindowPlan(p.dim, p.n, r, p.world, p.k, p.smoother, p.iters, p.colors, p.even_start, p.tamper)

    def _check_colors(self):
        p = self.plan
        if p.smoother != SM_MULTICOLOR:
            return
        ok = 1
        prev = self._plan_of(self.rank - 1) if self.rank > 0 else None
        for l in range(p.k):
            col, nc = self.eng.colors(l)
            if nc > p.colors[l]:
                ok = 0                                       # longer colour chains than the halo assumes
            col8 = torch.from_numpy(col.astype(np.int8))
            a, b = p.owned_local(l)
            # received: the rows of my window below / above my block; sent: the rows of my block that
            # the neighbour's window holds (the window of rank-1 may end in the level's missing
            # last row: its plan tells how many rows it has above its block)
            got_p = torch.full((a,), -1, dtype=torch.int8)
            got_n = torch.full((col.size - b,), -1, dtype=torch.int8)
            n_to_prev = (prev.window_rows(l) - prev.owned_local(l)[1]) if prev is not None else 0
            n_to_next = self._plan_of(self.rank + 1).owned_local(l)[0] if self.rank < self.world - 1 else 0
            self.comm.neighbor_exchange(col8[a:a + n_to_prev].clone(), got_p,
                                        col8[b - n_to_next:b].clone(), got_n)
            if self.rank > 0 and not torch.equal(got_p, col8[:a]):
                ok = 0
            if self.rank < self.world - 1 and not torch.equal(got_n, col8[b:]):
                ok = 0
        flag = self.comm.all_reduce_sum(torch.tensor([1 - ok], dtype=torch.int64))
        if int(flag.item()) != 0:
            raise WindowUnsupported("the windows' greedy colourings disagree on an overlap (or use more "
                                    "colours than the halo depth assumes)")

    # ---- exchange 1: halo units of the level-0 solution, in place in the window ----
    def _exchange_u0(self):
        p, u = self.plan, self.eng.u0
        rp, rn, sp, sn = p.halo_sizes()
        m = p.pitch(0)
        a, b = p.owned_local(0)
        self.comm.neighbor_exchange(u[a:a + sp * m] if sp else None, u[a - rp * m:a] if rp else None,
                                    u[b - sn * m:b] if sn else None, u[b:b + rn * m] if rn else None)

    def vcycle(self):
        p, e = self.plan, self.eng
        if self.libcomm is not None:
            self.libcomm.window_cycle(e.mg, e.tail, self._plan_c)
            self._cycles_run += 1
            return
        if self.world > 1:
            self._exchange_u0()
        e.run(1)                                              # multigrid.hpp:265-283, levels < k
        a, b = p.owned_local(p.k)
        self._stage_in[:b - a].copy_(e.fk[a:b])
        self.comm.all_gather_blocks(self._stage_in, self._gathered)
        uk = e.tail_cycle(self._gathered[:self.nk])          # levels >= k incl. :287-288
        off = p.w0 * self.pk
        e.uk.copy_(uk[off:off + e.uk.numel()])
        e.run(3)                                              # :291-302, levels < k
        self._cycles_run += 1

    # ---- diagnostics ----
    def rss(self):
        """AMG::rss(A_0, u_0, b) (common.hpp:17-27): each rank's owned rows, all-reduced (the
        summation order follows the partition: equal to the single-GPU figure to rounding)."""
        if self.world > 1:
            self._exchange_u0()
        a, b = self.plan.owned_local(0)
        part = self.eng.residual_sumsq(a, b).to(torch.float64).reshape(1)
        return float(self.comm.all_reduce_sum(part).item())

    def solution_checksum(self):
        """Sum of the 64-bit patterns of the level-0 solution modulo 2^64: partition-independent
        (dist_vcycle.DistributedVcycle.solution_checksum)."""
        a, b = self.plan.owned_local(0)
        self.eng.sync()
        part = self.eng.u0[a:b].contiguous().view(torch.int64).sum().reshape(1)
        return int(self.comm.all_reduce_sum(part).item())

    def gather_solution(self):
        """level-0 solution on every rank (tests only)"""
        p = self.plan
        a, b = p.owned_local(0)
        blk = p.chunk * p.pitch(0)
        dev = self.eng.u0.device
        inp = torch.zeros(blk, dtype=torch.float64, device=dev)
        inp[:b - a].copy_(self.eng.u0[a:b])
        out = torch.zeros(blk * self.world, dtype=torch.float64, device=dev)
        self.comm.all_gather_blocks(inp, out)
        self.eng.sync()
        return out[:p.global_rows(0)].cpu().numpy().copy()

    def timed_out(self):
        return False

    def close(self):
        self.eng.close()


# ---------------------------------------------------------------- engine ---------
class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


class HipWindowEngine:
    """The C-ABI solver (libamg_hip.so) as the engine: amg_hip_create_poisson_window on this
    rank's GPU for the distributed levels, an ordinary solver for the replicated tail; u0, fk, uk
    are torch views of the window solver's memory."""

    def __init__(self, amg, device, stream, plan, omega, patch_min_rows=None, use_graph=True):
        self.amg, self.plan, self.device, self._stream = amg, plan, device, stream
        self.omega, self.use_graph = omega, use_graph
        self.tail = None
        sm = amg.SM_JACOBI if plan.smoother == SM_JACOBI else amg.SM_MULTICOLOR_GS
        if patch_min_rows is not None:
            amg.set_patch_min_rows(patch_min_rows)
        try:
            self.mg = amg.Multigrid.poisson_window(plan.n, plan.w0, plan.w1, plan.k + 1, dim=plan.dim,
                                                   smoother=sm, smoother_iters=plan.iters, omega=omega,
                                                   device=device.index, stream=stream.cuda_stream,
                                                   use_graph=use_graph)
        finally:
            if patch_min_rows is not None:
                amg.set_patch_min_rows(amg.PATCH_MIN_ROWS_DEFAULT)
        for l in range(plan.k + 1):
            if self.mg.get_n_dofs(l) != plan.window_rows(l):
                raise RuntimeError("window hierarchy does not have the planned level sizes")
        if plan.dim == 2:
            self.mg.window_setup(*plan.patch_ranges())
        else:
            self.mg.window_setup()
        self.u0 = self._view(0, "u")
        self.fk = self._view(plan.k, "f")
        self.uk = self._view(plan.k, "u")
        self._r0 = self._view(0, "r")
        self._scratch = torch.zeros(1100, dtype=torch.float64, device=device)

    def _view(self, level, which, mg=None):
        ptr, n = (mg or self.mg).vec_dev_ptr(level, which)
        return torch.as_tensor(_DevArray(ptr, n), device=self.device)

    def run(self, part):
        self.mg.window_run(part)

    def level_matrix(self, l):
        return self.mg.get_coefficient_matrix(l)

    def colors(self, l):
        return self.mg.get_colors(l)

    def build_tail(self, colptr, rowind, val, n_levels):
        amg = self.amg
        sm = amg.SM_JACOBI if self.plan.smoother == SM_JACOBI else amg.SM_MULTICOLOR_GS
        n = colptr.size - 1
        self.tail = amg.Multigrid(colptr, rowind, val, np.zeros(n), n_levels, smoother=sm,
                                  smoother_iters=self.plan.iters, omega=self.omega,
                                  device=self.device.index, stream=self._stream.cuda_stream,
                                  use_graph=self.use_graph)
        self._tail_f = self._view(0, "f", self.tail)
        self._tail_u = self._view(0, "u", self.tail)

    def tail_cycle(self, f_full):
        """one V-cycle of the replicated levels from a zero guess (multigrid.hpp:278)"""
        self._tail_f.copy_(f_full)
        self.tail.zero_vec(0, "u")
        self.tail.vcycle(1)
        return self._tail_u

    def residual_sumsq(self, a, b):
        self.mg.level_op(0, 1)                                # r_0 = f_0 - A_0 u_0 over the window
        out = self._scratch[1024:1025]
        lib = self.amg.lib()
        st = lib.amg_hip_dev_sumsq(b - a, self._r0[a:b].data_ptr(), out.data_ptr(), self._scratch.data_ptr(),
                                   self._stream.cuda_stream)
        if st != 0:
            raise self.amg.AmgHipError(st, lib.amg_hip_last_error().decode())
        return out.clone()

    def sync(self):
        self._stream.synchronize()

    def close(self):
        self.u0 = self.fk = self.uk = self._r0 = self._tail_f = self._tail_u = None
        if self.tail is not None:
            self.tail.close()
            self.tail = None
        if self.mg is not None:
            self.mg.close()
            self.mg = None


# ------------------------------------------------------- bench.py --gpus N -------
class ReplicatedVcycle:
    """Nothing distributed: every rank runs the whole single-GPU cycle (amg_hip_create_poisson).
    No exchange on the data path; it is the result every sharded candidate has to reproduce."""

    n_dist = 0

    def __init__(self, amg, device, stream, dim, n, L, smoother, iters, omega, use_graph=True):
        sm = amg.SM_JACOBI if smoother == SM_JACOBI else amg.SM_MULTICOLOR_GS
        self.device, self._stream = device, stream
        self.mg = amg.Multigrid.poisson(n, L, dim=dim, smoother=sm, smoother_iters=iters, omega=omega,
                                        device=device.index, stream=stream.cuda_stream, use_graph=use_graph)
        ptr, m = self.mg.vec_dev_ptr(0, "u")
        self._u0 = torch.as_tensor(_DevArray(ptr, m), device=device)

    def vcycle(self):
        self.mg.vcycle(1)

    def rss(self):
        return self.mg.rss()

    def solution_checksum(self):
        self._stream.synchronize()
        return int(self._u0.view(torch.int64).sum().item())

    def timed_out(self):
        return False

    def close(self):
        self._u0 = None
        self.mg.close()


def auto_levels(dim, n, world, L, smoother, iters, min_rows, max_overhead=0.35):
    """Distributed levels for bench.py: every level of at least `min_rows` rows, as long as the
    windows stay feasible and the redundantly recomputed halo stays below `max_overhead` of the
    owned block."""
    best = 0
    for k in range(1, L):
        try:
            p = WindowPlan(dim, n, 0, world, k, smoother, iters)
        except ValueError:
            break
        if p.global_rows(k - 1) < min_rows or 2.0 * p.halo / p.chunk > max_overhead:
            break
        best = k
    return best
