"""Slab-sharded V-cycle: the multi-GPU form of the K-Patch cycle.

Every rank builds the SAME solver (whole hierarchy, full-size vectors; 288 GB of HBM per GPU
make that cheap) and runs the K-Patch levels only over its own block of grid lines plus a halo
of `halo_lines` lines that it recomputes redundantly (include/amg_hip.h, "slab sharding").
One V-cycle then needs exactly two exchanges, both issued here through torch.distributed
("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests and the one-GPU rehearsal):

  1. halo: `halo_lines` lines of the level-0 solution from rank-1 and rank+1 (grouped
     isend/irecv, in place in the full-size vector; 22 lines = 720 KB per side at 4096^2
     with four slab levels);
  2. all-gather of the right-hand side of the first non-slab level (1 M rows = 8 MB at
     4096^2), after which the rest of the cycle (coarse solve included) runs replicated.

Compare dist_vcycle.DistributedVcycle: one exchange per sweep, residual and transfer (about
seven per level and cycle).  Per-row arithmetic is the single-GPU kernels', so the assembled
solution equals the single-GPU one bit for bit (tests/test_dist_gloo.py on an emulated
engine, tests/test_gpu_slab.py on the device).

All compute sits behind an `engine`:
  engine.info            amg_ctypes.SlabInfo (line ranges, pitches)
  engine.u0, engine.fg   1-D float64 tensors over the padded level-0 solution / gathered rhs
  engine.run(part)       1 = down-legs, 2 = replicated rest, 3 = up-legs (stream-ordered)
  engine.rss()           AMG::rss on the full level-0 vectors
  engine.sync()
"""
import numpy as np
import torch
import torch.distributed as dist


def exchange(ops, group, host_staged, sync):
    """Runs grouped isend/irecv ops; host_staged: through host copies (a process group that
    cannot move device tensors, e.g. gloo next to a GPU engine)."""
    if not ops:
        return
    if host_staged:
        sync()
        staged, back = [], []
        for op in ops:
            h = op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            staged.append(dist.P2POp(op.op, h, op.peer, group))
            if op.op is dist.irecv:
                back.append((op.tensor, h))
        for req in dist.batch_isend_irecv(staged):
            req.wait()
        for dst, h in back:
            dst.copy_(h)
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


class SlabVcycle:
    """multigrid.hpp:263-305 over row-block shards with redundant halos (true Jacobi 2+2)."""

    def __init__(self, engine, rank, world, group=None, host_staged=False):
        self.eng, self.rank, self.world, self.group = engine, rank, world, group
        self.host_staged = bool(host_staged)
        i = engine.info
        self.info = i
        self.n_dist = int(i.levels) if world > 1 else 0
        self.n0 = int(i.lines) * int(i.pitch0)
        self._cycles_run = 0
        self._inplace_ok = True
        if world > 1 and (i.line_end - i.line_begin < i.halo_lines):
            raise ValueError("slab sharding: fewer owned lines than the halo depth")

    # ---- exchange 1: halo lines of the level-0 solution, in place ----
    def _exchange_u0(self):
        i, r, w, u = self.info, self.rank, self.world, self.eng.u0
        m, H = int(i.pitch0), int(i.halo_lines) * int(i.pitch0)
        a, b = int(i.line_begin) * m, int(i.line_end) * m
        ops = []
        if r > 0:
            ops.append(dist.P2POp(dist.irecv, u[a - H:a], r - 1, self.group))
            ops.append(dist.P2POp(dist.isend, u[a:a + H], r - 1, self.group))
        if r < w - 1:
            ops.append(dist.P2POp(dist.irecv, u[b:b + H], r + 1, self.group))
            ops.append(dist.P2POp(dist.isend, u[b - H:b], r + 1, self.group))
        exchange(ops, self.group, self.host_staged, self.eng.sync)

    # ---- exchange 2 (and the diagnostics): equal blocks of a padded vector, in place ----
    def _all_gather_blocks(self, vec, block):
        r = self.rank
        mine = vec[r * block:(r + 1) * block]
        if self.host_staged:
            self.eng.sync()
            ho = torch.empty(block * self.world, dtype=vec.dtype)
            dist.all_gather_into_tensor(ho, mine.cpu(), group=self.group)
            vec[:block * self.world].copy_(ho)
        elif vec.device.type == "cpu":   # gloo: no aliasing of input and output
            dist.all_gather_into_tensor(vec[:block * self.world], mine.clone(), group=self.group)
        elif self._inplace_ok:           # RCCL in-place all-gather (input = its own block of the output)
            try:
                dist.all_gather_into_tensor(vec[:block * self.world], mine, group=self.group)
            except (RuntimeError, TypeError, ValueError):
                # refused at argument checking (before anything is enqueued, hence on every rank
                # alike): gather from a copy of the block instead
                self._inplace_ok = False
                dist.all_gather_into_tensor(vec[:block * self.world], mine.clone(), group=self.group)
        else:
            dist.all_gather_into_tensor(vec[:block * self.world], mine.clone(), group=self.group)

    def use_library_comm(self, libcomm):
        """one V-cycle = ONE C call (amg_hip_slab_cycle): the two exchanges are the library's own
        RCCL calls on the solver's stream (amg_ctypes.Comm)"""
        self.libcomm = libcomm

    def vcycle(self):
        if getattr(self, "libcomm", None) is not None:
            self.libcomm.slab_cycle(self.eng.mg, self.eng.info)
            self._cycles_run += 1
            return
        if self.world == 1:
            self.eng.run(1)
            self.eng.run(2)
            self.eng.run(3)
        else:
            i = self.info
            self._exchange_u0()
            self.eng.run(1)
            self._all_gather_blocks(self.eng.fg, int(i.chunk_lines) * int(i.gather_pitch))
            self.eng.run(2)
            self.eng.run(3)
        self._cycles_run += 1

    # ---- diagnostics: assemble the level-0 solution on every rank ----
    def _assemble(self):
        if self.world > 1:
            i = self.info
            self._all_gather_blocks(self.eng.u0, int(i.chunk_lines) * int(i.pitch0))

    def rss(self):
        """AMG::rss(A_0, u_0, b) (common.hpp:17-27) of the assembled solution: the single-GPU
        reduction on every rank, hence the same bits as at N = 1."""
        self._assemble()
        return self.eng.rss()

    def solution_checksum(self):
        """Sum of the 64-bit patterns of the level-0 solution modulo 2^64
        (dist_vcycle.DistributedVcycle.solution_checksum: partition-independent)."""
        self._assemble()
        self.eng.sync()
        return int(self.eng.u0[:self.n0].view(torch.int64).sum().item())

    def gather_solution(self):
        self._assemble()
        self.eng.sync()
        return self.eng.u0[:self.n0].cpu().numpy().copy()

    def timed_out(self):
        return False

    def close(self):
        self.eng.close()


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


class HipSlabEngine:
    """The C-ABI solver (libamg_hip.so) as the engine: amg_hip_create_poisson on this rank's
    GPU, amg_hip_slab_setup / amg_hip_slab_run; u0 and fg are torch views of its memory."""

    def __init__(self, amg, device, stream, n, n_levels, omega, sweeps, rank, world, max_levels=-1,
                 patch_min_rows=None):
        """patch_min_rows: K-Patch threshold for THIS solver (amg_hip_set_patch_min_rows is read
        when a solver is created; the default is restored afterwards).  A sharded level is
        1/world of the work per rank, so smaller levels pay off as slab levels than as K-Patch
        levels of a single GPU."""
        self.amg = amg
        if patch_min_rows is not None:
            amg.set_patch_min_rows(patch_min_rows)
        try:
            self.mg = amg.Multigrid.poisson(n, n_levels, smoother=amg.SM_JACOBI, smoother_iters=sweeps,
                                            omega=omega, device=device.index, stream=stream.cuda_stream)
        finally:
            if patch_min_rows is not None:
                amg.set_patch_min_rows(amg.PATCH_MIN_ROWS_DEFAULT)
        try:
            self.info = self.mg.slab_setup(rank, world, max_levels)
        except Exception:
            self.mg.close()
            raise
        i = self.info
        self._stream = stream
        cap = int(i.chunk_lines) * world
        self.u0 = torch.as_tensor(_DevArray(i.u0, cap * int(i.pitch0)), device=device)
        self.fg = torch.as_tensor(_DevArray(i.f_gather, cap * int(i.gather_pitch)), device=device)

    def run(self, part):
        self.mg.slab_run(part)

    def rss(self):
        return self.mg.rss()

    def sync(self):
        self._stream.synchronize()

    def close(self):
        self.u0 = self.fg = None
        self.mg.close()
