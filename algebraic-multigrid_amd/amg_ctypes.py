"""ctypes binding of libamg_hip.so (C ABI: include/amg_hip.h).

Thin host-side mirror used by tests/, bench.py and the multi-GPU driver.  It
contains no arithmetic: every operation is a call through the C ABI into the
HIP kernels.  If the shared library is missing the import fails loudly; if no
HIP device is present every compute call raises AmgHipError (status EHIP) --
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libamg_hip.so")

OK, EINVAL, EHIP, ENOMEM, EUNSUPPORTED, ECOMM = 0, 1, 2, 3, 4, 5
SM_SPGS, SM_REF_JACOBI, SM_SOR, SM_JACOBI, SM_MULTICOLOR_GS = 0, 1, 2, 3, 4
LAYOUT_AUTO, LAYOUT_CSR, LAYOUT_SELL, LAYOUT_DICT = 0, 1, 2, 3

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)


class AmgHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"amg_hip status {status}: {msg}")
        self.status = status
        self.message = msg


class Options(C.Structure):
    _fields_ = [("smoother", C.c_int32), ("smoother_iters", C.c_int32),
                ("omega", C.c_double), ("device", C.c_int32), ("use_graph", C.c_int32),
                ("stencil_transfers", C.c_int32), ("layout", C.c_int32),
                ("host_only", C.c_int32), ("keep_structural_zeros", C.c_int32),
                ("no_fusion", C.c_int32), ("fuse_prolong", C.c_int32),
                ("fast_coarse_solve", C.c_int32), ("host_galerkin", C.c_int32),
                ("keep_residual", C.c_int32), ("exact_coarse_solve", C.c_int32),
                ("exact_gs", C.c_int32),
                ("stream", C.c_void_p), ("window", C.c_int32), ("reserved0", C.c_int32)]


SLAB_MAX_LEVELS = 8


class SlabInfo(C.Structure):
    """amg_hip_slab_info (include/amg_hip.h)"""
    _fields_ = [("levels", C.c_int32), ("halo_lines", C.c_int32), ("lines", C.c_int64),
                ("chunk_lines", C.c_int64), ("line_begin", C.c_int64), ("line_end", C.c_int64),
                ("pitch0", C.c_int64), ("gather_pitch", C.c_int64), ("gather_rows", C.c_int64),
                ("down_lo", C.c_int64 * SLAB_MAX_LEVELS), ("down_hi", C.c_int64 * SLAB_MAX_LEVELS),
                ("up_lo", C.c_int64 * SLAB_MAX_LEVELS), ("up_hi", C.c_int64 * SLAB_MAX_LEVELS),
                ("u0", C.c_void_p), ("f_gather", C.c_void_p)]


class WindowPlanC(C.Structure):
    """amg_hip_window_plan (include/amg_hip.h)"""
    _fields_ = [(k, C.c_int64) for k in ("own0_off", "own0_end", "send_prev_cnt", "recv_prev_cnt",
                                         "send_next_cnt", "recv_next_cnt", "own_k_off", "own_k_cnt",
                                         "block_k", "uk_off")]


class HaloDesc(C.Structure):
    _fields_ = [("dst_prev", C.c_void_p), ("src_prev", C.c_void_p), ("bytes_prev", C.c_int64),
                ("dst_next", C.c_void_p), ("src_next", C.c_void_p), ("bytes_next", C.c_int64),
                ("data_flag_at_prev", C.c_void_p), ("data_flag_at_next", C.c_void_p),
                ("my_data_from_prev", C.c_void_p), ("my_data_from_next", C.c_void_p),
                ("ack_flag_at_prev", C.c_void_p), ("ack_flag_at_next", C.c_void_p),
                ("my_ack_from_prev", C.c_void_p), ("my_ack_from_next", C.c_void_p),
                ("epoch", C.c_uint32), ("recv_from_prev", C.c_int32), ("recv_from_next", C.c_int32)]


class HaloKDesc(C.Structure):
    _fields_ = [("dst_prev", C.c_void_p), ("src_prev", C.c_void_p), ("cnt_prev", C.c_int64),
                ("dst_next", C.c_void_p), ("src_next", C.c_void_p), ("cnt_next", C.c_int64),
                ("my_free_from_prev", C.c_void_p), ("my_free_from_next", C.c_void_p),
                ("data_at_prev", C.c_void_p), ("data_at_next", C.c_void_p),
                ("my_data_from_prev", C.c_void_p), ("my_data_from_next", C.c_void_p),
                ("recv_prev", C.c_int32), ("recv_next", C.c_int32), ("timeout", C.c_void_p)]


class GatherKDesc(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("src", C.c_void_p),
                ("cnt", C.c_int64), ("off", C.c_int64), ("dst", C.c_void_p * 16),
                ("data_at", C.c_void_p * 16), ("free_at", C.c_void_p * 16),
                ("my_data_from", C.c_void_p), ("my_free_from", C.c_void_p),
                ("timeout", C.c_void_p)]


# name -> (restype, argtypes).  Must list EVERY symbol include/amg_hip.h declares
# (tests/test_cabi_symbols.py checks the header against this table).
_SIGS = {
    "amg_hip_last_error": (C.c_char_p, []),
    "amg_hip_default_options": (None, [C.POINTER(Options)]),
    "amg_hip_device_count": (C.c_int, []),
    "amg_hip_set_default_layout": (None, [C.c_int32]),
    "amg_hip_create": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_int32,
                                 C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "amg_hip_create_custom": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_int32,
                                        C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_f64p),
                                        C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_f64p),
                                        C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "amg_hip_create_poisson": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.POINTER(Options),
                                         C.POINTER(C.c_void_p)]),
    "amg_hip_destroy": (None, [C.c_void_p]),
    "amg_hip_vcycle": (C.c_int, [C.c_void_p]),
    "amg_hip_vcycles": (C.c_int, [C.c_void_p, C.c_int32]),
    "amg_hip_sync": (C.c_int, [C.c_void_p]),
    "amg_hip_solve": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_int64, _i64p, _f64p,
                                _i32p]),
    "amg_hip_rss": (C.c_int, [C.c_void_p, _f64p]),
    "amg_hip_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "amg_hip_pcg": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, _i64p, _f64p]),
    "amg_hip_n_levels": (C.c_int32, [C.c_void_p]),
    "amg_hip_get_n_dofs": (C.c_int64, [C.c_void_p, C.c_int32]),
    "amg_hip_get_level_nnz": (C.c_int64, [C.c_void_p, C.c_int32]),
    "amg_hip_get_level_matrix": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, _f64p]),
    "amg_hip_get_transfer_nnz": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "amg_hip_get_transfer": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i32p, _f64p]),
    "amg_hip_get_vec": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f64p]),
    "amg_hip_set_vec": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _f64p]),
    "amg_hip_copy_vec_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "amg_hip_zero_vec": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "amg_hip_coarse_halfbw": (C.c_int64, [C.c_void_p]),
    "amg_hip_coarse_solve_kind": (C.c_int32, [C.c_void_p]),
    "amg_hip_fine_sweep_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, _i32p, _f64p]),
    "amg_hip_set_patch_min_rows": (None, [C.c_int64]),
    "amg_hip_set_band_chain": (None, [C.c_int32]),
    "amg_hip_set_tail_fusion": (None, [C.c_int32]),
    "amg_hip_set_patch_tile_flags": (None, [C.c_int32]),
    "amg_hip_create_rs": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_int32, C.c_double, C.c_int64,
                                    C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "amg_hip_slab_plan": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(SlabInfo)]),
    "amg_hip_slab_setup": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(SlabInfo)]),
    "amg_hip_slab_run": (C.c_int, [C.c_void_p, C.c_int32]),
    "amg_hip_create_poisson_window": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                                C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "amg_hip_window_setup": (C.c_int, [C.c_void_p, _i64p, _i64p, _i64p, _i64p]),
    "amg_hip_window_run": (C.c_int, [C.c_void_p, C.c_int32]),
    "amg_hip_vec_dev_ptr": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), _i64p]),
    "amg_hip_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "amg_hip_comm_unique_id": (C.c_int, [C.c_char_p]),
    "amg_hip_comm_create": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "amg_hip_comm_destroy": (None, [C.c_void_p]),
    "amg_hip_comm_rank": (C.c_int32, [C.c_void_p]),
    "amg_hip_comm_world": (C.c_int32, [C.c_void_p]),
    "amg_hip_comm_neighbor_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                 C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "amg_hip_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "amg_hip_slab_cycle": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SlabInfo)]),
    "amg_hip_window_cycle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(WindowPlanC)]),
    "amg_hip_level_layout": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "amg_hip_get_colors": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p]),
    "amg_hip_level_op": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "amg_hip_cycle_bytes": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "amg_hip_cycle_must_move": (C.c_int, [C.c_void_p, C.c_int32, _f64p]),
    "amg_hip_profile_fine_sweep": (C.c_int, [C.c_void_p, C.c_int32, _f64p, _f64p]),
    "amg_hip_smooth": (C.c_int, [C.c_int32, C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p,
                                 C.c_double, C.c_double, C.c_int64, C.c_int64, _i64p, _i32p]),
    "amg_hip_spgs_sweep": (C.c_int, [C.c_int32, C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]),
    "amg_hip_residual": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p]),
    "amg_hip_spmv": (C.c_int, [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p]),
    "amg_hip_linear_restrict": (C.c_int, [C.c_int64, C.c_int64, _f64p, _f64p]),
    "amg_hip_linear_prolong_add": (C.c_int, [C.c_int64, C.c_int64, _f64p, _f64p]),
    "amg_hip_rss_host": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p]),
    "amg_hip_coarse_solve": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, _i64p]),
    "amg_hip_coarse_solve_fast": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, _i64p,
                                            _i32p]),
    "amg_hip_laplacian": (C.c_int64, [C.c_int32, C.c_int64, _i32p, _i32p, _f64p]),
    "amg_hip_rhs": (C.c_int, [C.c_int32, C.c_int64, _f64p]),
    "amg_hip_csr_shape": (C.c_int, [C.c_int64, _i32p, _i32p, _i32p]),
    "amg_hip_dev_residual": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "amg_hip_dev_jacobi": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_double, C.c_int64, C.c_void_p]),
    "amg_hip_dev_spmv": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "amg_hip_devmat_create": (C.c_int, [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.c_int32,
                                        C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "amg_hip_set_index16": (None, [C.c_int32]),
    "amg_hip_set_nontemporal": (None, [C.c_int32]),
    "amg_hip_set_dict_rows": (None, [C.c_int32]),
    "amg_hip_set_dict_stencil": (None, [C.c_int32]),
    "amg_hip_set_xcd_mapping": (None, [C.c_int32]),
    "amg_hip_set_row_types": (None, [C.c_int32]),
    "amg_hip_dict_probe": (C.c_int, [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.c_int64,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "amg_hip_devmat_destroy": (None, [C.c_void_p]),
    "amg_hip_devmat_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "amg_hip_devmat_apply": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_double, C.c_int64, C.c_void_p]),
    "amg_hip_dev_jacobi_from_zero": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_double, C.c_void_p]),
    "amg_hip_dev_axpy1": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "amg_hip_arena_create": (C.c_int, [C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    "amg_hip_arena_destroy": (None, [C.c_void_p]),
    "amg_hip_arena_base": (C.c_void_p, [C.c_void_p]),
    "amg_hip_arena_export": (C.c_int, [C.c_void_p, C.c_char_p]),
    "amg_hip_arena_open_peer": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "amg_hip_arena_close_peer": (C.c_int, [C.c_void_p]),
    "amg_hip_halo_push_wait": (C.c_int, [C.POINTER(HaloDesc), C.c_void_p]),
    "amg_hip_halo_ack": (C.c_int, [C.POINTER(HaloDesc), C.c_void_p]),
    "amg_hip_halo_exchange_kernel": (C.c_int, [C.POINTER(HaloKDesc), C.c_void_p]),
    "amg_hip_halo_ack_kernel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "amg_hip_gather_kernel": (C.c_int, [C.POINTER(GatherKDesc), C.c_void_p]),
    "amg_hip_gather_ack_kernel": (C.c_int, [C.POINTER(GatherKDesc), C.c_void_p]),
    "amg_hip_fill_u32": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p]),
    "amg_hip_capture_begin": (C.c_int, [C.c_void_p]),
    "amg_hip_capture_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "amg_hip_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "amg_hip_graph_destroy": (None, [C.c_void_p]),
    "amg_hip_dev_sumsq": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("AMG_HIP_LIBRARY", LIB_PATH)   # another build of the same ABI
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `make -C {_HERE}` "
                "(or __graft_entry__.build()); there is no fallback implementation")
        L = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _chk(status):
    if status != OK:
        raise AmgHipError(status, lib().amg_hip_last_error().decode())


def _a32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _a64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p32(a):
    return a.ctypes.data_as(_i32p)


def _p64(a):
    return a.ctypes.data_as(_f64p)


def device_count():
    return lib().amg_hip_device_count()


def set_default_layout(layout):
    lib().amg_hip_set_default_layout(layout)


def set_index16(on):
    lib().amg_hip_set_index16(int(on))


def set_dict_rows(rows_per_lane):
    lib().amg_hip_set_dict_rows(int(rows_per_lane))


def set_dict_stencil(on):
    lib().amg_hip_set_dict_stencil(int(on))


def dict_probe(rowptr, col, val, ncols, diag_shift=0):
    """Host-only: (n_pairs, n_row_types, words) of the dictionary coding of a CSR block, or
    None when the block does not qualify.  Raises when the coding does not round-trip."""
    rowptr, col, val = _a32(rowptr), _a32(col), _a64(val)
    a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    st = lib().amg_hip_dict_probe(rowptr.size - 1, ncols, _p32(rowptr), _p32(col), _p64(val),
                                  diag_shift, C.byref(a), C.byref(b), C.byref(c))
    if st == EUNSUPPORTED:
        return None
    _chk(st)
    return a.value, b.value, c.value


class Comm:
    """amg_hip_comm: the library's own RCCL communicator (include/amg_hip.h "communicator").
    id_bytes: the 128 bytes rank 0 got from Comm.unique_id(), handed to every rank by the caller."""

    def __init__(self, id_bytes, rank, world, device=-1):
        h = C.c_void_p()
        _chk(lib().amg_hip_comm_create(bytes(id_bytes), int(rank), int(world), int(device), C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _chk(lib().amg_hip_comm_unique_id(buf))
        return buf.raw

    def slab_cycle(self, mg, info):
        _chk(lib().amg_hip_slab_cycle(mg._h, self._h, C.byref(info)))

    def window_cycle(self, window_mg, tail_mg, plan_c):
        _chk(lib().amg_hip_window_cycle(window_mg._h, tail_mg._h, self._h, C.byref(plan_c)))

    def close(self):
        if getattr(self, "_h", None):
            lib().amg_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def slab_plan(lines, rank, world, levels):
    """Host arithmetic of the slab sharding (no device): SlabInfo with the line ranges."""
    info = SlabInfo()
    _chk(lib().amg_hip_slab_plan(int(lines), int(rank), int(world), int(levels), C.byref(info)))
    return info


def set_patch_tile_flags(on):
    lib().amg_hip_set_patch_tile_flags(int(bool(on)))


def set_tail_fusion(on):
    lib().amg_hip_set_tail_fusion(int(bool(on)))


def set_band_chain(on):
    lib().amg_hip_set_band_chain(int(bool(on)))


PATCH_MIN_ROWS_DEFAULT = 1000000   # the library's default K-Patch threshold (amg_hip_set_patch_min_rows)


def set_patch_min_rows(rows):
    lib().amg_hip_set_patch_min_rows(int(rows))


def set_row_types(on):
    lib().amg_hip_set_row_types(int(on))


def set_xcd_mapping(on):
    lib().amg_hip_set_xcd_mapping(int(on))


# ---- Grid<double> ------------------------------------------------------------
def laplacian(n, dim=2):
    """grid.hpp:88-98.  Returns (colptr, rowind, val) of the CSC matrix."""
    nnz = lib().amg_hip_laplacian(dim, n, None, None, None)
    if nnz < 0:
        raise AmgHipError(EINVAL, lib().amg_hip_last_error().decode())
    N = n ** dim
    colptr = np.empty(N + 1, np.int32)
    rowind = np.empty(nnz, np.int32)
    val = np.empty(nnz, np.float64)
    got = lib().amg_hip_laplacian(dim, n, _p32(colptr), _p32(rowind), _p64(val))
    assert got == nnz
    return colptr, rowind, val


def rhs(n, dim=2):
    b = np.empty(n ** dim, np.float64)
    _chk(lib().amg_hip_rhs(dim, n, _p64(b)))
    return b


# ---- AMG::Multigrid<double> ----------------------------------------------------
class Multigrid:
    """Mirror of AMG::Multigrid<double> (multigrid.hpp) over the C ABI."""

    def __init__(self, colptr, rowind, val, b, n_levels, smoother=SM_SPGS,
                 smoother_iters=1, omega=1.0, tolerance=1e-9, compute_error_every_n_iters=10,
                 n_iters=100, device=-1, use_graph=True, stencil_transfers=True,
                 transfers=None, layout=None, host_only=False, keep_structural_zeros=False,
                 no_fusion=False, fuse_prolong=False, stream=None, fast_coarse_solve=False,
                 host_galerkin=False, keep_residual=False, exact_coarse_solve=False,
                 exact_gs=False):
        # multigrid.hpp:165-178 (same checks, same order)
        if compute_error_every_n_iters > n_iters:
            raise ValueError("`compute_error_every_n_iters` must be leq to `n_iters`, got "
                             f"{compute_error_every_n_iters} and {n_iters}")
        colptr, rowind, val, b = _a32(colptr), _a32(rowind), _a64(val), _a64(b)
        n = colptr.size - 1
        if n != b.size:
            raise ValueError("`A` and `b` must have the same number of degrees of freedom, "
                             f"got {n} and {b.size}")
        self.tolerance = tolerance
        self.every = compute_error_every_n_iters
        self.n_iters = n_iters
        o = self._options(smoother, smoother_iters, omega, device, use_graph, stencil_transfers, layout,
                          host_only, keep_structural_zeros, no_fusion, fuse_prolong, stream,
                          fast_coarse_solve, host_galerkin, keep_residual, exact_coarse_solve, exact_gs)
        h = C.c_void_p()
        if transfers is None:
            st = lib().amg_hip_create(n, _p32(colptr), _p32(rowind), _p64(val), _p64(b),
                                      n_levels, C.byref(o), C.byref(h))
        else:
            # transfers: list over levels of (P_csc, R_csc), each (colptr,rowind,val)
            keep = []
            arrs = [[], [], [], [], [], []]
            for (P, R) in transfers:
                for k, a in enumerate((_a32(P[0]), _a32(P[1]), _a64(P[2]),
                                       _a32(R[0]), _a32(R[1]), _a64(R[2]))):
                    keep.append(a)
                    arrs[k].append(a)
            nl = len(transfers)

            def tab(lst, ptr_t, conv):
                t = (ptr_t * max(nl, 1))()
                for i, a in enumerate(lst):
                    t[i] = conv(a)
                return t
            st = lib().amg_hip_create_custom(
                n, _p32(colptr), _p32(rowind), _p64(val), _p64(b), n_levels,
                tab(arrs[0], _i32p, _p32), tab(arrs[1], _i32p, _p32), tab(arrs[2], _f64p, _p64),
                tab(arrs[3], _i32p, _p32), tab(arrs[4], _i32p, _p32), tab(arrs[5], _f64p, _p64),
                C.byref(o), C.byref(h))
        if st == EINVAL:
            raise ValueError(lib().amg_hip_last_error().decode())
        _chk(st)
        self._h = h

    @staticmethod
    def _options(smoother, smoother_iters, omega, device, use_graph, stencil_transfers, layout,
                 host_only, keep_structural_zeros, no_fusion, fuse_prolong, stream,
                 fast_coarse_solve, host_galerkin, keep_residual, exact_coarse_solve, exact_gs):
        o = Options()
        lib().amg_hip_default_options(C.byref(o))
        o.smoother, o.smoother_iters, o.omega = smoother, smoother_iters, omega
        o.device, o.use_graph, o.stencil_transfers = device, int(use_graph), int(stencil_transfers)
        if layout is not None:
            o.layout = layout
        o.host_only = int(host_only)
        o.keep_structural_zeros = int(keep_structural_zeros)
        o.no_fusion = int(no_fusion)
        o.fuse_prolong = int(fuse_prolong)
        o.fast_coarse_solve = int(fast_coarse_solve)
        o.host_galerkin = int(host_galerkin)
        o.keep_residual = int(keep_residual)
        o.exact_coarse_solve = int(exact_coarse_solve)
        o.exact_gs = int(exact_gs)
        if stream:
            o.stream = C.c_void_p(stream)
        return o

    @classmethod
    def ruge_stueben(cls, colptr, rowind, val, b, max_levels=25, theta=0.25, min_coarse=500,
                     smoother=SM_SPGS, smoother_iters=1, omega=1.0, tolerance=1e-9,
                     compute_error_every_n_iters=10, n_iters=100, device=-1, use_graph=True, layout=None,
                     host_only=False, stream=None, host_galerkin=False, exact_coarse_solve=False,
                     exact_gs=False):
        """AMG::Multigrid on a strength-based C/F hierarchy (amg_hip_create_rs): same V-cycle,
        coarsening by the classical Ruge-Stueben first pass with direct interpolation.
        `self.n_levels` tells how many levels were built."""
        colptr, rowind, val, b = _a32(colptr), _a32(rowind), _a64(val), _a64(b)
        n = colptr.size - 1
        if n != b.size:
            raise ValueError("`A` and `b` must have the same number of degrees of freedom, "
                             f"got {n} and {b.size}")
        self = cls.__new__(cls)
        self.tolerance, self.every, self.n_iters = tolerance, compute_error_every_n_iters, n_iters
        o = cls._options(smoother, smoother_iters, omega, device, use_graph, True, layout, host_only,
                         False, False, False, stream, False, host_galerkin, False, exact_coarse_solve,
                         exact_gs)
        h = C.c_void_p()
        st = lib().amg_hip_create_rs(n, _p32(colptr), _p32(rowind), _p64(val), _p64(b), int(max_levels),
                                     float(theta), int(min_coarse), C.byref(o), C.byref(h))
        if st == EINVAL:
            raise ValueError(lib().amg_hip_last_error().decode())
        _chk(st)
        self._h = h
        return self

    @classmethod
    def poisson(cls, n, n_levels, dim=2, smoother=SM_SPGS, smoother_iters=1, omega=1.0, tolerance=1e-9,
                compute_error_every_n_iters=10, n_iters=100, device=-1, use_graph=True,
                stencil_transfers=True, layout=None, keep_structural_zeros=False, no_fusion=False,
                stream=None, fast_coarse_solve=False, keep_residual=False, exact_coarse_solve=False,
                exact_gs=False):
        """AMG::Multigrid on A = Grid::laplacian(n), b = Grid::rhs(n) with the setup on the device
        end to end (amg_hip_create_poisson): no host matrices."""
        if compute_error_every_n_iters > n_iters:
            raise ValueError("`compute_error_every_n_iters` must be leq to `n_iters`, got "
                             f"{compute_error_every_n_iters} and {n_iters}")
        self = cls.__new__(cls)
        self.tolerance, self.every, self.n_iters = tolerance, compute_error_every_n_iters, n_iters
        o = cls._options(smoother, smoother_iters, omega, device, use_graph, stencil_transfers, layout,
                         False, keep_structural_zeros, no_fusion, False, stream, fast_coarse_solve,
                         False, keep_residual, exact_coarse_solve, exact_gs)
        h = C.c_void_p()
        st = lib().amg_hip_create_poisson(dim, n, n_levels, C.byref(o), C.byref(h))
        if st == EINVAL:
            raise ValueError(lib().amg_hip_last_error().decode())
        _chk(st)
        self._h = h
        return self

    @classmethod
    def poisson_window(cls, n, unit_begin, unit_end, n_levels, dim=2, smoother=SM_JACOBI, smoother_iters=2,
                       omega=0.6, device=-1, use_graph=True, layout=None, no_fusion=False, stream=None,
                       host_only=False):
        """One rank's WINDOW of a sharded Grid::laplacian(n) hierarchy (amg_hip_create_poisson_window):
        units [unit_begin, unit_end) of the slowest axis, n_levels - 1 distributed levels; runs by
        parts (window_run)."""
        self = cls.__new__(cls)
        self.tolerance, self.every, self.n_iters = 1e-9, 10, 100
        o = cls._options(smoother, smoother_iters, omega, device, use_graph, True, layout, host_only, False,
                         no_fusion, False, stream, False, False, False, False, False)
        o.window = 1
        h = C.c_void_p()
        st = lib().amg_hip_create_poisson_window(dim, n, int(unit_begin), int(unit_end), n_levels,
                                                 C.byref(o), C.byref(h))
        if st == EINVAL:
            raise ValueError(lib().amg_hip_last_error().decode())
        _chk(st)
        self._h = h
        return self

    def window_setup(self, down_lo=None, down_hi=None, up_lo=None, up_hi=None):
        if down_lo is None:
            _chk(lib().amg_hip_window_setup(self._h, None, None, None, None))
            return
        arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (down_lo, down_hi, up_lo, up_hi)]
        _chk(lib().amg_hip_window_setup(self._h, *[a.ctypes.data_as(_i64p) for a in arrs]))

    def window_run(self, part):
        _chk(lib().amg_hip_window_run(self._h, int(part)))

    def vec_dev_ptr(self, level, which):
        """(device pointer, n) of a level vector; which: "u", "f" or "r"."""
        p, n = C.c_void_p(), C.c_int64(0)
        _chk(lib().amg_hip_vec_dev_ptr(self._h, level, {"u": 0, "f": 1, "r": 2}[which], C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().amg_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def n_levels(self):
        return lib().amg_hip_n_levels(self._h)

    def get_n_dofs(self, level):
        return int(lib().amg_hip_get_n_dofs(self._h, level))

    def get_coefficient_matrix(self, level):
        n = self.get_n_dofs(level)
        nnz = lib().amg_hip_get_level_nnz(self._h, level)
        colptr = np.empty(n + 1, np.int32)
        rowind = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.float64)
        _chk(lib().amg_hip_get_level_matrix(self._h, level, _p32(colptr), _p32(rowind), _p64(val)))
        return colptr, rowind, val

    def get_transfer(self, level, which):
        w = 1 if which == "R" else 0
        nnz = lib().amg_hip_get_transfer_nnz(self._h, level, w)
        if nnz < 0:
            raise ValueError("level out of range")
        n_h, n_H = self.get_n_dofs(level), self.get_n_dofs(level + 1)
        cols = n_h if w else n_H
        colptr = np.empty(cols + 1, np.int32)
        rowind = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.float64)
        _chk(lib().amg_hip_get_transfer(self._h, level, w, _p32(colptr), _p32(rowind), _p64(val)))
        return colptr, rowind, val

    def _get(self, level, which):
        out = np.empty(self.get_n_dofs(level), np.float64)
        _chk(lib().amg_hip_get_vec(self._h, level, which, _p64(out)))
        return out

    def get_soln(self, level=0):
        return self._get(level, 0)

    def get_rhs(self, level=0):
        return self._get(level, 1)

    def get_residual(self, level=0):
        return self._get(level, 2)

    def set_vec(self, level, which, v):
        v = _a64(v)
        assert v.size == self.get_n_dofs(level)
        _chk(lib().amg_hip_set_vec(self._h, level, {"u": 0, "f": 1, "r": 2}[which], _p64(v)))

    def copy_vec_dev(self, level, which, dev_ptr, to_solver):
        _chk(lib().amg_hip_copy_vec_dev(self._h, level, {"u": 0, "f": 1, "r": 2}[which],
                                        C.c_void_p(dev_ptr), int(to_solver)))

    def zero_vec(self, level, which):
        _chk(lib().amg_hip_zero_vec(self._h, level, {"u": 0, "f": 1, "r": 2}[which]))

    def get_tolerance(self):
        return self.tolerance

    def level_layout(self, level):
        """(layout, matrix stream bytes per sweep) of the level's device operator."""
        lay, nb = C.c_int32(0), C.c_int64(0)
        _chk(lib().amg_hip_level_layout(self._h, level, C.byref(lay), C.byref(nb)))
        return int(lay.value), int(nb.value)

    def coarse_halfbw(self):
        return int(lib().amg_hip_coarse_halfbw(self._h))

    def get_colors(self, level):
        color = np.empty(self.get_n_dofs(level), np.int32)
        nc = C.c_int32(0)
        _chk(lib().amg_hip_get_colors(self._h, level, _p32(color), C.byref(nc)))
        return color, nc.value

    def cycle_bytes(self):
        a, b = C.c_double(0), C.c_double(0)
        _chk(lib().amg_hip_cycle_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def cycle_must_move(self, part=0):
        """bytes the launches of one V-cycle (or of one part of a sharded one) have to move"""
        b = C.c_double(0)
        _chk(lib().amg_hip_cycle_must_move(self._h, int(part), C.byref(b)))
        return b.value

    def profile_fine_sweep(self, n_launches):
        """(avg_ms, min_ms, sweeps per launch, kernel name) of the level-0 Jacobi sweep
        kernel, HIP events on the solver's stream."""
        a, b = C.c_double(0), C.c_double(0)
        _chk(lib().amg_hip_profile_fine_sweep(self._h, n_launches, C.byref(a), C.byref(b)))
        name = C.create_string_buffer(256)
        k, nb = C.c_int32(0), C.c_double(0)
        _chk(lib().amg_hip_fine_sweep_info(self._h, name, 256, C.byref(k), C.byref(nb)))
        return a.value, b.value, k.value, name.value.decode(), nb.value

    def fine_sweep_info(self):
        """(kernel name, sweeps per launch, bytes one launch has to move) of the level-0 smoother
        launch (amg_hip_fine_sweep_info); no launch is made."""
        name = C.create_string_buffer(256)
        k, nb = C.c_int32(0), C.c_double(0)
        _chk(lib().amg_hip_fine_sweep_info(self._h, name, 256, C.byref(k), C.byref(nb)))
        return name.value.decode(), k.value, nb.value

    def coarse_solve_kind(self):
        return {0: "band (one wave, sequential, bit-exact)", 1: "spike (partitioned, parallel)",
                2: "band-wide (blocked sequential, any half-bandwidth, bit-exact)",
                3: "band-chain (LDS-resident scalar recurrence, half-bandwidth <= 3, bit-exact)"}[
                    int(lib().amg_hip_coarse_solve_kind(self._h))]

    def level_op(self, level, op):
        """op: 0 smooth (built-in), 1 residual, 2 zero+restrict, 3 prolong+add, 4 coarse solve."""
        _chk(lib().amg_hip_level_op(self._h, level, op))

    def vcycle(self, n=1):
        _chk(lib().amg_hip_vcycles(self._h, n))

    def sync(self):
        _chk(lib().amg_hip_sync(self._h))

    def rss(self):
        out = C.c_double(0)
        _chk(lib().amg_hip_rss(self._h, C.byref(out)))
        return out.value

    def slab_setup(self, rank, world, max_levels=-1):
        """Row-block sharding of the K-Patch levels over `world` ranks (amg_hip_slab_setup)."""
        info = SlabInfo()
        _chk(lib().amg_hip_slab_setup(self._h, int(rank), int(world), int(max_levels), C.byref(info)))
        return info

    def slab_run(self, part):
        """1: down-legs of the slab levels, 2: replicated rest, 3: up-legs (amg_hip_slab_run)."""
        _chk(lib().amg_hip_slab_run(self._h, int(part)))

    def apply_dev(self, v_dev, z_dev):
        """z = M^-1 v (one V-cycle from zero); device pointers."""
        _chk(lib().amg_hip_apply(self._h, C.c_void_p(v_dev), C.c_void_p(z_dev)))

    def pcg(self, rtol=1e-10, max_iters=100):
        """CG on A_0 x = b preconditioned with one V-cycle; returns (x, iters, relres)."""
        it, rel = C.c_int64(0), C.c_double(0)
        _chk(lib().amg_hip_pcg(self._h, rtol, max_iters, C.byref(it), C.byref(rel)))
        return self.get_soln(0), it.value, rel.value

    def solve(self):
        """multigrid.hpp:311-337.  Returns (u, iters, converged, last_rss) and
        prints the reference's convergence line."""
        it, last, conv = C.c_int64(0), C.c_double(0), C.c_int32(0)
        _chk(lib().amg_hip_solve(self._h, self.tolerance, self.every, self.n_iters,
                                 C.byref(it), C.byref(last), C.byref(conv)))
        if conv.value:
            print(f"AMG converged after {it.value} iterations.")
        else:
            print(f"AMG did not converge after {it.value} iterations.")
        return self.get_soln(0), it.value, bool(conv.value), last.value


# ---- stand-alone plug-in operations ---------------------------------------------
def smooth(kind, colptr, rowind, val, u, b, n_iters=1, omega=1.0, tol=1e-9, every=0):
    colptr, rowind, val, b = _a32(colptr), _a32(rowind), _a64(val), _a64(b)
    u = np.array(u, dtype=np.float64, copy=True)
    it, conv = C.c_int64(0), C.c_int32(0)
    st = lib().amg_hip_smooth(kind, colptr.size - 1, _p32(colptr), _p32(rowind), _p64(val),
                              _p64(u), _p64(b), omega, tol, every, n_iters, C.byref(it),
                              C.byref(conv))
    if st == EINVAL:
        raise ValueError(lib().amg_hip_last_error().decode())
    _chk(st)
    return u, it.value, bool(conv.value)


def spgs_sweep(direction, colptr, rowind, val, u, b):
    colptr, rowind, val, b = _a32(colptr), _a32(rowind), _a64(val), _a64(b)
    u = np.array(u, dtype=np.float64, copy=True)
    _chk(lib().amg_hip_spgs_sweep(direction, colptr.size - 1, _p32(colptr), _p32(rowind),
                                  _p64(val), _p64(u), _p64(b)))
    return u


def residual(colptr, rowind, val, u, f):
    colptr, rowind, val, u, f = _a32(colptr), _a32(rowind), _a64(val), _a64(u), _a64(f)
    r = np.empty(colptr.size - 1, np.float64)
    _chk(lib().amg_hip_residual(colptr.size - 1, _p32(colptr), _p32(rowind), _p64(val),
                                _p64(u), _p64(f), _p64(r)))
    return r


def spmv(rows, cols, colptr, rowind, val, v):
    colptr, rowind, val, v = _a32(colptr), _a32(rowind), _a64(val), _a64(v)
    out = np.empty(rows, np.float64)
    _chk(lib().amg_hip_spmv(rows, cols, _p32(colptr), _p32(rowind), _p64(val), _p64(v), _p64(out)))
    return out


def linear_restrict(n_h, n_H, r):
    r = _a64(r)
    out = np.empty(n_H, np.float64)
    _chk(lib().amg_hip_linear_restrict(n_h, n_H, _p64(r), _p64(out)))
    return out


def linear_prolong_add(n_h, n_H, u_H, u_h):
    u_H = _a64(u_H)
    u_h = np.array(u_h, dtype=np.float64, copy=True)
    _chk(lib().amg_hip_linear_prolong_add(n_h, n_H, _p64(u_H), _p64(u_h)))
    return u_h


def rss(colptr, rowind, val, u, b):
    colptr, rowind, val, u, b = _a32(colptr), _a32(rowind), _a64(val), _a64(u), _a64(b)
    out = C.c_double(0)
    _chk(lib().amg_hip_rss_host(colptr.size - 1, _p32(colptr), _p32(rowind), _p64(val),
                                _p64(u), _p64(b), C.byref(out)))
    return out.value


def coarse_solve(colptr, rowind, val, f):
    colptr, rowind, val, f = _a32(colptr), _a32(rowind), _a64(val), _a64(f)
    x = np.empty(f.size, np.float64)
    w = C.c_int64(0)
    _chk(lib().amg_hip_coarse_solve(f.size, _p32(colptr), _p32(rowind), _p64(val), _p64(f),
                                    _p64(x), C.byref(w)))
    return x, w.value


def coarse_solve_fast(colptr, rowind, val, f):
    colptr, rowind, val, f = _a32(colptr), _a32(rowind), _a64(val), _a64(f)
    x = np.empty(f.size, np.float64)
    w, c = C.c_int64(0), C.c_int32(0)
    _chk(lib().amg_hip_coarse_solve_fast(f.size, _p32(colptr), _p32(rowind), _p64(val), _p64(f),
                                         _p64(x), C.byref(w), C.byref(c)))
    return x, w.value, c.value


def csr_shape(rowptr):
    rowptr = _a32(rowptr)
    mb, mr = C.c_int32(0), C.c_int32(0)
    _chk(lib().amg_hip_csr_shape(rowptr.size - 1, _p32(rowptr), C.byref(mb), C.byref(mr)))
    return mb.value, mr.value
