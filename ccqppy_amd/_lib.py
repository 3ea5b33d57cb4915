"""ctypes binding of libccqp_hip.so (the C ABI declared in include/ccqp_hip.h).

There is NO CPU fallback: if the shared library is missing or fails to load, importing the
compute path raises `CCQPHipUnavailable` with the build instruction.  `import torch` happens
first so that the library binds to the HIP runtime torch already mapped (one runtime/process).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL: shares libamdhip64.so.7 with torch)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libccqp_hip.so")

NUM_SLOTS = 64
MAX_DOTS = 8
SMALL_SEG_MAX = 64
MAX_STEPS = 16

ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED = 1, 2, 3   # CCQP_ERR_*
ABI_VERSION = 2          # CCQP_ABI_VERSION of include/ccqp_hip.h this binding was written against
LOCKSTEP_LOOKAHEAD = 1 << 17   # CCQP_LOCKSTEP_LOOKAHEAD: flags bit of ccqp_bapgd_* (every problem carries ga / Aya)
LOCKSTEP_FUSED_ROUNDS = 1 << 18  # CCQP_LOCKSTEP_FUSED_ROUNDS: one round per accepted APGD iteration (budget in slab slot 48)
# per-context tuning options (ccqp_ctx_set_option): name -> key
OPTIONS = {name: key for key, name in enumerate((
    "GEMV_ROW_STRIDE", "GEMV2_WPB", "GEMV2_SHAPE", "GEMV2_GROUP", "BMV3_SHAPE", "LOOKAHEAD_MIN_N", "MPRGP_LOOKAHEAD_MIN_N",
    "MPRGP_LOOKAHEAD", "APGD_LOOKAHEAD", "BMV_STAGGER", "MPRGP_FUSED_TAIL", "PERS1_MIN_N", "BMV_LOADS",
    "SMALL_SHARING", "PERS_MIN_MIB", "BPIPE_STEP_IN_TAIL",
    "APGD_UPDATE_IN_DECISION"))}

OP_IDENTITY, OP_LOWER, OP_UPPER, OP_BOX, OP_SPHERE, OP_CONE = range(6)
SEG_LB_INT, SEG_UB_INT, SEG_P0_INT = 1, 2, 4


class CCQPHipUnavailable(ImportError):
    pass


class CCQPHipError(RuntimeError):
    pass


class Seg(ctypes.Structure):
    """struct ccqp_seg (include/ccqp_hip.h)."""
    _fields_ = [("kind", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("offset", ctypes.c_int64), ("dim", ctypes.c_int64),
                ("lb", ctypes.c_void_p), ("ub", ctypes.c_void_p),
                ("lb_s", ctypes.c_double), ("ub_s", ctypes.c_double)]


class SegWork(ctypes.Structure):
    """struct ccqp_segwork (include/ccqp_hip.h): one workgroup's share of a segmented projection."""
    _fields_ = [("type", ctypes.c_int32), ("seg", ctypes.c_int32), ("chunk", ctypes.c_int32),
                ("pbase", ctypes.c_int32)]


class Bop(ctypes.Structure):
    """struct ccqp_bop (include/ccqp_hip.h): one (problem, operation) descriptor of a batched launch."""
    _fields_ = [("p", ctypes.c_void_p * 16), ("s", ctypes.c_double * 4),
                ("n", ctypes.c_int64), ("n2", ctypes.c_int64), ("lda", ctypes.c_int64),
                ("slab", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("slot", ctypes.c_int32 * 4), ("k", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("seg", Seg)]


class PProb(ctypes.Structure):
    """struct ccqp_pprob (include/ccqp_hip.h): one problem of a batched pipelined BBPGD run."""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("b", ctypes.c_void_p),
                ("x", ctypes.c_void_p * 3), ("g", ctypes.c_void_p * 3), ("y", ctypes.c_void_p),
                ("seg", Seg), ("slab", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("counter", ctypes.c_void_p), ("c", ctypes.c_double), ("tol", ctypes.c_double),
                ("xmin", ctypes.c_void_p), ("gmin", ctypes.c_void_p)]


class PSegs(ctypes.Structure):
    """struct ccqp_psegs (include/ccqp_hip.h): the feasible set of one problem of a segmented lock-step run."""
    _fields_ = [("table", ctypes.c_void_p), ("work", ctypes.c_void_p), ("normwork", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("npartials", ctypes.c_int64), ("nwork", ctypes.c_int32),
                ("nnorm", ctypes.c_int32)]


class PApgd(ctypes.Structure):
    """struct ccqp_papgd (include/ccqp_hip.h): one problem of a lock-step APGD run."""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("b", ctypes.c_void_p),
                ("X", ctypes.c_void_p * 2), ("Y", ctypes.c_void_p * 2), ("gk", ctypes.c_void_p),
                ("g1", ctypes.c_void_p), ("Ayk", ctypes.c_void_p), ("Ax1", ctypes.c_void_p),
                ("xhat", ctypes.c_void_p), ("seg", Seg), ("slab", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("counter", ctypes.c_void_p), ("c", ctypes.c_double), ("tol", ctypes.c_double),
                ("ga", ctypes.c_void_p), ("Aya", ctypes.c_void_p)]


class PMprgp(ctypes.Structure):
    """struct ccqp_pmprgp (include/ccqp_hip.h): one problem of a lock-step MPRGP run."""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("b", ctypes.c_void_p),
                ("X", ctypes.c_void_p * 2), ("G", ctypes.c_void_p * 2)] + \
               [(name, ctypes.c_void_p) for name in ("p", "Ap", "psi", "w", "nv", "txk", "dx", "Adx", "ytmp")] + \
               [("seg", Seg), ("slab", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("counter", ctypes.c_void_p), ("c", ctypes.c_double), ("tol", ctypes.c_double)]


class PSpg(ctypes.Structure):
    """struct ccqp_pspg (include/ccqp_hip.h): one problem of a lock-step SPG run (device side)."""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("b", ctypes.c_void_p),
                ("x", ctypes.c_void_p), ("g", ctypes.c_void_p), ("d", ctypes.c_void_p), ("Ad", ctypes.c_void_p),
                ("seg", Seg), ("partials", ctypes.c_void_p), ("counter", ctypes.c_void_p)]


class SpgHost(ctypes.Structure):
    """struct ccqp_spg_host (include/ccqp_hip.h): host-side state of one problem of a lock-step SPG run."""
    _fields_ = [("tol", ctypes.c_double), ("max_mv", ctypes.c_double), ("tau", ctypes.c_double),
                ("sigma1", ctypes.c_double), ("sigma2", ctypes.c_double),
                ("alpha", ctypes.c_double), ("betak", ctypes.c_double), ("fk", ctypes.c_double),
                ("dd", ctypes.c_double),
                ("mv", ctypes.c_int64), ("used", ctypes.c_int64), ("reserved_", ctypes.c_int64),
                ("m", ctypes.c_int32), ("qlen", ctypes.c_int32), ("status", ctypes.c_int32),
                ("pending", ctypes.c_int32), ("have_dd", ctypes.c_int32), ("pad_", ctypes.c_int32)]


(BOP_GEMV, BOP_AXPBY, BOP_SCALE, BOP_COPY, BOP_FILL, BOP_DOTS, BOP_PROJ_ELEM, BOP_RESIDUAL_ELEM,
 BOP_FEASIBLE_ELEM, BOP_MPRGP_SPLIT, BOP_MPRGP_NORMAL_DOT, BOP_MPRGP_PROPORTION, BOP_FREE_GRADIENT_ELEM,
 BOP_CG_DIRECTION) = range(1, 15)

_c = ctypes
_P = _c.c_void_p
_I64 = _c.c_int64
_D = _c.c_double
_I = _c.c_int

# name -> argtypes (restype is int unless noted); must list EVERY symbol of include/ccqp_hip.h
SIGNATURES = {
    "ccqp_abi_version": [],
    "ccqp_last_error": [],
    "ccqp_ctx_create": [_I, _c.POINTER(_P)],
    "ccqp_ctx_destroy": [_P],
    "ccqp_ctx_info": [_P, _c.POINTER(_I), _c.POINTER(_I), _c.c_char_p, _I],
    "ccqp_ctx_set_option": [_P, _I, _I64],
    "ccqp_ctx_get_option": [_P, _I, _c.POINTER(_I64)],
    "ccqp_scalars_read": [_P, _I, _I, _c.POINTER(_D), _P],
    "ccqp_scalars_read_async": [_P, _I, _I, _P],
    "ccqp_scalars_host": [_P, _I, _I, _c.POINTER(_D)],
    "ccqp_scalars_devptr": [_P, _c.POINTER(_P)],
    "ccqp_gemv_f64": [_P, _P, _I64, _P, _P, _I64, _I64, _I, _P],
    "ccqp_gemv_plan": [_P, _I64, _I64, _I, _c.POINTER(_I64), _c.POINTER(_I64)],
    "ccqp_gemv_form": [_P, _I64, _I64, _I, _c.c_char_p, _I],
    "ccqp_gemv2_f64": [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _P],
    "ccqp_gemv2_form": [_P, _I64, _I64, _I, _c.c_char_p, _I],
    "ccqp_gemv3_f64": [_P, _P, _I64, _P, _P, _P, _P, _I64, _I, _P],
    "ccqp_gemv3_form": [_P, _I64, _I, _c.c_char_p, _I],
    "ccqp_gemvk_f64": [_P, _P, _I64, _I, _P, _P, _P, _P, _I64, _I64, _I, _P],
    "ccqp_gemvk_form": [_P, _I, _I64, _I64, _I, _c.c_char_p, _I],
    "ccqp_mv_timing_begin": [_I],
    "ccqp_mv_timing_end": [_c.POINTER(_c.c_double), _I, _c.POINTER(_I)],
    "ccqp_gemv_epilogue_f64": [_P, _P, _I64, _P, _P, _P, _P, _I64, _I64, _I, _P],
    "ccqp_axpby_f64": [_P, _D, _P, _D, _P, _P, _I64, _P],
    "ccqp_axpby2_f64": [_P, _D, _P, _D, _P, _P, _D, _P, _D, _P, _P, _I64, _P],
    "ccqp_scale_f64": [_P, _D, _P, _P, _I64, _P],
    "ccqp_copy_f64": [_P, _P, _P, _I64, _P],
    "ccqp_fill_f64": [_P, _D, _P, _I64, _P],
    "ccqp_mul_f64": [_P, _P, _P, _P, _I64, _P],
    "ccqp_muldiv_f64": [_P, _P, _P, _D, _P, _I64, _P],
    "ccqp_submul_f64": [_P, _P, _P, _P, _P, _I64, _P],
    "ccqp_dots_f64": [_P, _I, _c.POINTER(_P), _c.POINTER(_P), _I64, _I, _P],
    "ccqp_proj_elementwise_f64": [_P, _c.POINTER(Seg), _D, _P, _D, _P, _P, _P],
    "ccqp_proj_elementwise_sub_f64": [_P, _c.POINTER(Seg), _D, _P, _D, _P, _P, _P, _P],
    "ccqp_proj_normscale_f64": [_P, _c.POINTER(Seg), _P, _P, _I, _P],
    "ccqp_proj_small_segments_f64": [_P, _P, _I64, _P, _P, _I, _P],
    "ccqp_comm_rccl_version": [],
    "ccqp_comm_unique_id": [_P],
    "ccqp_comm_create": [_P, _I, _I, _P, _c.POINTER(_P)],
    "ccqp_comm_destroy": [_P],
    "ccqp_comm_info": [_P, _c.POINTER(_I), _c.POINTER(_I)],
    "ccqp_comm_allgather_f64": [_P, _P, _I64, _P],
    "ccqp_comm_allreduce_sum_f64": [_P, _P, _I64, _P],
    "ccqp_shard_gemv_allgather_f64": [_P, _P, _P, _I64, _P, _P, _I64, _I64, _I64, _I, _I, _P],
    "ccqp_shard_gemvk_allgather_f64": [_P, _P, _P, _I64, _I, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P],
    "ccqp_shard_gemv2_allgather_f64": [_P, _P, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P],
    "ccqp_unpack_gathered_f64": [_P, _P, _I, _I64, _I64, _P, _P, _P, _P],
    "ccqp_segs_plan": [_P, _I64, _P, _I64, _c.POINTER(_I64), _P, _I64, _c.POINTER(_I64), _c.POINTER(_I64)],
    "ccqp_segs_workspace_doubles": [_I64, _I64],
    "ccqp_segs_project_f64": [_P, _P, _P, _I64, _P, _I64, _I64, _P, _D, _P, _D, _P, _P, _P, _P],
    "ccqp_segs_residual_sq_f64": [_P, _P, _P, _I64, _P, _I64, _I64, _P, _P, _P, _D, _D, _I, _P],
    "ccqp_segs_normal_f64": [_P, _P, _P, _I64, _P, _I64, _I64, _P, _P, _P, _P],
    "ccqp_segs_feasible_steps_f64": [_P, _P, _P, _I64, _P, _I64, _I64, _P, _P, _P, _D, _I, _I, _P],
    "ccqp_residual_sq_elementwise_f64": [_P, _c.POINTER(Seg), _P, _P, _D, _D, _I, _P],
    "ccqp_fused_dots_f64": [_P, _I, _c.POINTER(_P), _c.POINTER(_P), _c.POINTER(_P), _c.POINTER(_P), _I64, _I,
                            _c.POINTER(Seg), _P, _P, _D, _D, _I, _P],
    "ccqp_pipe_reset": [_P, _D, _P],
    "ccqp_pipe_step_project_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _P],
    "ccqp_pipe_gemv_epilogue_f64": [_P, _P, _I64, _P, _P, _P, _P, _I64, _I64, _I, _P],
    "ccqp_pipe_bb_tail_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _P, _D, _D, _D, _D, _P],
    "ccqp_pipe_spmv_csr_f64": [_P, _P, _P, _P, _P, _c.POINTER(_I64), _P, _P, _P, _P, _I64, _I64, _I64, _I, _P],
    "ccqp_pipe_mul_f64": [_P, _P, _P, _P, _I64, _P],
    "ccqp_pipe_axpby_f64": [_P, _D, _P, _D, _P, _P, _I64, _P],
    "ccqp_pipe_poll": [_P, _c.POINTER(_D)],
    "ccqp_pipe_capture_begin": [_P, _P],
    "ccqp_pipe_capture_end": [_P, _P, _c.POINTER(_P)],
    "ccqp_pipe_graph_launch": [_P, _P, _P],
    "ccqp_pipe_graph_destroy": [_P],
    "ccqp_bpipe_partials_doubles": [_I64],
    "ccqp_bpipe_matvec_form": [_P, _I, _I64, _I, _c.c_char_p, _I],
    "ccqp_bmv3_shape": [_P, _I],
    "ccqp_bapgd_fused_form": [_P, _I, _I64, _I],
    "ccqp_bpipe_start_f64": [_P, _P, _I, _I64, _D, _I, _P, _P],
    "ccqp_bpipe_iterate_f64": [_P, _P, _I, _I64, _I64, _D, _D, _I, _I, _P, _P],
    "ccqp_bpipe_poll": [_P, _c.POINTER(_D)],
    "ccqp_bspg_host_step": [_P, _P, _I64, _P, _P, _P, _P, _I],
    "ccqp_bspg_run_f64": [_P, _P, _P, _P, _P, _I64, _P, _I, _I64, _I64, _I64, _I, _I, _I64, _P, _P, _P,
                          _c.POINTER(_I), _P],
    "ccqp_bsegs_start_f64": [_P, _P, _P, _I, _I64, _I64, _I64, _D, _I, _P, _P],
    "ccqp_bsegs_apgd_round_f64": [_P, _P, _P, _I, _I64, _I64, _I64, _I64, _D, _I, _P, _P],
    "ccqp_bsegs_iterate_f64": [_P, _P, _P, _I, _I64, _I64, _I64, _I64, _D, _D, _I, _I, _P, _P],
    "ccqp_bsolve_bbpgd_f64": [_P, _P, _I, _I64, _D, _D, _I64, _I, _P],
    "ccqp_bsolve_apgd_f64": [_P, _P, _I, _I64, _D, _D, _I, _P],
    "ccqp_bsolve_mprgp_f64": [_P, _P, _I, _I64, _D, _D, _D, _I, _P],
    "ccqp_bapgd_stage_one_f64": [_P, _c.POINTER(PApgd), _P, _P, _I64, _D, _P, _P],
    "ccqp_bpipe_stage_one_f64": [_P, _c.POINTER(PProb), _P, _P, _I64, _D, _P, _P],
    "ccqp_bmprgp_stage_one_f64": [_P, _c.POINTER(PMprgp), _P, _P, _I64, _D, _P, _P],
    "ccqp_bmprgp_fused_form": [_P, _I, _I64, _I],
    "ccqp_bspg_stage_one_f64": [_P, _c.POINTER(PSpg), _P, _P, _I64, _P, _P, _P],
    "ccqp_read_f64": [_P, _P, _I, _c.POINTER(_D), _P],
    "ccqp_bsolve_apgd_grid_f64": [_P, _P, _I64, _D, _I, _I64, _P, _P],
    "ccqp_bapgd_start_f64": [_P, _P, _I, _I64, _I, _P, _P],
    "ccqp_bapgd_round_f64": [_P, _P, _I, _I64, _I64, _D, _I, _P, _P],
    "ccqp_bmprgp_partial_rows": [],
    "ccqp_bmprgp_start_f64": [_P, _P, _I, _I64, _D, _I, _P, _P],
    "ccqp_bmprgp_round_f64": [_P, _P, _I, _I64, _I64, _D, _D, _I, _P, _P],
    "ccqp_bsegs_mprgp_start_f64": [_P, _P, _P, _I, _I64, _I64, _I64, _D, _I, _P, _P],
    "ccqp_bsegs_mprgp_round_f64": [_P, _P, _P, _I, _I64, _I64, _I64, _I64, _D, _D, _I, _P, _P],
    "ccqp_bb_tail_elementwise_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _P, _D, _D, _I, _P],
    "ccqp_isclose_mask_mul_f64": [_P, _P, _P, _P, _I, _P, _I64, _P],
    "ccqp_count_notclose_f64": [_P, _P, _P, _I64, _I, _P],
    "ccqp_feasible_steps_elementwise_f64": [_P, _c.POINTER(Seg), _P, _P, _D, _I, _I, _P],
    "ccqp_normal_elementwise_f64": [_P, _c.POINTER(Seg), _P, _P, _I, _P],
    "ccqp_normal_sphere_f64": [_P, _c.POINTER(Seg), _P, _P, _I, _I, _P],
    "ccqp_chopped_gradient_f64": [_P, _P, _P, _P, _I, _P, _I64, _P],
    "ccqp_mprgp_split_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _P, _P, _I, _P],
    "ccqp_mprgp_normal_dot_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _I, _I, _P],
    "ccqp_mprgp_proportion_f64": [_P, _P, _P, _P, _P, _I, _I64, _I, _P],
    "ccqp_free_gradient_elementwise_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _P],
    "ccqp_cg_direction_f64": [_P, _P, _P, _D, _P, _P, _I64, _P],
    "ccqp_projected_gradient_elementwise_f64": [_P, _c.POINTER(Seg), _P, _P, _P, _P, _P, _P],
    "ccqp_batch_serialize_matvecs": [_P, _I],
    "ccqp_ctx_workspace": [_P, _c.POINTER(_P), _c.POINTER(_P)],
    "ccqp_batch_run_f64": [_P, _I, _c.POINTER(_I), _c.POINTER(_I), _c.POINTER(Bop), _P],
    "ccqp_scalars_gather_async": [_P, _I, _c.POINTER(_P), _c.POINTER(_I), _c.POINTER(_I), _P],
    "ccqp_scalars_gather_host": [_P, _I, _c.POINTER(_D)],
    "ccqp_symv_workspace_doubles": [_I64],
    "ccqp_symv_upper_f64": [_P, _P, _I64, _P, _P, _P, _P, _I64, _P, _P],
    "ccqp_spmv_plan_words": [_I64, _I64],
    "ccqp_spmv_plan_csr": [_P, _P, _P, _I64, _I64, _P, _c.POINTER(_I64), _P],
    "ccqp_spmv_plan_set": [_P, _P, _I64, _I, _c.POINTER(_I64), _P],
    "ccqp_spmv_csr_f64": [_P, _P, _P, _P, _P, _c.POINTER(_I64), _P, _P, _P, _P, _I64, _I64, _I64, _I, _P],
    "ccqp_csr_validate": [_P, _P, _P, _I64, _I64, _I64, _I, _P],
    "ccqp_gen_spd_rows_f64": [_P, _P, _I64, _I64, _I64, _I64, _c.c_uint64, _P],
    "ccqp_gen_vec_f64": [_P, _P, _I64, _c.c_uint64, _P],
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises CCQPHipUnavailable if not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CCQPHipUnavailable(
            "ccqppy_amd: %s is missing.  Build it with `make` at the repository root "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise CCQPHipUnavailable("ccqppy_amd: cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CCQPHipUnavailable("ccqppy_amd: %s does not export %s (stale build?)"
                                     % (LIB_PATH, name)) from e
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == "ccqp_last_error" else
                      ctypes.c_int64 if name in ("ccqp_symv_workspace_doubles", "ccqp_spmv_plan_words",
                                                 "ccqp_bpipe_partials_doubles", "ccqp_segs_workspace_doubles")
                      else ctypes.c_int)
    if lib.ccqp_abi_version() != ABI_VERSION:
        raise CCQPHipUnavailable("ccqppy_amd: ABI version mismatch in %s (library %d, binding %d): rebuild with `make`"
                                 % (LIB_PATH, lib.ccqp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = _lib.ccqp_last_error()
        raise CCQPHipError("libccqp_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
