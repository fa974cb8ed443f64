"""ctypes binding of include/qcolloc.h (libqcolloc_hip.so).

This is the only way the Python host layer reaches the evaluator; there is no Python/CPU
implementation of the path behind it.  If the shared library has not been built the import fails
loudly (run `python -c "import __graft_entry__ as g; g.build()"` or `make -C
quantumcollocation.jl_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Import it FIRST so that our
# library's NEEDED libamdhip64.so.7 resolves to the runtime already in the process; two HIP runtimes
# in one process cannot share device pointers or streams.
import torch  # noqa: F401  (device memory, streams, torch.distributed: plumbing only)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libqcolloc_hip.so")
# kernel experiments (profiles/build_variant.sh): QCOLLOC_HIP_VARIANT=name loads csrc/libqcolloc_hip.<name>.so (always in-tree)
if os.environ.get("QCOLLOC_HIP_VARIANT"):
    LIB_PATH = LIB_PATH[:-2] + os.path.basename(os.environ["QCOLLOC_HIP_VARIANT"]) + LIB_PATH[-3:]   # libqcolloc_hip.<name>.so

QC_OK = 0
QC_ERR_INVALID = -1
QC_ERR_NO_DEVICE = -2
QC_ERR_HIP = -3
QC_ERR_UNSUPPORTED = -4
QC_PADE = 0
QC_EXPONENTIAL = 1
# value blocks of an interval (qc_desc.jac_block_order / hess_block_order: permutations of these, all zeros = this order)
QC_JB_F, QC_JB_B, QC_JB_A, QC_JB_H, QC_JB_D = range(5)
QC_JAC_BLOCKS = 5
QC_HB_UA, QC_HB_AU, QC_HB_UH, QC_HB_HU, QC_HB_AA, QC_HB_AH, QC_HB_HH, QC_HB_D = range(8)
QC_HESS_BLOCKS = 8
QC_KERNEL_AUTO = 0
QC_KERNEL_LDS = 1
QC_KERNEL_MFMA = 2
QC_MAX_DERIV = 8
QC_HESS_ALIGN_LINE = 16     # qc_desc.hess_align of the line-aligned (padded) Hessian value layout (device-resident consumers)
QC_FID_UNITARY, QC_FID_KET, QC_FID_DENSITY = 0, 1, 2
QC_REG_DT_SCALED = 2       # (0 and 1 are retired values: the library refuses them)
QC_REG_PLAIN = 3
QC_ABI_VERSION = 6          # QC_VERSION_MAJOR * 1000 + QC_VERSION_MINOR of the include/qcolloc.h this file mirrors
QC_FID_FORM_ABS, QC_FID_FORM_ABS2 = 0, 1
QC_ROWS_STACKED = 0
QC_ROWS_BY_COMPONENT = 1

_c_double_p = C.POINTER(C.c_double)
_c_int64_p = C.POINTER(C.c_int64)


class qc_desc(C.Structure):
    _fields_ = [
        ("N", C.c_int32),
        ("m", C.c_int32),
        ("T", C.c_int64),
        ("zdim", C.c_int32),
        ("global_dim", C.c_int64),
        ("off_U", C.c_int32),
        ("off_a", C.c_int32),
        ("off_dt", C.c_int32),
        ("dt_fixed", C.c_double),
        ("integrator", C.c_int32),
        ("pade_order", C.c_int32),
        ("n_deriv", C.c_int32),
        ("deriv_x_off", C.c_int32 * QC_MAX_DERIV),
        ("deriv_dx_off", C.c_int32 * QC_MAX_DERIV),
        ("deriv_dim", C.c_int32 * QC_MAX_DERIV),
        ("G_drift", _c_double_p),
        ("G_drives", _c_double_p),
        ("device", C.c_int32),
        ("kernel", C.c_int32),
        ("t_begin", C.c_int64),
        ("t_end", C.c_int64),
        ("state_cols", C.c_int32),
        ("hess_align", C.c_int32),
        ("rows_per_interval", C.c_int64),
        ("row_offset", C.c_int64),
        ("jac_per_interval", C.c_int64),
        ("jac_offset", C.c_int64),
        ("hess_per_interval", C.c_int64),
        ("hess_offset", C.c_int64),
        ("row_placement", C.c_int32),
        ("hess_tail_zeros", C.c_int32),
        ("deriv_row_off", C.c_int32 * QC_MAX_DERIV),
        ("jac_block_order", C.c_int32 * QC_JAC_BLOCKS),
        ("hess_block_order", C.c_int32 * QC_HESS_BLOCKS),
    ]


class qc_dims_t(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("n_cols", C.c_int64),
        ("ddim", C.c_int64),
        ("jac_nnz_interval", C.c_int64),
        ("hess_nnz_interval", C.c_int64),
        ("n_intervals", C.c_int64),
        ("F_len", C.c_int64),
        ("jac_nnz", C.c_int64),
        ("hess_nnz", C.c_int64),
        ("Z_len", C.c_int64),
        ("kernel", C.c_int32),
        ("reserved", C.c_int32),
    ]


class qc_fidelity_desc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("N", C.c_int32),
        ("goal_iso", _c_double_p),
        ("subspace", C.POINTER(C.c_int32)),
        ("n_sub", C.c_int32),
        ("form", C.c_int32),
        ("n_phases", C.c_int32),
        ("device", C.c_int32),
        ("phase_dims", C.POINTER(C.c_int32)),
        ("phase_ops", _c_double_p),
    ]


class qc_terms_desc(C.Structure):
    _fields_ = [
        ("T", C.c_int64),
        ("zdim", C.c_int32),
        ("off_dt", C.c_int32),
        ("global_dim", C.c_int64),
        ("dt_fixed", C.c_double),
        ("n_reg", C.c_int32),
        ("weighting", C.c_int32),
        ("reg_index", C.POINTER(C.c_int32)),
        ("reg_R", _c_double_p),
        ("reg_baseline", _c_double_p),
        ("min_time_D", C.c_double),
        ("min_time_knots", C.c_int64),
        ("device", C.c_int32),
        ("reserved0", C.c_int32),
    ]


# Every symbol include/qcolloc.h declares: (name, restype, argtypes).  tests/test_abi.py checks this
# table against the header and against the built library.
_DESC_P = C.POINTER(qc_desc)
_DIMS_P = C.POINTER(qc_dims_t)
_H = C.c_void_p
_TDESC_P = C.POINTER(qc_terms_desc)
SYMBOLS = {
    "qc_operator_to_iso_vec": (C.c_int, [C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "qc_iso_vec_to_operator": (C.c_int, [C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "qc_generator_from_hamiltonian": (C.c_int, [C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "qc_pade_coefficients": (C.c_int, [C.c_int32, _c_double_p]),
    "qc_desc_dims": (C.c_int, [_DESC_P, _DIMS_P]),
    "qc_desc_jac_structure": (C.c_int, [_DESC_P, _c_int64_p, _c_int64_p, C.c_int]),
    "qc_desc_hess_structure": (C.c_int, [_DESC_P, _c_int64_p, _c_int64_p, C.c_int]),
    "qc_create": (C.c_int, [_DESC_P, C.POINTER(_H)]),
    "qc_destroy": (None, [_H]),
    "qc_last_error": (C.c_char_p, [_H]),
    "qc_kernel_name": (C.c_char_p, [_H, C.c_int32]),
    "qc_dims": (C.c_int, [_H, _DIMS_P]),
    "qc_jac_structure": (C.c_int, [_H, _c_int64_p, _c_int64_p, C.c_int]),
    "qc_hess_structure": (C.c_int, [_H, _c_int64_p, _c_int64_p, C.c_int]),
    "qc_eval_F": (C.c_int, [_H, _c_double_p, _c_double_p]),
    "qc_eval_jac": (C.c_int, [_H, _c_double_p, _c_double_p]),
    "qc_eval_F_jac": (C.c_int, [_H, _c_double_p, _c_double_p, _c_double_p]),
    "qc_eval_hess": (C.c_int, [_H, _c_double_p, _c_double_p, _c_double_p]),
    "qc_set_new_x": (C.c_int, [_H, C.c_int]),
    "qc_knot_generation": (C.c_int64, [_H]),
    "qc_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "qc_host_free": (C.c_int, [C.c_void_p]),
    "qc_eval_F_jac_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_eval_hess_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_eval_F_jac_hess_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_eval_F_jac_dev_multi": (C.c_int, [C.POINTER(_H), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_eval_hess_dev_multi": (C.c_int, [C.POINTER(_H), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_eval_F_list": (C.c_int, [C.POINTER(_H), C.c_int32, _c_double_p, _c_double_p]),
    "qc_eval_jac_list": (C.c_int, [C.POINTER(_H), C.c_int32, _c_double_p, _c_double_p]),
    "qc_eval_F_jac_list": (C.c_int, [C.POINTER(_H), C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "qc_eval_hess_list": (C.c_int, [C.POINTER(_H), C.c_int32, _c_double_p, _c_double_p, _c_double_p]),
    "qc_create_multi": (C.c_int, [_DESC_P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(_H)]),
    "qc_multi_count": (C.c_int32, [_H]),
    "qc_multi_shard": (_H, [_H, C.c_int32]),
    "qc_multi_shard_info": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_int32), _c_int64_p, _c_int64_p]),
    "qc_multi_padded_len": (C.c_int64, [_H, C.c_int64]),
    "qc_multi_eval_F_jac_dev": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "qc_multi_eval_hess_dev": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "qc_multi_sync": (C.c_int, [_H]),
    "qc_multi_all_gather_dev": (C.c_int, [_H, C.POINTER(C.c_void_p), C.c_int64]),
    "qc_sizeof_desc": (C.c_int64, []),
    "qc_sizeof_dims": (C.c_int64, []),
    "qc_sizeof_terms_desc": (C.c_int64, []),
    "qc_fidelity_create": (C.c_int, [C.c_int32, _c_double_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(_H)]),
    "qc_fidelity_create_desc": (C.c_int, [C.POINTER(qc_fidelity_desc), C.POINTER(_H)]),
    "qc_fidelity_input_len": (C.c_int32, [_H]),
    "qc_hermitian_eig": (C.c_int, [C.c_int32, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "qc_fidelity_create_kind": (C.c_int, [C.c_int32, C.c_int32, _c_double_p, C.c_int32, C.POINTER(_H)]),
    "qc_fidelity_destroy": (None, [_H]),
    "qc_fidelity_last_error": (C.c_char_p, [_H]),
    "qc_fidelity_eval": (C.c_int, [_H, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "qc_fidelity_eval_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_rollout": (C.c_int, [_H, _c_double_p, _c_double_p, _c_double_p]),
    "qc_rollout_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_terms_desc_hess_nnz": (C.c_int, [_TDESC_P, _c_int64_p]),
    "qc_terms_desc_hess_structure": (C.c_int, [_TDESC_P, _c_int64_p, _c_int64_p, C.c_int]),
    "qc_terms_create": (C.c_int, [_TDESC_P, C.POINTER(_H)]),
    "qc_terms_destroy": (None, [_H]),
    "qc_terms_last_error": (C.c_char_p, [_H]),
    "qc_terms_hess_nnz": (C.c_int, [_H, _c_int64_p]),
    "qc_terms_hess_structure": (C.c_int, [_H, _c_int64_p, _c_int64_p, C.c_int]),
    "qc_terms_eval": (C.c_int, [_H, _c_double_p, _c_double_p, _c_double_p, _c_double_p]),
    "qc_terms_eval_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "qc_debug_read_stamps": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int64]),
    "qc_debug_host_expand_rate": (C.c_int, [_H, C.c_int32, _c_double_p]),
    "qc_version": (C.c_char_p, []),
    "qc_abi_version": (C.c_int32, []),
}


class QCollocError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libqcolloc_hip error {code}: {msg}")
        self.code = code


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for this path."
        )
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.qc_abi_version() != QC_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.qc_abi_version()} but this binding mirrors {QC_ABI_VERSION} "
                          "(constants were renumbered between minor versions; stale build? run __graft_entry__.build())")
    # the struct mirrors above must be the structs this build of the library was compiled with
    for name, mirror in (("qc_sizeof_desc", qc_desc), ("qc_sizeof_dims", qc_dims_t), ("qc_sizeof_terms_desc", qc_terms_desc)):
        if getattr(lib, name)() != C.sizeof(mirror):
            raise ImportError(f"{LIB_PATH}: {name}() = {getattr(lib, name)()} but the Python mirror has {C.sizeof(mirror)} bytes "
                              "(stale build? run __graft_entry__.build())")
    return lib


lib = _load()


def check(rc: int, handle=None) -> None:
    if rc != QC_OK:
        msg = lib.qc_last_error(handle)
        raise QCollocError(rc, msg.decode() if msg else "unknown error")


def dptr(a: np.ndarray):
    """double* of a C- or F-contiguous float64 numpy array (no copy)."""
    assert a.dtype == np.float64 and (a.flags.c_contiguous or a.flags.f_contiguous)
    return a.ctypes.data_as(_c_double_p)


def iptr(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(_c_int64_p)
