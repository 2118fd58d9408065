"""ctypes binding of libpyvb_hip.so (declared in include/pyvb_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load,
importing this module raises, and so does every product path that needs it.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYVB_HIP_LIB: another build of the same library (kernel experiments, profiles/); there is still no other backend
LIB_PATH = os.environ.get("PYVB_HIP_LIB") or os.path.join(_HERE, "libpyvb_hip.so")

OK, E_ARG, E_HIP, E_LINALG, E_STALE, E_RCCL, E_UNSUPPORTED = range(7)
NOISE_DIAGONAL_GAMMA, NOISE_GAMMA, NOISE_WISHART = 0, 1, 2
FORWARD, BACKWARD = 0, 1
K_PREP, K_SWEEP_FWD, K_STATS, K_PARAMS, K_STEP, K_SWEEP_BWD, K_ELBO, K_GY = range(8)

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_h = ctypes.c_void_p
# include/pyvb_hip.h: pyvb_host_allreduce_fn -- int (*)(double* buf, size_t count, void* user)
HOST_ALLREDUCE = ctypes.CFUNCTYPE(ctypes.c_int, _dp, ctypes.c_size_t, ctypes.c_void_p)

# name -> (restype, argtypes): every symbol include/pyvb_hip.h declares
SIGNATURES = {
    "pyvb_last_error": (ctypes.c_char_p, []),
    "pyvb_version": (ctypes.c_int, []),
    "pyvb_build_id": (ctypes.c_char_p, []),
    "pyvb_device_count": (ctypes.c_int, [_ip]),
    "pyvb_lds_create": (ctypes.c_int, [ctypes.POINTER(_h)] + [ctypes.c_int] * 6),
    "pyvb_lds_destroy": (ctypes.c_int, [_h]),
    "pyvb_lds_set_priors": (ctypes.c_int, [_h] + [_dp] * 10),
    "pyvb_lds_set_wishart_priors": (ctypes.c_int, [_h, ctypes.c_double, _dp, ctypes.c_double, _dp]),
    "pyvb_lds_set_wishart_state": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_get_wishart_state": (ctypes.c_int, [_h, _dp, _dp, _dp, _dp]),
    "pyvb_lds_set_column_cov": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_get_column_cov": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_set_observations": (ctypes.c_int, [_h, _dp]),
    "pyvb_lds_set_output_state": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_update_Y": (ctypes.c_int, [_h]),
    "pyvb_lds_get_outputs": (ctypes.c_int, [_h, _dp, _dp, _dp]),
    "pyvb_lds_set_column_observations": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_set_state": (ctypes.c_int, [_h] + [_dp] * 7),
    "pyvb_lds_get_state": (ctypes.c_int, [_h] + [_dp] * 9),
    "pyvb_lds_get_posterior_classes": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_set_posterior_classes": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_get_column_qld": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_lds_sweep": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_lds_update_x": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_lds_update_A": (ctypes.c_int, [_h]),
    "pyvb_lds_update_C": (ctypes.c_int, [_h]),
    "pyvb_lds_update_columns": (ctypes.c_int, [_h, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "pyvb_lds_update_Q": (ctypes.c_int, [_h]),
    "pyvb_lds_update_R": (ctypes.c_int, [_h]),
    "pyvb_lds_elbo": (ctypes.c_int, [_h]),
    "pyvb_lds_get_elbo": (ctypes.c_int, [_h, _dp]),
    "pyvb_lds_elbo_total": (ctypes.c_int, [_h, _dp]),
    "pyvb_lds_iterate": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_lds_get_elbo_history": (ctypes.c_int, [_h, _dp, ctypes.c_int, _ip]),
    "pyvb_lds_reset_elbo_history": (ctypes.c_int, [_h]),
    "pyvb_lds_sync": (ctypes.c_int, [_h]),
    "pyvb_lds_timing_enable": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_lds_timing_reset": (ctypes.c_int, [_h]),
    "pyvb_lds_timing_get": (ctypes.c_int, [_h, ctypes.c_int, _dp, _ip]),
    "pyvb_lds_get_warmup": (ctypes.c_int, [_h, _ip]),
    "pyvb_lds_get_time_split": (ctypes.c_int, [_h, _ip]),
    "pyvb_lds_set_time_split": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_comm_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "pyvb_lds_comm_init": (ctypes.c_int, [_h, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "pyvb_lds_comm_destroy": (ctypes.c_int, [_h]),
    "pyvb_lds_comm_init_host": (ctypes.c_int, [_h, HOST_ALLREDUCE, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "pyvb_graph_create": (ctypes.c_int, [ctypes.POINTER(_h), ctypes.c_int, ctypes.c_size_t]),
    "pyvb_graph_destroy": (ctypes.c_int, [_h]),
    "pyvb_graph_write": (ctypes.c_int, [_h, ctypes.c_size_t, _dp, ctypes.c_size_t]),
    "pyvb_graph_read": (ctypes.c_int, [_h, ctypes.c_size_t, _dp, ctypes.c_size_t]),
    "pyvb_graph_tape_create": (ctypes.c_int, [_h, _ip, ctypes.c_int, _ip]),
    "pyvb_graph_tape_set_program": (ctypes.c_int, [_h, ctypes.c_int, _ip, ctypes.c_int, _ip, ctypes.c_int]),
    "pyvb_graph_tape_run": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_graph_tape_destroy": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_graph_sync": (ctypes.c_int, [_h]),
    "pyvb_pca_create": (ctypes.c_int, [ctypes.POINTER(_h), ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_long]),
    "pyvb_pca_destroy": (ctypes.c_int, [_h]),
    "pyvb_pca_set_priors": (ctypes.c_int, [_h, _dp, _dp, _dp, _dp, ctypes.c_double, ctypes.c_double]),
    "pyvb_pca_set_data": (ctypes.c_int, [_h, _dp]),
    "pyvb_pca_set_state": (ctypes.c_int, [_h] + [_dp] * 6),
    "pyvb_pca_set_unpinned_rows": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_pca_set_initial_variances": (ctypes.c_int, [_h, _dp, _dp]),
    "pyvb_pca_get_state": (ctypes.c_int, [_h] + [_dp] * 9),
    "pyvb_pca_get_qld": (ctypes.c_int, [_h, _dp, _dp, _dp, _dp]),
    "pyvb_pca_update_W": (ctypes.c_int, [_h]),
    "pyvb_pca_update_Z": (ctypes.c_int, [_h]),
    "pyvb_pca_update_X": (ctypes.c_int, [_h, ctypes.c_long, ctypes.c_long]),
    "pyvb_pca_update_X0": (ctypes.c_int, [_h]),
    "pyvb_pca_update_Mu": (ctypes.c_int, [_h]),
    "pyvb_pca_update_Beta": (ctypes.c_int, [_h]),
    "pyvb_pca_elbo": (ctypes.c_int, [_h, _dp]),
    "pyvb_pca_iterate": (ctypes.c_int, [_h, ctypes.c_int]),
    "pyvb_pca_sync": (ctypes.c_int, [_h]),
    "pyvb_pca_comm_init": (ctypes.c_int, [_h, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]),
    "pyvb_pca_comm_init_host": (ctypes.c_int, [_h, HOST_ALLREDUCE, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
}


class PyvbHipError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libpyvb_hip status %d: %s" % (code, message))
        self.code = code


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C pyvb_amd/csrc` (or __graft_entry__.build()); "
            "pyvb_amd has no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def source_id():
    """What pyvb_build_id() must return for a library built from the sources in this tree (pyvb_amd/csrc/Makefile: BUILD_ID),
    or None when the sources are not beside the package."""
    import hashlib
    import re
    src = os.path.join(_HERE, "csrc")
    try:
        mk = open(os.path.join(src, "Makefile")).read()
        names = re.search(r"^SRCS\s*=\s*(.*)$", mk, re.M).group(1).split() + re.search(r"^HDRS\s*=\s*(.*)$", mk, re.M).group(1).split()
        h = hashlib.sha256()
        for n in names + ["Makefile"]:
            h.update(open(os.path.join(src, n), "rb").read())
        return h.hexdigest()[:32]
    except (OSError, AttributeError):
        return None


def check(rc):
    if rc != OK:
        msg = lib.pyvb_last_error().decode("utf-8", "replace")
        if rc == E_LINALG:
            raise np.linalg.LinAlgError(msg)
        raise PyvbHipError(rc, msg)


def dptr(a):
    """numpy float64 C-contiguous array (or None) -> double*"""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)
