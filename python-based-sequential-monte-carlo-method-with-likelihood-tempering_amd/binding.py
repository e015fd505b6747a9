"""ctypes binding of libsmc_hip.so - exactly the symbols include/smc_hip.h declares.

The product path fails loudly when the HIP library is missing or a call fails: there is no CPU
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMC_HIP_LIB: another build of the same library (A/B timing of kernel variants on one box, tools/); default: the in-tree one
LIB_PATH = os.environ.get("SMC_HIP_LIB") or os.path.join(_HERE, "libsmc_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "smc_hip.h")

SMC_SET_PRED, SMC_SET_FILT = 0, 1
SMC_PRIOR_UNIFORM, SMC_PRIOR_NORMAL, SMC_PRIOR_FLAT = 0, 1, 2
SMC_PRIOR_MODE_MASK, SMC_PRIOR_MODE_RATIO_MASK, SMC_PRIOR_MODE_RATIO = 0, 1, 2
PRIOR_MODES = {"mask": 0, "ratio_mask": 1, "ratio": 2}
RESAMPLING = {"residual_systematic": 0, "systematic": 1, "multinomial": 2}
SMC_MAX_ESS_CAND = 16
SMC_ABI_VERSION = 3
SMC_SWEEP_COUNTER_WORDS = 14
SWEEP_COUNTER_NAMES = ("n_failed", "rk_attempts", "accepted_now", "accepted_ever", "newton_iters", "factorisations", "failed_solves",
                       "expected_solves", "completed_solves", "unsolved_items", "wave_split", "cancelled_solves", "long_items", "solved_items")
SMC_MH_BATCH_MAX = 32
SMC_T_LOGLIK, SMC_T_MH, SMC_T_ESS, SMC_T_RESAMPLE, SMC_T_MOMENTS, SMC_T_MAX, SMC_T_SOLVE = range(7)
TIMING_NAMES = {SMC_T_LOGLIK: "loglik", SMC_T_MH: "mh", SMC_T_ESS: "ess", SMC_T_RESAMPLE: "resample",
                SMC_T_MOMENTS: "moments", SMC_T_MAX: "max", SMC_T_SOLVE: "solve"}


class SmcError(RuntimeError):
    """A call into libsmc_hip.so returned a non-zero status."""


c_dp = ctypes.POINTER(ctypes.c_double)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_ip = ctypes.POINTER(ctypes.c_int)
c_ctx = ctypes.c_void_p
i64, f64, cint, u64 = ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_uint64

# name -> (restype, argtypes); one entry per function of include/smc_hip.h
SIGNATURES = {
    "smc_abi_version": (cint, []),
    "smc_last_error": (ctypes.c_char_p, [c_ctx]),
    "smc_create": (cint, [ctypes.POINTER(c_ctx), cint, i64, i64, cint]),
    "smc_destroy": (None, [c_ctx]),
    "smc_synchronize": (cint, [c_ctx]),
    "smc_device_info": (cint, [c_ctx, ctypes.c_char_p, cint, ctypes.c_char_p, cint, c_ip]),
    "smc_set_model_mm": (cint, [c_ctx, c_dp, c_dp, c_dp, cint, cint, cint, f64, f64, f64]),
    "smc_set_model_methanation": (cint, [c_ctx, c_dp, c_dp, c_dp, cint, c_dp, c_ip, cint, f64, f64, f64, f64]),
    "smc_set_model_user": (cint, [c_ctx, ctypes.c_char_p, cint, c_dp, c_dp, c_dp, cint, cint, cint, cint, f64, f64, f64]),
    "smc_user_model_check": (cint, [ctypes.c_char_p, cint, cint, ctypes.c_char_p, cint]),
    "smc_user_model_dump_source": (cint, [ctypes.c_char_p, cint, cint, ctypes.c_char_p]),
    "smc_meth_sweep_counters": (cint, [c_ctx, c_i64p]),
    "smc_meth_sweep_check": (cint, [c_ctx, c_i64p]),
    "smc_meth_download_solves": (cint, [c_ctx, c_dp, ctypes.POINTER(ctypes.c_int32), i64]),
    "smc_set_prior": (cint, [c_ctx, c_ip, c_dp, c_dp, cint]),
    "smc_set_prior_mode": (cint, [c_ctx, cint]),
    "smc_set_resampling": (cint, [c_ctx, cint]),
    "smc_set_early_reject": (cint, [c_ctx, cint]),
    "smc_set_stiff_first": (cint, [c_ctx, cint]),
    "smc_set_cost_order": (cint, [c_ctx, cint]),
    "smc_set_fast_tail": (cint, [c_ctx, cint]),
    "smc_set_in_phase": (cint, [c_ctx, cint]),
    "smc_set_exact_pow": (cint, [c_ctx, cint]),
    "smc_exchange_plan": (cint, [cint, cint, i64, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p, c_i64p]),
    "smc_upload_particles": (cint, [c_ctx, cint, c_dp, i64]),
    "smc_download_particles": (cint, [c_ctx, cint, c_dp, i64]),
    "smc_upload_lk": (cint, [c_ctx, cint, c_dp, i64]),
    "smc_download_lk": (cint, [c_ctx, cint, c_dp, i64]),
    "smc_download_accept_flags": (cint, [c_ctx, c_u8p, i64]),
    "smc_download_item_info": (cint, [c_ctx, ctypes.POINTER(ctypes.c_int32), i64]),
    "smc_debug_set_order": (cint, [c_ctx, ctypes.POINTER(ctypes.c_int32), i64, cint]),
    "smc_commit_filt_to_pred": (cint, [c_ctx]),
    "smc_sample_prior_device": (cint, [c_ctx, u64, i64]),
    "smc_loglik": (cint, [c_ctx, cint, c_i64p, c_i64p]),
    "smc_mm_loglik_host": (cint, [c_ctx, c_dp, i64, c_dp, c_dp, c_i64p, c_i64p]),
    "smc_max_lk_local": (cint, [c_ctx, c_dp]),
    "smc_ess_partials": (cint, [c_ctx, f64, c_dp, cint, c_dp, c_dp]),
    "smc_max_lk_global": (cint, [c_ctx, c_dp]),
    "smc_ess_partials_global": (cint, [c_ctx, f64, c_dp, cint, c_dp, c_dp]),
    "smc_ess_search_global": (cint, [c_ctx, c_dp, cint, cint, c_dp, c_dp, c_dp]),
    "smc_resample_global": (cint, [c_ctx, f64, f64, f64, f64, cint, c_i64p, c_i64p]),
    "smc_resample_enqueue": (cint, [c_ctx, f64, f64, f64, f64, cint]),
    "smc_resample_result": (cint, [c_ctx, c_i64p, c_i64p]),
    "smc_pinned_alloc": (cint, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "smc_pinned_free": (cint, [ctypes.c_void_p]),
    "smc_mh_iteration_device_rng": (cint, [c_ctx, f64, f64, c_dp, u64, u64, i64, c_i64p, c_i64p, c_i64p, c_i64p, c_dp]),
    "smc_mh_sweeps_device_rng": (cint, [c_ctx, f64, f64, c_dp, u64, u64, cint, f64, f64, i64, c_ip, c_ip, c_dp, c_i64p, c_i64p,
                                        c_i64p, c_i64p, c_dp, c_dp, c_i64p]),
    "smc_mh_iteration_last_transform": (cint, [c_ctx, c_dp]),
    "smc_proposal_factor_device": (cint, [c_ctx, c_dp, c_dp, c_dp]),
    "smc_resample_phase1": (cint, [c_ctx, f64, f64, f64, c_dp, c_i64p]),
    "smc_resample_phase2": (cint, [c_ctx, f64, f64, f64, f64, f64, c_i64p]),
    "smc_download_offspring": (cint, [c_ctx, c_i64p, i64]),
    "smc_resample_phase3": (cint, [c_ctx, c_i64p, c_i64p, cint]),
    "smc_moment_sums_local": (cint, [c_ctx, c_dp]),
    "smc_moment_centered_local": (cint, [c_ctx, c_dp, c_dp]),
    "smc_mh_step_host_rng": (cint, [c_ctx, f64, f64, c_dp, c_dp, i64, c_i64p, c_i64p, c_i64p, c_i64p]),
    "smc_mh_step_device_rng": (cint, [c_ctx, f64, f64, c_dp, u64, u64, i64, c_i64p, c_i64p, c_i64p, c_i64p]),
    "smc_reset_accept_flags": (cint, [c_ctx]),
    "smc_set_debug_capture": (cint, [c_ctx, cint]),
    "smc_download_debug_proposals": (cint, [c_ctx, c_dp, c_dp, c_u8p, c_u8p, i64]),
    "smc_comm_get_unique_id": (cint, [c_u8p]),
    "smc_comm_init": (cint, [c_ctx, c_u8p, cint, cint]),
    "smc_comm_info": (cint, [c_ctx, c_ip, c_ip, c_ip]),
    "smc_comm_allreduce_sum_f64": (cint, [c_ctx, c_dp, cint]),
    "smc_comm_allreduce_max_f64": (cint, [c_ctx, c_dp, cint]),
    "smc_comm_allreduce_sum_i64": (cint, [c_ctx, c_i64p, cint]),
    "smc_comm_allgather_f64": (cint, [c_ctx, c_dp, cint, c_dp]),
    "smc_comm_allgather_i64": (cint, [c_ctx, c_i64p, cint, c_i64p]),
    "smc_comm_barrier": (cint, [c_ctx]),
    "smc_debug_rccl_self_exchange": (cint, [c_ctx, i64, i64, i64]),
    "smc_debug_set_local_peers": (cint, [c_ctx, ctypes.POINTER(c_ctx), cint, cint]),
    "smc_resample_phase3_pull": (cint, [c_ctx]),
    "smc_debug_peer_collectives": (cint, [c_ctx, cint]),
    "smc_timing_enable": (cint, [c_ctx, cint]),
    "smc_timing_reset": (cint, [c_ctx]),
    "smc_timing_get": (cint, [c_ctx, cint, c_i64p, c_dp]),
    "smc_work_totals": (cint, [c_ctx, c_i64p]),
    "smc_meth_last_error": (ctypes.c_char_p, []),
    "smc_meth_residual_host": (cint, [cint, c_dp, c_dp, c_dp, i64, c_dp]),
    "smc_meth_rate_host": (cint, [cint, c_dp, c_dp, i64, c_dp]),
    "smc_meth_loglike_host": (cint, [cint, c_dp, c_dp, c_dp, i64, cint, c_dp]),
    "smc_meth_dae_host": (cint, [cint, c_dp, c_dp, i64, f64, f64, f64, f64, f64, f64, c_dp, c_dp,
                                 ctypes.POINTER(ctypes.c_int32), c_i64p, c_dp]),
}

_LIB = None


def header_symbols(path: str = HEADER_PATH):
    """Function names declared in include/smc_hip.h (used by the CPU test that checks the exports)."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(smc_[a-z0-9_]+)\s*\(", txt)) - {"smc_ctx"})


MISSING = set()   # entry points an older A/B build lacks


def lib():
    """Load libsmc_hip.so (built by __graft_entry__.build() / csrc/Makefile). Raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SmcError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                           "(or __graft_entry__.build()); this package has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                # only an A/B build of an OLDER revision (SMC_HIP_LIB, tools/ab_rev.sh) may lack an entry point; a call
                # to it then raises.  The in-tree library must export everything (tests/test_abi.py).
                if not os.environ.get("SMC_HIP_LIB"):
                    raise
                MISSING.add(name)
                continue
            fn.restype = res
            fn.argtypes = args
        ver = L.smc_abi_version()
        # an A/B build of an older revision (SMC_HIP_LIB) is accepted from version 2 on (round 4: every entry point the default
        # driver path calls exists there; a call to a newer one raises through MISSING); version 1 lacks smc_resample_enqueue
        # and smc_mh_sweeps_device_rng, which run_smc uses by default, and is refused
        if ver != SMC_ABI_VERSION and not (os.environ.get("SMC_HIP_LIB") and 2 <= ver < SMC_ABI_VERSION):
            raise SmcError(f"libsmc_hip.so ABI version mismatch: library {ver}, binding {SMC_ABI_VERSION}")
        _LIB = L
    return _LIB


def check(ctx, status: int, what: str):
    if status != 0:
        msg = lib().smc_last_error(ctx)
        raise SmcError(f"{what}: {msg.decode() if msg else 'unknown error'}")
