"""MI355X-native engine for the particle inner loop of adaptive likelihood-tempered SMC.

The directory name contains hyphens (it mirrors the upstream repository's name), so the package is
loaded by path: `__graft_entry__.load_package()` registers it as the module `smc_lt_amd`.

  binding   ctypes declarations of include/smc_hip.h
  engine    HipEngine: one libsmc_hip.so context (device-resident particle sets + stages)
  comm      SingleComm / RcclComm
  driver    SMCSettings, run_smc: the tempering loop of the reference's driver scripts
  dropin/   shadow modules with the reference's names (Micmem_settings, Micmem_likelihood)
"""
from .binding import (SMC_SET_FILT, SMC_SET_PRED, SmcError, header_symbols, lib, LIB_PATH)  # noqa: F401
from .comm import RcclComm, SingleComm  # noqa: F401
from .driver import SMCSettings, ess_candidates, ess_search, mvn_transform, proposal_cov, resample, run_smc, sample_prior  # noqa: F401
from .engine import HipEngine, release_pinned_pool  # noqa: F401
from . import methanation  # noqa: F401
from . import datagen  # noqa: F401
from . import user_models  # noqa: F401
