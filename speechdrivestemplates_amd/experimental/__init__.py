"""Measured experiments that LOSE to the default path and are therefore not part of the product library:

  stage1d  : the generator's Conv1d stage as one launch per layer and direction (csrc/conv1d.hip)     -2 % end to end
  presplit : the bf16x6 pre-split operand pipeline (csrc/presplit.hip, switched by ops.PRESPLIT)        4060 vs 4445 clips/s
  graph    : hipGraph replay of a whole train step                                                      replay == eager

Their kernels are compiled only into the -DSDT_TUNING library (`python __graft_entry__.py --tuning`, loaded with
SDT_HIP_LIB=.../libsdt_hip_tuning.so by the tools and by the tests that cover them); `require()` raises when the loaded library
does not carry them.  Nothing here is switched by environment variables."""
from .. import _lib


def available():
    return _lib.has_experimental()


def require(what):
    if not available():
        raise RuntimeError("%s needs the -DSDT_TUNING library (python __graft_entry__.py --tuning; SDT_HIP_LIB=.../libsdt_hip_tuning.so): "
                           "the product library does not carry the experimental kernels" % what)
