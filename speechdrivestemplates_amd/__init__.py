"""MI355X-native SDT voice2pose training hot path (HIP kernels behind the reference core.networks / core.pipelines API)."""
import os as _os

# HIP maps streams onto a small pool of hardware queues (default 4, in creation order).  A data-parallel process creates more streams than that
# (communication stream, RCCL's own, capture streams) and the weight-gradient side stream then SHARES a hardware queue with the main stream: the two
# serialise and the step loses the overlap it has on one GPU (measured under a 1-rank RCCL group: 6.43 ms per step against 5.62 plain; with 8 queues
# 5.82).  Read by the HIP runtime at its initialisation, i.e. at the first torch.cuda call: importing this package first is enough; a launcher may
# also export it.
import sys as _sys

# True when the default below came too late to take effect: HIP was already initialised (a torch.cuda call before this import) and the variable was
# not exported -- the runtime then runs with its own default of 4 queues.  dp.GradReducer warns about it (it costs a data-parallel step ~8 %,
# profiles/r05_dp_one_gpu_ab.txt); nothing is wrong on a single GPU.
HW_QUEUES_SET_TOO_LATE = False
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _t = _sys.modules.get("torch")
    try:
        HW_QUEUES_SET_TOO_LATE = bool(_t is not None and _t.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        HW_QUEUES_SET_TOO_LATE = False
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"


def hw_queues():
    """(value of GPU_MAX_HW_QUEUES this process runs with as far as this package can tell, whether the package's default came too late to count)"""
    return _os.environ.get("GPU_MAX_HW_QUEUES"), HW_QUEUES_SET_TOO_LATE
