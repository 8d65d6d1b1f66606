"""MI355X-native SDT voice2pose training hot path (HIP kernels behind the reference core.networks / core.pipelines API)."""
import os as _os

# HIP maps streams onto a small pool of hardware queues (default 4, in creation order).  A data-parallel process creates more streams than that
# (communication stream, RCCL's own, capture streams) and the weight-gradient side stream then SHARES a hardware queue with the main stream: the two
# serialise and the step loses the overlap it has on one GPU (measured under a 1-rank RCCL group: 6.43 ms per step against 5.62 plain; with 8 queues
# 5.82).  Read by the HIP runtime at its initialisation, i.e. at the first torch.cuda call: importing this package first is enough; a launcher may
# also export it.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
