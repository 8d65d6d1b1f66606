#!/usr/bin/env python
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for conv_taps_kernel's own access pattern (MI355X_MICROARCH.md: "other
access widths are uncalibrated -- calibrate on a known byte count").  Launches convolutions whose minimum traffic is known:

  k1   1x1, 128 -> 64 channels: one n-tile, one tap -> every input byte is needed exactly once
  k3   3x3, 128 -> 64: one n-tile, 9 taps -> re-reads exist but can all hit in the L2
  k3n4 3x3, 128 -> 256: 4 n-tiles x 9 taps (layer L4 of the audio encoder)
  k4s2 4x4 stride 2, 64 -> 64 on the 80x427 map (layer L1, 280 MB of input)

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/cal_f -o c -- python tools/fetch_calibration.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/cal_w -o c -- python tools/fetch_calibration.py
    python tools/pmc_summary.py gpurun_out/cal_f/c_counter_collection.csv --match conv_ --per-dispatch

--encoder: instead, one forward + backward launch set of every audio-encoder layer L1..L7 at B=32 (per dispatch: forward,
then the input-gradient parity classes and the weight gradient), to compare each launch with its algorithmic bytes.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from speechdrivestemplates_amd import ops  # noqa: E402

B = 32
if "--encoder" in sys.argv:
    #        Cin  H    W   Cout k       s  p
    LAYERS = ((64, 80, 427, 64, (4, 4), 2, 1), (64, 40, 213, 128, (3, 3), 1, 1), (128, 40, 213, 128, (4, 4), 2, 1),
              (128, 20, 106, 256, (3, 3), 1, 1), (256, 20, 106, 256, (4, 4), 2, 1), (256, 10, 53, 256, (3, 3), 1, 1),
              (256, 10, 53, 256, (6, 3), 1, 0))
    for li, (Cin, H, W, Cout, k, s, p) in enumerate(LAYERS, 1):
        x = torch.randn(B, H, W, Cin, device="cuda", requires_grad=True)
        w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(Cout, Cin, *k, device="cuda") * 0.05))
        y = ops.ConvFn.apply(x, w, None, s, p)
        y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
        print("L%d  X %.1f MB  W %.2f MB  Y %.1f MB" % (li, x.numel() * 4 / 1e6, w.numel() * 4 / 1e6, y.numel() * 4 / 1e6))
    sys.exit(0)
CASES = (("k1", (40, 213, 128), 64, 1, 1, 0), ("k3", (40, 213, 128), 64, 3, 1, 1), ("k3n4", (20, 106, 128), 256, 3, 1, 1),
         ("k4s2", (80, 427, 64), 64, 4, 2, 1))
for name, (H, W, Cin), Cout, k, s, p in CASES:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.nn.Parameter(ops.to_weight_layout(torch.randn(Cout, Cin, k, k, device="cuda") * 0.05))
    for _ in range(3):
        y = ops.ConvFn.apply(x, w, None, s, p)
    torch.cuda.synchronize()
    print("%-5s X %.1f MB  W %.2f MB  Y %.1f MB  grid %d" % (name, x.numel() * 4 / 1e6, w.numel() * 4 / 1e6, y.numel() * 4 / 1e6,
                                                          -(-y.numel() // Cout // 64) * -(-Cout // 64)))
