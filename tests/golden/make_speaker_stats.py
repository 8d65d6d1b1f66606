#!/usr/bin/env python
"""Export the reference's per-speaker normalisation constants (core/datasets/speakers_stat.py:4-1492: SPEAKERS_STAT_121, 9 speakers, and
SPEAKERS_STAT_121_parted, 11 speakers; scale_factor, mean[242], std[242], float64) into ONE data file that ships with the package:

    speechdrivestemplates_amd/core/datasets/speakers_stat_121.npz     keys  <table>/<speaker>/{mean,std,scale_factor}

Only numbers are stored (a constants table is data, like the mean / std files of any dataset); the reference module is loaded straight
from its file, without its package (whose __init__ needs librosa).  Run in the authoring container:  python tests/golden/make_speaker_stats.py
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRC = "/root/reference/core/datasets/speakers_stat.py"
OUT = os.path.join(REPO, "speechdrivestemplates_amd", "core", "datasets", "speakers_stat_121.npz")


def main():
    spec = importlib.util.spec_from_file_location("_ref_speakers_stat", SRC)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for table in ("SPEAKERS_STAT_121", "SPEAKERS_STAT_121_parted"):
        for name, st in sorted(getattr(mod, table).items()):
            mean, std = np.asarray(st["mean"], dtype=np.float64), np.asarray(st["std"], dtype=np.float64)
            assert mean.shape == std.shape == (242,), (table, name, mean.shape, std.shape)
            out["%s/%s/mean" % (table, name)] = mean
            out["%s/%s/std" % (table, name)] = std
            out["%s/%s/scale_factor" % (table, name)] = np.float64(st["scale_factor"])
    np.savez_compressed(OUT, **out)
    names = {t: sorted({k.split("/")[1] for k in out if k.startswith(t + "/")}) for t in ("SPEAKERS_STAT_121", "SPEAKERS_STAT_121_parted")}
    print("wrote %s (%d bytes): %s" % (OUT, os.path.getsize(OUT), names))


if __name__ == "__main__":
    main()
