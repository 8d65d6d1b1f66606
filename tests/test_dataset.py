"""Row f-3: the on-disk GestureDataset reader against outputs of the reference's own GestureDataset on the same seeded
synthetic clip files (tests/golden/make_dataset_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from synth_clips import write_synthetic_speaker  # noqa: E402


@pytest.mark.parametrize("hier", [True, False])
def test_reader_matches_reference(tmp_path, hier):
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import get_dataset
    from speechdrivestemplates_amd.core.datasets.gesture_dataset import load_speaker_stats
    g = dict(np.load(os.path.join(GOLDEN, "dataset_clips.npz")))
    write_synthetic_speaker(str(tmp_path), "oliver", n=5, seed=11)
    load_speaker_stats(os.path.join(GOLDEN, "speaker_stat_oliver.npz"), "oliver")
    cfg = get_cfg_defaults()
    cfg.merge_from_list(["DATASET.HIERARCHICAL_POSE", hier])
    for split in ("train", "val"):
        ds = get_dataset("GestureDataset")(str(tmp_path), "oliver", split, cfg)
        assert len(ds) == int(g["%s/%s/len" % (hier, split)])
        for i in range(len(ds)):
            s = ds[i]
            tag = "%s/%s/%d" % (hier, split, i)
            assert s["poses"].shape == (64, 2, 121) and s["poses"].dtype == torch.float32
            np.testing.assert_array_equal(s["poses"].numpy(), g[tag + "/poses"])  # bit-exact: same fp32 op order
            assert float(s["poses_score"].double().sum()) == float(g[tag + "/score_sum"])
            assert len(s["audio"]) == int(g[tag + "/audio_len"]) == 68266
            np.testing.assert_array_equal(np.asarray(s["audio"][:64]), g[tag + "/audio_head"])
            assert float(np.asarray(s["audio"], dtype=np.float64).sum()) == float(g[tag + "/audio_sum"])
            assert s["num_frames"] == int(g[tag + "/num_frames"]) == 64 and s["clip_index"] == int(g[tag + "/clip_index"])
    # collated batch has the layout the model consumes
    batch = torch.utils.data.default_collate([ds[0]])
    assert batch["speaker_stat"]["mean"].dtype == torch.float64 and batch["poses"].shape == (1, 64, 2, 121)


def test_missing_csv_and_unknown_speaker(tmp_path):
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import get_dataset
    with pytest.raises(FileNotFoundError):
        get_dataset("GestureDataset")(str(tmp_path), "nobody", "train", get_cfg_defaults())
    with pytest.raises(KeyError):
        get_dataset("NoSuchDataset")
