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


@pytest.mark.gpu
@pytest.mark.parametrize("hier", [True, False])
def test_device_clip_store_matches_reference_dataset(tmp_path, hier):
    """Row f-3, device side: batches assembled on the GPU from an HBM-resident clip store are bit-identical to the
    reference GestureDataset's samples (the committed fixture) and to the host reader + default_collate."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import get_dataset
    from speechdrivestemplates_amd.core.datasets.gesture_dataset import DeviceClipStore, load_speaker_stats
    g = dict(np.load(os.path.join(GOLDEN, "dataset_clips.npz")))
    write_synthetic_speaker(str(tmp_path), "oliver", n=5, seed=11)
    load_speaker_stats(os.path.join(GOLDEN, "speaker_stat_oliver.npz"), "oliver")
    cfg = get_cfg_defaults()
    cfg.merge_from_list(["DATASET.HIERARCHICAL_POSE", hier])
    ds = get_dataset("GestureDataset")(str(tmp_path), "oliver", "train", cfg)
    store = DeviceClipStore(ds)
    assert len(store) == len(ds)
    order = list(reversed(range(len(ds)))) + [0, 0]  # permuted, with a repeated clip
    got = store.batch(order)
    want = torch.utils.data.default_collate([ds[i] for i in order])
    for k in ("poses", "poses_score", "audio"):
        assert got[k].is_cuda and got[k].dtype == torch.float32
        assert torch.equal(got[k].cpu(), want[k].float()), k
    assert torch.equal(got["clip_index"].cpu(), want["clip_index"]) and torch.equal(got["num_frames"], want["num_frames"])
    for k in ("mean", "std", "scale_factor"):
        assert got["speaker_stat"][k].dtype == torch.float64
        assert torch.equal(got["speaker_stat"][k].cpu(), want["speaker_stat"][k]), k
    assert got["speaker"] == want["speaker"]
    for j, i in enumerate(order):  # and directly against what the reference's own dataset class produced
        np.testing.assert_array_equal(got["poses"][j].cpu().numpy(), g["%s/train/%d/poses" % (hier, i)])
    with pytest.raises(IndexError):
        store.batch([len(ds)])


@pytest.mark.gpu
def test_train_step_from_device_clip_store(tmp_path):
    """A batch assembled by DeviceClipStore drives the train step exactly like the host-collated one."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import get_dataset
    from speechdrivestemplates_amd.core.datasets.gesture_dataset import DeviceClipStore, load_speaker_stats
    from speechdrivestemplates_amd.core.pipelines import get_pipeline
    write_synthetic_speaker(str(tmp_path), "oliver", n=5, seed=11)
    load_speaker_stats(os.path.join(GOLDEN, "speaker_stat_oliver.npz"), "oliver")
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLDEN), "..", "configs", "voice2pose_sdt_bp.yaml"))
    cfg.merge_from_list(["SYS.LOG_INTERVAL", 10 ** 9])
    ds = get_dataset("GestureDataset")(str(tmp_path), "oliver", "train", cfg)
    store = DeviceClipStore(ds)
    out = []
    for src in ("host", "device"):
        torch.manual_seed(0)
        pipe = get_pipeline(cfg.PIPELINE_TYPE)(cfg)
        pipe.num_train_samples, pipe.train_dataset = len(ds), ds
        pipe.setup_model(cfg)
        pipe.setup_optimizer()
        pipe.model.train()
        idx = [2, 0, 1]
        batch = store.batch(idx) if src == "device" else torch.utils.data.default_collate([ds[i] for i in idx])
        losses, results = pipe.forward_backward(batch)
        pipe.optimizer_updates(losses)
        torch.cuda.synchronize()
        out.append((losses["G_reg_loss"].detach().clone(), losses["L2_dist"].detach().clone(), results["poses_pred_batch"].detach().clone(),
                    losses["lip_sync_error_n"].detach().clone()))
    (reg_a, l2_a, pred_a, lip_a), (reg_b, l2_b, pred_b, lip_b) = out
    assert torch.equal(reg_a, reg_b) and torch.equal(pred_a, pred_b)
    assert abs(float(l2_a) - float(l2_b)) <= 1e-12 * abs(float(l2_a)), (float(l2_a), float(l2_b))  # float64 atomics: summation order only
    # the store's statistics are expand()ed views: the metrics kernel once read them through pointers of freed contiguous temporaries
    assert abs(float(lip_a) - float(lip_b)) <= 1e-12 * abs(float(lip_a)), (float(lip_a), float(lip_b))


def test_demo_split_reads_wav(tmp_path):
    """Row f-4, host side: the demo split turns wav files into the sample dict of gesture_dataset.py:54-79 (mono float32 at
    16 kHz, cropped to a whole number of 1/15 s frames, MAX_DEMO_LENGTH honoured)."""
    from scipy.io import wavfile
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import get_dataset
    from speechdrivestemplates_amd.core.datasets.gesture_dataset import load_speaker_stats
    load_speaker_stats(os.path.join(GOLDEN, "speaker_stat_oliver.npz"), "oliver")
    rng = np.random.default_rng(5)
    a16 = (rng.standard_normal(16000 * 3 + 123) * 3000).astype(np.int16)           # 3.0077 s mono 16 kHz
    a22 = (rng.standard_normal((22050 * 2, 2)) * 3000).astype(np.int16)             # 2 s stereo 22.05 kHz
    long = (rng.standard_normal(16000 * 30) * 3000).astype(np.int16)                # 30 s > MAX_DEMO_LENGTH = 24 s
    for name, sr, a in (("a.wav", 16000, a16), ("b.wav", 22050, a22), ("c.wav", 16000, long)):
        wavfile.write(str(tmp_path / name), sr, a)
    cfg = get_cfg_defaults()
    ds = get_dataset("GestureDataset")("unused_root", "oliver", "demo", cfg,
                                       demo_input=" ".join(str(tmp_path / n) for n in ("a.wav", "b.wav", "c.wav")))
    assert len(ds) == 3
    s = ds[0]
    assert s["num_frames"] == int(len(a16) / (16000 / 15)) == 45 and len(s["audio"]) == int(45 * 16000 / 15) == 48000
    np.testing.assert_array_equal(s["audio"], a16[:48000].astype(np.float32) / 32768.0)  # what librosa.load returns at 16 kHz
    assert s["audio"].dtype == np.float32 and s["speaker"] == "oliver" and set(s["speaker_stat"]) == {"mean", "std", "scale_factor"}
    s = ds[1]
    # 2 s resampled to 32000 samples; parse_audio_length's float arithmetic gives int(32000 / (16000 / 15)) = 29 frames
    assert s["num_frames"] == int(32000 / (16000 / 15)) == 29 and len(s["audio"]) == int(29 * (16000 / 15)) and np.abs(s["audio"]).max() < 1.0
    s = ds[2]
    assert s["num_frames"] == 360 and len(s["audio"]) == 384000  # cropped to 24 s
    d = get_dataset("GestureDataset")("unused_root", "oliver", "demo", cfg, demo_input=str(tmp_path))
    assert len(d) == 1  # a directory: DEMO.NUM_SAMPLES (=1) wav files of it
    with pytest.raises(NotImplementedError):
        get_dataset("GestureDataset")("unused_root", "oliver", "demo", cfg, demo_input="x.m4a")[0]


def test_builtin_speaker_statistics_cover_the_reference_table(tmp_path):
    """All speakers of the reference's speakers_stat.py (9 global-relative, 11 hierarchical) ship as data and load lazily; the shipped oliver
    entry equals the fixture the other tests use; BASELINE config 5's speaker (kubinec) reads a clip file through them; where the reference
    sources are present the whole table is compared bit for bit."""
    from speechdrivestemplates_amd.config import get_cfg_defaults
    from speechdrivestemplates_amd.core.datasets import gesture_dataset as gd
    saved = (dict(gd.SPEAKERS_STAT_121), dict(gd.SPEAKERS_STAT_121_parted), gd._BUILTIN_LOADED[0])
    try:
        gd.SPEAKERS_STAT_121.clear()
        gd.SPEAKERS_STAT_121_parted.clear()
        gd._BUILTIN_LOADED[0] = False
        cfg = get_cfg_defaults()
        write_synthetic_speaker(str(tmp_path), "kubinec", n=3, seed=5)
        ds = gd.GestureDataset(str(tmp_path), "kubinec", "train", cfg)  # no registration by the caller: the built-in table serves it
        s = ds[1]
        assert s["speaker"] == "kubinec" and s["speaker_stat"]["mean"].shape == (242,) and s["poses"].shape == (64, 2, 121)
        assert sorted(gd.SPEAKERS_STAT_121) == ['almaram', 'conan', 'ellen', 'jon', 'kubinec', 'luo', 'oliver', 'shelly', 'xing']
        assert sorted(gd.SPEAKERS_STAT_121_parted) == ['almaram', 'angelica', 'conan', 'ellen', 'jon', 'kubinec', 'luo', 'oliver', 'seth', 'shelly', 'xing']
        sp = np.load(os.path.join(GOLDEN, "speaker_stat_oliver.npz"))
        assert np.array_equal(gd.SPEAKERS_STAT_121_parted["oliver"]["mean"], sp["parted_mean"])
        assert np.array_equal(gd.SPEAKERS_STAT_121["oliver"]["std"], sp["global_std"])
        assert gd.SPEAKERS_STAT_121_parted["oliver"]["scale_factor"] == float(sp["parted_scale"])
        ref = "/root/reference/core/datasets/speakers_stat.py"
        if os.path.exists(ref):
            import importlib.util
            spec = importlib.util.spec_from_file_location("_ref_speakers_stat_check", ref)
            mod = importlib.util.module_from_spec(spec)
            sys.dont_write_bytecode, prev = True, sys.dont_write_bytecode
            try:
                spec.loader.exec_module(mod)
            finally:
                sys.dont_write_bytecode = prev
            for table, mine in (("SPEAKERS_STAT_121", gd.SPEAKERS_STAT_121), ("SPEAKERS_STAT_121_parted", gd.SPEAKERS_STAT_121_parted)):
                theirs = getattr(mod, table)
                assert sorted(theirs) == sorted(mine)
                for name, st in theirs.items():
                    assert np.array_equal(np.asarray(st["mean"], dtype=np.float64), mine[name]["mean"]), (table, name)
                    assert np.array_equal(np.asarray(st["std"], dtype=np.float64), mine[name]["std"]), (table, name)
                    assert float(st["scale_factor"]) == mine[name]["scale_factor"], (table, name)
    finally:
        gd.SPEAKERS_STAT_121.clear()
        gd.SPEAKERS_STAT_121.update(saved[0])
        gd.SPEAKERS_STAT_121_parted.clear()
        gd.SPEAKERS_STAT_121_parted.update(saved[1])
        gd._BUILTIN_LOADED[0] = saved[2]
