"""Golden vectors for the on-disk dataset path: the REFERENCE's GestureDataset (imported from /root/reference under the
same shim as make_golden.py) reads a seeded synthetic speaker directory; only its outputs are stored."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from make_golden import install_shim  # noqa: E402
from synth_clips import write_synthetic_speaker  # noqa: E402


def main():
    install_shim()
    from core.datasets.gesture_dataset import GestureDataset
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_synthetic_speaker(root, "oliver", n=5, seed=11)
        for hier in (True, False):
            cfg = types.SimpleNamespace(DATASET=types.SimpleNamespace(
                SUBSET=None, CACHING=False, AUDIO_LENGTH=68267, AUDIO_SR=16000, FPS=15, NUM_FRAMES=64, NUM_LANDMARKS=121,
                HIERARCHICAL_POSE=hier, MAX_DEMO_LENGTH=24))
            for split in ("train", "val"):
                ds = GestureDataset(root, "oliver", split, cfg)
                out["%s/%s/len" % (hier, split)] = np.array(len(ds))
                for i in range(len(ds)):
                    s = ds[i]
                    tag = "%s/%s/%d" % (hier, split, i)
                    out[tag + "/poses"] = s["poses"].numpy()
                    out[tag + "/score_sum"] = np.array(float(s["poses_score"].double().sum()))
                    out[tag + "/audio_head"] = np.asarray(s["audio"][:64])
                    out[tag + "/audio_len"] = np.array(len(s["audio"]))
                    out[tag + "/audio_sum"] = np.array(float(np.asarray(s["audio"], dtype=np.float64).sum()))
                    out[tag + "/num_frames"] = np.array(s["num_frames"])
                    out[tag + "/clip_index"] = np.array(s["clip_index"])
    np.savez_compressed(os.path.join(HERE, "dataset_clips.npz"), **out)
    print("dataset_clips.npz", os.path.getsize(os.path.join(HERE, "dataset_clips.npz")) // 1024, "KiB", sorted(out)[:6])


if __name__ == "__main__":
    main()
