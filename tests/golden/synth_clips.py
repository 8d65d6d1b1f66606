"""Seeded synthetic speaker directory in the reference's on-disk format (data_preprocess/3_1_generate_clips.py:140-141,190;
gesture_dataset.py:85-92): processed_137.csv + one npz per clip with pose (frames,3,137) in pixels and raw audio."""
import os

import numpy as np
import pandas as pd


def write_synthetic_speaker(root, speaker="oliver", n=5, seed=11):
    rng = np.random.Generator(np.random.PCG64(seed))
    d = os.path.join(root, speaker)
    os.makedirs(os.path.join(d, "clips"), exist_ok=True)
    rows = []
    for i in range(n):
        frames = [64, 70, 64, 66, 64][i % 5]
        pose = np.empty((frames, 3, 137), dtype=np.float64)
        pose[:, 0] = 640 + 150 * rng.standard_normal((frames, 137))
        pose[:, 1] = 360 + 120 * rng.standard_normal((frames, 137))
        pose[:, 2] = rng.uniform(0, 1, (frames, 137))
        audio = (0.1 * rng.standard_normal([68266, 70000, 60000, 68267, 68266][i % 5])).astype(np.float32)
        fn = "clips/%05d.npz" % i
        np.savez(os.path.join(d, fn), pose=pose, audio=audio, imgs=np.array(["f%03d.jpg" % k for k in range(frames)]))
        rows.append({"dataset": "train" if i % 4 else "dev", "start": i * 4.0, "end": i * 4.0 + 4.27, "interval_id": "iv%d" % i,
                     "pose_fn": fn, "audio_fn": "a%d.wav" % i, "video_fn": "v.mp4", "speaker": speaker})
    pd.DataFrame(rows).to_csv(os.path.join(d, "processed_137.csv"), index=False)
    return d
