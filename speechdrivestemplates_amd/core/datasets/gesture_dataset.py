"""Dataset-side contract of the hot path (core/datasets/gesture_dataset.py): the sample dict the model
consumes (:107-119) and the pose transforms that run inside every train/test step (:147-236).

Round-1 scope: the transforms (torch, device-agnostic; the per-step fused float64 version is the
``sdt_final_metrics_f64`` kernel) and a seeded synthetic dataset with the reference's field layout.  The
on-disk csv/npz reader of the reference (:85-105,124-145) is a "next" row."""
import numpy as np
import torch
from torch.utils.data import Dataset

# keypoint indices in the 121-point layout, gesture_dataset.py:42-45
HAND_ROOT_L, HAND_ROOT_R, HEAD_ROOT = 6, 3, 39
_HEAD_IDX = [k for k in range(9, 79) if k != HEAD_ROOT]

SPEAKERS_STAT_121 = {}         # name -> {'scale_factor', 'mean'(242,), 'std'(242,)}   (global-relative poses)
SPEAKERS_STAT_121_parted = {}  # same for hierarchical ("parted") poses


def register_speaker_stat(name, parted=None, global_=None):
    """Make a speaker's normalisation constants available to ``get_speaker_stat`` (the reference hard-codes
    them in core/datasets/speakers_stat.py; here they are data supplied by the caller / a checkpoint)."""
    if parted is not None:
        SPEAKERS_STAT_121_parted[name] = parted
    if global_ is not None:
        SPEAKERS_STAT_121[name] = global_


class PoseTransforms:
    """normalize / denormalize / parted<->global / get_final_results with the reference semantics."""
    root_node, hand_root_l, hand_root_r, head_root = 1, HAND_ROOT_L, HAND_ROOT_R, HEAD_ROOT

    def _stat(self, t, kp):
        K = self.cfg.NUM_LANDMARKS
        if isinstance(t, np.ndarray):
            t = torch.tensor(t.astype(np.float64), dtype=torch.float32)  # torch.Tensor(ndarray) -> float32, :174-176
        t = t.to(kp.device)
        if t.dim() == 1:
            return t.reshape(1, 2, K)
        if t.dim() == 2:
            return t.reshape(kp.shape[0], 1, 2, K)
        raise NotImplementedError

    def normalize_poses(self, kp, speaker_stat):
        return (kp - self._stat(speaker_stat['mean'], kp)) / self._stat(speaker_stat['std'], kp)

    def denormalize_poses(self, kp, speaker_stat):
        return kp * self._stat(speaker_stat['std'], kp) + self._stat(speaker_stat['mean'], kp)

    def parted_to_global(self, poses):
        poses[..., :2, _HEAD_IDX] = poses[..., :2, _HEAD_IDX] + poses[..., :2, HEAD_ROOT, None]
        poses[..., :2, 79:100] = poses[..., :2, 79:100] + poses[..., :2, HAND_ROOT_L, None]
        poses[..., :2, 100:121] = poses[..., :2, 100:121] + poses[..., :2, HAND_ROOT_R, None]
        return poses

    def global_to_parted(self, poses):
        poses[..., :2, _HEAD_IDX] = poses[..., :2, _HEAD_IDX] - poses[..., :2, HEAD_ROOT, None]
        poses[..., :2, 79:100] = poses[..., :2, 79:100] - poses[..., :2, HAND_ROOT_L, None]
        poses[..., :2, 100:121] = poses[..., :2, 100:121] - poses[..., :2, HAND_ROOT_R, None]
        return poses

    def get_speaker_stat(self, speaker, num_kp, parted):
        table = SPEAKERS_STAT_121_parted if parted else SPEAKERS_STAT_121
        if num_kp != 121 or speaker not in table:
            raise KeyError('no %s statistics registered for speaker %r (see register_speaker_stat)'
                           % ('parted' if parted else 'global', speaker))
        return table[speaker]

    def get_final_results(self, poses, speaker_stat):
        poses = self.denormalize_poses(poses, speaker_stat)
        if self.cfg.HIERARCHICAL_POSE:
            poses = self.parted_to_global(poses)
        scale = speaker_stat['scale_factor'].to(poses.device)
        return poses * scale.reshape(scale.shape[0], 1, 1, -1)

    def transform_normalized_parted2global(self, poses, speaker):
        """gesture_dataset.py:222-236 (assumes one speaker per batch, like the reference)."""
        stat_g = self.get_speaker_stat(speaker[0], poses.shape[-1], False)
        stat_p = self.get_speaker_stat(speaker[0], poses.shape[-1], True)
        poses = self.parted_to_global(self.denormalize_poses(poses, stat_p))
        return self.normalize_poses(poses, stat_g)


class GestureDataset(PoseTransforms, Dataset):
    """Transform-only stand-in with the reference's class name: it carries cfg.DATASET and the pose
    transforms; reading processed_137.csv / clip npz files is not implemented in this round."""

    def __init__(self, root_dir=None, speaker=None, split='train', cfg=None, demo_input=None):
        self.cfg = cfg.DATASET
        self.speaker, self.split = speaker, split
        if root_dir is not None:
            raise NotImplementedError('on-disk GestureDataset reading is a next-round item; use SyntheticGestureDataset')

    def __len__(self):
        return 0


class SyntheticGestureDataset(PoseTransforms, Dataset):
    """Seeded synthetic clips with the field layout of GestureDataset.__getitem__ (gesture_dataset.py:107-119):
    audio 0.1*N(0,1) of 68266 samples, normalised poses N(0,1) (64,2,121), per-clip float64 statistics."""

    def __init__(self, root_dir=None, speaker='synthetic', split='train', cfg=None, demo_input=None, num_clips=None, seed=1):
        self.cfg = cfg.DATASET
        self.speaker, self.split, self.seed = speaker or 'synthetic', split, seed
        self.num_clips = int(num_clips if num_clips is not None else getattr(self.cfg, 'SYNTHETIC_CLIPS', 4096))
        if self.speaker not in SPEAKERS_STAT_121:  # s2g's parted->global re-normalisation looks statistics up by name
            rng = np.random.Generator(np.random.PCG64([seed, 7]))
            K2 = 2 * self.cfg.NUM_LANDMARKS
            register_speaker_stat(self.speaker,
                                  parted={'scale_factor': 1.0, 'mean': rng.standard_normal(K2) * 20.0, 'std': rng.uniform(2.0, 30.0, K2)},
                                  global_={'scale_factor': 1.0, 'mean': rng.standard_normal(K2) * 60.0, 'std': rng.uniform(5.0, 80.0, K2)})
        per_frame = self.cfg.AUDIO_SR / self.cfg.FPS  # parse_audio_length, audio_processing.py:5-11
        self.num_frames = int(self.cfg.AUDIO_LENGTH / per_frame)
        self.audio_length = int(self.num_frames * per_frame)

    def __len__(self):
        return self.num_clips

    def __getitem__(self, idx):
        K = self.cfg.NUM_LANDMARKS
        rng = np.random.Generator(np.random.PCG64([self.seed, 0 if self.split == 'train' else 1, int(idx)]))
        audio = (0.1 * rng.standard_normal(self.audio_length)).astype(np.float32)
        poses = rng.standard_normal((self.num_frames, 2, K)).astype(np.float32)
        score = rng.uniform(0, 1, (self.num_frames, 1, K)).astype(np.float32)
        stat = {'scale_factor': float(rng.uniform(0.8, 1.3)), 'mean': rng.standard_normal(2 * K) * 20.0,
                'std': rng.uniform(2.0, 30.0, 2 * K)}
        return {'speaker': self.speaker, 'audio': audio, 'num_frames': self.num_frames, 'clip_index': idx,
                'poses': torch.from_numpy(poses), 'poses_score': torch.from_numpy(np.repeat(score, 2, axis=1)),
                'speaker_stat': stat,
                'anchors': {'hand_root_l': HAND_ROOT_L, 'hand_root_r': HAND_ROOT_R, 'head_root': HEAD_ROOT}}
