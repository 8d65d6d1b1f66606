"""Dataset-side contract of the hot path (core/datasets/gesture_dataset.py): the sample dict the model
consumes (:107-119) and the pose transforms that run inside every train/test step (:147-236).

Round-1 scope: the transforms (torch, device-agnostic; the per-step fused float64 version is the
``sdt_final_metrics_f64`` kernel) and a seeded synthetic dataset with the reference's field layout.  The
on-disk csv/npz reader of the reference (:85-105,124-145) is a "next" row."""
import os

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
    them in core/datasets/speakers_stat.py; here they are data supplied by the caller / a checkpoint).
    Replacing a registered table keeps the DEVICE copies of the old one alive and refreshes them in place with the new values: a captured
    hipGraph reads them by raw pointer, so they are never freed or re-allocated behind it and a replay sees the new statistics."""
    for table, new in ((SPEAKERS_STAT_121_parted, parted), (SPEAKERS_STAT_121, global_)):
        if new is None:
            continue
        old = table.get(name)
        table[name] = new
        if old is not None and old is not new:
            PoseTransforms._rehome(old, new)


class PoseTransforms:
    """normalize / denormalize / parted<->global / get_final_results with the reference semantics."""
    root_node, hand_root_l, hand_root_r, head_root = 1, HAND_ROOT_L, HAND_ROOT_R, HEAD_ROOT

    # (id(ndarray), device) -> [ndarray kept alive, fp32 device tensor, the bytes that were uploaded]: registered statistics are uploaded once, not
    # per step.  The device tensor of an entry is NEVER freed or replaced (a captured hipGraph reads it by raw pointer: the caching allocator
    # would hand the memory to somebody else and every replay would normalise with garbage, silently -- ADVICE r5); an array whose bytes changed
    # (in-place edit of a registered mean / std: a 2 KB compare per call) is copied INTO the same tensor, which also makes the edit visible to
    # replays.  The table only grows; past _STAT_CACHE_MAX distinct arrays further ones are uploaded per call (correct, slower, not capturable).
    _STAT_ON_DEVICE = {}
    _STAT_CACHE_MAX = 256

    @staticmethod
    def _no_capture(what):
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("speaker statistics %s inside a hipGraph capture" % what)

    @staticmethod
    def _rehome(old, new):
        """register_speaker_stat replaced table ``old`` by ``new``: the device copies of old's arrays now carry new's values (same memory)"""
        for k in ('mean', 'std'):
            o, n = old.get(k), new.get(k)
            if not (isinstance(o, np.ndarray) and isinstance(n, np.ndarray)) or o is n:
                continue
            for key in [kk for kk in PoseTransforms._STAT_ON_DEVICE if kk[0] == id(o)]:
                ent = PoseTransforms._STAT_ON_DEVICE[key]
                if ent[0] is not o or ent[1].numel() != n.size or (id(n), key[1]) in PoseTransforms._STAT_ON_DEVICE:
                    continue  # (the old entry stays: its tensor must not be freed)
                PoseTransforms._no_capture("were replaced")
                ent[1].copy_(torch.tensor(n.astype(np.float64), dtype=torch.float32).reshape(ent[1].shape))
                PoseTransforms._STAT_ON_DEVICE[(id(n), key[1])] = [n, ent[1], n.tobytes()]
                ent[2] = None  # the old array no longer describes the tensor: a later call with it re-uploads (into the same tensor again)

    def _stat(self, t, kp):
        K = self.cfg.NUM_LANDMARKS
        if isinstance(t, np.ndarray):
            key = (id(t), str(kp.device))
            hit = PoseTransforms._STAT_ON_DEVICE.get(key)
            raw = t.tobytes()
            if hit is not None and hit[0] is t and hit[2] != raw:  # edited in place (or re-homed away): refresh the SAME device tensor
                self._no_capture("changed")
                hit[1].copy_(torch.tensor(t.astype(np.float64), dtype=torch.float32).reshape(hit[1].shape))
                hit[2] = raw
            elif hit is None or hit[0] is not t:
                self._no_capture("were never uploaded")
                # torch.Tensor(ndarray) -> float32, :174-176.  (A host-to-device copy per call also made the step un-capturable in a hipGraph.)
                dev_t = torch.tensor(t.astype(np.float64), dtype=torch.float32).to(kp.device)
                hit = [t, dev_t, raw]
                if len(PoseTransforms._STAT_ON_DEVICE) < PoseTransforms._STAT_CACHE_MAX:
                    PoseTransforms._STAT_ON_DEVICE[key] = hit
            t = hit[1]
        else:
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

    _PART_INDEX = {}  # device -> (keypoints that hang off a part root, the root of each): built once, so that the transforms below copy no
                      # index list to the device per call (which also kept the s2g train step from being captured into a hipGraph)

    @staticmethod
    def _part_index(device):
        key = str(device)
        if key not in PoseTransforms._PART_INDEX:
            sel = list(_HEAD_IDX) + list(range(79, 100)) + list(range(100, 121))
            root = [HEAD_ROOT] * len(_HEAD_IDX) + [HAND_ROOT_L] * 21 + [HAND_ROOT_R] * 21
            PoseTransforms._PART_INDEX[key] = (torch.tensor(sel, dtype=torch.int64, device=device), torch.tensor(root, dtype=torch.int64, device=device))
        return PoseTransforms._PART_INDEX[key]

    def parted_to_global(self, poses):
        """gesture_dataset.py:147-155, in place: head / hand keypoints += their part's root (the roots themselves are in no part)."""
        sel, root = self._part_index(poses.device)
        xy = poses[..., :2, :]
        xy.index_add_(-1, sel, xy.index_select(-1, root))
        return poses

    def global_to_parted(self, poses):
        sel, root = self._part_index(poses.device)
        xy = poses[..., :2, :]
        xy.index_add_(-1, sel, xy.index_select(-1, root), alpha=-1)
        return poses

    def get_speaker_stat(self, speaker, num_kp, parted):
        table = SPEAKERS_STAT_121_parted if parted else SPEAKERS_STAT_121
        if speaker not in table:
            load_builtin_speaker_stats()  # the reference's own speakers (speakers_stat.py:4-1492), shipped as data
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


_KEEP_137 = list(range(0, 8)) + [15, 16] + list(range(25, 137))  # OpenPose-137 -> 122: drop lower body (:131-136)
_DROP_ROOT = [0] + list(range(2, 122))                            # 122 -> 121: remove the root joint (:138-145)


def crop_pad_audio(wav, audio_length):
    """core/utils/audio_processing.py:14-19."""
    if len(wav) > audio_length:
        return wav[:audio_length]
    if len(wav) < audio_length:
        return np.pad(wav, [0, audio_length - len(wav)], mode='constant', constant_values=0)
    return wav


def parse_audio_length(audio_length, sr, fps):
    """core/utils/audio_processing.py:5-11 -- (68267, 16000, 15) -> (68266, 64)."""
    per_frame = sr / fps
    num_frames = int(audio_length / per_frame)
    return int(num_frames * per_frame), num_frames


class GestureDataset(PoseTransforms, Dataset):
    """On-disk clips in the reference's format (gesture_dataset.py:14-119): ``<root>/<speaker>/processed_137.csv``
    with columns dataset,start,end,interval_id,pose_fn,audio_fn,video_fn,speaker and one npz per clip holding
    ``pose (>=64,3,137)`` (x, y, confidence in pixels) and ``audio``.  ``__getitem__`` reproduces :85-119: crop/pad the
    audio to 68266 samples, 137 -> 122 -> 121 keypoints relative to the root joint, hierarchical ("parted") offsets,
    normalisation with the speaker's statistics.  Speaker statistics are data: register them with
    ``register_speaker_stat`` / ``load_speaker_stats`` (the reference hard-codes them in speakers_stat.py).
    ``root_dir=None`` gives a transform-only instance (what Voice2PoseModel.forward needs of a dataset)."""

    def __init__(self, root_dir=None, speaker=None, split='train', cfg=None, demo_input=None):
        self.cfg = cfg.DATASET
        self.speaker, self.split = speaker, split
        self.clips = None
        if root_dir is None:
            return
        assert speaker is not None, 'The speaker is "None"!'
        self.root_dir = os.path.join(root_dir, speaker)
        if split == 'demo':  # wav file(s) / a directory of wav files (gesture_dataset.py:28-36)
            assert demo_input is not None, 'demo split needs demo_input'
            if len(demo_input.split()) == 1 and os.path.isdir(demo_input):
                files = os.listdir(demo_input)
                np.random.shuffle(files)
                files = [f for f in files[:1000] if f.split('.')[-1] == 'wav'][:cfg.DEMO.NUM_SAMPLES]
                self.clips = [os.path.join(demo_input, f) for f in files]
            else:
                self.clips = demo_input.split()
            if self.cfg.SUBSET is not None:
                self.clips = self.clips[:self.cfg.SUBSET]
            return
        if split not in ('train', 'val'):
            raise NotImplementedError(split)
        csv_path = os.path.join(self.root_dir, 'processed_137.csv')
        if not os.path.exists(csv_path):
            raise FileNotFoundError('No csv file: %s' % csv_path)
        import pandas as pd
        clips = pd.read_csv(csv_path)
        self.clips = clips[clips['dataset'] == ('train' if split == 'train' else 'dev')]
        if self.cfg.SUBSET is not None:
            self.clips = self.clips[:self.cfg.SUBSET]

    def __len__(self):
        return 0 if self.clips is None else len(self.clips)

    def remove_unuesd_kp(self, poses):
        assert poses.shape[-1] == 137
        return poses[..., :, _KEEP_137]

    def absolute_to_relative(self, poses):
        poses[..., :2, :] = poses[..., :2, :] - poses[..., :2, self.root_node, None]
        return poses[..., :, _DROP_ROOT]

    def _demo_item(self, idx):
        """gesture_dataset.py:54-79.  The reference decodes with librosa.load(path, sr=16000) (mono, float32 in [-1,1]);
        here PCM wav files are read with scipy and, only if their rate differs from DATASET.AUDIO_SR, resampled with a
        polyphase filter (librosa's resampler is not available offline -- identical samples for 16 kHz input)."""
        feed = self.clips[idx]
        if feed.split('.')[-1] != 'wav':
            raise NotImplementedError('Audio format %s is not supported.' % feed.split('.')[-1])
        from scipy.io import wavfile
        sr, a = wavfile.read(feed)
        if a.dtype.kind == 'i':
            a = a.astype(np.float32) / float(2 ** (8 * a.dtype.itemsize - 1))
        elif a.dtype.kind == 'u':  # 8-bit PCM is unsigned
            a = (a.astype(np.float32) - 128.0) / 128.0
        else:
            a = a.astype(np.float32)
        if a.ndim == 2:
            a = a.mean(axis=1)
        if sr != self.cfg.AUDIO_SR:
            from math import gcd
            from scipy.signal import resample_poly
            k = gcd(int(sr), int(self.cfg.AUDIO_SR))
            a = resample_poly(a, self.cfg.AUDIO_SR // k, sr // k).astype(np.float32)
        if self.cfg.MAX_DEMO_LENGTH is not None:
            max_length = self.cfg.MAX_DEMO_LENGTH * self.cfg.AUDIO_SR
            if len(a) > max_length:
                start = np.random.randint(0, len(a) - max_length)
                a = a[start:start + max_length]
        audio_length, num_frames = parse_audio_length(len(a), self.cfg.AUDIO_SR, self.cfg.FPS)
        return {'speaker': self.speaker, 'audio': crop_pad_audio(a, audio_length), 'clip_index': idx,
                'speaker_stat': self.get_speaker_stat(self.speaker, 121, self.cfg.HIERARCHICAL_POSE), 'num_frames': num_frames}

    def __getitem__(self, idx):
        if self.split == 'demo':
            return self._demo_item(idx)
        clip = self.clips.iloc[idx]
        speaker = clip['speaker']
        arr = np.load(os.path.join(self.root_dir, clip['pose_fn']))
        audio_length, num_frames = parse_audio_length(self.cfg.AUDIO_LENGTH, self.cfg.AUDIO_SR, self.cfg.FPS)
        audio = crop_pad_audio(arr['audio'], audio_length)
        p = torch.Tensor(arr['pose'][:self.cfg.NUM_FRAMES, ...])
        p = self.absolute_to_relative(self.remove_unuesd_kp(p))
        if self.cfg.HIERARCHICAL_POSE:
            p = self.global_to_parted(p)
        poses, score = p[:, :2, :], p[:, 2:, :].repeat(1, 2, 1)
        stat = self.get_speaker_stat(speaker, poses.shape[-1], parted=self.cfg.HIERARCHICAL_POSE)
        return {'speaker': speaker, 'audio': audio, 'num_frames': num_frames, 'clip_index': idx,
                'poses': self.normalize_poses(poses, stat), 'poses_score': score, 'speaker_stat': stat,
                'anchors': {'hand_root_l': HAND_ROOT_L, 'hand_root_r': HAND_ROOT_R, 'head_root': HEAD_ROOT}}


class DeviceClipStore(PoseTransforms):
    """A speaker's clips resident in HBM + batch assembly on the GPU ("GPU-side batched version to remove
    the DataLoader bottleneck").  The reference reads one npz per sample in DataLoader worker processes and collates on
    the host (gesture_dataset.py:85-119, trainer.py:77); an MI355X holds the whole dataset instead -- 379 KB per clip
    (64x3x137 pose + 68266 audio samples, fp32), i.e. ~11 GB for 30k clips of 288 GB -- and ``batch(indices)`` is two
    kernel launches (ops.clip_poses_prepare, ops.rows_gather) producing the model's batch dict on the device,
    bit-identical to ``default_collate([dataset[i] for i in indices])``.  One speaker per store, like the reference."""

    def __init__(self, dataset, device='cuda', indices=None):
        self.cfg = dataset.cfg
        self.speaker = dataset.speaker
        idx = range(len(dataset)) if indices is None else indices
        audio_length, self.num_frames = parse_audio_length(self.cfg.AUDIO_LENGTH, self.cfg.AUDIO_SR, self.cfg.FPS)
        poses, audio = [], []
        for i in idx:  # one pass over the files, at start-up
            clip = dataset.clips.iloc[i]
            assert clip['speaker'] == self.speaker, 'one speaker per store'
            arr = np.load(os.path.join(dataset.root_dir, clip['pose_fn']))
            poses.append(torch.Tensor(arr['pose'][:self.cfg.NUM_FRAMES, ...]))  # fp32 like :95
            audio.append(torch.from_numpy(np.ascontiguousarray(crop_pad_audio(arr['audio'], audio_length), dtype=np.float32)))
        self.raw_poses = torch.stack(poses).contiguous().to(device)   # (N, T, 3, 137)
        self.audio = torch.stack(audio).contiguous().to(device)       # (N, 68266)
        stat = self.get_speaker_stat(self.speaker, self.cfg.NUM_LANDMARKS, parted=self.cfg.HIERARCHICAL_POSE)
        self.stat = stat
        to32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=torch.float32, device=device)  # noqa: E731  (:174-176)
        self._mean32, self._std32 = to32(stat['mean']), to32(stat['std'])
        self._mean64 = torch.tensor(np.asarray(stat['mean'], dtype=np.float64), device=device)
        self._std64 = torch.tensor(np.asarray(stat['std'], dtype=np.float64), device=device)
        self._scale64 = torch.tensor(float(stat['scale_factor']), dtype=torch.float64, device=device)

    def __len__(self):
        return self.raw_poses.shape[0]

    def batch(self, indices):
        """indices: int64 tensor / sequence of clip positions in the store -> the collated sample dict, on the device."""
        from ... import ops
        dev = self.raw_poses.device
        idx = torch.as_tensor(indices, dtype=torch.int64).to(dev)
        if idx.numel() == 0 or int(idx.min()) < 0 or int(idx.max()) >= len(self):
            raise IndexError('clip index out of range')
        B = idx.numel()
        poses, score = ops.clip_poses_prepare(self.raw_poses, idx, self._mean32, self._std32, self.num_frames,
                                              self.cfg.HIERARCHICAL_POSE)
        return {'speaker': [self.speaker] * B, 'audio': ops.rows_gather(self.audio, idx),
                'num_frames': torch.full((B,), self.num_frames, dtype=torch.int64), 'clip_index': idx,
                'poses': poses, 'poses_score': score,
                'speaker_stat': {'scale_factor': self._scale64.expand(B), 'mean': self._mean64.expand(B, -1),
                                 'std': self._std64.expand(B, -1)},
                'anchors': {'hand_root_l': torch.full((B,), HAND_ROOT_L), 'hand_root_r': torch.full((B,), HAND_ROOT_R),
                            'head_root': torch.full((B,), HEAD_ROOT)}}


_BUILTIN_LOADED = [False]


def load_builtin_speaker_stats():
    """Register the reference's speakers (core/datasets/speakers_stat.py:4-1492: 9 with global-relative statistics, 11 with hierarchical
    ones -- oliver at :497/:1265, kubinec at :440/:1208 ...) from the data file tests/golden/make_speaker_stats.py exported.  Speakers already
    registered by the caller are left alone.  Called lazily by ``get_speaker_stat``."""
    if _BUILTIN_LOADED[0]:
        return
    _BUILTIN_LOADED[0] = True
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'speakers_stat_121.npz')
    if not os.path.exists(path):
        return
    data = np.load(path)
    for key in data.files:
        table, name, field = key.split('/')
        if field != 'mean':
            continue
        dst = SPEAKERS_STAT_121_parted if table.endswith('_parted') else SPEAKERS_STAT_121
        if name not in dst:
            pre = '%s/%s/' % (table, name)
            dst[name] = {'scale_factor': float(data[pre + 'scale_factor']), 'mean': data[pre + 'mean'], 'std': data[pre + 'std']}


def load_speaker_stats(npz_path, name):
    """Register statistics stored as {parted,global}_{mean,std,scale} arrays (see tests/golden/speaker_stat_oliver.npz)."""
    sp = np.load(npz_path)
    register_speaker_stat(name,
                          parted={'mean': sp['parted_mean'], 'std': sp['parted_std'], 'scale_factor': float(sp['parted_scale'])},
                          global_={'mean': sp['global_mean'], 'std': sp['global_std'], 'scale_factor': float(sp['global_scale'])})


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
