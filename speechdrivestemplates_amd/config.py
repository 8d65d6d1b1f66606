"""Config shim with the reference's keys and merge order (configs/default.py:4-97, main.py:29-33):
defaults <- yaml file <- CLI ``KEY VAL`` pairs, then frozen.  yacs is not a dependency: ``CfgNode`` here is
a ~60-line attribute-access dict with the four methods the reference calls (merge_from_file,
merge_from_list, freeze, clone)."""
import ast
import copy

import yaml

_DEFAULTS = {
    "PIPELINE_TYPE": None,
    "VOICE2POSE": {
        "STRICT_LOADING": True,
        "GENERATOR": {
            "NAME": None, "LEAKY_RELU": True, "NORM": "IN", "LAMBDA_REG": 1.0, "LAMBDA_CLIP_KL": 0.1,
            "CLIP_CODE": {"DIMENSION": None, "LR_SCALING": 1.0, "TRAIN": True, "FRAME_VARIANT": False,
                          "SAMPLE_FROM_NORMAL": False, "TEST_WITH_GT_CODE": False, "EXTERNAL_CODE": False,
                          "EXTERNAL_CODE_PTH": None},
        },
        "POSE_ENCODER": {"NAME": "PoseSeqEncoder", "AE_CHECKPOINT": None},
        "POSE_DISCRIMINATOR": {"NAME": None, "LEAKY_RELU": False, "LAMBDA_GAN": 1.0, "MOTION": True, "WHITE_LIST": None},
    },
    "POSE2POSE": {
        "AUTOENCODER": {"NAME": None, "LEAKY_RELU": True, "NORM": "BN", "CODE_DIM": 32},
        "LAMBDA_REG": 1.0, "LAMBDA_KL": 0.1,
    },
    "DATASET": {
        "NAME": "GestureDataset", "ROOT_DIR": "datasets/speakers", "SUBSET": None, "NUM_LANDMARKS": 121,
        "HIERARCHICAL_POSE": True, "SPEAKER": None, "NUM_FRAMES": 64, "AUDIO_LENGTH": 68267, "MAX_DEMO_LENGTH": 24,
        "AUDIO_SR": 16000, "FPS": 15, "CACHING": False,
        # extension (not in the reference): number of clips of the seeded synthetic dataset used by bench/tests
        "SYNTHETIC_CLIPS": 4096,
    },
    "TRAIN": {"NUM_EPOCHS": 100, "BATCH_SIZE": 32, "SAVE_VIDEO": True, "SAVE_NPZ": False, "LR": 1e-4, "WD": 0,
              "LR_SCHEDULER": True, "PRETRAIN_FROM": None, "VALIDATE": True, "NUM_RESULT_SAMPLE": 2,
              "CHECKPOINT_INTERVAL": 1},
    "TEST": {"BATCH_SIZE": 32, "NUM_RESULT_SAMPLE": 8, "SAVE_VIDEO": True, "SAVE_NPZ": True, "MULTIPLE": 1},
    "DEMO": {"MULTIPLE": 1, "NUM_SAMPLES": 1, "CODE_INDEX": None, "CODE_INDEX_B": None, "CODE_PATH": None},
    "SYS": {"OUTPUT_DIR": "output/", "CANVAS_SIZE": (720, 1280), "VISUALIZATION_SCALING": 0.85,
            "VIDEO_FORMAT": ["mp4", "img"], "ASYNC_VIDEO_SAVING": False, "LOG_INTERVAL": 100, "NUM_WORKERS": 8,
            "DISTRIBUTED": False, "WORLD_SIZE": 1, "MASTER_ADDR": "localhost", "MASTER_PORT": 21379,
            # extensions of this engine (absent from the reference's configs/default.py:4-97; defaults = the reference's behaviour):
            # STORAGE 'bf16' = BASELINE config 4's arithmetic for the Conv2d chain (ops.set_storage), HIP_GRAPH = replay the train step
            # from a captured hipGraph (graph.GraphedStep, single-GPU runs)
            # CHAIN1D False = the generator's Conv1d blocks one by one: the one-launch chain spins on clusters of co-resident workgroups, so a GPU
            # that two training processes share must not run two of them at once (each would wait for CUs the other one holds until the spin limit
            # trips and the Trainer raises)
            "STORAGE": "f32", "HIP_GRAPH": False, "CHAIN1D": True,
            # CONV_F32_SPLIT (fp32 tensors): Conv2d products as six bf16 MFMA products of an exact three-way bf16 split of both operands, fp32
            # accumulation (csrc/convbf.hip; fp32-grade results, 1.3x faster); False = the fp32-MFMA kernels of rounds 3-4
            "CONV_F32_SPLIT": True,
            # DDP_UNSYNCED_D True = the reference's data-parallel quirk (core/pipelines/voice2pose.py:301,308): DistributedDataParallel
            # all-reduces the gradients of the FIRST backward of a step only, so the discriminator's own backward (the second one of an s2g step)
            # leaves per-rank gradients and the rank's discriminators drift apart.  Default False: this engine synchronises them (dp.GradReducer).
            "DDP_UNSYNCED_D": False},
}


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError("Attempted to set %s on a frozen CfgNode" % k)
        self[k] = v

    def freeze(self, flag=True):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)

    def defrost(self):
        self.freeze(False)

    def clone(self):
        return copy.deepcopy(self)

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError("Non-existent config key: %s%s" % (path, k))
            if isinstance(v, dict):
                self[k]._merge(v, path + k + ".")
            else:
                if isinstance(self[k], float) and isinstance(v, (str, int)) and not isinstance(v, bool):
                    v = float(v)  # YAML 1.1 reads "1e-4" as a string
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "override list must be KEY VAL pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent config key: %s" % key)
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = val


def get_cfg_defaults():
    return CfgNode(_DEFAULTS)


def load_cfg(config_file=None, opts=()):
    """main.py:29-33 -- defaults <- yaml <- KEY VAL list, frozen."""
    cfg = get_cfg_defaults()
    if config_file:
        cfg.merge_from_file(config_file)
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    return cfg
