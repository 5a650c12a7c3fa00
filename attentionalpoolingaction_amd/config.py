"""Experiment configuration for the attentional-pooling head.

Keeps the reference's key names and defaults for everything the hot path reads, so the YAML files
under /root/reference/experiments/*.yaml load unchanged (same strict behaviour as
/root/reference/src/config.py:287-325: an unknown key or a value of the wrong type is an error).
Only the keys that the shipped YAMLs or the head touch are declared; the reference's input-pipeline
knobs (glimpses, rendered poses, video reading ...) are out of scope.
"""
from __future__ import annotations

import copy
from typing import Any, Dict

import yaml


class AttrDict(dict):
    """dict with attribute access (what the reference gets from easydict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _defaults() -> AttrDict:
    c = AttrDict()
    # --- top level (config.py:238-272)
    c.RNG_SEED = 42
    c.EPS = 1e-14
    c.EXP_DIR = 'expt_outputs/'
    c.DATASET_NAME = 'mpii'
    c.DATASET_DIR = 'data/mpii/mpii_tfrecords'
    c.DATASET_LIST_DIR = ''
    c.MODEL_NAME = 'inception_v3'
    c.NUM_READERS = 4
    c.NUM_PREPROCESSING_THREADS = 4
    c.GPUS = '2'
    c.HEATMAP_MARKER_WD_RATIO = 0.1
    c.MAX_INPUT_IMAGE_SIZE = 512
    c.INPUT_FILE_STYLE_LABEL = ''

    # --- INPUT (config.py:18-42): input formats of the data pipeline -- accepted so that any reference YAML
    # loads, not consumed by the head (out of scope)
    i = c.INPUT = AttrDict()
    i.INPUT_IMAGE_FORMAT = 'normal'
    i.INPUT_IMAGE_FORMAT_POSE_RENDER_TYPE = 'rgb'
    i.POSE_GLIMPSE_CONTEXT_RATIO = 0.0
    i.POSE_GLIMPSE_RESIZE = False
    i.POSE_GLIMPSE_PARTS_KEEP = []
    i.SPLIT_ID = 1
    i.VIDEO = AttrDict(MODALITY='rgb')

    # --- TRAIN (config.py:44-133)
    t = c.TRAIN = AttrDict()
    t.BATCH_SIZE = 10
    t.WEIGHT_DECAY = 0.0005
    t.CLIP_GRADIENTS = -1.0
    t.IMAGE_SIZE = 450
    t.RESIZE_SIDE = 480
    t.FINAL_POSE_HMAP_SIDE = 15
    t.LABEL_SMOOTHING = False
    t.LEARNING_RATE = 0.01
    t.LEARNING_RATE_DECAY_RATE = 0.33
    t.END_LEARNING_RATE = 0.00001
    t.NUM_STEPS_PER_DECAY = 0
    t.NUM_EPOCHS_PER_DECAY = 40.0
    t.LEARNING_RATE_DECAY_TYPE = 'exponential'
    t.OPTIMIZER = 'momentum'
    t.MOMENTUM = 0.9
    t.MAX_NUMBER_OF_STEPS = 100000
    t.LOG_EVERY_N_STEPS = 10
    t.CHECKPOINT_PATH = 'data/pretrained_models/inception_v3.ckpt'
    t.CHECKPOINT_EXCLUDE_SCOPES = ''
    t.DATASET_SPLIT_NAME = 'trainval_train'
    t.LOSS_FN_POSE = 'l2'
    t.LOSS_FN_POSE_WT = 1.0
    t.LOSS_FN_POSE_SAMPLED = False
    t.LOSS_FN_ACTION = 'softmax-xentropy'
    t.LOSS_FN_ACTION_WT = 1.0
    t.VIDEO_FRAMES_PER_VIDEO = 1
    t.ITER_SIZE = 1
    # the rest of the reference's TRAIN table: optimiser variants, checkpoint / summary plumbing of the TF
    # training loop -- accepted, not consumed here
    t.ADAM_BETA1 = 0.9
    t.ADAM_BETA2 = 0.999
    t.OPT_EPSILON = 1.0
    t.MOVING_AVERAGE_VARIABLES = None
    t.TRAINABLE_SCOPES = ''
    t.IGNORE_MISSING_VARS = True
    t.VAR_NAME_MAPPER = ''
    t.SAVE_SUMMARIES_SECS = 300
    t.SAVE_INTERVAL_SECS = 1800
    t.READ_SEGMENT_STYLE = False
    t.OTHER_IMG_SUMMARIES_TO_ADD = ['PosePrelogitsBasedAttention']

    # --- TEST (config.py:139-157)
    e = c.TEST = AttrDict()
    e.BATCH_SIZE = 10
    e.DATASET_SPLIT_NAME = 'trainval_val'
    e.CHECKPOINT_PATH = ''
    e.VIDEO_FRAMES_PER_VIDEO = 1
    e.EVAL_METRIC = ''
    e.MAX_NUM_BATCHES = None
    e.MOVING_AVERAGE_DECAY = None

    # --- NET (config.py:160-231): the flags that select / shape the head
    n = c.NET = AttrDict()
    n.USE_POSE_ATTENTION_LOGITS = False
    n.USE_POSE_ATTENTION_LOGITS_DIMS = [-1]
    n.USE_POSE_ATTENTION_LOGITS_AVGED_HMAP = False
    n.USE_POSE_LOGITS_DIRECTLY = False
    n.USE_POSE_LOGITS_DIRECTLY_PLUS_LOGITS = False
    n.USE_POSE_LOGITS_DIRECTLY_v2 = False
    n.USE_POSE_LOGITS_DIRECTLY_v2_EXTRA_LAYER = False
    n.USE_COMPACT_BILINEAR_POOLING = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_SOFTMAX_ATT = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_RELU_ATT = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_PER_CLASS = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_SINGLE_LAYER_ATT = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_WITH_POSE_FEAT_2LAYER = False
    n.USE_POSE_PRELOGITS_BASED_ATTENTION_RANK = 1
    n.USE_TEMPORAL_ATT = False
    n.LAST_CONV_MAP_FOR_POSE = AttrDict(
        inception_v2_tsn='InceptionV2_TSN/inception_5a', inception_v3='Mixed_7c',
        resnet_v1_101='resnet_v1_101/block4', vgg_16='vgg_16/conv5')
    n.TRAIN_TOP_BN = False
    n.DROPOUT = -1.0
    return c


cfg = _defaults()


def reset_cfg() -> AttrDict:
    """Restore the defaults in place (the reference uses one global cfg; tests need isolation)."""
    cfg.clear()
    cfg.update(_defaults())
    return cfg


def _merge(a: Dict[str, Any], b: AttrDict, path: str = '') -> None:
    for k, v in a.items():
        if k not in b:
            raise KeyError('{}{} is not a valid config key'.format(path, k))
        old = b[k]
        if isinstance(old, AttrDict):
            if not isinstance(v, dict):
                raise ValueError('{}{} must be a mapping'.format(path, k))
            _merge(v, old, path + k + '.')
            continue
        if old is not None and v is not None and type(old) is not type(v):
            # the reference tolerates nothing but identical types (config.py:299-309); we add the
            # one coercion YAML makes unavoidable: int literal for a float-typed key
            if isinstance(old, float) and isinstance(v, int) and not isinstance(v, bool):
                v = float(v)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}{}'.format(
                    type(old), type(v), path, k))
        b[k] = v


def cfg_from_file(filename: str) -> AttrDict:
    """Load a YAML experiment file and merge it into the global cfg (config.py:319-325)."""
    with open(filename, 'r') as f:
        y = yaml.safe_load(f) or {}
    _merge(y, cfg)
    return cfg


def cfg_from_dict(d: Dict[str, Any]) -> AttrDict:
    _merge(d, cfg)
    return cfg


def dropout_keep_prob(c: AttrDict = None) -> float:
    """nets_factory.py:143-146: keep_prob = 0.2 if cfg.NET.DROPOUT < 0 else 1 - DROPOUT."""
    c = cfg if c is None else c
    return 0.2 if c.NET.DROPOUT < 0 else 1.0 - c.NET.DROPOUT
