"""Global configuration ``cfg`` with the keys the hot path reads — same names, nesting and defaults as
the reference's lib/model/config.py:11-453 (an attribute dictionary; easydict is not required).

Only options consumed by the detector forward/backward path are kept; dataset, drawing and output-dir
options of the reference belong to subsystems outside this package.
"""
import os
from ast import literal_eval

import numpy as np


class AttrDict(dict):
    """dict with attribute access, recursively applied to nested dicts (EasyDict work-alike)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __setattr__(self, key, value):
        self[key] = value

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)


def _defaults():
    c = AttrDict()
    c.DEBUG = dict(EN=False, EN_TEST_MSG=True)
    # Bayesian heads (config.py:34-47) — all off by default; the uncertainty branches are not on the path
    c.UC = dict(EN_RPN_BBOX_ALEATORIC=False, EN_RPN_CLS_ALEATORIC=False, EN_RPN_BBOX_EPISTEMIC=False,
                EN_RPN_CLS_EPISTEMIC=False, EN_BBOX_ALEATORIC=False, EN_CLS_ALEATORIC=False,
                EN_BBOX_EPISTEMIC=False, EN_BBOX_EPISTEMIC_INV_TRANSFORM=False, EN_CLS_EPISTEMIC=False,
                A_NUM_CE_SAMPLE=200, A_NUM_BBOX_SAMPLE=200, E_NUM_SAMPLE=10, SORT_TYPE='')
    c.PRELOAD = False
    c.PRELOAD_FULL = False
    c.USE_FPN = False                      # config.py:52
    c.USE_LIDAR_FPN = False
    c.ENABLE_FULL_NET = True
    c.NET_TYPE = 'lidar'                   # config.py:55 (the CLIs overwrite it)
    c.DB_NAME = ''                         # config.py:57
    # where get_output_dir / get_output_tb_dir put their trees (config.py:349,358); the reference derives ROOT_DIR from
    # its own checkout, here it is the working directory unless the caller sets it
    c.ROOT_DIR = os.path.abspath(os.getcwd())
    c.EXP_DIR = 'res101'
    c.TRAIN = dict(
        LEARNING_RATE=0.001, MOMENTUM=0.5, WEIGHT_DECAY=0.0001, GAMMA=0.1, STEPSIZE=[70000, 140000, 210000],
        BATCH_SIZE=16, VAL_BATCH_SIZE=32, DOUBLE_BIAS=False, TRUNCATED=False, BIAS_DECAY=False, USE_GT=False,
        SCALES=(1.0,), FRAMES_PER_BATCH=1, ROI_BATCH_SIZE=256, FG_FRACTION=0.25, FG_THRESH=0.6, DC_THRESH=0.5,
        BG_THRESH_HI=0.5, BG_THRESH_LO=0.0, HAS_RPN=True, RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3,
        RPN_CLOBBER_POSITIVES=False, RPN_FG_FRACTION=0.5, RPN_BATCHSIZE=256, RPN_NMS_THRESH=0.7,
        RPN_PRE_NMS_TOP_N=12000, RPN_POST_NMS_TOP_N=2000, RPN_BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0),
        RPN_POSITIVE_WEIGHT=-1.0, IGNORE_DC=False, ITER=1, DISPLAY=512, SNAPSHOT_KEPT=30, SUMMARY_INTERVAL=15,
        SNAPSHOT_ITERS=5000, SNAPSHOT_PREFIX='res101_faster_rcnn',
        TOD_FILTER_LIST=['Day', 'Night', 'Dawn/Dusk'],
        # not in the reference: how SolverWrapper executes a step (model/train_graph.py)
        GRAPHS=True,             # replay each training step as a captured hipGraph (False: eager autograd launches)
        FRAMES_IN_FLIGHT=4,      # frames of a pseudo batch in flight as single-chain graphs, one per hardware queue (1: one captured step at a time)
        LIDAR=dict(BBOX_NORMALIZE_MEANS=(0.0,) * 7, BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.1, 0.2, 0.2, 0.2, 1.0)),
        IMAGE=dict(BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0), BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2)))
    c.TEST = dict(SCALES=(1.0,), NMS_THRESH=0.6, BBOX_REG=True, HAS_RPN=True, RPN_NMS_THRESH=0.7,
                  RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, MODE='nms', RPN_TOP_N=5000, IGNORE_DC=False,
                  ITER=1, AUGMENT_EN=False, TOD_FILTER_LIST=['Day', 'Night', 'Dawn/Dusk'],
                  # not in the reference: how test_net / test_frame / run_eval execute a frame (model/frame_graph.FramePool)
                  FRAME_GRAPHS=True,       # replay each frame as a captured hipGraph (False: eager launches)
                  FRAMES_IN_FLIGHT=4,      # test_net: frames in flight, one HIP stream each
                  GRAPH_MAX_SHAPES=4,      # distinct frame problems held as graphs (least recently used one is dropped)
                  GRAPH_AUTOTUNE=True)     # time the convolution plans of a new frame shape during its warm-up frames
    c.RESNET = dict(MAX_POOL=False, FIXED_BLOCKS=1)
    c.PIXEL_MEANS = np.array([[[96.866, 98.76, 93.85]]])
    c.PIXEL_STDDEVS = np.array([[[1, 1, 1]]])
    c.PIXEL_ARRANGE = [0, 1, 2]
    c.GRAD_MAX_CLIP = 20
    c.RNG_SEED = 3
    c.POOLING_MODE = 'align'               # config.py:364
    c.POOLING_SIZE = 7
    c.ANCHOR_SCALES = [2, 4, 8, 16, 32]    # config.py:373
    c.ANCHOR_RATIOS = [0.5, 0.75, 1, 1.25, 2]
    c.RPN_CHANNELS = 512
    c.ENABLE_CUSTOM_TAIL = False
    c.LIDAR = dict(X_RANGE=[0, 70], Y_RANGE=[-40, 40], Z_RANGE=[-3, 3], VOXEL_LEN=0.1, VOXEL_HEIGHT=0.5,
                   NUM_SLICES=12, NUM_META_CHANNEL=3, NUM_CHANNEL=15, MAX_PTS_PER_VOXEL=32, MAX_NUM_VOXEL=25000,
                   ANCHORS=np.array([[4.73, 2.08, 1.77]]), ANCHOR_SCALES=np.array([[1]]),
                   ANCHOR_ANGLES=np.array([0, np.pi / 2]), NUM_BBOX_ELEM=7,
                   REG_LOSS_WEIGHT=[1.0] * 7, EN_RY_SIN=True)
    c.IMAGE = dict(NUM_BBOX_ELEM=4)
    return c


cfg = _defaults()


def reset_cfg():
    """Restore the defaults in place (handy for tests that mutate the global like the CLIs do)."""
    fresh = _defaults()
    cfg.clear()
    for k, v in fresh.items():
        cfg[k] = v


def _merge(src, dst, path=''):
    for k, v in src.items():
        if k not in dst:
            raise KeyError('{} is not a valid config key'.format(path + k))
        old = dst[k]
        if isinstance(old, AttrDict):
            if not isinstance(v, dict):
                raise ValueError('config key {} expects a mapping'.format(path + k))
            _merge(v, old, path + k + '.')
            continue
        if isinstance(old, np.ndarray):
            v = np.array(v, dtype=old.dtype)
        elif type(old) is not type(v):
            raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(v), path + k))
        dst[k] = v


def cfg_from_file(filename):
    """Merge a yaml file into cfg (reference: config.py:580-586); unknown keys / type changes raise."""
    import yaml
    with open(filename, 'r') as f:
        _merge(yaml.safe_load(f) or {}, cfg)


def cfg_from_list(cfg_list):
    """Set keys from ['A.B', 'value', ...] pairs (reference: config.py:589-609)."""
    if len(cfg_list) % 2 != 0:
        raise AssertionError('cfg_from_list needs key/value pairs')
    for dotted, raw in zip(cfg_list[0::2], cfg_list[1::2]):
        node = cfg
        parts = dotted.split('.')
        for sub in parts[:-1]:
            assert sub in node, dotted
            node = node[sub]
        leaf = parts[-1]
        assert leaf in node, dotted
        try:
            value = literal_eval(raw)
        except (ValueError, SyntaxError):
            value = raw
        assert type(value) == type(node[leaf]), 'type {} does not match original type {}'.format(
            type(value), type(node[leaf]))
        node[leaf] = value


def _run_name(mode):
    """The directory leaf config.py:462-495 / :508-541 build: enabled-heads prefix + mode + time-of-day filter + ITER."""
    net_type = '{}_'.format(cfg.NET_TYPE)
    if cfg.ENABLE_FULL_NET is False:
        net_type += 'rpn_only_'
    for flag, tag in (('EN_BBOX_ALEATORIC', 'a_bbox_'), ('EN_CLS_ALEATORIC', 'a_cls_'), ('EN_BBOX_EPISTEMIC', 'e_bbox_'),
                      ('EN_CLS_EPISTEMIC', 'e_cls_'), ('EN_RPN_BBOX_ALEATORIC', 'a_rpn_bbox_'),
                      ('EN_RPN_CLS_ALEATORIC', 'a_rpn_cls_'), ('EN_RPN_BBOX_EPISTEMIC', 'e_rpn_bbox_'),
                      ('EN_RPN_CLS_EPISTEMIC', 'e_rpn_cls_')):
        if cfg.UC[flag]:
            net_type += tag
    tod = cfg.TRAIN.TOD_FILTER_LIST
    names = {'Day': 'day', 'Night': 'night', 'Dawn/Dusk': 'dawn_dusk'}
    train_filter = 'all' if len(tod) == 3 else names[tod[0]]
    return '{}{}_{}_{}'.format(net_type, mode, train_filter, cfg[mode.upper()].ITER)


def get_output_dir(db, mode='train', weights_filename=None):
    """ROOT_DIR/output/EXP_DIR/<db.name>/<run name>, created when missing (reference: config.py:454-498)."""
    outdir = os.path.join(os.path.abspath(os.path.join(cfg.ROOT_DIR, 'output', cfg.EXP_DIR, db.name)),
                          weights_filename if weights_filename is not None else _run_name(mode))
    os.makedirs(outdir, exist_ok=True)
    return outdir


def get_output_tb_dir(db, weights_filename):
    """ROOT_DIR/tensorboard/EXP_DIR/<db.name>/<run name> (reference: config.py:500-545; the run name is the 'train' one)."""
    outdir = os.path.join(os.path.abspath(os.path.join(cfg.ROOT_DIR, 'tensorboard', cfg.EXP_DIR, db.name)),
                          weights_filename if weights_filename is not None else _run_name('train'))
    os.makedirs(outdir, exist_ok=True)
    return outdir
