#!/usr/bin/env python
"""Golden vectors for the pose-LABEL path, produced by the REFERENCE'S OWN code.

The reference builds the pose labels inside `train_preprocess_pipeline` (src/preprocess_pipeline.py:135-219):
the label rasteriser is called once per frame (:150-167), the image's crop / flip is replayed on the canvases
(`_replay_augmentation`, :21-45, called at :193), the result is converted to float, min-max normalised with
cfg.EPS and resized to FINAL_POSE_HMAP_SIDE^2 (:194-207), then split back into frames (:212-214).

This script exec's that file (and src/config.py) from /root/reference behind the TF1 stand-in
(tests/golden/tf1_shim.py, in FLOAT32 mode: the label graph is float32 and its crop truncations depend on
it) and runs `train_preprocess_pipeline` itself.  What is stubbed, and why that does not weaken the pin:
  * the compiled OpenCV op behind `tf.load_op_library('pose_to_heatmap.so')` (pose_to_heatmap.cc), which is
    not buildable here; its python wrapper (src/custom_ops/custom_ops_factory.py:20-28: *= 255.0, cast to uint8)
    IS the reference's, exec'd from its file.  The op stand-in records the arguments the reference passes
    (canvas width max(200, side), out_channels, do_gauss_blur=False, marker_wd_ratio =
    cfg.HEATMAP_MARKER_WD_RATIO) and returns a FLOAT canvas: either
      - the canvas of oracle/labels_eval_oracle.py's restatement of the op for those arguments (cases
        'raster_*': the raster RULE itself stays unpinned -- cv::circle is third-party; everything AFTER
        the canvas is the reference's code), or
      - a seeded random uint8 canvas with a non-zero minimum (cases 'rand_*': pins the replay / normalise /
        resize arithmetic on non-binary data, independent of any rasteriser).
  * `image_preprocessing_fn` -- the image half of the input pipeline (out of scope).  The stub writes the
    three `preproc_info` entries exactly where models/slim/preprocessing/vgg_preprocessing.py writes them:
    'image_shape' = shape AFTER the aspect-preserving resize (:325), 'crop_info' = [offset_h, offset_w,
    crop_h, crop_w] (:177-178), 'whether_flip' (:348).
  * the dataset `provider`.

Run in the build container (it reads /root/reference):   python tests/golden/make_label_reference.py
Output: tests/golden/label_reference.npz.  Test infrastructure only.
"""
from __future__ import annotations

import json
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf1_shim as tfs                                   # noqa: E402
import make_head_reference as mhr                        # noqa: E402
from oracle import labels_eval_oracle as leo             # noqa: E402

REF = mhr.REF


def _rs(*keys):
    return np.random.RandomState(zlib.crc32('|'.join(str(k) for k in keys).encode()) & 0x7fffffff)


# ------------------------------------------------------------------------------------------------ cases
# geometry: im = ORIGINAL image (keypoint frame); aug = after the aspect-preserving resize to RESIZE_SIDE
# (vgg_preprocessing.py:325); crop on aug.
def _aug_shape(im_ht, im_wd, side):
    """models/slim/preprocessing/vgg_preprocessing.py `_smallest_size_at_least`: scale = side / min(h, w),
    new = to_int32(h * scale), to_int32(w * scale) in float32."""
    h, w = np.float32(im_ht), np.float32(im_wd)
    scale = np.float32(side) / (h if h < w else w)       # tf.cond(greater(height, width), s/width, s/height)
    return int(h * scale), int(w * scale)


def _pose(rs, n_people, J, im_ht, im_wd, p_missing=0.25):
    out = []
    for _ in range(n_people):
        for _j in range(J):
            if rs.rand() < p_missing:
                out += [-1, -1, 0]
            else:                                         # a few keypoints fall outside the image on purpose
                out += [int(rs.randint(-20, im_wd + 20)), int(rs.randint(-20, im_ht + 20)), int(rs.randint(0, 2))]
    return np.asarray(out, dtype=np.int64)


CASES = [
    # name, kind, T, J, (im_ht, im_wd), resize_side, crop side, (oy, ox) as fractions of the slack, flip, side
    dict(name='raster_mpii_landscape', kind='raster', T=1, J=16, im=(480, 640), rside=256, crop=224, off=(0.5, 0.3), flip=False, side=15, people=1),
    dict(name='raster_mpii_landscape_flip', kind='raster', T=1, J=16, im=(480, 640), rside=256, crop=224, off=(0.9, 0.7), flip=True, side=15, people=2),
    dict(name='raster_mpii_portrait', kind='raster', T=1, J=16, im=(720, 405), rside=256, crop=224, off=(0.2, 1.0), flip=True, side=15, people=1),
    dict(name='raster_cfg003_450', kind='raster', T=1, J=16, im=(1080, 1920), rside=512, crop=450, off=(0.0, 0.55), flip=False, side=15, people=3),
    dict(name='raster_video_3frames', kind='raster', T=3, J=16, im=(240, 320), rside=256, crop=224, off=(0.4, 0.6), flip=True, side=15, people=1),
    dict(name='raster_video_empty_frame', kind='raster', T=2, J=16, im=(240, 320), rside=256, crop=224, off=(1.0, 0.0), flip=False, side=15, people=1, empty_frames=(1,)),
    dict(name='raster_no_keypoints', kind='raster', T=1, J=16, im=(300, 300), rside=256, crop=224, off=(0.5, 0.5), flip=False, side=15, people=0),
    dict(name='rand_landscape', kind='rand', T=1, J=4, im=(480, 640), rside=256, crop=224, off=(0.37, 0.81), flip=False, side=15),
    dict(name='rand_portrait_flip', kind='rand', T=1, J=3, im=(500, 333), rside=256, crop=224, off=(0.66, 0.5), flip=True, side=15),
    dict(name='rand_video_joint_minmax', kind='rand', T=2, J=2, im=(360, 480), rside=256, crop=224, off=(0.1, 0.9), flip=True, side=15),
    dict(name='rand_square_whole_image', kind='rand', T=1, J=2, im=(256, 256), rside=224, crop=224, off=(0.0, 0.0), flip=False, side=15),
    dict(name='rand_side_above_200', kind='rand', T=1, J=2, im=(300, 400), rside=256, crop=224, off=(0.5, 0.5), flip=False, side=210),
    dict(name='rand_odd_ratio', kind='rand', T=1, J=2, im=(333, 517), rside=300, crop=299, off=(1.0, 0.43), flip=True, side=17),
    # the crop reaches past the resized image: tf.slice refuses it (the product reports an error status)
    dict(name='rand_crop_out_of_range', kind='rand', T=1, J=2, im=(480, 640), rside=256, crop=224, off=(0.0, 0.0), flip=False, side=15,
         force_crop=(40, 0, 224, 224)),
]


def load_pipeline():
    """-> (cfg module, preprocess_pipeline module with the stubbed custom-op module, call log)."""
    cfgmod, _nf, _loss = mhr.load_reference()
    calls = []
    state = {}

    def pose_to_heatmap_op(pl, im_ht, im_wd, out_wd, out_channels=16, do_gauss_blur=True, marker_wd_ratio=0.1):
        """stand-in of the compiled op (pose_to_heatmap.cc): FLOAT canvas in [0, 1] + valid flags"""
        args = dict(im_ht=int(tfs._raw(im_ht)), im_wd=int(tfs._raw(im_wd)), out_wd=int(out_wd),
                    out_channels=int(out_channels), do_gauss_blur=bool(do_gauss_blur),
                    marker_wd_ratio=float(marker_wd_ratio))
        calls.append(args)
        hm, valid = state['canvas_fn'](np.asarray(tfs._raw(pl)), args, len(calls) - 1)
        return tfs.Tensor(torch.from_numpy(hm)), tfs.Tensor(torch.from_numpy(valid))

    # the reference's OWN python wrapper (src/custom_ops/custom_ops_factory.py:20-28: set_shape, *= 255.0, cast
    # to uint8) around the op stand-in: tf.load_op_library hands back a namespace with the op
    tf = sys.modules['tensorflow']
    tf.load_op_library = lambda path: types.SimpleNamespace(
        pose_to_heatmap=pose_to_heatmap_op, zero_out_channels=None, render_pose=None, render_objects=None)
    cof = mhr._exec_ref(os.path.join(REF, 'src', 'custom_ops', 'custom_ops_factory.py'),
                        'custom_ops.custom_ops_factory')
    pkg = types.ModuleType('custom_ops')
    pkg.__path__ = []
    pkg.custom_ops_factory = cof
    sys.modules['custom_ops'] = pkg
    sys.modules['custom_ops.custom_ops_factory'] = cof
    pp = mhr._exec_ref(os.path.join(REF, 'src', 'preprocess_pipeline.py'), 'refpreproc')
    replay = pp._replay_augmentation

    def spy(H, aug_info):                                  # records the uint8 canvas stack the replay receives
        state['canvas_u8'] = tfs._raw(H).numpy().copy()
        assert state['canvas_u8'].dtype == np.uint8
        return replay(H, aug_info)
    pp._replay_augmentation = spy
    return cfgmod, pp, calls, state


def run_case(cfgmod, pp, calls, state, c):
    cfg = cfgmod.cfg
    T, J = c['T'], c['J']
    im_ht, im_wd = c['im']
    aug_ht, aug_wd = _aug_shape(im_ht, im_wd, c['rside'])
    crop = c['crop']
    if 'force_crop' in c:
        oy, ox, ch, cw = c['force_crop']
    else:
        oy = int(round(c['off'][0] * (aug_ht - crop)))
        ox = int(round(c['off'][1] * (aug_wd - crop)))
        ch = cw = crop
    rs = _rs('label', c['name'])
    poses = []
    for t in range(T):
        if c['kind'] == 'raster' and t in c.get('empty_frames', ()):
            poses.append(_pose(rs, 1, J, im_ht, im_wd, p_missing=1.1))
        elif c['kind'] == 'raster':
            poses.append(_pose(rs, c['people'], J, im_ht, im_wd) if c['people'] else np.zeros((0,), np.int64))
        else:
            poses.append(np.full((3 * J,), -1, dtype=np.int64))


    def canvas_fn(pl, args, idx):
        if c['kind'] == 'raster':
            hm, valid = leo.pose_to_heatmap(
                pl, args['im_ht'], args['im_wd'], args['out_wd'], out_channels=args['out_channels'],
                do_gauss_blur=args['do_gauss_blur'], marker_wd_ratio=args['marker_wd_ratio'])
            hm = np.asarray(hm, dtype=np.float32)
            valid = np.asarray(valid, dtype=bool)
        else:
            out_ht = int(args['im_ht'] * args['out_wd'] * 1.0 / args['im_wd'])      # pose_to_heatmap.cc:52
            r = _rs('canvas', c['name'], idx)
            coarse = r.rand(out_ht // 8 + 2, args['out_wd'] // 8 + 2, args['out_channels'])
            hm = np.kron(coarse, np.ones((8, 8, 1)))[:out_ht, :args['out_wd']]
            u8 = (20 + 200 * hm + 10 * r.rand(*hm.shape)).astype(np.uint8)           # min > 0, max < 255
            hm = ((u8.astype(np.float32) + np.float32(0.5)) / np.float32(255.0))      # the wrapper's *255 -> uint8 gives u8 back
            valid = r.rand(args['out_channels']) < 0.7
        return hm, valid

    state['canvas_fn'] = canvas_fn
    del calls[:]

    # the configuration knobs this path reads
    cfg.TRAIN.FINAL_POSE_HMAP_SIDE = c['side']
    cfg.TRAIN.IMAGE_SIZE = crop
    cfg.TRAIN.RESIZE_SIDE = c['rside']
    cfg.TRAIN.LOSS_FN_POSE = 'l2'
    cfg.INPUT.INPUT_IMAGE_FORMAT = 'image'

    class Provider(object):
        def get(self, items):
            img = torch.zeros((T, 8, 8, 3) if T > 1 else (8, 8, 3), dtype=torch.uint8)
            vals = {'image': tfs.Tensor(img),
                    'pose': [tfs.Tensor(torch.from_numpy(p)) for p in poses] if T > 1
                    else tfs.Tensor(torch.from_numpy(poses[0])),
                    'im_ht': tfs.Tensor(torch.tensor(im_ht, dtype=torch.int64)),
                    'im_wd': tfs.Tensor(torch.tensor(im_wd, dtype=torch.int64)),
                    'action_label': tfs.Tensor(torch.tensor(3, dtype=torch.int64))}
            return [vals[i] for i in items]

    seen = {}

    def image_preprocessing_fn(image, out_h, out_w, resize_side_min=None, resize_side_max=None,
                               preproc_info=None, modality='rgb'):
        seen.update(out_hw=(int(out_h), int(out_w)), rside=(int(resize_side_min), int(resize_side_max)),
                    in_channels=int(image.get_shape().as_list()[-1]))
        preproc_info['image_shape'] = [aug_ht, aug_wd, int(image.get_shape().as_list()[-1])]   # vgg_preprocessing.py:325
        preproc_info['crop_info'] = [oy, ox, ch, cw]                                            # :177-178
        preproc_info['whether_flip'] = tfs.Tensor(torch.tensor(bool(c['flip'])))                # :348
        return tfs.Tensor(torch.zeros((int(out_h), int(out_w), seen['in_channels']), dtype=tfs.DT))

    network_fn = types.SimpleNamespace(default_image_size=224)
    err = None
    try:
        image, hmap, valid, action = pp.train_preprocess_pipeline(Provider(), cfg, network_fn, J,
                                                                  image_preprocessing_fn)
    except ValueError as e:
        err = str(e)

    max_vals = max(max(p.size for p in poses), 3 * J)
    pose_pad = np.full((T, max_vals), -1, dtype=np.int64)
    for t, p in enumerate(poses):
        pose_pad[t, :p.size] = p
    out = {
        'in/pose': pose_pad,
        'in/n_vals': np.asarray([p.size for p in poses], dtype=np.int32),
        'in/geom': np.asarray([im_ht, im_wd, aug_ht, aug_wd, oy, ox, ch, cw, int(c['flip'])], dtype=np.int32),
        'in/canvas': state['canvas_u8'],                                    # uint8 [h, w, J*T] (:172)
        'meta': np.asarray(json.dumps({
            'kind': c['kind'], 'T': T, 'J': J, 'side': c['side'], 'eps': float(cfg.EPS),
            'calls': list(calls), 'image_preprocessing_fn_saw': seen, 'raises': err,
            'marker_wd_ratio_cfg': float(cfg.HEATMAP_MARKER_WD_RATIO)}, sort_keys=True)),
    }
    if err is None:
        lab = tfs._raw(hmap)
        assert lab.dtype == torch.float32 and list(lab.shape) == [T, c['side'], c['side'], J], lab.shape
        out['out/labels'] = lab.numpy()
        out['out/valid'] = tfs._raw(valid).numpy().astype(bool)
        assert int(tfs._raw(action)) == 3 and list(tfs._raw(image).shape) == [T, crop, crop, 3]
    return out


def generate():
    prev = tfs.DT
    tfs.set_float_dtype(torch.float32)
    tfs.set_graph(tfs.Graph(None, None))                # scopes only: the label graph has no variables / randomness
    try:
        cfgmod, pp, calls, state = load_pipeline()
        import copy
        base = copy.deepcopy(cfgmod.cfg)
        blobs = {}
        for c in CASES:
            cfgmod.cfg.clear()
            cfgmod.cfg.update(copy.deepcopy(base))
            for k, v in run_case(cfgmod, pp, calls, state, c).items():
                if v is not None:
                    blobs['{}/{}'.format(c['name'], k)] = v
        blobs['cases'] = np.asarray(json.dumps([c['name'] for c in CASES]))
        return blobs
    finally:
        tfs.set_float_dtype(prev)
        tfs.set_graph(None)


if __name__ == '__main__':
    blobs = generate()
    path = os.path.join(HERE, 'label_reference.npz')
    np.savez_compressed(path, **blobs)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(CASES), 'cases')
