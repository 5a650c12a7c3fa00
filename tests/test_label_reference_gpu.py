"""The DEVICE label path (apa_pose_labels_device: rasterise + replay + normalise + resize, one block per image)
against the vectors the reference's own `train_preprocess_pipeline` produced (tests/golden/label_reference.npz,
see tests/golden/make_label_reference.py).  Frames of a video are separate images of the device batch: on the
binary canvases of the rasteriser the joint min-max normalisation of :201-202 equals the per-image one
(tests/test_label_reference_cpu.py::test_frames_are_normalised_jointly)."""
import json
import os

import numpy as np
import pytest
import torch

from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
Z = np.load(os.path.join(GOLD, 'label_reference.npz'))
RASTER = [n for n in json.loads(str(Z['cases'])) if n.startswith('raster_')]


def _case(name):
    pre = name + '/'
    d = {k[len(pre):]: Z[k] for k in Z.files if k.startswith(pre)}
    d['meta'] = json.loads(str(d['meta']))
    return d


@pytest.mark.parametrize('name', RASTER)
def test_device_label_path_is_the_reference(name):
    c = _case(name)
    m = c['meta']
    T, J = m['T'], m['J']
    poses = [c['in/pose'][t, :int(c['in/n_vals'][t])] for t in range(T)]
    labels, valid, status = cof.pose_labels_device(
        poses, [c['in/geom'].tolist()] * T, out_wd=max(200, m['side']), J=J,
        marker_wd_ratio=m['marker_wd_ratio_cfg'], out_side=m['side'])
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0] * T
    assert np.array_equal(valid.cpu().numpy(), c['out/valid'])
    assert np.array_equal(labels.cpu().numpy(), c['out/labels'])


def test_device_label_batch_mixes_cases_and_flags_the_bad_crop():
    """All reference cases in ONE launch (different geometry per image), plus the crop tf.slice refuses."""
    poses, geoms, want, want_valid = [], [], [], []
    for name in RASTER:
        c = _case(name)
        if c['meta']['side'] != 15:
            continue
        for t in range(c['meta']['T']):
            poses.append(c['in/pose'][t, :int(c['in/n_vals'][t])])
            geoms.append(c['in/geom'].tolist())
            want.append(c['out/labels'][t])
            want_valid.append(c['out/valid'][t])
    bad = _case('rand_crop_out_of_range')
    assert bad['meta']['raises']
    poses.append(np.full((48,), 5, dtype=np.int64))
    geoms.append(bad['in/geom'].tolist())
    labels, valid, status = cof.pose_labels_device(poses, geoms, out_wd=200, J=16, marker_wd_ratio=0.1, out_side=15)
    torch.cuda.synchronize()
    assert status.cpu().tolist() == [0] * len(want) + [1]
    assert np.array_equal(labels[:-1].cpu().numpy(), np.stack(want))
    assert np.array_equal(valid[:-1].cpu().numpy(), np.stack(want_valid))
    assert float(labels[-1].abs().max()) == 0.0
