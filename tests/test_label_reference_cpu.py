"""The pose-LABEL path against vectors produced by the reference's own
`train_preprocess_pipeline` (src/preprocess_pipeline.py:135-219, exec'd behind the float32 TF1 stand-in by
tests/golden/make_label_reference.py; what that script stubs is listed in its header):

  * the oracle's restatement of the replay / normalise / resize lines == the reference's output, bit for bit
  * the product's HOST entry points (apa_pose_to_heatmap + apa_pose_label_replay_resize) == the same vectors
  * the call-site facts the product mirrors: canvas width max(200, side), do_gauss_blur=False, the config's
    marker ratio, one rasteriser call per frame, labels of all frames normalised JOINTLY (:172, :201-202)
  * a crop that reaches past the resized image is an error in both
No GPU compute call is made here (the device twin is covered by tests/test_label_reference_gpu.py).
"""
import json
import os

import numpy as np
import pytest

from attentionalpoolingaction_amd.custom_ops import custom_ops_factory as cof
from oracle import labels_eval_oracle as leo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
Z = np.load(os.path.join(GOLD, 'label_reference.npz'))
CASES = json.loads(str(Z['cases']))


def case(name):
    pre = name + '/'
    d = {k[len(pre):]: Z[k] for k in Z.files if k.startswith(pre)}
    d['meta'] = json.loads(str(d['meta']))
    return d


def frames_to_channels(labels):
    """[T, S, S, J] -> [S, S, T*J]: the layout before the final stack(split()) (:212-214)."""
    T, S, _, J = labels.shape
    return np.transpose(labels, (1, 2, 0, 3)).reshape(S, S, T * J)


def test_fixture_inventory():
    assert len(CASES) >= 14
    kinds = {case(n)['meta']['kind'] for n in CASES}
    assert kinds == {'raster', 'rand'}
    assert sum(1 for n in CASES if case(n)['meta']['raises']) == 1
    assert any(case(n)['meta']['T'] > 1 for n in CASES)


@pytest.mark.parametrize('name', CASES)
def test_reference_call_site_facts(name):
    c = case(name)
    m = c['meta']
    assert len(m['calls']) == m['T']                               # one rasteriser call per frame (:153)
    im_ht, im_wd = int(c['in/geom'][0]), int(c['in/geom'][1])
    for call in m['calls']:
        assert call['out_wd'] == max(200, m['side'])               # :158
        assert call['do_gauss_blur'] is False                      # :162
        assert call['out_channels'] == m['J']
        assert call['marker_wd_ratio'] == m['marker_wd_ratio_cfg'] # :163
        assert (call['im_ht'], call['im_wd']) == (im_ht, im_wd)    # keypoints scale with the ORIGINAL size
    assert m['eps'] == 1e-14                                       # config.py:241
    # the canvas height rule of the op (pose_to_heatmap.cc:52) as the product computes it
    lib = cof.load_library()
    assert c['in/canvas'].shape[0] == lib.apa_pose_to_heatmap_out_ht(im_ht, im_wd, max(200, m['side']))
    assert c['in/canvas'].shape[2] == m['T'] * m['J']              # frames concatenated on channels (:172)


@pytest.mark.parametrize('name', CASES)
def test_oracle_replay_normalise_resize_is_the_reference(name):
    c = case(name)
    m = c['meta']
    g = [int(v) for v in c['in/geom']]
    if m['raises']:
        with pytest.raises(AssertionError):
            leo.replay_normalise_resize(c['in/canvas'], g[2:4], g[4:8], bool(g[8]), m['side'], eps=m['eps'])
        return
    got = leo.replay_normalise_resize(c['in/canvas'], g[2:4], g[4:8], bool(g[8]), m['side'], eps=m['eps'])
    assert np.array_equal(got, frames_to_channels(c['out/labels'])), np.abs(got - frames_to_channels(c['out/labels'])).max()


@pytest.mark.parametrize('name', CASES)
def test_product_host_label_path_is_the_reference(name):
    c = case(name)
    m = c['meta']
    g = [int(v) for v in c['in/geom']]
    T, J = m['T'], m['J']
    canvas = c['in/canvas']
    if m['kind'] == 'raster':
        # the product's rasteriser draws the same canvas as the one the reference pipeline consumed
        for t in range(T):
            pose = c['in/pose'][t, :int(c['in/n_vals'][t])]
            hm, valid = cof.pose_to_heatmap(pose, g[0], g[1], max(200, m['side']), out_channels=J,
                                            marker_wd_ratio=m['marker_wd_ratio_cfg'], do_gauss_blur=False)
            assert np.array_equal(hm, canvas[:, :, t * J:(t + 1) * J])
            assert np.array_equal(valid, c['out/valid'][t])
    if m['raises']:
        with pytest.raises(cof.ApaError):
            cof.pose_label_replay_resize(canvas, g[2:4], g[4:8], bool(g[8]), m['side'], eps=m['eps'])
        return
    got = cof.pose_label_replay_resize(canvas, g[2:4], g[4:8], bool(g[8]), m['side'], eps=m['eps'])
    assert np.array_equal(got, frames_to_channels(c['out/labels']))


def test_frames_are_normalised_jointly():
    """:201-202 run on the channel-concatenated canvases: min / max are taken over ALL frames.  On non-binary
    canvases a per-frame normalisation gives a different answer (so the product call must see T*J channels);
    on the binary canvases of the real rasteriser both agree."""
    c = case('rand_video_joint_minmax')
    g = [int(v) for v in c['in/geom']]
    J = c['meta']['J']
    per_frame = np.stack([cof.pose_label_replay_resize(c['in/canvas'][:, :, t * J:(t + 1) * J], g[2:4], g[4:8],
                                                       bool(g[8]), 15) for t in range(2)])
    assert not np.array_equal(per_frame, c['out/labels'])
    c = case('raster_video_3frames')
    g = [int(v) for v in c['in/geom']]
    J = c['meta']['J']
    per_frame = np.stack([cof.pose_label_replay_resize(c['in/canvas'][:, :, t * J:(t + 1) * J], g[2:4], g[4:8],
                                                       bool(g[8]), 15) for t in range(3)])
    assert np.array_equal(per_frame, c['out/labels'])


@pytest.mark.regen
def test_generator_reproduces_the_committed_label_fixtures():
    import importlib.util
    import sys
    saved = dict(sys.modules)
    saved_path = list(sys.path)
    try:
        spec = importlib.util.spec_from_file_location('make_label_reference',
                                                      os.path.join(GOLD, 'make_label_reference.py'))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        blobs = gen.generate()
        assert set(blobs) == set(Z.files)
        for k in Z.files:
            if k.endswith('/meta') or k == 'cases':
                assert json.loads(str(blobs[k])) == json.loads(str(Z[k])), k
            else:
                assert np.array_equal(blobs[k], Z[k]), k
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
