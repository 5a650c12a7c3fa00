"""CPU oracle for the label generator and the eval consumers -- TEST INFRASTRUCTURE ONLY.

    PARITY STATUS: pinned from the raster canvas onwards, restated for the raster itself.
    * Everything AFTER the canvas -- the op's python wrapper (custom_ops_factory.py:20-28), crop / flip replay,
      uint8 -> float conversion, min-max normalisation, legacy-bilinear resize (preprocess_pipeline.py:21-45,
      150-214) -- is held to outputs of the reference's own train_preprocess_pipeline executed in the build
      container (tests/golden/make_label_reference.py, 14 cases, bit for bit: tests/test_label_reference_cpu.py).
    * The mAP functions at the end of this file are pinned too: tests/golden/map_reference.npz holds outputs of the
      reference's src/eval/cap_eval_utils.py / src/eval/utils.py (tests/golden/make_map_reference.py),
      reproduced to 1e-13 with identical tie order.
    * cv::circle / cv::GaussianBlur (pose_to_heatmap.cc:83-87) are OpenCV routines; OpenCV is absent from this
      image and pose_to_heatmap.cc cannot be compiled here.  Their behaviour is restated from OpenCV's published
      algorithm and checked against the one input the reference supplies
      (src/custom_ops/test/pose_to_heatmap_op_test.py:10-23, expected valid = [T]*5+[F]*11): that rule alone
      remains unpinned.

numpy + pure-Python loops (small sizes only).  Citations are relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------
# cv::circle(img, center, radius, Scalar(1), thickness=-1, LINE_8, shift=0)
#   OpenCV 2.4/3.x drawing.cpp `Circle(img, center, radius, color, fill=1)` -- integer midpoint
# --------------------------------------------------------------------------------------
def cv_filled_circle(img: np.ndarray, cx: int, cy: int, radius: int, value: float = 1.0) -> None:
    """In-place filled midpoint circle on a 2-D float array (rows = y, cols = x).

    Restates OpenCV's Circle(): dx = r, dy = 0, err = 0, plus = 1, minus = 2r-1; while dx >= dy
    fill the horizontal spans [cx-dx, cx+dx] on rows cy+-dy and [cx-dy, cx+dy] on rows cy+-dx
    (each clipped to the image), then step the midpoint error.  The result is NOT the Euclidean
    disc dx^2+dy^2<=r^2 (they differ on a few boundary pixels).
    """
    h, w = img.shape

    def hline(y: int, xa: int, xb: int) -> None:
        if 0 <= y < h:
            xa = max(xa, 0)
            xb = min(xb, w - 1)
            if xa <= xb:
                img[y, xa:xb + 1] = value

    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        y11, y12 = cy - dy, cy + dy
        y21, y22 = cy - dx, cy + dx
        x11, x12 = cx - dx, cx + dx
        x21, x22 = cx - dy, cx + dy
        hline(y11, x11, x12)
        hline(y12, x11, x12)
        hline(y21, x21, x22)
        hline(y22, x21, x22)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0          # OpenCV: mask = (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2


def _gauss_kernel_7() -> np.ndarray:
    """cv::getGaussianKernel(7, sigma<=0): OpenCV uses the fixed small-kernel table for
    ksize 7 when sigma <= 0: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]."""
    return np.array([0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125],
                    dtype=np.float32)


def cv_gaussian_blur_7(img: np.ndarray) -> np.ndarray:
    """cv::GaussianBlur(ch, ch, Size(7,7), 0) with BORDER_REFLECT_101 (pose_to_heatmap.cc:86-87)."""
    k = _gauss_kernel_7()
    h, w = img.shape

    def reflect101(i: int, n: int) -> int:
        if n == 1:
            return 0
        while i < 0 or i >= n:
            i = -i if i < 0 else 2 * (n - 1) - i
        return i

    tmp = np.zeros_like(img, dtype=np.float32)
    for x in range(w):
        acc = np.zeros(h, dtype=np.float32)
        for t in range(7):
            acc += k[t] * img[:, reflect101(x + t - 3, w)]
        tmp[:, x] = acc
    out = np.zeros_like(img, dtype=np.float32)
    for y in range(h):
        acc = np.zeros(w, dtype=np.float32)
        for t in range(7):
            acc += k[t] * tmp[reflect101(y + t - 3, h), :]
        out[y, :] = acc
    return out


def pose_to_heatmap(pose_label, im_ht: int, im_wd: int, out_wd: int, out_channels: int = 16,
                    marker_wd_ratio: float = 0.1, do_gauss_blur: bool = True):
    """PoseToHeatmapOp::Compute -- src/custom_ops/pose_to_heatmap.cc:35-96.

    pose_label: int64 [3*out_channels*n_people] as (x, y, is_visible) triples.
    Returns (heatmap float32 [out_ht, out_wd, out_channels], is_valid bool [out_channels]).
    """
    pose = [int(v) for v in pose_label]
    out_ht = int((im_ht * out_wd * 1.0) / im_wd)                           # :45
    nk = out_channels
    assert len(pose) % (3 * nk) == 0                                       # :49
    n_rects = len(pose) // (3 * nk)
    hm = np.zeros((out_ht, out_wd, nk), dtype=np.float32)
    valid = np.zeros((nk,), dtype=bool)
    elts = nk * 3
    # (int) out_wd * marker_wd_ratio_ : the cast binds to out_wd, the product is evaluated in
    # float and converted to int by the cv::circle(int radius) parameter  (:84)
    radius = int(np.float32(out_wd) * np.float32(marker_wd_ratio))

    def cdiv(a: int, b: int) -> int:            # C++ integer division truncates toward zero
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b >= 0) else -q

    for i in range(nk):
        ch = np.zeros((out_ht, out_wd), dtype=np.float32)
        for rid in range(n_rects):
            px = pose[rid * elts + i * 3]
            py = pose[rid * elts + i * 3 + 1]
            x = cdiv(px * out_wd, im_wd)                                   # :77
            y = cdiv(py * out_ht, im_ht)                                   # :78
            if px >= 0 and py >= 0:                                        # :80-81
                valid[i] = True
                cv_filled_circle(ch, x, y, radius, 1.0)                    # :83-85
                if do_gauss_blur:
                    ch = cv_gaussian_blur_7(ch)                            # :86-87 (after EVERY circle)
        hm[:, :, i] = ch                                                   # :90-94
    return hm, valid


def pose_to_heatmap_py_wrapper(*args, **kwargs):
    """custom_ops_factory.py:20-28: heatmap *= 255, cast to uint8 (truncation)."""
    hm, valid = pose_to_heatmap(*args, **kwargs)
    return (hm * np.float32(255.0)).astype(np.uint8), valid


def replay_normalise_resize(hm_u8: np.ndarray, orig_hw, crop_info, whether_flip: bool,
                            out_side: int, eps: float = 1e-14) -> np.ndarray:
    """src/preprocess_pipeline.py:21-45 (_replay_augmentation) followed by :195-214.

    hm_u8 uint8 [h,w,J]; crop_info = [offset_y, offset_x, crop_h, crop_w] in the coordinates of
    the orig_hw image (aug_info['image_shape']).  float32 arithmetic like TF:
      ratio = H_size / orig_size; start = to_int32(crop[:2]*ratio); size = to_int32(crop[2:]*ratio)
      slice, optional flip_left_right, convert_image_dtype (x * (1/255)),
      x -= min(x); x /= (max(x) + EPS); legacy bilinear resize to [out_side, out_side].
    """
    h, w, J = hm_u8.shape
    ratio_x = np.float32(w) / np.float32(orig_hw[1])
    ratio_y = np.float32(h) / np.float32(orig_hw[0])
    y0 = int(np.float32(crop_info[0]) * ratio_y)
    x0 = int(np.float32(crop_info[1]) * ratio_x)
    ch = int(np.float32(crop_info[2]) * ratio_y)
    cw = int(np.float32(crop_info[3]) * ratio_x)
    img = hm_u8[y0:y0 + ch, x0:x0 + cw, :]
    assert img.shape[:2] == (ch, cw), 'tf.slice out of range'
    if whether_flip:
        img = img[:, ::-1, :]
    img = img.astype(np.float32) * np.float32(1.0 / 255.0)
    img = img - img.min()
    img = img / (img.max() + np.float32(eps))
    out = np.zeros((out_side, out_side, J), dtype=np.float32)
    sy = np.float32(ch) / np.float32(out_side)
    sx = np.float32(cw) / np.float32(out_side)
    for oy in range(out_side):
        fy = np.float32(oy) * sy
        ylo = int(np.floor(fy)); yhi = min(ylo + 1, ch - 1); wy = fy - np.float32(ylo)
        for ox in range(out_side):
            fx = np.float32(ox) * sx
            xlo = int(np.floor(fx)); xhi = min(xlo + 1, cw - 1); wx = fx - np.float32(xlo)
            top = img[ylo, xlo] + (img[ylo, xhi] - img[ylo, xlo]) * wx
            bot = img[yhi, xlo] + (img[yhi, xhi] - img[yhi, xlo]) * wx
            out[oy, ox] = top + (bot - top) * wy
    return out


# --------------------------------------------------------------------------------------
# mAP -- src/eval/utils.py:4-16, src/eval/cap_eval_utils.py:55-109
# --------------------------------------------------------------------------------------
def voc_ap(rec: np.ndarray, prec: np.ndarray) -> float:
    """cap_eval_utils.py:92-109: area under the monotone (right-max) precision envelope."""
    mrec = np.concatenate(([0.0], rec.astype(np.float64), [1.0]))
    mpre = np.concatenate(([0.0], prec.astype(np.float64), [0.0]))
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    idx = np.where(mrec[1:] != mrec[:-1])[0] + 1
    ap = 0.0
    for i in idx:
        ap += (mrec[i] - mrec[i - 1]) * mpre[i]
    return float(ap)


def calc_pr_ovr_noref(counts: np.ndarray, out: np.ndarray):
    """cap_eval_utils.py:55-89: sort by score descending via np.argsort(out)[::-1] (so ties are
    broken by *descending original index* -- argsort is ascending-stable-ish quicksort in the
    reference; we use the same call), cumulative precision / recall, voc_ap."""
    counts = np.array(counts > 0, dtype=np.float32)
    ind = np.argsort(out)[::-1]
    score = out[ind].astype(np.float64)
    sortcounts = counts[ind].astype(np.float64)
    tp = sortcounts
    fp = np.where(sortcounts >= 1, 0.0, 1.0)
    P = np.cumsum(tp) / (np.cumsum(tp) + np.cumsum(fp))
    numinst = np.sum(counts)
    R = np.cumsum(tp) / numinst
    return P, R, score, voc_ap(R, P)


def compute_map(all_logits: np.ndarray, all_labels: np.ndarray):
    """eval/utils.py:4-16: per-class AP over classes with >=1 positive, mean."""
    aps = []
    for cid in range(all_logits.shape[1]):
        this_labels = (all_labels == cid).astype('float32')
        if np.sum(this_labels) == 0:
            continue
        aps.append(calc_pr_ovr_noref(this_labels, all_logits[:, cid])[3])
    return float(np.mean(aps)), aps


def eval_consumer(logits: np.ndarray, labels: np.ndarray):
    """src/eval.py:193-197, 303-306: predictions = argmax(logits,1); scores = softmax(logits,-1);
    accuracy = mean(argmax(scores) == labels); mAP = compute_map(scores, labels)."""
    z = logits.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    sm = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    acc = float(np.mean(sm.argmax(axis=1) == labels))
    return sm, acc, compute_map(sm, labels)[0]
