"""Base anchor table (A x 4, float64) for one feature-map cell.

Same values as the reference's lib/layer_utils/generate_anchors.py:41-105 — window (0,0,15,15) -> aspect
ratio enumeration with half-to-even ``np.round`` on sqrt(size/ratio) and on ws*ratio (:82-93) -> scale
enumeration (:96-105) -> rows ordered ratio-major, scale-minor (:49-53) — computed here in closed form
over a (ratio, scale) grid instead of by repeated window re-centring.  The table stays in float64; the
device kernel adds the shift grid in float64 and rounds once (frcnn_generate_anchors).
"""
import numpy as np


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    ratios = np.asarray(ratios, dtype=np.float64).reshape(-1, 1)   # (nr, 1)
    scales = np.asarray(scales, dtype=np.float64).reshape(1, -1)   # (1, ns)
    ctr = 0.5 * (base_size - 1)                                     # centre of the (0,0,bs-1,bs-1) window
    area = float(base_size) * float(base_size)
    ws = np.round(np.sqrt(area / ratios))                          # ratio step (integer-valued)
    hs = np.round(ws * ratios)
    # re-centring a (ws,hs) window on ctr keeps the centre: x_ctr = (ctr - (ws-1)/2) + (ws-1)/2
    x_ctr = (ctr - 0.5 * (ws - 1)) + 0.5 * (ws - 1)
    y_ctr = (ctr - 0.5 * (hs - 1)) + 0.5 * (hs - 1)
    half_w = 0.5 * (ws * scales - 1)                               # (nr, ns)
    half_h = 0.5 * (hs * scales - 1)
    table = np.stack((x_ctr - half_w, y_ctr - half_h, x_ctr + half_w, y_ctr + half_h), axis=-1)
    return table.reshape(-1, 4)
