"""Frame sampler of the training loop - counterpart of lib/roi_data_layer/layer.py:20-86: a permutation of the roidb
(numpy global RNG, so ``np.random.seed`` / the snapshot's RNG state reproduce it), a cursor, one frame per minibatch;
the permutation is redrawn when the cursor would run past the end.  ``random=True`` (validation) draws its permutations
from the wall clock and leaves the global RNG stream untouched."""
import time

import numpy as np

from ..model.config import cfg
from .minibatch import get_minibatch


class RoIDataLayer(object):
    def __init__(self, roidb, num_classes, mode, random=False):
        self._roidb, self._num_classes, self._mode, self._random = roidb, num_classes, mode, random
        self._cnt = 0
        self._shuffle_roidb_inds()

    def _shuffle_roidb_inds(self):
        n = len(self._roidb)
        if self._random:
            keep = np.random.get_state()
            np.random.seed(int(round(time.time() * 1000)) % 4294967295)
            self._perm = np.random.permutation(np.arange(n))
            np.random.set_state(keep)
        else:
            self._perm = np.random.permutation(np.arange(n))
        self._cur = 0

    def _get_next_minibatch_inds(self):
        step = int(cfg.TRAIN.FRAMES_PER_BATCH)
        if self._cur + step >= len(self._roidb):
            self._shuffle_roidb_inds()
        inds = self._perm[self._cur:self._cur + step]
        self._cur += step
        return inds

    def _get_next_minibatch(self, augment_en):
        blobs = None
        while blobs is None:                      # frames without usable ground truth are skipped (:66-82)
            entries = [self._roidb[i] for i in self._get_next_minibatch_inds()]
            blobs = get_minibatch(entries, self._num_classes, augment_en, self._cnt)
            self._cnt += 1
        return blobs

    def forward(self, augment_en):
        return self._get_next_minibatch(augment_en)
