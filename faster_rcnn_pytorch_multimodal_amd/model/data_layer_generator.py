"""``data_layer_generator`` - the frame source ``SolverWrapper.train_model`` pulls from (lib/model/data_layer_generator.py;
created at lib/model/train_val.py:309-310, driven through ``start / next / get_pointer / set_pointer / kill / join``).

The reference fills a queue from a background process with numpy blobs.  Here a blob is produced by device kernels
(``frcnn_prep_image`` / ``frcnn_bev_voxelize``) on the caller's stream, and the library is main-thread only
(include/frcnn_hip.h), so ``next()`` produces the frame on demand; the pointer protocol the snapshots rely on
(lib/model/train_val.py:100-165) is the same."""
from ..roi_data_layer.layer import RoIDataLayer


class data_layer_generator(object):
    def __init__(self, mode='train', roidb=None, augment_en=False, num_classes=0):
        self.data_layer = RoIDataLayer(roidb, num_classes, mode, random=(mode == 'val'))
        self._augment_en = bool(augment_en)
        self._cur, self._perm = self.data_layer._cur, self.data_layer._perm
        self.finished = False

    def set_pointer(self, cur_val, perm_val):
        self.data_layer._cur = self._cur = cur_val
        if perm_val is not None:
            self.data_layer._perm = perm_val
        self._perm = self.data_layer._perm

    def get_pointer(self):
        return self._cur, self._perm

    def start(self):
        pass

    def kill(self):
        self.finished = True

    def join(self):
        pass

    def clear(self):
        pass

    def next(self):
        blobs = self.data_layer.forward(self._augment_en)
        self._cur, self._perm = self.data_layer._cur, self.data_layer._perm
        return blobs

    __next__ = next

    def __iter__(self):
        return self
