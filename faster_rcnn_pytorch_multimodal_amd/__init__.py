"""MI355X-native (gfx950) Faster R-CNN hot path behind the Python API of
mathild7/faster_rcnn_pytorch_multimodal (``nets.network.Network`` and friends).

Every computation on the path is a hand-written HIP kernel in ``libfrcnn_hip.so`` (C ABI:
``include/frcnn_hip.h``); this package is the host side that mirrors the reference's interface.
There is no CPU execution path: without the built library (or without a GPU) the ops raise.
"""
__version__ = "0.1.0"
