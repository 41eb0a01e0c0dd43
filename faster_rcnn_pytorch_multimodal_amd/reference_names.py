"""The reference's import names for this package's modules.

The reference is a directory of top-level packages put on ``sys.path`` by ``tools/_init_paths.py`` (``lib/nets``,
``lib/model``, ``lib/layer_utils``, ``lib/utils``, ``lib/roi_data_layer``, ``lib/datasets``); its callers say
``from nets.imagenet import imagenet`` / ``from model.config import cfg`` (``tools/test_net.py:14-31``,
``tools/trainval_net.py:10-30``, and every module under ``lib/`` among themselves).  ``install()`` registers each module
of this package under that name IN ``sys.modules`` - the same module object, not a second copy, so there is one
``cfg``, one ``Network`` class, one loaded ``libfrcnn_hip.so``.

Two ways in:
  * ``compat/`` next to this file holds one stub package per reference name whose ``__init__`` calls ``install()``;
    pointing ``tools/_init_paths.py`` at that directory (ONE ``sys.path`` entry) makes the reference's import lines
    resolve unchanged;
  * ``import faster_rcnn_pytorch_multimodal_amd.reference_names as rn; rn.install()`` from code that already imports
    this package.
``uninstall()`` restores whatever the names pointed to before (tests).
"""
import importlib
import os
import pkgutil
import sys

PACKAGES = ('nets', 'model', 'layer_utils', 'utils', 'roi_data_layer', 'datasets')

_saved = None


def install():
    """Alias every module of the six mirrored packages under its reference name.  Idempotent."""
    global _saved
    if _saved is not None:
        return sorted(k for k in _saved)
    root = __name__.rsplit('.', 1)[0]
    saved = {}

    stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'compat') + os.sep

    def alias(name, module):
        old = sys.modules.get(name)
        if old is not None and os.path.abspath(getattr(old, '__file__', None) or '').startswith(stubs):
            old = None              # a compat/ stub in the middle of replacing itself: nothing to restore later
        saved[name] = old
        sys.modules[name] = module

    for pkg in PACKAGES:
        real = importlib.import_module(root + '.' + pkg)
        alias(pkg, real)
        for info in pkgutil.iter_modules(real.__path__):
            alias(pkg + '.' + info.name, importlib.import_module(real.__name__ + '.' + info.name))
    _saved = saved
    return sorted(saved)


def uninstall():
    global _saved
    if _saved is None:
        return
    for name, old in _saved.items():
        if old is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = old
    _saved = None
