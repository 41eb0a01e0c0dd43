"""Stub for the reference's top-level package ``layer_utils`` (lib/layer_utils): replaces itself in sys.modules by
faster_rcnn_pytorch_multimodal_amd.layer_utils and aliases its submodules, see ../../reference_names.py."""
import os
import sys

_site = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..'))
if _site not in sys.path:
    sys.path.append(_site)
from faster_rcnn_pytorch_multimodal_amd import reference_names  # noqa: E402

reference_names.install()
