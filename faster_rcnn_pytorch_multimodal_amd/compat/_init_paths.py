"""Drop-in for the reference's tools/_init_paths.py: puts THIS directory (the stub packages nets/, model/, layer_utils/,
utils/, roi_data_layer/, datasets/) on sys.path, where the reference puts its lib/."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
