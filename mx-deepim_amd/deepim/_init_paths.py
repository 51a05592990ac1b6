"""Same role as the reference's deepim/_init_paths.py: put the package root on sys.path for `python deepim/test.py`."""
import os
import sys

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for _p in (_ROOT, os.path.join(_ROOT, "..")):
    _p = os.path.abspath(_p)
    if _p not in sys.path:
        sys.path.insert(0, _p)
