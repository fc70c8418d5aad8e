"""Locates the native op layer (<package>/ops.py) whether ``dropin`` is imported as a sub-package
or has been put on ``sys.path`` directly."""
import importlib
import os
import sys

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG_NAME = os.path.basename(_PKG_DIR)


def _load():
    if _PKG_NAME in sys.modules:
        return importlib.import_module(_PKG_NAME + ".ops")
    parent = os.path.dirname(_PKG_DIR)
    if parent not in sys.path:
        sys.path.insert(0, parent)
    return importlib.import_module(_PKG_NAME + ".ops")


ops = _load()
