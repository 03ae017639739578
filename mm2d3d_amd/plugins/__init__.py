"""Model plugins importable by the reference's bare names (``importlib.import_module("2d_net")``, train.py:522)."""
import inspect
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def install():
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    return HERE


def load_model(name: str, **kwargs):
    """ModelWrapper semantics (train.py:508-531): import by name, keep only the kwargs the constructor accepts."""
    install()
    mod = importlib.import_module(name)
    params = inspect.signature(mod.Model.__init__).parameters
    return mod.Model(**{k: v for k, v in kwargs.items() if k in params})
