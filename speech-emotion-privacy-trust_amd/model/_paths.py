"""Makes `sept_amd` importable whether the reference-style flat import
(`sys.path.append('../model'); import baseline_models`) or the package import
(`from model.baseline_models import ...`) is used."""
import os
import sys

_PKG = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
