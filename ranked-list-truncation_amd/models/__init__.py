"""Plug-in surface of the hot path: the same names the reference exports from models/__init__.py:2-6
for the five in-scope model families, plus BiCut / MOECut / PLECut from the "next" row N4
(section 8f; models/__init__.py:1,7-8)."""
from .Bicut import BiCut
from .Choopy import Choopy
from .AttnCut import AttnCut
from .MtChoopy import MtChoopy
from .MtAttnCut import MtAttnCut
from .MMOECut import MMOECut
from .MOECut import MOECut
from .PLECut import PLECut

__all__ = ["Choopy", "AttnCut", "MtChoopy", "MtAttnCut", "MMOECut", "MOECut", "PLECut", "BiCut"]
