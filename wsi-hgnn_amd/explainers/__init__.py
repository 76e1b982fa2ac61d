"""Leave-one-node-out explainers of the reference (``explainers/GEM.py``, ``explainers/gem_het.py``) on the batched engine."""
from .gem import GemExplainer, HetGemExplainer  # noqa: F401

__all__ = ["GemExplainer", "HetGemExplainer"]
