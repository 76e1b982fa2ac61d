"""Readout poolings with the reference's ``pooling/*`` API: ``Pool().forward(graph, feat, ntype=None) -> [B, D]``
(pooling/avg_pooling.py:11-19, sum_pooling.py:10-18, max_pooling.py:11-19, nt_pooling.py:4-10)."""
from .readout import AvgPooling, SumPooling, MaxPooling, NTPooling, GlobalAttentionPooling  # noqa: F401
from .ASAP import ASAPPooling  # noqa: F401  (commented out in the reference's pooling/__init__.py:1,7)

__all__ = ["AvgPooling", "SumPooling", "MaxPooling", "NTPooling", "GlobalAttentionPooling", "ASAPPooling"]
