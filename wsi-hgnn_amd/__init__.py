"""wsi-hgnn_amd: MI355X-native hot path of HKU-MedAI/WSI-HGNN (HEAT message passing + readout).

Import as ``wsi_hgnn_amd`` (alias package at the repo root).  Layout:
  graph.py      HeteroGraph container + CSR/CSC kernel plan (replaces the DGLGraph argument)
  synthetic.py  synthetic WSI patch graphs of the BASELINE shapes
  csrc/         hand-written HIP kernels for gfx950 + the C-ABI (include/wsi_hgnn.h)
  _native.py    ctypes loader of csrc/libwsi_hgnn.so (fails loudly when missing)
  ops.py        torch.autograd.Function wrappers calling the C-ABI
  models/, pooling/   nn.Module mirror of the reference's models/* and pooling/* API
  dist.py       WSI-sharded data parallelism (RCCL gradient all-reduce)
"""
from .graph import HeteroGraph, GraphPlan, batch, to_homogeneous, permute_nodes, remove_nodes, locality_order, apply_locality_order  # noqa: F401
