"""The optimizer step of the reference's trainer on the GPU in ONE launch.

``parser.parse_optimizer`` (parser.py:33-38) builds ``torch.optim.Adam(model.parameters(), lr, weight_decay)`` and
``trainer/train_gnn.py:72`` calls ``optimizer.step()`` after ``loss.backward()``.  :class:`Adam` is that optimizer with the same
arithmetic (L2 penalty added to the gradient, bias-corrected moments, ``amsgrad=False``) and the same ``state_dict`` layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter: checkpoints move between the two), stepping every parameter of a group through
``wsi_adam_step`` (csrc/optim.hip): one kernel, 28 bytes of HBM traffic per element, where torch's fused path takes two launches at half
the bandwidth on this model's 54 tensors.  GPU only - there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Iterable

import torch

from . import _native as N


class Adam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = N.load()
        from . import ops
        ops._background_recover()       # a backward pass that raised never joined its side stream: order this step behind it (no-op otherwise)
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("wsi_hgnn_amd.optim.Adam steps contiguous fp32 parameters on the GPU only (no CPU path)")
                if p.grad.is_sparse:
                    raise RuntimeError("wsi_hgnn_amd.optim.Adam does not take sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                # the count is a plain Python number (a state loaded from torch.optim.Adam brings a tensor: converted once; torch converts back
                # when it loads ours).  54 host-tensor updates per step, as torch keeps them, were the one thing in this loop that touched
                # the CPU tensor machinery - and with it, once in ~30 steps, an 80 ms stall of the whole process on a CPU-quota'd box.
                t = int(st["step"]) + 1
                st["step"] = t
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(t, []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            for t, items in by_step.items():            # (one launch when every parameter has been stepped equally often - the usual case)
                # the descriptor table is kept between steps and only the gradient pointers are refreshed: p / m / v never move, and a fresh
                # table of 54 structs per step is allocation churn that brings Python's cyclic collector round sooner (an 80 ms pause)
                key = tuple(id(p) for p, _, _, _ in items)
                tables = self.__dict__.setdefault("_wsi_tables", {})          # (on the optimizer, not in param_groups: state_dict() stays plain data)
                cached = tables.get(gi)
                if cached is None or cached[0] != key:
                    arr = (N.AdamTensor * len(items))(*[N.AdamTensor(p.data_ptr(), 0, m.data_ptr(), v.data_ptr(), p.numel()) for p, _, m, v in items])
                    cached = tables[gi] = (key, arr, ctypes.cast(arr, ctypes.c_void_p))
                _, arr, arr_p = cached
                for i, (p, g, m, v) in enumerate(items):
                    a = arr[i]
                    a.p, a.g, a.m, a.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                N.check(lib.wsi_adam_step(arr_p, len(items), float(group["lr"]), float(group["betas"][0]),
                                          float(group["betas"][1]), float(group["eps"]), float(group["weight_decay"]), t, N.stream()), "wsi_adam_step")
                # the kernel wrote p / m / v through raw pointers: move their version counters as torch.optim.Adam's in-place ops would, so that
                # autograd's "modified by an inplace operation" check and every version-keyed fact about these tensors (ops._annotate) see the step
                torch.autograd.graph.increment_version([t_ for p, _, m, v in items for t_ in (p, m, v)])
        ops.repack_weights()            # the packed fp16 planes of the weights the projections read (ops._PACKED): all of them in one launch per op
        return loss
