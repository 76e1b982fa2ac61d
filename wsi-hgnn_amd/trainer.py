"""The caller of the hot path: one optimisation step, mirroring ``GNNTrainer.train_one_step``
(trainer/train_gnn.py:55-79) on the MI355X path.

Differences from the reference, all on the caller side of the same semantics:
  * a tuple/list of heterogeneous graphs is block-diagonally batched and run as ONE forward (the reference loops
    ``[self.gnn(g) for g in graphs]`` and concatenates, :59-62) — identical logits, one set of launches;
  * with ``torch.distributed`` initialised, gradients are averaged over ranks with one flat RCCL all-reduce
    (``dist.GradBucket``) between ``backward`` and ``optimizer.step``;
  * loss / accuracy stay on the device unless ``sync=True`` (the reference's ``.item()`` / ``.cpu().numpy()`` at :73-79
    force a host sync every step).
The training loop, datasets, checkpointing and evaluators around it are out of scope.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch
import torch.nn.functional as F

from .dist import GradBucket
from .graph import HeteroGraph, batch as batch_graphs


def acc(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """utils.py ``acc``: fraction of argmax hits (kept on the device)."""
    return (pred.argmax(dim=1) == label).to(torch.float32).mean()


def apply_loss(loss_fcn, pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """``loss_fcn(pred, label)``; the reference's classification loss - ``nn.CrossEntropyLoss()`` with its defaults, parser.py:182-183 - on GPU logits
    runs as ONE launch forward and one backward (``ops.cross_entropy``: the same arithmetic) instead of torch's four kernels and two fills."""
    if (type(loss_fcn) is torch.nn.CrossEntropyLoss and loss_fcn.weight is None and loss_fcn.reduction == "mean" and loss_fcn.ignore_index == -100
            and loss_fcn.label_smoothing == 0.0 and pred.is_cuda and pred.dim() == 2 and pred.dtype == torch.float32 and label.dtype == torch.int64
            and label.dim() == 1 and pred.numel() <= 65536):
        from . import ops
        return ops.cross_entropy(pred, label)
    return loss_fcn(pred, label)


def train_one_step(gnn: torch.nn.Module, optimizer: torch.optim.Optimizer, loss_fcn, graphs: Union[HeteroGraph, Sequence[HeteroGraph]],
                   label: torch.Tensor, device, bucket: Optional[GradBucket] = None, sync: bool = True):
    """trainer/train_gnn.py:55-79.  Returns (loss, accuracy, pred, prob, label) — python float / numpy arrays when
    ``sync`` (as the reference), device tensors otherwise."""
    optimizer.zero_grad(set_to_none=True)                           # :56 (every parameter, inside or outside a bucket)
    label = label.to(device)                                        # :57
    if isinstance(graphs, (tuple, list)):                           # :59-62 heterogeneous graphs arrive as a tuple
        gs = [x.to(device) for x in graphs]
        same = all(x.ntypes == gs[0].ntypes and x.canonical_etypes == gs[0].canonical_etypes for x in gs[1:])
        if same:                                                    # one block-diagonal batch, one forward
            pred = gnn(batch_graphs(gs) if len(gs) > 1 else gs[0])
        else:                                                       # dgl.to_heterogeneous keeps only the relations that occur, so
            pred = torch.cat([gnn(x) for x in gs])                  # slides may differ in schema: per-graph forward as :61
    else:
        pred = gnn(graphs.to(device))                               # :64-65
    prob = F.softmax(pred, dim=1)                                   # :67
    loss = apply_loss(loss_fcn, pred, label)                        # :68
    if bucket is not None:
        bucket.arm()                                                # data parallel: reduce pieces of the gradient while backward runs
    loss.backward()                                                 # :70
    if bucket is not None:
        if bucket.world_size() > 1:
            bucket.check_outside(p for grp in optimizer.param_groups for p in grp["params"])
        bucket.all_reduce_mean()
    optimizer.step()                                                # :71
    accuracy = acc(pred, label)                                     # :73
    if not sync:
        return loss.detach(), accuracy, pred.detach().argmax(dim=1), prob.detach(), label
    bad = getattr(loss, "_wsi_bad_label", None)                      # ops.cross_entropy: a label outside [0, C) other than ignore_index
    if bad is not None and int(bad.item()):
        raise RuntimeError("train_one_step: a label lies outside [0, num_classes) (torch.nn.CrossEntropyLoss would trip its device assert)")
    return (loss.item(), float(accuracy.item()), pred.detach().cpu().numpy().argmax(axis=1),                # :75-79
            prob.detach().cpu().numpy(), label.detach().cpu().numpy())


class CapturedStep:
    """One training step (forward + loss + backward + optimizer) on a RESIDENT batch, captured once into a hipGraph and replayed.

    The reference trains slide by slide (``trainer/train_gnn.py:48-79``): the same few hundred graphs every epoch, each a step whose ~100 kernel
    launches the host takes longer to issue than the GPU to run (one 5k-node BRCA-shaped graph under HEATNet2: 1.80 ms eager, 0.83 ms replayed -
    ``tools/graph_capture_probe.py``).  Everything the step launches goes to the capturing stream - the library keeps no stream or state of its
    own, never allocates or synchronises (``include/wsi_hgnn.h``), and the hub kernels' side stream forks from and joins that stream with events -
    so the whole step records as one graph.  What the capture bakes in: the graph's kernel plan and every shape, i.e. one ``CapturedStep`` per
    resident batch (captures may share a memory ``pool``); the optimizer must keep its step count on the device
    (``torch.optim.Adam(..., capturable=True)``).  A model stepped eagerly before is fine - as long as the caller holds no tensor of those
    steps' autograd graphs any more (a previous loss, logits).  At the benchmark's size the step is GPU-bound and replay changes nothing
    (6.87 vs 6.93 ms).

    >>> step = CapturedStep(model, torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True), torch.nn.CrossEntropyLoss(), G, labels)
    >>> for _ in range(epochs): loss = step()          # a device tensor, overwritten by the next replay
    """

    def __init__(self, gnn: torch.nn.Module, optimizer: torch.optim.Optimizer, loss_fcn, graph: HeteroGraph, label: torch.Tensor,
                 warmup: int = 3, pool=None):
        if not label.is_cuda:
            raise RuntimeError("CapturedStep: the batch and its labels must be resident on the GPU")
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise RuntimeError("CapturedStep: the optimizer must be capturable (torch.optim.Adam(..., capturable=True)): its step count has to "
                                   "live on the device, a host count would be frozen into the graph")
        # Train-mode dropout: the HEAT layers draw their masks as a function of (host seed + a DEVICE word, row, column) - ops.CounterDropout.  The
        # capture freezes the host seeds; the recorded step itself advances the word, so every replay drops other entries (forward and backward of a
        # replay the same ones).  Any other dropout (a module that draws a mask tensor from a host-seeded generator state) cannot be replayed.
        self.seed_base = None
        if gnn.training and any(isinstance(mod, torch.nn.Dropout) and mod.p > 0.0 for mod in gnn.modules()):
            from .models.heat_layer import HEATLayer
            owners = [m_ for m_ in gnn.modules() if any(isinstance(c, torch.nn.Dropout) and c.p > 0.0 for c in m_.children())]
            if not all(isinstance(m_, HEATLayer) and getattr(m_, "counter_dropout", False) for m_ in owners):
                raise RuntimeError("CapturedStep: the model draws dropout masks outside the HEAT layers' counter-based draw (train mode, p > 0); their "
                                   "generator state is a host value a capture would freeze - every replay would drop the same entries.  Step such a "
                                   "model eagerly")
            self.seed_base = torch.empty((), dtype=torch.int64).random_(-(1 << 31), 1 << 31).to(torch.int32).reshape(1).to(label.device)
        self.gnn, self.optimizer, self.loss_fcn, self.graph, self.label = gnn, optimizer, loss_fcn, graph, label
        # A model that has been stepped before keeps its AccumulateGrad nodes bound to the stream of that step for as long as ANYTHING keeps
        # its last autograd graph alive; such a node makes the capture synchronise with the default stream, which is invalid and, on this
        # ROCm, a segfault in capture_end rather than an error.  What this class can release it does: the gradients (the path itself keeps no
        # registry: what producers know about a tensor travels on the tensor, ops._annotate).  What it cannot: tensors of an earlier step the
        # CALLER still holds (a previous loss or logits) - drop them before building a CapturedStep (PyTorch's general rule for captures).
        optimizer.zero_grad(set_to_none=True)
        self.params = [p for group in optimizer.param_groups for p in group["params"] if p.requires_grad]
        side = torch.cuda.Stream(device=label.device)
        side.wait_stream(torch.cuda.current_stream(label.device))
        with torch.cuda.stream(side):                      # (plans, caches and allocator pools settle before the capture; these ARE steps)
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(label.device).wait_stream(side)
        torch.cuda.synchronize(label.device)
        self.cuda_graph = torch.cuda.CUDAGraph()
        try:
            # thread_local: only what THIS thread does during the capture can invalidate it (another thread's allocation or sync must not), and an
            # illegal call here surfaces as a Python error at that call instead of a broken capture found at capture_end
            with torch.cuda.graph(self.cuda_graph, pool=pool, capture_error_mode="thread_local"):
                self.loss = self._eager()
        except Exception as exc:
            self.cuda_graph = None
            raise RuntimeError("CapturedStep: the step could not be captured into a hipGraph (is a tensor of an earlier eager step's autograd "
                               "graph - a previous loss or logits - still alive in the caller?); run the step eagerly instead") from exc
        self.steps_taken = max(1, warmup)                  # (the capture records the step without executing it)

    def _eager(self) -> torch.Tensor:
        from . import ops
        self.optimizer.zero_grad(set_to_none=True)
        with ops.dropout_seed_base(self.seed_base):
            loss = self.loss_fcn(self.gnn(self.graph), self.label)
            grads = torch.autograd.grad(loss, self.params, allow_unused=True)   # (same gradients as loss.backward(); parameters the loss does not reach: None)
        for p, g in zip(self.params, grads):
            p.grad = g
        self.optimizer.step()
        if self.seed_base is not None:
            ops.advance_dropout_seed_base(self.seed_base)                       # part of the recorded step: the next replay draws other masks
        return loss.detach()

    def __call__(self) -> torch.Tensor:
        self.cuda_graph.replay()
        self.steps_taken += 1
        return self.loss

    def pool(self):
        """The memory pool of this capture, for further ``CapturedStep(..., pool=...)`` over other resident batches."""
        return self.cuda_graph.pool()
