"""WSI-sharded data parallelism: one process per GPU, graphs sharded across ranks, ONE flat fp32
gradient all-reduce per step over RCCL/xGMI.

The reference has no distributed training at all (single device, trainer/trainer.py:32-34);
this is the build's own design for the 8-GPU node.  WSI graphs are independent (no cross-graph edges),
so the only exchange is the parameter-gradient sum.  After ``backward`` the freshly computed gradients are
packed into one contiguous fp32 buffer with a single multi-tensor copy, ONE blocking ``all_reduce`` of that
buffer (36 MB for HEATNet4 with 3 node types) replaces per-parameter collectives, and every ``.grad`` is
re-pointed at its slice of the buffer so the optimizer reads the averaged values in place — on MI355X's
point-to-point xGMI fabric one large collective keeps all 7 links busy, many small ones are latency-bound.
With ``arm()`` before ``backward`` the buffer is reduced in a few (default 4) contiguous pieces: a piece is launched
asynchronously from a post-accumulate-grad hook as soon as every parameter in it holds its gradient (the head and the last
layer finish first; always in the same order on every rank), so most of the ~3 % of a step the collective costs at 8 GPUs hides under the rest of backward; the piece
holding the first parameters carries the used-flags and goes last, from ``all_reduce_mean()``.  Without ``arm()`` it is ONE
blocking collective.  Every rank ends up with the same bits either way; between the two forms the results agree bit for bit on two ranks and to
fp32 summation order beyond (a ring / tree all-reduce adds each element's 8 contributions in an order that depends on the element's position in
the buffer it is handed: found by running the tests at world size 8, tests/test_dist.py).  Backend-agnostic (``nccl`` = RCCL on ROCm, ``gloo``
in the CPU tests).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_assignment(n_items: int, world_size: int, weights: Optional[Sequence[float]] = None) -> List[int]:
    """Rank of every item.  ``weights`` None: round-robin (item i -> rank i % world_size).  With weights (the slides' edge counts - or node
    counts - : real slides span 10^3..10^4 patches, construct_graph/graph_constructor.py:267, and a step costs what its edges cost):
    longest-processing-time-first under an equal-count constraint - items by decreasing weight (ties: lower index first), each to the currently
    lightest rank that still has room for one of its ceil(n / world_size) items (ties: lower rank).  Every rank keeps the SAME number of graphs
    per epoch (+- 1), so steps stay aligned across ranks, while the per-rank edge totals are equalised.  Deterministic and a pure function of
    its arguments: every rank computes the same table without communicating."""
    if world_size <= 0:
        raise ValueError("shard: world_size must be positive")
    if weights is None:
        return [i % world_size for i in range(n_items)]
    if len(weights) != n_items:
        raise ValueError("shard: one weight per item")
    cap_hi = -(-n_items // world_size)
    n_hi = n_items - (cap_hi - 1) * world_size if cap_hi > 0 else 0       # how many ranks hold cap_hi items (the others one fewer)
    load = [0.0] * world_size
    held = [0] * world_size
    full_hi = 0                                                            # ranks that already hold cap_hi items
    out = [0] * n_items
    for i in sorted(range(n_items), key=lambda j: (-float(weights[j]), j)):
        best = -1
        for r in range(world_size):
            room = cap_hi if full_hi < n_hi else cap_hi - 1                # once n_hi ranks are at cap_hi the rest stop one short
            if held[r] >= room:
                continue
            if best < 0 or load[r] < load[best]:
                best = r
        out[i] = best
        load[best] += float(weights[i])
        held[best] += 1
        if held[best] == cap_hi:
            full_hi += 1
    return out


def shard(items: Sequence, rank: int, world_size: int, weights: Optional[Sequence[float]] = None) -> List:
    """This rank's share of a list of WSI graphs (or file names): round-robin, or - with ``weights`` (e.g. ``[g.num_edges() for g in graphs]``) -
    size-balanced (``shard_assignment``).  The order of ``items`` is kept inside a shard."""
    owner = shard_assignment(len(items), world_size, weights)
    return [x for x, r in zip(items, owner) if r == rank]


def shard_imbalance(weights: Sequence[float], owner: Sequence[int], world_size: int) -> float:
    """max over ranks of the rank's total weight / the mean over ranks (1.0 = perfectly even): how much longer the slowest rank's epoch is
    than it had to be."""
    load = [0.0] * world_size
    for w, r in zip(weights, owner):
        load[r] += float(w)
    mean = sum(load) / world_size
    return max(load) / mean if mean > 0 else 1.0


class GradBucket:
    """Flat fp32 gradient buffer for ONE all-reduce per step.

    Usage per step:  ``optimizer.zero_grad(set_to_none=True)`` -> forward/backward -> ``all_reduce_mean()``.

    Which parameters receive a gradient depends on the batch: a HEAT/HGT layer only touches the projections of the
    node types and relations PRESENT in the batch, and the loader emits batches of different schemas.  The rule that
    keeps every rank's parameters identical and matches the single-process semantics (an optimizer skips
    ``grad is None``):

    * a parameter used by ANY rank in this step gets the rank-averaged gradient on EVERY rank (ranks that did not use
      it contribute zeros);
    * a parameter used by NO rank keeps ``grad = None`` on every rank.

    To decide the second case without a collective of its own, the flat buffer carries one "used" flag per parameter
    behind the gradients, summed by the same all-reduce.  A rank reads the flags back (one small device->host copy,
    a host sync) only in a step in which it has itself left a bucket parameter unused; in the steady state — every
    bucket parameter used locally — no rank synchronises.  Parameters the architecture never reaches
    (``model.dead_parameter_names()``, e.g. HEATNet4's ``gcs.{l}.weight``, models/HEATNet4.py:54) stay out of the
    bucket so they do not force that read-back every step."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, pieces: int = 4, overlap: bool = True,
                 single_rank_collectives: bool = False):
        """``overlap``: ``arm()`` launches pieces of the all-reduce from hooks while backward runs (False: ``arm()`` is a no-op and
        ``all_reduce_mean`` is ONE blocking collective - same sums, bit for bit).  ``single_rank_collectives``: at world size 1 the bucket
        normally does nothing at all; with this flag it runs its whole path - hooks, packing, the collectives on the backend's stream, the flag
        piece, re-pointing ``.grad`` - against a one-rank group (tests / measurements of RCCL on a 1-GPU box: sums over one rank are the values)."""
        self.overlap = bool(overlap)
        self.single_rank_collectives = bool(single_rank_collectives)
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters to bucket")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        nflag = len(self.params)
        self.numel = total
        # [flags | gradients]: the flags sit next to the FIRST parameters, whose gradients arrive last in backward - the piece
        # that is reduced last carries them (they are only known once backward is over)
        self._buf = torch.zeros(nflag + total, dtype=torch.float32, device=dev)
        self.flags = self._buf[:nflag]                # per parameter: number of ranks that produced a gradient
        self.flat = self._buf[nflag:]                 # the gradients
        self.group = process_group
        self.views = []
        offs = [0]
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[offs[-1]:offs[-1] + n].view_as(p))
            offs.append(offs[-1] + n)
        # contiguous parameter ranges of roughly equal size: piece 0 = [flags | first parameters], launched last
        pieces = max(1, min(int(pieces), len(self.params)))
        self._piece_lo = [0]
        for b in range(1, pieces):
            tgt = total * b // pieces
            i = min(range(len(offs)), key=lambda j: abs(offs[j] - tgt))
            if i > self._piece_lo[-1] and i < len(self.params):
                self._piece_lo.append(i)
        self._piece_hi = self._piece_lo[1:] + [len(self.params)]
        self._piece_of = [0] * len(self.params)
        for b, (lo, hi) in enumerate(zip(self._piece_lo, self._piece_hi)):
            for i in range(lo, hi):
                self._piece_of[i] = b
        self._piece_slice = [self._buf[(0 if b == 0 else nflag + offs[lo]):nflag + offs[hi]]
                             for b, (lo, hi) in enumerate(zip(self._piece_lo, self._piece_hi))]
        self._armed = False
        self._hooks = None
        self._ready: List[int] = []
        self._handles: List = []
        self._next = -1
        self._flags_uploaded: Optional[List[bool]] = None
        self.flag_readbacks = 0                       # host syncs taken so far (diagnostics / tests)
        self.overlapped_pieces = 0                    # pieces launched from a hook so far (diagnostics / tests)

    @classmethod
    def from_model(cls, model: torch.nn.Module, process_group=None, **kw) -> "GradBucket":
        """Every trainable parameter except those the architecture never reaches (``model.dead_parameter_names()``)."""
        dead = set(model.dead_parameter_names()) if hasattr(model, "dead_parameter_names") else set()
        return cls([p for n, p in model.named_parameters() if n not in dead], process_group, **kw)

    @classmethod
    def from_used_parameters(cls, model: torch.nn.Module, process_group=None, **kw) -> "GradBucket":
        """Bucket only the parameters that hold a gradient right now (call after one probe backward).  Correct ONLY when
        every later batch on every rank uses exactly the same parameters (one fixed graph schema): a parameter outside
        the bucket is never reduced.  ``from_model`` has no such condition."""
        used = [p for p in model.parameters() if p.grad is not None]
        return cls(used, process_group, **kw)

    def zero(self) -> None:
        for p in self.params:
            p.grad = None

    def world_size(self) -> int:
        if not dist.is_available() or not dist.is_initialized():
            return 1
        return dist.get_world_size(self.group)

    def check_outside(self, all_params: Iterable[torch.nn.Parameter]) -> None:
        """Raise if a parameter outside the bucket holds a gradient: it would never be reduced and the ranks would drift."""
        mine = {id(p) for p in self.params}
        for p in all_params:
            if p.grad is not None and id(p) not in mine:
                raise RuntimeError("a parameter outside the GradBucket received a gradient: build the bucket with "
                                   "GradBucket.from_model (all trainable parameters) when batches differ in schema")

    # ---------------------------------------------------------------------------------------- overlap with backward
    def arm(self) -> None:
        """Call before ``backward`` (after ``zero_grad(set_to_none=True)``): pieces of the buffer are then all-reduced
        asynchronously as soon as their gradients exist.  No-op for a single rank or a bucket built with ``overlap=False``.
        Valid for EXACTLY ONE backward pass before ``all_reduce_mean``: a piece is packed and sent when its last gradient of that
        pass arrives, so a second pass (gradient accumulation over micro-batches) would add to ``.grad`` behind a piece already
        on the wire - the hook raises instead of dropping that contribution.  Accumulate without ``arm()`` (the blocking path
        packs everything at the end)."""
        if (self.world_size() == 1 and not self.single_rank_collectives) or not self.overlap or len(self._piece_lo) == 1:
            return
        # the hooks below pack gradients DURING backward: every weight-gradient launch must stay on the autograd stream (a gradient written by the
        # side stream of ops._gemm_tn_background would be packed and sent before it exists); released in all_reduce_mean / disarm
        from . import ops
        ops.block_background_weight_gradients(True, who="bucket")
        if self._hooks is None:
            # first use: bring the communicator up from THIS thread (the pieces are launched from autograd's worker thread; every
            # rank arms at the same point of its step, so this blocking one-element collective lines up)
            dist.all_reduce(torch.zeros(1, dtype=torch.float32, device=self.flat.device), group=self.group)
            self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]
        n = len(self._piece_lo)
        self._ready = [0] * n
        self._handles = [None] * n
        self._next = n - 1                            # collectives must be issued in the same order on every rank: n-1, ..., 1, 0
        self._armed = True

    def disarm(self) -> None:
        """Leave the armed state (``all_reduce_mean`` does; call it yourself when a backward pass raised between ``arm()`` and it): the
        hooks go quiet and the background weight-gradient path is released."""
        self._armed = False
        from . import ops
        ops.block_background_weight_gradients(False, who="bucket")

    def _make_hook(self, i: int):
        def hook(_param):
            if not self._armed:
                return
            if self._piece_of[i] > self._next:
                raise RuntimeError("GradBucket: a gradient arrived for a piece of the buffer that is already being all-reduced "
                                   "(a second backward between arm() and all_reduce_mean()); accumulate over micro-batches "
                                   "without arm()")
            self._ready[self._piece_of[i]] += 1
            # launch every complete piece at the head of the fixed order (a piece with a parameter this batch does not use never
            # completes: it and everything behind it wait for all_reduce_mean)
            while self._next >= 1 and self._ready[self._next] == self._piece_hi[self._next] - self._piece_lo[self._next]:
                self._launch(self._next, from_hook=True)
                self._next -= 1
        return hook

    def _pack(self, lo: int, hi: int) -> None:
        """Copy the gradients of parameters [lo, hi) into their slices (zeros where a parameter got none)."""
        grads = []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is not None:
                grads.append(p.grad)
            else:
                v.zero_()
                grads.append(v)
        torch._foreach_copy_(self.views[lo:hi], grads)

    def _launch(self, b: int, from_hook: bool = False) -> None:
        self._pack(self._piece_lo[b], self._piece_hi[b])
        self._handles[b] = dist.all_reduce(self._piece_slice[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if from_hook:
            self.overlapped_pieces += 1

    def all_reduce_mean(self) -> None:
        """Average gradients over ranks (global-batch mean when every rank holds the same batch size)."""
        ws = self.world_size()
        if ws == 1 and not self.single_rank_collectives:
            return
        used = [p.grad is not None for p in self.params]
        if used != self._flags_uploaded:              # the flags change only when the batch schema does
            from .graph import host_to_device
            self._local_flags = host_to_device([1.0 if u else 0.0 for u in used], torch.float32, self.flat.device)
            self._flags_uploaded = used
        self.flags.copy_(self._local_flags)
        if self._armed:
            # the rest of the fixed order: pieces behind one with an unused parameter, and piece 0 (it carries the flags)
            while self._next >= 0:
                self._launch(self._next)
                self._next -= 1
            for h in self._handles:
                h.wait()
            self.disarm()
        else:
            self._pack(0, len(self.params))
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / ws)
        if all(used):
            for p, v in zip(self.params, self.views):
                p.grad = v
            return
        anyone = (self.flags > 0.5).tolist()          # host sync, only on a rank that skipped a bucket parameter itself
        self.flag_readbacks += 1
        for p, v, u, a in zip(self.params, self.views, used, anyone):
            p.grad = v if (u or a) else None
