"""Graph container for the WSI-HGNN hot path (replaces the DGLGraph the reference passes around).

The reference hands a ``dgl.DGLGraph`` to ``model.forward`` and only touches a small part of
its API: ``G.ntypes``, ``G.canonical_etypes``, ``G.nodes[t].data['feat']``
(models/HEATNet4.py:202), ``G.edata['sim']`` (models/HEATNet4.py:209), ``G.ndata`` /
``G.local_scope()`` (pooling/avg_pooling.py:12-13), ``batch_num_nodes(ntype)`` (behind
``dgl.readout.mean_nodes``), ``G.to(device)`` (trainer/train_gnn.py:60).  ``HeteroGraph``
offers exactly that surface, plus the device-side *plan* the HIP kernels consume:

* nodes of all types live in ONE type-major id space (global id = type offset + local id), so
  K/Q/V of every node type sit in one ``[N, 3*D]`` table in HBM and a kernel never branches on
  the node type;
* forward layout = two-level CSR by destination: ``node_seg[w] .. node_seg[w+1]`` are the
  relation slots of dst node ``w`` (one slot per canonical relation whose dst type is w's type,
  empty relations included — DGL's ``cross_reducer='mean'`` denominator),
  ``rowptr[seg] .. rowptr[seg+1]`` the in-edges of that (node, relation) segment;
* backward layout = CSC by source over the same edge numbering (``csc_eid`` points back into the
  CSR edge order), so gradients w.r.t. K/V are reduced without atomics.

Graph construction / pickles (construct_graph/, data.py) are out of scope; ``from_coo`` takes
plain COO tensors.
"""
from __future__ import annotations

import collections

import os

import contextlib
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch

CanonicalEType = Tuple[str, str, str]


class _Frame(dict):
    """Per-type feature dict (``G.nodes['0'].data`` / per-relation edge data)."""


class _NodeTypeView:
    __slots__ = ("data",)

    def __init__(self, frame: _Frame):
        self.data = frame


class _NodesAccessor:
    def __init__(self, g: "HeteroGraph"):
        self._g = g

    def __getitem__(self, ntype: str) -> _NodeTypeView:
        return _NodeTypeView(self._g._nframes[ntype])


class _NDataAccessor:
    """``G.ndata[key]``: tensor for a one-type graph, ``{ntype: tensor}`` otherwise (DGL semantics)."""

    def __init__(self, g: "HeteroGraph"):
        self._g = g

    def __getitem__(self, key: str):
        g = self._g
        if len(g.ntypes) == 1:
            return g._nframes[g.ntypes[0]][key]
        return {t: g._nframes[t][key] for t in g.ntypes if key in g._nframes[t]}

    def __setitem__(self, key: str, value) -> None:
        g = self._g
        if isinstance(value, dict):
            for t, v in value.items():
                g._nframes[t][key] = v
        else:
            if len(g.ntypes) != 1:
                raise ValueError("assigning a tensor to ndata needs a single node type; pass a dict")
            g._nframes[g.ntypes[0]][key] = value

    def __contains__(self, key: str) -> bool:
        return any(key in self._g._nframes[t] for t in self._g.ntypes)


class _EDataAccessor:
    """``G.edata[key]``: tensor for a one-relation graph, ``{canonical_etype: tensor}`` otherwise."""

    def __init__(self, g: "HeteroGraph"):
        self._g = g

    def __getitem__(self, key: str):
        g = self._g
        if len(g.canonical_etypes) == 1:
            return g._eframes[g.canonical_etypes[0]][key]
        return {r: g._eframes[r][key] for r in g.canonical_etypes if key in g._eframes[r]}

    def __setitem__(self, key: str, value) -> None:
        g = self._g
        if isinstance(value, dict):
            for r, v in value.items():
                g._eframes[r][key] = v
        else:
            if len(g.canonical_etypes) != 1:
                raise ValueError("assigning a tensor to edata needs a single relation; pass a dict")
            g._eframes[g.canonical_etypes[0]][key] = value

    def __contains__(self, key: str) -> bool:
        return any(key in self._g._eframes[r] for r in self._g.canonical_etypes)


def _resolve_device(device) -> torch.device:
    """torch.device with the implicit CUDA index made explicit ('cuda' -> 'cuda:<current>'), so devices compare by value."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        return torch.device("cuda", torch.cuda.current_device())
    return device


class GraphPlan:
    """Device-resident int32/fp32 index structures consumed by the HIP kernels (see module doc).

    All tensors live on ``device``.  Edge order "CSR" = sorted by (global dst, relation slot),
    stable in the input edge order; every per-edge tensor the kernels exchange uses that order.
    """

    def __init__(self):
        self.device = None
        self.num_nodes = 0          # N, all types
        self.num_edges = 0          # E, all relations
        self.num_segs = 0           # sum_t N_t * R_t
        self.type_off: List[int] = []   # [T+1] host ints, global id range of each ntype
        self.rel_slots: List[int] = []  # R_t per ntype (relations whose dst type is t)
        self.node_seg = None        # int32 [N+1]
        self.rowptr = None          # int32 [S+1]
        self.src = None             # int32 [E]   global src id, CSR order
        self.perm = None            # int64 [E]   CSR position -> position in the concatenated input edge list (None for
                                    #             loader-assembled plans: HeteroGraph._csr_perm recomputes it on demand)
        self.inv_rd = None          # fp32  [N]   1/R_t(type of node), 0 when R_t == 0
        self.colptr = None          # int32 [N+1] CSC by global src
        self.csc_eid = None         # int32 [E]   CSR edge id of the j-th CSC entry
        self.csc_dst = None         # int32 [E]   global dst of the j-th CSC entry
        self.order_dst = None       # int32 [N]   dst nodes, heaviest (most in-edges) first
        self.order_src = None       # int32 [N]   src nodes, heaviest (most out-edges) first
        self.num_heavy = 0          # leading entries of order_dst = the highest in-degree nodes (see wsi_heat_attn_fwd)
        self.locality = False       # orders follow the slides' locality positions ('_pos'): kernels walk them XCD-contiguously
        self.heavy_degree = HEAVY_DEGREE   # threshold the first `num_heavy` entries of order_dst were chosen with
        self.readout_ptr = None     # int32 [T*B+1] rows of (ntype t, graph b) = [ptr[t*B+b], ptr[t*B+b+1])
        self.batch_size = 1
        self.num_src_rows = 0       # rows of the k/v tables the CSC indexes: N, or sum_r N_src(r) (per-relation tables)
        self.rel_rows: List[Tuple[int, int]] = []   # per canonical relation: its row range in the per-relation tables
        self.graph_sizes: List[int] = []            # host: nodes (all types) of every graph of the batch - the graph-major pieces of the orders (attn_tiles)


class HeteroGraph:
    """Heterogeneous (or homogeneous) multi-relation graph, optionally a block-diagonal batch."""

    def __init__(
        self,
        num_nodes: "OrderedDict[str, int] | Dict[str, int]",
        edges: "OrderedDict[CanonicalEType, Tuple[torch.Tensor, torch.Tensor]]",
        batch_num_nodes: Optional[Dict[str, torch.Tensor]] = None,
    ):
        # DGL keeps node types and canonical relations sorted as STRINGS ('10' < '2'), whatever order they were given in
        # (dgl.heterograph / dgl.to_heterogeneous); HEATNet4 concatenates per-type blocks in G.ntypes order (HEATNet4.py:236-242),
        # so the order is part of the arithmetic.
        self._num_nodes = OrderedDict((k, int(num_nodes[k0])) for k, k0 in sorted((str(k), k) for k in num_nodes))
        pairs = []
        for (s, e, d), (u, v) in edges.items():
            s, e, d = str(s), str(e), str(d)
            if s not in self._num_nodes or d not in self._num_nodes:
                raise KeyError(f"relation {(s, e, d)} uses an unknown node type")
            u = torch.as_tensor(u, dtype=torch.int64)
            v = torch.as_tensor(v, dtype=torch.int64)
            if u.shape != v.shape or u.dim() != 1:
                raise ValueError("src/dst must be 1-D tensors of equal length")
            pairs.append(((s, e, d), (u, v)))
        pairs.sort(key=lambda kv: kv[0])
        self._edges_store: "Optional[OrderedDict[CanonicalEType, Tuple[torch.Tensor, torch.Tensor]]]" = OrderedDict(pairs)
        self._rels: List[CanonicalEType] = [k for k, _ in pairs]
        self._edge_thunk = None         # loader batches: per-relation COO (+ per-relation edge fields) built on first use
        self._nframes: Dict[str, _Frame] = {t: _Frame() for t in self._num_nodes}
        self._eframes: Dict[CanonicalEType, _Frame] = {r: _Frame() for r in self._rels}
        if batch_num_nodes is None:
            self._batch_num_nodes = None
        else:
            self._batch_num_nodes = {t: torch.as_tensor(batch_num_nodes[t], dtype=torch.int64).cpu()
                                     for t in self._num_nodes}
        self._plan: Optional[GraphPlan] = None

    @property
    def _edges(self) -> "OrderedDict[CanonicalEType, Tuple[torch.Tensor, torch.Tensor]]":
        """Per-relation COO.  A loader batch (data.GraphBatchLoader) carries only its assembled kernel plan; its COO and
        per-relation edge fields are an offset-concatenation of the stored graphs' edges, done here on first use."""
        if self._edges_store is None:
            thunk, self._edge_thunk = self._edge_thunk, None
            edges, efields = thunk()
            self._edges_store = OrderedDict((r, edges[r]) for r in self._rels)
            for r in self._rels:
                for k, x in efields.get(r, {}).items():
                    self._eframes[r].setdefault(k, x)
        return self._edges_store

    @classmethod
    def _from_plan(cls, num_nodes, rels, batch_num_nodes, plan, edge_thunk) -> "HeteroGraph":
        """Loader batch: schema + assembled plan now, COO later (see ``_edges``).  ``rels`` must be sorted."""
        g = cls(num_nodes, OrderedDict(), batch_num_nodes)
        g._rels = [tuple(r) for r in rels]
        g._eframes = {r: _Frame() for r in g._rels}
        g._edges_store = None
        g._edge_thunk = edge_thunk
        g._plan = plan
        return g

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_coo(cls, num_nodes, edges, feat=None, sim=None) -> "HeteroGraph":
        g = cls(num_nodes, edges)
        if feat is not None:
            for t, x in feat.items():
                g._nframes[str(t)]["feat"] = x
        if sim is not None:
            for r, x in sim.items():
                g._eframes[tuple(str(a) for a in r)]["sim"] = x
        return g

    @classmethod
    def homogeneous(cls, num_nodes: int, src, dst, feat=None) -> "HeteroGraph":
        g = cls(OrderedDict([("_N", num_nodes)]), OrderedDict([(("_N", "_E", "_N"), (src, dst))]))
        if feat is not None:
            g._nframes["_N"]["feat"] = feat
        return g

    # ------------------------------------------------------------------ DGL-like surface
    @property
    def ntypes(self) -> List[str]:
        return list(self._num_nodes.keys())

    @property
    def canonical_etypes(self) -> List[CanonicalEType]:
        return list(self._rels)

    @property
    def etypes(self) -> List[str]:
        return [r[1] for r in self._rels]

    @property
    def is_homogeneous(self) -> bool:
        return len(self._num_nodes) == 1 and len(self._rels) == 1

    @property
    def nodes(self) -> _NodesAccessor:
        return _NodesAccessor(self)

    @property
    def ndata(self) -> _NDataAccessor:
        return _NDataAccessor(self)

    @property
    def edata(self) -> _EDataAccessor:
        return _EDataAccessor(self)

    def num_nodes(self, ntype: Optional[str] = None) -> int:
        if ntype is None:
            return sum(self._num_nodes.values())
        return self._num_nodes[ntype]

    number_of_nodes = num_nodes

    def num_edges(self, etype: Optional[CanonicalEType] = None) -> int:
        if etype is None:
            if self._edges_store is None:
                return int(self._plan.num_edges)
            return sum(int(u.numel()) for u, _ in self._edges.values())
        return int(self._edges[etype][0].numel())

    number_of_edges = num_edges

    def edges(self, etype: Optional[CanonicalEType] = None):
        if etype is None:
            if len(self._rels) != 1:
                raise ValueError("etype is required for a multi-relation graph")
            etype = self.canonical_etypes[0]
        return self._edges[etype]

    @property
    def batch_size(self) -> int:
        if self._batch_num_nodes is None:
            return 1
        return int(next(iter(self._batch_num_nodes.values())).numel())

    def batch_num_nodes(self, ntype: Optional[str] = None) -> torch.Tensor:
        if ntype is None:
            if len(self._num_nodes) != 1:
                raise ValueError("ntype is required for a multi-type graph")
            ntype = self.ntypes[0]
        if self._batch_num_nodes is None:
            return torch.tensor([self._num_nodes[ntype]], dtype=torch.int64)
        return self._batch_num_nodes[ntype]

    @property
    def device(self) -> torch.device:
        for fr in self._nframes.values():
            for v in fr.values():
                return v.device
        if self._edges_store is None:
            return self._plan.device
        for u, _ in self._edges_store.values():
            return u.device
        return torch.device("cpu")

    @contextlib.contextmanager
    def local_scope(self):
        """Frames written inside the scope are dropped on exit (DGL ``local_scope``)."""
        nsnap = {t: dict(fr) for t, fr in self._nframes.items()}
        esnap = {r: dict(fr) for r, fr in self._eframes.items()}
        try:
            yield self
        finally:
            for t, fr in self._nframes.items():
                fr.clear()
                fr.update(nsnap[t])
            for r, fr in self._eframes.items():
                fr.clear()
                fr.update(esnap[r])

    def to(self, device) -> "HeteroGraph":
        """``g.to(device)`` (trainer/train_gnn.py:60,64).  Returns ``self`` — with its cached kernel plan, contexts and
        concatenated tables — whenever no tensor would move ('cuda' and 'cuda:<current>' are the same device)."""
        device = _resolve_device(device)
        if self._all_on(device):
            return self
        edges = self._edges                      # a loader batch materialises its COO before it moves
        g = HeteroGraph(self._num_nodes, OrderedDict((r, (u.to(device), v.to(device))) for r, (u, v) in edges.items()),
                        self._batch_num_nodes)
        for t, fr in self._nframes.items():
            for k, v in fr.items():
                g._nframes[t][k] = v.to(device)
        for r, fr in self._eframes.items():
            for k, v in fr.items():
                g._eframes[r][k] = v.to(device)
        return g

    def _all_on(self, device) -> bool:
        for fr in list(self._nframes.values()) + list(self._eframes.values()):
            for v in fr.values():
                if _resolve_device(v.device) != device:
                    return False
        if self._edges_store is None:
            return _resolve_device(self._plan.device) == device
        for u, v in self._edges_store.values():
            if _resolve_device(u.device) != device or _resolve_device(v.device) != device:
                return False
        return True

    # ------------------------------------------------------------------ type-major helpers
    def type_offsets(self) -> List[int]:
        off = [0]
        for t in self.ntypes:
            off.append(off[-1] + self._num_nodes[t])
        return off

    def cat_ndata(self, key: str = "feat") -> torch.Tensor:
        """Concatenate a node field type-major into one ``[N, F]`` fp32 tensor (the kernels' layout).

        Cached per graph (keyed by the parts' storage), so a resident batch is concatenated once."""
        parts = [self._nframes[t][key] for t in self.ntypes]
        sig = tuple((p.data_ptr(), tuple(p.shape), p.dtype, p._version) for p in parts)
        cache = self.__dict__.setdefault("_cat_cache", {})
        hit = cache.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        out = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        out = out.to(torch.float32).contiguous()
        cache[key] = (sig, out)
        return out

    def cat_edata_csr(self, key: str = "sim") -> torch.Tensor:
        """Edge field of all relations, fp32, permuted into the plan's CSR edge order (cached; the cache follows
        re-assignment and in-place edits of the per-relation tensors)."""
        cache = self.__dict__.setdefault("_cat_cache", {})
        if self._edges_store is None and ("e", key) in cache:
            return cache[("e", key)][1]             # loader batch, per-relation fields never touched: stored at assembly time
        parts = [self._eframes[r][key] for r in self.canonical_etypes]
        sig = tuple((p.data_ptr(), tuple(p.shape), p.dtype, p._version) for p in parts)
        hit = cache.get(("e", key))
        if hit is not None and hit[0] == sig:
            return hit[1]
        plan = self.plan()
        if parts:
            flat = torch.cat([p.reshape(-1) for p in parts]).to(device=plan.device, dtype=torch.float32)
            out = flat[self._csr_perm()].contiguous() if flat.numel() else flat
        else:
            out = torch.empty(0, dtype=torch.float32, device=plan.device)
        cache[("e", key)] = (sig, out)
        return out

    def _csr_perm(self) -> torch.Tensor:
        """CSR position -> position in the relation-major concatenated edge list.  Loader-assembled plans do not carry
        it; their edge order is the same stable (segment, input order) sort, so it is recomputed from the COO."""
        plan = self.plan()
        if plan.perm is None:
            hd = PlanHeader(self.ntypes, self.canonical_etypes, [self.num_nodes(t) for t in self.ntypes])
            gseg = [hd.seg_off[hd.tindex[d]] + v.to(plan.device) * hd.R[hd.tindex[d]] + hd.slot_of_rel[ri]
                    for ri, ((s, e, d), (u, v)) in enumerate(self._edges.items())]
            gseg = torch.cat(gseg) if gseg else torch.empty(0, dtype=torch.int64, device=plan.device)
            plan.perm = torch.sort(gseg, stable=True).indices
        return plan.perm

    # ------------------------------------------------------------------ kernel plan
    def plan(self, per_relation_src: bool = False) -> GraphPlan:
        """Kernel plan.  ``per_relation_src``: source rows are numbered per (relation, source node) — the layout
        HGT needs, where every relation has its own transformed K/V table (models/HGT.py:92-97)."""
        if per_relation_src:
            cache = self.__dict__.setdefault("_plan_rel", None)
            if cache is None:
                self.__dict__["_plan_rel"] = cache = _build_plan(self, per_relation_src=True)
            return cache
        if self._plan is None:
            self._plan = _build_plan(self)
        return self._plan


class _PinnedArena:
    """Ring of page-locked host memory for small host->device transfers.

    On this ROCm stack a pageable ``torch.tensor(list, device='cuda')`` copy blocks the host until the GPU has drained
    (measured: the call took as long as the training step still queued), and ``tensor.pin_memory()`` per call costs a
    ``hipHostMalloc`` (also synchronising).  Both serialised batch preparation with the running step.  Carving the
    staging space out of one long-lived pinned buffer makes every small upload a plain asynchronous copy."""

    def __init__(self, nbytes: int = 32 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.size = nbytes
        self.off = 0
        self.lap = 0
        self.pending = collections.deque()     # (lap, start, end, event) of the copies issued, in issue (= address, lap-major) order

    def stage(self, t: torch.Tensor, device: torch.device) -> torch.Tensor:
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return torch.empty(t.shape, dtype=t.dtype, device=device)
        need = (nbytes + 255) // 256 * 256
        if need > self.size:
            return t.to(device)                       # oversized: plain (blocking) copy
        if self.off + need > self.size:
            self.off = 0
            self.lap += 1
        start, end = self.off, self.off + need
        # a ring: only the copies of the PREVIOUS lap that still read from [start, end) must have left the buffer - the oldest
        # entries, long complete in steady state.  (Waiting for every pending copy at the wrap, as an earlier version did, also waits
        # for the ones just queued on the compute stream behind a whole training step: the host then idles until the GPU drains.)
        while self.pending:
            lap, a, b, evt = self.pending[0]
            if lap == self.lap or (lap == self.lap - 1 and a >= end):
                break
            evt.synchronize()
            self.pending.popleft()
        view = self.buf[start:start + nbytes].view(t.dtype).view(t.shape)
        view.copy_(t)
        out = view.to(device, non_blocking=True)
        evt = torch.cuda.Event()
        evt.record(torch.cuda.current_stream(device))
        self.off = end
        self.pending.append((self.lap, start, end, evt))
        return out


_ARENAS: dict = {}


def host_to_device(values, dtype, device) -> torch.Tensor:
    """Small host list/tensor -> device tensor as an ASYNCHRONOUS copy on the current stream (see _PinnedArena)."""
    t = values.to(dtype).contiguous() if isinstance(values, torch.Tensor) else torch.tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != "cuda":
        return t
    key = device.index if device.index is not None else torch.cuda.current_device()
    arena = _ARENAS.get(key)
    if arena is None:
        arena = _ARENAS[key] = _PinnedArena()
    return arena.stage(t, device)


def _count(idx: torch.Tensor, size: int) -> torch.Tensor:
    """``torch.bincount(idx, minlength=size)`` without its device->host sync (bincount reads max(idx) on the host)."""
    out = torch.zeros(size, dtype=torch.int64, device=idx.device)
    if idx.numel():
        out.index_add_(0, idx, torch.ones_like(idx))
    return out


class PlanHeader:
    """Host-side part of a plan: schema, per-type offsets, relation slots (no device work)."""

    def __init__(self, ntypes: List[str], rels: List[CanonicalEType], counts: List[int]):
        self.ntypes, self.rels, self.counts = ntypes, rels, counts
        self.tindex = {t: i for i, t in enumerate(ntypes)}
        self.type_off = [0]
        for c in counts:
            self.type_off.append(self.type_off[-1] + c)
        self.N = self.type_off[-1]
        slots: List[List[int]] = [[] for _ in ntypes]
        self.slot_of_rel: List[int] = []
        for ri, (s, e, d) in enumerate(rels):
            self.slot_of_rel.append(len(slots[self.tindex[d]]))
            slots[self.tindex[d]].append(ri)
        self.R = [len(x) for x in slots]                    # relations whose dst type is t (empty relations included)
        self.seg_off = [0]
        for ti in range(len(ntypes)):
            self.seg_off.append(self.seg_off[-1] + counts[ti] * self.R[ti])
        self.S = self.seg_off[-1]
        self.rel_rows: List[Tuple[int, int]] = []           # per-relation source row ranges (per_relation_src layout)
        off = 0
        for (s, e, d) in rels:
            ns = counts[self.tindex[s]]
            self.rel_rows.append((off, off + ns))
            off += ns
        self.rel_rows_total = off


# In-degree above which a node goes to the cooperative hub kernels (passed to the kernels with every call: ops._attn_flags).
# Attention ms per step by threshold 32 / 64 / 96 / 128 / 192: synthetic hub batch 2.50 / 2.44 / 2.44 / 2.44 / 2.49; kNN study graphs
# in construction order 2.58 / 2.39 / - / 2.35 / -: one workgroup per node only pays for long chains.
HEAVY_DEGREE = 128
# Locality-ordered (kNN) graphs: every node that is pulled out of the position order into the hub prefix costs locality, and a
# single wave walks a few dozen neighbouring rows out of L2 quickly; measured on the WSI-like study graphs (attention ms per step):
# threshold 32 -> 2.15, 64 -> 2.56, 128 -> 3.04 with the top-N/32 candidates in the prefix; no hub split at all -> 1.93.  So only
# the nodes ABOVE a high threshold go to the hub kernels and nothing else leaves the position order.
HEAVY_DEGREE_LOCALITY = 128
# Two more plan switches (module attributes like the thresholds above; `set_plan_options` sets any of them - the package reads no environment
# variable): HUB_SPLIT = False never routes a node to the hub kernels; LOCALITY = False ignores the slides' '_pos' (heaviest-first orders).
HUB_SPLIT = True
LOCALITY = True


def set_plan_options(heavy_degree: Optional[int] = None, heavy_degree_locality: Optional[int] = None, hub_split: Optional[bool] = None,
                     locality: Optional[bool] = None) -> None:
    """Options of every kernel plan built FROM NOW ON (cached plans keep what they were built with)."""
    global HEAVY_DEGREE, HEAVY_DEGREE_LOCALITY, HUB_SPLIT, LOCALITY
    if heavy_degree is not None:
        HEAVY_DEGREE = int(heavy_degree)
    if heavy_degree_locality is not None:
        HEAVY_DEGREE_LOCALITY = int(heavy_degree_locality)
    if hub_split is not None:
        HUB_SPLIT = bool(hub_split)
    if locality is not None:
        LOCALITY = bool(locality)


def finish_plan(hd: PlanHeader, gsrc, gdst, gseg, dev, per_relation_src: bool,
                batch_counts: List[List[int]], max_in_degree: Optional[int] = None, pos: Optional[torch.Tensor] = None) -> GraphPlan:
    """Device part of the plan from the concatenated global edge arrays (int64, any order): CSR by (dst, relation slot),
    CSC by source row, degree orders, readout pointers.  No device->host synchronisation when the caller knows
    ``max_in_degree`` (the loader does, per stored graph); otherwise it is read back once (one sync per plan)."""
    p = GraphPlan()
    p.device = dev
    N, S = hd.N, hd.S
    p.type_off, p.num_nodes, p.rel_slots, p.num_segs, p.rel_rows = hd.type_off, N, hd.R, S, list(hd.rel_rows)
    node_seg = torch.empty(N + 1, dtype=torch.int64, device=dev)
    inv_rd = torch.empty(N, dtype=torch.float32, device=dev)
    for ti in range(len(hd.ntypes)):
        n = hd.counts[ti]
        a, b = hd.type_off[ti], hd.type_off[ti + 1]
        node_seg[a:b] = hd.seg_off[ti] + torch.arange(n, device=dev, dtype=torch.int64) * hd.R[ti]
        inv_rd[a:b] = (1.0 / hd.R[ti]) if hd.R[ti] > 0 else 0.0
    node_seg[N:].fill_(S)        # (not `node_seg[N] = S`: a scalar __setitem__ is a synchronising pageable copy)
    E = int(gsrc.numel())
    p.num_edges = E
    if E >= 2 ** 31 - 1 or S >= 2 ** 31 - 1:
        raise ValueError("graph too large for the int32 kernel plan")
    perm = torch.sort(gseg, stable=True).indices if E else gseg
    src_c = gsrc[perm]
    dst_c = gdst[perm]
    seg_c = gseg[perm]
    rowptr = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    if E:
        rowptr[1:] = torch.cumsum(_count(seg_c, S), 0)
    p.perm = perm
    p.src = src_c.to(torch.int32).contiguous()
    p.rowptr = rowptr.to(torch.int32).contiguous()
    p.node_seg = node_seg.to(torch.int32).contiguous()
    p.inv_rd = inv_rd.contiguous()
    NS = hd.rel_rows_total if per_relation_src else N
    p.num_src_rows = NS
    cperm = torch.sort(src_c, stable=True).indices if E else src_c
    colptr = torch.zeros(NS + 1, dtype=torch.int64, device=dev)
    if E:
        colptr[1:] = torch.cumsum(_count(src_c, NS), 0)
    p.colptr = colptr.to(torch.int32).contiguous()
    p.csc_eid = cperm.to(torch.int32).contiguous()
    p.csc_dst = dst_c[cperm].to(torch.int32).contiguous() if E else dst_c.to(torch.int32)
    indeg = _count(dst_c, N)
    outdeg = colptr[1:] - colptr[:-1]
    B = len(batch_counts[0]) if batch_counts else 1
    p.batch_size = B
    p.graph_sizes = [sum(int(batch_counts[ti][b]) for ti in range(len(hd.ntypes))) for b in range(B)] if batch_counts else [N]
    # Processing orders.  dst side: the M highest in-degree nodes first (candidates for the hub kernel, wsi_heat_attn_fwd's
    # num_heavy), then graph-major and heaviest-first inside a graph: all CUs work on ONE graph's K/V rows at a time (41 MB
    # at 10k nodes, D=512), which the 256 MB Infinity Cache holds, instead of sweeping the whole batch's tables.
    M = 0 if not HUB_SPLIT else min(N, max(64, N // 32))
    if M > 0:
        if max_in_degree is None:
            max_in_degree = int(indeg.max().item()) if E else 0
        if max_in_degree <= HEAVY_DEGREE:      # no hubs in this batch: skip the second launch and its fork/join
            M = 0
    p.num_heavy = M
    if pos is not None and NS == N:
        # Locality order (graph.apply_locality_order): inside a graph, nodes are visited in the order of their positions in the
        # slide's bandwidth-reducing order, across node types, so that the waves in flight at any time gather K/V rows of
        # neighbouring patches (kNN graphs: a small, L2-sized set) — instead of heaviest-first, which scatters them.
        flat = [int(batch_counts[ti][b]) for ti in range(len(hd.ntypes)) for b in range(B)]
        gid = torch.arange(B, device=dev).repeat(len(hd.ntypes)).repeat_interleave(host_to_device(flat, torch.int64, dev), output_size=N)
        key = gid * (int(N) + 1) + pos.to(device=dev, dtype=torch.int64)
        kd = key.clone()
        p.heavy_degree = HEAVY_DEGREE_LOCALITY
        M = int((indeg > HEAVY_DEGREE_LOCALITY).sum().item()) if (M > 0 and max_in_degree > HEAVY_DEGREE_LOCALITY) else 0
        p.num_heavy = M
        if M > 0:                  # exactly the nodes above the threshold, heaviest first; everything else stays in position order
            top = torch.topk(indeg, M, sorted=True).indices
            kd[top] = torch.arange(M, device=dev, dtype=kd.dtype) - M
        p.order_dst = torch.sort(kd, stable=True).indices.to(torch.int32).contiguous()
        p.order_src = torch.sort(key, stable=True).indices.to(torch.int32).contiguous()
        p.locality = True
    elif B == 1 and M == 0:
        p.order_dst = torch.sort(indeg, descending=True, stable=True).indices.to(torch.int32).contiguous()
        p.order_src = torch.sort(outdeg, descending=True, stable=True).indices.to(torch.int32).contiguous()
    else:
        flat = [int(batch_counts[ti][b]) for ti in range(len(hd.ntypes)) for b in range(B)] if batch_counts else [N]
        reps = len(hd.ntypes) if batch_counts else 1
        gid = torch.arange(B, device=dev).repeat(reps).repeat_interleave(host_to_device(flat, torch.int64, dev), output_size=N)
        big = E + 1
        kd = gid * big + (E - indeg)
        if M > 0:
            top = torch.topk(indeg, M, sorted=True).indices
            kd[top] = torch.arange(M, device=dev, dtype=kd.dtype) - M           # negative keys: ahead of everything, by rank
        p.order_dst = torch.sort(kd, stable=True).indices.to(torch.int32).contiguous()
        p.order_src = torch.sort(gid[:NS] * big + (E - outdeg), stable=True).indices.to(torch.int32).contiguous() \
            if NS == N else torch.sort(outdeg, descending=True, stable=True).indices.to(torch.int32).contiguous()
    ptr = [0]
    for ti in range(len(hd.ntypes)):
        base, acc = hd.type_off[ti], 0
        for b in range(B):
            acc += int(batch_counts[ti][b])
            ptr.append(base + acc)
        if acc != hd.counts[ti]:
            raise ValueError(f"batch_num_nodes of type {hd.ntypes[ti]} does not sum to its node count")
    p.readout_ptr = host_to_device(ptr, torch.int32, dev)
    return p


def attn_tiles(plan: GraphPlan, parts: int = 8):
    """``wsi_attn_tiles_t`` of a plan, or None when the L2-blocked attention kernels do not apply to it: the processing orders are cut into
    8 parts of equal node count (part p runs on XCD p) and every part into SPANS that lie inside one graph - both orders are graph-major, so a
    span's gathers touch one graph's table rows only.  Needs orders without a hub prefix (``num_heavy == 0``: the hub kernels are not blocked)
    and HEAT-style source rows.  Host-only arithmetic on the batch's graph sizes; cached on the plan."""
    hit = plan.__dict__.get("_attn_tiles", False)
    if hit is not False:
        return hit
    from . import _native as N
    tiles = None
    sizes = [int(x) for x in (plan.graph_sizes or []) if int(x) > 0]
    n = int(plan.num_nodes)
    if sizes and sum(sizes) == n and plan.num_heavy == 0 and plan.num_src_rows == n and len(sizes) + parts <= N.WSI_ATTN_MAX_SPANS:
        cuts = [0]
        for x in sizes:
            cuts.append(cuts[-1] + x)
        tiles = N.AttnTiles()
        k = 0
        tiles.part_ptr[0] = 0
        for p_ in range(parts):
            a, b = (p_ * n) // parts, ((p_ + 1) * n) // parts
            for gi in range(len(sizes)):
                lo, hi = max(a, cuts[gi]), min(b, cuts[gi + 1])
                if hi > lo:
                    tiles.begin[k], tiles.end[k] = lo, hi
                    k += 1
            tiles.part_ptr[p_ + 1] = k
    plan.__dict__["_attn_tiles"] = tiles
    return tiles


class PlanPieces:
    """Per-graph, per-node-type pieces of a single graph's kernel plan, kept on the device so that the plan of ANY batch
    containing the graph is a concatenation plus offset additions — no sort, no host synchronisation (data.py).

    The batch layout is type-major (all graphs' type-0 nodes, then type-1, ...), a single graph's plan is type-major too,
    so the edges whose destination has type t are one contiguous CSR range of the graph and the CSC entries whose source
    has type s likewise; inside a piece only local ids are stored, with the node type of the other endpoint per entry."""

    def __init__(self, hd: PlanHeader, plan: GraphPlan, sim_csr: torch.Tensor, pos: Optional[torch.Tensor] = None):
        dev = plan.device
        T = len(hd.ntypes)
        toff = torch.tensor(hd.type_off, dtype=torch.int64, device=dev)
        rowptr, colptr = plan.rowptr.long(), plan.colptr.long()
        e_start = rowptr[torch.tensor(hd.seg_off, dtype=torch.int64, device=dev)].tolist()          # one sync, once per graph
        c_start = colptr[toff].tolist()
        self.counts = list(hd.counts)
        self.ecount = [e_start[t + 1] - e_start[t] for t in range(T)]
        self.ccount = [c_start[t + 1] - c_start[t] for t in range(T)]
        src = plan.src.long()
        src_t = torch.bucketize(src, toff[1:], right=True)
        src_l = src - toff[src_t]
        eid = plan.csc_eid.long()
        es = torch.tensor(e_start, dtype=torch.int64, device=dev)
        dt = torch.bucketize(eid, es[1:], right=True)                    # destination type of each CSC entry's edge
        eid_l = eid - es[dt]
        dst_l = plan.csc_dst.long() - toff[dt]
        ns = plan.node_seg.long()
        indeg = rowptr[ns[1:]] - rowptr[ns[:-1]]
        outdeg = colptr[1:] - colptr[:-1]
        self.rp, self.src_l, self.src_t, self.sim = [], [], [], []
        self.cp, self.eid_l, self.ent_t, self.dst_l = [], [], [], []
        heavy_l, heavy_t, light_l, light_t, so_l, so_t, nheavy = [], [], [], [], [], [], []
        for t in range(T):
            a, b = hd.seg_off[t], hd.seg_off[t + 1]
            self.rp.append(rowptr[a:b] - e_start[t])
            sl = slice(e_start[t], e_start[t + 1])
            self.src_l.append(src_l[sl]); self.src_t.append(src_t[sl]); self.sim.append(sim_csr[sl])
            na, nb = hd.type_off[t], hd.type_off[t + 1]
            self.cp.append(colptr[na:nb] - c_start[t])
            cl = slice(c_start[t], c_start[t + 1])
            self.eid_l.append(eid_l[cl]); self.ent_t.append(dt[cl]); self.dst_l.append(dst_l[cl])
            deg = indeg[na:nb]
            od = torch.sort(deg, descending=True, stable=True).indices
            h = int((deg > (HEAVY_DEGREE_LOCALITY if pos is not None else HEAVY_DEGREE)).sum().item())
            nheavy.append(h)
            tt = torch.full((nb - na,), t, dtype=torch.int64, device=dev)
            heavy_l.append(od[:h]); heavy_t.append(tt[:h]); light_l.append(od[h:]); light_t.append(tt[h:])
            so_l.append(torch.sort(outdeg[na:nb], descending=True, stable=True).indices); so_t.append(tt)
        cat = lambda xs: torch.cat(xs) if xs else torch.empty(0, dtype=torch.int64, device=dev)
        self.heavy_l, self.heavy_t, self.light_l, self.light_t = cat(heavy_l), cat(heavy_t), cat(light_l), cat(light_t)
        self.so_l, self.so_t = cat(so_l), cat(so_t)
        self.locality = pos is not None
        if pos is not None:      # locality order (apply_locality_order): light destinations and all sources by slide-wide position
            pos = pos.to(device=dev, dtype=torch.int64)
            gl = self.light_l + toff[self.light_t]
            o = torch.sort(pos[gl], stable=True).indices
            self.light_l, self.light_t = self.light_l[o], self.light_t[o]
            gs_ = self.so_l + toff[self.so_t]
            o = torch.sort(pos[gs_], stable=True).indices
            self.so_l, self.so_t = self.so_l[o], self.so_t[o]
        self.num_heavy = sum(nheavy)
        self.max_in_degree = int(indeg.max().item()) if indeg.numel() else 0


def plan_frame(hd: PlanHeader, dev, batch_counts: List[List[int]]) -> GraphPlan:
    """The parts of a plan that depend on the node counts only: node_seg, inv_rd, readout pointers."""
    p = GraphPlan()
    p.device = dev
    N, S = hd.N, hd.S
    p.type_off, p.num_nodes, p.rel_slots, p.num_segs, p.rel_rows = hd.type_off, N, hd.R, S, list(hd.rel_rows)
    node_seg = torch.empty(N + 1, dtype=torch.int64, device=dev)
    inv_rd = torch.empty(N, dtype=torch.float32, device=dev)
    for ti in range(len(hd.ntypes)):
        n = hd.counts[ti]
        a, b = hd.type_off[ti], hd.type_off[ti + 1]
        node_seg[a:b] = hd.seg_off[ti] + torch.arange(n, device=dev, dtype=torch.int64) * hd.R[ti]
        inv_rd[a:b] = (1.0 / hd.R[ti]) if hd.R[ti] > 0 else 0.0
    node_seg[N:].fill_(S)        # (not `node_seg[N] = S`: a scalar __setitem__ is a synchronising pageable copy)
    p.node_seg = node_seg.to(torch.int32).contiguous()
    p.inv_rd = inv_rd.contiguous()
    B = len(batch_counts[0]) if batch_counts else 1
    p.batch_size = B
    p.graph_sizes = [sum(int(batch_counts[ti][b]) for ti in range(len(hd.ntypes))) for b in range(B)] if batch_counts else [hd.N]
    ptr = [0]
    for ti in range(len(hd.ntypes)):
        base, acc = hd.type_off[ti], 0
        for b in range(B):
            acc += int(batch_counts[ti][b])
            ptr.append(base + acc)
        if acc != hd.counts[ti]:
            raise ValueError(f"batch_num_nodes of type {hd.ntypes[ti]} does not sum to its node count")
    p.readout_ptr = host_to_device(ptr, torch.int32, dev)
    return p


def _plan_frame_host(hd: PlanHeader, dev, batch_counts: List[List[int]]) -> GraphPlan:
    """plan_frame without its device tables (node_seg, inv_rd: written by the assembly kernel)."""
    p = GraphPlan()
    p.device = dev
    p.type_off, p.num_nodes, p.rel_slots, p.num_segs, p.rel_rows = hd.type_off, hd.N, hd.R, hd.S, list(hd.rel_rows)
    B = len(batch_counts[0]) if batch_counts else 1
    p.batch_size = B
    p.graph_sizes = [sum(int(batch_counts[ti][b]) for ti in range(len(hd.ntypes))) for b in range(B)] if batch_counts else [hd.N]
    ptr = [0]
    for ti in range(len(hd.ntypes)):
        base, acc = hd.type_off[ti], 0
        for b in range(B):
            acc += int(batch_counts[ti][b])
            ptr.append(base + acc)
        if acc != hd.counts[ti]:
            raise ValueError(f"batch_num_nodes of type {hd.ntypes[ti]} does not sum to its node count")
    p.readout_ptr = host_to_device(ptr, torch.int32, dev)
    return p


def assemble_plan(hd: PlanHeader, pieces: Sequence[PlanPieces], dev, batch_counts: List[List[int]]):
    """Plan of the block-diagonal batch of the graphs whose pieces are given (+ the CSR-ordered ``sim``).  Every table is a concatenation of
    the pieces with per-piece offsets (node / edge ids through two small lookup tables): on the GPU ONE launch over a table of segment
    descriptors (``wsi_plan_assemble``; one upload), no sort, no device->host synchronisation.  ``assemble_plan_torch`` is the same thing as
    ~90 tensor operations (CPU graphs; the kernel's test compares the two bit for bit)."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        return assemble_plan_torch(hd, pieces, dev, batch_counts)
    import struct
    from . import _native as N
    T, B = len(hd.ntypes), len(pieces)
    p = _plan_frame_host(hd, dev, batch_counts)
    Nn, S = hd.N, hd.S
    pre = [[0] * T for _ in range(B + 1)]
    for b in range(B):
        for t in range(T):
            pre[b + 1][t] = pre[b][t] + batch_counts[t][b]
    node_tab = [hd.type_off[t] + pre[b][t] for b in range(B) for t in range(T)]                       # [b*T + t]
    eoff, coff, acc_e, acc_c = {}, {}, 0, 0
    for t in range(T):
        for b in range(B):
            eoff[(t, b)], coff[(t, b)] = acc_e, acc_c
            acc_e += pieces[b].ecount[t]
            acc_c += pieces[b].ccount[t]
    E = acc_e
    if E >= 2 ** 31 - 1 or S >= 2 ** 31 - 1:
        raise ValueError("graph too large for the int32 kernel plan")
    p.num_edges, p.num_src_rows = E, Nn
    edge_tab = [eoff[(t, b)] for b in range(B) for t in range(T)]
    H = sum(pc.num_heavy for pc in pieces)
    i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)[:int(n)]
    rowptr, colptr, node_seg = i32(S + 1), i32(Nn + 1), i32(Nn + 1)
    src, csc_eid, csc_dst = i32(E), i32(E), i32(E)
    order_dst, order_src = i32(Nn), i32(Nn)
    sim = torch.empty(max(E, 1), dtype=torch.float32, device=dev)[:E]
    inv_rd = torch.empty(max(Nn, 1), dtype=torch.float32, device=dev)[:Nn]
    segs: List[List[int]] = []

    def seg(out, off, esize, n, in1=None, in2=None, tab=-1, key=0, add=0, stride=0, mode=0):
        if n > 0:
            segs.append([out.data_ptr() + off * esize, 0 if in1 is None else in1.data_ptr(), 0 if in2 is None else in2.data_ptr(),
                         tab, key, add, stride, int(n), mode, 0])

    NODE, EDGE = 0, B * T              # offsets of the two lookup tables behind the descriptors (filled in below)
    so = co = 0
    for t in range(T):
        for b in range(B):
            pc = pieces[b]
            ns, nc = pc.counts[t] * hd.R[t], pc.counts[t]
            seg(rowptr, so, 4, ns, in1=pc.rp[t], add=eoff[(t, b)])
            seg(colptr, co, 4, nc, in1=pc.cp[t], add=coff[(t, b)])
            so += ns
            co += nc
            ne, ncc = pc.ecount[t], pc.ccount[t]
            seg(src, eoff[(t, b)], 4, ne, in1=pc.src_l[t], in2=pc.src_t[t], tab=NODE, key=b * T)
            seg(sim, eoff[(t, b)], 4, ne, in1=pc.sim[t], mode=1)
            seg(csc_eid, coff[(t, b)], 4, ncc, in1=pc.eid_l[t], in2=pc.ent_t[t], tab=EDGE, key=b * T)
            seg(csc_dst, coff[(t, b)], 4, ncc, in1=pc.dst_l[t], in2=pc.ent_t[t], tab=NODE, key=b * T)
    seg(rowptr, S, 4, 1, add=E)
    seg(colptr, Nn, 4, 1, add=E)
    ho, lo, oo = 0, H, 0
    for b, pc in enumerate(pieces):                 # processing orders: exact hub list first, then graph-major / heaviest-first inside (graph, type)
        nh, nl, no = int(pc.heavy_l.numel()), int(pc.light_l.numel()), int(pc.so_l.numel())
        seg(order_dst, ho, 4, nh, in1=pc.heavy_l, in2=pc.heavy_t, tab=NODE, key=b * T)
        seg(order_dst, lo, 4, nl, in1=pc.light_l, in2=pc.light_t, tab=NODE, key=b * T)
        seg(order_src, oo, 4, no, in1=pc.so_l, in2=pc.so_t, tab=NODE, key=b * T)
        ho, lo, oo = ho + nh, lo + nl, oo + no
    for t in range(T):
        seg(node_seg, hd.type_off[t], 4, hd.counts[t], add=hd.seg_off[t], stride=hd.R[t])
        seg(inv_rd, hd.type_off[t], 4, hd.counts[t], add=struct.unpack("<i", struct.pack("<f", (1.0 / hd.R[t]) if hd.R[t] > 0 else 0.0))[0], mode=2)
    seg(node_seg, Nn, 4, 1, add=S)
    blocks = 0
    tab0 = len(segs) * 10
    for s_ in segs:
        if s_[3] >= 0:
            s_[3] += tab0
        s_[9] = blocks
        blocks += (s_[7] + 1023) // 1024
    desc = host_to_device([w for s_ in segs for w in s_] + node_tab + edge_tab, torch.int64, dev)
    N.check(N.load().wsi_plan_assemble(N.ptr(desc), len(segs), blocks, N.stream()), "wsi_plan_assemble")
    p.node_seg, p.inv_rd = node_seg, inv_rd
    p.rowptr, p.colptr, p.src, p.csc_eid, p.csc_dst = rowptr, colptr, src, csc_eid, csc_dst
    p.order_dst, p.order_src = order_dst, order_src
    p._assembly_desc = desc             # (the pieces are owned by the stored graphs; the descriptor table must outlive the launch: held by the plan)
    p.num_heavy = H if HUB_SPLIT else 0
    p.locality = all(pc.locality for pc in pieces)
    if any(pc.locality for pc in pieces) and not p.locality:
        raise ValueError("a batch mixes locality-ordered and plain graphs: apply graph.apply_locality_order to all of a data set's slides or none")
    p.heavy_degree = HEAVY_DEGREE_LOCALITY if p.locality else HEAVY_DEGREE
    return p, sim


def assemble_plan_torch(hd: PlanHeader, pieces: Sequence[PlanPieces], dev, batch_counts: List[List[int]]):
    """``assemble_plan`` as tensor operations: concatenations, two small offset tables and a few gathers (~90 launches on a GPU)."""
    T, B = len(hd.ntypes), len(pieces)
    p = plan_frame(hd, dev, batch_counts)
    N, S = hd.N, hd.S
    pre = [[0] * T for _ in range(B + 1)]
    for b in range(B):
        for t in range(T):
            pre[b + 1][t] = pre[b][t] + batch_counts[t][b]
    node_tab = [hd.type_off[t] + pre[b][t] for b in range(B) for t in range(T)]                       # [b*T + t]
    eoff, coff, acc_e, acc_c = {}, {}, 0, 0
    for t in range(T):
        for b in range(B):
            eoff[(t, b)], coff[(t, b)] = acc_e, acc_c
            acc_e += pieces[b].ecount[t]
            acc_c += pieces[b].ccount[t]
    E = acc_e
    if E >= 2 ** 31 - 1 or S >= 2 ** 31 - 1:
        raise ValueError("graph too large for the int32 kernel plan")
    p.num_edges, p.num_src_rows = E, N
    order = [(t, b) for t in range(T) for b in range(B)]
    tabs = host_to_device([node_tab, [eoff[(t, b)] for b in range(B) for t in range(T)]], torch.int64, dev)     # [2, B*T]
    meta = host_to_device([[eoff[k] for k in order], [pieces[b].counts[t] * hd.R[t] for (t, b) in order],
                           [coff[k] for k in order], [pieces[b].counts[t] for (t, b) in order],
                           [b * T for (t, b) in order], [pieces[b].ecount[t] for (t, b) in order],
                           [pieces[b].ccount[t] for (t, b) in order]], torch.int64, dev)                        # [7, T*B]
    ri = torch.repeat_interleave
    rowptr = torch.empty(S + 1, dtype=torch.int64, device=dev)
    torch.add(torch.cat([pieces[b].rp[t] for (t, b) in order]), ri(meta[0], meta[1], output_size=S), out=rowptr[:S])
    rowptr[S:].fill_(E)
    colptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
    torch.add(torch.cat([pieces[b].cp[t] for (t, b) in order]), ri(meta[2], meta[3], output_size=N), out=colptr[:N])
    colptr[N:].fill_(E)
    ekey = ri(meta[4], meta[5], output_size=E)                        # b*T of every CSR edge
    ckey = ri(meta[4], meta[6], output_size=E)                        # b*T of every CSC entry
    src = torch.cat([pieces[b].src_l[t] for (t, b) in order]) + tabs[0][ekey + torch.cat([pieces[b].src_t[t] for (t, b) in order])]
    ent = ckey + torch.cat([pieces[b].ent_t[t] for (t, b) in order])
    csc_eid = torch.cat([pieces[b].eid_l[t] for (t, b) in order]) + tabs[1][ent]
    csc_dst = torch.cat([pieces[b].dst_l[t] for (t, b) in order]) + tabs[0][ent]
    sim = torch.cat([pieces[b].sim[t] for (t, b) in order])
    # processing orders: exact hub list first, then graph-major / heaviest-first inside (graph, type)
    bkeys = host_to_device([[b * T for b in range(B)], [int(pc.heavy_l.numel()) for pc in pieces],
                            [int(pc.light_l.numel()) for pc in pieces], [int(pc.so_l.numel()) for pc in pieces]], torch.int64, dev)
    H = sum(pc.num_heavy for pc in pieces)
    heavy = torch.cat([pc.heavy_l for pc in pieces]) + tabs[0][ri(bkeys[0], bkeys[1], output_size=H) + torch.cat([pc.heavy_t for pc in pieces])]
    light = torch.cat([pc.light_l for pc in pieces]) + tabs[0][ri(bkeys[0], bkeys[2], output_size=N - H) + torch.cat([pc.light_t for pc in pieces])]
    osrc = torch.cat([pc.so_l for pc in pieces]) + tabs[0][ri(bkeys[0], bkeys[3], output_size=N) + torch.cat([pc.so_t for pc in pieces])]
    p.rowptr, p.colptr = rowptr.to(torch.int32), colptr.to(torch.int32)
    p.src, p.csc_eid, p.csc_dst = src.to(torch.int32), csc_eid.to(torch.int32), csc_dst.to(torch.int32)
    p.order_dst = torch.cat([heavy, light]).to(torch.int32)
    p.order_src = osrc.to(torch.int32)
    p.num_heavy = H if HUB_SPLIT else 0
    p.locality = all(pc.locality for pc in pieces)
    if any(pc.locality for pc in pieces) and not p.locality:
        raise ValueError("a batch mixes locality-ordered and plain graphs: apply graph.apply_locality_order to all of a data set's slides or none")
    p.heavy_degree = HEAVY_DEGREE_LOCALITY if p.locality else HEAVY_DEGREE
    return p, sim


def _build_plan(g: HeteroGraph, per_relation_src: bool = False) -> GraphPlan:
    dev = g.device
    hd = PlanHeader(g.ntypes, g.canonical_etypes, [g.num_nodes(t) for t in g.ntypes])
    gsrc, gdst, gseg = [], [], []
    for ri, (s, e, d) in enumerate(hd.rels):
        u, v = g._edges[(s, e, d)]
        u = u.to(dev)
        v = v.to(dev)
        ti_d = hd.tindex[d]
        gsrc.append(u + (hd.rel_rows[ri][0] if per_relation_src else hd.type_off[hd.tindex[s]]))
        gdst.append(v + hd.type_off[ti_d])
        gseg.append(hd.seg_off[ti_d] + v * hd.R[ti_d] + hd.slot_of_rel[ri])
    if gsrc:
        gsrc, gdst, gseg = torch.cat(gsrc), torch.cat(gdst), torch.cat(gseg)
    else:
        gsrc = gdst = gseg = torch.empty(0, dtype=torch.int64, device=dev)
    pos = None
    if g.ntypes and all("_pos" in g._nframes[t] for t in g.ntypes) and LOCALITY:
        pos = torch.cat([g._nframes[t]["_pos"].reshape(-1) for t in g.ntypes])
    return finish_plan(hd, gsrc, gdst, gseg, dev, per_relation_src,
                       [g.batch_num_nodes(t).tolist() for t in g.ntypes], pos=pos)


def batch(graphs: Sequence[HeteroGraph]) -> HeteroGraph:
    """Block-diagonal batch (``dgl.batch``): same ntypes/relations required."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("empty batch")
    g0 = graphs[0]
    for g in graphs[1:]:
        if g.ntypes != g0.ntypes or g.canonical_etypes != g0.canonical_etypes:
            raise ValueError("dgl.batch semantics: all graphs must share node types and relations")
    num_nodes = OrderedDict((t, sum(g.num_nodes(t) for g in graphs)) for t in g0.ntypes)
    bnn = {t: torch.cat([g.batch_num_nodes(t) for g in graphs]) for t in g0.ntypes}
    edges = OrderedDict()
    for r in g0.canonical_etypes:
        s, _, d = r
        us, vs = [], []
        so = do = 0
        for g in graphs:
            u, v = g._edges[r]
            us.append(u + so)
            vs.append(v + do)
            so += g.num_nodes(s)
            do += g.num_nodes(d)
        edges[r] = (torch.cat(us), torch.cat(vs))
    out = HeteroGraph(num_nodes, edges, bnn)
    for t in g0.ntypes:
        for k in g0._nframes[t]:
            out._nframes[t][k] = torch.cat([g._nframes[t][k] for g in graphs], dim=0)
    for r in g0.canonical_etypes:
        for k in g0._eframes[r]:
            out._eframes[r][k] = torch.cat([g._eframes[r][k] for g in graphs], dim=0)
    return out


def to_homogeneous(g: HeteroGraph, add_self_loop: bool = False) -> HeteroGraph:
    """``dgl.to_homogeneous(g, ndata=['feat', ...])`` (+ ``dgl.add_self_loop``): one node type holding all nodes in
    type-major order (DGL's order), one relation holding every edge (relations concatenated in canonical order);
    used by models/GCN_NTPool.py:90-91."""
    off = g.type_offsets()
    tindex = {t: i for i, t in enumerate(g.ntypes)}
    us, vs = [], []
    for (s, e, d) in g.canonical_etypes:
        u, v = g._edges[(s, e, d)]
        us.append(u + off[tindex[s]])
        vs.append(v + off[tindex[d]])
    n = off[-1]
    dev = g.device
    u = torch.cat(us) if us else torch.empty(0, dtype=torch.int64, device=dev)
    v = torch.cat(vs) if vs else torch.empty(0, dtype=torch.int64, device=dev)
    if add_self_loop:
        loop = torch.arange(n, dtype=torch.int64, device=u.device)
        u = torch.cat([u, loop])
        v = torch.cat([v, loop])
    out = HeteroGraph.homogeneous(n, u, v)
    keys = set.intersection(*[set(g._nframes[t].keys()) for t in g.ntypes]) if g.ntypes else set()
    for k in keys:
        if k == "_ID":
            continue
        out._nframes["_N"][k] = g.cat_ndata(k) if k == "feat" else torch.cat([g._nframes[t][k] for t in g.ntypes], dim=0)
    return out


def remove_nodes(g: HeteroGraph, nids: torch.Tensor, ntype: Optional[str] = None) -> HeteroGraph:
    """``dgl.remove_nodes(g, nids, ntype=...)`` on a single (unbatched) graph: the nodes and every edge that touches them go, the
    remaining nodes of that type keep their relative order (ids shift down), node and edge fields follow, and the set of
    relations is kept even when one becomes empty (DGL keeps the metagraph: SURVEY Appendix A.1.5)."""
    if g._batch_num_nodes is not None and g.batch_size > 1:
        raise ValueError("remove_nodes works on single graphs (the reference resets _batch_num_nodes before calling it)")
    if ntype is None:
        if len(g.ntypes) != 1:
            raise ValueError("ntype is required for a multi-type graph")
        ntype = g.ntypes[0]
    ntype = str(ntype)
    n = g.num_nodes(ntype)
    dev = g.device
    keep = torch.ones(n, dtype=torch.bool, device=dev)
    keep[torch.as_tensor(nids, dtype=torch.int64, device=dev)] = False
    new_id = torch.cumsum(keep, 0) - 1
    counts = OrderedDict((t, g.num_nodes(t)) for t in g.ntypes)
    counts[ntype] = int(keep.sum().item()) if dev.type != "cpu" else int(keep.sum())
    edges, emask = OrderedDict(), {}
    for (s, e, d) in g.canonical_etypes:
        u, v = g._edges[(s, e, d)]
        m = torch.ones(u.numel(), dtype=torch.bool, device=u.device)
        if s == ntype:
            m &= keep.to(u.device)[u]
        if d == ntype:
            m &= keep.to(v.device)[v]
        uu, vv = u[m], v[m]
        if s == ntype:
            uu = new_id.to(u.device)[uu]
        if d == ntype:
            vv = new_id.to(v.device)[vv]
        edges[(s, e, d)] = (uu, vv)
        emask[(s, e, d)] = m
    out = HeteroGraph(counts, edges)
    for t in g.ntypes:
        for k, x in g._nframes[t].items():
            out._nframes[t][k] = x[keep.to(x.device)] if t == ntype else x
    for r in g.canonical_etypes:
        for k, x in g._eframes[r].items():
            out._eframes[r][k] = x[emask[r].to(x.device)]
    return out


def permute_nodes(g: HeteroGraph, perm: Dict[str, torch.Tensor]) -> HeteroGraph:
    """The same graph with the nodes of every type renumbered: new node i of type t is old node ``perm[t][i]``.
    Node fields follow their nodes, edges are relabelled, edge order and edge fields are untouched, so every model output
    is unchanged (message passing is permutation-equivariant, the readouts are permutation-invariant)."""
    inv = {}
    for t in g.ntypes:
        p = perm[t].to(torch.int64)
        if p.numel() != g.num_nodes(t):
            raise ValueError(f"perm[{t!r}] has {p.numel()} entries for {g.num_nodes(t)} nodes")
        q = torch.empty_like(p)
        q[p] = torch.arange(p.numel(), dtype=torch.int64, device=p.device)
        inv[t] = q
    edges = OrderedDict()
    for (s, e, d) in g.canonical_etypes:
        u, v = g._edges[(s, e, d)]
        edges[(s, e, d)] = (inv[s].to(u.device)[u], inv[d].to(v.device)[v])
    if g._batch_num_nodes is not None and g.batch_size > 1:
        raise ValueError("permute single graphs before batching them (a batch's node order encodes its graphs)")
    out = HeteroGraph(OrderedDict((t, g.num_nodes(t)) for t in g.ntypes), edges)
    for t in g.ntypes:
        for k, x in g._nframes[t].items():
            out._nframes[t][k] = x[perm[t].to(x.device)]
    for r in g.canonical_etypes:
        for k, x in g._eframes[r].items():
            out._eframes[r][k] = x
    return out


def locality_order(g: HeteroGraph, positions: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """A node order under which graph neighbours get nearby ids: reverse Cuthill-McKee on the symmetrised homogeneous
    adjacency, restricted to each node type.  WSI graphs are k-NN graphs in feature space, i.e. strongly clustered; with
    this order the K/V rows one workgroup gathers for neighbouring destinations fall into a few hundred KB instead of the
    whole 20 MB table, so the gathers hit the 4 MiB L2 instead of streaming from the Infinity Cache.  One-off, on the CPU,
    per slide (scipy); apply with ``permute_nodes`` before the graph is stored / batched.  Pure performance hint: results
    do not depend on it."""
    import numpy as np
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    off = g.type_offsets()
    n = off[-1]
    tindex = {t: i for i, t in enumerate(g.ntypes)}
    rows, cols = [], []
    for (s, e, d) in g.canonical_etypes:
        u, v = g._edges[(s, e, d)]
        rows.append(u.cpu().numpy() + off[tindex[s]])
        cols.append(v.cpu().numpy() + off[tindex[d]])
    if not rows or n == 0:
        if positions is not None:
            for i, t in enumerate(g.ntypes):
                positions[t] = off[i] + torch.arange(g.num_nodes(t))
        return {t: torch.arange(g.num_nodes(t)) for t in g.ntypes}
    r, c = np.concatenate(rows), np.concatenate(cols)
    a = coo_matrix((np.ones(2 * r.size, dtype=np.int8), (np.concatenate([r, c]), np.concatenate([c, r]))), shape=(n, n)).tocsr()
    order = np.asarray(reverse_cuthill_mckee(a, symmetric_mode=True), dtype=np.int64)       # order[i] = old global id at new position i
    out = {}
    for i, t in enumerate(g.ntypes):
        m = (order >= off[i]) & (order < off[i + 1])
        out[t] = torch.from_numpy((order[m] - off[i]).copy())
        if positions is not None:
            positions[t] = torch.from_numpy(np.nonzero(m)[0].astype(np.int64))      # slide-wide position of each node of type t, new order
    return out


def apply_locality_order(g: HeteroGraph) -> HeteroGraph:
    """``permute_nodes(g, locality_order(g))`` + the node field ``'_pos'`` (position of every node in the slide-wide order,
    across node types).  A plan built from graphs that carry ``'_pos'`` walks destination and source nodes in that order and
    the attention kernels walk it XCD-contiguously (``GraphPlan.locality``): the K/V rows gathered by the waves in flight on
    one XCD are those of neighbouring patches — on kNN (real WSI) graphs a working set its 4 MiB L2 can hold; on the random
    benchmark graphs there is no such order and the default heaviest-first one is used.  Results do not depend on it."""
    positions: Dict[str, torch.Tensor] = {}
    perm = locality_order(g, positions)
    out = permute_nodes(g, perm)
    for t in out.ntypes:
        # on the graph's own device: a CPU field on a device-resident graph would make every ``g.to(device)`` copy the whole graph
        # (and drop its cached plan) - once per trainer step
        out._nframes[t]["_pos"] = positions.get(t, torch.arange(out.num_nodes(t))).to(out.device)
    return out
