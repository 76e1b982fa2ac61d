"""Batched graph loader — the MI355X replacement for the reference's ``GraphDataLoader`` use
(trainer/train_gnn.py:48-53: ``batch_size``, ``shuffle=True``, ``drop_last=False``) and for the per-step
``g.to(device)`` of trainer/train_gnn.py:60,64.

The reference moves every graph host->device inside the step (a 10k-node WSI graph is 41 MB of 1024-d features) and
runs one forward per graph.  Sized for 288 GB of HBM3E instead:

* ``resident=True`` (default when the data set fits in half of the free HBM — a 1000-slide TCGA cohort is ~41 GB):
  every graph's features and packed edge arrays are uploaded ONCE; a batch is assembled by device-to-device copies
  straight into the kernels' type-major ``[N, F]`` layout (no PCIe traffic in the step at all);
* ``resident=False``: features stay in pinned host memory and a side HIP stream copies each (graph, node type) block
  into its slice of one of two alternating device buffers while the previous step computes (events order buffer reuse).

Either way the CSR/CSC kernel plan of the batch is assembled on the device from per-graph, per-node-type PIECES of each
graph's own plan (``graph.PlanPieces``, built once per stored graph): the batch layout is type-major and block-diagonal,
so the plan of a batch is a concatenation of pieces plus offset additions — ~45 small launches, **no sort**, no host
sync — and the returned ``HeteroGraph`` carries the plan, the CSR-ordered ``sim`` and the already-concatenated feature
table, so the model does no further preprocessing.
Dataset parsing (DGL pickles, labels from TCGA barcodes — data.py:67-123) stays out of scope; graphs arrive as
``HeteroGraph`` objects (see INTEGRATION.md for the one-off DGL conversion).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Iterator, List, Optional, Sequence, Tuple

import torch

from . import graph as _graph_mod
from .graph import HeteroGraph, PlanHeader, PlanPieces, _resolve_device, assemble_plan, host_to_device


class StoredGraph:
    """One WSI graph in loader form: per-type features + the per-node-type pieces of its own kernel plan."""

    def __init__(self, g: HeteroGraph, label: int, device: torch.device, resident: bool):
        self.ntypes = g.ntypes
        self.rels = g.canonical_etypes
        self.num_nodes = [g.num_nodes(t) for t in self.ntypes]
        self.label = int(label)
        # the graph's own kernel plan, cut into per-node-type pieces (device-resident, ~2 MB per 10k-node graph)
        edges = OrderedDict((r, tuple(x.to(device) for x in g.edges(r))) for r in self.rels)
        sims = {r: g._eframes[r]["sim"].to(device=device, dtype=torch.float32) for r in self.rels}
        topo = HeteroGraph.from_coo(OrderedDict(zip(self.ntypes, self.num_nodes)), edges, sim=sims)
        self.edges, self.sims = edges, sims      # per-relation COO (local ids): the batch's COO is an offset-concat of these, built
        plan = topo.plan()                       # only if something asks for it (HeteroGraph._edges)
        self.num_edges = plan.num_edges
        pos = None
        if all("_pos" in g.nodes[t].data for t in self.ntypes) and _graph_mod.LOCALITY:   # graph.apply_locality_order was applied to this slide (same switch as graph._build_plan)
            pos = torch.cat([g.nodes[t].data["_pos"].reshape(-1) for t in self.ntypes])
        self.pieces = PlanPieces(PlanHeader(self.ntypes, self.rels, self.num_nodes), plan, topo.cat_edata_csr("sim"), pos)
        self.max_in_degree = self.pieces.max_in_degree
        feats = [g.nodes[t].data["feat"].to(torch.float32).contiguous() for t in self.ntypes]
        if resident:
            place = lambda x: x.to(device)
        elif device.type == "cuda":
            place = lambda x: x.pin_memory()
        else:
            place = lambda x: x
        self.feat = [place(f) for f in feats]
        self.bytes = sum(f.numel() * 4 for f in feats)
        self._feat_scale: List[Optional[torch.Tensor]] = [None] * len(feats)

    def feat_scale(self, t: int) -> torch.Tensor:
        """Row scales (absmax bits, ops.row_absmax) of the resident features of node type ``t``: taken once per stored graph."""
        if self._feat_scale[t] is None:
            from . import ops
            self._feat_scale[t] = ops.row_absmax(self.feat[t]) if self.feat[t].shape[0] else torch.empty((0, 1), dtype=torch.int32, device=self.feat[t].device)
        return self._feat_scale[t]


def _batch_coo(its: Sequence[StoredGraph], ntypes, rels):
    """Per-relation COO and ``sim`` of the block-diagonal batch of ``its`` (what ``graph.batch`` would hold): every stored
    graph's local ids shifted by the number of same-type nodes of the graphs before it.  Needed only by consumers that
    re-derive structure from the edges (HGT's per-relation-source plan, ``to_homogeneous``, ``batch``, ``save_graph``);
    the HEAT path runs on the assembled plan and never calls this."""
    tindex = {t: i for i, t in enumerate(ntypes)}
    off = [[0] * len(ntypes)]
    for it in its:
        off.append([a + b for a, b in zip(off[-1], it.num_nodes)])
    edges, efields = OrderedDict(), {}
    for r in rels:
        si, di = tindex[r[0]], tindex[r[2]]
        edges[r] = (torch.cat([it.edges[r][0] + off[b][si] for b, it in enumerate(its)]),
                    torch.cat([it.edges[r][1] + off[b][di] for b, it in enumerate(its)]))
        efields[r] = {"sim": torch.cat([it.sims[r] for it in its])}
    return edges, efields


class GraphBatchLoader:
    """Iterates over (batched HeteroGraph on ``device``, labels on ``device``)."""

    def __init__(self, graphs: Sequence[HeteroGraph], labels: Sequence[int], batch_size: int, device,
                 shuffle: bool = True, drop_last: bool = False, seed: int = 611, resident: Optional[bool] = None, passes: int = 1,
                 assemble_on_side_stream: bool = False):
        if len(graphs) != len(labels):
            raise ValueError("graphs and labels differ in length")
        if len(graphs) == 0:
            raise ValueError("empty data set")
        self.device = _resolve_device(device)          # 'cuda' -> 'cuda:<current>': devices compare by value downstream
        total = sum(g.num_nodes(t) * g.nodes[t].data["feat"].shape[1] * 4 for g in graphs for t in g.ntypes)
        if resident is None:
            free = torch.cuda.mem_get_info(self.device)[0] if self.device.type == "cuda" else 0
            resident = total < 0.5 * free
        self.resident = bool(resident)
        self.items = [StoredGraph(g, y, self.device, self.resident) for g, y in zip(graphs, labels)]
        # Slides may differ in schema (dgl.to_heterogeneous keeps only the relations that occur, and an ABSENT relation is not
        # an EMPTY one: the cross-relation mean's denominator differs), and only same-schema graphs can be batched
        # block-diagonally: every batch is drawn from one schema bucket (the reference runs such graphs one by one,
        # trainer/train_gnn.py:59-62, which is the batch-size-1 case of the same thing).
        self.buckets = OrderedDict()
        for i, it in enumerate(self.items):
            self.buckets.setdefault((tuple(it.ntypes), tuple(it.rels)), []).append(i)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        # one iterator = ``passes`` (re-shuffled) passes over the data set: the prefetch crosses the boundary between them.  (The first batch of an
        # iterator has nothing in front of it to hide its transfer behind; with a handful of batches per pass - bench.py --pcie: two - that start-up
        # cost is paid every other step.)
        self.passes = max(1, int(passes))
        self.gen = torch.Generator().manual_seed(seed)
        self.in_dim = self.items[0].feat[0].shape[1]
        # pinned-host mode: the transfers travel on ops' "work" side stream - idle while this loader feeds steps (it blocks the side streams then,
        # __iter__) - rather than on one more stream of its own: a fourth hardware queue in use stretches the step (ops._SIDE_STREAMS)
        self.copy_stream = None
        if not self.resident:
            from . import ops
            self.copy_stream = ops.side_stream(self.device, "work")
        # device-resident data set: the NEXT batch (feature concatenation + kernel plan, ~50 small kernels and one 328 MB copy) is put together
        # in line on the caller's stream, behind the step just enqueued.  ``assemble_on_side_stream=True`` moves it to a side stream; measured in round 4
        # (bench.py --pcie, hbm_resident, same box) that is the SLOWER choice - 7.86 vs 7.52 ms per step: the side stream has to start behind the
        # caller's stream anyway (the stored graphs' tensors may have work pending there), so nothing overlaps and the hand-over costs - and beside
        # the background weight gradients (ops._gemm_tn_background, a second side stream) it doubled the step (14.1 ms); with it the loader
        # therefore keeps those launches in order
        self.side_stream = (torch.cuda.Stream(device=self.device)
                            if self.resident and self.device.type == "cuda" and assemble_on_side_stream else None)
        if self.side_stream is not None:
            from . import ops
            ops.block_background_weight_gradients(True, who="loader")
        self._bufs: List[Optional[torch.Tensor]] = [None, None]
        self._free_evt: List[Optional[torch.cuda.Event]] = [None, None]

    def __len__(self) -> int:
        bs = self.batch_size
        return self.passes * sum(len(ix) // bs if self.drop_last else (len(ix) + bs - 1) // bs for ix in self.buckets.values())

    # ------------------------------------------------------------------ batch assembly
    def _assemble(self, idxs: List[int], slot: int):
        its = [self.items[i] for i in idxs]
        ntypes, rels = its[0].ntypes, its[0].rels
        B, T = len(its), len(ntypes)
        dev = self.device
        counts = [[it.num_nodes[t] for it in its] for t in range(T)]
        hd = PlanHeader(ntypes, rels, [sum(c) for c in counts])
        n = hd.N
        # ---- features -> type-major [N, F] buffer
        ready = None
        if self.resident and self.side_stream is not None:
            main = torch.cuda.current_stream(dev)
            # the stored graphs' tensors may still have work pending on the caller's stream (graphs handed over already on the
            # device are kept as they are; first-use scale scans): the side stream starts behind it
            self.side_stream.wait_stream(main)
            with torch.cuda.stream(self.side_stream):
                feat = torch.empty((n, self.in_dim), dtype=torch.float32, device=dev)
                self._copy_features(feat, its, hd)
                scales = self._cat_feature_scales(its, hd, n, dev)
                plan, sim = assemble_plan(hd, [it.pieces for it in its], dev, counts)
                labels = host_to_device([it.label for it in its], torch.int64, dev)
                ready = torch.cuda.Event()
                ready.record(self.side_stream)
            # allocated under the side stream, consumed on the caller's: the caching allocator must not hand this memory to the next
            # side-stream assembly while the caller's kernels still read it
            for t_ in [feat, sim, labels, scales] + [v for v in vars(plan).values() if isinstance(v, torch.Tensor)]:
                if t_ is not None and t_.is_cuda:
                    t_.record_stream(main)
        elif self.resident:
            feat = torch.empty((n, self.in_dim), dtype=torch.float32, device=dev)
            self._copy_features(feat, its, hd)
            scales = self._cat_feature_scales(its, hd, n, dev)
            plan, sim = assemble_plan(hd, [it.pieces for it in its], dev, counts)
            labels = host_to_device([it.label for it in its], torch.int64, dev)
        else:
            with torch.cuda.stream(self.copy_stream):
                if self._free_evt[slot] is not None:
                    self.copy_stream.wait_event(self._free_evt[slot])    # the step that read this buffer has finished
                buf = self._bufs[slot]
                if buf is None or buf.shape[0] < n:
                    buf = self._bufs[slot] = torch.empty((int(n * 1.1) + 1, self.in_dim), dtype=torch.float32, device=dev)
                feat = buf[:n]
                self._copy_features(feat, its, hd)
                # ---- kernel plan of the batch from the stored pieces (no sort, no sync), on the copy stream as well
                plan, sim = assemble_plan(hd, [it.pieces for it in its], dev, counts)
                labels = host_to_device([it.label for it in its], torch.int64, dev)
                ready = torch.cuda.Event()
                ready.record(self.copy_stream)
            scales = None
            main = torch.cuda.current_stream(dev)
            for t_ in [sim, labels] + [v for v in vars(plan).values() if isinstance(v, torch.Tensor)]:
                if t_ is not None and t_.is_cuda:
                    t_.record_stream(main)       # (the feature buffers are persistent and guarded by events instead)
        # ---- the graph object the models consume
        nn_ = OrderedDict((t, hd.counts[i]) for i, t in enumerate(ntypes))
        G = HeteroGraph._from_plan(nn_, rels, {t: torch.tensor(counts[i], dtype=torch.int64) for i, t in enumerate(ntypes)},
                                   plan, lambda: _batch_coo(its, ntypes, rels))
        parts = []
        for i, t in enumerate(ntypes):
            v = feat[hd.type_off[i]:hd.type_off[i + 1]]
            G._nframes[t]["feat"] = v
            parts.append(v)
        sig = tuple((p.data_ptr(), tuple(p.shape), p.dtype, p._version) for p in parts)
        cache = G.__dict__.setdefault("_cat_cache", {})
        cache["feat"] = (sig, feat)                       # the type-major table already IS the concatenation
        cache[("e", "sim")] = ((), sim)                   # CSR-ordered; valid while the per-relation fields are untouched
        if scales is not None:                            # fp16x3 / auto: the input projection finds the features' row scales ready
            from . import ops
            ops.attach_row_scales(feat, scales)
        return G, labels, ready

    def _cat_feature_scales(self, its, hd, n, dev) -> Optional[torch.Tensor]:
        """Resident data set under a scaled GEMM arithmetic: the row scales of the batch's feature table, concatenated from the
        stored graphs' own (each scanned once, at its first use) in the table's type-major order."""
        from . import ops
        if dev.type != "cuda" or not ops.scaled_gemm_mode():
            return None
        scales = torch.empty((n, 1), dtype=torch.int32, device=dev)
        for t in range(len(hd.ntypes)):
            a, b = hd.type_off[t], hd.type_off[t + 1]
            if b > a:
                torch.cat([it.feat_scale(t) for it in its], dim=0, out=scales[a:b])
        return scales

    def _copy_features(self, feat, its, hd) -> None:
        if self.resident:
            # device-resident data set: one concatenation KERNEL per node type.  (Per-block ``copy_`` calls are D2D
            # hipMemcpyAsync's that the runtime sometimes routes through the slow SDMA engine: sporadic 50 ms stalls.)
            for t in range(len(hd.ntypes)):
                a, b = hd.type_off[t], hd.type_off[t + 1]
                if b > a:
                    torch.cat([it.feat[t] for it in its], dim=0, out=feat[a:b])
            return
        for t in range(len(hd.ntypes)):
            row = hd.type_off[t]
            for it in its:
                k = it.num_nodes[t]
                if k:
                    feat[row:row + k].copy_(it.feat[t], non_blocking=True)
                row += k

    def __iter__(self) -> Iterator[Tuple[HeteroGraph, torch.Tensor]]:
        batches = []
        for _ in range(self.passes):
            one = []
            for ix in self.buckets.values():
                order = [ix[j] for j in torch.randperm(len(ix), generator=self.gen).tolist()] if self.shuffle else list(ix)
                bb = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
                if self.drop_last and bb and len(bb[-1]) < self.batch_size:
                    bb.pop()
                one += bb
            if self.shuffle and len(self.buckets) > 1:
                one = [one[j] for j in torch.randperm(len(one), generator=self.gen).tolist()]
            batches += one
        if not batches:
            return
        # pinned-host mode: while this iterator feeds steps, every launch of those steps stays on the caller's stream (ops.block_side_streams: beside
        # H2D transfers a second compute stream stretches the step from 6.7 to 10 ms; the transfer bounds it at ~6 ms either way)
        pinned = not self.resident and self.device.type == "cuda"
        who = f"loader-{id(self)}"
        if pinned:
            from . import ops
            ops.block_side_streams(True, who)
        try:
            slot = 0
            nxt = self._assemble(batches[0], slot)
            for bi in range(len(batches)):
                G, labels, ready = nxt
                used = slot
                # pinned-host mode: the NEXT batch is put together BEFORE this one is handed over - its H2D copies are then already queued on the
                # copy stream (behind the event of the step that last read that buffer) when the consumer enqueues its step, and run beside that
                # step on the GPU whether or not the host is running ahead of it.  A device-resident data set assembles in line on the caller's
                # stream: there the next batch's ~50 assembly kernels and feature concatenation go BEHIND the step just enqueued (after the yield),
                # not in front of it
                if not self.resident and bi + 1 < len(batches):
                    slot ^= 1
                    nxt = self._assemble(batches[bi + 1], slot)
                cur = torch.cuda.current_stream(self.device)
                if ready is not None:
                    cur.wait_event(ready)
                yield G, labels
                if self.resident and bi + 1 < len(batches):
                    slot ^= 1
                    nxt = self._assemble(batches[bi + 1], slot)
                if not self.resident:
                    evt = torch.cuda.Event()
                    evt.record(cur)                               # the consumer has enqueued its step on `cur`: the buffer is free behind it
                    self._free_evt[used] = evt
        finally:
            if pinned:
                ops.block_side_streams(False, who)
