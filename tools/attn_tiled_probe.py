"""L2-blocked attention kernels (csrc/heat_attn_tiled.hip) against the shipped one-wave-per-node kernels (csrc/heat_attn.hip) on the
bench batch: results (max relative difference of every output), time per launch (device events, median), for several rows-in-flight
settings.  ``--pmc`` marks the section to profile: run the script under ``rocprofv3 --pmc ... --kernel-trace`` with ``--iters 3``.

    python tools/attn_tiled_probe.py [--graphs 8] [--nodes 10000] [--hidden 512] [--heads 8] [--iters 30] [--json out.json]
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

# ---- the stream-form experiment lives in the measurement library only (csrc/heat_attn_stream_ablate.inc): its table and helpers are here, not in the package
WSI_ATTN_MAX_UNITS, WSI_ATTN_MAX_SPANS = 64, 200


class AttnStream(ctypes.Structure):
    _fields_ = [("part_ptr", ctypes.c_int32 * 9), ("unit_ptr", ctypes.c_int32 * (WSI_ATTN_MAX_UNITS + 1)), ("begin", ctypes.c_int32 * WSI_ATTN_MAX_SPANS),
                ("end", ctypes.c_int32 * WSI_ATTN_MAX_SPANS), ("inv_r", ctypes.c_float * WSI_ATTN_MAX_SPANS)]


def bind_stream(lib):
    from ctypes import POINTER, c_int32, c_int64, c_void_p
    f = lib.wsi_heat_attn_stream_aggregate
    f.restype = ctypes.c_int
    f.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(AttnStream), c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
    return f


def attn_stream_map(plan, bc, parts: int = 8):
    """``wsi_attn_stream_t`` of a plan (None when the stream kernels do not apply): a UNIT is one graph of the batch (or, for batches of fewer
    than 8 graphs, one of ``8 // B`` equal pieces of it) with one SPAN per node type - the node-id range of that (graph, type) - and units are
    dealt to the 8 parts (XCDs) largest first onto the least loaded part.  Host arithmetic only; cached on the plan."""
    hit = plan.__dict__.get("_attn_stream", False)
    if hit is not False:
        return hit
    m = None
    n = int(plan.num_nodes)
    if bc and plan.num_src_rows == n and len(bc) == len(plan.rel_slots):
        T, B = len(bc), len(bc[0])
        pieces = max(1, parts // B) if B < parts else 1
        units = []                                   # (weight, [(begin, end, inv_r), ...])
        for b in range(B):
            for pc in range(pieces):
                spans = []
                for t in range(T):
                    base = plan.type_off[t] + sum(bc[t][:b])
                    c = bc[t][b]
                    lo, hi = base + (pc * c) // pieces, base + ((pc + 1) * c) // pieces
                    if hi > lo:
                        spans.append((lo, hi, (1.0 / plan.rel_slots[t]) if plan.rel_slots[t] > 0 else 0.0))
                if spans:
                    units.append((sum(h_ - l_ for l_, h_, _ in spans), spans))
        if units and len(units) <= WSI_ATTN_MAX_UNITS and sum(len(u[1]) for u in units) <= WSI_ATTN_MAX_SPANS:
            load = [0] * parts
            of_part = [[] for _ in range(parts)]
            for i in sorted(range(len(units)), key=lambda i_: (-units[i_][0], i_)):
                p_ = min(range(parts), key=lambda q: (load[q], q))
                of_part[p_].append(i)
                load[p_] += units[i][0]
            m = AttnStream()
            g = k = 0
            m.part_ptr[0] = 0
            m.unit_ptr[0] = 0
            for p_ in range(parts):
                for i in sorted(of_part[p_]):
                    for (lo, hi, ir) in units[i][1]:
                        m.begin[k], m.end[k], m.inv_r[k] = lo, hi, ir
                        k += 1
                    g += 1
                    m.unit_ptr[g] = k
                m.part_ptr[p_ + 1] = g
    plan.__dict__["_attn_stream"] = m
    return m


def node_edge_ptr(plan) -> torch.Tensor:
    """[N + 1] int32: first CSR edge of every destination node (``rowptr[node_seg]``), once per plan."""
    ep = plan.__dict__.get("_node_eptr")
    if ep is None:
        ep = plan.__dict__["_node_eptr"] = plan.rowptr[plan.node_seg.long()].contiguous()
    return ep


def edge_dst(plan) -> torch.Tensor:
    """[E] int32: destination node of every CSR edge, expanded from ``node_edge_ptr`` once per plan."""
    ed = plan.__dict__.get("_edge_dst")
    if ed is None:
        ep = node_edge_ptr(plan).long()
        ed = torch.repeat_interleave(torch.arange(plan.num_nodes, dtype=torch.int32, device=ep.device), ep[1:] - ep[:-1], output_size=plan.num_edges)
        plan.__dict__["_edge_dst"] = ed
    return ed



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=8)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dst-mode", default="uniform")
    ap.add_argument("--only", default="", help="comma list of: old,tiled (default both); stream = the stream-form aggregate experiment and "
                    "stride = its row-pitch test (both need the measurement library: build.build_native(ablate=True))")
    ap.add_argument("--u", default="2,4,8")
    ap.add_argument("--json", default="")
    ap.add_argument("--stream-cfgs", default="4:8:256:0,4:6:1024:100,8:6:1024:100,4:5:1024:100,4:6:1024:0,4:3:1024:60,4:12:512:100,4:6:512:60,4:3:512:40,4:2:256:0,4:8:256:20,4:4:256:20")
    args = ap.parse_args()
    if args.only == "stride":
        return stride_test(args)

    from wsi_hgnn_amd import _native as N, ops, synthetic
    from wsi_hgnn_amd.graph import attn_tiles
    if "stream" in args.only:
        N.use_measurement_library()
    lib = N.load()
    dev = torch.device("cuda:0")
    g, _ = synthetic.hetero_batch(args.graphs, args.nodes, in_dim=8, dst_mode=args.dst_mode)
    bc = [g.batch_num_nodes(t).tolist() for t in g.ntypes]
    g = g.to(dev)
    plan = g.plan()
    sim = g.cat_edata_csr("sim")
    D, H = args.hidden, args.heads
    n, E, S = plan.num_nodes, plan.num_edges, plan.num_segs
    torch.manual_seed(3)
    kqv = torch.randn(n, 3 * D, device=dev) * 0.5
    g_t = torch.randn(n, D, device=dev)
    ew, eb = torch.tensor([0.7], device=dev), torch.tensor([0.3], device=dev)
    tiles = attn_tiles(plan)
    assert tiles is not None, "this plan has no tile table (hub prefix?)"
    kO, qO, vO = 0, D * 4, 2 * D * 4
    want = set(args.only.split(",")) if args.only else {"old", "tiled"}

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_.record()
            fn()
            b_.record()
            b_.synchronize()
            ts.append(a_.elapsed_time(b_) * 1e3)
        return statistics.median(ts)

    out = {"shape": dict(graphs=args.graphs, nodes=args.nodes, D=D, H=H, N=n, E=E, S=S, heavy=plan.num_heavy)}
    # ---------------------------------------------------------------- shipped kernels
    t0 = torch.empty(n, D, device=dev)
    sc0 = torch.empty(E, H, device=dev)
    ls0 = torch.zeros(S, H, device=dev)
    gargs = (N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim), N.ptr(plan.order_dst), plan.num_heavy, ops._attn_flags(plan),
             N.ptr(ew), N.ptr(eb))

    def old_fwd():
        N.check(lib.wsi_heat_attn_fwd(N.ptr(kqv, qO), 3 * D, N.ptr(kqv, kO), 3 * D, N.ptr(kqv, vO), 3 * D, n, D, H, *gargs,
                                      N.ptr(t0), D, N.ptr(sc0), N.ptr(ls0), None, N.context(), N.stream()), "fwd")

    a0 = torch.empty(E, H, device=dev)
    scr0 = torch.empty(3, E, H, device=dev)
    red = torch.empty(1024, device=dev)
    gkqv0 = torch.empty_like(kqv)
    ge0 = torch.empty(2, device=dev)

    def old_bwd():
        N.check(lib.wsi_heat_attn_bwd(
            N.ptr(kqv, qO), 3 * D, N.ptr(kqv, kO), 3 * D, N.ptr(kqv, vO), 3 * D, n, plan.num_src_rows, E, D, H,
            N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim), N.ptr(plan.colptr), N.ptr(plan.csc_eid), N.ptr(plan.csc_dst),
            N.ptr(plan.inv_rd), N.ptr(plan.order_dst), plan.num_heavy, N.ptr(plan.order_src), ops._attn_flags(plan), N.ptr(ew), N.ptr(eb),
            N.ptr(g_t), D, None, N.ptr(sc0), N.ptr(a0), N.ptr(ls0), N.ptr(scr0[0]), N.ptr(scr0[1]), N.ptr(scr0[2]), N.ptr(red),
            N.ptr(gkqv0, qO), 3 * D, N.ptr(gkqv0, kO), 3 * D, N.ptr(gkqv0, vO), 3 * D, N.ptr(ge0), None, None, N.context(), N.stream()), "bwd")

    old_fwd()
    old_bwd()
    torch.cuda.synchronize()
    if "old" in want:
        out["old"] = dict(fwd_us=timed(old_fwd, args.iters), bwd_us=timed(old_bwd, args.iters))
        print("old  ", out["old"], flush=True)

    # ---------------------------------------------------------------- tiled kernels
    def rel(x, y):
        return ((x.double() - y.double()).abs().max() / y.double().abs().max().clamp_min(1e-30)).item()

    if "tiled" in want:
        for u in [int(x) for x in args.u.split(",")]:
            flags = u << 4
            t1 = torch.empty(n, D, device=dev)
            sc1 = torch.empty(H, E, device=dev)
            ls1 = torch.zeros(H, S, device=dev)
            tmax = torch.zeros(n, H, dtype=torch.int32, device=dev)

            def tiled_fwd(v=True):
                N.check(lib.wsi_heat_attn_tiled_fwd(N.ptr(kqv, qO), 3 * D, N.ptr(kqv, kO), 3 * D, N.ptr(kqv, vO) if v else None, 3 * D,
                                                    n, E, S, D, H, N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim),
                                                    N.ptr(plan.order_dst), ctypes.byref(tiles), flags, N.ptr(ew), N.ptr(eb),
                                                    N.ptr(t1), D, N.ptr(sc1), N.ptr(ls1), N.ptr(tmax), N.stream()), "tiled fwd")

            a1 = torch.empty(H, E, device=dev)
            scr1 = torch.empty(3, H, E, device=dev)
            gkqv1 = torch.empty_like(kqv)
            ge1 = torch.empty(2, device=dev)
            gmax = torch.zeros(n, 3 * H, dtype=torch.int32, device=dev)

            def tiled_bwd():
                N.check(lib.wsi_heat_attn_tiled_bwd(
                    N.ptr(kqv, qO), 3 * D, N.ptr(kqv, kO), 3 * D, N.ptr(kqv, vO), 3 * D, n, E, S, D, H,
                    N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim), N.ptr(plan.colptr), N.ptr(plan.csc_eid), N.ptr(plan.csc_dst),
                    N.ptr(plan.inv_rd), N.ptr(plan.order_dst), N.ptr(plan.order_src), ctypes.byref(tiles), flags, N.ptr(ew), N.ptr(eb),
                    N.ptr(g_t), D, None, N.ptr(sc1), N.ptr(ls1), N.ptr(a1), N.ptr(scr1[0]), N.ptr(scr1[1]), N.ptr(scr1[2]), N.ptr(red),
                    N.ptr(gkqv1, qO), 3 * D, N.ptr(gkqv1, kO), 3 * D, N.ptr(gkqv1, vO), 3 * D, N.ptr(ge1), N.ptr(gmax), N.stream()), "tiled bwd")

            tiled_fwd()
            tiled_bwd()
            torch.cuda.synchronize()
            res = dict(
                t=rel(t1, t0), score=rel(sc1.t(), sc0), lse=rel(ls1.t(), ls0), a=rel(a1.t(), a0),
                gq=rel(gkqv1[:, D:2 * D], gkqv0[:, D:2 * D]), gk=rel(gkqv1[:, :D], gkqv0[:, :D]), gv=rel(gkqv1[:, 2 * D:], gkqv0[:, 2 * D:]),
                ge=rel(ge1, ge0),
                tmax_ok=bool(torch.equal(tmax.view(torch.float32).max(dim=1).values, t1.abs().max(dim=1).values)),
                gmax_ok=bool(torch.equal(gmax.view(torch.float32).max(dim=1).values, gkqv1.abs().max(dim=1).values)))
            t1b = t1.clone()
            g1b = gkqv1.clone()
            tiled_fwd()
            tiled_bwd()
            torch.cuda.synchronize()
            res["bit_equal_twice"] = bool(torch.equal(t1b, t1) and torch.equal(g1b, gkqv1))
            res["fwd_us"] = timed(tiled_fwd, args.iters)
            res["scores_only_us"] = timed(lambda: tiled_fwd(False), args.iters)
            res["bwd_us"] = timed(tiled_bwd, args.iters)
            out[f"tiled_u{u}"] = res
            print(f"tiled u={u}", res, flush=True)
    # ---------------------------------------------------------------- stream aggregate (experiment)
    if "stream" in want:
        stream_aggregate = bind_stream(lib)
        smap = attn_stream_map(plan, bc)
        eptr, edst = node_edge_ptr(plan), edge_dst(plan)
        eseg = ops._edge_segments(plan).long()
        a_hm = torch.exp(sc0 - ls0[eseg]).t().contiguous()             # [H, E] probabilities from the shipped forward
        cfgs = [tuple(int(x) for x in c.split(":")) for c in args.stream_cfgs.split(",")]
        for cfg in cfgs:
            (u, npg, bs, ldskb), resident = cfg[:4], (cfg[4] if len(cfg) > 4 else 0)
            t2 = torch.full((n, D), float("nan"), device=dev)
            tmax2 = torch.full((n, H), -1, dtype=torch.int32, device=dev)
            flags = (u << 4) | (npg << 8) | ({256: 0, 512: 1, 1024: 2}[bs] << 16) | (ldskb << 20) | resident      # resident: sources masked to 8192 rows

            def stream_agg():
                N.check(stream_aggregate(N.ptr(kqv, vO), 3 * D, n, E, D, H, N.ptr(eptr), N.ptr(plan.src), N.ptr(edst),
                                                           ctypes.byref(smap), flags, N.ptr(a_hm), N.ptr(t2), D, N.ptr(tmax2), N.stream()), "stream agg")
            stream_agg()
            torch.cuda.synchronize()
            res = dict(t=rel(t2, t0), tmax_ok=bool(torch.equal(tmax2.view(torch.float32).max(dim=1).values, t2.abs().max(dim=1).values)),
                       us=timed(stream_agg, args.iters))
            out[f"stream_agg_u{u}_npg{npg}_bs{bs}_lds{ldskb}_res{resident}"] = res
            print(f"stream aggregate u={u} npg={npg} bs={bs} ldsKB={ldskb} resident={resident}", res, flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


def stride_test(args):
    """One head (D = 64, H = 1) over ``--graphs`` graphs (64 = the work of 8 graphs x 8 heads): the stream aggregate gathering 256-byte slices from a
    COMPACT table (row pitch 256 B) against the same slices inside 6 KB rows (the [N, 3D] K|Q|V table): is the row pitch what the L2 trips over?"""
    from wsi_hgnn_amd import _native as N, ops, synthetic
    N.use_measurement_library()
    lib = N.load()
    stream_aggregate = bind_stream(lib)
    dev = torch.device("cuda:0")
    g, _ = synthetic.hetero_batch(args.graphs, args.nodes, in_dim=8, dst_mode=args.dst_mode)
    bc = [g.batch_num_nodes(t).tolist() for t in g.ntypes]
    g = g.to(dev)
    plan = g.plan()
    n, E = plan.num_nodes, plan.num_edges
    smap = attn_stream_map(plan, bc)
    eptr, edst = node_edge_ptr(plan), edge_dst(plan)
    torch.manual_seed(1)
    a = torch.rand(1, E, device=dev)
    out = {}
    for name, ld in (("compact_256B", 64), ("pitch_512B", 128), ("pitch_2KB", 512), ("pitch_6KB", 1536), ("pitch_6KB+128", 1568), ("pitch_8KB", 2048)):
        tab = torch.randn(n, ld, device=dev)
        t2 = torch.empty(n, 64, device=dev)
        for u, npg in ((4, 8), (8, 8)):
            flags = (u << 4) | (npg << 8)

            def run():
                N.check(stream_aggregate(N.ptr(tab), ld, n, E, 64, 1, N.ptr(eptr), N.ptr(plan.src), N.ptr(edst),
                                                           ctypes.byref(smap), flags, N.ptr(a), N.ptr(t2), 64, None, N.stream()), "stream agg")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.iters):
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record(); run(); b_.record(); b_.synchronize()
                ts.append(a_.elapsed_time(b_) * 1e3)
            out[f"{name}_u{u}"] = statistics.median(ts)
            print(name, "u", u, out[f"{name}_u{u}"], "us", flush=True)
        del tab
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
