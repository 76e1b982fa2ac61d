"""torch.autograd.Function wrappers around the C-ABI (include/wsi_hgnn.h).

PyTorch is plumbing here: it owns the device memory, the stream and the autograd tape; every
arithmetic kernel on the hot path is a hand-written HIP kernel reached through ``_native``.

Operator API mirrored (there is no native operator layer in the reference; these are the DGL /
torch calls of models/HEATNet4.py the ops stand in for):
  grouped_linear      <- nn.Linear per node type        (HEATNet4.py:100-102,134,202,219)
  heat_attention      <- apply_edges(v_dot_u) * ea / sqrt_dk ; edge_softmax ; multi_update_all(u_mul_e, sum, 'mean')
                         (HEATNet4.py:103-119)
  segment_reduce      <- dgl.readout.{sum,mean,max}_nodes  (pooling/*_pooling.py)
"""
from __future__ import annotations

import ctypes
import contextlib
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as N
from .graph import GraphPlan, host_to_device



# ------------------------------------------------------------------------------------------------
# optional per-kernel timing (bench.py): HIP events recorded on the launch stream around each C-ABI call
# ------------------------------------------------------------------------------------------------
_TIMING = {"on": False, "records": []}


def enable_kernel_timing(on: bool) -> None:
    _TIMING["on"] = bool(on)
    _TIMING["records"] = []


class _Timed:
    def __init__(self, name: str, flops: float = 0.0, mfma_flops: float = 0.0):
        self.name, self.flops, self.mfma_flops = name, flops, mfma_flops

    def __enter__(self):
        if _TIMING["on"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _TIMING["on"]:
            self.e1.record()
            _TIMING["records"].append((self.name, self.flops, self.mfma_flops, self.e0, self.e1))
        return False


def kernel_timing_summary() -> dict:
    """{family: {ms, launches, flops, mfma_flops}} over everything recorded since enable_kernel_timing(True); ``flops`` are
    algorithmic (2MNK), ``mfma_flops`` what the matrix cores execute for them (x6 under bf16x6, x3 / x6 under fp16x3)."""
    torch.cuda.synchronize()
    out = {}
    for name, flops, mfma, e0, e1 in _TIMING["records"]:
        d = out.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0.0, "mfma_flops": 0.0})
        d["ms"] += e0.elapsed_time(e1)
        d["launches"] += 1
        d["flops"] += flops
        d["mfma_flops"] += mfma
    return out

# ------------------------------------------------------------------------------------------------
# grouped GEMM
# ------------------------------------------------------------------------------------------------
_GEMM_MODES = {"fp32": N.WSI_GEMM_FP32, "bf16x6": N.WSI_GEMM_BF16X6, "fp16x3": N.WSI_GEMM_FP16X3, "auto": N.WSI_GEMM_AUTO}
_SCALED_MODES = ("fp16x3", "auto")          # modes whose launches exchange row scales
_PRECISION = {"mode": "fp32"}


def set_gemm_precision(mode: str) -> None:
    """Arithmetic of every projection GEMM this host module launches from now on (passed PER CALL to wsi_gemm_grouped; the
    library keeps no mode).  "fp32" (default): IEEE fp32 MFMA.  "bf16x6": exact 3-way bf16 split of
    both operands, 6 cross products accumulated in fp32 on the bf16 matrix cores — fp32-class error, not a reduced-precision
    mode.  "fp16x3": per-row power-of-two scaling, 2-way fp16 split (2^-23 relative, per product <= 2^-21), 3 cross products on the fp16 matrix
    cores — the same error class at half the matrix work (include/wsi_hgnn.h).  "auto": per launch, fp16x3 where its pre-pass
    is amortised (large projections), bf16x6 otherwise."""
    if mode not in _GEMM_MODES:
        raise ValueError(f"unknown GEMM precision {mode!r}; expected one of {sorted(_GEMM_MODES)}")
    _PRECISION["mode"] = mode


def gemm_precision() -> str:
    return _PRECISION["mode"]


# ------------------------------------------------------------------------------------------------
# facts that travel WITH a tensor between autograd nodes
# ------------------------------------------------------------------------------------------------
# A producer on the path knows things about the tensor it hands on that its consumer would otherwise recompute with a pass over it (the absmax
# bits of its rows: the fp16x3 scales) or could never recover (that a readout's gradient is a broadcast of S distinct rows).  The fact is attached
# to THE TENSOR OBJECT, with the version counter it was true at, and read from the object the consumer receives: autograd hands a Python tensor
# from one node to the next unchanged (PyTorch preserves a tensor's Python object while the tensor lives), so this is an argument passed
# explicitly from producer to consumer - there is no process-wide table to match against, nothing another model or thread in the process can
# evict or alias.  Whatever is not that very object at that very version (a view, a detached alias, a gradient the engine accumulated with
# another contribution - in place: the version moves - or into a new tensor) carries no valid fact and the consumer takes its general path.
def _annotate(t: torch.Tensor, name: str, payload) -> None:
    setattr(t, name, (t._version, payload))


def _annotation(t: torch.Tensor, name: str):
    hit = getattr(t, name, None)
    if hit is None or hit[0] != t._version:
        return None
    return hit[1]


EXCHANGE_STATS = {"row_scale_hits": 0, "broadcast_hits": 0}       # how often a consumer found its producer's fact (tests read these)


def attach_row_scales(t: torch.Tensor, bits: torch.Tensor) -> None:
    """``bits`` [rows, parts] int32: partial absmax bit patterns of the rows of ``t`` as it is NOW (include/wsi_hgnn.h, wsi_gemm_group_t.a_absmax)."""
    _annotate(t, "_wsi_row_scales", bits)


def row_scales_of(t: torch.Tensor) -> Optional[torch.Tensor]:
    """The row scales ``t``'s producer attached, if the arithmetic uses scales and ``t`` has not been written since; else None."""
    if _PRECISION["mode"] not in _SCALED_MODES:
        return None
    bits = _annotation(t, "_wsi_row_scales")
    if bits is not None:
        EXCHANGE_STATS["row_scale_hits"] += 1
    return bits


def _auto_scaled(rows: int, width: int) -> bool:
    """WSI_GEMM_AUTO's NT / NN rule (csrc/gemm_f32.hip::kernel_precision) for a projection that reads a [rows, width] tensor over all node types
    (K = width, about as many output columns, up to three projections per launch), conservatively: >= 12 GFLOP and K >= 384 always; on LARGE batches
    (>= 49152 rows: the step is GPU-bound) from 5 GFLOP and K >= 256."""
    flop = float(rows) * width * width * 2.0 * 3.0
    return (width >= 384 and flop >= 12e9) or (rows >= 49152 and width >= 256 and flop >= 5e9)


def _new_row_scale(rows: int, parts: int, device, width: int = 1 << 30, zero: bool = True) -> Optional[torch.Tensor]:
    """A zeroed [rows, parts] table of partial absmax bits for producers to fill (each its own slots, plain stores; the
    consumer takes the row maximum - include/wsi_hgnn.h, wsi_gemm_group_t.a_absmax), or None outside the scaled modes.
    ``width``: columns of the tensor the scales describe = the K of the projection that would consume them; under "auto" a
    narrow or short tensor gets no table (its consumer runs bf16x6 anyway - WSI_GEMM_AUTO's rule - and the table would only
    cost a fill and epilogue stores per launch: 0.4 ms per HGT step).  ``zero=False`` when the producers are known to write every
    slot of every row (slots nobody writes must read 0)."""
    mode = _PRECISION["mode"]
    if mode not in _SCALED_MODES or (mode == "auto" and not _auto_scaled(rows, width)):
        return None
    if rows <= 32:
        return None     # a tensor this short is produced and consumed by the skinny kernels (fp32 FMAs, no scales; a launch that
                        # asks for c_absmax would be kept off them, and the arithmetic must not depend on whether scales are wanted)
    return (torch.zeros if zero else torch.empty)((max(int(rows), 1), int(parts)), dtype=torch.int32, device=device)


def remember_constant_rows(x: torch.Tensor, holder=None) -> None:
    """Scaled modes: make sure ``x`` - a tensor that does not change from step to step, e.g. the input features of a resident graph - carries
    its row scales, so that the projection reading it skips its absmax pass: computed once (``wsi_row_absmax``) and attached to the tensor
    itself (``holder``, the graph that keeps ``x`` alive, is accepted for the callers' sake and not used)."""
    if _PRECISION["mode"] not in _SCALED_MODES or x.dim() != 2 or not x.is_cuda or x.stride(1) != 1:
        return
    if _PRECISION["mode"] == "auto" and not _auto_scaled(x.shape[0], x.shape[1]):
        return                                       # its consumer runs bf16x6 (WSI_GEMM_AUTO's rule)
    if _annotation(x, "_wsi_row_scales") is None:
        attach_row_scales(x, row_absmax(x))


def scaled_gemm_mode() -> bool:
    """True when the projections run (or may run) scaled-fp16 and row scales are worth keeping."""
    return _PRECISION["mode"] in _SCALED_MODES


def row_absmax(x: torch.Tensor) -> torch.Tensor:
    """[rows, 1] int32 absmax bit patterns of the rows of a 2-D fp32 tensor (``wsi_row_absmax``)."""
    N.require_cuda(x)
    bits = torch.empty((x.shape[0], 1), dtype=torch.int32, device=x.device)
    N.check(N.load().wsi_row_absmax(N.ptr(x), x.stride(0), x.shape[0], x.shape[1], N.ptr(bits), N.stream()), "wsi_row_absmax")
    return bits


def _scale_in(bits: Optional[torch.Tensor], r0: int) -> dict:
    """Group fields that hand the row scales ``bits`` ([rows, parts], starting at row ``r0``) to a projection as its A scales."""
    if bits is None:
        return {}
    return dict(a_absmax=N.ptr(bits, r0 * bits.shape[1] * 4), a_absmax_parts=bits.shape[1])


def _scale_out(bits: Optional[torch.Tensor], r0: int, col_off: int = 0) -> dict:
    """Group fields that make a projection leave the scales of the rows it writes (from row ``r0``, columns from ``col_off``)."""
    if bits is None:
        return {}
    return dict(c_absmax=N.ptr(bits, r0 * bits.shape[1] * 4), c_absmax_parts=bits.shape[1], c_absmax_first=2 * (col_off // 128))


# ------------------------------------------------------------------------------------------------
# column statistics for the scaled-fp16 weight gradients (include/wsi_hgnn.h: c_colmax / a_colmax ...)
# ------------------------------------------------------------------------------------------------
class ColStats:
    """What a producer knows about the COLUMNS of a tensor it wrote: per row range (one per grouped-GEMM group = node type) partial absmax bits
    (and partial sums): ``bits`` [parts, width] int32, ``sums`` [parts, width] fp32 or None; ``ranges[(r0, r1)] = (first part, parts)`` of the rows
    [r0, r1) (a GEMM epilogue: one part per 128-row tile).  ``col0``: column of the tables that column 0 of the annotated tensor is (the aggregate t
    of a HEAT layer is bounded column by column by V, whose statistics sit at columns [2D, 3D) of the K|Q|V table's)."""

    def __init__(self, bits, sums, ranges, col0=0):
        self.bits, self.sums, self.ranges, self.col0 = bits, sums, dict(ranges), int(col0)

    @classmethod
    def allocate(cls, rows, width: int, device, sums: bool) -> Optional["ColStats"]:
        """Tables for a grouped NT / NN launch whose groups write the row ranges ``rows`` (one part per 128-row tile of each distinct range)."""
        ranges, p = {}, 0
        for (a, b) in rows:
            if (a, b) not in ranges and b > a:
                n = (b - a + 127) // 128
                ranges[(a, b)] = (p, n)
                p += n
        if p == 0:
            return None
        bits = torch.empty((p, width), dtype=torch.int32, device=device)
        return cls(bits, torch.empty((p, width), dtype=torch.float32, device=device) if sums else None, ranges)

    def shifted(self, col0: int) -> "ColStats":
        """The same tables seen from a tensor whose column 0 is their column ``col0``."""
        return ColStats(self.bits, self.sums, self.ranges, self.col0 + col0)

    def produce(self, r0: int, r1: int, col: int) -> dict:
        """Group fields that make an NT / NN group writing rows [r0, r1), columns from ``col`` on, leave its statistics."""
        hit = self.ranges.get((r0, r1))
        if hit is None:
            return {}
        w = self.bits.shape[1]
        d = dict(c_colmax=N.ptr(self.bits, (hit[0] * w + col) * 4), c_col_ld=w)
        if self.sums is not None:
            d["c_colsum"] = N.ptr(self.sums, (hit[0] * w + col) * 4)
        return d

    def consume(self, side: str, r0: int, r1: int, col: int, want_sums: bool = False) -> dict:
        """Group fields of a TN group whose operand ``side`` ('a' / 'b') is rows [r0, r1), columns from ``col`` on, of the annotated tensor
        (empty when the range is not one the producer wrote, or sums are wanted and were not kept)."""
        hit = self.ranges.get((r0, r1))
        if hit is None or (want_sums and self.sums is None):
            return {}
        w = self.bits.shape[1]
        off = (hit[0] * w + self.col0 + col) * 4
        d = {side + "_colmax": N.ptr(self.bits, off), side + "_col_ld": w, side + "_col_parts": hit[1]}
        if side == "a" and want_sums:
            d["a_colsum"] = N.ptr(self.sums, off)
        return d


def attach_col_stats(t: torch.Tensor, stats: Optional[ColStats]) -> None:
    if stats is not None:
        _annotate(t, "_wsi_col_stats", stats)


def col_stats_of(t: torch.Tensor) -> Optional[ColStats]:
    """The column statistics ``t``'s producer attached, if the arithmetic uses scales and ``t`` has not been written since; else None."""
    if _PRECISION["mode"] not in _SCALED_MODES:
        return None
    st = _annotation(t, "_wsi_col_stats")
    if st is not None:
        EXCHANGE_STATS["col_stat_hits"] = EXCHANGE_STATS.get("col_stat_hits", 0) + 1
    return st


_TN_AUTO = {"flop": 4e9}          # WSI_GEMM_AUTO's weight-gradient threshold (csrc/gemm_f32.hip::kernel_precision; tools move both for A/B runs)


def want_col_stats(total_rows: int, out_cols: int, in_cols: int) -> bool:
    """Will the weight gradient dW [out_cols, in_cols] over ``total_rows`` rows run on the scaled-fp16 TN kernel (so that its operands' producers
    should leave column statistics)?  WSI_GEMM_AUTO's rule for TN launches (csrc/gemm_f32.hip::kernel_precision), conservatively."""
    mode = _PRECISION["mode"]
    if mode not in _SCALED_MODES:
        return False
    flop = 2.0 * total_rows * out_cols * in_cols
    return mode == "fp16x3" or (min(out_cols, in_cols) >= 192 and ((flop >= 3e10 and total_rows >= 3 * 2048) or (flop >= _TN_AUTO["flop"] and total_rows >= 49152)))


_SIDE_STATS = {"enabled": True, "launches": 0, "reasons": set()}
# HIP streams of this module beside the caller's.  ONE by default: the background weight gradients ("work") and the column statistics / early
# skip-gate gradient ("stats") take turns on it (measured equal to two: 5.85 against 5.87 ms per step, bench.py --side-streams).  Why the count
# matters: the runtime gives every stream in use a hardware queue of its own up to GPU_MAX_HW_QUEUES (4) and lets further streams SHARE queues.
# Measured on this GPU (round 5, bench.py --pcie; profiles/r05_pcie_side_streams.txt):
#   * a FOURTH hardware queue in use beside H2D transfers stretches every kernel of the step (small launches to ~50 us each): 10.0 ms per
#     loader-fed step against 6.7 - with the caller's stream, two streams here and the loader's copy stream; gone under GPU_MAX_HW_QUEUES=3 and
#     with four unrelated streams used in between (the copy stream then shares a queue);
#   * two of these streams SHARING one queue serialise behind each other: 7.5 ms per step against 5.9.
# So: one side stream; the pinned-host loader borrows it for its transfers (it keeps the steps it feeds in order anyway - block_side_streams) rather
# than making a stream of its own; a data-parallel step has the caller's stream, this one and RCCL's.
_SIDE_STREAMS = {"count": 1, "streams": {}}


def set_side_stream_count(n: int) -> None:
    if n not in (1, 2):
        raise ValueError("set_side_stream_count: 1 or 2")
    _SIDE_STREAMS["count"] = int(n)          # (streams already made are kept: a new one would take - or, past the runtime's limit, SHARE - one more hardware queue)


def side_stream(device, role: str = "work") -> "torch.cuda.Stream":
    dev = torch.device(device)
    key = (dev.index, role if _SIDE_STREAMS["count"] == 2 else "work")
    st = _SIDE_STREAMS["streams"].get(key)
    if st is None:
        st = _SIDE_STREAMS["streams"][key] = torch.cuda.Stream(device=dev)
    return st


def _side_stats_on() -> bool:
    return _SIDE_STATS["enabled"] and not _SIDE_STATS["reasons"] and not torch.cuda.is_current_stream_capturing()


def block_side_streams(blocked: bool, who: str) -> None:
    """Keep EVERY launch of the step on the caller's stream while ``who`` says so: no background weight gradients, no statistics stream.
    (data.GraphBatchLoader while it feeds steps from pinned host memory: with H2D transfers in flight, kernels of a second compute stream
    stretch the whole step - the small launches of the caller's stream to ~50 us each - 10.0 ms per loader-fed step against 6.7 in order,
    profiles/r05_pcie_side_streams.txt; the transfer, 5.9 ms at 55 GB/s, bounds that step anyway.)"""
    (_SIDE_STATS["reasons"].add if blocked else _SIDE_STATS["reasons"].discard)(who)
    block_background_weight_gradients(blocked, who=who)


def set_side_column_statistics(on: bool) -> None:
    """On (default): the column statistics of the attention gradients (no producer leaves them: one wave per node, four nodes per workgroup) are taken
    by ``wsi_col_stats`` on a stream of its own, beside the matrix-bound dX projection that reads the same table, instead of in line inside the weight
    gradient (its own pass: 0.15 ms per step on the bench batch, bandwidth-bound).  Off: the weight gradient makes its pass."""
    _SIDE_STATS["enabled"] = bool(on)


def _col_stats_side(x: torch.Tensor, rows: Sequence[Tuple[int, int]], device):
    """(ColStats of the row ranges of ``x`` - partial absmax bits and sums per 256 rows -, event) computed on the statistics stream, which first waits
    for everything the caller's stream holds so far; None inside a stream capture or when switched off.  The consumer waits for the event."""
    if not _side_stats_on():
        return None
    lib = N.load()
    dev = torch.device(device)
    width = x.shape[1]
    wpad = (width + 3) & ~3
    ranges, p = {}, 0
    for (a, b) in rows:
        if (a, b) not in ranges and b > a:
            n_ = lib.wsi_col_stats_parts(b - a)
            ranges[(a, b)] = (p, n_)
            p += n_
    if p == 0:
        return None
    bits = torch.empty((p, wpad), dtype=torch.int32, device=dev)
    sums = torch.empty((p, wpad), dtype=torch.float32, device=dev)
    side = side_stream(dev, "stats")
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    for (a, b), (p0, n_) in ranges.items():
        N.check(lib.wsi_col_stats(N.ptr(x, a * x.stride(0) * 4), x.stride(0), b - a, width, N.ptr(bits, p0 * wpad * 4), N.ptr(sums, p0 * wpad * 4), wpad,
                                  ctypes.c_void_p(side.cuda_stream)), "wsi_col_stats")
        _SIDE_STATS["launches"] += 1
    ev = torch.cuda.Event()
    ev.record(side)
    for t_ in (x, bits, sums):
        t_.record_stream(side)
    return ColStats(bits, sums, ranges), ev


def remember_constant_cols(x: torch.Tensor, rows: Sequence[Tuple[int, int]]) -> None:
    """Scaled modes: make sure ``x`` - a weight-gradient operand that does not change from step to step: the input features of a resident graph -
    carries its column statistics (one part per row range: ``wsi_col_absmax``, once), so that the input projection's weight gradient skips its pass
    over it."""
    if _PRECISION["mode"] not in _SCALED_MODES or x.dim() != 2 or not x.is_cuda or x.stride(1) != 1:
        return
    st = _annotation(x, "_wsi_col_stats")
    if st is not None and all(r in st.ranges for r in rows if r[1] > r[0]):
        return
    lib = N.load()
    width = x.shape[1]
    ranges, bits_rows = {}, []
    for (a, b) in rows:
        if (a, b) in ranges or b <= a:
            continue
        out = torch.empty(width, dtype=torch.int32, device=x.device)
        nbytes = lib.wsi_col_absmax_workspace_bytes(b - a, width)
        ws = torch.empty(max(nbytes // 4, 4), dtype=torch.int32, device=x.device)
        N.check(lib.wsi_col_absmax(N.ptr(x, a * x.stride(0) * 4), x.stride(0), b - a, width, N.ptr(out), N.ptr(ws), nbytes, N.stream()), "wsi_col_absmax")
        ranges[(a, b)] = (len(bits_rows), 1)
        bits_rows.append(out)
    if bits_rows:
        attach_col_stats(x, ColStats(torch.stack(bits_rows), None, ranges))


# ------------------------------------------------------------------------------------------------
# weights packed once per optimizer step (wsi_gemm_group_t.b_packed)
# ------------------------------------------------------------------------------------------------
# The scaled-fp16 NT / NN kernel reads its B operand (the weights) as two fp16 planes in MFMA fragment order; wsi_gemm_grouped packs them per call -
# one more launch in front of every projection, seven per step, for weights that change once per step.  Here the packed form is kept per
# (op, weights, chunking): ``repack_weights`` (called by optim.Adam.step behind its update) refreshes every entry in ONE launch per op and ARMS
# it; a projection uses an armed entry once (a weight is read once per step in each form: NT in the forward, NN in the backward) and disarms it.
# An entry nobody re-armed is packed again at its next use - so under any OTHER optimizer nothing is ever reused across a parameter update:
# version counters alone cannot be trusted with that (torch.optim.Adam(fused=True, capturable=True) updates parameters without moving them -
# found with tools/graph_capture_probe.py: a cache keyed on versions trained on stale weights).  They are still checked (load_state_dict, manual
# in-place surgery between the optimizer step and the next forward); a change through ``.data`` between those two points would go unnoticed:
# ``invalidate_packed_weights()`` after such surgery, or ``set_packed_weight_cache(False)``.
_PACKED = {"enabled": True, "entries": {}, "hits": 0, "packs": 0}


def set_packed_weight_cache(on: bool) -> None:
    _PACKED["enabled"] = bool(on)
    if not on:
        _PACKED["entries"].clear()


def invalidate_packed_weights() -> None:
    _PACKED["entries"].clear()


def _pack_groups(op: int, items, arm: bool) -> None:
    """``items``: (entry, group dict) pairs to (re)pack; one wsi_gemm_pack_b launch per WSI_GEMM_MAX_GROUPS of them."""
    lib = N.load()
    for i in range(0, len(items), N.WSI_GEMM_MAX_GROUPS):
        chunk = items[i:i + N.WSI_GEMM_MAX_GROUPS]
        arr = _group_array([g for _, g in chunk])
        N.check(lib.wsi_gemm_pack_b(op, arr, len(chunk), N.stream()), "wsi_gemm_pack_b")
        _PACKED["packs"] += 1
        for e, _ in chunk:
            e["versions"] = tuple(w._version for w in e["weights"])
            e["armed"] = arm


def _attach_packed(op: int, chunk: Sequence[dict], arr) -> None:
    """Give every group of an NT / NN launch that runs scaled-fp16 the packed form of its weights if an ARMED one exists (``repack_weights``), and
    register the weights so that the next ``repack_weights`` packs them; a group without an armed entry is packed by the call itself, as always."""
    import weakref
    ent = _PACKED["entries"]
    for gi, g in enumerate(chunk):
        ws_ = g.get("Bw")
        if not ws_ or any(not isinstance(w, torch.nn.Parameter) for w in ws_):
            continue                                  # (only module PARAMETERS live from step to step.  A computed weight - HGT's relation-folded projections, and
                                                      # under no_grad also every per-call temporary: LEConv's concatenated weight, a sliced attention vector -
                                                      # is a new tensor per call whose entry would push the real parameters out of the table)
        key = (op, tuple(w.data_ptr() for w in ws_), g["N"], g["K"], g.get("b_chunk", 0), g["ldb"])
        e = ent.get(key)
        if e is not None and any(r() is not w for r, w in zip(e["refs"], ws_)):
            e = None                                  # the address was recycled for another tensor
        if e is None:
            if len(ent) >= 256:
                dead = [k_ for k_, e_ in ent.items() if any(r() is None for r in e_["refs"])]
                for k_ in dead:
                    ent.pop(k_)                      # entries of weights that no longer exist go first
                if len(ent) >= 256:
                    ent.pop(next(iter(ent)))
            nbytes = N.load().wsi_gemm_packed_b_bytes(g["N"], g["K"])
            e = ent[key] = {"buf": torch.empty(nbytes // 4, dtype=torch.int32, device=ws_[0].device), "refs": [weakref.ref(w) for w in ws_],
                            "weights": None, "versions": None, "op": op, "armed": False,
                            "group": dict(B=g["B"], B1=g.get("B1"), B2=g.get("B2"), ldb=g["ldb"], N=g["N"], K=g["K"], b_chunk=g.get("b_chunk", 0))}
            e["group"]["b_packed"] = N.ptr(e["buf"])
            continue
        if e["armed"] and e["versions"] == tuple(w._version for w in ws_):
            _PACKED["hits"] += 1
            e["armed"] = False                        # one use per refresh
            arr[gi].b_packed = e["group"]["b_packed"]


def repack_weights() -> None:
    """Refresh and arm every packed weight (optim.Adam.step calls this behind its update): one launch per op instead of one in front of every
    projection of the next step."""
    if not _PACKED["enabled"] or not _PACKED["entries"] or torch.cuda.is_current_stream_capturing():
        return                                       # (inside a stream capture every projection packs for itself: the recorded step must not depend on this cache)
    by_op = {}
    dead = []
    for key, e in _PACKED["entries"].items():
        ws_ = [r() for r in e["refs"]]
        if any(w is None for w in ws_):
            dead.append(key)
            continue
        e["weights"] = ws_
        by_op.setdefault(e["op"], []).append((e, e["group"]))
    for key in dead:
        _PACKED["entries"].pop(key, None)
    for op, items in by_op.items():
        _pack_groups(op, items, arm=True)
    for e in _PACKED["entries"].values():
        e["weights"] = None


def _gemm(op: int, epilogue: int, groups: Sequence[dict], device) -> bool:
    """Launch wsi_gemm_grouped (chunks of WSI_GEMM_MAX_GROUPS). Each group dict: A,B,C(+bias,R,gate) as
    (tensor, byte_offset) or raw ints, lda/ldb/ldc/ldr, M,N,K.  Returns True when every group that asked for column statistics
    (``c_colmax``) got them (``wsi_gemm_writes_colstats``: only the LDS-DMA scaled-fp16 kernel leaves them)."""
    lib = N.load()
    prec = _GEMM_MODES[_PRECISION["mode"]]
    groups = [g for g in groups if g["M"] > 0 and g["N"] > 0]
    wrote = True
    for i in range(0, len(groups), N.WSI_GEMM_MAX_GROUPS):
        chunk = groups[i:i + N.WSI_GEMM_MAX_GROUPS]
        # positional construction: one C call per group (field-by-field assignment costs ~25 attribute stores each)
        arr = _group_array(chunk)
        if any(g.get("c_colmax") for g in chunk):
            wrote = wrote and bool(lib.wsi_gemm_writes_colstats(op, prec, arr, len(chunk)))
        ws = None
        ws_bytes = 0
        kernel = lib.wsi_gemm_kernel_precision(op, prec, arr, len(chunk))      # resolves "auto" / the TN launches of fp16x3
        if kernel == N.WSI_GEMM_FP16X3 and op != N.WSI_GEMM_TN and _PACKED["enabled"] and not torch.cuda.is_current_stream_capturing():
            _attach_packed(op, chunk, arr)
        if op == N.WSI_GEMM_TN or kernel == N.WSI_GEMM_FP16X3:
            ws_bytes = lib.wsi_gemm_workspace_bytes(op, prec, arr, len(chunk))
            ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=device)
        if not _TIMING["on"]:               # (the common case: no event pairs, no flop count - this function runs ~20 times per step)
            N.check(lib.wsi_gemm_grouped(op, epilogue, prec, arr, len(chunk), N.ptr(ws), ws_bytes, N.stream()), "wsi_gemm_grouped")
            continue
        flops = sum(2.0 * g["M"] * g["N"] * g["K"] for g in chunk)
        products = {N.WSI_GEMM_FP32: 1.0, N.WSI_GEMM_BF16X6: 6.0, N.WSI_GEMM_FP16X3: 3.0}[kernel]
        with _Timed("gemm", flops, flops * products), _Timed(("gemm_nt", "gemm_nn", "gemm_tn")[op] + ("_fp32", "_bf16x6", "_fp16x3")[kernel], flops, flops * products):
            N.check(lib.wsi_gemm_grouped(op, epilogue, prec, arr, len(chunk), N.ptr(ws), ws_bytes, N.stream()), "wsi_gemm_grouped")
    return wrote


_GROUP_FIELDS = [f[0] for f in N.GemmGroup._fields_]
_GROUP_INDEX = {name: i for i, name in enumerate(_GROUP_FIELDS)}
_GROUP_DEFAULT = tuple(1.0 if name == "drop_scale" else (None if N.GemmGroup._fields_[i][1] is ctypes.c_void_p else 0) for i, name in enumerate(_GROUP_FIELDS))


def _group_array(chunk):
    """The wsi_gemm_group_t table of a launch from its group dicts (keys = the struct's field names; anything else - ``Bw`` - is the host's own).
    Positional construction from a default row with only the PRESENT keys written: the struct has 43 fields, a group sets about a dozen, and this
    runs ~35 times per step (one ``dict.get`` per field was 0.6 ms of a launch-bound one-slide step)."""
    index, default = _GROUP_INDEX, _GROUP_DEFAULT
    rows = []
    for g in chunk:
        vals = list(default)
        for k, v in g.items():
            i = index.get(k)
            if i is not None:
                vals[i] = v
        rows.append(N.GemmGroup(*vals))
    return (N.GemmGroup * len(rows))(*rows)


_SMALL_PAIR = {"enabled": True}


def _gemm_small_pair(dx_groups: Sequence[dict], dx_epilogue: int, dw_groups: Sequence[dict], dw_epilogue: int, device) -> bool:
    """dX = dY W (NN groups) and dW = dY^T X (+ db; TN groups) of small Linear layers in ONE launch (``wsi_gemm_small_pair``) - the classifier
    head's levels are one row per graph.  False (nothing launched) when a group is not small or asks for scales: the caller makes its two calls."""
    dx_groups = [g for g in dx_groups if g["M"] > 0 and g["N"] > 0]
    dw_groups = [g for g in dw_groups if g["M"] > 0 and g["N"] > 0]
    if (not _SMALL_PAIR["enabled"] or not dx_groups or not dw_groups or len(dx_groups) + len(dw_groups) > 16
            or any(g["M"] > 32 or g.get("c_absmax") or g.get("b_chunk") for g in dx_groups) or any(g["K"] > 32 or g.get("c_absmax") for g in dw_groups)):
        return False
    flops = sum(2.0 * g["M"] * g["N"] * g["K"] for g in list(dx_groups) + list(dw_groups))
    with _Timed("gemm", flops, flops), _Timed("gemm_small_pair", flops, flops):
        N.check(N.load().wsi_gemm_small_pair(_group_array(dx_groups), len(dx_groups), dx_epilogue, _group_array(dw_groups), len(dw_groups), dw_epilogue,
                                             N.stream()), "wsi_gemm_small_pair")
    return True


# ------------------------------------------------------------------------------------------------
# weight gradients in the background
# ------------------------------------------------------------------------------------------------
# A layer's weight gradients are read by nobody before the optimizer; the attention backward of the layer BELOW is bound by what the fabric
# delivers to its gathers and leaves the matrix cores idle.  The dW launches issued ahead of such a phase (a layer's a_linear dW, and the
# K|Q|V dW of the layer above) therefore go to a second stream, capped at one workgroup per CU (WSI_EPI_BACKGROUND) so that the attention
# waves keep their share of registers and LDS: measured on the bench shapes 1.47 -> 1.29 ms for one dW + one attention backward
# (tools/overlap_probe.py; uncapped: 1.37).  The main stream waits for the side stream once, when the whole backward pass is over
# (autograd's final callback) - before anything can read a gradient.  Off while a data-parallel bucket is armed: its hooks pack gradients
# while backward is still running.
_BACKGROUND = {"enabled": True, "min_flop": 3.0e10, "queued": [], "pending": [], "armed": False, "blocked": False,
               "launches": 0, "task": -1, "written": set()}
# "armed" / "task": the backward pass (autograd graph task id) whose final callback will join the side stream.  The state is keyed to that pass: a pass
# that RAISES skips its final callbacks (OOM on a large slide, a hook error), and whatever it left behind - the flag, queued launches that pin their
# operands, streams nobody waits for - is cleared by the next thing that could be hurt by it (``_background_recover``: the next pass's first queueing
# attempt, every fused-layer forward, ``optim.Adam.step``).  "written": ids of the parameters whose gradient buffers the side stream is still writing.


def set_background_weight_gradients(on: bool) -> None:
    _BACKGROUND["enabled"] = bool(on)


def block_background_weight_gradients(blocked: bool, who: str = "bucket") -> None:
    """dist.GradBucket.arm(): gradients are consumed by hooks DURING backward - every launch stays on the caller's stream.  (``who``: the
    blockers are independent - an armed bucket, a loader that prefetches on a side stream of its own.)"""
    reasons = _BACKGROUND.setdefault("reasons", set())
    (reasons.add if blocked else reasons.discard)(who)
    _BACKGROUND["blocked"] = bool(reasons)


def _background_flush(device, to_side: bool = True) -> None:
    """Launch the queued weight-gradient GEMMs.  Called where an attention backward is about to be launched (``to_side``: on the side stream, which
    first waits for everything the caller's stream holds so far - they then run UNDER that attention phase and not beside the projection
    GEMMs in front of it) and at the join (whatever is still queued has no such phase left to hide under: in order, on the caller's stream)."""
    st = _BACKGROUND
    if not st["queued"]:
        return
    queued, st["queued"] = [q for q in st["queued"] if _outputs_alive(q[4])], []
    queued = [q[:4] for q in queued]
    if not queued:
        return
    dev = torch.device(device)
    if not to_side:
        for epilogue, groups, keep, events in queued:
            for ev in events:
                torch.cuda.current_stream(dev).wait_event(ev)
            _gemm(N.WSI_GEMM_TN, epilogue, groups, dev)
        return
    side = side_stream(dev, "work")
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for epilogue, groups, keep, events in queued:
            for ev in events:
                side.wait_event(ev)
            _gemm(N.WSI_GEMM_TN, epilogue | N.WSI_EPI_BACKGROUND, groups, dev)
            st["launches"] += 1
    st["pending"].append((side, [k for _, _, keep, _ in queued for k in keep]))


def _outputs_alive(outs) -> bool:
    """Do the buffers a queued launch WRITES still exist?  The queue must not hold them (AccumulateGrad adopts a gradient only when nobody else
    references it), so each is remembered as (weak reference to the tensor the backward returned, weak reference to its parameter, address): alive
    while autograd still holds the tensor, or once the parameter's ``.grad`` IS that buffer.  Neither: the pass that queued the launch died and its
    gradients were released - the launch is dropped."""
    for tref, pref, ptr in outs:
        if tref() is not None:
            continue
        p = pref()
        if p is None or p.grad is None or p.grad.data_ptr() != ptr:
            return False
    return True


def _background_wait_pending() -> None:
    """Order the caller's stream behind every side stream that carries weight gradients; drop the references that kept the operands alive."""
    st = _BACKGROUND
    pend, st["pending"] = st["pending"], []
    done = set()
    for side, _ in pend:
        if id(side) not in done:
            torch.cuda.current_stream(side.device).wait_stream(side)
            done.add(id(side))
    st["written"].clear()


def _background_join() -> None:
    """End of the backward pass: launch what is still queued, order the caller's stream behind the side stream, drop the references that kept the
    operands alive."""
    st = _BACKGROUND
    st["armed"] = False
    st["task"] = -1
    if st["queued"]:
        _background_flush(st["queued"][0][2][0].device, to_side=False)
    _background_wait_pending()


def _background_recover() -> None:
    """Leftovers of a backward pass that never reached its final callback (it raised): queued launches are DROPPED (their pass is dead, nobody will
    read those gradients; launching them now would only write into buffers of a finished step), what the side stream already holds is waited for.
    A no-op in the normal case - called from places that run between passes (a fused layer's forward, the optimizer step) and from the first
    queueing attempt of a new pass."""
    st = _BACKGROUND
    if not (st["armed"] or st["queued"] or st["pending"]):
        return
    cur = torch._C._current_graph_task_id()
    if st["armed"] and st["task"] == cur:
        return                                       # the pass that armed it is still running (a forward inside a backward: checkpointing)
    if cur >= 0:
        return                                       # inside ANOTHER backward pass (nested in the arming one): the arming pass owns the queue and the flag
    st["armed"] = False
    st["task"] = -1
    st["queued"] = []
    _background_wait_pending()


def _background_safe(params: Sequence[Optional[torch.Tensor]]) -> bool:
    """May the gradients of ``params`` be written by the side stream?  Only when each goes STRAIGHT into an AccumulateGrad that adopts the buffer
    without reading it: a leaf with an empty ``.grad``, no tensor hook and no post-accumulate-grad hook (a data-parallel bucket's hooks, a user's
    gradient clipping hook: they would read the buffer on the caller's stream before the side stream has written it), and not already being
    written by this pass (a parameter used twice in the graph: weight tying - the second contribution would be ADDED to the first at once)."""
    st = _BACKGROUND
    foreign = st["armed"] and st["task"] != torch._C._current_graph_task_id()      # another pass holds the queue: see _gemm_tn_background
    for p in params:
        if p is None:
            continue
        if foreign and id(p) not in st["written"]:
            continue
        if (not p.is_leaf or p.grad_fn is not None or p.grad is not None or getattr(p, "_backward_hooks", None)
                or getattr(p, "_post_accumulate_grad_hooks", None) or id(p) in st["written"]):
            if id(p) in st["written"]:
                # the earlier contribution is still in flight on the side stream: AccumulateGrad is about to read it - finish it first
                if st["queued"]:
                    _background_flush(p.device, to_side=False)
                _background_wait_pending()
            return False
    return not foreign


def _gemm_tn_background(epilogue: int, groups: Sequence[dict], device, keep: Sequence[torch.Tensor], written: Sequence[torch.Tensor] = (), events=(),
                        outs=()) -> bool:
    """Queue a weight-gradient GEMM for the side stream (returns False - nothing queued - when the mechanism is off).  ``keep``: every tensor
    the launch READS (at least one); held until the join so that the allocator cannot hand their memory to a later allocation.  The gradients it
    WRITES must reach autograd with no second reference to them and meet an empty ``.grad`` of a plain leaf without hooks (``written``: those
    parameters; the callers check them with ``_background_safe`` first): AccumulateGrad then adopts the buffer without reading it; with a reference
    left, a gradient to add to or a hook, something would read it on the caller's stream at once - before the values exist."""
    st = _BACKGROUND
    if not st["enabled"] or st["blocked"] or torch.is_grad_enabled():        # (create_graph=True: AccumulateGrad never adopts a buffer)
        return False
    dev = torch.device(device)
    if dev.type != "cuda":
        return False
    # worth it only under a long attention backward: on one 10k-node slide the step is launch-bound and the capped launch + the join cost more than
    # they hide (HEATNet4 2.75 -> 3.35 ms eager; replayed as a hipGraph 1.6 -> 2.8 ms: tools/r04_probe_a.sh) - small launches and captures stay in order
    if sum(2.0 * g["M"] * g["N"] * g["K"] for g in groups) < st["min_flop"] or torch.cuda.is_current_stream_capturing():
        return False
    task = torch._C._current_graph_task_id()
    if task < 0:
        return False                                                                       # not inside a backward pass: stay in order
    if st["armed"] and st["task"] != task:
        # Another backward pass holds the queue: a reentrant pass nested inside it (torch.utils.checkpoint(use_reentrant=True), autograd.backward in a
        # hook: the outer pass is ALIVE and will read its gradients) - or a new pass right after one that raised.  The two cannot be told apart from
        # here, so nothing is dropped and nothing is touched: this pass's launches stay in order (a parameter the other pass is still writing is
        # finished first: _background_safe), the queue stays with the pass that armed it - its own final callback joins it; a dead pass's leftovers
        # are dropped by the next forward / optimizer step (_background_recover outside any backward pass) and a launch whose output buffers are
        # gone is never issued (_outputs_alive).
        return False
    if not st["armed"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_background_join)     # runs once, after the last node of this backward pass
        except RuntimeError:
            return False
        st["armed"] = True
        st["task"] = task
    import weakref
    st["queued"].append((epilogue, list(groups), list(keep), list(events),
                         [(weakref.ref(g_), weakref.ref(p_), g_.data_ptr()) for p_, g_ in outs if p_ is not None and g_ is not None]))
    st["written"].update(id(p) for p in written if p is not None)
    return True


class LinearSpec:
    """Static description of a grouped linear: group i maps rows ``rows[i]`` of x through weight i into rows
    ``out_rows[i]`` (default: the same rows) and columns ``[col_off[i], col_off[i]+out_i)`` of y."""

    def __init__(self, rows: Sequence[Tuple[int, int]], col_off: Sequence[int], out_cols: int, num_rows: int,
                 out_rows: Optional[Sequence[Tuple[int, int]]] = None, num_out_rows: Optional[int] = None):
        self.rows = [(int(a), int(b)) for a, b in rows]
        self.out_rows = [(int(a), int(b)) for a, b in (out_rows if out_rows is not None else rows)]
        for (a, b), (c, d) in zip(self.rows, self.out_rows):
            if b - a != d - c:
                raise ValueError("LinearSpec: input and output row ranges differ in length")
        self.col_off = [int(c) for c in col_off]
        self.out_cols = int(out_cols)
        self.num_rows = int(num_rows)
        self.num_out_rows = int(num_out_rows if num_out_rows is not None else num_rows)
        # distinct INPUT row ranges (dX accumulation rounds) and distinct OUTPUT row ranges (bias column sums)
        self.ranges: List[Tuple[int, int]] = []
        self.range_of: List[int] = []
        self.oranges: List[Tuple[int, int]] = []
        self.orange_of: List[int] = []
        for r, o in zip(self.rows, self.out_rows):
            if r not in self.ranges:
                self.ranges.append(r)
            self.range_of.append(self.ranges.index(r))
            if o not in self.oranges:
                self.oranges.append(o)
            self.orange_of.append(self.oranges.index(o))
        self.in_covered = sum(b - a for a, b in self.ranges) == self.num_rows
        self._rplan = None

    def out_covered(self, widths: Sequence[int]) -> bool:
        return sum((b - a) * w for (a, b), w in zip(self.out_rows, widths)) == self.num_out_rows * self.out_cols

    def bias_rplan(self, device):
        """(ReducePlan over the sorted, gap-filled OUTPUT row ranges, segment index of each distinct range)."""
        if self._rplan is None or self._rplan[0].device != device:
            order = sorted(range(len(self.oranges)), key=lambda i: self.oranges[i])
            filled: List[Tuple[int, int]] = []
            seg_of = [0] * len(self.oranges)
            pos = self.oranges[order[0]][0] if order else 0
            for i in order:
                a, b = self.oranges[i]
                if a > pos:
                    filled.append((pos, a))      # gap rows: reduced but never read back
                elif a < pos and b > a:
                    raise ValueError("LinearSpec: overlapping output row ranges")
                seg_of[i] = len(filled)
                filled.append((a, b))
                pos = max(pos, b)
            self._rplan = (ReducePlan.from_ranges(filled, device, chunk=512), seg_of)
        return self._rplan


class _GroupedLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec: LinearSpec, epilogue: int, n_w: int, *params):
        N.require_cuda(x)
        weights = params[:n_w]
        biases = params[n_w:]
        x = x.contiguous()
        covered = spec.out_covered([w.shape[0] for w in weights])
        y = (torch.empty if covered else torch.zeros)((spec.num_out_rows, spec.out_cols), dtype=torch.float32, device=x.device)
        K = x.shape[1]
        x_max = row_scales_of(x)                       # fp16x3: row scales of x if its producer left them
        # (a producer's slots are addressed by 128-column tile: only column blocks that start on one can leave scales)
        # (every group a whole number of 128-column tiles and the groups tile the output: every slot of every row gets written - no fill)
        y_max = (_new_row_scale(spec.num_out_rows, N.gemm_absmax_parts(spec.out_cols), x.device, spec.out_cols,
                                zero=not (covered and all(w.shape[0] % 128 == 0 for w in weights)))
                 if all(c % 128 == 0 for c in spec.col_off) else None)
        # column statistics of y for the weight gradient that will read it as its second operand (a layer's K|Q|V gradients read the layer input)
        y_cols = (ColStats.allocate(spec.out_rows, spec.out_cols, x.device, sums=False)
                  if (covered and spec.num_out_rows > 32 and want_col_stats(spec.num_out_rows, spec.out_cols, spec.out_cols)) else None)
        groups = []
        for i, w in enumerate(weights):
            r0, r1 = spec.rows[i]
            o0, o1 = spec.out_rows[i]
            b = biases[i]
            groups.append(dict(A=N.ptr(x, r0 * K * 4), lda=K, B=N.ptr(w), ldb=w.stride(0), Bw=[w],
                               C=N.ptr(y, (o0 * spec.out_cols + spec.col_off[i]) * 4), ldc=spec.out_cols,
                               bias=N.ptr(b), M=r1 - r0, N=w.shape[0], K=K,
                               **_scale_in(x_max, r0), **_scale_out(y_max, o0, spec.col_off[i]),
                               **(y_cols.produce(o0, o1, spec.col_off[i]) if y_cols is not None else {})))
        epi = (N.WSI_EPI_BIAS if any(b is not None for b in biases) else 0) | epilogue
        wrote = _gemm(N.WSI_GEMM_NT, epi, groups, x.device)
        if y_max is not None:
            attach_row_scales(y, y_max)
        if y_cols is not None and wrote:
            attach_col_stats(y, y_cols)
        ctx.x_cols = col_stats_of(x)                   # (the weight gradient's second operand: constant features carry theirs - remember_constant_cols)
        ctx.spec, ctx.n_w = spec, n_w
        ctx.has_bias = [b is not None for b in biases]
        ctx.save_for_backward(x, *weights)
        return y

    @staticmethod
    def backward(ctx, gy):
        spec, n_w = ctx.spec, ctx.n_w
        x, *weights = ctx.saved_tensors
        gy = gy.contiguous()
        K = x.shape[1]
        dev = x.device
        gws: List[Optional[torch.Tensor]] = [None] * n_w
        gbs: List[Optional[torch.Tensor]] = [None] * n_w
        need_w = [ctx.needs_input_grad[4 + i] for i in range(n_w)]
        need_b = [ctx.has_bias[i] and ctx.needs_input_grad[4 + n_w + i] for i in range(n_w)]
        wgroups = []
        gy_cols, x_cols = col_stats_of(gy), getattr(ctx, "x_cols", None)
        for i in range(n_w):
            if not need_w[i]:
                continue
            r0, r1 = spec.rows[i]
            o0, o1 = spec.out_rows[i]
            w = weights[i]
            gws[i] = torch.empty_like(w, memory_format=torch.contiguous_format)
            if need_b[i]:      # bias gradient = column sums of dY, taken from the tiles the dW GEMM stages anyway
                gbs[i] = torch.empty(w.shape[0], dtype=torch.float32, device=dev)
                need_b[i] = False
            wgroups.append(dict(A=N.ptr(gy, (o0 * spec.out_cols + spec.col_off[i]) * 4), lda=spec.out_cols,
                                B=N.ptr(x, r0 * K * 4), ldb=K, C=N.ptr(gws[i]), ldc=K, colsum_out=N.ptr(gbs[i]),
                                M=w.shape[0], N=K, K=r1 - r0,
                                **(gy_cols.consume("a", o0, o1, spec.col_off[i], want_sums=gbs[i] is not None) if gy_cols is not None else {}),
                                **(x_cols.consume("b", r0, r1, 0) if x_cols is not None else {})))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = (torch.empty if spec.in_covered else torch.zeros)((spec.num_rows, K), dtype=torch.float32, device=dev)
            # rounds: the r-th group of every distinct input row range; round 0 overwrites, later rounds accumulate
            seen = [0] * len(spec.ranges)
            rounds: List[List[int]] = []
            for i in range(n_w):
                r = seen[spec.range_of[i]]
                seen[spec.range_of[i]] += 1
                while len(rounds) <= r:
                    rounds.append([])
                rounds[r].append(i)
            # fp16x3 row scales: dY's are usable when every group reads whole rows of it; dX's are final after ONE round only
            whole = all(spec.col_off[i] == 0 and weights[i].shape[0] == spec.out_cols for i in range(n_w))
            small = len(rounds) == 1 and spec.num_rows <= 32       # (the classifier head: one row per graph - no scales, one launch for dX and dW)
            gy_max = row_scales_of(gy) if (whole and not small) else None
            gx_max = _new_row_scale(spec.num_rows, N.gemm_absmax_parts(K), dev, K) if (len(rounds) == 1 and not small) else None
            for r, idxs in enumerate(rounds):
                groups = []
                for i in idxs:
                    r0, r1 = spec.rows[i]
                    o0 = spec.out_rows[i][0]
                    w = weights[i]
                    groups.append(dict(A=N.ptr(gy, (o0 * spec.out_cols + spec.col_off[i]) * 4), lda=spec.out_cols,
                                       B=N.ptr(w), ldb=w.stride(0), Bw=[w], C=N.ptr(gx, r0 * K * 4), ldc=K,
                                       M=r1 - r0, N=K, K=w.shape[0], **_scale_in(gy_max, o0), **_scale_out(gx_max, r0)))
                if small and wgroups and _gemm_small_pair(groups, 0, wgroups, 0, dev):
                    wgroups = []
                    continue
                _gemm(N.WSI_GEMM_NN, N.WSI_EPI_ACCUMULATE if r > 0 else 0, groups, dev)
            if gx_max is not None:
                attach_row_scales(gx, gx_max)
        if wgroups:
            _gemm(N.WSI_GEMM_TN, 0, wgroups, dev)
        if any(need_b):
            rp, seg_of = spec.bias_rplan(dev)
            colsum = _segment_reduce_raw(gy, rp, N.WSI_RED_SUM)[0]   # [n_segments, out_cols]
            for i in range(n_w):
                if need_b[i]:
                    c0 = spec.col_off[i]
                    gbs[i] = colsum[seg_of[spec.orange_of[i]], c0:c0 + weights[i].shape[0]]
        return (gx, None, None, None, *gws, *gbs)


def grouped_linear(x: torch.Tensor, spec: LinearSpec, weights: Sequence[torch.Tensor],
                   biases: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
    """y[rows_i, col_off_i : col_off_i+out_i] = x[rows_i] @ W_i^T + b_i for every group, one launch."""
    return _GroupedLinear.apply(x, spec, 0, len(weights), *weights, *biases)


_LINEAR_SPECS: dict = {}


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Linear forward/backward on the MFMA GEMM (single group)."""
    key = (x.shape[0], weight.shape[0], str(x.device))
    spec = _LINEAR_SPECS.get(key)
    if spec is None:   # cached: the spec owns device-side chunk tables (bias gradient) that must not be rebuilt per step
        if len(_LINEAR_SPECS) >= 512:                   # bounded: graphs of ever-changing sizes must not grow it forever
            _LINEAR_SPECS.pop(next(iter(_LINEAR_SPECS)))
        spec = _LINEAR_SPECS[key] = LinearSpec([(0, x.shape[0])], [0], weight.shape[0], x.shape[0])
    return _GroupedLinear.apply(x, spec, 0, 1, weight, bias)


# ------------------------------------------------------------------------------------------------
# HEAT relation attention
# ------------------------------------------------------------------------------------------------
def _attn_flags(plan: GraphPlan) -> int:
    return (N.WSI_ATTN_XCD_CONTIGUOUS if getattr(plan, "locality", False) else 0) | ((int(plan.heavy_degree) & 0xffff) << 8)



class _HeatAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kqv, e_weight, e_bias, plan: GraphPlan, sim_csr, D: int, H: int):
        N.require_cuda(kqv, sim_csr)
        lib = N.load()
        kqv = kqv.contiguous()
        n, E, S = plan.num_nodes, plan.num_edges, plan.num_segs
        ld = kqv.shape[1]
        dev = kqv.device
        t = torch.empty((n, D), dtype=torch.float32, device=dev)
        score = torch.empty((max(E, 1), H), dtype=torch.float32, device=dev)
        lse = torch.empty((max(S, 1), H), dtype=torch.float32, device=dev)
        ew = e_weight.reshape(-1)
        eb = e_bias.reshape(-1)
        with _Timed("heat_attn"):
            N.check(lib.wsi_heat_attn_fwd(
                N.ptr(kqv, D * 4), ld, N.ptr(kqv, 0), ld, N.ptr(kqv, 2 * D * 4), ld,
                n, D, H,
                N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr), N.ptr(plan.order_dst), plan.num_heavy, _attn_flags(plan),
                N.ptr(ew), N.ptr(eb),
                N.ptr(t), D, N.ptr(score), N.ptr(lse), None, N.context(), N.stream()), "wsi_heat_attn_fwd")
        ctx.plan, ctx.D, ctx.H = plan, D, H
        ctx.save_for_backward(kqv, ew, eb, sim_csr, score, lse)
        return t

    @staticmethod
    def backward(ctx, g_t):
        lib = N.load()
        plan, D, H = ctx.plan, ctx.D, ctx.H
        kqv, ew, eb, sim_csr, score, lse = ctx.saved_tensors
        g_t = g_t.contiguous()
        n, E = plan.num_nodes, plan.num_edges
        ld = kqv.shape[1]
        dev = kqv.device
        a = torch.empty_like(score)   # pass 1 reads the saved logits and writes the probabilities here
        scratch = torch.empty((3, max(E, 1), H), dtype=torch.float32, device=dev)
        red_ws = torch.empty(1024, dtype=torch.float32, device=dev)
        gkqv = torch.empty_like(kqv)
        g_e = torch.empty(2, dtype=torch.float32, device=dev)
        with _Timed("heat_attn"):
          N.check(lib.wsi_heat_attn_bwd(
            N.ptr(kqv, D * 4), ld, N.ptr(kqv, 0), ld, N.ptr(kqv, 2 * D * 4), ld,
            n, plan.num_src_rows, E, D, H,
            N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr),
            N.ptr(plan.colptr), N.ptr(plan.csc_eid), N.ptr(plan.csc_dst),
            N.ptr(plan.inv_rd), N.ptr(plan.order_dst), plan.num_heavy, N.ptr(plan.order_src), _attn_flags(plan),
            N.ptr(ew), N.ptr(eb),
            N.ptr(g_t), g_t.shape[1], None, N.ptr(score), N.ptr(a), N.ptr(lse),
            N.ptr(scratch[0]), N.ptr(scratch[1]), N.ptr(scratch[2]), N.ptr(red_ws),
            N.ptr(gkqv, D * 4), ld, N.ptr(gkqv, 0), ld, N.ptr(gkqv, 2 * D * 4), ld,
            N.ptr(g_e), None, None, N.context(), N.stream()), "wsi_heat_attn_bwd")
        return gkqv, g_e[0:1].view(1, 1), g_e[1:2], None, None, None, None


def heat_attention(kqv: torch.Tensor, e_weight: torch.Tensor, e_bias: torch.Tensor, plan: GraphPlan,
                   sim_csr: torch.Tensor, D: int, H: int) -> torch.Tensor:
    """t[N,D] from the fused K|Q|V table ``kqv`` [N,3D] (K at column 0, Q at D, V at 2D)."""
    return _HeatAttention.apply(kqv, e_weight, e_bias, plan, sim_csr, D, H)


# ------------------------------------------------------------------------------------------------
# segmented reduction
# ------------------------------------------------------------------------------------------------
class ReducePlan:
    """Chunk tables for wsi_segment_reduce_* (see include/wsi_hgnn.h)."""

    def __init__(self):
        self.device = None
        self.num_segs = 0
        self.num_chunks = 0
        self.num_rows = 0
        self.first_row = 0
        self.chunk_row = None
        self.chunk_seg = None
        self.seg_chunk = None
        self.ranges: List[Tuple[int, int]] = []      # host copy of the segments' row ranges
        self._counts = None
        self._inv_counts = None
        self._row_seg = None
        self._nonempty = None
        self._seg_of = {}
        self.type_rows = None          # optional record of the builder: the row ranges the plan was made for ...
        self.type_segments = None      # ... and the run of segments of each

    @classmethod
    def from_ptr(cls, seg_ptr: Sequence[int], device, chunk: int = 128) -> "ReducePlan":
        return cls.from_ranges([(int(seg_ptr[i]), int(seg_ptr[i + 1])) for i in range(len(seg_ptr) - 1)], device, chunk)

    @classmethod
    def from_ranges(cls, ranges: Sequence[Tuple[int, int]], device, chunk: int = 128) -> "ReducePlan":
        """``ranges`` = row range of every segment, sorted and gap-free (each start == previous end)."""
        p = cls()
        chunk_row: List[int] = []
        chunk_seg: List[int] = []
        seg_chunk = [0]
        pos = ranges[0][0] if ranges else 0
        first = pos
        for s, (a, b) in enumerate(ranges):
            if a != pos or b < a:
                raise ValueError("ReducePlan: segments must be sorted and contiguous")
            r = a
            while r < b:
                chunk_row.append(r)
                chunk_seg.append(s)
                r = min(b, r + chunk)
            pos = b
            seg_chunk.append(len(chunk_row))
        chunk_row.append(pos)
        p.device = device
        p.ranges = [(int(a), int(b)) for a, b in ranges]
        p.num_segs = len(ranges)
        p.num_chunks = len(chunk_seg)
        p.num_rows = pos - first
        p.first_row = first
        # ONE upload for everything the plan keeps on the device (a new batch of a loader-fed run builds two plans per step: seven small
        # uploads each were a quarter of a millisecond of host time): the three chunk tables, then counts / inverse counts / non-empty flags as
        # float bits.  The attributes are views into that buffer.
        cs = chunk_seg if chunk_seg else [0]
        cnt = [float(b - a) for a, b in p.ranges] or [0.0]
        inv = [1.0 / (b - a) if b > a else 0.0 for a, b in p.ranges] or [0.0]
        non = [1.0 if b > a else 0.0 for a, b in p.ranges] or [0.0]
        host = torch.cat([torch.tensor(chunk_row + cs + seg_chunk, dtype=torch.int32), torch.tensor(cnt + inv + non, dtype=torch.float32).view(torch.int32)])
        buf = host_to_device(host, torch.int32, device)
        o1 = len(chunk_row); o2 = o1 + len(cs); o3 = o2 + len(seg_chunk); k = len(cnt)
        p.chunk_row, p.chunk_seg, p.seg_chunk = buf[:o1], buf[o1:o2], buf[o2:o3]
        p._counts = buf[o3:o3 + k].view(torch.float32).view(-1, 1)
        p._inv_counts = buf[o3 + k:o3 + 2 * k].view(torch.float32).view(-1, 1)
        p._nonempty = buf[o3 + 2 * k:o3 + 3 * k].view(torch.float32).view(-1, 1)
        return p


    def counts(self) -> torch.Tensor:
        """[num_segs, 1] fp32 row counts of the segments (device)."""
        if self._counts is None:
            self._counts = host_to_device([float(b - a) for a, b in self.ranges] or [0.0], torch.float32, self.device).view(-1, 1)
        return self._counts

    def inv_counts(self) -> torch.Tensor:
        """[num_segs, 1]: 1 / rows of the segment, 0 for an empty one."""
        if self._inv_counts is None:
            self._inv_counts = host_to_device([1.0 / (b - a) if b > a else 0.0 for a, b in self.ranges] or [0.0], torch.float32, self.device).view(-1, 1)
        return self._inv_counts

    def prepare_broadcast(self) -> None:
        """Upload the small per-segment tables the low-rank readout gradient path uses (``SegmentBroadcast``), from the thread and
        at the time the plan is first used - not from inside the backward pass of the first step that meets a new batch."""
        self.counts(); self.inv_counts(); self.nonempty(); self.row_segment()

    def has_empty(self) -> bool:
        return any(b == a for a, b in self.ranges)

    def nonempty(self) -> torch.Tensor:
        """[num_segs, 1]: 1.0 for a segment with rows, 0.0 for an empty one."""
        if self._nonempty is None:
            self._nonempty = host_to_device([1.0 if b > a else 0.0 for a, b in self.ranges] or [0.0], torch.float32, self.device).view(-1, 1)
        return self._nonempty

    def row_segment(self) -> torch.Tensor:
        """[num_rows] int32: the segment of every row (device)."""
        if self._row_seg is None:
            # expanded ON THE DEVICE from the S counts (output size known: no sync).  A [num_rows] table built on the host costs a
            # multi-threaded CPU copy into the staging buffer per new batch - measured 70-100 ms stalls of the whole process on a
            # CPU-quota'd box (the OpenMP workers spin after the copy and the cgroup gets throttled).
            dev = torch.device(self.device)
            if dev.type == "cuda":
                # one upload + one launch: a run of `rows` copies of the segment number per segment (wsi_plan_assemble, mode 0 with add = s)
                out = torch.empty(max(self.num_rows, 1), dtype=torch.int32, device=dev)[:self.num_rows]
                desc, blocks = [], 0
                for s_, (a, b) in enumerate(self.ranges):
                    if b > a:
                        desc += [out.data_ptr() + 4 * (a - self.first_row), 0, 0, -1, 0, s_, 0, b - a, 0, blocks]
                        blocks += (b - a + 1023) // 1024
                if desc:
                    table = host_to_device(desc, torch.int64, dev)
                    N.check(N.load().wsi_plan_assemble(N.ptr(table), len(desc) // 10, blocks, N.stream()), "wsi_plan_assemble")
                    self._row_seg_desc = table
                self._row_seg = out
            else:
                reps = torch.tensor([b - a for a, b in self.ranges], dtype=torch.int64)
                self._row_seg = torch.repeat_interleave(torch.arange(len(self.ranges), dtype=torch.int32), reps)
        return self._row_seg

    def segments_of(self, rows: Sequence[Tuple[int, int]]) -> Optional[List[Tuple[int, int]]]:
        """For row ranges that are unions of consecutive segments: the segment index range of each; None if one is not.  A plan built for
        known row ranges (``type_rows`` / ``type_segments``, set by its builder: the readout plan's B segments per node type) answers from
        that record - an EMPTY segment on a boundary between two ranges belongs to exactly one of them, which row numbers alone cannot
        tell.  Otherwise sorted, gap-free ranges are matched by a walk in which such a segment goes with the range on its LEFT."""
        key = tuple(rows)
        hit = self._seg_of.get(key)
        if hit is None:
            res: Optional[List[Tuple[int, int]]]
            if self.type_rows is not None and list(self.type_rows) == [tuple(r) for r in rows]:
                res = list(self.type_segments)
            else:
                res, cur = [], 0
                for a, b in rows:
                    if a == b:
                        res.append((0, 0))
                        continue
                    s0 = cur
                    if s0 >= len(self.ranges) or self.ranges[s0][0] != a:
                        res = None
                        break
                    s1 = s0
                    while s1 < len(self.ranges) and self.ranges[s1][1] <= b and self.ranges[s1][0] >= a:
                        s1 += 1
                    if s1 == s0 or self.ranges[s1 - 1][1] != b:
                        res = None
                        break
                    res.append((s0, s1))
                    cur = s1
            hit = self._seg_of[key] = (res,)
        return hit[0]


def _segment_reduce_raw(x: torch.Tensor, rp: ReducePlan, op: int):
    lib = N.load()
    D = x.shape[1]
    dev = x.device
    out = torch.empty((rp.num_segs, D), dtype=torch.float32, device=dev)
    extra = 2 if op == N.WSI_RED_MAX else 1
    partial = torch.empty(max(rp.num_chunks * D * extra, 1), dtype=torch.float32, device=dev)
    argmax = torch.empty((rp.num_segs, D), dtype=torch.int32, device=dev) if op == N.WSI_RED_MAX else None
    N.check(lib.wsi_segment_reduce_fwd(N.ptr(x), x.stride(0), D, op, N.ptr(rp.chunk_row), rp.num_chunks,
                                       N.ptr(rp.seg_chunk), rp.num_segs, N.ptr(partial), N.ptr(out), D,
                                       N.ptr(argmax), N.stream()), "wsi_segment_reduce_fwd")
    return out, argmax


class SegmentBroadcast:
    """What the backward of a sum / mean readout knows about the gradient it hands down: ``gx[row] = g_row[segment of row]`` -
    a matrix of rank <= num_segs (graphs x node types), not of rank num_rows.  The layer that receives ``gx`` unchanged
    (``_HeatLayerFused.backward``) computes its output-projection gradients from the [num_segs, D] factors instead of running
    [num_rows]-deep GEMMs over identical rows.  ``x_ptr`` / ``x_version`` identify the tensor the readout reduced and ``x_mean`` its segment means."""

    def __init__(self, g_seg, rp, op, pooled, x_ptr, x_version):
        if op == N.WSI_RED_MEAN:
            self.g_row = g_seg * rp.inv_counts()        # gradient of every row of the segment
            self.g_sum = g_seg * rp.nonempty() if rp.has_empty() else g_seg      # count x g_row: the segment's rows summed
            self.x_mean = pooled
        else:
            self.g_row = g_seg
            self.g_sum = g_seg * rp.counts()
            self.x_mean = pooled * rp.inv_counts()
        self.rp, self.x_ptr, self.x_version = rp, x_ptr, x_version

    @classmethod
    def from_parts(cls, rp, g_row: torch.Tensor, g_sum: torch.Tensor, x_mean: torch.Tensor) -> "SegmentBroadcast":
        """The factors as ``wsi_pool_bwd_prep`` leaves them (same meaning as ``from_pooled_gradient`` computes with tensor operations)."""
        self = cls.__new__(cls)
        self.rp, self.x_ptr, self.x_version, self.x_mean, self.g_row, self.g_sum = rp, None, None, x_mean, g_row, g_sum
        return self

    @classmethod
    def from_pooled_gradient(cls, g_pool: torch.Tensor, rp, op: int, x_mean: torch.Tensor) -> "SegmentBroadcast":
        """The same factors for a layer that returned its readout itself (``_HeatLayerFused(pool=)``): ``g_pool`` is the gradient of the pooled
        rows [num_segs, D] (zero weight on empty segments: the forward masked them), ``x_mean`` the segment means of the never-formed output."""
        self = cls.__new__(cls)
        self.rp, self.x_ptr, self.x_version, self.x_mean = rp, None, None, x_mean
        if op == N.WSI_RED_MEAN:
            self.g_row = g_pool * rp.inv_counts()
            self.g_sum = g_pool * rp.nonempty() if rp.has_empty() else g_pool
        else:
            self.g_row, self.g_sum = g_pool, g_pool * rp.counts()
        return self


_LOW_RANK = {"enabled": True}


# Column block of K / Q / V in a fused layer's [n, 3D] table.  "kqv": K | Q | V (the default).  "kvq": K | V | Q - the two rows an edge GATHERS by its
# source (k for the logit, v for the message) and the two gradient rows pass 3 writes per source node (g_k, g_v) are ONE contiguous 2 D run
# (4 KB at D = 512) instead of two runs 2 KB apart.  The kernels take three pointers: the layout is this module's choice.  Measured in round 5
# (tools/layout_probe.py, both layers at full depth, interleaved): 6.731 vs 6.718 ms per step - no difference (a gather moves whole 128-byte lines
# either way and the fabric does not care whether a node's two 2 KB runs touch): the default stays, the switch stays for the record.
_KQV_LAYOUT = {"order": "kqv"}


def set_kqv_layout(order: str) -> None:
    if order not in ("kqv", "kvq"):
        raise ValueError("kqv layout: 'kqv' or 'kvq'")
    _KQV_LAYOUT["order"] = order


def _kqv_blocks(nproj: int):
    """(column block of K, of Q, of V) in a table of ``nproj`` blocks (2: K | Q - the last layer under a readout never forms V)."""
    return (0, 2, 1) if (nproj == 3 and _KQV_LAYOUT["order"] == "kvq") else (0, 1, 2)


_COLLAPSE_V = {"enabled": True, "min_work": 4.0e9}


def set_value_collapse(on: bool, min_work: Optional[float] = None) -> None:
    """On (default): a readout-fused last layer never forms V or g_v (rank <= segments x heads): ``wsi_attn_pool_t``.  Off: V goes through
    the K|Q|V projections like K and Q (A/B measurements, parity tests of both forms).  ``min_work``: rows x D x D below which the layer
    keeps V anyway - the collapse trades three [rows, D] x [D, D] projections for ~25 small launches and a heavier pass 1 (measured on one
    MI355X: 80 000 x 512^2 = 2.1e10: 7.38 -> 7.02 ms per step; 40 000 x 256^2 = 2.6e9: 2.24 -> 2.57 ms; default threshold 4e9)."""
    _COLLAPSE_V["enabled"] = bool(on)
    if min_work is not None:
        _COLLAPSE_V["min_work"] = float(min_work)


def set_low_rank_readout_grad(on: bool) -> None:
    """On (default): the layer under a sum / mean readout uses the rank-(graphs x node types) structure of the gradient it
    receives (see ``SegmentBroadcast``).  Off: every gradient goes through the full-depth GEMMs (A/B measurements, parity tests)."""
    _LOW_RANK["enabled"] = bool(on)


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rp: ReducePlan, op: int):
        N.require_cuda(x)
        x = x.contiguous()
        out, argmax = _segment_reduce_raw(x, rp, op)
        ctx.rp, ctx.op, ctx.shape = rp, op, x.shape
        ctx.x_id = (x.data_ptr(), x._version)
        ctx.save_for_backward(argmax) if argmax is not None else ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = N.load()
        rp, op = ctx.rp, ctx.op
        gout = gout.contiguous()
        n, D = ctx.shape
        argmax = ctx.saved_tensors[0] if op == N.WSI_RED_MAX else None
        # rows not covered by any segment (none in practice) and max-pool non-argmax rows get zero
        covered = rp.num_rows == n and op != N.WSI_RED_MAX
        gx = (torch.empty if covered else torch.zeros)((n, D), dtype=torch.float32, device=gout.device)
        N.check(lib.wsi_segment_reduce_bwd(N.ptr(gout), gout.stride(0), D, op, N.ptr(rp.chunk_row), N.ptr(rp.chunk_seg),
                                           rp.num_chunks, N.ptr(rp.seg_chunk), rp.num_segs, N.ptr(argmax),
                                           N.ptr(gx), D, N.stream()), "wsi_segment_reduce_bwd")
        if covered and _LOW_RANK["enabled"] and rp.first_row == 0 and rp.num_segs * 8 <= n:
            _annotate(gx, "_wsi_broadcast", SegmentBroadcast(gout, rp, op, ctx.saved_tensors[0], *ctx.x_id))
        return gx, None, None


def segment_reduce(x: torch.Tensor, rp: ReducePlan, op: str) -> torch.Tensor:
    """[num_rows, D] -> [num_segs, D]; op in {'sum','mean','max'}; empty segments give 0."""
    code = {"sum": N.WSI_RED_SUM, "mean": N.WSI_RED_MEAN, "max": N.WSI_RED_MAX}[op]
    return _SegmentReduce.apply(x, rp, code)


def segment_weighted_sums(x: torch.Tensor, w: torch.Tensor, rp: "ReducePlan") -> torch.Tensor:
    """out[s, j, :] = sum over the rows r of segment s of w[r, j] * x[r, :];  x [rows, D], w [rows, J] -> [num_segs, J, D]."""
    N.require_cuda(x, w)
    x, w = x.contiguous(), w.contiguous()
    D, J = x.shape[1], w.shape[1]
    out = torch.empty((rp.num_segs, J, D), dtype=torch.float32, device=x.device)
    partial = torch.empty(max(rp.num_chunks * J * D, 1), dtype=torch.float32, device=x.device)
    N.check(N.load().wsi_segment_weighted_sums(N.ptr(x), D, D, N.ptr(w), J, J, N.ptr(rp.chunk_row), rp.num_chunks, N.ptr(rp.seg_chunk),
                                               rp.num_segs, N.ptr(partial), N.ptr(out), N.stream()), "wsi_segment_weighted_sums")
    return out


# ------------------------------------------------------------------------------------------------
# segment dot (skip-gate gradient)
# ------------------------------------------------------------------------------------------------
def segment_dot_diff(g: torch.Tensor, a: torch.Tensor, b: torch.Tensor, rp: "ReducePlan") -> torch.Tensor:
    """out[s] = sum over rows of segment s and all columns of g * (a - b)."""
    lib = N.load()
    D = g.shape[1]
    out = torch.empty(rp.num_segs, dtype=torch.float32, device=g.device)
    partial = torch.empty(max(rp.num_chunks * ((D + 255) // 256), 1), dtype=torch.float32, device=g.device)
    N.check(lib.wsi_segment_dot_diff(N.ptr(g), g.stride(0), N.ptr(a), a.stride(0), N.ptr(b), b.stride(0), D,
                                     N.ptr(rp.chunk_row), rp.num_chunks, N.ptr(rp.seg_chunk), rp.num_segs,
                                     N.ptr(partial), N.ptr(out), N.stream()), "wsi_segment_dot_diff")
    return out


# ------------------------------------------------------------------------------------------------
# counter-based dropout (WSI_EPI_DROPOUT / wsi_dropout_apply; the hash is specified in include/wsi_hgnn.h)
# ------------------------------------------------------------------------------------------------
class CounterDropout:
    """One draw of ``nn.Dropout(p)`` over a [rows, cols] tensor as a FUNCTION of (seed, row, col) instead of a mask tensor: the projection's
    epilogue applies it while it writes the tensor, the backward regenerates it (``dropout_apply``) - nothing is generated, stored or read back
    (models/HEATNet4.py:135 ``self.drop(self.a_linears[...](t))``; two passes over [N, D] for torch's mask, one read in the epilogue and one in the
    backward go away).  ``p`` is quantised to 1/65536; kept values are scaled by 1 / (1 - p) exactly as nn.Dropout scales them."""

    def __init__(self, p: float, seed: int, seed_base: Optional[torch.Tensor] = None):
        """``seed_base``: an optional one-element int32 DEVICE tensor whose value is added to ``seed`` (mod 2^32) when the kernels run - see
        ``dropout_seed_base``: what lets a step captured into a hipGraph draw new masks at every replay."""
        if not 0.0 <= p < 1.0:
            raise ValueError("dropout probability must be in [0, 1)")
        self.p = float(p)
        self.seed = int(seed) & 0xffffffff
        self.seed_base = seed_base
        self.threshold = int(round(self.p * 65536.0))
        self.scale = 1.0 / (1.0 - self.p)

    def effective_seed(self) -> int:
        """seed + the CURRENT value of the device word (a synchronising read: tests only)."""
        return (self.seed + (int(self.seed_base.item()) if self.seed_base is not None else 0)) & 0xffffffff

    def group_fields(self, row0: int, cols: int, col0: int = 0) -> dict:
        return dict(drop_seed=self.seed, drop_threshold=self.threshold, drop_scale=self.scale, drop_row0=int(row0), drop_cols=int(cols), drop_col0=int(col0),
                    drop_seed_base=N.ptr(self.seed_base))


def next_dropout_seed() -> int:
    """A fresh 32-bit seed per dropout draw, taken from torch's default CPU generator (a host-side draw: no device round trip): the sequence of
    masks follows ``torch.manual_seed`` exactly as nn.Dropout's does - re-seeding replays it."""
    return int(torch.empty((), dtype=torch.int64).random_(0, 1 << 32).item())


_SEED_BASE: dict = {}           # device index -> the one-element int32 tensor every CounterDropout created meanwhile adds to its seed
SEED_STRIDE = -1640531527       # 0x9E3779B9 as int32: what one step advances the word by


@contextlib.contextmanager
def dropout_seed_base(base: Optional[torch.Tensor]):
    """While active, every dropout draw on ``base``'s device is ``CounterDropout(p, host seed, seed_base=base)``: its mask is a function of
    host seed + the value ``base`` holds WHEN THE KERNEL RUNS.  ``trainer.CapturedStep`` records a step under it and lets the recorded step advance
    ``base`` (``advance_dropout_seed_base``): the host seeds are frozen into the graph, the word is not - every replay draws new masks, forward and
    backward of one replay the same ones."""
    if base is None:
        yield
        return
    if not (base.is_cuda and base.dtype == torch.int32 and base.numel() == 1):
        raise ValueError("dropout_seed_base: a one-element int32 CUDA tensor")
    key = base.device.index
    prev = _SEED_BASE.get(key)
    _SEED_BASE[key] = base
    try:
        yield
    finally:
        if prev is None:
            _SEED_BASE.pop(key, None)
        else:
            _SEED_BASE[key] = prev


def current_dropout_seed_base(device) -> Optional[torch.Tensor]:
    device = torch.device(device)
    return _SEED_BASE.get(device.index if device.index is not None else torch.cuda.current_device()) if device.type == "cuda" else None


def advance_dropout_seed_base(base: torch.Tensor) -> None:
    """One step further (an in-place device add: part of the recorded step)."""
    base.add_(SEED_STRIDE)


def _mul32(a: torch.Tensor, b: int) -> torch.Tensor:
    """(a * b) mod 2^32 for int64 tensors holding 32-bit values, without overflowing int64."""
    lo = (a & 0xffff) * b
    hi = (((a >> 16) * b) & 0xffff) << 16
    return (lo + hi) & 0xffffffff


def dropout_keep_mask(drop: CounterDropout, rows: int, cols: int, device="cpu", row0: int = 0) -> torch.Tensor:
    """The boolean keep mask of ``drop`` for rows [row0, row0 + rows) of a tensor with ``cols`` columns, replayed with integer tensor arithmetic from
    the specification in include/wsi_hgnn.h (independent of the kernels: the tests compare them with it)."""
    r = torch.arange(row0, row0 + rows, dtype=torch.int64, device=device).view(-1, 1)
    c = torch.arange(cols, dtype=torch.int64, device=device).view(1, -1)
    pairs = (cols + 1) // 2
    idx = (_mul32(r & 0xffffffff, pairs) + (c >> 1)) & 0xffffffff
    h = (_mul32(idx, 0x9E3779B1) + drop.effective_seed()) & 0xffffffff
    h = h ^ (h >> 16)
    h = _mul32(h, 0x85EBCA6B)
    h = h ^ (h >> 13)
    h = _mul32(h, 0xC2B2AE35)
    h = h ^ (h >> 16)
    bits = torch.where((c & 1) == 1, h >> 16, h & 0xffff)
    return bits >= drop.threshold


def dropout_apply(x: torch.Tensor, drop: CounterDropout, row0: int = 0) -> torch.Tensor:
    """x * mask * 1/(1-p) with the regenerated mask of ``drop`` (``wsi_dropout_apply``); x: [rows, cols] whose row 0 is row ``row0`` of the masked tensor."""
    N.require_cuda(x)
    x = x.contiguous()
    out = torch.empty_like(x)
    N.check(N.load().wsi_dropout_apply(N.ptr(x), x.stride(0), N.ptr(out), out.stride(0), x.shape[0], x.shape[1], int(row0), x.shape[1], 0,
                                       drop.seed, N.ptr(drop.seed_base), drop.threshold, drop.scale, N.stream()), "wsi_dropout_apply")
    return out


# ------------------------------------------------------------------------------------------------
# fused HEAT layer: K|Q|V GEMM -> relation attention -> output GEMM with the sigmoid-gated skip in its
# epilogue; hand-written backward so no elementwise pass, gather or gradient accumulation is left to
# eager PyTorch (models/HEATNet4.py:85-138 as ONE autograd node).
# ------------------------------------------------------------------------------------------------
def _value_collapse_applies(hctx, prp, n: int, D: int, H: int) -> bool:
    """The last layer's V never needs forming (wsi_attn_pool_t): fast attention kernels, <= 8 node types, HEAT-style source rows, and the
    readout plan numbered type-major (segment = type * graphs + graph) over exactly the layer's node-type row ranges."""
    T = len(hctx.rows)
    bseg = prp.num_segs // max(T, 1)
    return (_COLLAPSE_V["enabled"] and float(n) * D * D >= _COLLAPSE_V["min_work"]
            and D in (128, 256, 512) and H in (1, 2, 4, 8, 16) and 1 <= T <= 8 and hctx.plan.num_src_rows == n
            and prp.num_rows == n and prp.num_segs == T * bseg
            and prp.segments_of(hctx.rows) == [(i * bseg, (i + 1) * bseg) for i in range(T)])


def _edge_segments(plan) -> torch.Tensor:
    """[E] int32: the softmax segment (row of ``lse``) of every CSR edge, expanded on the device from ``rowptr`` once per plan."""
    es = plan.__dict__.get("_edge_seg")
    if es is None:
        counts = (plan.rowptr[1:] - plan.rowptr[:-1]).to(torch.int64)
        es = torch.repeat_interleave(torch.arange(plan.num_segs, dtype=torch.int32, device=counts.device), counts, output_size=plan.num_edges)
        plan.__dict__["_edge_seg"] = es
    return es


def _segment_dst(plan) -> torch.Tensor:
    """[num softmax segments] int32: the destination node of every softmax segment, expanded on the device from ``node_seg`` once per plan."""
    sd = plan.__dict__.get("_seg_dst")
    if sd is None:
        counts = (plan.node_seg[1:] - plan.node_seg[:-1]).to(torch.int64)
        sd = torch.repeat_interleave(torch.arange(counts.shape[0], dtype=torch.int32, device=counts.device), counts, output_size=plan.num_segs)
        plan.__dict__["_seg_dst"] = sd
    return sd


def _pooled_factors(h, ctab, prp, T: int, H: int, with_mean: bool = False):
    """hp[seg, h, tau, :] = sum over the source rows u of type tau in seg's graph of ctab[u, type(seg), h] * h[u, :]  and  csum[tau, seg, h] = the same
    sum of the coefficients alone - one weighted-sums pass over the (source type, graph) segments whose second stage writes both where their
    consumers read them (``wsi_pool_factors``: two launches)."""
    n, D = h.shape
    S, J = prp.num_segs, T * H
    hp = torch.empty((S, H, T, D), dtype=torch.float32, device=h.device)
    csum = torch.empty((T, S, H), dtype=torch.float32, device=h.device)
    partial = torch.empty(max(prp.num_chunks * (J + 1) * (D + 1), 1), dtype=torch.float32, device=h.device)
    h_mean = torch.empty((S, D), dtype=torch.float32, device=h.device) if with_mean else None      # (the same pass: one more weight column of ones)
    N.check(N.load().wsi_pool_factors(N.ptr(h), D, D, N.ptr(ctab), J, T, H, S // T, N.ptr(prp.chunk_row), prp.num_chunks, N.ptr(prp.seg_chunk),
                                      N.ptr(partial), N.ptr(hp), N.ptr(csum), N.ptr(h_mean), N.stream()), "wsi_pool_factors")
    return (hp, csum, h_mean) if with_mean else (hp, csum)


def _ptr_array(tensors):
    """HOST array of device pointers (``const float* const*`` arguments of the wsi_pool_* entry points)."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _gate_tables(hctx, skip, segs, num_segs: int):
    """(seg_gate [num_segs] int32: the model gate (index into ``skip``) of the node type a readout segment belongs to, -1 for a type the layer
    passes through;  type_gate [T] int32: the same per graph node type) - device tables, cached on the layer context per readout plan."""
    from .graph import host_to_device
    key = ("gate_tables", tuple(segs), num_segs)
    hit = hctx.cache.get(key)
    if hit is None:
        T = len(hctx.rows)
        tg = [hctx.nid[i] if (i in hctx.a_types) else -1 for i in range(T)]
        sg = [-1] * num_segs
        for i, (s0, s1) in enumerate(segs):
            for s_ in range(s0, s1):
                sg[s_] = tg[i]
        hit = hctx.cache[key] = (host_to_device(sg or [-1], torch.int32, skip.device), host_to_device(tg, torch.int32, skip.device))
    return hit


class _HeatLayerFused(torch.autograd.Function):
    """inputs: h [N,D], hctx (HeatContext), H, skip [T_model], e_weight [1,1], e_bias [1], drop_mask ([N,D] keep mask
    scaled by 1/(1-p) for the nn.Dropout of HEATNet4.py:135, or None), then per graph node type i (in hctx order) 8 tensors:
    Wk, Wq, Wv, Wa, bk, bq, bv, ba.

    ``pool`` = (ReducePlan, WSI_RED_SUM | WSI_RED_MEAN) or None.  With a pool the function returns the READOUT of the layer's output,
    [num_segs, D], and never forms the output itself: a sum / mean over node rows commutes with the affine output stage,
        mean_seg( s (t Wa^T + ba) + (1 - s) h ) = s (mean_seg(t) Wa^T + ba) + (1 - s) mean_seg(h)
    (models/HEATNet4.py:128-135 followed by pools[0] at :219, with no dropout between them) - the [N, D] x [D, D] projection becomes
    two segment means and an [S, D] x [D, D] one, and the backward starts from the S gradient rows it would otherwise meet broadcast
    to N (``SegmentBroadcast``).  Every segment of the plan must lie inside one node type's row range."""

    @staticmethod
    def forward(ctx, h, hctx, H, skip, e_weight, e_bias, drop_mask, pool, background_dw, *params):
        N.require_cuda(h)
        _background_recover()                  # (leftovers of a backward pass that raised: nothing in the normal case)
        ctx.background_dw = bool(background_dw)
        lib = N.load()
        h = h.contiguous()
        dev = h.device
        T = len(hctx.rows)
        n, D = h.shape
        plan = hctx.plan
        P = [params[8 * i:8 * i + 8] for i in range(T)]
        # fp16x3 row scales (absmax bits) travel with the activations: h's from its producer, t's from the attention kernel,
        # out's from the epilogue that writes it - no projection makes its own pass over an operand the path just produced
        h_max = row_scales_of(h)
        everywhere = all(hctx.incoming)              # every node type gets the out projection: its epilogue writes all slots of all rows
        t_max = _new_row_scale(n, 1, dev, D, zero=False)                          # the attention kernel writes every row
        out_max = None if pool is not None else _new_row_scale(n, N.gemm_absmax_parts(D), dev, D, zero=not everywhere)
        # a readout-fused last layer never needs V (its aggregate is read through S x H weighted sums of h): K and Q only
        counter = drop_mask if isinstance(drop_mask, CounterDropout) else None       # the draw as a function (no tensor) ...
        if counter is not None and counter.threshold == 0:
            counter = drop_mask = None                                                  # p quantises to 0: nothing is dropped
        no_v = pool is not None and drop_mask is None and _value_collapse_applies(hctx, pool[0], n, D, H)
        nproj = 2 if no_v else 3
        ldp = nproj * D
        # column statistics for the weight gradients (scaled-fp16 TN: ColStats).  h's come with it; a row of t is a convex combination of V rows
        # (softmax weights within a relation slot, then a mean over the slots: |t[w, c]| <= max_u |v[u, c]| over the source nodes u), so V's column
        # maxima bound t's - a few binades loose at most, which the 17-binade full-precision window of the split absorbs: the V groups of the
        # projection leave them; the output projection leaves those of `out` for the next layer
        full = pool is None
        cs_on = full and want_col_stats(n, D, D)
        ctx.h_cols = col_stats_of(h)
        v_cols = ColStats.allocate(hctx.rows, D, dev, sums=False) if cs_on else None
        # 1) K|Q(|V) table
        kqv = torch.empty((n, ldp), dtype=torch.float32, device=dev)
        blk = _kqv_blocks(nproj) if pool is None else (0, 1, 2)      # column block of K, Q, V (under a readout the backward may walk K | Q as columns [0, 2D))
        ctx.blk = blk
        kO, qO, vO = blk[0] * D * 4, blk[1] * D * 4, blk[2] * D * 4
        groups = []
        for i, (r0, r1) in enumerate(hctx.rows):
            for j in range(nproj):
                groups.append(dict(A=N.ptr(h, r0 * D * 4), lda=D, B=N.ptr(P[i][j]), ldb=D, Bw=[P[i][j]],
                                   C=N.ptr(kqv, (r0 * ldp + blk[j] * D) * 4), ldc=ldp, bias=N.ptr(P[i][4 + j]),
                                   M=r1 - r0, N=D, K=D, **_scale_in(h_max, r0),
                                   **(v_cols.produce(r0, r1, 0) if (v_cols is not None and j == 2) else {})))
        if not _gemm(N.WSI_GEMM_NT, N.WSI_EPI_BIAS, groups, dev):
            v_cols = None
        # a row of t mixes V rows of EVERY source type that reaches its node: the bound of each of t's row ranges is the maximum over ALL parts
        ctx.t_cols = ColStats(v_cols.bits, None, {r: (0, v_cols.bits.shape[0]) for r in hctx.rows if r[1] > r[0]}) if v_cols is not None else None
        # 2) relation attention
        score = torch.empty((max(plan.num_edges, 1), H), dtype=torch.float32, device=dev)
        lse = torch.empty((max(plan.num_segs, 1), H), dtype=torch.float32, device=dev)
        ew, eb = e_weight.reshape(-1), e_bias.reshape(-1)
        sim_csr = hctx.sim_csr                  # fetched once per forward; backward uses the same tensor
        ctx.no_v = no_v
        if no_v:
            prp = pool[0]
            S, dk = prp.num_segs, D // H
            with _Timed("heat_attn"), _Timed("heat_attn_fwd_pooled"):
                N.check(lib.wsi_heat_attn_scores_fwd(
                    N.ptr(kqv, qO), ldp, N.ptr(kqv, kO), ldp, n, D, H,
                    N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr), N.ptr(plan.order_dst), plan.num_heavy, _attn_flags(plan),
                    N.ptr(ew), N.ptr(eb), N.ptr(score), N.ptr(lse), N.context(), N.stream()), "wsi_heat_attn_scores_fwd")
                ctab = torch.empty((n, T, H), dtype=torch.float32, device=dev)
                N.check(lib.wsi_heat_pool_coeff(N.ptr(score), N.ptr(lse), N.ptr(_edge_segments(plan)), N.ptr(plan.colptr), N.ptr(plan.csc_eid),
                                                N.ptr(plan.csc_dst), N.ptr(plan.inv_rd), N.ptr(prp.row_segment()), S // T, T, H, n,
                                                N.ptr(ctab), N.stream()), "wsi_heat_pool_coeff")
            # sum_seg(t)[:, head h] = sum_tau ( hp[:, h, tau, :] (W_v^tau rows of head h)^T + csum[tau, :, h] b_v^tau (head h) )
            hp, csum, h_mean_pre = _pooled_factors(h, ctab, prp, T, H, with_mean=True)
            # one launch per stage: the per-(head, source type) products with W_v into tpart[tau], then their sum over tau + the value-bias
            # term + the 1 / count scaling in wsi_pool_tmean
            tpart = torch.empty((T, S, D), dtype=torch.float32, device=dev)
            groups = [dict(A=N.ptr(hp, (hh * T + tau) * D * 4), lda=H * T * D, B=N.ptr(P[tau][2], hh * dk * D * 4), ldb=D,
                           C=N.ptr(tpart, (tau * S * D + hh * dk) * 4), ldc=D, M=S, N=dk, K=D) for tau in range(T) for hh in range(H)]
            _gemm(N.WSI_GEMM_NT, 0, groups, dev)
            t = None
            t_mean_pre = torch.empty((S, D), dtype=torch.float32, device=dev)
            N.check(lib.wsi_pool_tmean(N.ptr(tpart), T, S, D, H, N.ptr(csum), _ptr_array([P[tau][6] for tau in range(T)]),
                                       N.ptr(prp.inv_counts()), N.ptr(t_mean_pre), N.stream()), "wsi_pool_tmean")
        else:
            t = torch.empty((n, D), dtype=torch.float32, device=dev)
            with _Timed("heat_attn"), _Timed("heat_attn_fwd_full"):
                N.check(lib.wsi_heat_attn_fwd(
                    N.ptr(kqv, qO), 3 * D, N.ptr(kqv, kO), 3 * D, N.ptr(kqv, vO), 3 * D, n, D, H,
                    N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr), N.ptr(plan.order_dst), plan.num_heavy, _attn_flags(plan),
                    N.ptr(ew), N.ptr(eb), N.ptr(t), D, N.ptr(score), N.ptr(lse), N.ptr(t_max), N.context(), N.stream()), "wsi_heat_attn_fwd")
        ctx.hctx, ctx.H, ctx.T = hctx, H, T
        ctx.has_mask = drop_mask is not None
        ctx.pool = pool
        if pool is not None:
            # 3') the readout of the output, straight from the segment means of t and h
            prp, pop = pool
            segs = prp.segments_of(hctx.rows)
            if drop_mask is not None or segs is None or prp.num_rows != n or pop not in (N.WSI_RED_SUM, N.WSI_RED_MEAN):
                raise ValueError("heat_layer_fused(pool=...): needs a sum / mean plan over all rows whose segments respect the node types, and no dropout mask")
            t_mean = t_mean_pre if no_v else _segment_reduce_raw(t, prp, N.WSI_RED_MEAN)[0]
            h_mean = h_mean_pre if no_v else _segment_reduce_raw(h, prp, N.WSI_RED_MEAN)[0]     # (no_v: taken by the weighted-sums pass over h)
            z_mean = torch.empty_like(h_mean) if everywhere else h_mean.clone()       # passthrough types (:129-133): mean_seg(h)
            groups = []
            for i in hctx.a_types:
                s0, s1 = segs[i]
                groups.append(dict(A=N.ptr(t_mean, s0 * D * 4), lda=D, B=N.ptr(P[i][3]), ldb=D, C=N.ptr(z_mean, s0 * D * 4), ldc=D,
                                   bias=N.ptr(P[i][7]), R=N.ptr(h_mean, s0 * D * 4), ldr=D, gate=N.ptr(skip, 4 * hctx.nid[i]),
                                   M=s1 - s0, N=D, K=D))
            _gemm(N.WSI_GEMM_NT, N.WSI_EPI_GATED_SKIP, groups, dev)
            # an empty segment reads 0 (not s * ba): the readout kernels' convention, dgl.readout semantics
            pooled = z_mean * prp.counts() if pop == N.WSI_RED_SUM else (z_mean * prp.nonempty() if prp.has_empty() else z_mean)
            extra = (ctab, hp, csum) if no_v else ()
            ctx.save_for_backward(h, kqv, t_mean, h_mean, z_mean, score, lse, skip, ew, eb, sim_csr, *extra, *params)
            return pooled
        # 3) out = sigma(skip) * (t Wa^T + ba) + (1 - sigma(skip)) * h      (HEATNet4.py:128-135)
        out = torch.empty((n, D), dtype=torch.float32, device=dev)
        out_cols = ColStats.allocate([hctx.rows[i] for i in hctx.a_types], D, dev, sums=False) if cs_on else None
        groups = []
        for i in hctx.a_types:
            r0, r1 = hctx.rows[i]
            groups.append(dict(A=N.ptr(t, r0 * D * 4), lda=D, B=N.ptr(P[i][3]), ldb=D, Bw=[P[i][3]], C=N.ptr(out, r0 * D * 4), ldc=D,
                               bias=N.ptr(P[i][7]), R=N.ptr(h, r0 * D * 4), ldr=D, gate=N.ptr(skip, 4 * hctx.nid[i]),
                               Mm=N.ptr(drop_mask, r0 * D * 4) if (drop_mask is not None and counter is None) else None, ldm=D,
                               M=r1 - r0, N=D, K=D, **_scale_in(t_max, r0), **_scale_out(out_max, r0),
                               **(counter.group_fields(r0, D) if counter is not None else {}),
                               **(out_cols.produce(r0, r1, 0) if out_cols is not None else {})))
        if not _gemm(N.WSI_GEMM_NT, N.WSI_EPI_GATED_SKIP | (N.WSI_EPI_DROPOUT if counter is not None else (N.WSI_EPI_MUL_M if drop_mask is not None else 0)), groups, dev):
            out_cols = None
        for i, (r0, r1) in enumerate(hctx.rows):
            if not hctx.incoming[i]:
                out[r0:r1] = h[r0:r1]                       # no incoming relation: passthrough (:129-133)
                if out_max is not None:
                    if h_max is not None and h_max.shape[1] <= out_max.shape[1]:
                        out_max[r0:r1, :h_max.shape[1]] = h_max[r0:r1]
                    else:
                        out_max = None                      # (scales of those rows unknown: the consumer makes its own pass)
        if out_max is not None:
            attach_row_scales(out, out_max)
        attach_col_stats(out, out_cols)                    # (after the pass-through copies: they move the version; those row ranges have no entry)
        ctx.counter = counter
        ctx.save_for_backward(h, kqv, t, out, score, lse, skip, ew, eb, sim_csr, *(() if (drop_mask is None or counter is not None) else (drop_mask,)), *params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = N.load()
        hctx, H, T = ctx.hctx, ctx.H, ctx.T
        bc = None
        if ctx.pool is not None:
            # the gradient arrives as S rows (one per segment of the readout): build the factors the low-rank path below works with,
            # and the [N, D] broadcast only the residual term of the K|Q|V dX epilogue still reads
            h, kqv, t_mean, h_mean, z_mean, score, lse, skip, ew, eb, sim_csr, *params = ctx.saved_tensors
            fwd_factors = None
            if ctx.no_v:
                fwd_factors, params = params[:3], params[3:]
            t = out = None
            prp, pop = ctx.pool
            n_rows, D_ = h.shape
            pre = None
            segs_p = prp.segments_of(hctx.rows)
            if prp.num_segs <= 8192 and segs_p is not None:
                # one launch: the two scalings of the gradient, the skip-gate gradient (segment dots against mean(out) - mean(h), gate map,
                # 1 - sigmoid) and the 1 - sigmoid(skip) factors of the residual term
                seg_gate, type_gate = _gate_tables(hctx, skip, segs_p, prp.num_segs)
                g_pool = g_out.contiguous()
                g_row, g_sum = torch.empty_like(g_pool), torch.empty_like(g_pool)
                pre = (torch.empty_like(skip), torch.empty(T, dtype=torch.float32, device=h.device))
                N.check(lib.wsi_pool_bwd_prep(N.ptr(g_pool), prp.num_segs, D_, pop, N.ptr(prp.counts()), N.ptr(z_mean), N.ptr(h_mean), N.ptr(seg_gate),
                                              N.ptr(skip), skip.shape[0], N.ptr(type_gate), T, N.ptr(g_row), N.ptr(g_sum), N.ptr(pre[0]), N.ptr(pre[1]),
                                              N.stream()), "wsi_pool_bwd_prep")
                bc = SegmentBroadcast.from_parts(prp, g_row, g_sum, z_mean)
            else:
                bc = SegmentBroadcast.from_pooled_gradient(g_out.contiguous(), prp, pop, z_mean)
            # g_v has rank <= S x H: never formed when the fast attention kernels apply (wsi_attn_pool_t; segments numbered type-major)
            collapse = ctx.no_v or _value_collapse_applies(hctx, prp, n_rows, D_, H)
            if collapse:
                g_out = None                      # (the residual term of the dX epilogue comes out of pass 3)
            else:
                g_out = torch.empty((n_rows, D_), dtype=torch.float32, device=h.device)
                N.check(lib.wsi_segment_reduce_bwd(N.ptr(bc.g_row), D_, D_, N.WSI_RED_SUM, N.ptr(prp.chunk_row), N.ptr(prp.chunk_seg),
                                                   prp.num_chunks, N.ptr(prp.seg_chunk), prp.num_segs, None,
                                                   N.ptr(g_out), D_, N.stream()), "wsi_segment_reduce_bwd")
        else:
            h, kqv, t, out, score, lse, skip, ew, eb, sim_csr, *params = ctx.saved_tensors
            t_mean = h_mean = None
            collapse = False
            fwd_factors = None
            pre = None
            g_out = g_out.contiguous()
        g_y = g_out                       # gradient w.r.t. the (un-dropped) a_linear output, before the gate scaling
        if ctx.has_mask and getattr(ctx, "counter", None) is not None:
            g_y = dropout_apply(g_out, ctx.counter)             # the mask regenerated from (seed, row, col): never stored
        elif ctx.has_mask:
            g_y = g_out * params[0]
            params = params[1:]
        P = [params[8 * i:8 * i + 8] for i in range(T)]
        dev = h.device
        n, D = h.shape
        plan = hctx.plan
        E = plan.num_edges
        a_types = hctx.a_types
        rp = hctx.type_rplan()
        grads = [None] * (8 * T)
        gate = lambda i: N.ptr(skip, 4 * hctx.nid[i])

        def gate_grad_launch(stream_ptr, before_launch=None):
            # two launches: the per-type dots sum_rows g_out (out - h), then gate map and sigmoid factor (wsi_gate_grad): a streaming pass over three [n, D] tensors.
            # Everything the launch reads that is produced on the CALLER's stream - the gate table (an asynchronous upload on a cache miss: the first
            # backward of every new graph context) and the buffers (the caching allocator hands out blocks whose last use is ordered on that stream) -
            # exists before `before_launch` (the side stream's wait for the caller's stream) runs.
            seg_gate, _ = _gate_tables(hctx, skip, [(i, i + 1) for i in range(T)], T)
            gs_ = torch.empty_like(skip)
            partial = torch.empty(max(rp.num_chunks * ((D + 255) // 256), 1), dtype=torch.float32, device=dev)
            if before_launch is not None:
                before_launch()
            N.check(lib.wsi_gate_grad(N.ptr(g_out), g_out.stride(0), N.ptr(out), out.stride(0), N.ptr(h), h.stride(0), D, N.ptr(rp.chunk_row), rp.num_chunks,
                                      N.ptr(rp.seg_chunk), rp.num_segs, N.ptr(seg_gate), N.ptr(skip), skip.shape[0], N.ptr(partial), N.ptr(gs_),
                                      stream_ptr), "wsi_gate_grad")
            return gs_, partial

        # a full-depth layer's skip-gate gradient is bandwidth-bound and nobody reads it before this function returns: on the statistics stream,
        # beside the matrix-bound output-projection gradient below (joined at the end)
        early_gate = None
        if (out is not None and g_out is not None and pre is None and _side_stats_on()
                and float(n) * D >= 2.0e7 and not (_LOW_RANK["enabled"] and _annotation(g_out, "_wsi_broadcast") is not None)):
            side = side_stream(dev, "stats")
            gs_, partial_ = gate_grad_launch(ctypes.c_void_p(side.cuda_stream), lambda: side.wait_stream(torch.cuda.current_stream(dev)))
            ev_ = torch.cuda.Event()
            ev_.record(side)
            for t_ in (g_out, out, h, gs_, partial_, skip):
                t_.record_stream(side)
            early_gate = (gs_, ev_)
        # --- output projection: g_t = s * g_out Wa ; gWa = s * g_out^T t ; gba = s * colsum(g_out)
        gkqv_max = _new_row_scale(max(n, plan.num_src_rows), 2, dev, 3 * D, zero=plan.num_src_rows != n)   # pass 2: slot 0 of all n, pass 3: slot 1 of the source rows
        gh_max = _new_row_scale(n, N.gemm_absmax_parts(D), dev, D, zero=False)   # the two dX launches cover every row
        # the layer under a sum / mean readout receives a gradient with ONE distinct row per (graph, node type): rank S = graphs x types
        if bc is None and _LOW_RANK["enabled"] and g_out is not None:
            bc = _annotation(g_out, "_wsi_broadcast")       # attached by the readout's backward to the very tensor it handed down
            if bc is not None:
                EXCHANGE_STATS["broadcast_hits"] += 1
        segs = bc.rp.segments_of(hctx.rows) if bc is not None and bc.rp.num_rows == n else None
        if segs is None:
            bc = None
        groups, wgroups = [], []
        if bc is not None and not ctx.has_mask:
            # g_t[row] = s * g_row[seg] Wa: S rows through the projection, then one broadcast;  gWa = s * sum_seg g_sum[seg] (x) mean_seg(t)
            # (= s * g_out^T t with the rows of a segment summed first);  gba = s * sum_seg g_sum[seg]
            S = bc.rp.num_segs
            gt_seg = (torch.empty if len(a_types) == T else torch.zeros)((S, D), dtype=torch.float32, device=dev)
            if t_mean is None:
                t_mean, _ = _segment_reduce_raw(t, bc.rp, N.WSI_RED_MEAN)
            for i in a_types:
                s0, s1 = segs[i]
                groups.append(dict(A=N.ptr(bc.g_row, s0 * D * 4), lda=D, B=N.ptr(P[i][3]), ldb=D, C=N.ptr(gt_seg, s0 * D * 4), ldc=D,
                                   gate=gate(i), M=s1 - s0, N=D, K=D))
                gw = torch.empty_like(P[i][3])
                gb = torch.empty_like(P[i][7])
                grads[8 * i + 3] = gw
                grads[8 * i + 7] = gb
                wgroups.append(dict(A=N.ptr(bc.g_sum, s0 * D * 4), lda=D, B=N.ptr(t_mean, s0 * D * 4), ldb=D, C=N.ptr(gw), ldc=D,
                                    gate=gate(i), colsum_out=N.ptr(gb), M=D, N=D, K=s1 - s0))
            if not _gemm_small_pair(groups, N.WSI_EPI_SCALE_GATE, wgroups, N.WSI_EPI_SCALE_GATE, dev):
                _gemm(N.WSI_GEMM_NN, N.WSI_EPI_SCALE_GATE, groups, dev)
                _gemm(N.WSI_GEMM_TN, N.WSI_EPI_SCALE_GATE, wgroups, dev)
            g_t, gt_row = gt_seg, bc.rp.row_segment()       # the attention backward reads g_t[gt_row[w]]: S rows that stay in the L2
        else:
            gt_row = None
            g_t = torch.empty((n, D), dtype=torch.float32, device=dev)
            gy_max = row_scales_of(g_y)                # fp16x3 row scales: left by the layer above (its dX epilogue), if any
            gy_cols, t_cols = col_stats_of(g_y), getattr(ctx, "t_cols", None)      # column statistics for the weight gradient, likewise
            for i in a_types:
                r0, r1 = hctx.rows[i]
                groups.append(dict(A=N.ptr(g_y, r0 * D * 4), lda=D, B=N.ptr(P[i][3]), ldb=D, Bw=[P[i][3]], C=N.ptr(g_t, r0 * D * 4), ldc=D,
                                   gate=gate(i), M=r1 - r0, N=D, K=D, **_scale_in(gy_max, r0)))
                gw = torch.empty_like(P[i][3])
                gb = torch.empty_like(P[i][7])
                grads[8 * i + 3] = gw
                grads[8 * i + 7] = gb
                wgroups.append(dict(A=N.ptr(g_y, r0 * D * 4), lda=D, B=N.ptr(t, r0 * D * 4), ldb=D, C=N.ptr(gw), ldc=D,
                                    gate=gate(i), colsum_out=N.ptr(gb), M=D, N=D, K=r1 - r0,
                                    **(gy_cols.consume("a", r0, r1, 0, want_sums=True) if gy_cols is not None else {}),
                                    **(t_cols.consume("b", r0, r1, 0) if t_cols is not None else {})))
            _gemm(N.WSI_GEMM_NN, N.WSI_EPI_SCALE_GATE, groups, dev)
            # the a_linear weight gradient has the whole attention backward of this layer in front of it: in the background
            wr = [P[i][k] for i in a_types for k in (3, 7)]
            keep = [g_y, t, skip] + [c_.bits for c_ in (gy_cols, t_cols) if c_ is not None] + [c_.sums for c_ in (gy_cols,) if c_ is not None and c_.sums is not None]
            outs = [(P[i][k], grads[8 * i + k]) for i in a_types for k in (3, 7)]
            if not (_background_safe(wr) and _gemm_tn_background(N.WSI_EPI_SCALE_GATE, wgroups, dev, keep, wr, outs=outs)):
                _gemm(N.WSI_GEMM_TN, N.WSI_EPI_SCALE_GATE, wgroups, dev)
        # d loss / d skip[nid] = (1 - sigmoid(skip[nid])) * sum over the graph node types i mapped to nid of dots[i],
        # dots[i] = sum over the rows of type i of g_out * (out - h)
        sig_skip = None
        if pre is not None:
            g_skip = pre[0]                            # (wsi_pool_bwd_prep: from the segment means, with the scalings of the gradient)
        elif bc is not None and (out is None or (bc.x_ptr == out.data_ptr() and bc.x_version == out._version)):
            # sum_rows g_out * (out - h) = sum_seg g_sum[seg] . (mean_seg(out) - mean_seg(h)); the readout already holds mean_seg(out)
            from .graph import host_to_device
            q = hctx.cache.get("gate_of_type")
            if q is None:
                q = hctx.cache["gate_of_type"] = host_to_device(
                    [[1.0 if (i in a_types and hctx.nid[i] == g_) else 0.0 for i in range(T)] for g_ in range(skip.shape[0])], torch.float32, dev).view(skip.shape[0], T)
            if h_mean is None:
                h_mean, _ = _segment_reduce_raw(h, bc.rp, N.WSI_RED_MEAN)
            hit = hctx.cache.get("gate_of_seg")            # (rp, matrix): matched by identity of the plan object, which the entry keeps alive
            if hit is None or hit[0] is not bc.rp:
                m = host_to_device([[1.0 if segs[i][0] <= s_ < segs[i][1] else 0.0 for s_ in range(bc.rp.num_segs)] for i in range(T)], torch.float32, dev)
                hit = hctx.cache["gate_of_seg"] = (bc.rp, q @ m.view(T, -1))
            qs = hit[1]
            sig_skip = torch.sigmoid(skip)
            g_skip = (qs @ (bc.g_sum * (bc.x_mean - h_mean)).sum(dim=1)) * (1.0 - sig_skip)
        elif early_gate is not None:
            g_skip = early_gate[0]                     # (launched on the statistics stream at the top of this function; joined before it returns)
        else:
            g_skip = gate_grad_launch(N.stream())[0]
        # --- relation attention backward (pass 1 reads the saved logits and writes the probabilities to `a`)
        a = torch.empty_like(score)
        scratch = torch.empty((3, max(E, 1), H), dtype=torch.float32, device=dev)
        red_ws = torch.empty(1024, dtype=torch.float32, device=dev)
        no_v = fwd_factors is not None            # the forward never computed V: kqv is [n, 2D] (K | Q) and so is its gradient
        ldp = kqv.shape[1]
        blk = ctx.blk                             # column block of K, Q, V (the forward's choice)
        kO, qO, vO = blk[0] * D * 4, blk[1] * D * 4, blk[2] * D * 4
        gkqv = torch.empty_like(kqv)
        g_e = torch.empty(2, dtype=torch.float32, device=dev)
        pool_arg = None
        if collapse:
            S, dk = bc.rp.num_segs, D // H
            from .graph import host_to_device
            # y[tau, seg, h, :] = g_t[seg]_h (W_v^tau rows of head h): what one unit of c[u, bin, h] adds to g_h[u]
            ytab = torch.empty((T, S, H, D), dtype=torch.float32, device=dev)
            groups = []
            for tau in range(T):
                for hh in range(H):
                    groups.append(dict(A=N.ptr(gt_seg, hh * dk * 4), lda=D, B=N.ptr(P[tau][2], hh * dk * D * 4), ldb=D,
                                       C=N.ptr(ytab, ((tau * S) * H + hh) * D * 4), ldc=H * D, M=S, N=D, K=dk))
            _gemm(N.WSI_GEMM_NN, 0, groups, dev)
            if pre is not None:
                omg = pre[1]
            else:
                qt = hctx.cache.get("gate_of_row_type")
                if qt is None:
                    qt = hctx.cache["gate_of_row_type"] = host_to_device(
                        [[1.0 if (hctx.incoming[i] and hctx.nid[i] == g_) else 0.0 for g_ in range(skip.shape[0])] for i in range(T)], torch.float32, dev).view(T, -1)
                omg = 1.0 - qt @ (sig_skip if sig_skip is not None else torch.sigmoid(skip))      # [T]: 1 - s of the type; 1 where the layer passes h through
            r_out = torch.empty((n, D), dtype=torch.float32, device=dev)
            bv_ptrs = _ptr_array([P[tau][6] for tau in range(T)])
            gbv = torch.empty((T, D), dtype=torch.float32, device=dev)
            if no_v:
                ctab, hp, csum = fwd_factors
                # beta[tau, s, h] = g_t[seg]_h . b_v^tau (head h), and the value-bias gradients from the forward's coefficient sums: one launch
                beta = torch.empty((T, S, H), dtype=torch.float32, device=dev)
                N.check(lib.wsi_pool_bwd_bias(N.ptr(gt_seg), T, S, D, H, bv_ptrs, N.ptr(csum), N.ptr(beta), N.ptr(gbv), N.stream()), "wsi_pool_bwd_bias")
            else:
                ctab = torch.empty((n, T, H), dtype=torch.float32, device=dev)
                beta = None
            gtab = None
            if no_v and T * H <= 32 and T * H * (D + 4) * 4 <= 64 * 1024:     # (wsi_heat_pool_gtab's limits: J = T*H <= 32 columns, table in 64 KB of LDS)
                # pass 1's dot products taken once per SOURCE node (T*H per node, one pass over h) instead of H per edge against gathered rows
                gtab = torch.empty((n, T, H), dtype=torch.float32, device=dev)
                with _Timed("heat_attn"), _Timed("heat_attn_bwd_pooled"):
                    N.check(lib.wsi_heat_pool_gtab(N.ptr(h), D, D, H, N.ptr(ytab), N.ptr(beta), N.ptr(bc.rp.chunk_row), N.ptr(bc.rp.chunk_seg),
                                                   bc.rp.num_chunks, S // T, T, N.ptr(gtab), N.stream()), "wsi_heat_pool_gtab")
            pool_desc = N.AttnPool(row_seg=N.ptr(bc.rp.row_segment()), segs_per_type=S // T, n_types=T, y=N.ptr(ytab), g_row=N.ptr(bc.g_row),
                                   omg=N.ptr(omg), r_out=N.ptr(r_out), ldr=D, ctab=N.ptr(ctab), ctab_ready=1 if no_v else 0,
                                   h=N.ptr(h) if no_v else None, ldh=D, beta=N.ptr(beta), gtab=N.ptr(gtab),
                                   edge_seg=N.ptr(_edge_segments(plan)) if gtab is not None else None,
                                   seg_dst=N.ptr(_segment_dst(plan)) if gtab is not None else None)
            pool_arg = ctypes.byref(pool_desc)
        _background_flush(dev)                 # queued weight gradients (this layer's a_linear, the layer above's K|Q|V) run under the attention backward
        with _Timed("heat_attn"), _Timed("heat_attn_bwd_pooled" if collapse else "heat_attn_bwd_full"):
            N.check(lib.wsi_heat_attn_bwd(
                N.ptr(kqv, qO), ldp, N.ptr(kqv, kO), ldp, None if no_v else N.ptr(kqv, vO), ldp, n, plan.num_src_rows, E, D, H,
                N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr),
                N.ptr(plan.colptr), N.ptr(plan.csc_eid), N.ptr(plan.csc_dst),
                N.ptr(plan.inv_rd), N.ptr(plan.order_dst), plan.num_heavy, N.ptr(plan.order_src), _attn_flags(plan), N.ptr(ew), N.ptr(eb),
                N.ptr(g_t), D, N.ptr(gt_row), N.ptr(score), N.ptr(a), N.ptr(lse), N.ptr(scratch[0]), N.ptr(scratch[1]), N.ptr(scratch[2]), N.ptr(red_ws),
                N.ptr(gkqv, qO), ldp, N.ptr(gkqv, kO), ldp, None if no_v else N.ptr(gkqv, vO), ldp,
                N.ptr(g_e), N.ptr(gkqv_max), pool_arg, N.context(), N.stream()), "wsi_heat_attn_bwd")
        # column statistics of the attention gradients for their weight gradient, beside the dX projection below (ops._col_stats_side)
        gk_side = _col_stats_side(gkqv, hctx.rows, dev) if (want_col_stats(n, D, D) and (D % 32 == 0)) else None
        # --- K|Q|V projections: g_h = gkqv [Wk;Wq;Wv] + (1-s) g_out ; gW = gkqv^T h ; gb = colsum(gkqv)
        g_h = torch.empty((n, D), dtype=torch.float32, device=dev)
        chunked = (D % 32 == 0)
        nproj = 2 if collapse else 3              # collapse: K and Q chunks only (columns [0, 2D) of gkqv; V columns, if any, are not written)
        # g_h is the dY of the weight gradient below (the lower layer's output projection, or the input projection): its dX epilogue leaves the
        # column statistics - absmax and sums (the bias gradient) - that launch would otherwise take a pass over g_h for
        gh_cols = ColStats.allocate(hctx.rows, D, dev, sums=True) if (chunked and want_col_stats(n, D, D)) else None
        gh_wrote = True
        if collapse:
            groups = []
            for i, (r0, r1) in enumerate(hctx.rows):
                groups.append(dict(A=N.ptr(gkqv, r0 * ldp * 4), lda=ldp, B=N.ptr(P[i][0]), B1=N.ptr(P[i][1]), b_chunk=D, ldb=D, Bw=[P[i][0], P[i][1]],
                                   C=N.ptr(g_h, r0 * D * 4), ldc=D, R=N.ptr(r_out, r0 * D * 4), ldr=D, M=r1 - r0, N=D, K=2 * D,
                                   **_scale_in(gkqv_max, r0), **_scale_out(gh_max, r0),
                                   **(gh_cols.produce(r0, r1, 0) if gh_cols is not None else {})))
            gh_wrote = _gemm(N.WSI_GEMM_NN, N.WSI_EPI_ADD_R, groups, dev)
        for with_gate in ((True, False) if not collapse else ()):
            idxs = [i for i in range(T) if hctx.incoming[i] == with_gate]
            if not idxs:
                continue
            epi = N.WSI_EPI_ADD_R | (N.WSI_EPI_R_1MG if with_gate else 0)
            if chunked:
                groups = []
                for i in idxs:
                    r0, r1 = hctx.rows[i]
                    wb = sorted(range(3), key=lambda j_: blk[j_])      # the weight of column block 0, 1, 2
                    groups.append(dict(A=N.ptr(gkqv, r0 * 3 * D * 4), lda=3 * D, B=N.ptr(P[i][wb[0]]), B1=N.ptr(P[i][wb[1]]), B2=N.ptr(P[i][wb[2]]),
                                       Bw=[P[i][wb[0]], P[i][wb[1]], P[i][wb[2]]], b_chunk=D, ldb=D, C=N.ptr(g_h, r0 * D * 4), ldc=D, R=N.ptr(g_out, r0 * D * 4), ldr=D,
                                       gate=gate(i) if with_gate else None, M=r1 - r0, N=D, K=3 * D,
                                       **_scale_in(gkqv_max, r0), **_scale_out(gh_max, r0),
                                       **(gh_cols.produce(r0, r1, 0) if gh_cols is not None else {})))
                gh_wrote = _gemm(N.WSI_GEMM_NN, epi, groups, dev) and gh_wrote
            else:
                for j in range(3):
                    groups = []
                    for i in idxs:
                        r0, r1 = hctx.rows[i]
                        groups.append(dict(A=N.ptr(gkqv, (r0 * 3 * D + blk[j] * D) * 4), lda=3 * D, B=N.ptr(P[i][j]), ldb=D,
                                           C=N.ptr(g_h, r0 * D * 4), ldc=D, R=N.ptr(g_out, r0 * D * 4), ldr=D,
                                           gate=gate(i) if with_gate else None, M=r1 - r0, N=D, K=D))
                    _gemm(N.WSI_GEMM_NN, epi if j == 0 else N.WSI_EPI_ACCUMULATE, groups, dev)
        wgroups = []
        h_cols = getattr(ctx, "h_cols", None)
        for i, (r0, r1) in enumerate(hctx.rows):
            for j in range(nproj):
                gw = torch.empty_like(P[i][j])
                gb = torch.empty_like(P[i][4 + j])
                grads[8 * i + j] = gw
                grads[8 * i + 4 + j] = gb
                wgroups.append(dict(A=N.ptr(gkqv, (r0 * ldp + blk[j] * D) * 4), lda=ldp, B=N.ptr(h, r0 * D * 4), ldb=D,
                                    C=N.ptr(gw), ldc=D, colsum_out=N.ptr(gb), M=D, N=D, K=r1 - r0,
                                    **(gk_side[0].consume("a", r0, r1, blk[j] * D, want_sums=True) if gk_side is not None else {}),
                                    **(h_cols.consume("b", r0, r1, 0) if h_cols is not None else {})))
        # a layer with another HEAT layer below it: its K|Q|V weight gradient runs under THAT layer's attention backward
        wr = [P[i][k] for i in range(T) for j in range(nproj) for k in (j, 4 + j)]
        keep = [gkqv, h] + ([h_cols.bits] if h_cols is not None else []) + ([gk_side[0].bits, gk_side[0].sums] if gk_side is not None else [])
        outs = [(P[i][k], grads[8 * i + k]) for i in range(T) for j in range(nproj) for k in (j, 4 + j)]
        evs = [gk_side[1]] if gk_side is not None else []
        if not (ctx.background_dw and _background_safe(wr) and _gemm_tn_background(0, wgroups, dev, keep, wr, evs, outs=outs)):
            for ev in evs:
                torch.cuda.current_stream(dev).wait_event(ev)
            _gemm(N.WSI_GEMM_TN, 0, wgroups, dev)
        if collapse:
            # dW_v^tau (rows of head h) = sum_seg g_t[seg]_h (x) hp[seg, h, tau, :]  (hp: weighted sums of h over the (source type, graph) segments, from
            # the forward when it never computed V, else taken here from pass 3's coefficients);  db_v likewise with the sums of the coefficients
            if not no_v:
                hp, csum = _pooled_factors(h, ctab, bc.rp, T, H)
                N.check(lib.wsi_pool_bwd_bias(N.ptr(gt_seg), T, S, D, H, None, N.ptr(csum), None, N.ptr(gbv), N.stream()), "wsi_pool_bwd_bias")
            wgroups = []
            for tau in range(T):
                gw = torch.empty_like(P[tau][2])
                grads[8 * tau + 2] = gw
                grads[8 * tau + 6] = gbv[tau]
                for hh in range(H):
                    wgroups.append(dict(A=N.ptr(gt_seg, hh * dk * 4), lda=D, B=N.ptr(hp, (hh * T + tau) * D * 4), ldb=H * T * D,
                                        C=N.ptr(gw, hh * dk * D * 4), ldc=D, M=dk, N=D, K=S))
            _gemm(N.WSI_GEMM_TN, 0, wgroups, dev)
        if gh_max is not None and chunked:
            attach_row_scales(g_h, gh_max)
        if gh_cols is not None and gh_wrote and chunked:
            attach_col_stats(g_h, gh_cols)
        if early_gate is not None:
            torch.cuda.current_stream(dev).wait_event(early_gate[1])
        return (g_h, None, None, g_skip, g_e[0:1].view(1, 1), g_e[1:2], None, None, None, *grads)


def heat_layer_fused(h, hctx, H, skip, e_weight, e_bias, params, drop_mask=None, pool=None, background_dw=False):
    """``pool`` = (ReducePlan, "sum" | "mean"): return the readout of the layer's output instead of the output (see _HeatLayerFused).
    ``background_dw``: another HEAT layer's backward follows this one's (it is not the first layer): its K|Q|V weight gradient may run on the
    side stream under that layer's attention backward (``_gemm_tn_background``)."""
    if pool is not None:
        pool = (pool[0], {"sum": N.WSI_RED_SUM, "mean": N.WSI_RED_MEAN}[pool[1]])
    return _HeatLayerFused.apply(h, hctx, H, skip, e_weight, e_bias, drop_mask, pool, background_dw, *params)


# ------------------------------------------------------------------------------------------------
# the trainer's loss
# ------------------------------------------------------------------------------------------------
class _CrossEntropy(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss() with its defaults (mean over the batch; parser.py:182-183) as ONE launch forward (``wsi_cross_entropy``: the loss
    and the gradient factor together) and one tiny scaling backward, where torch takes log_softmax + nll_loss and their two backward kernels
    plus fills.  A label of -100 (the default ``ignore_index``) is ignored as torch ignores it (zero gradient row, mean over the others); any
    other label outside [0, C) - torch's device assert - makes the loss NaN with a zero gradient row and sets the flag the returned tensor carries
    as ``_wsi_bad_label`` (a one-element int32 device tensor; ``trainer.train_one_step`` raises on it where it synchronises anyway)."""

    @staticmethod
    def forward(ctx, logits, labels, bad):
        N.require_cuda(logits, labels)
        logits = logits.contiguous()
        labels = labels.contiguous()
        if labels.dtype != torch.int64 or logits.dim() != 2 or labels.shape != logits.shape[:1]:
            raise ValueError("cross_entropy: logits [B, C] fp32 and int64 labels [B]")
        B, C = logits.shape
        out = torch.empty(1 + B * C, dtype=torch.float32, device=logits.device)
        N.check(N.load().wsi_cross_entropy(N.ptr(logits), N.ptr(labels), B, C, N.ptr(out), N.ptr(out, 4), N.ptr(bad), N.stream()), "wsi_cross_entropy")
        ctx.save_for_backward(out)
        ctx.shape = (B, C)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return out[1:].view(ctx.shape) * g, None, None


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Mean cross entropy (a 0-dim tensor), ``F.cross_entropy(logits, labels)`` with default arguments."""
    bad = torch.zeros(1, dtype=torch.int32, device=logits.device)
    loss = _CrossEntropy.apply(logits, labels, bad)
    loss._wsi_bad_label = bad
    return loss


# ------------------------------------------------------------------------------------------------
# general relation attention (separate q table and stacked K|V table) — HGT
# ------------------------------------------------------------------------------------------------
class _RelationAttention(torch.autograd.Function):
    """t[N,D] from q [N,D] and kv [R,2D] (K at column 0, V at D); plan.src / plan.colptr index kv rows."""

    @staticmethod
    def forward(ctx, q, kv, e_weight, e_bias, plan: GraphPlan, sim_csr, D: int, H: int):
        N.require_cuda(q, kv)
        lib = N.load()
        q = q.contiguous()
        kv = kv.contiguous()
        n, E, S = plan.num_nodes, plan.num_edges, plan.num_segs
        dev = q.device
        t = torch.empty((n, D), dtype=torch.float32, device=dev)
        score = torch.empty((max(E, 1), H), dtype=torch.float32, device=dev)
        lse = torch.empty((max(S, 1), H), dtype=torch.float32, device=dev)
        ew, eb = e_weight.reshape(-1), e_bias.reshape(-1)
        with _Timed("heat_attn"):
            N.check(lib.wsi_heat_attn_fwd(
                N.ptr(q), q.stride(0), N.ptr(kv, 0), kv.stride(0), N.ptr(kv, D * 4), kv.stride(0), n, D, H,
                N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr), N.ptr(plan.order_dst), plan.num_heavy, _attn_flags(plan),
                N.ptr(ew), N.ptr(eb), N.ptr(t), D, N.ptr(score), N.ptr(lse), None, N.context(), N.stream()), "wsi_heat_attn_fwd")
        ctx.plan, ctx.D, ctx.H = plan, D, H
        ctx.save_for_backward(q, kv, ew, eb, sim_csr, score, lse)
        return t

    @staticmethod
    def backward(ctx, g_t):
        lib = N.load()
        plan, D, H = ctx.plan, ctx.D, ctx.H
        q, kv, ew, eb, sim_csr, score, lse = ctx.saved_tensors
        g_t = g_t.contiguous()
        n, E = plan.num_nodes, plan.num_edges
        dev = q.device
        a = torch.empty_like(score)
        scratch = torch.empty((3, max(E, 1), H), dtype=torch.float32, device=dev)
        red_ws = torch.empty(1024, dtype=torch.float32, device=dev)
        gq = torch.empty_like(q)
        gkv = torch.empty_like(kv)
        g_e = torch.empty(2, dtype=torch.float32, device=dev)
        with _Timed("heat_attn"):
            N.check(lib.wsi_heat_attn_bwd(
                N.ptr(q), q.stride(0), N.ptr(kv, 0), kv.stride(0), N.ptr(kv, D * 4), kv.stride(0),
                n, plan.num_src_rows, E, D, H,
                N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src), N.ptr(sim_csr),
                N.ptr(plan.colptr), N.ptr(plan.csc_eid), N.ptr(plan.csc_dst),
                N.ptr(plan.inv_rd), N.ptr(plan.order_dst), plan.num_heavy, N.ptr(plan.order_src), _attn_flags(plan), N.ptr(ew), N.ptr(eb),
                N.ptr(g_t), g_t.stride(0), None, N.ptr(score), N.ptr(a), N.ptr(lse), N.ptr(scratch[0]), N.ptr(scratch[1]), N.ptr(scratch[2]), N.ptr(red_ws),
                N.ptr(gq), gq.stride(0), N.ptr(gkv, 0), gkv.stride(0), N.ptr(gkv, D * 4), gkv.stride(0),
                N.ptr(g_e), None, None, N.context(), N.stream()), "wsi_heat_attn_bwd")
        return gq, gkv, g_e[0:1].view(1, 1), g_e[1:2], None, None, None, None


def relation_attention(q, kv, e_weight, e_bias, plan: GraphPlan, sim_csr, D: int, H: int) -> torch.Tensor:
    return _RelationAttention.apply(q, kv, e_weight, e_bias, plan, sim_csr, D, H)


# ------------------------------------------------------------------------------------------------
# gated linear:  z[rows_i] = s_i * (t[rows_i] W_i^T + b_i) + (1 - s_i) * h[rows_i],  s_i = sigmoid(skip[nid_i])
# (models/HGT.py:121-122, models/HEATNet4.py:134-135 when not fused into the whole layer)
# ------------------------------------------------------------------------------------------------
class _GatedLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, h, skip, rows, nids, rplan, seg_of, n_g, drop_mask, *params):
        N.require_cuda(t, h)
        t = t.contiguous()
        h = h.contiguous()
        dev = t.device
        n, D = h.shape
        K = t.shape[1]
        ws, bs = params[:n_g], params[n_g:]
        covered = sum(b - a for a, b in rows) == n
        z = torch.empty((n, D), dtype=torch.float32, device=dev) if covered else h.clone()
        groups = []
        for i, (r0, r1) in enumerate(rows):
            groups.append(dict(A=N.ptr(t, r0 * K * 4), lda=K, B=N.ptr(ws[i]), ldb=K, C=N.ptr(z, r0 * D * 4), ldc=D,
                               bias=N.ptr(bs[i]), R=N.ptr(h, r0 * D * 4), ldr=D, gate=N.ptr(skip, 4 * nids[i]),
                               Mm=N.ptr(drop_mask, r0 * D * 4) if drop_mask is not None else None, ldm=D,
                               M=r1 - r0, N=D, K=K))
        _gemm(N.WSI_GEMM_NT, N.WSI_EPI_GATED_SKIP | (N.WSI_EPI_MUL_M if drop_mask is not None else 0), groups, dev)
        ctx.rows, ctx.nids, ctx.rplan, ctx.seg_of, ctx.n_g, ctx.covered = rows, nids, rplan, seg_of, n_g, covered
        ctx.has_mask = drop_mask is not None
        ctx.save_for_backward(t, h, z, skip, *(() if drop_mask is None else (drop_mask,)), *ws)
        return z

    @staticmethod
    def backward(ctx, g_z):
        rows, nids, rp, seg_of, n_g = ctx.rows, ctx.nids, ctx.rplan, ctx.seg_of, ctx.n_g
        t, h, z, skip, *ws = ctx.saved_tensors
        g_z = g_z.contiguous()
        g_y = g_z                         # gradient w.r.t. the (un-dropped) linear output, before the gate scaling
        if ctx.has_mask:
            g_y = g_z * ws[0]
            ws = ws[1:]
        dev = t.device
        n, D = h.shape
        K = t.shape[1]
        gate = lambda i: N.ptr(skip, 4 * nids[i])
        g_t = (torch.empty if ctx.covered else torch.zeros)((n, K), dtype=torch.float32, device=dev)
        gws, gbs, groups, wgroups = [], [], [], []
        for i, (r0, r1) in enumerate(rows):
            groups.append(dict(A=N.ptr(g_y, r0 * D * 4), lda=D, B=N.ptr(ws[i]), ldb=K, C=N.ptr(g_t, r0 * K * 4), ldc=K,
                               gate=gate(i), M=r1 - r0, N=K, K=D))
            gw = torch.empty_like(ws[i])
            gws.append(gw)
            gbs.append(torch.empty(D, dtype=torch.float32, device=dev))
            wgroups.append(dict(A=N.ptr(g_y, r0 * D * 4), lda=D, B=N.ptr(t, r0 * K * 4), ldb=K, C=N.ptr(gw), ldc=K,
                                gate=gate(i), colsum_out=N.ptr(gbs[-1]), M=D, N=K, K=r1 - r0))
        _gemm(N.WSI_GEMM_NN, N.WSI_EPI_SCALE_GATE, groups, dev)
        _gemm(N.WSI_GEMM_TN, N.WSI_EPI_SCALE_GATE, wgroups, dev)
        sig = torch.sigmoid(skip)
        dots = segment_dot_diff(g_z, z, h, rp)                          # segment seg_of[i] of rp == rows[i]
        g_skip = torch.zeros_like(skip)
        scale = torch.ones(n, 1, dtype=torch.float32, device=dev)       # rows outside `rows` pass h through: dz/dh = 1
        for i, (r0, r1) in enumerate(rows):
            s_i = sig[nids[i]]
            g_skip[nids[i]] = g_skip[nids[i]] + dots[seg_of[i]] * (1.0 - s_i)
            scale[r0:r1] = 1.0 - s_i
        g_h = g_z * scale
        return (g_t, g_h, g_skip, None, None, None, None, None, None, *gws, *gbs)


def gated_linear(t, h, skip, rows, nids, rplan, seg_of, weights, biases, drop_mask=None):
    """``rplan``: ReducePlan over (gap-filled) row ranges; ``seg_of[i]`` = its segment holding ``rows[i]``.  ``drop_mask``: [n, D]
    keep mask already scaled by 1 / (1 - p) (the nn.Dropout between the linear and the gate, models/HGT.py:121), or None."""
    return _GatedLinear.apply(t, h, skip, rows, nids, rplan, seg_of, len(weights), drop_mask, *weights, *biases)


# ------------------------------------------------------------------------------------------------
# GELU, LayerNorm, GraphConv aggregation
# ------------------------------------------------------------------------------------------------
class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N.require_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        N.check(N.load().wsi_gelu_fwd(N.ptr(x), N.ptr(y), x.numel(), N.stream()), "wsi_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        N.check(N.load().wsi_gelu_bwd(N.ptr(x), N.ptr(gy), N.ptr(gx), x.numel(), N.stream()), "wsi_gelu_bwd")
        return gx


def gelu(x: torch.Tensor) -> torch.Tensor:
    return _Gelu.apply(x)


class _LayerNorm(torch.autograd.Function):
    """Per-row LayerNorm with per-node-type affine parameters: gamma/beta [P,D], row_param[n] int32 -> P;
    ``rplan`` segments = row ranges sharing a parameter row, ``seg_param[s]`` = that parameter row."""

    @staticmethod
    def forward(ctx, x, gamma, beta, row_param, rplan, seg_param, eps):
        N.require_cuda(x)
        x = x.contiguous()
        gamma = gamma.contiguous()
        beta = beta.contiguous()
        n, D = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((max(n, 1), 2), dtype=torch.float32, device=x.device)
        N.check(N.load().wsi_layernorm_fwd(N.ptr(x), D, n, D, float(eps), N.ptr(gamma), N.ptr(beta), N.ptr(row_param),
                                           N.ptr(y), D, N.ptr(stats), N.stream()), "wsi_layernorm_fwd")
        ctx.rplan, ctx.seg_param = rplan, seg_param
        ctx.save_for_backward(x, gamma, row_param, stats)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, row_param, stats = ctx.saved_tensors
        gy = gy.contiguous()
        n, D = x.shape
        gx = torch.empty_like(x)
        xhat_gy = torch.empty_like(x)
        N.check(N.load().wsi_layernorm_bwd(N.ptr(gy), D, N.ptr(x), D, n, D, N.ptr(gamma), N.ptr(row_param), N.ptr(stats),
                                           N.ptr(gx), D, N.ptr(xhat_gy), D, N.stream()), "wsi_layernorm_bwd")
        gg_seg = _segment_reduce_raw(xhat_gy, ctx.rplan, N.WSI_RED_SUM)[0]     # [S, D]
        gb_seg = _segment_reduce_raw(gy, ctx.rplan, N.WSI_RED_SUM)[0]
        ggamma = torch.zeros_like(gamma)
        gbeta = torch.zeros_like(gamma)
        for s_, p_ in enumerate(ctx.seg_param):
            ggamma[p_] = ggamma[p_] + gg_seg[s_]
            gbeta[p_] = gbeta[p_] + gb_seg[s_]
        return gx, ggamma, gbeta, None, None, None, None


def layer_norm(x, gamma, beta, row_param, rplan, seg_param, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, row_param, rplan, seg_param, eps)


class _GraphConvAggregate(torch.autograd.Function):
    """y = act(indeg^-1/2 * sum_{u->w} outdeg^-1/2[u] * z[u] + bias)   (DGL GraphConv norm='both', after/before the GEMM)."""

    @staticmethod
    def forward(ctx, z, bias, hp, relu: bool):
        N.require_cuda(z)
        z = z.contiguous()
        n, D = z.shape
        y = torch.empty_like(z)
        N.check(N.load().wsi_spmm_sum(N.ptr(z), D, n, D, N.ptr(hp.rowptr), N.ptr(hp.src), N.ptr(getattr(hp, "edge_w", None)), N.ptr(hp.out_norm), N.ptr(hp.in_norm),
                                      N.ptr(bias), 1 if relu else 0, None, 0, N.ptr(y), D, N.stream()), "wsi_spmm_sum")
        ctx.hp, ctx.relu, ctx.has_bias = hp, relu, bias is not None
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        hp = ctx.hp
        gy = gy.contiguous()
        n, D = y.shape
        gz = torch.empty_like(y)
        N.check(N.load().wsi_spmm_sum(N.ptr(gy), D, n, D, N.ptr(hp.colptr), N.ptr(hp.csc_dst), N.ptr(getattr(hp, "edge_w_csc", None)), N.ptr(hp.in_norm), N.ptr(hp.out_norm),
                                      None, 0, N.ptr(y) if ctx.relu else None, D, N.ptr(gz), D, N.stream()), "wsi_spmm_sum")
        gb = None
        if ctx.has_bias:
            gb = (gy * (y > 0).to(gy.dtype)).sum(0) if ctx.relu else gy.sum(0)
        return gz, gb, None, None


def graph_conv_aggregate(z, bias, hplan, relu: bool):
    return _GraphConvAggregate.apply(z, bias, hplan, relu)


# ------------------------------------------------------------------------------------------------
# ASAPPooling edge kernels (pooling/ASAP.py:158-179)
# ------------------------------------------------------------------------------------------------
class EdgeCSR:
    """An edge list grouped by the node that aggregates it (CSR: ``rowptr``/``src`` = gathered node per edge) and by the
    gathered node (CSC over the same edge numbering), plus ``perm`` (CSR position -> original edge).  Attribute names
    follow GraphPlan so that ``graph_conv_aggregate`` accepts it; ``in_norm``/``out_norm`` (may stay None = 1) scale the
    aggregating / gathered node."""

    def __init__(self, group: torch.Tensor, other: torch.Tensor, n: int):
        from .graph import _count
        dev = group.device
        E = int(group.numel())
        self.num_nodes, self.num_edges = int(n), E
        perm = torch.sort(group, stable=True).indices if E else group
        g_s, o_s = group[perm], other[perm]
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        colptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if E:
            rowptr[1:] = torch.cumsum(_count(g_s, n), 0)
            colptr[1:] = torch.cumsum(_count(o_s, n), 0)
        cperm = torch.sort(o_s, stable=True).indices if E else o_s
        self.perm = perm
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.src = o_s.to(torch.int32).contiguous()
        self.colptr = colptr.to(torch.int32).contiguous()
        self.csc_eid = cperm.to(torch.int32).contiguous()
        self.csc_dst = g_s[cperm].to(torch.int32).contiguous() if E else g_s.to(torch.int32)
        self.in_norm = None
        self.out_norm = None
        self.edge_w = None          # optional per-edge weights in CSR order (``with_weights``) ...
        self.edge_w_csc = None      # ... and in CSC order (the backward gathers along it)

    def with_weights(self, w: Optional[torch.Tensor]) -> "EdgeCSR":
        """A shallow copy carrying per-edge weights ``w`` (ORIGINAL edge order; constants: no gradient flows to them) for
        ``graph_conv_aggregate`` (``wsi_spmm_sum``'s edge_w) and ``wsi_stas``."""
        import copy
        if w is not None and w.requires_grad:
            raise RuntimeError("EdgeCSR.with_weights: edge weights are treated as constants (ASAPPooling only ever passes detached values, pooling/ASAP.py:97)")
        c = copy.copy(self)
        if w is None:
            c.edge_w = c.edge_w_csc = None
        else:
            c.edge_w = w.detach().reshape(-1).to(torch.float32)[self.perm].contiguous()
            c.edge_w_csc = c.edge_w[self.csc_eid.long()].contiguous()
        return c


class _CsrGatherMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ec: EdgeCSR):
        N.require_cuda(x)
        x = x.contiguous()
        n, D = ec.num_nodes, x.shape[1]
        out = torch.empty((n, D), dtype=torch.float32, device=x.device)
        arg = torch.empty((n, D), dtype=torch.int32, device=x.device)
        N.check(N.load().wsi_csr_gather_max_fwd(N.ptr(x), D, n, D, N.ptr(ec.rowptr), N.ptr(ec.src), N.ptr(out), D, N.ptr(arg), N.stream()),
                "wsi_csr_gather_max_fwd")
        ctx.ec, ctx.rows = ec, x.shape[0]
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        ec = ctx.ec
        g = g.contiguous()
        D = g.shape[1]
        gx = torch.empty((ctx.rows, D), dtype=torch.float32, device=g.device)
        N.check(N.load().wsi_csr_gather_max_bwd(N.ptr(g), D, N.ptr(arg), ctx.rows, D, N.ptr(ec.colptr), N.ptr(ec.csc_eid), N.ptr(ec.csc_dst),
                                                N.ptr(gx), D, N.stream()), "wsi_csr_gather_max_bwd")
        return gx, None


def csr_gather_max(x: torch.Tensor, ec: EdgeCSR) -> torch.Tensor:
    """out[i] = max over the edges grouped at i of x[j]   (torch_scatter.scatter_max of pooling/ASAP.py:163; 0 for empty groups)."""
    return _CsrGatherMax.apply(x, ec)


class _AsapAttend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, x, ec: EdgeCSR, slope: float):
        N.require_cuda(a, b, x)
        a_shape, b_shape = a.shape, b.shape
        a, b, x = a.contiguous().view(-1), b.contiguous().view(-1), x.contiguous()
        n, D = ec.num_nodes, x.shape[1]
        out = torch.empty((n, D), dtype=torch.float32, device=x.device)
        score = torch.empty(max(ec.num_edges, 1), dtype=torch.float32, device=x.device)
        N.check(N.load().wsi_asap_attend_fwd(N.ptr(a), N.ptr(b), N.ptr(x), D, n, D, N.ptr(ec.rowptr), N.ptr(ec.src), float(slope),
                                             N.ptr(score), N.ptr(out), D, N.stream()), "wsi_asap_attend_fwd")
        ctx.ec, ctx.slope = ec, float(slope)
        ctx.shapes = (a_shape, b_shape)
        ctx.save_for_backward(a, b, x, score)
        score_orig = torch.empty(ec.num_edges, dtype=torch.float32, device=x.device)
        score_orig[ec.perm] = score[:ec.num_edges]
        ctx.mark_non_differentiable(score_orig)        # the reference detaches it where it is reused (ASAP.py:97)
        return out, score_orig

    @staticmethod
    def backward(ctx, g_out, _g_score):
        a, b, x, score = ctx.saved_tensors
        ec = ctx.ec
        g_out = g_out.contiguous()
        n, D = ec.num_nodes, x.shape[1]
        dev = x.device
        gpre = torch.empty(max(ec.num_edges, 1), dtype=torch.float32, device=dev)
        g_a = torch.empty(n, dtype=torch.float32, device=dev)
        g_b = torch.empty(n, dtype=torch.float32, device=dev)
        gx = torch.empty((n, D), dtype=torch.float32, device=dev)
        N.check(N.load().wsi_asap_attend_bwd(N.ptr(a), N.ptr(b), N.ptr(x), D, n, D, N.ptr(ec.rowptr), N.ptr(ec.src),
                                             N.ptr(ec.colptr), N.ptr(ec.csc_eid), N.ptr(ec.csc_dst), ctx.slope,
                                             N.ptr(score), N.ptr(g_out), D, N.ptr(gpre), N.ptr(g_a), N.ptr(g_b), N.ptr(gx), D, N.stream()),
                "wsi_asap_attend_bwd")
        return g_a.view(ctx.shapes[0]), g_b.view(ctx.shapes[1]), gx, None, None


def asap_attend(a: torch.Tensor, b: torch.Tensor, x: torch.Tensor, ec: EdgeCSR, negative_slope: float):
    """(out [n,D], score [E] in the ORIGINAL edge order) of pooling/ASAP.py:167-179; ``a`` [n] already holds the bias."""
    return _AsapAttend.apply(a, b, x, ec, negative_slope)
