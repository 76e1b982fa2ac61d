#!/usr/bin/env python
"""bench.py — HEATNet4 training-step throughput on synthetic WSI graphs (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: forward + cross-entropy + backward of
HEATNet4(1024,512,2,2 layers,4 heads) on a block-diagonal batch of 8 synthetic 10k-node / 6-relation
graphs per GPU (BASELINE configs[2]/[3]), gradient all-reduce (N>1) and the Adam step
(the reference's optimizer, parser.py:33-38) — inputs resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="graphs per GPU")
    ap.add_argument("--nodes", type=int, default=10000, help="nodes per graph")
    ap.add_argument("--in-dim", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--model", default="HEATNet4", choices=["HEATNet4", "HEATNet2"],
                    help="HEATNet4 is the metric's model; HEATNet2 (configs[1]: --hidden 256 --nodes 5000) is a side measurement")
    ap.add_argument("--dst-mode", default="uniform", choices=["uniform", "hub"])
    ap.add_argument("--schema", default="synthetic", choices=["synthetic", "real"],
                    help="synthetic = SURVEY 8d's 3 node types / 6 relations (the metric's graph); real = the reference graph "
                         "constructor's schema: 6 node types, up to 72 (src type, sign, dst type) relations, 12 relation slots per "
                         "node (graph_constructor.py:276-297; side measurement, never `value` of the BASELINE metric)")
    ap.add_argument("--slide-sizes", default="equal", choices=["equal", "mixed"],
                    help="equal (the metric: every slide has --nodes patches) | mixed: the world x batch slides of a step have 2 / 0.2 / 2 / 0.2 / 1 / 1 / 1 / 1 "
                         "x --nodes patches (real slides span 10^3..10^4) and are dealt to the ranks by --shard; side measurement of the sharding policy")
    ap.add_argument("--shard", default="balanced", choices=["balanced", "round_robin"],
                    help="--slide-sizes mixed: dist.shard by edge count (same slide count per rank, equalised edges) or round-robin")
    ap.add_argument("--dropout", type=float, default=0.0, help="feat_drop of the HEAT layers (SURVEY 8d fixes 0.0 for the metric; "
                    "the reference's training configs use 0.2, which takes the layers' train-mode branch)")
    ap.add_argument("--gemm", default="auto", choices=["fp32", "bf16x6", "fp16x3", "auto"],
                    help="arithmetic of the projection GEMMs in the timed region, all three fp32-class (every model-level parity test "
                         "runs under each with the same 1e-4 tolerance; error against float64 <= the fp32 MFMA path's own): "
                         "fp16x3 = fp32 EMULATED on the fp16 matrix cores: operands scaled per row by a power of two, split into 2 fp16 "
                         "terms (2^-23 relative; <= 2^-21 per product), 3 cross products summed in fp32 (weight gradients: the same split scaled per column); bf16x6 = exact 3-way "
                         "bf16 split, 6 cross products; fp32 = v_mfma_f32_32x32x2_f32; auto (default, what `value` is quoted on) = per "
                         "launch fp16x3 where its pre-pass is amortised (every projection of the default workload), else bf16x6.  The "
                         "other modes are timed too and reported as other_gemm_modes")
    ap.add_argument("--no-alt-gemm", action="store_true", help="skip the extra timed leg in the other GEMM arithmetic")
    ap.add_argument("--no-knn", action="store_true", help="skip the extra leg on WSI-like kNN graphs in locality order (`knn_locality`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-captured", action="store_true", help="skip the single_graph_step leg (one graph per step, eager vs one hipGraph)")
    ap.add_argument("--no-training-config", action="store_true", help="skip the training_config leg (the same steps with the reference's feat_drop 0.2)")
    ap.add_argument("--no-full-depth", action="store_true", help="skip the full_depth_last_layer leg (kernel traces of the headline formulation alone)")
    ap.add_argument("--torch-adam", action="store_true", help="step with torch.optim.Adam(fused=True) instead of wsi_hgnn_amd.optim.Adam (same arithmetic)")
    ap.add_argument("--side-statistics", default="on", choices=["on", "off"], help="off: the column statistics of the attention gradients and the early skip-gate "
                                                                                    "gradient stay on the caller's stream (ops.set_side_column_statistics)")
    ap.add_argument("--side-streams", type=int, default=1, choices=[1, 2], help="2: background weight gradients and column statistics on a side stream each instead of sharing one (ops.set_side_stream_count)")
    ap.add_argument("--pcie", action="store_true", help="additionally time steps fed by the prefetching host->device loader "
                                                        "(PCIe-inclusive rate; reported as an extra field, never as `value`)")
    ap.add_argument("--background-dw", choices=["on", "off"], default="on",
                    help="off: every weight-gradient GEMM in order on the caller's stream (ops.set_background_weight_gradients(False)): what a kernel trace "
                         "needs for per-launch durations (tools/profile_round.sh)")
    ap.add_argument("--pmc", dest="pmc", action="store_true", default=None,
                    help="measure roofline.traffic in THIS run: re-execute the workload (2 steps) twice under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` "
                         "(separate passes, no trace domains, as the MI355X guide prescribes; FETCH x2 gfx950 correction) and read the per-kernel fabric-side "
                         "bytes from their CSVs.  DEFAULT since round 6 whenever rocprofv3 is on the box and the run is the one-GPU default workload "
                         "(~40 s for the two child runs); --no-pmc reads the figure from the committed summary of the same passes instead (labelled)")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false")
    ap.add_argument("--dp-overlap", type=int, default=1, choices=[0, 1],
                    help="N > 1: 1 = the gradient all-reduce goes out in pieces from autograd hooks while backward runs (GradBucket(overlap=True)), "
                         "0 = one blocking collective after backward")
    ap.add_argument("--one-device", action="store_true",
                    help="TESTING ONLY: every rank uses cuda:0 (dry run of the multi-rank code path on a 1-GPU box; never for reported numbers)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo with --one-device)")
    return ap.parse_args()


def dense_flops(N, in_dim, D, L):
    """Algorithmic flops of the projections for one fwd+bwd (K/Q/V/A once per node type);
    the input projection has no dX (features need no gradient)."""
    fwd = 2.0 * N * in_dim * D + L * 4 * 2.0 * N * D * D
    bwd = 2.0 * N * in_dim * D + L * 4 * 2 * 2.0 * N * D * D
    return fwd + bwd


def edge_bytes(N, E, D, L):
    """Compulsory-traffic model of the relation-attention kernels, fwd+bwd."""
    return L * ((4 * N * D * 4 + 24 * E) + (8 * N * D * 4 + 24 * E))


def whole_model_bytes(N, E, F, D, L):
    """Compulsory HBM bytes of the whole step, fwd+bwd: every dense activation touched once per pass
    (input features, adapted states, per layer the K|Q|V table, the aggregated t, the gated output), x3 for forward +
    backward, plus the edge phase."""
    dense_fwd = N * 4 * (F + D + L * 7 * D)
    return 3 * dense_fwd + edge_bytes(N, E, D, L)


def launch_ranks(args, one_dev: bool) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: check the box has N GPUs, then replace this process
    by `torch.distributed.run --nproc-per-node N bench.py <same arguments>` (one rank per GPU over RCCL).  Never returns."""
    import socket
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not one_dev:
        print(f"bench.py: --gpus {args.gpus} requested but this box has {ndev} GPU(s); refusing to report an {args.gpus}-GPU "
              f"number from fewer ranks", file=sys.stderr)
        sys.exit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def cpu_info():
    """(model name, physical cores, logical cpus) of the host, from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    return model, (min(len(cores), logical) if cores else logical), logical


def main():
    args = parse()
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP kernels have no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # --one-device --backend gloo: dry-run of the multi-rank code path on a 1-GPU box
    # (all ranks share cuda:0, collectives go through gloo) - for testing only, never for reported numbers
    one_dev = args.one_device
    backend = args.backend
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args, one_dev)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); the line would misreport n_gpus", file=sys.stderr)
        sys.exit(3)
    if world > 1 and not one_dev and torch.cuda.device_count() < world:
        if rank == 0:
            print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        sys.exit(3)
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__
    __graft_entry__.build()
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic, ops
    from wsi_hgnn_amd.dist import GradBucket

    n_types = 6 if args.schema == "real" else 3
    nd = {str(i): i for i in range(n_types)}
    torch.manual_seed(611)
    model = getattr(models, args.model)(args.in_dim, args.hidden, 2, args.layers, args.heads, nd, args.dropout, "mean").to(dev)
    model.train()

    if args.slide_sizes == "mixed":
        # a pool of world x batch slides of very different sizes, dealt to the ranks by dist.shard: every rank computes the same table from the
        # slides' edge counts (8 per patch) without communicating and builds only its own share
        from wsi_hgnn_amd.dist import shard
        factors = [2.0, 0.2, 2.0, 0.2, 1.0, 1.0, 1.0, 1.0]
        pool_nodes = [int(args.nodes * factors[i % len(factors)]) for i in range(world * args.batch)]
        mine = shard(list(range(len(pool_nodes))), rank, world, weights=[8 * n for n in pool_nodes] if args.shard == "balanced" else None)
        gs = [synthetic.hetero_graph(pool_nodes[i], args.in_dim, seed=611 + i, dst_mode=args.dst_mode) for i in mine]
        G_cpu = W.batch(gs)
        labels = torch.randint(0, 2, (len(gs),), generator=torch.Generator().manual_seed(611 + 1000 * rank + 999))
    elif args.schema == "real":
        # slides differ in which of the 72 relations occur; the batch uses the union schema (synthetic.real_schema_batch)
        G_cpu, labels = synthetic.real_schema_batch(args.batch, args.nodes, args.in_dim, rank=rank, dst_mode=args.dst_mode)
    else:
        G_cpu, labels = synthetic.hetero_batch(args.batch, args.nodes, args.in_dim, rank=rank, dst_mode=args.dst_mode)
    G = G_cpu.to(dev)
    labels = labels.to(dev)
    n_nodes, n_edges = G.num_nodes(), G.num_edges()
    ce = torch.nn.CrossEntropyLoss()                    # the reference's loss (parser.py:182-183) ...
    from wsi_hgnn_amd.trainer import apply_loss
    loss_fn = lambda pred, y: apply_loss(ce, pred, y)   # ... applied the way trainer.train_one_step applies it (one launch each way on GPU logits)

    # probe step: builds the kernel plan
    out = model(G)
    loss_fn(out, labels).backward()
    bucket = GradBucket.from_model(model, overlap=bool(args.dp_overlap))       # every parameter the architecture can reach (dist.py); dead ones stay out
    # the reference's optimizer (torch.optim.Adam(lr, weight_decay), parser.py:33-38) with the same arithmetic in one launch
    # (wsi_hgnn_amd.optim.Adam -> wsi_adam_step); --torch-adam: torch's own fused implementation (two launches on this model)
    if args.torch_adam:
        opt = torch.optim.Adam(bucket.params, lr=1e-5, weight_decay=5e-3, fused=True)
    else:
        from wsi_hgnn_amd.optim import Adam as WsiAdam
        opt = WsiAdam(bucket.params, lr=1e-5, weight_decay=5e-3)

    def step():
        opt.zero_grad(set_to_none=True)
        o = model(G)
        l = loss_fn(o, labels)
        bucket.arm()                                # N > 1: pieces of the gradient are all-reduced while backward runs (dist.py)
        l.backward()
        with ops._Timed("grad_allreduce"):          # what is left of the collective after backward
            bucket.all_reduce_mean()
        opt.step()
        return l

    ops.set_gemm_precision(args.gemm)
    ops.set_background_weight_gradients(args.background_dw == "on")
    ops.set_side_column_statistics(args.side_statistics == "on")
    ops.set_side_stream_count(args.side_streams)
    for _ in range(args.warmup):
        step()
    # benchmark hygiene: everything allocated so far (torch, the model, the plan caches) goes to the collector's permanent generation, so that a
    # full cyclic collection cannot land in a timed region - on this process image one takes ~80 ms, i.e. +4 ms on the mean of 20 steps
    # (found with tools/host_time_probe.py: one 78 ms host step around the 20th optimizer step, none ever after).  No work is skipped by it.
    import gc
    gc.collect()
    gc.freeze()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
            torch.cuda.synchronize()

    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step marks on the launch stream: the median beside the mean
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        last = step()
        marks[i + 1].record()
    sync()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
        et = torch.tensor([float(n_edges)], device=dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.SUM)
        total_edges = et.item()
        per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([float(n_edges)], device=dev, dtype=torch.float64))
        edges_per_rank = [int(x.item()) for x in per_rank]
    else:
        total_edges = float(n_edges)
        edges_per_rank = [int(n_edges)]
    # the slowest rank sets the step: max over ranks of the edges a rank processes / the mean (1.0 = even shards)
    shard_balance = {"policy": ("equal slides" if args.slide_sizes == "equal" else args.shard), "edges_per_rank": edges_per_rank,
                     "imbalance": round(max(edges_per_rank) / (sum(edges_per_rank) / len(edges_per_rank)), 4),
                     "note": "dist.shard(weights=edge counts): same number of slides on every rank, edge totals equalised (longest-processing-time-first); "
                             "the metric's slides are all the same size, --slide-sizes mixed exercises the policy"}
    ms_per_step = dt / args.steps * 1e3
    value = total_edges * args.steps / dt

    # ---- per-kernel HIP-event pass (same steps, events on the launch stream = torch's current stream)
    roofline = None
    edge_phase = None
    allreduce_ms = None

    PMC_CSV = "profiles/r06_hbm_traffic_pmc.csv"
    pmc_live = None          # --pmc: {kernel name: (launches, fabric-side bytes per launch)} measured by two rocprofv3 passes of this workload

    def measure_pmc():
        """Two child runs of this script (same workload arguments, 2 steps, no side legs) under rocprofv3 --pmc, one counter each."""
        import csv, shutil, subprocess, tempfile
        rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
        if not os.path.exists(rocprof):
            return None
        child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt-gemm", "--no-knn", "--no-full-depth",
                 "--no-captured", "--no-training-config", "--no-kernel-timing", "--no-pmc", "--batch", str(args.batch), "--nodes", str(args.nodes), "--in-dim", str(args.in_dim),
                 "--hidden", str(args.hidden), "--layers", str(args.layers), "--heads", str(args.heads), "--model", args.model, "--dst-mode", args.dst_mode,
                 "--schema", args.schema, "--dropout", str(args.dropout), "--gemm", args.gemm]
        vals = {}
        for counter, factor in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):      # KiB; FETCH_SIZE reports half the bytes of 16-byte-per-lane reads on gfx950
            d = tempfile.mkdtemp(prefix="wsi_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pm", "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150)
            found = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not found:
                shutil.rmtree(d, ignore_errors=True)
                return None
            for row in csv.DictReader(open(found[0])):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0]
                v = vals.setdefault(name, {"n": {}, "b": 0.0})
                v["n"][counter] = v["n"].get(counter, 0) + 1
                v["b"] += factor * float(row["Counter_Value"]) * 1024.0
            shutil.rmtree(d, ignore_errors=True)
        return {k: (max(v["n"].values()), v["b"] / max(v["n"].values())) for k, v in vals.items() if "wsi::" in k}

    default_run = (args.schema == "synthetic" and args.model == "HEATNet4" and args.batch == 8 and args.nodes == 10000 and args.hidden == 512
                   and args.dst_mode == "uniform" and args.dropout == 0.0 and args.slide_sizes == "equal" and not args.no_kernel_timing)
    want_pmc = args.pmc if args.pmc is not None else default_run          # default: on for the driver's plain `python bench.py`
    if want_pmc and world == 1:
        try:
            pmc_live = measure_pmc()
        except Exception as exc:          # (the profiler missing or refusing must not cost the bench line)
            print(f"[bench] --pmc failed: {exc}", file=sys.stderr)

    def pmc_source():
        return ("measured by this run: two child runs of the same workload under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; FETCH x2 gfx950 correction)"
                if pmc_live else f"{PMC_CSV} (separate rocprofv3 --pmc passes of this command, FETCH x2 gfx950 correction); not measured by this run (bench.py --pmc measures it)")

    def pmc_traffic(prefixes):
        """Average fabric-side bytes per launch of the kernels named by `prefixes`.  NOT measured by this run (PMC counters
        need a rocprofv3 wrapper): read from the committed summary of separate `rocprofv3 --pmc` passes over this same command
        (tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE passes, gfx950 x2 read correction); None when that file is absent or
        this run is not the default workload it was collected on."""
        if pmc_live:
            tot = n = 0.0
            for k, (launches, b) in pmc_live.items():
                if any(pfx in k for pfx in prefixes):
                    tot += b * launches
                    n += launches
            return round(tot / n) if n else None
        path = os.path.join(ROOT, PMC_CSV)
        default_workload = (args.schema == "synthetic" and args.model == "HEATNet4" and args.batch == 8 and args.nodes == 10000
                            and args.hidden == 512 and args.dst_mode == "uniform" and args.dropout == 0.0)
        if not os.path.exists(path) or not default_workload:
            return None
        import csv
        tot = n = 0.0
        for r in csv.DictReader(open(path)):
            if any(pfx in r["kernel"] for pfx in prefixes):
                tot += float(r["hbm_MB_per_launch"]) * 1e6 * int(r["launches"])
                n += int(r["launches"])
        return round(tot / n) if n else None
    if not args.no_kernel_timing:
        ops.enable_kernel_timing(True)
        ops.set_background_weight_gradients(False)     # every launch in order on ONE stream while it is bracketed by events: a dW running under
        ksteps = max(3, min(args.steps, 10))           # an attention backward has no duration of its own
        for _ in range(ksteps):
            step()
        torch.cuda.synchronize()
        stats = ops.kernel_timing_summary()
        ops.enable_kernel_timing(False)
        ops.set_background_weight_gradients(args.background_dw == "on")
        ar = stats.get("grad_allreduce")
        if ar is not None:
            allreduce_ms = ar["ms"] / ksteps
        gemm = stats.get("gemm")
        if gemm and gemm["ms"] > 0:
            equiv = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12           # algorithmic (fp32-equivalent) rate
            achieved = gemm["mfma_flops"] / (gemm["ms"] * 1e-3) / 1e12   # what the matrix cores execute
            if args.gemm == "fp32":
                kname, peak, pfx = "wsi::gemm_f32_kernel (v_mfma_f32_32x32x2_f32, NT/NN/TN)", 157.3, ["gemm_f32_kernel"]
            elif args.gemm == "bf16x6":   # six bf16 MFMA products per algorithmic fp32 product, priced against the dense bf16 peak
                kname, peak, pfx = "wsi::gemm_bf16x6_kernel (6x v_mfma_f32_32x32x16_bf16 per fp32 product)", 2500.0, ["gemm_bf16x6"]
            else:                         # fp16x3 / auto: three fp16 products on every large launch (NT / NN: row-scaled, TN: column-scaled), six bf16 products on small ones
                kname, peak, pfx = ("wsi::gemm_fp16x3g_kernel (3x v_mfma_f32_32x32x16_f16 per fp32 product, LDS-DMA staged; Y = XW^T and dX = dY W) + "
                                    "wsi::gemm_tn16_kernel (dW = dY^T X, column-scaled, 3 products), absmax / pack / column-statistics pre-passes included in the time"), 2500.0, ["gemm_fp16x3g", "gemm_fp16x3w", "gemm_tn16", "gemm_bf16x6"]
            # the DOMINANT kernel by time: under fp16x3 / auto the scaled-fp16 projection kernel (Y = X W^T and dX = dY W); the weight-gradient
            # kernel and the aggregate over every projection launch follow as objects of their own
            fam = {k: v for k, v in stats.items() if k.startswith(("gemm_nt_", "gemm_nn_", "gemm_tn_"))}
            dom = None
            if args.gemm in ("fp16x3", "auto"):
                parts = [fam[k] for k in ("gemm_nt_fp16x3", "gemm_nn_fp16x3") if k in fam]
                if parts:
                    dom = {f: sum(p_[f] for p_ in parts) for f in ("ms", "launches", "flops", "mfma_flops")}
            aggregate = {"achieved": round(achieved, 2), "frac": round(achieved / peak, 4), "fp32_equivalent_tflops": round(equiv, 2),
                         "ms_per_step": round(gemm["ms"] / ksteps, 3), "launches_per_step": gemm["launches"] / ksteps,
                         "algorithmic_gflop_per_step": round(gemm["flops"] / ksteps / 1e9, 2), "kernels": kname,
                         "by_family_ms_per_step": {k: round(v["ms"] / ksteps, 3) for k, v in sorted(fam.items())}}
            tn = fam.get("gemm_tn_fp16x3") or fam.get("gemm_tn_bf16x6")
            weight_gradient = None
            if tn and tn["ms"] > 0:
                weight_gradient = {"kernel": ("wsi::gemm_tn16_kernel (dW = dY^T X: column-scaled 2-way fp16 split, 3 products, ds_read_b64_tr_b16 fragments, split-K + reduce; "
                                              "its column-statistics pass over the attention gradients included in the time)" if "gemm_tn_fp16x3" in fam else
                                              "wsi::gemm_bf16x6_kernel<TN> (dW = dY^T X, 6 bf16 products, split-K + reduce)"),
                                   "achieved": round(tn["mfma_flops"] / (tn["ms"] * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": round(tn["mfma_flops"] / (tn["ms"] * 1e-3) / 1e12 / 2500.0, 4),
                                   "fp32_equivalent_tflops": round(tn["flops"] / (tn["ms"] * 1e-3) / 1e12, 2),
                                   "ms_per_step": round(tn["ms"] / ksteps, 3), "avg_launch_ms": round(tn["ms"] / tn["launches"], 4)}
            if dom is not None and dom["ms"] > 0:
                gemm, kname, pfx = dom, ("wsi::gemm_fp16x3g_kernel (3x v_mfma_f32_32x32x16_f16 per fp32 product, LDS-DMA staged; Y = XW^T and dX = dY W; "
                                         "its weight-pack pre-pass included in the time)"), ["gemm_fp16x3g"]
                equiv = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
                achieved = gemm["mfma_flops"] / (gemm["ms"] * 1e-3) / 1e12
            roofline = {"kernel": kname, "bound": "mfma",
                        "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": pmc_traffic(pfx),
                        "weight_gradient_kernel": weight_gradient, "all_projection_launches": aggregate,
                        "measured": "HIP events on the launch stream around every wsi_gemm_grouped call of " + str(ksteps) + " steps run with every launch in order "
                                    "(the background weight gradients of DESIGN 3.8 switched off for this pass: a kernel running under another has no duration of its own)",
                        "fp32_equivalent_tflops": round(equiv, 2),
                        "mfma_flops_per_algorithmic_flop": round(gemm["mfma_flops"] / gemm["flops"], 3),
                        "sustained_mfma_ceiling": (None if args.gemm == "fp32" else
                                                   {"tflops": 1650.0, "frac": round(achieved / 1650.0, 4),
                                                    "note": "what a loop of nothing but v_mfma_f32_32x32x16_bf16 sustains on random operands on this chip "
                                                            "(power-limited; 2470 on zeros): tools/ubench/mfma_rate.hip, profiles/r02_gemm_pmc.md"}),
                        "traffic_source": pmc_source(),
                        "launches_per_step": gemm["launches"] / ksteps,
                        "avg_launch_ms": round(gemm["ms"] / gemm["launches"], 4),
                        "ms_per_step": round(gemm["ms"] / ksteps, 3),
                        "algorithmic_gflop_per_step": round(gemm["flops"] / ksteps / 1e9, 2)}
        attn = stats.get("heat_attn")
        if attn and attn["ms"] > 0:
            nbytes = edge_bytes(n_nodes, n_edges, args.hidden, args.layers)
            gbs = nbytes / (attn["ms"] / ksteps * 1e-3) / 1e9
            # per formulation, with the compulsory bytes of what THAT formulation touches (each tensor once per pass; row = D floats):
            #   full layer      fwd: read K, Q, V + write t, 24 B per edge (index, sim, 4 logits)           = 4 N D 4 + 24 E      
            #                   bwd: read K, Q, V, t-side g_t (2) + write g_K, g_Q, g_V                     = 8 N D 4 + 24 E      
            #   layer under the readout (no V, no t, no g_V):
            #                   fwd: read K, Q, 24 B per edge, the coefficient pass re-reads the logits (4 H B per edge) and writes T H floats per node
            #                        = 2 N D 4 + (24 + 4 H) E + 4 T H N
            #                   bwd: read K, Q (pass 2), Q again (pass 3) + write g_K, g_Q and the residual term r_out; per edge 24 B + the per-(edge, head)
            #                        lookups of pass 1 (8 H B)                                                  = 6 N D 4 + (24 + 8 H) E
            Dh, Hh, Tn = args.hidden, args.heads, len(G.ntypes)
            models_b = {"heat_attn_fwd_full": 4 * n_nodes * Dh * 4 + 24 * n_edges, "heat_attn_bwd_full": 8 * n_nodes * Dh * 4 + 24 * n_edges,
                        "heat_attn_fwd_pooled": 2 * n_nodes * Dh * 4 + (24 + 4 * Hh) * n_edges + 4 * Tn * Hh * n_nodes,
                        "heat_attn_bwd_pooled": 6 * n_nodes * Dh * 4 + (24 + 8 * Hh) * n_edges}
            per_form = {}
            for key, bmodel in models_b.items():
                rec = stats.get(key)
                if rec and rec["ms"] > 0:
                    calls = rec["launches"] / ksteps                     # timed regions per step (one per layer that takes this formulation)
                    layers_ = max(1, round(calls)) if not key.endswith("bwd_pooled") else 1      # (bwd_pooled: two timed regions of ONE layer: table + passes)
                    ms_layer = rec["ms"] / ksteps / layers_
                    g_ = bmodel / (ms_layer * 1e-3) / 1e9
                    per_form[key] = {"layers_per_step": layers_, "ms_per_layer": round(ms_layer, 4), "compulsory_MB": round(bmodel / 1e6, 1),
                                     "achieved_GBps": round(g_, 1), "frac": round(g_ / 8000.0, 4)}
            edge_phase = {"kernel": "wsi::heat_attn_{fwd,bwd_p1,bwd_p2,bwd_p3}", "bound": "hbm",
                          "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                          "algorithmic_bytes_per_edge": round(nbytes / n_edges, 1),
                          "note": "`achieved` / `frac` price every layer with SURVEY 8d's reference-order byte model (6272 B/edge for L = 2) - what the reference's "
                                  "order of operations would have to move; `per_formulation` prices each layer with the bytes of what it actually does "
                                  "(the layer under the readout gathers no V and writes no t)",
                          "per_formulation": per_form,
                          "ms_per_step": round(attn["ms"] / ksteps, 3), "traffic": pmc_traffic(["heat_attn_"]),
                          "traffic_source": pmc_source() + "; fabric-side bytes per attention-kernel launch, Infinity-Cache hits included"}

    # ---- SURVEY 8(d)'s narrower definition of the metric: forward + loss + backward only (no all-reduce, no optimizer)
    def fb_step():
        opt.zero_grad(set_to_none=True)
        l = loss_fn(model(G), labels)
        l.backward()
        return l
    for _ in range(2):
        fb_step()
    sync()
    f0 = time.perf_counter()
    for _ in range(args.steps):
        fb_step()
    sync()
    fdt = time.perf_counter() - f0
    if world > 1:
        ft = torch.tensor([fdt], device=dev, dtype=torch.float64)
        dist.all_reduce(ft, op=dist.ReduceOp.MAX)
        fdt = ft.item()
    fwd_bwd_only = {"value": total_edges * args.steps / fdt, "unit": "edges/s", "ms_per_step": fdt / args.steps * 1e3,
                    "note": "forward + CrossEntropy + backward only (the timed region SURVEY 8d defines); `value` above also "
                            "includes the gradient all-reduce and the Adam step"}

    # ---- the same K steps with the last layer run in the reference's order of operations (DESIGN 3.7 switched off: its output formed for every
    # node, a readout pass over it, V projected, N-deep backward).  Same outputs and gradients to fp32 rounding (tests); reported beside
    # `value` so that both formulations of the same arithmetic are on record.
    full_depth = None
    if args.model in ("HEATNet4", "HEATNet2") and getattr(model, "fuse_readout", False) and not args.no_full_depth:
        model.fuse_readout = False
        ops.set_low_rank_readout_grad(False)
        ops.set_value_collapse(False)
        try:
            for _ in range(max(2, args.warmup)):
                step()
            sync()
            d0 = time.perf_counter()
            for _ in range(args.steps):
                dlast = step()
            sync()
            ddt = time.perf_counter() - d0
            if world > 1:
                dt_t = torch.tensor([ddt], device=dev, dtype=torch.float64)
                dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
                ddt = dt_t.item()
            full_depth = {"value": total_edges * args.steps / ddt, "unit": "edges/s", "ms_per_step": ddt / args.steps * 1e3, "loss": float(dlast.item()),
                          "note": "same step with model.fuse_readout = False, ops.set_low_rank_readout_grad(False), ops.set_value_collapse(False): the last layer's output projection, "
                                  "readout, V projection and their backward at full depth (N rows) instead of on the S = graphs x node-types rows the "
                                  "sum / mean readout reduces them to; identical results to fp32 summation order "
                                  "(tests/test_kernels_gpu.py::test_readout_shortcuts_equal_the_full_depth_path)"}
        finally:
            del model.fuse_readout                     # back to the class default
            ops.set_low_rank_readout_grad(True)
            ops.set_value_collapse(True)
        for _ in range(2):
            step()
        sync()

    # ---- the same K steps in the reference's TRAINING configuration: feat_drop 0.2 drawn on every layer (configs/COAD/HEAT4_kimia_classification_v2.yml;
    # models/HEATNet4.py:77,134).  The dropout sits between the last layer's output projection and the readout, so that layer runs at full depth
    # (DESIGN 3.7 does not apply); the masks are functions of (seed, row, column) applied in the projection epilogues (ops.CounterDropout).
    training_config = None
    if args.model in ("HEATNet4", "HEATNet2") and args.dropout == 0.0 and not args.no_training_config:
        p_ref = 0.2
        for layer in model.gcs:
            layer.drop.p = p_ref
        try:
            for _ in range(max(2, args.warmup)):
                step()
            sync()
            d0 = time.perf_counter()
            for _ in range(args.steps):
                tlast = step()
            sync()
            tdt = time.perf_counter() - d0
            if world > 1:
                dt_t = torch.tensor([tdt], device=dev, dtype=torch.float64)
                dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
                tdt = dt_t.item()
            training_config = {"value": total_edges * args.steps / tdt, "unit": "edges/s", "ms_per_step": tdt / args.steps * 1e3, "feat_drop": p_ref,
                               "loss": float(tlast.item()), "dropout": "counter-based, drawn in the GEMM epilogue and regenerated in the backward (no mask tensors)",
                               "note": "same model / batch / optimizer / timed region as `value` with the reference's training-time dropout "
                                       "(configs/COAD/HEAT4_kimia_classification_v2.yml feat_drop 0.2) drawn on every HEAT layer; the metric's configuration "
                                       "(SURVEY 8d) fixes dropout 0.0, which is what `value` runs"}
        finally:
            for layer in model.gcs:
                layer.drop.p = 0.0
        for _ in range(2):
            step()
        sync()

    # ---- the same K steps in the other GEMM arithmetic (reported beside `value`, never as `value`)
    alts = []
    if world == 1 and not args.no_alt_gemm:
        for other in [m for m in ("bf16x6", "fp16x3", "auto", "fp32") if m != args.gemm]:   # exact fp32 last: its power draw lowers the clocks of a short leg timed right after it
            ops.set_gemm_precision(other)
            for _ in range(max(2, args.warmup)):
                step()
            sync()
            a0 = time.perf_counter()
            for _ in range(args.steps):
                alast = step()
            sync()
            adt = time.perf_counter() - a0
            alts.append({"gemm": other, "value": total_edges * args.steps / adt, "unit": "edges/s", "ms_per_step": adt / args.steps * 1e3,
                         "loss": float(alast.item())})
        ops.set_gemm_precision(args.gemm)
    alt = None
    if alts:
        alt = dict(next(a for a in alts if a["gemm"] == ("bf16x6" if args.gemm == "fp32" else "fp32")))
        alt["note"] = ("same timed region with the GEMM precision argument = %s (exact fp32 MFMA = the reference's arithmetic; the emulated "
                       "modes' error against float64 is <= its own: tests/test_kernels_gpu.py::test_gemm_emulated_error_vs_fp32_mfma, "
                       "test_gemm_fp16x3_scaling_cases)" % alt["gemm"])

    # ---- the reference's own regime: ONE slide per step (trainer/train_gnn.py:48-79).  Such a step is ~100 launches the host takes longer to
    # issue than the GPU to run; trainer.CapturedStep records it into one hipGraph.  Eager and replayed, same model / graph / optimizer.
    # An extra field, never `value`.
    captured = None
    if world == 1 and not args.no_captured and args.schema == "synthetic" and args.model in ("HEATNet4", "HEATNet2"):
        # in a SUBPROCESS: an invalid capture is a segfault on this ROCm, not an exception, and a side measurement must not be able to take
        # the headline line down with it
        import subprocess
        try:
            cmd = [sys.executable, os.path.join(ROOT, "tools", "graph_capture_probe.py"), "--json", "--model", args.model, "--hidden", str(args.hidden),
                   "--nodes", str(args.nodes), "--batch", "1", "--steps", str(3 * args.steps)]
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            rec = json.loads(res.stdout.strip().splitlines()[-1])
            captured = {"workload": f"{args.model}, ONE {args.nodes}-node graph per step (the reference's slide-by-slide regime), fwd + CE + bwd + Adam",
                        "eager_ms_per_step": rec["eager_ms_per_step"], "hipgraph_ms_per_step": rec["hipgraph_ms_per_step"],
                        "edges_per_s_hipgraph": rec["edges"] / (rec["hipgraph_ms_per_step"] * 1e-3), "trajectories_equal": rec["trajectories_equal"],
                        "note": "trainer.CapturedStep: the whole step as ONE hipGraph on a resident graph (tools/graph_capture_probe.py in a subprocess); "
                                "same loss trajectory as eager steps (tests/test_kernels_gpu.py::test_captured_step_replays_the_eager_trajectory)"}
            if not args.no_training_config:
                # the same with the reference's feat_drop 0.2: the masks' seed is a host value + a device word the recorded step advances (new masks per replay)
                res = subprocess.run(cmd + ["--dropout", "0.2"], capture_output=True, text=True, timeout=240)
                rec = json.loads(res.stdout.strip().splitlines()[-1])
                captured["with_feat_drop_0.2"] = {"eager_ms_per_step": rec["eager_ms_per_step"], "hipgraph_ms_per_step": rec["hipgraph_ms_per_step"],
                                                  "note": "eager and replayed steps draw through different seed sequences: times only "
                                                          "(tests/test_kernels_gpu.py::test_captured_step_with_train_mode_dropout_draws_new_masks_every_replay "
                                                          "compares the trajectories under one sequence, bit for bit)"}
        except Exception as exc:
            captured = captured if isinstance(captured, dict) and "eager_ms_per_step" in captured else {"error": repr(exc)}

    # ---- the graphs the reference actually produces: kNN in feature space (8 out-edges per patch, skewed in-degree), in locality
    # order.  An extra field, never `value` (BASELINE's metric is quoted on the uniformly random synthetic graphs above).
    knn = None
    if world == 1 and not args.no_knn and args.schema == "synthetic" and args.model in ("HEATNet4", "HEATNet2"):
        try:
            t_build = time.perf_counter()
            kg = [synthetic.knn_slide(args.nodes, args.in_dim, seed=100 + i, device=dev, n_types=n_types) for i in range(args.batch)]
            t_build = (time.perf_counter() - t_build) / args.batch
            KG = W.batch(kg).to(dev)
            klabels = (torch.arange(args.batch, device=dev) % 2)

            def kstep():
                opt.zero_grad(set_to_none=True)
                l = loss_fn(model(KG), klabels)
                l.backward()
                opt.step()
                return l
            for _ in range(max(3, args.warmup)):
                kstep()
            sync()
            k0 = time.perf_counter()
            for _ in range(args.steps):
                klast = kstep()
            sync()
            kdt = (time.perf_counter() - k0) / args.steps
            ops.enable_kernel_timing(True)
            for _ in range(5):
                kstep()
            kst = ops.kernel_timing_summary()
            ops.enable_kernel_timing(False)
            kp = KG.plan()
            ns_, rp_ = kp.node_seg.long(), kp.rowptr.long()
            knn = {"workload": f"{args.batch} WSI-like slides: {args.nodes} patches, {args.in_dim}-d features in 40 clusters, exact 8-NN edges typed by "
                               "Pearson sign (construct.construct_graph = graph_constructor.py:256-303), locality order (graph.apply_locality_order)",
                   "ms_per_step": round(kdt * 1e3, 4), "edges": KG.num_edges(), "edges_per_s": KG.num_edges() / kdt,
                   "edge_phase_ms": round(kst["heat_attn"]["ms"] / 5, 4), "gemm_ms": round(kst["gemm"]["ms"] / 5, 4),
                   "max_in_degree": int((rp_[ns_[1:]] - rp_[ns_[:-1]]).max()), "locality_plan": bool(kp.locality),
                   "build_s_per_slide": round(t_build, 3), "loss": float(klast.item()),
                   "fabric_bytes_per_edge_and_layer": {"value": 5400, "as_constructed": 13100,
                                                       "source": "profiles/r02_locality_traffic_locality.csv / _raw.csv (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over the four "
                                                                 "attention kernels, tools/pmc_locality.sh); not measured by this run"},
                   "note": "same model / optimizer / timed region as `value`, on graphs with the reference's real structure; reported beside `value`, never as it"}
            del KG, kg
        except Exception as e:                             # the headline must not depend on this leg
            knn = {"error": f"{type(e).__name__}: {e}"}

    # ---- PCIe-inclusive leg: every step consumes a fresh batch assembled host->device by the prefetching loader
    pcie = None
    if args.pcie:
        from wsi_hgnn_amd.data import GraphBatchLoader
        pool = [synthetic.hetero_graph(args.nodes, args.in_dim, seed=7000 + 1000 * rank + i, dst_mode=args.dst_mode)
                for i in range(2 * args.batch)]
        pcie = {}
        for mode, resident in (("pinned_host", False), ("hbm_resident", True)):
          loader = GraphBatchLoader(pool, [i % 2 for i in range(len(pool))], args.batch, dev, shuffle=True, drop_last=True,
                                    resident=resident, passes=4)          # (8 batches per iterator: the start-up transfer of an iterator, which nothing hides, is paid once in 8 steps)

          host = {"loader": 0.0, "step": 0.0}

          def feed(nsteps):
            done = 0
            edges = 0
            while done < nsteps:
                it = iter(loader)
                while True:
                    h0 = time.perf_counter()
                    try:
                        Gb, yb = next(it)                 # (assembles the NEXT batch before handing this one over: host time of the loader)
                    except StopIteration:
                        break
                    h1 = time.perf_counter()
                    opt.zero_grad(set_to_none=True)
                    l = loss_fn(model(Gb), yb)
                    hf = time.perf_counter()
                    bucket.arm()
                    l.backward()
                    hb = time.perf_counter()
                    bucket.all_reduce_mean()
                    opt.step()
                    h2 = time.perf_counter()
                    host["loader"] += h1 - h0
                    host["step"] += h2 - h1
                    host["forward"] = host.get("forward", 0.0) + hf - h1
                    host["backward"] = host.get("backward", 0.0) + hb - hf
                    edges += Gb.num_edges()
                    done += 1
                    if done >= nsteps:
                        break
            return edges
          feed(8)          # two passes over the pool: freshly pinned host pages are slow on their first transfers
          sync()
          host["loader"] = host["step"] = host["forward"] = host["backward"] = 0.0
          p0 = time.perf_counter()
          pe = feed(args.steps)
          sync()
          pdt = time.perf_counter() - p0
          pcie[mode] = {"value": pe / pdt, "unit": "edges/s (this rank)", "ms_per_step": pdt / args.steps * 1e3,
                        "host_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in host.items()}}
        pcie["note"] = ("every step consumes a NEW shuffled batch from the loader (assembly + kernel-plan build included): pinned_host = "
                        "features cross PCIe each step on a side stream; hbm_resident = data set uploaded once, batches assembled D2D")

    # ---- CPU baseline: the oracle (pure-PyTorch restatement of the reference; DGL is unavailable) on a bounded sample
    # one graph of the workload, fwd+loss+bwd, median of 10 after 2 warm-ups, at k = 8 threads and k = all physical
    # cores; CPU model and core counts stated.
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import statistics
        from oracle import models as OM
        torch.manual_seed(611)
        o = getattr(OM, args.model)(args.in_dim, args.hidden, 2, args.layers, args.heads, nd, 0.0, "mean")
        o.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
        g1 = (synthetic.real_schema_graph if args.schema == "real" else synthetic.hetero_graph)(args.nodes, args.in_dim, seed=611, dst_mode=args.dst_mode)
        y1 = torch.tensor([0])
        model_name, phys, logical = cpu_info()
        restore = torch.get_num_threads()
        runs = {}

        def cpu_step():
            for p in o.parameters():
                p.grad = None
            ce(o(g1), y1).backward()
        for cores in sorted({min(8, phys), phys}):
            torch.set_num_threads(cores)
            for _ in range(2):
                cpu_step()
            ts = []
            for _ in range(10):
                c0 = time.perf_counter()
                cpu_step()
                ts.append(time.perf_counter() - c0)
            runs[cores] = statistics.median(ts)
        torch.set_num_threads(restore)
        cores = min(runs, key=runs.get)
        cdt = runs[cores]
        cpu_baseline = {"value": round(g1.num_edges() / cdt, 1), "unit": "edges/s", "cores": cores, "kind": "port",
                        "sample": f"1 graph ({args.nodes} nodes, {g1.num_edges()} edges) fwd+loss+bwd, median of 10 after 2 warm-ups per "
                                  f"thread count; s/graph by threads: { {k: round(v, 3) for k, v in runs.items()} }; value = the faster one; "
                                  f"CPU restatement of the reference (oracle/models.py; DGL unavailable)",
                        "cpu_model": model_name, "physical_cores": phys, "logical_cpus": logical, "torch": torch.__version__,
                        "s_per_graph": round(cdt, 3),
                        "edges_per_s_by_threads": {str(k): round(g1.num_edges() / v, 1) for k, v in runs.items()}}

    wm = whole_model_bytes(n_nodes, n_edges, args.in_dim, args.hidden, args.layers)
    hbm_roof = n_edges / (wm / 8.0e12)                # edges/s one GPU could sustain if the step only moved its compulsory bytes
    hbm_roofline = {"bytes_per_edge": round(wm / n_edges, 1), "roofline_edges_per_s_per_gpu": round(hbm_roof),
                    "frac": round(value / world / hbm_roof, 4),
                    "note": "whole-step compulsory-traffic model at 8 TB/s (SURVEY 8d; the north star's '40 % of the HBM roofline'). "
                            "The projections (1174 GFLOP/step in the reference's order of operations"
                            + (f"; {roofline['all_projection_launches']['algorithmic_gflop_per_step']:.0f} executed: the last layer's V and output projections are folded into the sum / mean readout" if roofline else "")
                            + ") need >= 7.5 ms at the 157.3 TFLOP/s fp32 matrix peak, >= 1.5 ms as 3 fp16 products (forward, dX) / 6 bf16 "
                            "products (dW) at the 2.5 PFLOP/s dense peak and >= 2.3 ms at the 1.65 PFLOP/s a pure-MFMA loop sustains on random "
                            "operands, tools/ubench/mfma_rate.hip) vs 1.7 ms of HBM time: the step is matrix-bound, not HBM-bound"}
    if rank == 0:
        line = {
            "metric": "edges/s fwd+bwd HEATNet4, 10k-node/6-rel synth graph, 1->8 MI355X" if (args.model == "HEATNet4" and args.schema == "synthetic" and args.slide_sizes == "equal") else f"edges/s fwd+bwd {args.model} (side measurement, not the BASELINE metric)",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x6": "f32 (bf16x6 emulation, fp32-class error)",
                      "fp16x3": "f32 (fp16x3 emulation: row- / column-scaled 2-way fp16 split, 3 products; fp32-class error)",
                      "auto": "f32 (fp16x3 emulation: 2-way fp16 split scaled per row (Y = XW^T, dX) or per column (dW), 3 products - bf16x6 for small launches; fp32-class error)"}[args.gemm],
            "data": "synthetic",
            "config": {"workload": f"{args.model} fwd+loss+bwd+grad-allreduce+Adam, batch of {args.batch} synthetic hetero graphs per GPU "
                                   f"({args.nodes} nodes, {len(G.ntypes)} node types, {len(G.canonical_etypes)} relations, {n_edges // args.batch} edges each, {args.in_dim}-d features, "
                                   f"hidden {args.hidden}, {args.layers} layers, {args.heads} heads), dst={args.dst_mode}",
                       "graphs_per_gpu": args.batch, "nodes_per_graph": args.nodes, "edges_per_gpu_step": n_edges,
                       "schema": args.schema, "relations": len(G.canonical_etypes), "node_types": len(G.ntypes),
                       "parallelism": f"dp{world} (WSI-sharded, flat fp32 grad all-reduce over RCCL)",
                       "includes_optimizer_step": True},
            "ranks": (dist.get_world_size() if world > 1 else 1), "collective_backend": (dist.get_backend() if world > 1 else None),
            "grad_allreduce": {"ms_per_step": (round(allreduce_ms, 4) if allreduce_ms is not None else None),
                               "bytes": bucket._buf.numel() * 4, "flag_readbacks": bucket.flag_readbacks,
                               "pieces": len(bucket._piece_lo), "pieces_launched_during_backward": bucket.overlapped_pieces,
                               "overlap": bool(args.dp_overlap),
                               "NCCL_ALGO": os.environ.get("NCCL_ALGO", "(unset: RCCL's choice)"), "NCCL_PROTO": os.environ.get("NCCL_PROTO", "(unset)"),
                               "note": "one flat fp32 buffer per step (dist.GradBucket), all-reduced in `pieces` contiguous parts launched from "
                                       "autograd hooks while backward runs (the part with the first parameters and the used-flags goes last); "
                                       "ms_per_step = what is left after backward: HIP events on the launch stream of rank 0"},
            "shard_balance": shard_balance,
            "loss": float(last.item()),
            "ms_per_step_median": round(ms_median, 4),      # SURVEY 8d asks for the median: GPU time between per-step marks on rank 0 (`ms_per_step` is the contract's wall-clock mean)
            "roofline": roofline,
            "edge_phase_roofline": edge_phase,
            "hbm_roofline": hbm_roofline,
            "cpu_baseline": cpu_baseline,
            "fwd_bwd_only": fwd_bwd_only,
            "single_graph_step": captured,
            "knn_locality": knn,
            "full_depth_last_layer": full_depth,
            "training_config": training_config,
            "alt_gemm": alt,
            "other_gemm_modes": alts or None,
            "pcie_inclusive": pcie,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
