"""The data-parallel leg on a 1-GPU box (`-m gpu`): two gloo ranks sharing cuda:0 run the HIP model, and bench.py's own
launcher is exercised.  RCCL itself needs >= 2 GPUs: `test_two_nccl_ranks_match_union_batch` runs the same check over backend
"nccl" with one GPU per rank and is skipped below two devices, so a multi-GPU lease verifies the RCCL path by itself; everything
above the collective backend — sharding, the flat bucket with its used flags, train_one_step, bench.py's rank launch and refusal
to misreport — is the same code on both backends."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ND = {"0": 0, "1": 1, "2": 2}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    from wsi_hgnn_amd import models
    torch.manual_seed(611)
    return models.HEATNet4(32, 128, 2, 2, 4, ND, 0.0, "mean").to(dev)


def _graphs():
    from wsi_hgnn_amd import synthetic
    return [synthetic.hetero_graph(200, 32, seed=10 + i, dst_mode="hub") for i in range(4)]


def _worker(rank, world, port, out_dir, backend="gloo", overlap="1", background_min_flop=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo: both ranks share cuda:0 (a 1-GPU box); nccl (= RCCL): one GPU per rank
    dev = torch.device("cuda:0" if backend == "gloo" else f"cuda:{rank}")
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
    try:
        import wsi_hgnn_amd as W
        from wsi_hgnn_amd.dist import GradBucket, shard
        from wsi_hgnn_amd.trainer import train_one_step
        from wsi_hgnn_amd import ops
        if background_min_flop is not None:               # let this test's small dW launches qualify for the side stream (ops._gemm_tn_background)
            ops._BACKGROUND["min_flop"] = float(background_min_flop)
        m = _model(dev)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-3)
        bucket = GradBucket.from_model(m, overlap=(overlap == "1"))
        gs = _graphs()
        labels = torch.tensor([0, 1, 1, 0])
        mine = shard(list(range(4)), rank, world)
        loss, *_ = train_one_step(m, opt, torch.nn.CrossEntropyLoss(), tuple(gs[i] for i in mine), labels[mine], dev, bucket=bucket, sync=True)
        torch.cuda.synchronize()
        live = [n for n, p in m.named_parameters() if any(p is q for q in bucket.params)]
        torch.save({"params": {k: v.detach().cpu() for k, v in m.state_dict().items()}, "flat": bucket.flat.cpu(), "live": live,
                    "readbacks": bucket.flag_readbacks, "loss": loss, "overlapped": bucket.overlapped_pieces,
                    "pieces": len(bucket._piece_lo), "background_launches": ops._BACKGROUND["launches"],
                    "background_blocked_after": ops._BACKGROUND["blocked"]}, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_of_the_hip_model_match_the_union_batch():
    """SURVEY 8e determinism check with the PRODUCT model: after one train_one_step on two ranks (2 graphs each) the
    parameters of both ranks are identical and equal those of one process stepping on the 4-graph union batch."""
    from wsi_hgnn_amd.trainer import train_one_step
    res = _run_two("gloo", "1")
    assert torch.equal(res[0]["flat"], res[1]["flat"])
    assert res[0]["readbacks"] == 0 and res[1]["readbacks"] == 0
    for r in res:                       # train_one_step arms the bucket: all pieces but the last went out while backward was running
        assert r["pieces"] >= 3 and r["overlapped"] == r["pieces"] - 1
    for k in res[0]["params"]:
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k
    # the averaged gradient both ranks stepped with == the gradient of one process on the 4-graph union batch
    import wsi_hgnn_amd as W
    dev = torch.device("cuda:0")
    m = _model(dev)
    G = W.batch(_graphs()).to(dev)
    torch.nn.functional.cross_entropy(m(G), torch.tensor([0, 1, 1, 0], device=dev)).backward()
    named = dict(m.named_parameters())
    ref = torch.cat([named[n].grad.reshape(-1) for n in res[0]["live"]]).cpu()
    err = (res[0]["flat"] - ref).abs().max().item()
    assert err <= 1e-7 + 1e-5 * ref.abs().max().item(), err
    # and one optimizer step from it lands within a small fraction of the step size (lr = 1e-3) of the single-process step
    m2 = _model(dev)
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-3)
    train_one_step(m2, opt, torch.nn.CrossEntropyLoss(), tuple(_graphs()), torch.tensor([0, 1, 1, 0]), dev, sync=True)
    for k, v in m2.state_dict().items():
        assert (res[0]["params"][k] - v.detach().cpu()).abs().max().item() <= 5e-5, k


def test_armed_bucket_keeps_weight_gradients_off_the_side_stream():
    """dist.GradBucket.arm() packs gradients from hooks DURING backward: a weight gradient written by the side stream of ops._gemm_tn_background
     would be packed and reduced before it exists.  With the size gate of that path opened (min_flop = 0: every dW launch of this small
    model would qualify) two ranks with the overlapped all-reduce must produce, bit for bit, what the blocking path produces, must not have used
    the side stream while armed, and must leave the path unblocked afterwards; the blocking path (no hooks during backward) DOES use it."""
    armed = _run_two("gloo", "1", background_min_flop=0.0)
    blocking = _run_two("gloo", "0", background_min_flop=0.0)
    for r in armed:
        assert r["overlapped"] == r["pieces"] - 1 and r["background_launches"] == 0 and not r["background_blocked_after"]
    for r in blocking:
        assert r["overlapped"] == 0 and r["background_launches"] > 0
    assert torch.equal(armed[0]["flat"], armed[1]["flat"]) and torch.equal(armed[0]["flat"], blocking[0]["flat"])
    for k in armed[0]["params"]:
        assert torch.equal(armed[0]["params"][k], blocking[0]["params"][k]), k


def _one_rank_worker(port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import wsi_hgnn_amd as W
        from wsi_hgnn_amd.dist import GradBucket
        from wsi_hgnn_amd.trainer import train_one_step
        from wsi_hgnn_amd import ops
        ops._BACKGROUND["min_flop"] = 0.0                 # this model's dW launches would all qualify for the side stream
        gs = _graphs()
        labels = torch.tensor([0, 1, 1, 0])
        out = {}
        for name, kw in (("plain", None), ("overlap", dict(overlap=True, single_rank_collectives=True)),
                         ("blocking", dict(overlap=False, single_rank_collectives=True))):
            m = _model(dev)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-3)
            bucket = GradBucket.from_model(m, **kw) if kw is not None else None
            before = ops._BACKGROUND["launches"]
            losses = []
            for _ in range(3):
                loss, *_ = train_one_step(m, opt, torch.nn.CrossEntropyLoss(), tuple(gs), labels, dev, bucket=bucket, sync=True)
                losses.append(loss)
            torch.cuda.synchronize()
            out[name] = {"params": {k: v.detach().cpu() for k, v in m.state_dict().items()}, "losses": losses,
                         "grads": {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None},
                         "overlapped": bucket.overlapped_pieces if bucket else 0, "pieces": len(bucket._piece_lo) if bucket else 0,
                         "side_launches": ops._BACKGROUND["launches"] - before, "blocked_after": ops._BACKGROUND["blocked"]}
        torch.save(out, os.path.join(out_dir, "one_rank.pt"))
    finally:
        dist.destroy_process_group()


def test_one_rccl_rank_carries_the_bucket():
    """RCCL on the one GPU there is: a one-rank "nccl" process group with the bucket's whole path switched on (GradBucket(single_rank_collectives=True)):
    the pieces go out from autograd's hooks on RCCL's stream beside the caller's stream, the flag piece and the re-pointing of ``.grad`` run, the
    side stream stays unused while armed.  A sum over one rank is the value itself, so three optimizer steps must end bit-identical to the plain
    (bucket-less) steps - overlapped and blocking alike.  (What this cannot show - a ring over xGMI - needs the second GPU of
    test_two_nccl_ranks_match_union_batch.)"""
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as out_dir:
        p = ctx.Process(target=_one_rank_worker, args=(_free_port(), out_dir))
        p.start()
        p.join(timeout=900)
        assert (not p.is_alive()) and p.exitcode == 0, p.exitcode
        res = torch.load(os.path.join(out_dir, "one_rank.pt"))
    plain, ov, bl = res["plain"], res["overlap"], res["blocking"]
    assert ov["pieces"] >= 3 and ov["overlapped"] == 3 * (ov["pieces"] - 1) and bl["overlapped"] == 0
    assert ov["side_launches"] == 0 and not ov["blocked_after"] and plain["side_launches"] > 0
    for other in (ov, bl):
        assert other["losses"] == plain["losses"]
        for k in plain["params"]:
            assert torch.equal(other["params"][k], plain["params"][k]), k
        for k in plain["grads"]:
            assert torch.equal(other["grads"][k], plain["grads"][k]), k


def _run_two(backend, overlap, background_min_flop=None):
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as out_dir:
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out_dir, backend, overlap, background_min_flop)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=900)
        assert all((not p.is_alive()) and p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        return [torch.load(os.path.join(out_dir, f"rank{r}.pt")) for r in range(2)]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: runs on a box with >= 2 GPUs")
def test_two_nccl_ranks_match_union_batch():
    """The same determinism check over RCCL (backend "nccl"), one GPU per rank, with the overlapped gradient all-reduce on: the
    asynchronous pieces launched from autograd's worker thread must (1) actually be launched during backward, (2) give both ranks
    bit-identical averaged gradients and parameters, (3) equal - bit for bit - what the single blocking collective
    (GradBucket(overlap=False)) produces, and (4) match the gradient of one process on the 4-graph union batch.  Skipped on a 1-GPU box;
    an 8-GPU lease verifies the RCCL path by running it."""
    res = _run_two("nccl", "1")
    blocking = _run_two("nccl", "0")
    assert torch.equal(res[0]["flat"], res[1]["flat"])
    for r in res:
        assert r["pieces"] >= 3 and r["overlapped"] > 0 and r["overlapped"] == r["pieces"] - 1
    for r in blocking:
        assert r["overlapped"] == 0
    assert torch.equal(res[0]["flat"], blocking[0]["flat"])                      # overlap changes when the sums happen, not the sums
    for k in res[0]["params"]:
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k
        assert torch.equal(res[0]["params"][k], blocking[0]["params"][k]), k
    import wsi_hgnn_amd as W
    dev = torch.device("cuda:0")
    m = _model(dev)
    G = W.batch(_graphs()).to(dev)
    torch.nn.functional.cross_entropy(m(G), torch.tensor([0, 1, 1, 0], device=dev)).backward()
    named = dict(m.named_parameters())
    ref = torch.cat([named[n].grad.reshape(-1) for n in res[0]["live"]]).cpu()
    err = (res[0]["flat"] - ref).abs().max().item()
    assert err <= 1e-7 + 1e-5 * ref.abs().max().item(), err


SMALL = ["--steps", "2", "--warmup", "1", "--batch", "2", "--nodes", "600", "--in-dim", "64", "--hidden", "128",
         "--no-cpu-baseline", "--no-alt-gemm", "--no-knn", "--no-captured"]


def _run_bench(extra_args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bench_refuses_more_gpus_than_the_box_has():
    n = torch.cuda.device_count()
    r = _run_bench(["--gpus", str(n + 1)] + SMALL)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not r.stdout.strip()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts 2 ranks itself (here: both on cuda:0 over gloo, the
    documented dry-run knobs) and reports n_gpus = ranks = 2 with the all-reduce timed."""
    r = _run_bench(["--gpus", "2", "--one-device", "--backend", "gloo"] + SMALL, {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["collective_backend"] == "gloo"
    assert out["grad_allreduce"]["ms_per_step"] is not None and out["grad_allreduce"]["flag_readbacks"] == 0
    assert out["grad_allreduce"]["pieces_launched_during_backward"] > 0
    assert out["value"] > 0 and out["scaling"] == "weak"
    # single rank through the same entry point
    r1 = _run_bench(["--gpus", "1"] + SMALL)
    assert r1.returncode == 0, r1.stderr[-2000:]
    o1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert o1["n_gpus"] == 1 and o1["ranks"] == 1
    assert {"roofline", "cpu_baseline", "metric", "unit", "ms_per_step", "config"} <= set(o1)
    # a launcher that started a different number of ranks than --gpus is refused as well
    r2 = _run_bench(["--gpus", "4"] + SMALL, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r2.returncode != 0 and "misreport" in r2.stderr


def test_bench_dry_run_at_the_world_size_of_the_target_node():
    """`python bench.py --gpus 8` exactly as the driver launches it on an 8-GPU node - 8 ranks under torch.distributed.run, the flat gradient bucket
    cut in pieces that leave from autograd hooks, the used flags, max-over-ranks timing, one JSON line from rank 0 - with the two dry-run arguments
    that put every rank on cuda:0 and carry the collective over gloo (this box has one GPU): the code path of configs[3] at its real world size,
    short of RCCL itself."""
    r = _run_bench(["--gpus", "8", "--one-device", "--backend", "gloo"] + SMALL + ["--no-training-config", "--no-full-depth", "--no-kernel-timing"], {}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["collective_backend"] == "gloo" and out["scaling"] == "weak"
    assert out["grad_allreduce"]["flag_readbacks"] == 0 and out["grad_allreduce"]["pieces_launched_during_backward"] > 0
    assert out["value"] > 0 and out["config"]["graphs_per_gpu"] == 2 and out["config"]["parallelism"].startswith("dp8")

