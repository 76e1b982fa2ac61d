"""RCCL beside the caller's stream and the side stream on ONE GPU (the fourth-hardware-queue question that rested on an argument).
A one-rank "nccl" process group; the bench batch (8 x 10k nodes, HEATNet4 hidden 512) stepped three ways:
  plain     no bucket (side stream carries the background weight gradients and the column statistics)
  overlap   GradBucket(single_rank_collectives=True): 4 pieces all-reduced from autograd's hooks on RCCL's stream while backward runs
  blocking  the same bucket with overlap=False: ONE all-reduce of the 34 MB buffer behind backward
-> ms per step (median of device-event pairs) and the final parameters' equality.   python tools/rccl_one_rank.py [--json out.json]"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from wsi_hgnn_amd import models, synthetic, ops
    from wsi_hgnn_amd.dist import GradBucket
    from wsi_hgnn_amd.optim import Adam
    from wsi_hgnn_amd.trainer import apply_loss
    G, y = synthetic.hetero_batch(8, 10000, 1024)
    G, y = G.to(dev), y.to(dev)
    ce = torch.nn.CrossEntropyLoss()
    ops.set_gemm_precision("auto")
    out = {}
    finals = {}
    for name, kw in (("plain", None), ("overlap", dict(overlap=True, single_rank_collectives=True)), ("blocking", dict(overlap=False, single_rank_collectives=True))):
        torch.manual_seed(611)
        m = models.HEATNet4(1024, 512, 2, 2, 4, {"0": 0, "1": 1, "2": 2}, 0.0, "mean").to(dev)
        m.train()
        apply_loss(ce, m(G), y).backward()
        bucket = GradBucket.from_model(m, **kw) if kw else None
        opt = Adam([p for p in m.parameters() if p.grad is not None], lr=1e-5, weight_decay=5e-3)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = apply_loss(ce, m(G), y)
            if bucket is not None:
                bucket.arm()
            loss.backward()
            if bucket is not None:
                bucket.all_reduce_mean()
            opt.step()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        side0 = ops._BACKGROUND["launches"]
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        marks[0].record()
        for i in range(args.steps):
            step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        ts = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        out[name] = {"ms_per_step_median": round(statistics.median(ts), 4), "ms_per_step_mean": round(sum(ts) / len(ts), 4),
                     "side_stream_weight_gradient_launches": ops._BACKGROUND["launches"] - side0,
                     "pieces": len(bucket._piece_lo) if bucket else 0, "pieces_sent_from_hooks": bucket.overlapped_pieces if bucket else 0,
                     "bucket_MB": round(bucket._buf.numel() * 4 / 1e6, 1) if bucket else 0}
        finals[name] = [p.detach().clone() for p in m.parameters()]
        print(name, out[name], flush=True)
    out["final_parameters_bit_equal"] = {k: all(torch.equal(a, b) for a, b in zip(finals[k], finals["plain"])) for k in ("overlap", "blocking")}
    out["note"] = ("one MI355X, one-rank nccl group: RCCL's stream is a THIRD stream beside the caller's and the module's one side stream; an armed bucket keeps "
                   "every launch of backward on the caller's stream (the side stream idles), so 'overlap' also prices the weight gradients coming back in order")
    print(json.dumps(out["final_parameters_bit_equal"]))
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
