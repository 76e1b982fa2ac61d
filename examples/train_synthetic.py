#!/usr/bin/env python
"""End-to-end use of the package the way the reference's ``main.py`` -> ``GNNTrainer`` uses its own modules
(trainer/train_gnn.py:19-120), on synthetic WSI-shaped graphs:

  graph files (io.save_graph / load_graph)  ->  GraphBatchLoader (replaces GraphDataLoader + g.to(device))  ->
  HEATNet4 + Adam + CrossEntropy via trainer.train_one_step  ->  CheckpointStore (reference file layout)  ->  io.evaluate.

Run on one GPU:            python examples/train_synthetic.py --epochs 2
Run data-parallel on N:    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_synthetic.py
"""
import argparse
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_hgnn_amd import data, dist, io, models, synthetic, trainer  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=16)
    ap.add_argument("--nodes", type=int, default=2000)
    ap.add_argument("--in-dim", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--workdir", default=None)
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    work = args.workdir or tempfile.mkdtemp(prefix="wsi_example_")
    os.makedirs(os.path.join(work, "graphs"), exist_ok=True)
    # 1. "data set": flat safetensors graph files named like TCGA slides (label = tumour vs normal from the barcode list)
    paths, normal = [], []
    for i in range(args.graphs):
        name = f"TCGA-AA-{i:04d}-01Z-00-DX1"
        p = os.path.join(work, "graphs", name + ".safetensors")
        if rank == 0 and not os.path.exists(p):
            io.save_graph(p, synthetic.hetero_graph(args.nodes, args.in_dim, seed=100 + i, dst_mode="hub"))
        paths.append(p)
        if i % 2 == 0:
            normal.append(name[:16])
    if world > 1:
        td.barrier(device_ids=[local])
    mine = dist.shard(paths, rank, world)                                   # WSI-sharded data parallelism
    graphs = [io.load_graph(p) for p in mine]
    labels = [io.label_tumour_vs_normal(p, normal) for p in mine]
    loader = data.GraphBatchLoader(graphs, labels, args.batch, dev, shuffle=True, drop_last=False, seed=611 + rank)

    # 2. model / optimizer / loss exactly as parser.py builds them (Adam lr 1e-5 wd 5e-3; CrossEntropyLoss)
    nd = {"0": 0, "1": 1, "2": 2}
    torch.manual_seed(611)
    gnn = models.HEATNet4(args.in_dim, args.hidden, 2, 2, 4, nd, args.dropout, "mean").to(dev)
    opt = torch.optim.Adam(gnn.parameters(), lr=1e-5, weight_decay=5e-3)
    loss_fn = torch.nn.CrossEntropyLoss()
    bucket = dist.GradBucket.from_model(gnn) if world > 1 else None         # every parameter the architecture reaches, with used flags
    store = io.CheckpointStore(os.path.join(work, "ckpt"))

    # 3. epochs
    for epoch in range(args.epochs):
        gnn.train()
        tot, n = 0.0, 0
        for G, y in loader:
            loss, acc, *_ = trainer.train_one_step(gnn, opt, loss_fn, G, y, dev, bucket=bucket, sync=True)
            tot, n = tot + loss, n + 1
        gnn.eval()
        metrics = io.evaluate(gnn, loader)
        if rank == 0:
            print(f"epoch {epoch}: train loss {tot / max(n, 1):.4f}  eval {metrics}")
            store.save_model(gnn.state_dict(), version=epoch + 1, stats={"epoch": epoch, "loss": tot / max(n, 1), **metrics})
    if rank == 0:
        sd = store.load_model()
        gnn.load_state_dict(sd)
        print("checkpoint reloaded from", store.model_file(store.load_version()))
    if world > 1:
        td.destroy_process_group()
    return work


if __name__ == "__main__":
    main()
