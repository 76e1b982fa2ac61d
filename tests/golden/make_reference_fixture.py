#!/usr/bin/env python
"""Generate tests/golden/linear_attention_block.npz by EXECUTING the reference's own code.

The only fragment of the reference hot path that runs without DGL is ``LinearAttentionBlock``
(/root/reference/models/HEATNet4.py lines 20-42, pure torch).  This script (run in the build
container, where /root/reference exists) execs exactly those source lines from the reference file,
runs the class on seeded inputs and stores inputs, weight, output and gradients.  Only DATA is
committed; no reference source text is stored.  Re-run: ``python tests/golden/make_reference_fixture.py``.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/models/HEATNet4.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linear_attention_block.npz")


def main():
    lines = open(REF).read().splitlines()
    src = "\n".join(lines[19:42])          # file lines 20..42: class LinearAttentionBlock
    ns = {}
    exec("import torch\nimport torch.nn as nn\nimport torch.nn.functional as F\n" + src, ns)
    Block = ns["LinearAttentionBlock"]
    torch.manual_seed(611)
    blk = Block(in_features=256, normalize_attn=True)
    l = torch.randn(5, 256, requires_grad=True)
    g = torch.randn(5, 256)
    out = blk(l, g)
    gout = torch.randn(5, 256)
    out.backward(gout)
    np.savez(OUT, weight=blk.op.weight.detach().numpy(), l=l.detach().numpy(), g=g.detach().numpy(),
             out=out.detach().numpy(), gout=gout.numpy(), grad_l=l.grad.numpy(), grad_weight=blk.op.weight.grad.numpy())
    print("wrote", OUT, "max|out-l| =", float((out - l).abs().max()), "max|grad_w| =", float(blk.op.weight.grad.abs().max()))


if __name__ == "__main__":
    sys.exit(main())
