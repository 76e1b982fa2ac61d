"""GPU tests of the pre-split ("P3" planes) GEMM path of csrc/gemm_p3.hip through the C-ABI: the split itself (exact), the
NT and TN kernels against fp64 references with the tolerances of the fp32 MFMA path, fused epilogues, plane outputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _relerr(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()


@pytest.mark.parametrize("R,C", [(300, 512), (77, 200), (5, 16), (129, 33), (1, 1)])
def test_split_planes_is_exact(R, C):
    from wsi_hgnn_amd import ops
    torch.manual_seed(R + C)
    x = (torch.randn(R, C, device=_dev()) * torch.exp(4 * torch.randn(R, C, device=_dev()))).contiguous()
    p = ops.split_planes(x)
    assert p.shape == (R, ops.planes_ld(C))
    assert torch.equal(ops.planes_to_float(p, C), x)                      # x0 + x1 + x2 == x bit for bit
    nb = (C + 15) // 16
    pad = p.reshape(R, nb, 3, 16).float()[:, -1, :, (C - 1) % 16 + 1:]
    assert float(pad.abs().sum()) == 0.0                                  # columns beyond C are zero
    pt = ops.split_planes(x, transpose=True)
    assert torch.equal(ops.planes_to_float(pt, R), x.t())
    # a strided view (column slice of a wider table) and a column-block destination
    wide = torch.randn(R, C + 64, device=_dev())
    pv = ops.split_planes(wide[:, 32:32 + C])
    assert torch.equal(ops.planes_to_float(pv, C), wide[:, 32:32 + C])
    out = torch.zeros(R, ops.planes_ld(C + 32), dtype=torch.bfloat16, device=_dev())
    ops.split_planes(x, out=out, col_block=2)
    assert torch.equal(ops.planes_to_float(out, C + 32)[:, 32:], x)


def _nt(A, B, epilogue=0, bias=None, R=None, gate=None, Mm=None, want_planes=False, want_c=True, C_init=None):
    from wsi_hgnn_amd import ops, _native as N
    M, K = A.shape
    Nn = B.shape[0]
    Ap, Bp = ops.split_planes(A), ops.split_planes(B)
    C = (C_init.clone() if C_init is not None else torch.empty(M, Nn, device=A.device)) if want_c else None
    Cp = ops.empty_planes(M, Nn, A.device).zero_() if want_planes else None
    ops.gemm_p3(N.WSI_GEMM_NT, epilogue, [dict(Ap=N.ptr(Ap), ldap=Ap.stride(0), Bp=N.ptr(Bp), ldbp=Bp.stride(0),
                                               C=N.ptr(C), ldc=Nn, Cp=N.ptr(Cp), ldcp=(Cp.stride(0) if Cp is not None else 0),
                                               bias=N.ptr(bias), R=N.ptr(R), ldr=(R.stride(0) if R is not None else 0), gate=N.ptr(gate),
                                               Mm=N.ptr(Mm), ldm=(Mm.stride(0) if Mm is not None else 0), M=M, N=Nn, K=K)], A.device)
    return C, Cp


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 64), (1000, 512, 1024), (37, 5, 20), (8, 2, 64),
                                   (130, 129, 33), (1, 1, 1), (513, 200, 200), (2500, 1536, 512)])
def test_gemm_p3_nt(M, N, K):
    from wsi_hgnn_amd import _native as Nn_
    torch.manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device=_dev())
    B = torch.randn(N, K, device=_dev())
    bias = torch.randn(N, device=_dev())
    C, Cp = _nt(A, B, Nn_.WSI_EPI_BIAS, bias=bias, want_planes=True)
    ref = A.double().cpu() @ B.double().cpu().t() + bias.double().cpu()
    assert _relerr(C, ref) < 2e-6, (M, N, K, _relerr(C, ref))
    from wsi_hgnn_amd import ops
    assert torch.equal(ops.planes_to_float(Cp, N), C)                      # the plane output IS the fp32 output, split exactly


def test_gemm_p3_nt_asymmetric_layout_and_planes_only():
    """A = I with an asymmetric B catches a transposed C write; C may be NULL when only the planes are wanted."""
    from wsi_hgnn_amd import ops
    n = 160
    A = torch.eye(n, device=_dev())
    B = (torch.arange(n * n, device=_dev(), dtype=torch.float32).reshape(n, n) % 97) - 40.0
    C, Cp = _nt(A, B, want_planes=True, want_c=False)
    assert C is None
    assert torch.equal(ops.planes_to_float(Cp, n).cpu(), B.t().cpu())


def test_gemm_p3_nt_epilogues():
    from wsi_hgnn_amd import _native as N
    torch.manual_seed(4)
    M, Nn, K = 300, 256, 128
    A = torch.randn(M, K, device=_dev())
    B = torch.randn(Nn, K, device=_dev()) / math.sqrt(K)
    bias = torch.randn(Nn, device=_dev())
    R = torch.randn(M, Nn, device=_dev())
    gate = torch.tensor([0.3], device=_dev())
    Mm = (torch.rand(M, Nn, device=_dev()) > 0.2).float() / 0.8
    s = torch.sigmoid(gate.double().cpu())
    y = A.double().cpu() @ B.double().cpu().t() + bias.double().cpu()
    C, _ = _nt(A, B, N.WSI_EPI_GATED_SKIP, bias=bias, R=R, gate=gate)
    assert _relerr(C, s * y + (1 - s) * R.double().cpu()) < 3e-6
    C, _ = _nt(A, B, N.WSI_EPI_GATED_SKIP | N.WSI_EPI_MUL_M, bias=bias, R=R, gate=gate, Mm=Mm)
    assert _relerr(C, s * (y * Mm.double().cpu()) + (1 - s) * R.double().cpu()) < 3e-6
    C, _ = _nt(A, B, N.WSI_EPI_SCALE_GATE, gate=gate)
    assert _relerr(C, s * (y - bias.double().cpu())) < 3e-6
    C, _ = _nt(A, B, N.WSI_EPI_ADD_R | N.WSI_EPI_R_1MG, R=R, gate=gate)
    assert _relerr(C, (y - bias.double().cpu()) + (1 - s) * R.double().cpu()) < 3e-6
    C0 = torch.randn(M, Nn, device=_dev())
    C, _ = _nt(A, B, N.WSI_EPI_ACCUMULATE | N.WSI_EPI_BIAS, bias=bias, C_init=C0)
    assert _relerr(C, y + C0.double().cpu()) < 3e-6
    C, _ = _nt(A, B, N.WSI_EPI_BIAS | N.WSI_EPI_GELU, bias=bias)
    assert _relerr(C, torch.nn.functional.gelu(y)) < 3e-6


@pytest.mark.parametrize("K,M,N", [(300, 256, 128), (4099, 64, 96), (77, 2, 64), (2500, 512, 512), (40000, 128, 256), (33, 130, 129),
                                   (80000, 512, 512)])
def test_gemm_p3_tn_with_colsum(K, M, N):
    """dW = dY^T X over K rows (split-K, deterministic) + the bias gradient colsum(dY) from the ones-fragment MFMAs."""
    from wsi_hgnn_amd import ops, _native as Nn_
    torch.manual_seed(K + M + N)
    dY = torch.randn(K, M, device=_dev())
    X = torch.randn(K, N, device=_dev())
    Ap, Bp = ops.split_planes(dY), ops.split_planes(X)
    C = torch.empty(M, N, device=_dev())
    cs = torch.empty(M, device=_dev())
    g = [dict(Ap=Nn_.ptr(Ap), ldap=Ap.stride(0), Bp=Nn_.ptr(Bp), ldbp=Bp.stride(0), C=Nn_.ptr(C), ldc=N, colsum_out=Nn_.ptr(cs), M=M, N=N, K=K)]
    ops.gemm_p3(Nn_.WSI_GEMM_TN, 0, g, _dev())
    ref = dY.double().cpu().t() @ X.double().cpu()
    assert _relerr(C, ref) < 5e-6, _relerr(C, ref)
    assert _relerr(cs, dY.double().cpu().sum(0)) < 5e-6
    C2 = torch.empty_like(C)
    cs2 = torch.empty_like(cs)
    g[0].update(C=Nn_.ptr(C2), colsum_out=Nn_.ptr(cs2))
    ops.gemm_p3(Nn_.WSI_GEMM_TN, 0, g, _dev())
    assert torch.equal(C, C2) and torch.equal(cs, cs2)                      # deterministic


def test_gemm_p3_grouped_launch_and_argument_errors():
    from wsi_hgnn_amd import ops, _native as N
    torch.manual_seed(9)
    K = 96
    groups, refs, outs = [], [], []
    keep = []
    for M, Nn in ((150, 64), (1, 64), (249, 200)):
        A = torch.randn(M, K, device=_dev())
        B = torch.randn(Nn, K, device=_dev())
        Ap, Bp = ops.split_planes(A), ops.split_planes(B)
        C = torch.empty(M, Nn, device=_dev())
        keep += [Ap, Bp]
        groups.append(dict(Ap=N.ptr(Ap), ldap=Ap.stride(0), Bp=N.ptr(Bp), ldbp=Bp.stride(0), C=N.ptr(C), ldc=Nn, M=M, N=Nn, K=K))
        refs.append(A.double().cpu() @ B.double().cpu().t())
        outs.append(C)
    ops.gemm_p3(N.WSI_GEMM_NT, 0, groups, _dev())
    for C, ref in zip(outs, refs):
        assert _relerr(C, ref) < 2e-6
    with pytest.raises(RuntimeError, match="NT or TN"):
        ops.gemm_p3(N.WSI_GEMM_NN, 0, groups, _dev())
    bad = dict(groups[0])
    bad["ldap"] = 8
    with pytest.raises(RuntimeError, match="plane ld"):
        ops.gemm_p3(N.WSI_GEMM_NT, 0, [bad], _dev())
