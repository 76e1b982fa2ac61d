import sys, torch
sys.path.insert(0, '/root/repo')
from wsi_hgnn_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(77)
for (M, N, K, spread) in [(512, 384, 4096, 6), (512, 384, 4096, 0), (512, 384, 512, 0), (256, 256, 32768, 0)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    if spread:
        x = x * torch.exp2(torch.randint(-spread, spread + 1, (M, K), device=dev).float())
        w = w * torch.exp2(torch.randint(-spread, spread + 1, (N, K), device=dev).float())
    ref = x.double().cpu() @ w.double().cpu().t()
    scale = x.abs().double().cpu() @ w.abs().double().cpu().t()
    out = {}
    for mode in ('fp32', 'bf16x6'):
        ops.set_gemm_precision(mode)
        y = ops.linear(x, w, None).double().cpu()
        out[mode] = (((y - ref).abs() / scale).max().item(), ((y - ref).abs().max() / ref.abs().max()).item(), ((y-ref)/scale).mean().item())
    t = (x @ w.t()).double().cpu()
    out['torch.mm'] = (((t - ref).abs() / scale).max().item(), ((t - ref).abs().max() / ref.abs().max()).item(), ((t-ref)/scale).mean().item())
    print(M, N, K, spread, out)
