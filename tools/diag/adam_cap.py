import torch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for scale in (1e-3, 1e-5, 1e-6, 1e-7, 1e-8, 1e-10):
    p0 = torch.randn(100000, device=dev)
    gs = [torch.randn(100000, device=dev) * scale for _ in range(4)]
    out = {}
    for name, kw in (("plain", {}), ("capturable", dict(capturable=True)), ("cap_fused", dict(capturable=True, fused=True)), ("fused", dict(fused=True)), ("plain_single", dict(foreach=False))):
        p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([p], lr=1e-2, **kw)
        for g in gs:
            p.grad = g.clone(); opt.step()
        out[name] = p.detach().double().clone()
    # fp64 truth
    p = p0.double().clone(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for t, g in enumerate(gs, 1):
        g = g.double(); m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        p = p - 1e-2 * (m / (1 - 0.9 ** t)) / ((v / (1 - 0.999 ** t)).sqrt() + 1e-8)
    print(scale, {k: f"{(o - p).abs().max().item():.2e}" for k, o in out.items()})
