"""Development probe: what a plain read-only sweep of 400 MB achieves on this box."""
import torch

x = torch.randn(50_000_000, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in (("sum f64 400MB", lambda: x.sum(), 400e6), ("copy 400->400MB", lambda: y.copy_(x), 800e6),
                         ("max f64 400MB", lambda: x.max(), 400e6)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    print(f"{name}: {t*1e6:.1f} us  {nbytes/t/1e12:.2f} TB/s")
