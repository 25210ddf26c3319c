"""Practical HBM ceiling on this box: device-to-device copy and read-only reduction rates (torch), for DESIGN.md context."""
import torch, time, json
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")  # 4 GiB
y = torch.empty_like(x)
x.fill_(1.0)
res = {}
for name, fn, nbytes in (("copy_rw", lambda: y.copy_(x), 2 * x.numel() * 4), ("read_sum", lambda: x.sum(), x.numel() * 4), ("fill_w", lambda: y.fill_(2.0), x.numel() * 4)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    res[name] = round(nbytes / dt / 1e12, 3)
print(json.dumps({"TB_per_s": res}))
