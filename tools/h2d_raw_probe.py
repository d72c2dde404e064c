"""Raw host-to-device rate of this box's link, no kernels beside: one 75.5 MB page-locked buffer (256 x 96 chunks of s16) in 1..4 pieces on as many
streams.  The ceiling the host-fed rate (tools/host_fed_probe*.py) is measured against.  python tools/h2d_raw_probe.py"""
import time, torch
n = 256 * 96 * 3072
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for parts in (1, 2, 3, 4):
    ss = [torch.cuda.Stream() for _ in range(parts)]
    piece = (n + parts - 1) // parts
    def go():
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                d[i * piece:(i + 1) * piece].copy_(h[i * piece:(i + 1) * piece], non_blocking=True)
    for _ in range(3): go()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print(f"pieces={parts}: {dt * 1e3:.3f} ms, {n / dt / 1e9:.1f} GB/s", flush=True)
