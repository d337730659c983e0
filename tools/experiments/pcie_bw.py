"""Experiment: what the PCIe link of the GPU box delivers for pinned host <-> device copies (bounds flgpu_transform)."""
import time, torch
for mb in (6, 64, 400):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        reps = max(4, 2000 // mb)
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} {mb} MiB x {reps}: {mb * reps / 1024 / dt:.1f} GiB/s")
