"""Device rates of plain fills (writes only), reads (a reduction) and copies at the sizes of the step's tensors (MB): what a
write-only / read-only / copy kernel can reach on this box, beside the 8 TB/s of the data sheet."""
import torch

d = torch.device("cuda:0")


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for mb in (42, 168, 672):
    n = mb * 1024 * 1024 // 4
    a, b = torch.empty(n, device=d), torch.empty(n, device=d)
    tf = t_us(lambda: a.fill_(1.0))
    tc = t_us(lambda: b.copy_(a))
    tr = t_us(lambda: a.max())
    print(f"{mb:4d} MB: fill {tf:6.1f} us = {mb * 1.048576 / tf:5.2f} TB/s | copy {tc:6.1f} us = {2 * mb * 1.048576 / tc:5.2f} TB/s | "
          f"read (max) {tr:6.1f} us = {mb * 1.048576 / tr:5.2f} TB/s")
