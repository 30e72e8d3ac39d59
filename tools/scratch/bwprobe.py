import torch, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
from recon_probe import timeit
d = torch.device("cuda:0")
for mb in (20, 79, 160, 316, 1000):
    n = mb * 1000 * 1000 // 4
    src = torch.randn(n, device=d); dst = torch.empty_like(src)
    out = torch.empty((), device=d)
    t_sum = timeit(lambda: src.sum())
    t_copy = timeit(lambda: dst.copy_(src))
    t_fill = timeit(lambda: dst.fill_(1.0))
    print(f"{mb} MB: sum {t_sum:.1f} us = {n*4/t_sum/1e6:.2f} TB/s read; copy {t_copy:.1f} us = {2*n*4/t_copy/1e6:.2f} TB/s r+w; fill {t_fill:.1f} us = {n*4/t_fill/1e6:.2f} TB/s write")
