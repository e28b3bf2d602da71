"""plain write / copy bandwidth of the box (torch fill_ and copy_ at the stride-2 dgrad's output size)"""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (67, 134, 268, 536, 1072):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.randn(n, device=dev).bfloat16()
    us_f = t(lambda: a.fill_(1.0)); us_c = t(lambda: a.copy_(b))
    print(f"{mb:5d} MB: fill {us_f:7.1f} us = {mb / us_f:5.2f} TB/s written; copy {us_c:7.1f} us = {2 * mb / us_c:5.2f} TB/s read + written")
