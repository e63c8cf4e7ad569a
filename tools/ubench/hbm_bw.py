import torch
dev = torch.device("cuda:0")
n = 1 << 30   # floats: 4 GiB
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
ms = t(lambda: x.fill_(1.0)); print(f"fill 4 GiB: {ms:.3f} ms = {4.295/ms*1e3/1e3:.2f} TB/s written")
ms = t(lambda: x.zero_()); print(f"zero 4 GiB: {ms:.3f} ms = {4.295/ms:.2f} TB/s written")
ms = t(lambda: x.copy_(y)); print(f"copy 4 GiB: {ms:.3f} ms = {2*4.295/ms:.2f} TB/s read+written")
ms = t(lambda: torch.sum(x)); print(f"sum  4 GiB: {ms:.3f} ms = {4.295/ms:.2f} TB/s read")
