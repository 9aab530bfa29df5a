import torch, time
dev = torch.device("cuda:0")
x = torch.rand(512, 512, 6, device=dev)
pin = torch.empty(512, 512, 6).pin_memory()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("x.cpu()                         %.2f ms" % t(lambda: x.cpu()))
def viapin():
    pin.copy_(x, non_blocking=True); torch.cuda.synchronize(); return pin
print("pinned copy + sync              %.2f ms" % t(viapin))
def viapin_clone():
    pin.copy_(x, non_blocking=True); torch.cuda.synchronize(); return pin.clone()
print("pinned copy + sync + clone      %.2f ms" % t(viapin_clone))
def split():
    p = viapin(); return p[..., 0:3].contiguous(), p[..., 3:4].contiguous(), p[..., 4:5].contiguous(), p[..., 5:6].contiguous()
print("pinned + 4 contiguous slices    %.2f ms" % t(split))
c = x.cpu()
print("4 contiguous slices of a CPU tensor %.2f ms" % t(lambda: (c[..., 0:3].contiguous(), c[..., 3:4].contiguous(), c[..., 4:5].contiguous(), c[..., 5:6].contiguous())))
print("torch.empty(512,512,6).zero_()  %.2f ms" % t(lambda: torch.empty(512, 512, 6).zero_()))
