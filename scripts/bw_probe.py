import torch, time
dev = torch.device("cuda:0")
x = torch.rand(1 << 28, device=dev)  # 1 GiB
y = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
gb = x.numel() * 4 / 1e9
print("sum  (read 1.07 GB):  %.3f ms = %.2f TB/s" % (t(lambda: x.sum()), gb / t(lambda: x.sum())))
print("copy (read+write):    %.3f ms = %.2f TB/s" % (t(lambda: y.copy_(x)), 2 * gb / t(lambda: y.copy_(x))))
print("fill (write 1.07 GB): %.3f ms = %.2f TB/s" % (t(lambda: y.zero_()), gb / t(lambda: y.zero_())))
