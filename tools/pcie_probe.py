"""D2H / H2D bandwidth of pinned copies on this box (one stream, two streams, with HSA_ENABLE_SDMA as set in the environment)"""
import time, torch
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device="cuda:0"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda:0")
h = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
print("D2H one stream: %.1f GB/s" % (n / t(lambda: h.copy_(d, non_blocking=True)) / 1e9))
print("H2D one stream: %.1f GB/s" % (n / t(lambda: d.copy_(h, non_blocking=True)) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
print("D2H two streams: %.1f GB/s" % (2 * n / t(two) / 1e9))
def both():
    with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
print("D2H + H2D at once: %.1f GB/s each way" % (n / t(both) / 1e9))
