import torch, time
dev = torch.device("cuda", 0)
m = 1 << 27
full = torch.zeros(m + 4, device=dev)
mask = torch.rand(m, device=dev) < 0.39
idx = torch.nonzero(mask).flatten()
src = torch.rand(idx.numel(), device=dev)
torch.cuda.synchronize()
for name, fn in (("index_copy_", lambda: full.index_copy_(0, idx, src)), ("index_select", lambda: full.index_select(0, idx))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, idx.numel(), "elements: %.3f ms" % ((time.perf_counter() - t) * 100))
