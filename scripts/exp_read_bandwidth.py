import torch
n = 19_700_000_000 // 8
v = torch.zeros(n, dtype=torch.float64, device="cuda")
for _ in range(2): v.sum()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): s = v.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("torch sum over 19.7 GB: %.3f ms  %.2f TB/s" % (ms, n * 8 / ms / 1e9))
w = torch.empty_like(v)
e0.record()
for _ in range(3): w.copy_(v)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("torch copy 19.7 GB -> 19.7 GB: %.3f ms  %.2f TB/s (read + write)" % (ms, 2 * n * 8 / ms / 1e9))
