import torch, time, os, sys
import torch.nn.functional as F
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
x = torch.randn(1, 512, 64, 64); w = torch.randn(512, 512, 3, 3, requires_grad=True)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    y = F.conv2d(x, w, padding=1); y.sum().backward()
    t0 = time.time()
    for _ in range(3):
        y = F.conv2d(x, w, padding=1); y.sum().backward()
    print(nt, "threads:", round((time.time() - t0) / 3 * 1e3, 1), "ms per conv fwd+bwd", flush=True)
