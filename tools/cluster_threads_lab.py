"""Lab: three Python threads, each with its own stream, 400 one-launch selects each, concurrently (ctypes drops the GIL inside the library): exercises the
mutex and the cross-stream events of ClusterLaunchScope (topk_cluster.hip).  Measurement / stress aid, not part of the product."""
import sys, threading, numpy as np, torch
sys.path.insert(0, "/root/repo")
from kvpress_amd import _native as n
from oracle import kvpress_oracle as O
DEV = "cuda:0"
g = torch.Generator().manual_seed(0)
rows = [torch.randn(8, 131008, generator=g, dtype=torch.float32) for _ in range(3)]
wants = [O.topk_select(r.numpy(), 60000 + 1000 * i) for i, r in enumerate(rows)]
devs = [r.to(DEV) for r in rows]
errs = []
def worker(i):
    try:
        torch.cuda.set_device(0)
        s = torch.cuda.Stream(device=DEV)
        with torch.cuda.stream(s):
            for it in range(400):
                got = n.topk_select(devs[i], 60000 + 1000 * i)
                if it % 50 == 0:
                    s.synchronize()
                    assert np.array_equal(got.cpu().numpy(), wants[i]), f"thread {i} iteration {it}: WRONG"
            s.synchronize()
            assert np.array_equal(got.cpu().numpy(), wants[i])
    except Exception as e:
        errs.append((i, repr(e)))
ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
n.async_error_check()
print("errors:", errs if errs else "none", "-- three threads x 400 cluster selects on three streams")
