"""Per-kernel time of the SnapKV passes vs sequence length (HIP events via kvp_prof_*)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kvpress_amd import _native
dev = "cuda:0"
import sys as _s
SIZES = [int(x) for x in _s.argv[1:]] or [4096, 16384, 32768, 65536, 131072, 262144]
for S in SIZES:
    g = torch.Generator(device=dev); g.manual_seed(0)
    k = torch.randn((1, 8, S, 128), generator=g, device=dev).to(torch.bfloat16)
    q = torch.randn((1, 32, 64, 128), generator=g, device=dev).to(torch.bfloat16)
    for _ in range(3):
        _native.snapkv_score(q, k, 5)
    torch.cuda.synchronize()
    _native.prof_enable(True)
    for _ in range(5):
        _native.snapkv_score(q, k, 5)
    torch.cuda.synchronize()
    t = {}
    for name, ms in _native.prof_records():
        t.setdefault(name, []).append(ms * 1e3)
    _native.prof_enable(False)
    print(S, {n: round(sum(v) / len(v), 1) for n, v in t.items()}, flush=True)
