import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench
from kvpress_amd import _native
dev = torch.device('cuda', 0)
kind, S, ratio = bench.WORKLOADS['snapkv128k']
gen = torch.Generator(device=dev); gen.manual_seed(1)
bf = torch.bfloat16
keys = torch.randn((1, 8, S, 128), generator=gen, device=dev, dtype=torch.float32).to(bf)
values = torch.randn((1, 8, S, 128), generator=gen, device=dev, dtype=torch.float32).to(bf)
hidden = torch.randn((1, S, 4096), generator=gen, device=dev, dtype=bf)
att, rot = bench.build_module(dev)
with torch.no_grad():
    pe = rot(hidden, torch.arange(S, device=dev)[None])
press = bench.make_press(kind, ratio)
kw = {"position_embeddings": pe}
from kvpress_amd.utils import get_prerope_query_states
with torch.no_grad():
    q_pre = get_prerope_query_states(att, hidden[:, -64:])
cos, sin = pe[0][:, -64:], pe[1][:, -64:]
def run(name, fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    probes = []
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % max(1, n // 8) == 0: probes.append(_native.clock_probe(dev, 5))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    _native.prof_enable(True); fn(); torch.cuda.synchronize(); inside = _native.prof_kernel_clock(); _native.prof_enable(False)
    print(f"{name}: {dt*1e6:.1f} us/iter, clock probes MHz: {[int(p.item()) for p in probes]}; inside snapkv_p1_mfma right after the loop: {inside:.0f} MHz")
idle = _native.clock_probe(dev, 50); torch.cuda.synchronize(); print("idle probe MHz:", int(idle.item()))
with torch.no_grad():
    run("compress loop (bench step)", lambda: press.compress(att, hidden, keys, values, None, kw), 300)
    run("score only loop (rope+p1+combine+p2+pool+fill)", lambda: _native.snapkv_score_rope(q_pre, cos, sin, keys, 5), 300)
    idx = _native.topk_select(_native.snapkv_score_rope(q_pre, cos, sin, keys, 5), S // 2)
    run("gather only loop", lambda: _native.gather_kv(keys, values, idx), 300)
