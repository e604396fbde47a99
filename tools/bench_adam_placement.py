"""Adam step time against the relative placement of its four streams (parameters, gradients, two moments) at BASELINE
config #4 size (6 M Gaussians x 59 floats): the same launch runs at 5.4 .. 6.05 TB/s depending on where the allocator put the
buffers — the run-to-run spread of config #4 / #5 Adam times.  Output: profiles/r3_adam_placement.txt."""
import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fusionsense_amd import ops
dev = torch.device("cuda:0")
N = 6_000_000
sizes = [N * 3, N * 3, N * 4, N * 3, N * 45, N]   # the six groups
tot = sum(sizes)
def run(stagger_bytes, label):
    st = stagger_bytes // 4
    span = (tot + (1 << 20)) // 1024 * 1024 + (1 << 18)
    big = torch.zeros(4 * span + 16 * st + 1024, dtype=torch.float32, device=dev)
    views = []
    for i in range(4):  # p, g, m, v
        base = i * span + i * st
        vs, off = [], base
        for s in sizes:
            vs.append(big[off:off + s]); off += s
        views.append(vs)
    p, g, m, v = views
    for t in g: t.normal_()
    lrs = [1e-4] * 6
    for _ in range(3): ops.adam_step_(p, g, m, v, lrs, 1)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for k in range(20):
        ops.adam_step_(p, g, m, v, lrs, k + 2); ev[k + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(20))
    print(f"{label:28s} base%2MiB={[hex(t[0].data_ptr() % (1<<21)) for t in views]} min {d[0]:.3f} med {d[10]:.3f} max {d[-1]:.3f} ms  {tot*28/d[10]/1e9*1e3:.0f} GB/s")
    del big
for rep in range(2):
    run(0, "no stagger")
    run(256, "256 B")
    run(4096, "4 KiB")
    run(4096 * 3 + 256, "12.25 KiB")
    run(65536 + 4096, "68 KiB")
    run((1 << 20) + 8192, "1 MiB + 8 KiB")
# separate allocations as the trainer makes them
def sep():
    p = [torch.zeros(s, device=dev) for s in sizes]; g = [torch.randn(s, device=dev) for s in sizes]
    m = [torch.zeros(s, device=dev) for s in sizes]; v = [torch.zeros(s, device=dev) for s in sizes]
    lrs = [1e-4] * 6
    for _ in range(3): ops.adam_step_(p, g, m, v, lrs, 1)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for k in range(20):
        ops.adam_step_(p, g, m, v, lrs, k + 2); ev[k + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(20))
    print(f"separate allocations: ptr%2MiB of rest group p,g,m,v = {[hex(t[4].data_ptr() % (1<<21)) for t in (p,g,m,v)]} med {d[10]:.3f} ms {tot*28/d[10]/1e9*1e3:.0f} GB/s")
sep(); sep()
