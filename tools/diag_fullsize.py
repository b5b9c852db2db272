"""Diagnostic (GPU box): which side is off at BASELINE size -- HIP kernel or torch-CPU fp32 reference?  fp64 CPU is truth."""
import math, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probabilisticteacher_amd import ops
DEV = "cuda:0"
torch.set_num_threads(int(os.environ.get("NT", "64")))
print("threads", torch.get_num_threads(), "mkldnn", torch.backends.mkldnn.is_available())

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    d = (a - b).abs()
    i = int(d.argmax())
    idx = []
    for s in reversed(b.shape):
        idx.append(i % s); i //= s
    return f"{float(d.max()) / float(b.abs().max()):.2e} at {tuple(reversed(idx))} (frac>1e-4*max: {float((d > 1e-4 * b.abs().max()).double().mean()):.2e})"

def conv_case(n, cin, cout, h, w, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g))
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=g) * 0.1
    gy = torch.randn(n, cout, h, w, generator=g)
    outs = {}
    for name, dt, flag in (("f64", torch.float64, True), ("f32", torch.float32, True), ("f32_nomkldnn", torch.float32, False)):
        with torch.backends.mkldnn.flags(enabled=flag):
            xr, wr, br = (t.to(dt).clone().requires_grad_() for t in (x, wt, b))
            t0 = time.time()
            y = F.relu(F.conv2d(xr, wr, br, padding=1))
            y.backward(gy.to(dt))
            outs[name] = (y.detach(), xr.grad, wr.grad, br.grad, time.time() - t0)
    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, wt, b))
    yd = ops.conv3x3(xd, wd, bd, True)
    yd.backward(gy.to(DEV))
    hip = (yd.detach(), xd.grad, wd.grad, bd.grad)
    print(f"--- conv n={n} {cin}->{cout} {h}x{w}  (cpu times f64 {outs['f64'][4]:.1f}s f32 {outs['f32'][4]:.1f}s)")
    for j, nm in enumerate(("fwd", "dgrad", "wgrad", "bgrad")):
        print(f"  {nm:6s} HIP vs f64 {rel(hip[j], outs['f64'][j])} | cpu-f32 vs f64 {rel(outs['f32'][j], outs['f64'][j])} | cpu-f32-nomkldnn vs f64 {rel(outs['f32_nomkldnn'][j], outs['f64'][j])}")

for case in [(2, 512, 512, 50, 83), (1, 512, 512, 50, 83), (2, 256, 512, 100, 166), (2, 128, 256, 200, 333), (1, 256, 512, 9, 83)]:
    conv_case(*case)

# linear at fc1 size
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn(1024, 25088, generator=g)); w = torch.randn(1024, 25088, generator=g) / math.sqrt(25088); b = torch.zeros(1024)
ref64 = F.linear(x.double(), w.double(), b.double())
ref32 = F.linear(x, w, b)
hip = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), False)
print("--- linear 1024x25088x1024: HIP vs f64", rel(hip, ref64), "| cpu-f32 vs f64", rel(ref32, ref64))
