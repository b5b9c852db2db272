import math, sys, os
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_wino4_gpu import _wino4, g
import itertools
for (n, cin, cout, h, w), epi in itertools.product([(1, 8, 64, 8, 64), (1, 8, 16, 8, 63), (1, 16, 64, 8, 64), (2, 64, 64, 24, 72)], [0, 2, 0]):
    gen = g(1)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    ref = F.conv2d(x, wt, b if epi == 0 else None, padding=1)
    got = _wino4(x.cuda(), wt.cuda(), b.cuda(), None, epi).cpu()
    err = (got - ref).abs()
    bad = (err > 1e-4) | ~torch.isfinite(got)
    print((n, cin, cout, h, w), "epi", epi, "bad", int(bad.sum()), "channels with bad", int(bad.any(-1).any(-1).sum()))
    if bad.any():
        c0 = int(bad.any(-1).any(-1)[0].nonzero()[0])
        ys, xs = bad[0, c0].nonzero(as_tuple=True)
        print("  channel", c0, "bad (row, col):", sorted(set(zip(ys.tolist(), xs.tolist())))[:40])
        print("  got", got[0, c0][bad[0, c0]][:8].tolist(), "ref", ref[0, c0][bad[0, c0]][:8].tolist())
