import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.nn.functional as F
from tests.test_wino_gpu import _wino_wgrad
DEV = "cuda:0"
n, cin, cout, h, w = [int(v) for v in sys.argv[1:6]]
ch = int(sys.argv[6])
bad = []
for img in range(n):
    for y in range(h):
        for c in range(w):
            x = torch.zeros(n, cin, h, w); x[img, ch, y, c] = 1
            gy = torch.zeros(n, cout, h, w); gy[:, 0] = 1
            wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
            F.conv2d(x, wt, None, padding=1).backward(gy)
            dw, _ = _wino_wgrad(x.to(DEV), gy.to(DEV), cout)
            d = (dw.cpu() - wt.grad).abs()
            if d.max() > 1e-4:
                bad.append((img, y, c, d[0, ch].tolist(), float(d.max()), torch.nonzero(d > 1e-4)[:4].tolist()))
print(len(bad), "bad input pixels")
for b in bad[:40]:
    print(b)
