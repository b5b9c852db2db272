#!/usr/bin/env python
"""CPU study behind the interpolation points of csrc/wino4.hip / wino4w.hip: fp32 Winograd F(4x4,3x3) (transforms in fp32, products
accumulated in fp32) against an fp64 convolution, for the textbook points and for symmetric point sets (0, +-a, +-b, inf), and the
F(4x4,3x3)-domain weight gradient against fp64.  Runs anywhere (no GPU):
    python tools/exp/wino4_numerics.py [--scan]
Prints max / rms absolute error for unit-scale outputs (He-initialised weights, post-ReLU inputs) next to direct fp32 and F(2x2,3x3)."""
import argparse
import math

import numpy as np
import torch


def cook(points, m=4, r=3):
    """Cook-Toom matrices A^T (m x a), G (a x r), B^T (a x a) for the given finite points + infinity"""
    a = len(points) + 1
    P = np.array(points, dtype=np.float64)
    AT, G = np.zeros((m, a)), np.zeros((a, r))
    for i, p in enumerate(P):
        N = np.prod([p - q for j, q in enumerate(P) if j != i])
        AT[:, i] = [p ** k for k in range(m)]
        G[i, :] = [p ** k / N for k in range(r)]
    AT[m - 1, a - 1] = 1
    G[a - 1, r - 1] = 1
    BT, Mx = np.zeros((a, a)), np.zeros((m * r, a))
    for b in range(a):
        rhs = np.zeros(m * r)
        for k in range(m):
            for aa in range(r):
                Mx[k * r + aa, :] = AT[k, :] * G[:, aa]
                rhs[k * r + aa] = 1.0 if b == k + aa else 0.0
        sol = np.linalg.lstsq(Mx, rhs, rcond=None)[0]
        assert np.abs(Mx @ sol - rhs).max() < 1e-9
        BT[:, b] = sol
    return AT, G, BT


def wino(x, w, AT, G, BT, m):
    AT, BTt, G = torch.tensor(AT, dtype=torch.float32), torch.tensor(BT, dtype=torch.float32), torch.tensor(G)
    a = BTt.shape[0]
    C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    U = (G @ w.double() @ G.T).float()
    win = xp.unfold(1, a, m).unfold(2, a, m)
    V = BTt @ win @ BTt.T
    M = torch.einsum('ocij,ctuij->otuij', U, V)
    Y = AT @ M @ AT.T
    return Y.permute(0, 1, 3, 2, 4).reshape(-1, H, W)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scan", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    cin, cout, H, W = 512, 64, 48, 64
    x = torch.relu(torch.randn(cin, H, W))
    w = torch.randn(cout, cin, 3, 3) * math.sqrt(2 / (9 * cin))
    ref = torch.nn.functional.conv2d(x.double()[None], w.double(), padding=1)[0]
    d32 = torch.nn.functional.conv2d(x[None], w, padding=1)[0]

    def rep(name, y):
        e = (y.double() - ref).abs()
        print(f"{name:44s} max {float(e.max()):.2e}  rms {float(e.pow(2).mean().sqrt()):.2e}  (max |y| {float(ref.abs().max()):.2f})")
    rep("direct fp32", d32)
    rep("F(2x2,3x3) points 0, +-1, inf", wino(x, w, *cook([0, 1, -1], m=2), 2))
    rep("F(4x4,3x3) textbook 0, +-1, +-2, inf", wino(x, w, *cook([0, 1, -1, 2, -2]), 4))
    rep("F(4x4,3x3) 0, +-3/4, +-3/2, inf (the kernels')", wino(x, w, *cook([0, .75, -.75, 1.5, -1.5]), 4))
    if a.scan:
        res = []
        for a16 in range(6, 17):
            for b16 in range(a16 + 4, 40, 2):
                pa, pb = a16 / 16, b16 / 16
                e = (wino(x, w, *cook([0, pa, -pa, pb, -pb]), 4).double() - ref).abs()
                res.append((float(e.max()), float(e.pow(2).mean().sqrt()), pa, pb))
        for r in sorted(res)[:12]:
            print("   scan: max %.2e rms %.2e a=%.4f b=%.4f" % r)
    # weight gradient through the same domain: dU_p = sum_tiles (A dY A^T)_p (B^T d B)_p, dg = G^T dU G
    AT, G, BT = cook([0, .75, -.75, 1.5, -1.5])
    n, ci, co, h, wd = 4, 128, 64, 100, 168
    xx, dy = torch.relu(torch.randn(n, ci, h, wd)), torch.randn(n, co, h, wd)
    wref = torch.nn.grad.conv2d_weight(xx.double(), (co, ci, 3, 3), dy.double(), padding=1)
    w32 = torch.nn.grad.conv2d_weight(xx, (co, ci, 3, 3), dy, padding=1)
    BTt, A = torch.tensor(BT, dtype=torch.float32), torch.tensor(AT.T, dtype=torch.float32)
    win = torch.nn.functional.pad(xx, (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)
    V = BTt @ win @ BTt.T
    Wp = A @ dy.unfold(2, 4, 4).unfold(3, 4, 4) @ A.T
    dU = torch.einsum('nothij,ncthij->ocij', Wp, V)
    got = torch.tensor(G).T @ dU.double() @ torch.tensor(G)
    sc = float(wref.abs().max())
    print(f"weight gradient {ci}->{co} {h}x{wd} n={n}: direct fp32 err / max|dW| {float((w32.double() - wref).abs().max()) / sc:.2e}, "
          f"F(4x4,3x3) domain {float((got - wref).abs().max()) / sc:.2e}")


if __name__ == "__main__":
    main()
