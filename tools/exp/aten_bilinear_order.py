"""How does ATen's CPU bilinear upsample (the F.interpolate of reference trainer.py:573-576) round?

Findings (torch 2.10, AVX-512 build, this container), which csrc/misc.hip and oracle/csrc/ref_ops.c restate:
  * more than one intra-op thread (the training process of the reference): generic NCHW kernel, FMA-contracted --
        src = max(fma(scale, dst + .5, -.5), 0);  t_r = fma(p[r][x0], lx0, p[r][x1] * lx1);  out = fma(t_y0, ly0, t_y1 * ly1)
    -> `oracle.d2.bilinear_shrink_u8` reproduces every float of F.interpolate (checked below on ~1e7 pixels);
  * exactly one thread and C == 3: ATen switches to its channels-last kernel (2-D weights w = ly * lx, accumulated as
    fma(p11, w11, fma(p10, w10, fma(p00, w00, p01 * w01)))) whose uint8 truncation differs in ~1e-3 of the bytes.
Run:  python tools/exp/aten_bilinear_order.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import d2  # noqa: E402


def main():
    rng = np.random.RandomState(0)
    for threads in (torch.get_num_threads(), 1):
        torch.set_num_threads(threads)
        bad = tot = 0
        for (h, w) in [(90, 120), (600, 800), (800, 1333)]:
            img = torch.from_numpy(rng.randint(0, 256, (3, h, w)).astype(np.uint8))
            for r in (0.5, 0.61803, 0.75, 0.83, 0.9999, 1.0):
                dh, dw = int(h * r), int(w * r)
                ref = torch.zeros((3, dh, dw), dtype=torch.uint8)
                ref[:] = F.interpolate(img.unsqueeze(0).float(), size=(dh, dw), align_corners=False, mode="bilinear")[0]
                bad += int((d2.bilinear_shrink_u8(img, dh, dw) != ref).sum())
                tot += ref.numel()
        print(f"threads={threads}: {bad} of {tot} bytes differ from F.interpolate")


if __name__ == "__main__":
    main()
