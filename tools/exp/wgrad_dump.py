import sys, os, ctypes, torch
n, cin, cout, h, w = [int(v) for v in sys.argv[1:6]]
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin/libptmi355_dump.so"))
lib.ptmi_conv3x3_wino_wgrad_ws_floats.restype = ctypes.c_int64
DEV = "cuda:0"
x = (torch.arange(n * cin * h * w, dtype=torch.float32).reshape(n, cin, h, w) + 1).to(DEV)
dy = -(torch.arange(n * cout * h * w, dtype=torch.float32).reshape(n, cout, h, w) + 1).to(DEV)
nws = lib.ptmi_conv3x3_wino_wgrad_ws_floats(n, cin, cout, h, w)
ws = torch.full((max(nws, 16384),), float("nan"), device=DEV)
dw = torch.empty(cout, cin, 3, 3, device=DEV)
vp = ctypes.c_void_p
lib.ptmi_conv3x3_wino_wgrad(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), None, vp(ws.data_ptr()), n, cin, cout, h, w, 0, None)
torch.cuda.synchronize()
st = ws[:16384].cpu()
xs = st[5120:5120 + 64 * 164].reshape(64, 164)
xc = x.cpu()
bad = 0
# chunk 0 of block 0: image 0, y0 = 0, x0 = 0
for ch in range(min(cin, 64)):
    for r in range(4):
        for c in range(40):
            gy, gx = r - 1, c - 4
            exp = float(xc[0, ch, gy, gx]) if (0 <= gy < h and 0 <= gx < w) else 0.0
            got = float(xs[ch, r * 40 + c])
            if got != exp:
                bad += 1
                if bad < 30:
                    print("x plane", ch, "row", r, "col", c, "got", got, "expected", exp)
print("bad x words", bad)
ds = st[:64 * 68].reshape(64, 68)
dc = dy.cpu(); bad = 0
for ch in range(min(cout, 64)):
    for r in range(2):
        for c in range(32):
            exp = float(dc[0, ch, r, c]) if (r < h and c < w) else 0.0
            got = float(ds[ch, r * 32 + c])
            if got != exp:
                bad += 1
                if bad < 30:
                    print("dy plane", ch, "row", r, "col", c, "got", got, "expected", exp)
print("bad dy words", bad)
info = ws[16384:16384 + 11 * 256 * 4].cpu().reshape(11, 256, 4)
for px in (255, 296, 297, 295, 337):
    i, t = divmod(px, 256)
    print("piece", px, "idx", i, "tid", t, "voff r q4 nC", info[i, t].tolist(), "expected r q4", (px % 41) // 10, 4 * ((px % 41) % 10) - 4)
