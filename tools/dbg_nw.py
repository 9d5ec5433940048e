import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from mi355 import lib as milib
L = milib.get()
st = torch.cuda.current_stream().cuda_stream
B, IH, IW, Ci, Co, k = 1, 10, 12, 3, 32, 4
OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
mode = sys.argv[1] if len(sys.argv) > 1 else "x"
x = torch.zeros(B, IH, IW, Ci, device="cuda")
dy = torch.zeros(B, OH, OW, Co, device="cuda")
if mode == "x":      # x = index of (kh..), dy = 1 at one pixel/channel -> dw[kc][n0] = patch of that pixel
    x = torch.arange(B * IH * IW * Ci, device="cuda", dtype=torch.float32).reshape(B, IH, IW, Ci) % 64
    dy[0, 1, 2, 5] = 1.0
else:
    x[:] = 1.0
    dy = (torch.arange(B * OH * OW * Co, device="cuda", dtype=torch.float32).reshape(B, OH, OW, Co) % 16)
res = []
for on in (0, 1):
    L.mi_set_tuning(4, on)
    dw = torch.zeros(k, k, Ci, Co, device="cuda")
    L.mi_conv2d_nhwc_wgrad(st, 1, x.data_ptr(), None, 1, B, IH, IW, Ci, dy.to(torch.bfloat16).data_ptr(), k, k, Co, dw.data_ptr())
    torch.cuda.synchronize()
    res.append(dw.cpu().numpy().reshape(48, 32))
np.set_printoptions(linewidth=250, precision=0, suppress=True)
if mode == "x":
    print("old col5:", res[0][:, 5]); print("new col5:", res[1][:, 5])
    nz = np.argwhere(res[1] != 0)
    print("new nonzero cols:", sorted(set(nz[:, 1].tolist()))[:40], "rows:", sorted(set(nz[:, 0].tolist()))[:64])
else:
    print("old row0:", res[0][0]); print("new row0:", res[1][0]); print("new row1:", res[1][1]); print("new row 47", res[1][47])
