"""Op-level parity: every C-ABI kernel launcher vs a float64 torch-CPU statement of the same TF op, on the layer
geometries of the ConvVAE (odd sizes, k=5, C=3 and C=1 edge layers, fused minibatch gather) and the PPO MLP."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hip_helpers import DT, DTS, P, X3, alloc, assert_close, dev, host, rounded, stream, tols  # noqa: E402
from mi355 import lib as milib  # noqa: E402

CONVS = [  # IH, IW, Cin, Cout, k
    (80, 160, 3, 32, 4), (39, 79, 32, 64, 4), (18, 38, 64, 128, 4), (8, 18, 128, 256, 4)]
DECONVS = [  # IH, IW, Cin, Cout, k
    (3, 8, 256, 128, 4), (8, 18, 128, 64, 4), (18, 38, 64, 32, 5), (39, 79, 32, 3, 4), (39, 79, 32, 1, 4)]



# Every conv / deconv entry point is served by several kernel generations (first-generation register-staged tiles, gemm2 LDS-DMA
# tiles, raw-staged tapconv / tapwgrad, the narrow-layer kernels); `auto` is what production picks at these (small) sizes, the
# other two pin the dispatch through mi_set_tuning so that EVERY generation meets the same float64 reference on every geometry.
# `rwconv` = `newest` with the register-weight kernels forced for the thin gather-form layers (key 13 = 2; auto takes them only on chip-filling grids)
# and for all conv-form shapes (key 15 = 3: 32 -> 64 channels k = 5 and k = 4, 64 -> 128 channels k = 4), on 8 persistent blocks (key 16 = 1: several chunks per block)
GENERATIONS = {"auto": None, "gen1": {0: 0, 1: -1, 3: 0, 4: 0, 13: 0}, "newest": {0: 1, 1: 1, 3: 1, 4: 1, 13: 0},
               "rwconv": {0: 1, 1: 1, 3: 1, 4: 1, 13: 2, 15: 3, 16: 1}}


@pytest.fixture(params=list(GENERATIONS))
def kernels(request):
    L = milib.get()
    want = GENERATIONS[request.param]
    prev = {}
    if want:
        for k, v in want.items():
            prev[k] = L.mi_set_tuning(k, v)
    yield request.param
    for k, v in prev.items():
        L.mi_set_tuning(k, v)


def _nchw(a):
    return a.permute(0, 3, 1, 2)


def _nhwc(a):
    return a.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("geom", CONVS)
def test_conv_fwd_dgrad_wgrad(dt, geom, kernels):
    if kernels == "rwconv" and dt != "bf16":
        pytest.skip("the register-weight kernel is bf16 only")
    L = milib.get()
    code, td = DT[dt]
    IH, IW, Ci, Co, k = geom
    B = 3
    rng = np.random.RandomState(Ci)
    first = Ci == 3
    x = rng.rand(5 if first else B, IH, IW, Ci).astype(np.float32)
    idx = np.array([3, 0, 4], np.int32) if first else None
    w = (rng.randn(k, k, Ci, Co) / np.sqrt(k * k * Ci)).astype(np.float32)
    b = (0.1 * rng.randn(Co)).astype(np.float32)
    OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
    dy = rng.randn(B, OH, OW, Co).astype(np.float32)
    mask = rng.randn(B, IH, IW, Ci).astype(np.float32)

    # reference in float64 on the values the kernel reads (frames stay fp32 in HBM but are staged as T)
    xr = rounded(x[idx] if first else x, td).requires_grad_(True)
    wr = rounded(w, td).requires_grad_(True)
    y = F.conv2d(_nchw(xr), wr.permute(3, 2, 0, 1), torch.from_numpy(b).double(), stride=2)
    yref = _nhwc(F.relu(y))
    dyr = rounded(dy, td)
    y.backward(_nchw(dyr))
    dxref = xr.grad * (rounded(mask, td) > 0)
    dwref = wr.grad

    xd = dev(x, torch.float32 if first else td)
    wd, bd = dev(w, td), dev(b)
    out = alloc(td, B, OH, OW, Co)
    L.mi_conv2d_nhwc_fwd(stream(), code, xd.data_ptr(), P(dev(idx, torch.int32)) if first else None, int(first),
                         B, IH, IW, Ci, wd.data_ptr(), 0, bd.data_ptr(), k, k, Co, 1, out.data_ptr())
    rt, at = tols(dt, float(yref.abs().max()))
    assert_close(host(out), yref.detach().numpy(), rt, at, "conv fwd")
    # K-contiguous kernel copy made by mi_transpose_weights: same result through the conflict-free B staging
    wt = alloc(td, k * k * Ci * Co, fill=0.0)
    offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Ci], np.int32), np.array([Co], np.int32)
    L.mi_transpose_weights(stream(), code, P(dev(w)), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
    if td is X3:
        assert np.array_equal(host(wt).reshape(Co, k * k * Ci), rounded(w.reshape(-1, Co).T.copy(), td).numpy())
    else:
        assert torch.equal(wt.view(Co, k * k * Ci).cpu(), torch.from_numpy(w.reshape(-1, Co).T.copy()).to(td))
    out2 = torch.empty_like(out)
    L.mi_conv2d_nhwc_fwd(stream(), code, xd.data_ptr(), P(dev(idx, torch.int32)) if first else None, int(first),
                         B, IH, IW, Ci, wt.data_ptr(), 1, bd.data_ptr(), k, k, Co, 1, out2.data_ptr())
    assert_close(host(out2), yref.detach().numpy(), rt, at, "conv fwd (transposed kernel)")

    dyd = dev(dy, td)
    dw = torch.zeros(k, k, Ci, Co, device="cuda")
    idxd = dev(idx, torch.int32) if first else None
    L.mi_conv2d_nhwc_wgrad(stream(), code, xd.data_ptr(), idxd.data_ptr() if first else None, int(first), B, IH, IW, Ci,
                           dyd.data_ptr(), k, k, Co, dw.data_ptr())
    rt, at = tols("f32" if dt != "bf16" else "bf16", float(dwref.abs().max()))
    assert_close(host(dw), dwref.numpy(), rt if dt != "bf16" else 1e-4, at if dt != "bf16" else 1e-4 * float(dwref.abs().max()), "conv wgrad")

    if not first:                                        # conv1's input gradient is never needed
        dx = alloc(td, B, IH, IW, Ci, fill=7.0)
        L.mi_conv2d_nhwc_dgrad(stream(), code, dyd.data_ptr(), B, OH, OW, Co, wd.data_ptr(), k, k, Ci, IH, IW,
                               P(dev(mask, td)), dx.data_ptr())
        rt, at = tols(dt, float(dxref.abs().max()))
        assert_close(host(dx), dxref.numpy(), rt, at, "conv dgrad")


@pytest.mark.parametrize("blocks_per_xcd", [1, 2, 0])
@pytest.mark.parametrize("k,B,wide,hw", [(5, 7, 0, None), (4, 7, 0, None), (5, 20, 0, None), (4, 20, 1, None), (4, 50, 1, None),
                                         (5, 40, 0, (23, 47)), (4, 40, 0, (20, 44)), (4, 60, 1, (14, 40))])
def test_conv_form_register_weight_kernel_walks_runs_of_chunks(k, B, wide, hw, blocks_per_xcd):
    """rwconv_conv_kernel (32 -> 64 channels, or 64 -> 128 with wide = 1; stride 2): a block walks CONSECUTIVE 128- (64-) position chunks whose slot rows are
    staged in runs of three instalments; 8 or 16 blocks over 44 / 125 (54 / 134) chunks make every block cross run boundaries, start runs at every phase
    and end on every instalment; three more image sizes change the slot grid and the staged halo.  Checked against the float64 convolution of the same bf16
    values, with and without bias + ReLU / ReluGrad mask."""
    L = milib.get()
    code, td = DT["bf16"]
    prev = {key: L.mi_set_tuning(key, v) for key, v in ((13, 2), (15, 3), (16, blocks_per_xcd))}
    try:
        rng = np.random.RandomState(10 * k + B)
        IH, IW, Ci, Co = (18, 38, 64, 128) if wide else (39, 79, 32, 64)
        if hw:                                          # other slot grids: GW = 24 / 22 / 20, staged halos of 64 / 32 / 32 rows
            IH, IW = hw
        OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
        x = rng.randn(B, IH, IW, Ci).astype(np.float32)
        w = (rng.randn(k, k, Ci, Co) / np.sqrt(k * k * Ci)).astype(np.float32)
        b = (0.1 * rng.randn(Co)).astype(np.float32)
        mask = rng.randn(B, OH, OW, Co).astype(np.float32)
        y = F.conv2d(_nchw(rounded(x, td)), rounded(w, td).permute(3, 2, 0, 1), torch.from_numpy(b).double(), stride=2)
        wt = alloc(td, k * k * Ci * Co, fill=0.0)
        offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Ci], np.int32), np.array([Co], np.int32)
        L.mi_transpose_weights(stream(), code, P(dev(w)), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
        xd, bd = dev(x, td), dev(b)
        out = alloc(td, B, OH, OW, Co, fill=3.0)
        L.mi_conv2d_nhwc_fwd(stream(), code, xd.data_ptr(), None, 0, B, IH, IW, Ci, wt.data_ptr(), 1, bd.data_ptr(), k, k, Co, 1, out.data_ptr())
        yref = _nhwc(F.relu(y)).numpy()
        rt, at = tols("bf16", float(np.abs(yref).max()))
        assert_close(host(out), yref, rt, at, "conv-form fwd, bias + relu")
        # the same contraction as the input gradient of a transposed conv (deconv3.dgrad: no bias, ReluGrad mask): dY = x, weights [Cin_of_deconv = 64][k k 32]
        dx = alloc(td, B, OH, OW, Co, fill=3.0)
        L.mi_deconv2d_nhwc_dgrad(stream(), code, xd.data_ptr(), B, IH, IW, Ci, wt.data_ptr(), 1, k, k, Co, P(dev(mask, td)), dx.data_ptr())
        y0 = F.conv2d(_nchw(rounded(x, td)), rounded(w, td).permute(3, 2, 0, 1), None, stride=2)
        dref = (_nhwc(y0) * (rounded(mask, td) > 0)).numpy()
        rt, at = tols("bf16", float(np.abs(dref).max()))
        assert_close(host(dx), dref, rt, at, "conv-form as a deconv input gradient, mask")
    finally:
        for key, v in prev.items():
            L.mi_set_tuning(key, v)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("geom", DECONVS)
def test_deconv_fwd_dgrad_wgrad(dt, geom, kernels):
    if kernels == "rwconv" and dt != "bf16":
        pytest.skip("the register-weight kernel is bf16 only")
    L = milib.get()
    code, td = DT[dt]
    IH, IW, Ci, Co, k = geom
    B = 3
    rng = np.random.RandomState(Co + k)
    x = rng.randn(B, IH, IW, Ci).astype(np.float32)
    w = (rng.randn(k, k, Co, Ci) / np.sqrt(k * k * Ci / 4)).astype(np.float32)          # [kh,kw,out,in]
    b = (0.1 * rng.randn(Co)).astype(np.float32)
    OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
    dy = rng.randn(B, OH, OW, Co).astype(np.float32)
    mask = rng.randn(B, IH, IW, Ci).astype(np.float32)

    xr = rounded(x, td).requires_grad_(True)
    wr = rounded(w, td).requires_grad_(True)
    y = F.conv_transpose2d(_nchw(xr), wr.permute(3, 2, 0, 1), torch.from_numpy(b).double(), stride=2)
    assert y.shape[2:] == (OH, OW)
    relu = Co > 3
    yref = _nhwc(F.relu(y) if relu else y)
    y.backward(_nchw(rounded(dy, td)))
    dxref = xr.grad * (rounded(mask, td) > 0)
    dwref = wr.grad

    xd, wd, bd = dev(x, td), dev(w, td), dev(b)
    out = alloc(td, B, OH, OW, Co, fill=9.0)
    L.mi_deconv2d_nhwc_fwd(stream(), code, xd.data_ptr(), B, IH, IW, Ci, wd.data_ptr(), bd.data_ptr(), k, k, Co, int(relu), out.data_ptr())
    rt, at = tols(dt, float(yref.abs().max()))
    assert_close(host(out), yref.detach().numpy(), rt, at, "deconv fwd")

    dyd = dev(dy, td)
    dx = alloc(td, B, IH, IW, Ci, fill=5.0)
    L.mi_deconv2d_nhwc_dgrad(stream(), code, dyd.data_ptr(), B, OH, OW, Co, wd.data_ptr(), 0, k, k, Ci, P(dev(mask, td)), dx.data_ptr())
    rt, at = tols(dt, float(dxref.abs().max()))
    assert_close(host(dx), dxref.numpy(), rt, at, "deconv dgrad")
    wt = alloc(td, k * k * Co * Ci, fill=0.0)
    offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Co], np.int32), np.array([Ci], np.int32)
    L.mi_transpose_weights(stream(), code, P(dev(w)), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
    dx2 = alloc(td, B, IH, IW, Ci, fill=5.0)
    L.mi_deconv2d_nhwc_dgrad(stream(), code, dyd.data_ptr(), B, OH, OW, Co, wt.data_ptr(), 1, k, k, Ci, P(dev(mask, td)), dx2.data_ptr())
    assert_close(host(dx2), dxref.numpy(), rt, at, "deconv dgrad (transposed kernel)")

    dw = torch.zeros(k, k, Co, Ci, device="cuda")
    L.mi_deconv2d_nhwc_wgrad(stream(), code, dyd.data_ptr(), B, OH, OW, Co, xd.data_ptr(), k, k, Ci, dw.data_ptr())
    s = float(dwref.abs().max())
    assert_close(host(dw), dwref.numpy(), 1e-5 if dt != "bf16" else 1e-4, (2e-5 if dt != "bf16" else 1e-4) * s, "deconv wgrad")


@pytest.mark.parametrize("B", [1, 3, 8, 37, 512])
def test_activation_resident_small_grid_layers(B):
    """Round 4 (csrc/ares_tile.hpp): conv4 forward / deconv1 input gradient (conv form) and deconv1 forward / conv4 input gradient (gather
    form) with the frames of a group resident in LDS and fragment-ordered weights streamed through registers, against float64 statements of the four ops on the
    bf16-rounded operands (bias + ReLU / ReluGrad mask as the model uses them) -- at batch sizes below, at and across the frame-group sizes (4 / 8), with the
    last group ragged, and at the benchmarked 512 -- and bit for bit against the general kernels' results where both run the same products in fp32."""
    L = milib.get()
    code, td = DT["bf16"]
    rng = np.random.RandomState(B)
    nb = int(L.mi_ares_weight_bytes())
    assert nb == 16 * 128 * 256 * 2
    # ---- conv form: [B,8,18,128] -> [B,3,8,256] ----
    x = rng.randn(B, 8, 18, 128).astype(np.float32)
    w = (rng.randn(4, 4, 128, 256) / np.sqrt(16 * 128)).astype(np.float32)          # HWIO (conv4) == [kh,kw,out=128,in=256] (deconv1) as stored
    bias = (0.1 * rng.randn(256)).astype(np.float32)
    mask = rng.randn(B, 3, 8, 256).astype(np.float32)
    y0 = _nhwc(F.conv2d(_nchw(rounded(x, td)), rounded(w, td).permute(3, 2, 0, 1), None, stride=2))
    wf = torch.empty(nb, device="cuda", dtype=torch.uint8)
    L.mi_ares_pack_weights(stream(), 0, P(dev(w)), wf.data_ptr())
    xd = dev(x, td)
    launched = np.zeros(1, np.int32)
    out = alloc(td, B, 3, 8, 256, fill=7.0)
    L.mi_ares_conv(stream(), code, 0, xd.data_ptr(), B, wf.data_ptr(), P(dev(bias)), 1, None, out.data_ptr(), launched.ctypes.data)
    assert launched[0] == 1
    ref = F.relu(y0 + torch.from_numpy(bias).double()).numpy()
    rt, at = tols("bf16", float(np.abs(ref).max()))
    assert_close(host(out), ref, rt, at, "conv form: conv4 forward (bias + relu)")
    out2 = alloc(td, B, 3, 8, 256, fill=7.0)
    L.mi_ares_conv(stream(), code, 0, xd.data_ptr(), B, wf.data_ptr(), None, 0, P(dev(mask, td)), out2.data_ptr(), launched.ctypes.data)
    ref2 = (y0 * (rounded(mask, td) > 0)).numpy()
    rt, at = tols("bf16", float(np.abs(ref2).max()))
    assert_close(host(out2), ref2, rt, at, "conv form: input gradient of a transposed conv (mask)")
    # ---- gather form: [B,3,8,256] -> [B,8,18,128] ----
    xg = rng.randn(B, 3, 8, 256).astype(np.float32)
    wg = (rng.randn(4, 4, 128, 256) / np.sqrt(4 * 256)).astype(np.float32)          # [kh,kw,out=128,in=256] (deconv1) == HWIO [kh,kw,ci=128,co=256] (conv4)
    biasg = (0.1 * rng.randn(128)).astype(np.float32)
    maskg = rng.randn(B, 8, 18, 128).astype(np.float32)
    yg = _nhwc(F.conv_transpose2d(_nchw(rounded(xg, td)), rounded(wg, td).permute(3, 2, 0, 1), None, stride=2))
    assert tuple(yg.shape) == (B, 8, 18, 128)
    wfg = torch.empty(nb, device="cuda", dtype=torch.uint8)
    L.mi_ares_pack_weights(stream(), 1, P(dev(wg)), wfg.data_ptr())
    xgd = dev(xg, td)
    og = alloc(td, B, 8, 18, 128, fill=7.0)
    L.mi_ares_conv(stream(), code, 1, xgd.data_ptr(), B, wfg.data_ptr(), P(dev(biasg)), 1, None, og.data_ptr(), launched.ctypes.data)
    assert launched[0] == 1
    refg = F.relu(yg + torch.from_numpy(biasg).double()).numpy()
    rt, at = tols("bf16", float(np.abs(refg).max()))
    assert_close(host(og), refg, rt, at, "gather form: deconv1 forward (bias + relu)")
    og2 = alloc(td, B, 8, 18, 128, fill=7.0)
    L.mi_ares_conv(stream(), code, 1, xgd.data_ptr(), B, wfg.data_ptr(), None, 0, P(dev(maskg, td)), og2.data_ptr(), launched.ctypes.data)
    refg2 = (yg * (rounded(maskg, td) > 0)).numpy()
    rt, at = tols("bf16", float(np.abs(refg2).max()))
    assert_close(host(og2), refg2, rt, at, "gather form: input gradient of a conv (mask)")
    # ---- gather form, mid layer: [B,8,18,128] -> [B,18,38,64] (deconv2 forward / conv3's input gradient) ----
    if B <= 64:
        xm = rng.randn(B, 8, 18, 128).astype(np.float32)
        wm = (rng.randn(4, 4, 64, 128) / np.sqrt(4 * 128)).astype(np.float32)       # [kh,kw,out=64,in=128] (deconv2) == HWIO [kh,kw,ci=64,co=128] (conv3)
        biasm = (0.1 * rng.randn(64)).astype(np.float32)
        maskm = rng.randn(B, 18, 38, 64).astype(np.float32)
        ym = _nhwc(F.conv_transpose2d(_nchw(rounded(xm, td)), rounded(wm, td).permute(3, 2, 0, 1), None, stride=2))
        assert tuple(ym.shape) == (B, 18, 38, 64)
        wfm = torch.empty(nb, device="cuda", dtype=torch.uint8)
        L.mi_ares_pack_weights(stream(), 2, P(dev(wm)), wfm.data_ptr())
        xmd = dev(xm, td)
        om = alloc(td, B, 18, 38, 64, fill=7.0)
        L.mi_ares_conv(stream(), code, 2, xmd.data_ptr(), B, wfm.data_ptr(), P(dev(biasm)), 1, None, om.data_ptr(), launched.ctypes.data)
        assert launched[0] == 1
        refm = F.relu(ym + torch.from_numpy(biasm).double()).numpy()
        rt, at = tols("bf16", float(np.abs(refm).max()))
        assert_close(host(om), refm, rt, at, "mid gather form: deconv2 forward (bias + relu)")
        om2 = alloc(td, B, 18, 38, 64, fill=7.0)
        L.mi_ares_conv(stream(), code, 2, xmd.data_ptr(), B, wfm.data_ptr(), None, 0, P(dev(maskm, td)), om2.data_ptr(), launched.ctypes.data)
        refm2 = (ym * (rounded(maskm, td) > 0)).numpy()
        rt, at = tols("bf16", float(np.abs(refm2).max()))
        assert_close(host(om2), refm2, rt, at, "mid gather form: input gradient of conv3 (mask)")
    # not eligible: other storage types -> nothing launched, the caller takes the general kernels
    L.mi_ares_conv(stream(), DT["f32"][0], 0, xd.data_ptr(), B, wf.data_ptr(), None, 0, None, out.data_ptr(), launched.ctypes.data)
    assert launched[0] == 0


@pytest.mark.parametrize("dt", DTS)
def test_conv_dgrad_into_larger_input(dt, kernels):
    """conv2 reads a 39x79 map but its VALID s2 windows never touch the last row/col: their gradient must be 0."""
    L = milib.get()
    code, td = DT[dt]
    rng = np.random.RandomState(0)
    B, IH, IW, Ci, Co, k = 2, 39, 79, 32, 64, 4
    dy = rng.randn(B, 18, 38, Co).astype(np.float32)
    w = rng.randn(k, k, Ci, Co).astype(np.float32) * 0.05
    dx = alloc(td, B, IH, IW, Ci, fill=3.0)
    L.mi_conv2d_nhwc_dgrad(stream(), code, P(dev(dy, td)), B, 18, 38, Co, P(dev(w, td)), k, k, Ci, IH, IW, None, dx.data_ptr())
    g = host(dx)
    assert (g[:, 38] == 0).all() and (g[:, :, 78] == 0).all() and np.abs(g[:, :38, :78]).max() > 0


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(5, 6144, 128, 0, 12), (5, 64, 6144, 0, 1), (5, 128, 6144, 1, 1), (7, 6144, 64, 1, 8),
                                   (32, 72, 500, 0, 1), (130, 500, 300, 0, 1), (32, 300, 2, 0, 1), (32, 300, 1, 0, 1),
                                   (33, 304, 504, 1, 1)])
def test_dense_gemm_variants(dt, shape):
    L = milib.get()
    code, td = DT[dt]
    M, K, N, layout, nsplit = shape
    if dt == "bf16" and (K % 8):
        pytest.skip("bf16 vectors need K % 8 == 0")
    rng = np.random.RandomState(M + N)
    a = rng.randn(M, K).astype(np.float32)
    w = (rng.randn(K, N) / np.sqrt(K)).astype(np.float32)
    b = rng.randn(N).astype(np.float32)
    mask = rng.randn(M, N).astype(np.float32)
    ref = rounded(a, td) @ rounded(w, td)
    wd = dev(w if layout == 0 else np.ascontiguousarray(w.T), td)
    if nsplit > 1:
        out = torch.full((nsplit, M, N), 1.0, device="cuda")
        L.mi_gemm_bias_act(stream(), code, P(dev(a, td)), M, K, wd.data_ptr(), layout, N, None, 0, None, out.data_ptr(), 1, nsplit)
        got = host(out).sum(0)
        assert_close(got, ref.numpy(), 1e-5, 3e-5 * float(ref.abs().max()), "split-K gemm")
        return
    ref = ref + torch.from_numpy(b).double()
    ref = F.relu(ref) * (rounded(mask, td) > 0)
    out = alloc(td, M, N)
    L.mi_gemm_bias_act(stream(), code, P(dev(a, td)), M, K, wd.data_ptr(), layout, N, P(dev(b)), 1, P(dev(mask, td)), out.data_ptr(), 0, 1)
    rt, at = tols(dt, float(ref.abs().max()))
    assert_close(host(out), ref.numpy(), rt, at, "gemm+bias+relu+mask")
    # fp32 output regardless of storage type
    out32 = torch.empty(M, N, device="cuda")
    L.mi_gemm_bias_act(stream(), code, P(dev(a, td)), M, K, wd.data_ptr(), layout, N, P(dev(b)), 0, None, out32.data_ptr(), 1, 1)
    ref2 = rounded(a, td) @ rounded(w, td) + torch.from_numpy(b).double()
    assert_close(host(out32), ref2.numpy(), 1e-5, 3e-5 * float(ref2.abs().max()), "gemm f32 out")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(6, 6144, 128), (6, 64, 6144), (40, 72, 500), (40, 500, 300), (40, 304, 2), (40, 304, 1), (2100, 64, 64)])
def test_dense_wgrad(dt, shape):
    L = milib.get()
    code, td = DT[dt]
    M, K, N = shape
    if dt == "bf16" and (K % 8):
        pytest.skip("bf16 vectors need K % 8 == 0 (callers pad K)")
    rng = np.random.RandomState(K + N)
    a, dy = rng.randn(M, K).astype(np.float32), rng.randn(M, N).astype(np.float32)
    ref = rounded(a, td).T @ rounded(dy, td)
    dw = torch.zeros(K, N, device="cuda")
    L.mi_gemm_wgrad(stream(), code, P(dev(a, td)), P(dev(dy, td)), M, K, N, dw.data_ptr())
    assert_close(host(dw), ref.numpy(), 1e-5, 3e-5 * float(ref.abs().max()), "dense wgrad")
    L.mi_gemm_wgrad(stream(), code, P(dev(a, td)), P(dev(dy, td)), M, K, N, dw.data_ptr())      # accumulates
    assert_close(host(dw), 2 * ref.numpy(), 1e-5, 6e-5 * float(ref.abs().max()), "dense wgrad accumulate")


@pytest.mark.parametrize("shape", [(512, 4096, 1024), (77, 512, 8192), (33, 4096, 1024)])
def test_dense_wgrad_whole_tiles_over_all_rows(shape):
    """Round 4 (dwg_tile.hpp, the MlpVAE's two large layers): storing dense filter gradient with >= 256 tiles of 128 x 128 -- LDS-DMA stages of 32 rows, both operands
    read through the hardware transpose from swizzled rows, bias gradient by a ones operand -- against float64; both tile orders (k-major / n-major), a ragged last
    stage (rows past M are zero-filled by the descriptor), garbage in the output buffers beforehand, and two runs bitwise equal."""
    L = milib.get()
    code, td = DT["bf16"]
    M, K, N = shape
    rng = np.random.RandomState(M + K)
    a, dy = rng.randn(M, K).astype(np.float32), rng.randn(M, N).astype(np.float32)
    ref = (rounded(a, td).T @ rounded(dy, td)).numpy()
    refb = rounded(dy, td).sum(0).numpy()
    ad, dyd = dev(a, td), dev(dy, td)
    outs = []
    for fill in (3.0, -7.0):
        dw, db = torch.full((K, N), fill, device="cuda"), torch.full((N,), fill, device="cuda")
        L.mi_gemm_wgrad_bias_set(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), None, 0, 1)      # (no scratch: only the tile kernel can do this)
        outs.append((host(dw), host(db)))
    assert_close(outs[0][0], ref, 1e-5, 3e-5 * float(np.abs(ref).max()), "dense wgrad (whole tiles)")
    assert_close(outs[0][1], refb, 1e-5, 3e-5 * float(np.abs(refb).max()), "bias row (ones operand)")
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    dw = torch.full((K, N), 5.0, device="cuda")
    L.mi_gemm_wgrad_bias_set(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), None, None, 0, 1)                   # without the bias row
    assert np.array_equal(host(dw), outs[0][0])


@pytest.mark.parametrize("shape", [(512, 512, 256), (512, 256, 128), (512, 64, 256), (16, 64, 64), (48, 128, 192), (80, 512, 256), (2048, 256, 128), (32, 256, 512)])
def test_dense_wgrad_one_wave_per_tile_without_lds(shape):
    """Round 5 (dwgs_tile.hpp): the dense filter gradient without operand staging -- one block of 1 / 2 / 4 waves per 64 x 64 tile of dW over ALL rows
    (the waves split the rows and meet in a fixed order through a 16 KB LDS tile), fragments gathered as column-pair dwords and split by v_perm_b32, four steps of loads in
    flight, the bias row as a ones operand -- against float64 and against the first-generation kernel (tuning key 22 off); the MlpVAE's small layers at batch 512, row
    counts that give 1 / 2 / 4 waves and tail steps, a long reduction; storing form over garbage and adding form onto a previous result; no scratch needed; two runs
    bitwise equal."""
    L = milib.get()
    code, td = DT["bf16"]
    M, K, N = shape
    rng = np.random.RandomState(M + K + N)
    a, dy = rng.randn(M, K).astype(np.float32), rng.randn(M, N).astype(np.float32)
    ref = (rounded(a, td).T @ rounded(dy, td)).numpy()
    refb = rounded(dy, td).sum(0).numpy()
    ad, dyd = dev(a, td), dev(dy, td)
    outs = []
    for fill in (3.0, -7.0):
        dw, db = torch.full((K, N), fill, device="cuda"), torch.full((N,), fill, device="cuda")
        L.mi_gemm_wgrad_bias_set(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), None, 0, 1)      # storing form, no scratch
        outs.append((host(dw), host(db)))
    tol = 3e-5 * float(np.abs(ref).max()) * max(1.0, (M / 512.0) ** 0.5)
    assert_close(outs[0][0], ref, 1e-5, tol, "dense wgrad (one wave per tile)")
    assert_close(outs[0][1], refb, 1e-5, 3e-5 * float(np.abs(refb).max()) * max(1.0, (M / 512.0) ** 0.5), "bias row (ones operand)")
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    # adding form: onto the previous result, with and without the bias row
    w32, b32 = outs[0][0].astype(np.float32), outs[0][1].astype(np.float32)
    dw, db = torch.from_numpy(w32).cuda(), torch.from_numpy(b32).cuda()
    L.mi_gemm_wgrad_bias_ws(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), None, 0)
    assert np.array_equal(host(dw).astype(np.float32), w32 + w32) and np.array_equal(host(db).astype(np.float32), b32 + b32)
    L.mi_gemm_wgrad_ws(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), None, 0)
    assert np.array_equal(host(dw).astype(np.float32), (w32 + w32) + w32) and np.array_equal(host(db).astype(np.float32), b32 + b32)
    # with caller scratch: the same kernel (the scratch is not needed); storing and adding forms, twice (bitwise equal)
    nbs = int(L.mi_gemm_wgrad_scratch_bytes(code, M, K, N))
    wss = torch.empty(max(nbs, 256), device="cuda", dtype=torch.uint8)
    split_runs = []
    for fill in (11.0, -2.0):
        dw, db = torch.full((K, N), fill, device="cuda"), torch.full((N,), fill, device="cuda")
        L.mi_gemm_wgrad_bias_set(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), wss.data_ptr(), nbs, 1)
        split_runs.append((host(dw), host(db)))
    assert_close(split_runs[0][0], ref, 1e-5, tol, "dense wgrad (row splits + ordered sum)")
    assert_close(split_runs[0][1], refb, 1e-5, 3e-5 * float(np.abs(refb).max()) * max(1.0, (M / 512.0) ** 0.5), "bias row (row splits)")
    assert np.array_equal(split_runs[0][0], split_runs[1][0]) and np.array_equal(split_runs[0][1], split_runs[1][1])
    dw, db = torch.from_numpy(w32).cuda(), torch.from_numpy(b32).cuda()
    L.mi_gemm_wgrad_bias_ws(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), wss.data_ptr(), nbs)
    assert_close(host(dw), 2 * ref, 1e-5, 2 * tol, "adding form with row splits")
    # the first-generation kernel on the same operands (same bf16 products, fp32 sums in another order)
    prev = L.mi_set_tuning(22, 0)
    try:
        nb = int(L.mi_gemm_wgrad_scratch_bytes(code, M, K, N))
        ws = torch.empty(max(nb, 256), device="cuda", dtype=torch.uint8)
        dw1, db1 = torch.zeros(K, N, device="cuda"), torch.zeros(N, device="cuda")
        L.mi_gemm_wgrad_bias_ws(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw1.data_ptr(), db1.data_ptr(), ws.data_ptr(), nb)
        assert_close(host(dw1), outs[0][0], 1e-5, tol, "one wave per tile vs first generation")
        assert_close(host(db1), outs[0][1], 1e-5, tol, "bias row vs first generation")
    finally:
        L.mi_set_tuning(22, prev)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_adam_writing_both_weight_layouts_equals_adam_then_transposes(dt):
    """Round 4 (MlpVAE engine): mi_adam_tf_layouts = mi_adam_tf_flat + mi_transpose_weights in one launch -- p / m / v, the storage-type copy and the K-contiguous
    kernel copies BITWISE equal to the two-pass form; kernels with partial 64 x 64 tiles, a K that is not a multiple of the 16-byte vector, biases between them and a tail."""
    L = milib.get()
    code, td = DT[dt]
    rng = np.random.RandomState(3)
    kern = [(100, 72), (64, 128), (8, 260), (130, 64)]                         # (K, N): N % 4 == 0
    offs, o = [], 0
    for K, N in kern:
        offs.append(o); o += K * N + N                                      # kernel, then its bias
    n = o + 12                                                               # + a tail outside every kernel
    p0 = rng.randn(n).astype(np.float32); g0 = (rng.randn(n) * 0.1).astype(np.float32)
    m0 = (rng.randn(n) * 0.01).astype(np.float32); v0 = (rng.rand(n) * 1e-3).astype(np.float32)
    off_a, K_a, N_a = np.array(offs, np.int64), np.array([k for k, _ in kern], np.int32), np.array([nn for _, nn in kern], np.int32)
    alpha = 1e-3
    skip = np.array([0, 1, 2, 3], np.int32)                                   # kernel 1: no shadow copy, 2: no K-contiguous copy, 3: neither
    def run(fused, clear, use_skip=False):
        p, m, v, g = dev(p0), dev(m0), dev(v0), dev(g0)
        sh = torch.zeros(n, device="cuda", dtype=td) if dt == "bf16" else None
        wt = torch.zeros(n, device="cuda", dtype=td)
        if fused:
            L.mi_adam_tf_layouts(stream(), code, p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), n, off_a.ctypes.data, K_a.ctypes.data, N_a.ctypes.data, skip.ctypes.data if use_skip else None, len(kern),
                                 alpha, None, 0.9, 0.999, 1e-8, sh.data_ptr() if sh is not None else None, wt.data_ptr(), clear)
        else:
            L.mi_adam_tf_flat(stream(), p.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), n, alpha, 0.9, 0.999, 1e-8, sh.data_ptr() if sh is not None else None, clear)
            L.mi_transpose_weights(stream(), code, p.data_ptr(), wt.data_ptr(), off_a.ctypes.data, K_a.ctypes.data, N_a.ctypes.data, len(kern))
        torch.cuda.synchronize()
        out = [host(p), host(m), host(v), host(g), wt.view(torch.int16 if dt == "bf16" else torch.int32).cpu().numpy()]
        if sh is not None:
            out.append(sh.view(torch.int16).cpu().numpy())
        return out
    for clear in (0, 1):
        a_, b_ = run(True, clear), run(False, clear)
        for i, (x, y) in enumerate(zip(a_, b_)):
            if i == 4:                                                       # the transposed copy: only the kernels' ranges are defined
                for o_, (K, N) in zip(offs, kern):
                    assert np.array_equal(x[o_:o_ + K * N], y[o_:o_ + K * N]), ("wt", K, N)
            else:
                assert np.array_equal(x, y), i
    assert not np.array_equal(a_[0], p0)
    # skipped copies stay untouched (zeros), everything else is as before
    a_, b_ = run(True, 0, use_skip=True), run(False, 0)
    assert all(np.array_equal(a_[i], b_[i]) for i in range(4))
    for i_, (o_, (K, N)) in enumerate(zip(offs, kern)):
        sl = slice(o_, o_ + K * N)
        assert np.array_equal(a_[4][sl], b_[4][sl]) if not (skip[i_] & 2) else not a_[4][sl].any()
        if dt == "bf16":
            assert np.array_equal(a_[5][sl], b_[5][sl]) if not (skip[i_] & 1) else not a_[5][sl].any()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_gather_rows_cast(dt):
    """mi_gather_rows_cast: rows idx[b] of a float32 table in the engine's storage type (the MlpVAE engine's frame staging), incl. a row length with a scalar tail."""
    L = milib.get()
    code, td = DT[dt]
    rng = np.random.RandomState(1)
    for row_len in (38400, 2052, 7):
        tab = rng.rand(9, row_len).astype(np.float32)
        idx = np.array([4, 0, 8, 8, 2], np.int32)
        tab_d, idx_d = dev(tab), torch.from_numpy(idx).cuda()
        out = torch.zeros(len(idx), row_len, device="cuda", dtype=td)
        L.mi_gather_rows_cast(stream(), code, tab_d.data_ptr(), idx_d.data_ptr(), len(idx), row_len, out.data_ptr())
        assert np.array_equal(host(out), rounded(tab[idx], td).numpy())
        out2 = torch.zeros(3, row_len, device="cuda", dtype=td)
        L.mi_gather_rows_cast(stream(), code, tab_d.data_ptr(), None, 3, row_len, out2.data_ptr())
        assert np.array_equal(host(out2), rounded(tab[:3], td).numpy())


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(512, 64, 6144), (512, 6144, 128), (40, 72, 500), (2100, 64, 64), (300, 304, 1)])
def test_ordered_dense_wgrad_with_bias_row_and_colsum(dt, shape):
    """Round 4: mi_gemm_wgrad_bias_ws (row splits through per-split slabs + ONE ordered reduce; the layer's BiasAddGrad as one more row of the same product:
    a column of ones appended to `a` in the kernel's loader) and mi_colsum_ws (per-block column sums + ordered reduce) against float64 -- and run three times
    into zeroed buffers: bitwise identical results (no fp32 atomics on this path).  Shapes: the latent layers at batch 512, the PPO trunk, odd row counts."""
    L = milib.get()
    code, td = DT[dt]
    M, K, N = shape
    if dt == "bf16" and (K % 8):
        pytest.skip("bf16 vectors need K % 8 == 0 (callers pad K)")
    rng = np.random.RandomState(K + N)
    a, dy = rng.randn(M, K).astype(np.float32), rng.randn(M, N).astype(np.float32)
    ref = rounded(a, td).T @ rounded(dy, td)
    refb = rounded(dy, td).sum(0)
    ad, dyd = dev(a, td), dev(dy, td)
    nb = int(L.mi_gemm_wgrad_scratch_bytes(code, M, K, N))
    nc = int(L.mi_colsum_scratch_bytes(code, M, N))
    assert nb < (64 << 20) and nc < (8 << 20)
    ws = torch.empty(max(nb, 256), device="cuda", dtype=torch.uint8)
    wc = torch.empty(max(nc, 256), device="cuda", dtype=torch.uint8)
    runs = []
    for _ in range(3):
        dw, db, db2 = torch.zeros(K, N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        L.mi_gemm_wgrad_bias_ws(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb)
        L.mi_colsum_ws(stream(), code, dyd.data_ptr(), M, N, db2.data_ptr(), wc.data_ptr(), nc)
        torch.cuda.synchronize()
        runs.append((dw.cpu().numpy().copy(), db.cpu().numpy().copy(), db2.cpu().numpy().copy()))
    assert_close(runs[0][0], ref.numpy(), 1e-5, 3e-5 * float(ref.abs().max()), "dense wgrad (slabs)")
    assert_close(runs[0][1], refb.numpy(), 1e-5, 3e-5 * float(refb.abs().max()), "bias row")
    assert_close(runs[0][2], refb.numpy(), 1e-5, 3e-5 * float(refb.abs().max()), "colsum (ordered)")
    for r in runs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(r, runs[0]))
    # accumulating call without a bias row; dbias = NULL leaves nothing behind the filter rows
    dw = torch.from_numpy(runs[0][0]).cuda()
    L.mi_gemm_wgrad_ws(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), ws.data_ptr(), nb)
    assert_close(host(dw), 2 * ref.numpy(), 1e-5, 6e-5 * float(ref.abs().max()), "dense wgrad accumulate")
    # storing form (the MlpVAE engine): garbage in the buffers, the same bits out as the first call into zeros -- plain stores (one row split) or the storing ordered sum
    dw, db = torch.full((K, N), 7.5, device="cuda"), torch.full((N,), -3.25, device="cuda")
    L.mi_gemm_wgrad_bias_set(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, 1)
    assert np.array_equal(host(dw), runs[0][0]) and np.array_equal(host(db), runs[0][1])
    lds_free = dt == "bf16" and M % 16 == 0 and K % 64 == 0 and N % 64 == 0 and K * N <= 262144        # round 5: whole tiles per block over all rows (dwgs_tile.hpp) -- no scratch
    if nb > 0 and not lds_free:                         # row splits without scratch cannot store: refused, not silently accumulated
        with pytest.raises(milib.MiError):
            L.mi_gemm_wgrad_bias_set(stream(), code, ad.data_ptr(), dyd.data_ptr(), M, K, N, dw.data_ptr(), db.data_ptr(), None, 0, 1)


@pytest.mark.parametrize("dt", DTS)
def test_reparam_kl_fwd_bwd(dt):
    L = milib.get()
    code, td = DT[dt]
    rng = np.random.RandomState(0)
    B, Z, ns = 37, 64, 3
    heads = rng.randn(ns, B, 2 * Z).astype(np.float32) * 0.3
    bm, bl = rng.randn(Z).astype(np.float32) * 0.1, rng.randn(Z).astype(np.float32) * 0.1
    eps = rng.randn(B, Z).astype(np.float32)
    dzs = rng.randn(2, B, Z).astype(np.float32)
    beta, tol = 1.5, 0.0
    h = torch.from_numpy(heads).double().sum(0)
    mu = (h[:, :Z] + torch.from_numpy(bm).double()).requires_grad_(True)
    lv = (h[:, Z:] + torch.from_numpy(bl).double()).requires_grad_(True)
    z = mu + torch.exp(0.5 * lv) * torch.from_numpy(eps).double()
    kl = -0.5 * (1 + lv - mu * mu - lv.exp()).sum(1)
    loss = (z * torch.from_numpy(dzs).double().sum(0)).sum() + beta * kl.mean()
    loss.backward()
    mean = torch.empty(B, Z, device="cuda"); logvar = torch.empty(B, Z, device="cuda")
    zd = alloc(td, B, Z); klr = torch.empty(B, device="cuda")
    epsd = dev(eps)
    L.mi_vae_reparam_kl_fwd(stream(), code, P(dev(heads)), ns, P(dev(bm)), P(dev(bl)), epsd.data_ptr(), 1, B, Z,
                            mean.data_ptr(), logvar.data_ptr(), zd.data_ptr(), klr.data_ptr())
    assert_close(host(mean), mu.detach().numpy(), 1e-6, 1e-6, "mean")
    assert_close(host(klr), kl.detach().numpy(), 1e-5, 1e-5, "kl rows")
    rt, at = tols(dt, float(z.abs().max()))
    assert_close(host(zd), z.detach().numpy(), rt, at, "z")
    dh = alloc(td, B, 2 * Z)
    L.mi_vae_reparam_kl_bwd(stream(), code, P(dev(dzs)), 2, mean.data_ptr(), logvar.data_ptr(), epsd.data_ptr(), klr.data_ptr(),
                            beta, tol, 1.0 / B, B, Z, dh.data_ptr())
    ref = torch.cat([mu.grad, lv.grad], 1).numpy()
    rt, at = tols(dt, float(np.abs(ref).max()))
    assert_close(host(dh), ref, rt, at, "dheads")
    # SURVEY 8b's single entry point: both halves in one call == the two calls above, bit for bit; each half alone as well
    mean2 = torch.empty(B, Z, device="cuda"); logvar2 = torch.empty(B, Z, device="cuda"); z2 = alloc(td, B, Z); klr2 = torch.empty(B, device="cuda"); dh2 = alloc(td, B, 2 * Z)
    L.mi_vae_reparam_kl_fwd_bwd(stream(), code, P(dev(heads)), ns, P(dev(bm)), P(dev(bl)), epsd.data_ptr(), 1, B, Z, mean2.data_ptr(), logvar2.data_ptr(), z2.data_ptr(),
                                klr2.data_ptr(), P(dev(dzs)), 2, beta, tol, 1.0 / B, dh2.data_ptr())
    for a, b_, what in ((mean2, mean, "mean"), (logvar2, logvar, "logvar"), (klr2, klr, "kl"), (z2, zd, "z"), (dh2, dh, "dheads")):
        assert np.array_equal(host(a), host(b_)), what
    dh3 = alloc(td, B, 2 * Z)
    L.mi_vae_reparam_kl_fwd_bwd(stream(), code, None, 0, None, None, epsd.data_ptr(), 1, B, Z, mean2.data_ptr(), logvar2.data_ptr(), None, klr2.data_ptr(),
                                P(dev(dzs)), 2, beta, tol, 1.0 / B, dh3.data_ptr())                    # backward half alone, on what the forward half left
    assert np.array_equal(host(dh3), host(dh))
    with pytest.raises(milib.MiError):
        L.mi_vae_reparam_kl_fwd_bwd(stream(), code, None, 0, None, None, epsd.data_ptr(), 1, B, Z, mean2.data_ptr(), logvar2.data_ptr(), None, klr2.data_ptr(),
                                    None, 0, beta, tol, 1.0 / B, None)
    # inference mode: z == mean, no eps needed
    L.mi_vae_reparam_kl_fwd(stream(), code, P(dev(heads)), ns, P(dev(bm)), P(dev(bl)), None, 0, B, Z,
                            mean.data_ptr(), logvar.data_ptr(), zd.data_ptr(), klr.data_ptr())
    assert_close(host(zd), rounded(host(mean).astype(np.float32), td).numpy(), 0, 0, "z==mean")
    # kl_tolerance: rows whose KL is below the floor get no KL gradient
    floor = float(np.median(host(klr)))
    L.mi_vae_reparam_kl_bwd(stream(), code, P(dev(np.zeros_like(dzs))), 2, mean.data_ptr(), logvar.data_ptr(), epsd.data_ptr(), klr.data_ptr(),
                            beta, floor, 1.0 / B, B, Z, dh.data_ptr())
    g = host(dh)
    below = host(klr) < floor
    assert (g[below] == 0).all() and (np.abs(g[~below]).sum(1) > 0).all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Co,kind", [(3, 0), (3, 2), (1, 0), (1, 1)])
def test_deconv_fwd_with_fused_reconstruction_loss(dt, Co, kind):
    """mi_deconv2d_nhwc_fwd_bce (deconv4 + loss in one kernel) against the float64 statement of the two reference ops
    (tf.layers.conv2d_transpose, vae/models.py:264; bce / bce_v2 / mse + reduce_sum, :11-22,123-128): logits, dlogits, the loss
    and the per-channel dlogits sums (BiasAddGrad), with the target frames gathered through frame_idx."""
    import ctypes
    L = milib.get()
    code, td = DT[dt]
    IH, IW, Ci, k, B = 39, 79, 32, 4, 3
    OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
    rng = np.random.RandomState(Co * 10 + kind)
    x = rng.randn(B, IH, IW, Ci).astype(np.float32)
    w = (rng.randn(k, k, Co, Ci) / np.sqrt(4 * Ci)).astype(np.float32)
    b = (0.1 * rng.randn(Co)).astype(np.float32)
    frames = rng.rand(5, OH * OW * Co).astype(np.float32)
    idx = np.array([4, 0, 3], np.int32)
    inv_b = 1.0 / 8.0
    xr, wr = rounded(x, td), rounded(w, td)
    y = F.conv_transpose2d(_nchw(xr), wr.permute(3, 2, 0, 1), torch.from_numpy(b).double(), stride=2)
    logits_ref = rounded(_nhwc(y).float().numpy(), td).requires_grad_(True)            # the loss reads the STORED logits
    t = torch.from_numpy(frames[idx]).double().reshape(B, OH, OW, Co)
    if kind == 0:
        per = torch.clamp(logits_ref, min=0) - logits_ref * t + torch.log1p(torch.exp(-logits_ref.abs()))
    elif kind == 1:
        sg = torch.sigmoid(logits_ref); per = -(t * torch.log(1e-10 + sg) + (1 - t) * torch.log(1e-10 + 1 - sg))
    else:
        per = (t - torch.sigmoid(logits_ref)) ** 2
    (per.sum() * inv_b).backward()
    cap = 4096
    lp, bp = torch.zeros(cap, device="cuda"), torch.zeros(cap, 4, device="cuda")
    logits = alloc(td, B, OH, OW, Co)
    dl = alloc(td, B, OH, OW, Co)
    n = ctypes.c_int(0)
    L.mi_deconv2d_nhwc_fwd_bce(stream(), code, P(dev(x, td)), B, IH, IW, Ci, P(dev(w, td)), P(dev(b)), k, k, Co, logits.data_ptr(),
                               P(dev(frames)), P(dev(idx, torch.int32)), OH * OW * Co, kind, inv_b, dl.data_ptr(), lp.data_ptr(), bp.data_ptr(), cap,
                               ctypes.addressof(n))
    torch.cuda.synchronize()
    assert 0 < n.value <= cap, "the 32 -> %d channel layer is eligible for the fused kernel" % Co
    rt, at = tols(dt, float(logits_ref.abs().max()))
    assert_close(host(logits), _nhwc(y).detach().numpy(), rt, at, "logits")
    assert abs(float(lp[:n.value].double().sum()) / float(per.sum()) - 1) < (1e-5 if dt != "bf16" else 3e-3)
    rt, at = tols(dt, float(logits_ref.grad.abs().max()))
    assert_close(host(dl), logits_ref.grad.numpy(), rt, at, "dlogits")
    bias_ref = logits_ref.grad.sum((0, 1, 2)).numpy()
    got = bp[:n.value, :Co].double().sum(0).cpu().numpy()
    assert_close(got, bias_ref, 2e-3 if dt == "bf16" else 1e-5, 2e-3 * float(np.abs(bias_ref).max()) + 1e-7, "fused bias gradient")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_recon_loss_fwd_bwd(dt, kind):
    L = milib.get()
    code, td = DT[dt]
    rng = np.random.RandomState(kind)
    B, NP = 5, 38400
    logits = (rng.randn(B, NP) * 3).astype(np.float32)
    frames = rng.rand(8, NP).astype(np.float32)
    idx = np.array([7, 2, 2, 0, 5], np.int32)
    x = rounded(logits, td).requires_grad_(True)
    y = torch.from_numpy(frames[idx]).double()
    if kind == 0:
        per = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    elif kind == 1:
        s = torch.sigmoid(x); per = -(y * torch.log(1e-10 + s) + (1 - y) * torch.log(1e-10 + 1 - s))
    else:
        per = (y - torch.sigmoid(x)) ** 2
    rows = per.sum(1)
    (rows.sum() / 16.0).backward()                       # inv_batch = 1/16 (global batch under data parallelism)
    nch = L.mi_recon_loss_chunks(NP)
    partial = torch.zeros(B, nch, device="cuda")
    dl = alloc(td, B, NP)
    L.mi_bce_logits_fwd_bwd(stream(), code, P(dev(logits, td)), P(dev(frames)), P(dev(idx, torch.int32)), NP, B, NP, kind,
                            1.0 / 16.0, dl.data_ptr(), partial.data_ptr())
    assert_close(host(partial).sum(1), rows.detach().numpy(), 2e-6, 1e-3, "row losses")
    rt, at = tols(dt, float(x.grad.abs().max()))
    assert_close(host(dl), x.grad.numpy(), rt, at, "dlogits")
    klr = dev(rng.rand(B).astype(np.float32))
    out2, met = torch.zeros(2, device="cuda"), torch.zeros(3, device="cuda")
    for _ in range(2):
        L.mi_vae_finalize_losses(stream(), partial.data_ptr(), nch, klr.data_ptr(), 0.5, B, 1.0 / B, out2.data_ptr(), met.data_ptr(), 1.0)
    o = host(out2)
    assert o[0] == pytest.approx(float(rows.mean()), rel=1e-5)
    assert o[1] == pytest.approx(float(np.maximum(host(klr), 0.5).mean()), rel=1e-6)
    assert np.allclose(host(met), [2 * o[0], 2 * o[1], 2.0], rtol=1e-6)


def test_adam_tf_flat_bit_exact_vs_c_restatement():
    import ctypes, os
    L = milib.get()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = ctypes.CDLL(os.path.join(root, "oracle", "libgae_ref.so"))
    ref.adam_tf_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t] + [ctypes.c_float] * 4
    rng = np.random.RandomState(0)
    n = 100003
    p, g = rng.randn(n).astype(np.float32), (rng.randn(n) * 10 ** rng.uniform(-6, 1, n)).astype(np.float32)
    m, v = (rng.randn(n) * 0.1).astype(np.float32), (rng.rand(n) * 0.01).astype(np.float32)
    npad = (n + 7) // 8 * 8
    buf = [torch.zeros(npad, device="cuda") for _ in range(4)]
    for t, a in zip(buf, (p, m, v, g)):
        t[:n] = torch.from_numpy(a).cuda()
    shadow = torch.zeros(npad, device="cuda", dtype=torch.bfloat16)
    alpha = np.float32(1e-4 * np.sqrt(1 - 0.999 ** 3) / (1 - 0.9 ** 3))
    L.mi_adam_tf_flat(stream(), buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr(), n, float(alpha), 0.9, 0.999, 1e-8, shadow.data_ptr(), 1)
    pc, mc, vc = p.copy(), m.copy(), v.copy()
    ref.adam_tf_f32(pc.ctypes.data, mc.ctypes.data, vc.ctypes.data, g.ctypes.data, n, alpha, np.float32(0.9), np.float32(0.999), np.float32(1e-8))
    torch.cuda.synchronize()
    assert np.array_equal(buf[0][:n].cpu().numpy(), pc) and np.array_equal(buf[1][:n].cpu().numpy(), mc) and np.array_equal(buf[2][:n].cpu().numpy(), vc)
    assert float(buf[3].abs().max()) == 0.0
    assert torch.equal(shadow[:n].cpu(), torch.from_numpy(pc).to(torch.bfloat16))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("MN", [(3 * 12800, 3), (3 * 3081, 32), (1000, 64), (77, 256), (9, 6144), (40, 500), (40, 300), (40, 2), (40, 1), (5, 128)])
def test_colsum(dt, MN):
    L = milib.get()
    code, td = DT[dt]
    M, N = MN
    x = np.random.RandomState(N).randn(M, N).astype(np.float32)
    out = torch.ones(N, device="cuda")
    L.mi_colsum(stream(), code, P(dev(x, td)), M, N, out.data_ptr())
    ref = rounded(x, td).sum(0).numpy() + 1.0
    assert_close(host(out), ref, 1e-5, 1e-4 * max(1.0, float(np.abs(ref).max())), "colsum")


def test_sigmoid_range_check_cast():
    L = milib.get()
    x = np.random.RandomState(0).randn(1000).astype(np.float32) * 5
    out = torch.empty(1000, device="cuda")
    L.mi_sigmoid(stream(), milib.MI_F32, P(dev(x)), out.data_ptr(), 1000)
    assert_close(host(out), 1 / (1 + np.exp(-x.astype(np.float64))), 1e-6, 1e-7, "sigmoid")
    flag = torch.zeros(1, device="cuda", dtype=torch.int32)
    ok = dev(np.random.rand(5000).astype(np.float32))
    L.mi_range_check(stream(), ok.data_ptr(), 5000, 0.0, 1.0, flag.data_ptr())
    assert int(flag.item()) == 0
    ok[4321] = 1.5
    L.mi_range_check(stream(), ok.data_ptr(), 5000, 0.0, 1.0, flag.data_ptr())
    assert int(flag.item()) == 1
    bf = torch.empty(1000, device="cuda", dtype=torch.bfloat16)
    L.mi_cast_f32_to_bf16(stream(), P(dev(x)), bf.data_ptr(), 1000)
    torch.cuda.synchronize()
    assert torch.equal(bf.cpu(), torch.from_numpy(x).to(torch.bfloat16))


def test_ppo_loss_and_head_vs_oracle_formulas():
    from oracle import ppo_oracle as po
    L = milib.get()
    rng = np.random.RandomState(3)
    M, A = 300, 2
    u, uo = rng.randn(M, A).astype(np.float32), rng.randn(M, A).astype(np.float32)
    uo = (u + 0.2 * uo).astype(np.float32)
    ls, lso = np.array([-0.3, 0.1], np.float32), np.array([-0.25, 0.05], np.float32)
    v, R, Ad = rng.randn(M).astype(np.float32), rng.randn(M).astype(np.float32), rng.randn(M).astype(np.float32)
    act = rng.uniform(-1, 1, (M, A)).astype(np.float32)
    low, high = np.array([-1, 0], np.float32), np.array([1, 1], np.float32)
    ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    lst = torch.tensor(ls, dtype=torch.float64, requires_grad=True)
    vt = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    lo, hi = torch.tensor(low).double(), torch.tensor(high).double()
    mean = lo + ((torch.tanh(ut) + 1) / 2) * (hi - lo)
    mean_o = lo + ((torch.tanh(torch.tensor(uo).double()) + 1) / 2) * (hi - lo)
    a = torch.tensor(act).double()
    logp = po.normal_log_prob(a, mean, lst).sum(-1, keepdim=True)
    logpo = po.normal_log_prob(a, mean_o, torch.tensor(lso).double()).sum(-1, keepdim=True)
    ratio = torch.exp(logp - logpo)
    adv = torch.tensor(Ad).double().unsqueeze(-1)
    pl = torch.minimum(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
    vl = ((vt - torch.tensor(R).double()) ** 2).mean() * 0.7
    el = (0.5 + po.HALF_LOG_2PI + torch.log(torch.exp(lst))).sum() * 0.02
    loss = -pl + vl - el
    loss.backward()
    du, dv = torch.empty(M, A, device="cuda"), torch.empty(M, device="cuda")
    part = torch.zeros(L.mi_ppo_loss_partial_floats(M), device="cuda")
    losses, dls = torch.zeros(5, device="cuda"), torch.zeros(A, device="cuda")
    L.mi_ppo_loss_fwd_bwd(stream(), P(dev(u)), P(dev(uo)), P(dev(ls)), P(dev(lso)), P(dev(v)), P(dev(act)),
                          P(dev(R)), P(dev(Ad)), P(dev(low)), P(dev(high)), M, A, 0.2, 0.7, 0.02, 1.0 / M, 1.0,
                          du.data_ptr(), dv.data_ptr(), part.data_ptr(), losses.data_ptr(), dls.data_ptr())
    got = host(losses)
    assert np.allclose(got[:4], [float(pl), float(vl), float(el), float(loss)], rtol=2e-5, atol=1e-6)
    assert got[4] == pytest.approx(float(ratio.mean()), rel=2e-5)
    assert_close(host(du), ut.grad.numpy(), 2e-4, 1e-6, "du")
    assert_close(host(dv), vt.grad.numpy(), 1e-5, 1e-7, "dv")
    assert_close(host(dls), lst.grad.numpy(), 2e-4, 1e-6, "dlogstd")
    noise = rng.randn(M, A).astype(np.float32)
    actd, meand = torch.empty(M, A, device="cuda"), torch.empty(M, A, device="cuda")
    L.mi_policy_head(stream(), P(dev(u)), P(dev(ls)), P(dev(noise)), P(dev(low)), P(dev(high)), M, A, 0, actd.data_ptr(), meand.data_ptr())
    ref = np.clip(mean.detach().numpy() + np.exp(ls.astype(np.float64)) * noise, low, high)
    assert_close(host(actd), ref, 1e-5, 1e-6, "sampled action")
    assert_close(host(meand), mean.detach().numpy(), 1e-5, 1e-6, "action mean")


def test_gae_scan_bit_exact_and_normalize():
    from oracle import ppo_oracle as po
    L = milib.get()
    rng = np.random.RandomState(9)
    R, T = 70, 128
    rew = rng.uniform(0, 1, (R, T))
    val = rng.randn(R, T + 1).astype(np.float32).astype(np.float64)
    done = np.zeros((R, T)); done[::3, -1] = 1.0
    adv = torch.empty(R, T, device="cuda", dtype=torch.float64)
    vd = dev(val, torch.float64)
    L.mi_gae_scan(stream(), P(dev(rew, torch.float64)), vd.data_ptr(), P(dev(done, torch.float64)), R, T, 0.99, 0.95, adv.data_ptr())
    got = host(adv)
    for r in range(R):
        ref = po.compute_gae(list(rew[r]), list(val[r, :T].astype(np.float32)), np.float32(val[r, T]), list(done[r].astype(bool)), 0.99, 0.95)
        assert np.array_equal(got[r], ref), r                         # bit-exact fp64
    ret = torch.empty(R, T, device="cuda", dtype=torch.float64)
    L.mi_adv_normalize(stream(), adv.data_ptr(), vd.data_ptr(), R, T, ret.data_ptr())
    a2, r2 = host(adv), host(ret)
    for r in range(0, R, 7):
        rr, aa = po.returns_and_normalized_advantages(got[r].copy(), val[r, :T])
        assert np.array_equal(r2[r], rr)
        assert np.allclose(a2[r], aa, rtol=1e-12, atol=1e-12)


def _relu_bit_words(act):
    """Reference packing of the ReLU bit words (mi355_carla.h): act [.., C] -> uint32 [.., C // 16]; bit i = channel 16 j + 2 i > 0,
    bit 16 + i = channel 16 j + 2 i + 1 > 0."""
    a = (np.asarray(act) > 0).astype(np.uint32)
    a = a.reshape(a.shape[:-1] + (a.shape[-1] // 16, 8, 2))
    sh = np.arange(8, dtype=np.uint32)
    return ((a[..., 0] << sh).sum(-1) | ((a[..., 1] << sh).sum(-1) << np.uint32(16))).astype(np.uint32)


def test_relu_bit_words_producers_and_consumers():
    """bf16 engine: conv1 forward (narrow kernel) and deconv3 forward (register-weight kernel) also write the ReLU bit words of their output;
    conv2's input gradient (register-weight kernel) and deconv4's (narrow kernel) read those 8 bytes per pixel instead of the 64-byte
    activation row.  Words equal the reference packing of (stored output > 0); gradients equal the full-mask results bit for bit."""
    L = milib.get()
    prev = L.mi_set_tuning(13, 2)
    try:
        code, td = DT["bf16"]
        rng = np.random.RandomState(3)
        B = 3
        # --- producer 1: conv1 forward 80x160x3 -> 39x79x32
        x = rng.rand(B, 80, 160, 3).astype(np.float32)
        w = (rng.randn(4, 4, 3, 32) / np.sqrt(48)).astype(np.float32)
        b = (0.1 * rng.randn(32)).astype(np.float32)
        wt = torch.from_numpy(w).permute(3, 0, 1, 2).reshape(32, -1).contiguous()
        out = alloc(td, B, 39, 79, 32)
        bits = torch.zeros(B * 39 * 79 * 2, device="cuda", dtype=torch.int32)
        wrote = np.zeros(1, np.int32)
        L.mi_conv2d_nhwc_fwd_bits(stream(), code, P(dev(x)), None, 1, B, 80, 160, 3, P(dev(wt, td)), 1, P(dev(b)), 4, 4, 32, 1, out.data_ptr(), bits.data_ptr(), wrote.ctypes.data)
        assert wrote[0] == 1
        act1 = host(out)
        assert np.array_equal(bits.cpu().numpy().view(np.uint32).reshape(B, 39, 79, 2), _relu_bit_words(act1))
        assert 0.2 < (act1 > 0).mean() < 0.8
        # --- consumer 1: conv2 input gradient with those words == with the full mask
        dy = rng.randn(B, 18, 38, 64).astype(np.float32)
        w2 = (rng.randn(4, 4, 32, 64) / np.sqrt(512)).astype(np.float32)
        dx_full = alloc(td, B, 39, 79, 32)
        dx_bits = torch.empty_like(dx_full)
        dyd, w2d = dev(dy, td), dev(w2, td)
        L.mi_conv2d_nhwc_dgrad(stream(), code, dyd.data_ptr(), B, 18, 38, 64, w2d.data_ptr(), 4, 4, 32, 39, 79, out.data_ptr(), dx_full.data_ptr())
        L.mi_conv2d_nhwc_dgrad_bits(stream(), code, dyd.data_ptr(), B, 18, 38, 64, w2d.data_ptr(), 4, 4, 32, 39, 79, out.data_ptr(), bits.data_ptr(), dx_bits.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(dx_full, dx_bits) and float(dx_full.float().abs().max()) > 0
        # --- producer 2: deconv3 forward 18x38x64 -> 39x79x32 (k5)
        x3 = np.maximum(rng.randn(B, 18, 38, 64), 0).astype(np.float32)
        w3 = (rng.randn(5, 5, 32, 64) / np.sqrt(25 * 64 / 4)).astype(np.float32)
        o3 = alloc(td, B, 39, 79, 32)
        o3_ref = torch.empty_like(o3)
        bits3 = torch.zeros(B * 39 * 79 * 2, device="cuda", dtype=torch.int32)
        x3d, w3d, b3d = dev(x3, td), dev(w3, td), dev(b)
        wrote[0] = 0
        L.mi_deconv2d_nhwc_fwd_bits(stream(), code, x3d.data_ptr(), B, 18, 38, 64, w3d.data_ptr(), b3d.data_ptr(), 5, 5, 32, 1, o3.data_ptr(), bits3.data_ptr(), wrote.ctypes.data)
        L.mi_deconv2d_nhwc_fwd(stream(), code, x3d.data_ptr(), B, 18, 38, 64, w3d.data_ptr(), b3d.data_ptr(), 5, 5, 32, 1, o3_ref.data_ptr())
        torch.cuda.synchronize()
        assert wrote[0] == 1 and torch.equal(o3, o3_ref)
        assert np.array_equal(bits3.cpu().numpy().view(np.uint32).reshape(B, 39, 79, 2), _relu_bit_words(host(o3)))
        # --- consumer 2: deconv4 input gradient (80x160x3 logits gradient -> 39x79x32)
        dl = rng.randn(B, 80, 160, 3).astype(np.float32)
        w4 = (rng.randn(4, 4, 3, 32) / np.sqrt(48)).astype(np.float32)                      # [kh,kw,co=3,ci=32]
        w4t = torch.from_numpy(w4).permute(3, 0, 1, 2).reshape(32, -1).contiguous()          # [Cin][kh*kw*co]
        g_full = alloc(td, B, 39, 79, 32)
        g_bits = torch.empty_like(g_full)
        dld, w4d = dev(dl, td), dev(w4t, td)
        L.mi_deconv2d_nhwc_dgrad(stream(), code, dld.data_ptr(), B, 80, 160, 3, w4d.data_ptr(), 1, 4, 4, 32, o3.data_ptr(), g_full.data_ptr())
        L.mi_deconv2d_nhwc_dgrad_bits(stream(), code, dld.data_ptr(), B, 80, 160, 3, w4d.data_ptr(), 1, 4, 4, 32, o3.data_ptr(), bits3.data_ptr(), g_bits.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(g_full, g_bits) and float(g_full.float().abs().max()) > 0
    finally:
        L.mi_set_tuning(13, prev)


def test_split_storage_casts_and_adam_shadow():
    """Split storage (precision "bf16x3"): the device's split_from_f32 / split_to_f32 against their host restatement (hip_helpers.split_encode /
    split_decode) bit for bit -- normal values over 60 binades, zeros, tiny values -- the representation error bound hi + lo = x (1 + 2^-17), and
    the split shadow weights written by the fused Adam kernel."""
    from hip_helpers import split_decode, split_encode
    L = milib.get()
    rng = np.random.RandomState(0)
    n = 200003
    x = (rng.randn(n) * 10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
    x[:7] = [0.0, -0.0, 1.0, -1.0, 1e-30, 3.0e38, 255.0 / 256.0]
    xs = torch.empty(n, device="cuda", dtype=torch.int32)
    L.mi_cast_f32_to_split(stream(), P(dev(x)), xs.data_ptr(), n)
    torch.cuda.synchronize()
    assert np.array_equal(xs.cpu().numpy().view(np.uint32), split_encode(x))
    back = torch.empty(n, device="cuda")
    L.mi_cast_split_to_f32(stream(), xs.data_ptr(), back.data_ptr(), n)
    got = host(back)
    assert np.array_equal(got, split_decode(split_encode(x)).astype(np.float32).astype(np.float64))
    big = np.abs(x) > 1e-30                                   # (the lo half of a value near the bottom of the fp32 range is subnormal in bf16)
    assert np.abs(got[big] / x[big].astype(np.float64) - 1).max() <= 2.0 ** -16
    # Adam: fp32 masters updated exactly as before, shadow = split(masters)
    m4 = (n + 7) // 8 * 8
    p, g = rng.randn(m4).astype(np.float32), rng.randn(m4).astype(np.float32)
    buf = [dev(p), torch.zeros(m4, device="cuda"), torch.zeros(m4, device="cuda"), dev(g)]
    ref = [dev(p), torch.zeros(m4, device="cuda"), torch.zeros(m4, device="cuda"), dev(g)]
    shadow = torch.zeros(m4, device="cuda", dtype=torch.int32)
    L.mi_adam_tf_flat_shadow(stream(), buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr(), n, 1e-3, None, 0.9, 0.999, 1e-8,
                             shadow.data_ptr(), milib.MI_BF16X3, 1)
    L.mi_adam_tf_flat(stream(), ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr(), ref[3].data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, None, 1)
    torch.cuda.synchronize()
    assert torch.equal(buf[0], ref[0]) and torch.equal(buf[1], ref[1]) and torch.equal(buf[2], ref[2])
    assert np.array_equal(shadow[:n].cpu().numpy().view(np.uint32), split_encode(buf[0][:n].cpu().numpy()))


def test_split_storage_products_carry_sixteen_bits():
    """What the split mode buys: a long dot product of UNROUNDED fp32 operands.  bf16 storage is ~2^-9 per operand, split storage ~2^-17: the
    dense layer of the model's longest reduction (K = 6144) on fp32 inputs through each storage type against float64."""
    L = milib.get()
    rng = np.random.RandomState(5)
    M, K, N = 64, 6144, 128
    a, w = rng.randn(M, K).astype(np.float32), (rng.randn(N, K) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    err = {}
    for dt in DTS:
        code, td = DT[dt]
        out = torch.empty(M, N, device="cuda")
        L.mi_gemm_bias_act(stream(), code, P(dev(a, td)), M, K, P(dev(w, td)), 1, N, None, 0, None, out.data_ptr(), 1, 1)
        err[dt] = float(np.abs(host(out) - ref).max() / np.abs(ref).max())
    print("dense K = 6144, fp32 operands: max error / max |ref| by storage type:", err)
    # measured on MI355X: fp32 2.5e-6 (K = 6144 fmaf chain), split 3.2e-6, bf16 1.9e-3
    assert err["f32"] < 6e-6 and err["x3"] < 2e-5 and err["bf16"] > 50 * err["x3"], err


@pytest.mark.parametrize("kind,u8", [(0, True), (0, False), (1, True), (2, False)])
@pytest.mark.parametrize("B,IH,IW", [(3, 39, 79), (2, 7, 15), (1, 10, 33), (512, 39, 79)])
def test_decoder_tail_fused_equals_the_three_ops(kind, u8, B, IH, IW):
    """mi_deconv2d_tail_fused (deconv4 forward + loss + its input gradient + its filter gradient in one launch, dectail_tile.hpp) against the three
    separately validated ops it replaces (mi_deconv2d_nhwc_fwd_bce_u8, mi_deconv2d_nhwc_dgrad with the activation as ReluGrad mask,
    mi_deconv2d_nhwc_wgrad) on the model's geometry and on two ragged ones (tiles of 8 x 16 pixels that overhang the image, a single tile row):
    the input gradient bit for bit (same summation order), loss and bias-gradient sums to fp32 summation order, the filter gradient to 1e-5 of its
    max; the gradient buffer is ACCUMULATED into."""
    import ctypes
    if B == 512 and (kind, u8) != (0, True):
        pytest.skip("the benchmarked batch (768-block grid, XCD-contiguous tile ranges, several tiles per block) runs once, in the production configuration")
    L = milib.get()
    code, td = DT["bf16"]
    Ci, Co, k = 32, 3, 4
    OH, OW = 2 * IH + 2, 2 * IW + 2
    rng = np.random.RandomState(B * 100 + IH + kind)
    x = np.maximum(rng.randn(B, IH, IW, Ci), 0).astype(np.float32)                       # post-ReLU activations (about half of them zero)
    w = (rng.randn(k, k, Co, Ci) / np.sqrt(4 * Ci)).astype(np.float32)
    b = (0.1 * rng.randn(Co)).astype(np.float32)
    n_frames = B + 2
    frames_u8 = rng.randint(0, 256, (n_frames, OH * OW * Co)).astype(np.uint8)
    idx = rng.permutation(n_frames)[:B].astype(np.int32)
    labels = dev(frames_u8, torch.uint8) if u8 else dev(frames_u8.astype(np.float32) / 255.0)
    inv_b = 1.0 / 16.0
    xd, wd, bd, idxd = dev(x, td), dev(w, td), dev(b), dev(idx, torch.int32)
    wt = torch.zeros(k * k * Co * Ci, device="cuda", dtype=td)
    offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Co], np.int32), np.array([Ci], np.int32)
    L.mi_transpose_weights(stream(), code, P(dev(w)), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
    cap = 16384 if B < 64 else 1 << 20
    # --- the three ops ---
    lp, bp = torch.zeros(cap, device="cuda"), torch.zeros(cap, 4, device="cuda")
    dl = alloc(td, B, OH, OW, Co, fill=0.0)
    n = ctypes.c_int(0)
    L.mi_deconv2d_nhwc_fwd_bce_u8(stream(), code, xd.data_ptr(), B, IH, IW, Ci, wd.data_ptr(), bd.data_ptr(), k, k, Co, None, labels.data_ptr(), int(u8), idxd.data_ptr(),
                                  OH * OW * Co, kind, inv_b, dl.data_ptr(), lp.data_ptr(), bp.data_ptr(), cap, ctypes.addressof(n))
    torch.cuda.synchronize()
    assert n.value > 0
    loss_ref = float(lp[:n.value].double().sum()); bias_ref = bp[:n.value, :Co].double().sum(0).cpu().numpy()
    dx_ref = alloc(td, B, IH, IW, Ci, fill=3.0)
    L.mi_deconv2d_nhwc_dgrad(stream(), code, dl.data_ptr(), B, OH, OW, Co, wt.data_ptr(), 1, k, k, Ci, xd.data_ptr(), dx_ref.data_ptr())
    dw_ref = torch.zeros(k, k, Co, Ci, device="cuda")
    L.mi_deconv2d_nhwc_wgrad(stream(), code, dl.data_ptr(), B, OH, OW, Co, xd.data_ptr(), k, k, Ci, dw_ref.data_ptr())
    # --- one launch ---
    nb = L.mi_deconv2d_tail_blocks()
    scratch = torch.empty(nb * 6144, device="cuda", dtype=torch.uint8)
    lp2, bp2 = torch.zeros(cap, device="cuda"), torch.zeros(cap, 4, device="cuda")
    dx = alloc(td, B, IH, IW, Ci, fill=5.0)
    dw = torch.full((k, k, Co, Ci), 0.25, device="cuda")
    n2 = ctypes.c_int(0)
    L.mi_deconv2d_tail_fused(stream(), code, xd.data_ptr(), B, IH, IW, Ci, wd.data_ptr(), wt.data_ptr(), bd.data_ptr(), k, k, Co, labels.data_ptr(), int(u8), idxd.data_ptr(),
                             OH * OW * Co, kind, inv_b, dx.data_ptr(), dw.data_ptr(), lp2.data_ptr(), bp2.data_ptr(), cap, ctypes.addressof(n2), scratch.data_ptr(), scratch.numel(), int(kind != 1))
    if kind == 1:                                            # deferred form: the partial sums wait in scratch until the caller reduces them
        torch.cuda.synchronize()
        assert float((dw - 0.25).abs().max()) == 0.0
        L.mi_deconv2d_tail_reduce(stream(), scratch.data_ptr(), n2.value, dw.data_ptr())
    torch.cuda.synchronize()
    assert 0 < n2.value <= nb
    assert abs(float(lp2[:n2.value].double().sum()) / loss_ref - 1) < 2e-6
    got_b = bp2[:n2.value, :Co].double().sum(0).cpu().numpy()
    assert_close(got_b, bias_ref, 1e-5, 1e-5 * float(np.abs(bias_ref).max()) + 1e-7, "bias gradient sums")
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)), float((dx.float() - dx_ref.float()).abs().max())
    s = float(dw_ref.abs().max())
    assert_close(host(dw) - 0.25, host(dw_ref), 1e-5, (2e-5 if B < 64 else 2e-4) * s, "filter gradient")      # (1.6 M positions per element at batch 512: fp32 summation order)


@pytest.mark.parametrize("u8", [True, False])
@pytest.mark.parametrize("B,FH,FW", [(3, 80, 160), (2, 38, 70), (1, 20, 132), (512, 80, 160)])
def test_encoder_head_backward_fused_equals_the_two_ops(u8, B, FH, FW):
    """mi_conv2d_head_bwd_fused (conv2's input gradient + conv1's filter and bias gradient in one launch, enchead_tile.hpp; the gradient of conv1's output never
    exists) against the two separately validated ops it replaces -- mi_conv2d_nhwc_dgrad_bits, then mi_conv2d_nhwc_wgrad_ws on its output -- on the model's geometry
    and two ragged ones (8 x 16 tiles overhanging the image, odd slot counts), camera bytes and fp32 frames, gathered through a frame index, ACCUMULATING into the
    gradient buffers; and against the float64 statement of the same two contractions (bf16 operands, the intermediate rounded to bf16 like the stored tensor)."""
    import ctypes
    if os.environ.get("MI355_ENC12") == "0":
        pytest.skip("MI355_ENC12=0 switches the fused op off (A/B runs): nothing to compare")
    if B == 512 and not u8:
        pytest.skip("the benchmarked batch runs once, on camera bytes (the production configuration)")
    L = milib.get()
    code, td = DT["bf16"]
    IH, IW = (FH - 4) // 2 + 1, (FW - 4) // 2 + 1
    OH, OW = (IH - 4) // 2 + 1, (IW - 4) // 2 + 1
    rng = np.random.RandomState(B * 10 + FH)
    n_frames = B + 2
    frames_u8 = rng.randint(0, 256, (n_frames, FH, FW, 3)).astype(np.uint8)
    idx = rng.permutation(n_frames)[:B].astype(np.int32)
    fr = dev(frames_u8, torch.uint8) if u8 else dev(frames_u8.astype(np.float32) / 255.0)
    idxd = dev(idx, torch.int32)
    w1 = (rng.randn(4, 4, 3, 32) / np.sqrt(48)).astype(np.float32)
    b1 = (0.1 * rng.randn(32)).astype(np.float32)
    w1t = torch.from_numpy(w1).permute(3, 0, 1, 2).reshape(32, -1).contiguous()
    w2 = (rng.randn(4, 4, 32, 64) / np.sqrt(512)).astype(np.float32)
    dy = rng.randn(B, OH, OW, 64).astype(np.float32)
    act1 = alloc(td, B, IH, IW, 32)
    bits = torch.zeros(B * IH * IW * 2, device="cuda", dtype=torch.int32)
    wrote = np.zeros(1, np.int32)
    fmt = 2 if u8 else 1
    w1d, b1d = dev(w1t, td), dev(b1)                                        # (kept alive: temporaries inside the argument list may share one freed block)
    L.mi_conv2d_nhwc_fwd_bits(stream(), code, fr.data_ptr(), idxd.data_ptr(), fmt, B, FH, FW, 3, w1d.data_ptr(), 1, b1d.data_ptr(), 4, 4, 32, 1, act1.data_ptr(), bits.data_ptr(), wrote.ctypes.data)
    torch.cuda.synchronize()
    if wrote[0] != 1:
        pytest.skip("conv1 forward did not take the bit-word kernel on this geometry")
    dyd, w2d = dev(dy, td), dev(w2, td)
    # --- the two ops ---
    g1 = alloc(td, B, IH, IW, 32, fill=7.0)
    L.mi_conv2d_nhwc_dgrad_bits(stream(), code, dyd.data_ptr(), B, OH, OW, 64, w2d.data_ptr(), 4, 4, 32, IH, IW, act1.data_ptr(), bits.data_ptr(), g1.data_ptr())
    dw_ref, db_ref = torch.zeros(4, 4, 3, 32, device="cuda"), torch.zeros(32, device="cuda")
    ws = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    L.mi_conv2d_nhwc_wgrad_ws(stream(), code, fr.data_ptr(), idxd.data_ptr(), fmt, B, FH, FW, 3, g1.data_ptr(), 4, 4, 32, dw_ref.data_ptr(), ws.data_ptr(), ws.numel(), db_ref.data_ptr())
    # --- one launch ---
    nb = L.mi_conv2d_head_bwd_blocks()
    scratch = torch.empty(nb * 8320, device="cuda", dtype=torch.uint8)
    dw, db = torch.full((4, 4, 3, 32), 0.5, device="cuda"), torch.full((32,), -0.25, device="cuda")
    n = ctypes.c_int(0)
    L.mi_conv2d_head_bwd_fused(stream(), code, fr.data_ptr(), fmt, idxd.data_ptr(), B, FH, FW, dyd.data_ptr(), w2d.data_ptr(), bits.data_ptr(), dw.data_ptr(), db.data_ptr(),
                               scratch.data_ptr(), scratch.numel(), ctypes.addressof(n))
    torch.cuda.synchronize()
    assert 0 < n.value <= nb
    sw, sb = float(dw_ref.abs().max()), float(db_ref.abs().max())
    assert sw > 0 and sb > 0
    # same bf16 operands, same rounding point of the intermediate; what differs is the fp32 summation order inside the K = 256 products (a last-bit flip of an
    # intermediate value now and then) and across blocks
    assert_close(host(dw) - 0.5, host(dw_ref), 1e-3, 2e-4 * sw, "conv1 filter gradient, fused vs two ops")
    assert_close(host(db) + 0.25, host(db_ref), 1e-3, 2e-4 * sb, "conv1 bias gradient, fused vs two ops")
    if B == 512:
        return                                              # (the float64 statement below is a CPU conv over the whole batch: the small geometries carry it)
    # --- float64 statement ---
    t64 = lambda a: torch.from_numpy(np.asarray(a, np.float64))      # noqa: E731
    dyb, w2b = host(dyd).astype(np.float64), host(w2d).astype(np.float64)
    g = torch.nn.functional.conv_transpose2d(t64(dyb).permute(0, 3, 1, 2), t64(w2b).permute(3, 2, 0, 1), stride=2)      # [B, 32, IH', IW']
    g = torch.nn.functional.pad(g, (0, IW - g.shape[3], 0, IH - g.shape[2]))
    g = g.permute(0, 2, 3, 1) * t64(host(act1) > 0)
    g = t64(host(dev(g.numpy().astype(np.float32), td)))                                                                   # rounded where the unfused path stores it
    xs = t64(host(dev(frames_u8[idx].astype(np.float32) / 255.0, td)))                                                      # the bf16 value of k / 255 the kernels form
    patches = xs.unfold(1, 4, 2).unfold(2, 4, 2)                                                                            # [B, IH, IW, 3, kh, kw]
    dw64 = torch.einsum("byxckl,byxn->klcn", patches, g).numpy()
    db64 = g.sum((0, 1, 2)).numpy()
    assert_close(host(dw) - 0.5, dw64, 2e-3, 5e-4 * float(np.abs(dw64).max()), "conv1 filter gradient vs float64")
    assert_close(host(db) + 0.25, db64, 2e-3, 5e-4 * float(np.abs(db64).max()), "conv1 bias gradient vs float64")


@pytest.mark.parametrize("u8", [True, False])
@pytest.mark.parametrize("B", [1, 3, 37, 512])
def test_encoder_head_forward_fused_equals_the_two_ops(B, u8):
    """Round 5 (enc12_tile.hpp): mi_conv2d_enc12_fwd -- conv1 computed band by band into LDS and consumed there by conv2, one launch --
    against the two separately validated layer ops it replaces (mi_conv2d_nhwc_fwd_bits on camera bytes or fp32 frames, mi_conv2d_nhwc_fwd on its output), frames gathered
    through a frame index: conv1's activation and its ReLU bit words BIT FOR BIT (every pixel, incl. the row / column conv2 never reads and the rows two bands compute), conv2's output
    against the float64 convolution of that stored bf16 activation and against the unfused kernel (same bf16 products, another fp32 summation order: a last-place flip of
    the bf16 result now and then); garbage in every output buffer beforehand.  Batch 1, a ragged number of bands per resident block, and the benchmarked 512."""
    import ctypes
    if B == 512 and not u8:
        pytest.skip("the benchmarked batch runs once, on camera bytes (the production configuration)")
    L = milib.get()
    code, td = DT["bf16"]
    rng = np.random.RandomState(100 + B)
    n_frames = B + 3
    frames_u8 = rng.randint(0, 256, (n_frames, 80, 160, 3)).astype(np.uint8)
    idx = rng.permutation(n_frames)[:B].astype(np.int32)
    fr = dev(frames_u8, torch.uint8) if u8 else dev(frames_u8.astype(np.float32) / np.float32(255.0))
    idxd = dev(idx, torch.int32)
    fmt = 2 if u8 else 1
    w1 = (rng.randn(4, 4, 3, 32) / np.sqrt(48)).astype(np.float32)
    b1 = (0.1 * rng.randn(32)).astype(np.float32)
    w2 = (rng.randn(4, 4, 32, 64) / np.sqrt(512)).astype(np.float32)
    b2 = (0.1 * rng.randn(64)).astype(np.float32)
    w1t = dev(torch.from_numpy(w1).permute(3, 0, 1, 2).reshape(32, -1).contiguous(), td)          # K-contiguous copies [N][K]
    w2t = dev(torch.from_numpy(w2).permute(3, 0, 1, 2).reshape(64, -1).contiguous(), td)
    b1d, b2d = dev(b1), dev(b2)
    # --- the two ops ---
    act1 = alloc(td, B, 39, 79, 32, fill=3.0)
    bits = torch.full((B * 39 * 79 * 2,), 0x55, device="cuda", dtype=torch.int32)
    wrote = np.zeros(1, np.int32)
    L.mi_conv2d_nhwc_fwd_bits(stream(), code, fr.data_ptr(), idxd.data_ptr(), fmt, B, 80, 160, 3, w1t.data_ptr(), 1, b1d.data_ptr(), 4, 4, 32, 1, act1.data_ptr(), bits.data_ptr(), wrote.ctypes.data)
    torch.cuda.synchronize()
    assert wrote[0] == 1
    act2 = alloc(td, B, 18, 38, 64, fill=3.0)
    L.mi_conv2d_nhwc_fwd(stream(), code, act1.data_ptr(), None, 0, B, 39, 79, 32, w2t.data_ptr(), 1, b2d.data_ptr(), 4, 4, 64, 1, act2.data_ptr())
    # --- one launch ---
    f_act1 = alloc(td, B, 39, 79, 32, fill=-7.0)
    f_bits = torch.full((B * 39 * 79 * 2,), 0x33, device="cuda", dtype=torch.int32)
    f_act2 = alloc(td, B, 18, 38, 64, fill=-7.0)
    launched = ctypes.c_int(0)
    L.mi_conv2d_enc12_fwd(stream(), code, fr.data_ptr(), fmt, idxd.data_ptr(), B, 80, 160, w1t.data_ptr(), b1d.data_ptr(), w2t.data_ptr(), b2d.data_ptr(),
                          f_act1.data_ptr(), f_bits.data_ptr(), f_act2.data_ptr(), ctypes.addressof(launched))
    torch.cuda.synchronize()
    assert launched.value == 1
    assert torch.equal(f_act1.view(torch.int16), act1.view(torch.int16)), "conv1 activation: fused vs layer op, bit for bit"
    assert torch.equal(f_bits, bits), "ReLU bit words: fused vs layer op"
    a2, r2 = host(f_act2), host(act2)
    scale = float(np.abs(r2).max())
    assert scale > 0.1
    assert_close(a2, r2, 2.0 ** -7, 2.0 ** -8 * scale, "conv2 output: fused vs layer op (bf16 results of two fp32 summation orders)")
    assert (a2 >= 0).all() and (a2 == 0).mean() > 0.05, "ReLU"
    # without the bit words (inference): same tensors
    g_act1, g_act2 = alloc(td, B, 39, 79, 32, fill=-7.0), alloc(td, B, 18, 38, 64, fill=-7.0)
    L.mi_conv2d_enc12_fwd(stream(), code, fr.data_ptr(), fmt, idxd.data_ptr(), B, 80, 160, w1t.data_ptr(), b1d.data_ptr(), w2t.data_ptr(), b2d.data_ptr(),
                          g_act1.data_ptr(), None, g_act2.data_ptr(), ctypes.addressof(launched))
    assert torch.equal(g_act1.view(torch.int16), act1.view(torch.int16)) and torch.equal(g_act2.view(torch.int16), f_act2.view(torch.int16))
    # not eligible: another geometry, another storage type -> nothing launched
    L.mi_conv2d_enc12_fwd(stream(), code, fr.data_ptr(), fmt, idxd.data_ptr(), B, 78, 160, w1t.data_ptr(), b1d.data_ptr(), w2t.data_ptr(), b2d.data_ptr(),
                          g_act1.data_ptr(), None, g_act2.data_ptr(), ctypes.addressof(launched))
    assert launched.value == 0
    L.mi_conv2d_enc12_fwd(stream(), milib.MI_F32, fr.data_ptr(), fmt, idxd.data_ptr(), B, 80, 160, w1t.data_ptr(), b1d.data_ptr(), w2t.data_ptr(), b2d.data_ptr(),
                          g_act1.data_ptr(), None, g_act2.data_ptr(), ctypes.addressof(launched))
    assert launched.value == 0
    if B > 37:
        return                                              # (the float64 statement is a CPU conv over the whole batch: the small batches carry it)
    x = torch.from_numpy(host(act1).astype(np.float64)).permute(0, 3, 1, 2)
    w2b = torch.from_numpy(host(w2t).astype(np.float64)).reshape(64, 4, 4, 32).permute(0, 3, 1, 2)      # [n][kh][kw][c] -> [n][c][kh][kw]
    ref = torch.nn.functional.conv2d(x, w2b, torch.from_numpy(b2.astype(np.float64)), stride=2).clamp_min(0).permute(0, 2, 3, 1).numpy()
    assert_close(a2, ref, 2.0 ** -8, 2.0 ** -9 * scale, "conv2 output vs the float64 convolution of the stored activation")


@pytest.mark.parametrize("geom", [(39, 79, 32, 64), (18, 38, 64, 128)])
def test_bf16_partial_sum_slabs_error_bound_at_batch_512(geom):
    """the bf16 engine stores the position-split partial sums of its raw-staged filter gradients rounded to bf16 (mi_set_tuning key 18).  What that costs,
    measured at the benchmarked batch on the op itself: every one of the ~250 partial sums carries a relative rounding error of up to 2^-9 (RMS ~ 2^-9 / sqrt 3), and for
    partial sums of independent sign the error of their total is that fraction of sqrt(sum s_i^2) ~ the total's own RMS size: ~1e-3 of the RMS element of dW
    (NOT 1e-4: the statement corrected in csrc/vae_engine.hip), bounded here by 2e-3 RMS / 1e-2 of the tensor max against the fp32-slab result of the same kernel."""
    L = milib.get()
    code, td = DT["bf16"]
    IH, IW, Ci, Co = geom
    k, B = 4, 512
    OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
    gen = torch.Generator(device="cuda").manual_seed(IH)
    xd = torch.randn(B, IH, IW, Ci, device="cuda", generator=gen).to(td)
    dyd = torch.randn(B, OH, OW, Co, device="cuda", generator=gen).to(td)
    ws = torch.empty(256 << 20, device="cuda", dtype=torch.uint8)
    res = {}
    for mode in (0, 1):
        prev = L.mi_set_tuning(18, mode)
        try:
            dw, db = torch.zeros(k, k, Ci, Co, device="cuda"), torch.zeros(Co, device="cuda")
            L.mi_conv2d_nhwc_wgrad_ws(stream(), code, xd.data_ptr(), None, 0, B, IH, IW, Ci, dyd.data_ptr(), k, k, Co, dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())
            res[mode] = host(dw)
        finally:
            L.mi_set_tuning(18, prev)
    err = res[1] - res[0]
    rms, scale = float(np.sqrt(np.mean(err ** 2))), float(np.sqrt(np.mean(res[0] ** 2)))
    assert scale > 0 and not np.array_equal(res[0], res[1])               # (the knob did something)
    assert rms <= 2e-3 * scale, (rms, scale)
    assert float(np.abs(err).max()) <= 1e-2 * float(np.abs(res[0]).max())


@pytest.mark.parametrize("form,geom", [("conv", (39, 79, 32, 64)), ("conv", (18, 38, 64, 128)), ("conv", (8, 18, 128, 256)), ("deconv", (8, 18, 128, 64)), ("deconv", (3, 8, 256, 128))])
def test_split_storage_filter_gradient_on_the_doubled_channel_bf16_kernel(form, geom):
    """DESIGN finding 32: a split tensor (MI_BF16X3) read as bf16 has twice the channels (2c = lo half, 2c + 1 = hi half); the bf16 raw-staged filter-gradient kernel run on the
    doubled channel counts + fold_split_kernel (sum of every 2 x 2 block = the four partial products) must give the fp32-limit result of the split kernels: filter AND bias gradient
    against float64 on the split-rounded operands, accumulated into a non-zero buffer.  (mi_set_tuning key 21 = 2: every eligible layer; the engine leaves it off.)"""
    L = milib.get()
    code, td = DT["x3"]
    IH, IW, Ci, Co = geom
    k, B = 4, 3
    rng = np.random.RandomState(IH + Ci)
    prev = L.mi_set_tuning(21, 2)
    try:
        ws = torch.empty(96 << 20, device="cuda", dtype=torch.uint8)
        if form == "conv":
            OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
            x = rng.randn(B, IH, IW, Ci).astype(np.float32); dy = rng.randn(B, OH, OW, Co).astype(np.float32)
            xr, dyr = rounded(x, td), rounded(dy, td)
            w = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
            F.conv2d(_nchw(xr), w, stride=2).backward(_nchw(dyr))
            dwref = w.grad.permute(2, 3, 1, 0).numpy()                      # HWIO
            dbref = dyr.sum((0, 1, 2)).numpy()
            dw, db = torch.full((k, k, Ci, Co), 0.5, device="cuda"), torch.full((Co,), -1.0, device="cuda")
            xd, dyd = dev(x, td), dev(dy, td)                                # (kept alive: the launch is asynchronous)
            L.mi_conv2d_nhwc_wgrad_ws(stream(), code, xd.data_ptr(), None, 0, B, IH, IW, Ci, dyd.data_ptr(), k, k, Co, dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())
        else:                                                              # transposed conv: x [B, IH, IW, Ci] -> y [B, OH, OW, Co], kernel [kh, kw, co, ci]
            OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
            x = rng.randn(B, IH, IW, Ci).astype(np.float32); dy = rng.randn(B, OH, OW, Co).astype(np.float32)
            xr, dyr = rounded(x, td), rounded(dy, td)
            w = torch.zeros(Ci, Co, k, k, dtype=torch.float64, requires_grad=True)
            F.conv_transpose2d(_nchw(xr), w, stride=2).backward(_nchw(dyr))
            dwref = w.grad.permute(2, 3, 1, 0).numpy()                      # [kh, kw, co, ci]
            dbref = dyr.sum((0, 1, 2)).numpy()
            dw, db = torch.full((k, k, Co, Ci), 0.5, device="cuda"), torch.full((Co,), -1.0, device="cuda")
            xd, dyd = dev(x, td), dev(dy, td)
            L.mi_deconv2d_nhwc_wgrad_ws(stream(), code, dyd.data_ptr(), B, OH, OW, Co, xd.data_ptr(), k, k, Ci, dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())
        torch.cuda.synchronize()
        sw, sb = float(np.abs(dwref).max()), float(np.abs(dbref).max())
        assert_close(host(dw) - 0.5, dwref, 2e-5, 2e-5 * sw, "filter gradient (split operands on the bf16 kernel)")
        assert_close(host(db) + 1.0, dbref, 2e-5, 2e-5 * sb, "bias gradient")
    finally:
        L.mi_set_tuning(21, prev)


@pytest.mark.parametrize("B", [5, 37])
@pytest.mark.parametrize("form,geom,k", [("conv", (39, 79, 32, 64), 4), ("conv", (18, 38, 64, 128), 4), ("conv", (8, 18, 128, 256), 4),
                                         ("deconv", (3, 8, 256, 128), 4), ("deconv", (8, 18, 128, 64), 4), ("deconv", (18, 38, 64, 32), 5)])
def test_filter_gradient_dma_rows_decoded_once_per_wave(form, geom, k, B):
    """Round 6 (tapwgrad_tile.hpp, LDEC): the raw-staged filter-gradient kernels decode the source offsets of a step's DMA rows ONCE per wave -- one row per
    lane, a step ahead -- and every DMA instruction fetches its rows with one ds_bpermute_b32, instead of two magic-number divisions + the image bounds per lane and instruction.
    Same addresses, same zero fill, same summation order: all six layers of the model (both 2 x 2-tap forms and the k = 5 class-wave kernel, image edges, a ragged last step,
    several position splits) give BITWISE the filter and bias gradient of the per-instruction decode (mi_set_tuning key 24 = 0), and both meet the float64 product."""
    L = milib.get()
    code, td = DT["bf16"]
    IH, IW, Ci, Co = geom
    rng = np.random.RandomState(IH + Ci + B)
    ws = torch.empty(128 << 20, device="cuda", dtype=torch.uint8)
    x = rng.randn(B, IH, IW, Ci).astype(np.float32)
    if form == "conv":
        OH, OW = (IH - k) // 2 + 1, (IW - k) // 2 + 1
        wshape = (k, k, Ci, Co)
    else:
        OH, OW = (IH - 1) * 2 + k, (IW - 1) * 2 + k
        wshape = (k, k, Co, Ci)
    dy = rng.randn(B, OH, OW, Co).astype(np.float32)
    xr, dyr = rounded(x, td), rounded(dy, td)
    if form == "conv":
        w = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
        F.conv2d(_nchw(xr), w, stride=2).backward(_nchw(dyr))
    else:
        w = torch.zeros(Ci, Co, k, k, dtype=torch.float64, requires_grad=True)
        F.conv_transpose2d(_nchw(xr), w, stride=2).backward(_nchw(dyr))
    dwref = w.grad.permute(2, 3, 1, 0).numpy()
    dbref = dyr.sum((0, 1, 2)).numpy()
    xd, dyd = dev(x, td), dev(dy, td)
    res = {}
    prev_blocks = L.mi_set_tuning(9, 64)                                   # 64 target blocks: several position splits even at these small batches
    try:
        for mode in (0, 3):
            prev = L.mi_set_tuning(24, mode)
            try:
                dw, db = torch.full(wshape, 0.25, device="cuda"), torch.full((Co,), -2.0, device="cuda")
                ws.fill_(0x7f)                                               # (stale scratch must not matter)
                if form == "conv":
                    L.mi_conv2d_nhwc_wgrad_ws(stream(), code, xd.data_ptr(), None, 0, B, IH, IW, Ci, dyd.data_ptr(), k, k, Co, dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())
                else:
                    L.mi_deconv2d_nhwc_wgrad_ws(stream(), code, dyd.data_ptr(), B, OH, OW, Co, xd.data_ptr(), k, k, Ci, dw.data_ptr(), ws.data_ptr(), ws.numel(), db.data_ptr())
                torch.cuda.synchronize()
                res[mode] = (host(dw), host(db))
            finally:
                L.mi_set_tuning(24, prev)
    finally:
        L.mi_set_tuning(9, prev_blocks)
    assert np.array_equal(res[0][0], res[3][0]) and np.array_equal(res[0][1], res[3][1])
    sw, sb = float(np.abs(dwref).max()), float(np.abs(dbref).max())
    assert_close(res[3][0] - 0.25, dwref, 1e-4, 1e-4 * sw, "filter gradient (rows decoded once per wave)")
    assert_close(res[3][1] + 2.0, dbref, 1e-4, 1e-4 * sb, "bias gradient")


@pytest.mark.parametrize("B,IH,IW", [(3, 39, 79), (37, 39, 79), (2, 7, 15)])
def test_decoder_tail_fifth_slot_group_shared_by_three_waves(B, IH, IW):
    """Round 6 (dectail_tile.hpp; VERDICT r05 item 1: the ablation of tools/dectail_ablate.py named it): a tile computes 9 x 17 = 153 slots = FIVE groups of 32 for four waves; the fifth
    (25 live slots) was wave 0's second group while three waves waited at the barrier -- 12.9 of 76.7 us.  Waves 0, 1, 2 now each run the fifth group's MFMAs and the loss of ONE
    of its three logit pairs.  Same logits, same dlogits tile: the input gradient is BITWISE that of the one-wave form (mi_set_tuning key 26 = 0), the filter gradient too (same
    patches, same order); loss and bias-gradient partial sums are regrouped (fp32 summation order)."""
    import ctypes
    L = milib.get()
    code, td = DT["bf16"]
    Ci, Co, k = 32, 3, 4
    OH, OW = 2 * IH + 2, 2 * IW + 2
    rng = np.random.RandomState(B + IH)
    x = np.maximum(rng.randn(B, IH, IW, Ci), 0).astype(np.float32)
    w = (rng.randn(k, k, Co, Ci) / np.sqrt(4 * Ci)).astype(np.float32)
    b = (0.1 * rng.randn(Co)).astype(np.float32)
    frames_u8 = rng.randint(0, 256, (B + 2, OH * OW * Co)).astype(np.uint8)
    idx = rng.permutation(B + 2)[:B].astype(np.int32)
    labels = dev(frames_u8, torch.uint8)
    xd, wd, bd, idxd = dev(x, td), dev(w, td), dev(b), dev(idx, torch.int32)
    wt = torch.zeros(k * k * Co * Ci, device="cuda", dtype=td)
    offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Co], np.int32), np.array([Ci], np.int32)
    L.mi_transpose_weights(stream(), code, P(dev(w)), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
    nb = L.mi_deconv2d_tail_blocks()
    res = {}
    for mode in (0, 1):
        prev = L.mi_set_tuning(26, mode)
        try:
            scratch = torch.empty(nb * 6144, device="cuda", dtype=torch.uint8)
            lp, bp = torch.zeros(16384, device="cuda"), torch.zeros(16384, 4, device="cuda")
            dx = alloc(td, B, IH, IW, Ci, fill=5.0)
            dw = torch.zeros(k, k, Co, Ci, device="cuda")
            n = ctypes.c_int(0)
            L.mi_deconv2d_tail_fused(stream(), code, xd.data_ptr(), B, IH, IW, Ci, wd.data_ptr(), wt.data_ptr(), bd.data_ptr(), k, k, Co, labels.data_ptr(), 1, idxd.data_ptr(),
                                     OH * OW * Co, 0, 1.0 / 16, dx.data_ptr(), dw.data_ptr(), lp.data_ptr(), bp.data_ptr(), 16384, ctypes.addressof(n), scratch.data_ptr(), scratch.numel(), 1)
            torch.cuda.synchronize()
            res[mode] = (dx.view(torch.int16).clone(), dw.clone(), float(lp[:n.value].double().sum()), bp[:n.value, :Co].double().sum(0).cpu().numpy())
        finally:
            L.mi_set_tuning(26, prev)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert abs(res[1][2] / res[0][2] - 1) < 2e-6
    assert_close(res[1][3], res[0][3], 1e-5, 1e-5 * float(np.abs(res[0][3]).max()) + 1e-7, "bias gradient sums")


@pytest.mark.parametrize("B", [3, 40])
@pytest.mark.parametrize("which", ["conv3.fwd", "deconv2.dgrad", "deconv3.dgrad"])
def test_register_weight_conv_reads_fragment_ordered_weights(which, B):
    """Round 6 (rwconv.hip, WFRAG): rwconv_conv_kernel<4, 2> (64 -> 128 channels, k = 4: conv3's forward pass, deconv2's input gradient) loads its 64 weight fragments per wave ONCE per
    block; from the K-contiguous copy every load touches 32 rows x 32 B at a 2 KB pitch, from the fragment-ordered copy (mi_ares_pack_weights8 form 3; the optimiser launch emits it)
    1 KB contiguous.  Same values in the same registers: the output is BITWISE that of the K-contiguous form, with ReLU + bias (forward) and with a ReluGrad mask (input gradient)."""
    L = milib.get()
    code, td = DT["bf16"]
    rng = np.random.RandomState(B + len(which))
    prev = {k: L.mi_set_tuning(k, v) for k, v in ((13, 2), (15, 3))}          # the register-weight kernels whenever eligible (auto takes them only on chip-filling grids)
    try:
        if which == "conv3.fwd":
            IH, IW, Ci, Co, k = 18, 38, 64, 128, 4
            OH, OW = 8, 18
            x = np.maximum(rng.randn(B, IH, IW, Ci), 0).astype(np.float32)
            w = (rng.randn(k, k, Ci, Co) / np.sqrt(k * k * Ci)).astype(np.float32)           # HWIO = [K = 1024][N = 128]
            b = (0.1 * rng.randn(Co)).astype(np.float32)
            xd, bd, wmaster = dev(x, td), dev(b), dev(w)
            wt = alloc(td, k * k * Ci * Co, fill=0.0)
            offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Ci], np.int32), np.array([Co], np.int32)
            L.mi_transpose_weights(stream(), code, wmaster.data_ptr(), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
            outs = []
            for frag in (False, True):
                out = alloc(td, B, OH, OW, Co, fill=3.0)
                if frag:
                    L.mi_rwconv_next_weights_fragment_ordered(wfrag.data_ptr())
                L.mi_conv2d_nhwc_fwd(stream(), code, xd.data_ptr(), None, 0, B, IH, IW, Ci, wt.data_ptr(), 1, bd.data_ptr(), k, k, Co, 1, out.data_ptr())
                torch.cuda.synchronize()
                outs.append(out.view(torch.int16).clone())
                if not frag:
                    wfrag = torch.zeros(1 << 20, device="cuda", dtype=torch.uint8)
                    junk = [torch.zeros(1 << 20, device="cuda", dtype=torch.uint8) for _ in range(4)]
                    big = dev(np.zeros((4, 4, 128, 256), np.float32))
                    L.mi_ares_pack_weights8(stream(), big.data_ptr(), big.data_ptr(), wmaster.data_ptr(), None, junk[0].data_ptr(), junk[1].data_ptr(), junk[2].data_ptr(), junk[3].data_ptr(),
                                            None, None, wfrag.data_ptr(), None)
        elif which == "deconv3.dgrad":
            IH, IW, Ci, Co, k = 18, 38, 64, 32, 5                                              # deconv3: x [B, 18, 38, 64] -> y [B, 39, 79, 32]; its input gradient is a k5 s2 conv of dy (rwconv_conv_kernel<5, 1>, pack form 6)
            OH, OW = 39, 79
            dy = rng.randn(B, OH, OW, Co).astype(np.float32)
            xmask = rng.randn(B, IH, IW, Ci).astype(np.float32)
            w = (rng.randn(k, k, Co, Ci) / np.sqrt(k * k * Co / 4)).astype(np.float32)       # [kh, kw, out = 32, in = 64] = [K = 800][N = 64]
            dyd, md, wmaster = dev(dy, td), dev(xmask, td), dev(w)
            wt = alloc(td, k * k * Co * Ci, fill=0.0)
            offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Co], np.int32), np.array([Ci], np.int32)
            L.mi_transpose_weights(stream(), code, wmaster.data_ptr(), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
            wfrag = torch.zeros(1 << 20, device="cuda", dtype=torch.uint8)
            L.mi_ares_pack_weights(stream(), 6, wmaster.data_ptr(), wfrag.data_ptr())
            outs = []
            for frag in (False, True):
                dx = alloc(td, B, IH, IW, Ci, fill=3.0)
                if frag:
                    L.mi_rwconv_next_weights_fragment_ordered(wfrag.data_ptr())
                L.mi_deconv2d_nhwc_dgrad(stream(), code, dyd.data_ptr(), B, OH, OW, Co, wt.data_ptr(), 1, k, k, Ci, md.data_ptr(), dx.data_ptr())
                torch.cuda.synchronize()
                outs.append(dx.view(torch.int16).clone())
        else:
            IH, IW, Ci, Co, k = 8, 18, 128, 64, 4                                              # deconv2: x [B, 8, 18, 128] -> y [B, 18, 38, 64]; its input gradient is a k4 s2 conv of dy
            OH, OW = 18, 38
            dy = rng.randn(B, OH, OW, Co).astype(np.float32)
            xmask = rng.randn(B, IH, IW, Ci).astype(np.float32)
            w = (rng.randn(k, k, Co, Ci) / np.sqrt(k * k * Co)).astype(np.float32)           # [kh, kw, out = 64, in = 128] = [K = 1024][N = 128]
            dyd, md, wmaster = dev(dy, td), dev(xmask, td), dev(w)
            wt = alloc(td, k * k * Co * Ci, fill=0.0)
            offs, Ks, Ns = np.array([0], np.int64), np.array([k * k * Co], np.int32), np.array([Ci], np.int32)
            L.mi_transpose_weights(stream(), code, wmaster.data_ptr(), wt.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
            outs = []
            for frag in (False, True):
                dx = alloc(td, B, IH, IW, Ci, fill=3.0)
                if frag:
                    L.mi_rwconv_next_weights_fragment_ordered(wfrag.data_ptr())
                L.mi_deconv2d_nhwc_dgrad(stream(), code, dyd.data_ptr(), B, OH, OW, Co, wt.data_ptr(), 1, k, k, Ci, md.data_ptr(), dx.data_ptr())
                torch.cuda.synchronize()
                outs.append(dx.view(torch.int16).clone())
                if not frag:
                    wfrag = torch.zeros(1 << 20, device="cuda", dtype=torch.uint8)
                    junk = [torch.zeros(1 << 20, device="cuda", dtype=torch.uint8) for _ in range(4)]
                    big = dev(np.zeros((4, 4, 128, 256), np.float32))
                    L.mi_ares_pack_weights8(stream(), big.data_ptr(), big.data_ptr(), None, wmaster.data_ptr(), junk[0].data_ptr(), junk[1].data_ptr(), junk[2].data_ptr(), junk[3].data_ptr(),
                                            None, None, None, wfrag.data_ptr())
        assert torch.equal(outs[0], outs[1])
        assert float(outs[0].float().abs().max()) > 0 and not bool((outs[0] == outs[0].flatten()[0]).all())      # (something was computed)
        L.mi_rwconv_next_weights_fragment_ordered(None)
    finally:
        for k_, v in prev.items():
            L.mi_set_tuning(k_, v)


@pytest.mark.parametrize("B", [2, 24])
def test_deconv3_forward_and_fused_encoder_head_read_fragment_ordered_weights(B):
    """Round 6: the gather-form register-weight kernel of deconv3's forward pass (k = 5, 64 -> 32 channels; pack form 4) and the fused encoder head's conv2 stage (pack form 5) load their
    weight registers from fragment-ordered copies when one is announced (mi_rwconv_next_weights_fragment_ordered): 1 KB contiguous per wave load in the prologue.  Same values in the
    same registers: outputs BITWISE equal to the K-contiguous / TF-layout forms."""
    import ctypes
    L = milib.get()
    code, td = DT["bf16"]
    rng = np.random.RandomState(B)
    prev = L.mi_set_tuning(13, 2)                          # the gather-form register-weight kernel whenever eligible
    try:
        # ---- deconv3 forward: x [B, 18, 38, 64] -> y [B, 39, 79, 32], kernel [5, 5, 32, 64] ----
        IH, IW, Ci, Co, k = 18, 38, 64, 32, 5
        OH, OW = 39, 79
        x = np.maximum(rng.randn(B, IH, IW, Ci), 0).astype(np.float32)
        w = (rng.randn(k, k, Co, Ci) / np.sqrt(k * k * Ci / 4)).astype(np.float32)
        b = (0.1 * rng.randn(Co)).astype(np.float32)
        xd, wd, wmaster, bd = dev(x, td), dev(w, td), dev(w), dev(b)
        wfrag = torch.zeros(1 << 20, device="cuda", dtype=torch.uint8)
        L.mi_ares_pack_weights(stream(), 4, wmaster.data_ptr(), wfrag.data_ptr())
        outs = []
        for frag in (False, True):
            out = alloc(td, B, OH, OW, Co, fill=3.0)
            if frag:
                L.mi_rwconv_next_weights_fragment_ordered(wfrag.data_ptr())
            L.mi_deconv2d_nhwc_fwd(stream(), code, xd.data_ptr(), B, IH, IW, Ci, wd.data_ptr(), bd.data_ptr(), k, k, Co, 1, out.data_ptr())
            torch.cuda.synchronize()
            outs.append(out.view(torch.int16).clone())
        assert torch.equal(outs[0], outs[1]) and float(outs[0].float().abs().max()) > 0
        L.mi_rwconv_next_weights_fragment_ordered(None)
    finally:
        L.mi_set_tuning(13, prev)
    # ---- the fused encoder head: conv2's kernel [4, 4, 32, 64] in fragment order ----
    frames = torch.from_numpy(rng.randint(0, 256, (B + 3, 80, 160, 3)).astype(np.uint8)).cuda()
    idx = torch.from_numpy(rng.permutation(B + 3)[:B].astype(np.int32)).cuda()
    w1 = (rng.randn(4, 4, 3, 32) / 7).astype(np.float32); w2 = (rng.randn(4, 4, 32, 64) / 22).astype(np.float32)
    b1d, b2d = dev((0.1 * rng.randn(32)).astype(np.float32)), dev((0.1 * rng.randn(64)).astype(np.float32))
    w1t, w2t = alloc(td, 32 * 48, fill=0.0), alloc(td, 64 * 512, fill=0.0)
    for wm, wt_, K_, N_ in ((w1, w1t, 48, 32), (w2, w2t, 512, 64)):
        offs, Ks, Ns = np.array([0], np.int64), np.array([K_], np.int32), np.array([N_], np.int32)
        L.mi_transpose_weights(stream(), code, P(dev(wm)), wt_.data_ptr(), offs.ctypes.data, Ks.ctypes.data, Ns.ctypes.data, 1)
    w2master = dev(w2)
    w2frag = torch.zeros(1 << 17, device="cuda", dtype=torch.uint8)
    L.mi_ares_pack_weights(stream(), 5, w2master.data_ptr(), w2frag.data_ptr())
    res = []
    for frag in (False, True):
        act1 = alloc(td, B, 39, 79, 32, fill=0.0); bits = torch.zeros(B * 39 * 79 * 2, device="cuda", dtype=torch.int32); act2 = alloc(td, B, 18, 38, 64, fill=0.0)
        launched = ctypes.c_int(0)
        if frag:
            L.mi_rwconv_next_weights_fragment_ordered(w2frag.data_ptr())
        L.mi_conv2d_enc12_fwd(stream(), code, frames.data_ptr(), 2, idx.data_ptr(), B, 80, 160, w1t.data_ptr(), b1d.data_ptr(), w2t.data_ptr(), b2d.data_ptr(),
                              act1.data_ptr(), bits.data_ptr(), act2.data_ptr(), ctypes.addressof(launched))
        torch.cuda.synchronize()
        assert launched.value == 1
        res.append((act1.view(torch.int16).clone(), bits.clone(), act2.view(torch.int16).clone()))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    assert float(res[0][2].float().abs().max()) > 0


@pytest.mark.parametrize("B", [48, 512])
def test_two_dense_filter_gradients_in_one_launch_equal_the_two_launches(B):
    """Round 6 (mi_gemm_wgrad_bias_pair_ws): dense1's ([B, 64]^T [B, 6144]) and the heads' ([B, 6144]^T [B, 128]) filter + bias gradients as ONE launch of the first-generation kernel --
    every block computes what it would have in its own launch, the ordered slab sums are the same: BITWISE the two single calls; a pair the one-launch form does not cover (fp32)
    runs as the two calls."""
    L = milib.get()
    for tag in ("bf16", "f32"):
        code, td = DT[tag]
        rng = np.random.RandomState(B)
        z = rng.randn(B, 64).astype(np.float32); g1 = rng.randn(B, 6144).astype(np.float32)
        a4 = np.maximum(rng.randn(B, 6144), 0).astype(np.float32); gh = rng.randn(B, 128).astype(np.float32)
        zd, g1d, a4d, ghd = dev(z, td), dev(g1, td), dev(a4, td), dev(gh, td)
        nb0, nb1 = L.mi_gemm_wgrad_scratch_bytes(code, B, 64, 6144), L.mi_gemm_wgrad_scratch_bytes(code, B, 6144, 128)
        ws0 = torch.zeros(max(nb0, 16), device="cuda", dtype=torch.uint8); ws1 = torch.zeros(max(nb1, 16), device="cuda", dtype=torch.uint8)
        res = []
        for pair in (False, True):
            dw0 = torch.zeros(64, 6144, device="cuda"); db0 = torch.zeros(6144, device="cuda"); dw1 = torch.zeros(6144, 128, device="cuda"); db1 = torch.zeros(128, device="cuda")
            if pair:
                rc = L.mi_gemm_wgrad_bias_pair_ws(stream(), code, zd.data_ptr(), g1d.data_ptr(), B, 64, 6144, dw0.data_ptr(), db0.data_ptr(), ws0.data_ptr(), nb0,
                                                  a4d.data_ptr(), ghd.data_ptr(), B, 6144, 128, dw1.data_ptr(), db1.data_ptr(), ws1.data_ptr(), nb1)
            else:
                rc = L.mi_gemm_wgrad_bias_ws(stream(), code, zd.data_ptr(), g1d.data_ptr(), B, 64, 6144, dw0.data_ptr(), db0.data_ptr(), ws0.data_ptr(), nb0)
                assert rc == 0
                rc = L.mi_gemm_wgrad_bias_ws(stream(), code, a4d.data_ptr(), ghd.data_ptr(), B, 6144, 128, dw1.data_ptr(), db1.data_ptr(), ws1.data_ptr(), nb1)
            assert rc == 0, L.mi_last_error()
            torch.cuda.synchronize()
            res.append([t.clone() for t in (dw0, db0, dw1, db1)])
        if tag == "bf16" or nb0 > 0:                       # (fp32 with one row split adds by atomics: order-dependent last bits, compared by value below)
            for x, y in zip(res[0], res[1]):
                if tag == "bf16":
                    assert torch.equal(x, y)
        zb, g1b = zd.float(), g1d.float()
        want = zb.t() @ g1b
        assert float((res[1][0] - want).abs().max()) <= 2e-3 * float(want.abs().max()) + 1e-4
        assert float((res[1][3] - ghd.float().sum(0)).abs().max()) <= 2e-3 * float(ghd.float().sum(0).abs().max()) + 1e-3
