"""GPU: every HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances: fp32 kernels vs fp32 CPU ops, different summation order -> relative L2
1e-5..1e-4 per tensor (stated per test); the end-to-end 1e-3 bar of BASELINE.json is
checked in test_network_gpu.py.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import med3d_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from bodyct_dram_emph_subtype_amd import ops as o
    import bodyct_dram_emph_subtype_amd as pkg
    pkg.load_library()
    return o


def to_ndhwc(t):  # NCDHW cpu -> NDHWC gpu
    return t.permute(0, 2, 3, 4, 1).contiguous().to(DEV)


def to_ncdhw(t):  # NDHWC gpu -> NCDHW cpu
    return t.permute(0, 4, 1, 2, 3).contiguous().cpu()


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


CONV_CASES = [
    # B, D, H, W, Cin, Cout, k, stride, dil
    (1, 6, 8, 10, 32, 32, 3, 1, 1),
    (2, 8, 8, 8, 64, 64, 3, 1, 1),
    (1, 5, 9, 7, 64, 128, 3, 1, 2),
    (1, 6, 10, 9, 64, 64, 3, 1, 4),
    (1, 8, 12, 10, 64, 128, 3, 2, 1),
    (2, 10, 18, 34, 128, 64, 3, 2, 1),          # stride 2: several tiles per lattice class, two N tiles in the data gradient
    (1, 9, 11, 13, 64, 64, 3, 2, 1),            # stride 2 on odd extents (the sub-lattices differ in size)
    (2, 4, 6, 5, 128, 64, 1, 1, 1),
    (1, 8, 16, 16, 96, 32, 3, 1, 1),
    (1, 3, 4, 5, 256, 256, 3, 1, 4),
]


def conv_ref(x, w, b, k, stride, dil):
    pad = dil * (k - 1) // 2
    return F.conv3d(x, w, b, stride, pad, dil)


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3d_fwd_bwd(ops, monkeypatch, case):
    monkeypatch.setenv("DRAM_CONV_ALGO", "1")          # the direct kernels (Winograd: test_conv3d_winograd_path)
    B, D, H, W, Cin, Cout, k, stride, dil = case
    pad = dil * (k - 1) // 2
    x = rnd(B, Cin, D, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, k, k, k, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = conv_ref(x, w, bias, k, stride, dil)
    gy = rnd(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy)

    g = ops.ConvGeom(B, D, H, W, Cin, Cout, k, stride, pad, dil)
    assert g.out_shape == (B,) + tuple(y_ref.shape[2:]) + (Cout,)
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV))
    xd = to_ndhwc(x.detach())
    y, stats = ops.conv3d_fwd(xd, wf, bias.to(DEV), g, True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < 2e-6
    # fused BN statistics epilogue
    s = ops.reduce_partials(stats).cpu()
    yr = y_ref.detach().double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    # no-bias / no-stats variant
    y2, st2 = ops.conv3d_fwd(xd, wf, None, g, False)
    assert st2 is None and rel_l2(to_ncdhw(y2), conv_ref(x, w, None, k, stride, dil).detach()) < 2e-6

    gyd = to_ndhwc(gy)
    dx = ops.conv3d_bwd_data(gyd, wb, g)
    assert rel_l2(to_ncdhw(dx), gx_ref) < 2e-6
    add = rnd(B, Cin, D, H, W, seed=5)
    gate = rnd(B, Cin, D, H, W, seed=6)
    dx2 = ops.conv3d_bwd_data(gyd, wb, g, to_ndhwc(add), to_ndhwc(gate))
    assert rel_l2(to_ncdhw(dx2), gx_ref + add * (gate > 0).float()) < 2e-6
    dx3 = ops.conv3d_bwd_data(gyd, wb, g, to_ndhwc(add), None)
    assert rel_l2(to_ncdhw(dx3), gx_ref + add) < 2e-6

    if Cin % 64 == 0:
        dw = ops.conv3d_bwd_weight(xd, gyd, g)
        assert rel_l2(dw.cpu(), gw_ref) < 5e-6


C1_CASES = [
    # B, D, H, W, Cin, Cout    (1x1x1 convolutions with a 256-multiple of voxels: the batched-GEMM kernels)
    (1, 4, 8, 8, 64, 64),
    (2, 8, 8, 8, 128, 256),
    (1, 8, 16, 16, 256, 64),
    (1, 8, 16, 32, 512, 192),           # Cout only a multiple of 64; the weight gradient splits over voxels
]


@pytest.mark.parametrize("case", C1_CASES, ids=[str(c) for c in C1_CASES])
def test_conv1x1_gemm_path(ops, monkeypatch, case):
    """Bottleneck 1x1x1 convolutions as plain GEMMs (plan 3): forward with bias + fused BN sums, data
    gradient with the fused shortcut-gradient epilogue, weight gradient; and agreement with the direct kernel."""
    monkeypatch.delenv("DRAM_CONV_ALGO", raising=False)
    B, D, H, W, Cin, Cout = case
    x = rnd(B, Cin, D, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, 1, 1, 1, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x, w, bias)
    gy = rnd(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 1, 1, 0, 1)
    assert ops.conv_algo(g) == 3
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV), True, True, g)
    assert wf.shape == (1, Cout, Cin) and wb.shape == (1, Cin, Cout)
    xd, gyd = to_ndhwc(x.detach()), to_ndhwc(gy)
    y, stats = ops.conv3d_fwd(xd, wf, bias.to(DEV), g, True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < 2e-6
    s = ops.reduce_partials(stats).cpu()
    yr = y_ref.detach().double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    y2, st2 = ops.conv3d_fwd(xd, wf, None, g, False)
    assert st2 is None and rel_l2(to_ncdhw(y2), F.conv3d(x, w).detach()) < 2e-6
    add = rnd(B, Cin, D, H, W, seed=5)
    gate = rnd(B, Cin, D, H, W, seed=6)
    dx = ops.conv3d_bwd_data(gyd, wb, g)
    assert rel_l2(to_ncdhw(dx), gx_ref) < 2e-6
    dx2 = ops.conv3d_bwd_data(gyd, wb, g, to_ndhwc(add), to_ndhwc(gate))
    assert rel_l2(to_ncdhw(dx2), gx_ref + add * (gate > 0).float()) < 2e-6
    dw = ops.conv3d_bwd_weight(xd, gyd, g)
    assert rel_l2(dw.cpu(), gw_ref) < 5e-6
    assert torch.equal(dw, ops.conv3d_bwd_weight(xd, gyd, g))
    monkeypatch.setenv("DRAM_CONV_ALGO", "1")
    y1, _ = ops.conv3d_fwd(xd, wf, bias.to(DEV), g, False)        # same packed layout (taps = 1): direct kernel
    assert rel_l2(y.cpu(), y1.cpu()) < 2e-6


V3_CASES = [
    # forced v3 plan "tz3,bn", then (B, D, H, W, Cin, Cout, dil)
    ("8,64", (1, 9, 10, 17, 64, 64, 1)),
    ("8,64", (2, 6, 8, 8, 32, 128, 2)),
    ("8,32", (1, 8, 8, 9, 64, 32, 1)),
    ("4,128", (1, 8, 16, 16, 64, 128, 2)),
    ("4,128", (2, 5, 9, 10, 32, 256, 1)),
    ("4,256", (1, 4, 8, 8, 64, 256, 1)),
    ("4,256", (1, 16, 32, 32, 64, 512, 4)),
]


@pytest.mark.parametrize("plan,case", V3_CASES, ids=[f"{p}-{c}" for p, c in V3_CASES])
def test_conv3d_v3_halo_kernel_variants(ops, monkeypatch, plan, case):
    """Every template instance of the LDS-resident-halo kernel (8x8x8 and 4x8x8 tiles, 32..256
    columns per workgroup), forward with fused BN statistics and data gradient with the fused
    shortcut-gradient epilogue, incl. ragged volumes and dilation lattices."""
    monkeypatch.setenv("DRAM_IGEMM_V3_FORCE", plan)
    monkeypatch.setenv("DRAM_CONV_ALGO", "1")
    B, D, H, W, Cin, Cout, dil = case
    x = rnd(B, Cin, D, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x, w, bias, 1, dil, dil)
    gy = rnd(*y_ref.shape, seed=4)
    (gx_ref,) = torch.autograd.grad(y_ref, [x], gy)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV))
    y, stats = ops.conv3d_fwd(to_ndhwc(x.detach()), wf, bias.to(DEV), g, True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < 2e-6
    s = ops.reduce_partials(stats).cpu()
    yr = y_ref.detach().double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    if Cin % 64 == 0 or Cin == 32:       # the data gradient writes N = Cin channels
        bn = int(plan.split(",")[1])
        if Cin % bn == 0:
            add = rnd(B, Cin, D, H, W, seed=5)
            gate = rnd(B, Cin, D, H, W, seed=6)
            dx = ops.conv3d_bwd_data(to_ndhwc(gy), wb, g, to_ndhwc(add), to_ndhwc(gate))
            assert rel_l2(to_ncdhw(dx), gx_ref + add * (gate > 0).float()) < 2e-6


WINO_CASES = [
    # B, D, H, W, Cin, Cout, dil      (DRAM_CONV_ALGO=2: Winograd wherever applicable)
    (1, 8, 8, 8, 64, 64, 1),
    (2, 7, 10, 9, 64, 128, 1),          # ragged: odd extents
    (1, 9, 12, 10, 128, 64, 2),         # dilation lattice, ragged residues
    (1, 5, 7, 6, 64, 256, 1),           # weight gradient on the TN GEMM, 64-column tiles
    (1, 16, 16, 16, 128, 256, 2),       # TN GEMM, 128-column tiles, split over t
    (1, 8, 16, 16, 256, 256, 4),        # layer3/4-like: 256-column tiles, dilation 4
    (2, 4, 8, 8, 192, 320, 1),          # channel counts that are only multiples of 64
    (1, 2, 5, 3, 64, 64, 4),            # dilation > extent: EMPTY residue sub-lattices (tiles whose origin lies outside)
]


@pytest.mark.parametrize("tiling", ["2,2,2", "4,2,2", "4,4,2", "4,4,4"], ids=["F222", "F422", "F442", "F444"])
@pytest.mark.parametrize("case", WINO_CASES, ids=[str(c) for c in WINO_CASES])
def test_conv3d_winograd_path(ops, monkeypatch, case, tiling):
    """Winograd F(2x2x2,3x3x3) pipeline (tile transforms + batched NN / TN GEMMs) against
    F.conv3d and its autograd: forward with bias + fused BN sums, data gradient with the fused
    shortcut-gradient epilogue, weight gradient."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "2")
    nz, ny, nx = tiling.split(",")                    # outputs per tile along z, y, x: F(2,3) or F(4,3) per axis
    monkeypatch.setenv("DRAM_WINO_TILING", tiling)
    tol = 3e-5 if tiling == "4,4,4" else 1e-5         # F(4,3) on all three axes: ~1e-5 vs fp64 (F(2,3): 7e-7)
    B, D, H, W, Cin, Cout, dil = case
    x = rnd(B, Cin, D, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x, w, bias, 1, dil, dil)
    gy = rnd(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    assert ops.conv_use_wino(g)
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV), True, True, g)
    npts = (int(nz) + 2) * (int(ny) + 2) * (int(nx) + 2)
    assert wf.shape == (npts, Cout, Cin) and wb.shape == (npts, Cin, Cout)
    xd, gyd = to_ndhwc(x.detach()), to_ndhwc(gy)
    y, stats = ops.conv3d_fwd(xd, wf, bias.to(DEV), g, True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < tol
    s = ops.reduce_partials(stats).cpu()
    yr = y_ref.detach().double()
    cnt, rms = B * D * H * W, float(yr.std())          # coherent-error bound on a sum of cnt values with rel. error tol
    assert float((s[0] - yr.sum((0, 2, 3, 4))).abs().max()) <= 2 * tol * rms * cnt + 1e-3
    assert float((s[1] - (yr * yr).sum((0, 2, 3, 4))).abs().max()) <= 4 * tol * rms * rms * cnt + 1e-3
    add = rnd(B, Cin, D, H, W, seed=5)
    gate = rnd(B, Cin, D, H, W, seed=6)
    dx = ops.conv3d_bwd_data(gyd, wb, g)
    assert rel_l2(to_ncdhw(dx), gx_ref) < tol
    dx2 = ops.conv3d_bwd_data(gyd, wb, g, to_ndhwc(add), to_ndhwc(gate))
    assert rel_l2(to_ncdhw(dx2), gx_ref + add * (gate > 0).float()) < tol
    dw = ops.conv3d_bwd_weight(xd, gyd, g)
    assert rel_l2(dw.cpu(), gw_ref) < 2 * tol
    # the forward pass can hand its transformed input to the weight gradient (bitwise same result)
    y_k, _, v = ops.conv3d_fwd_keep(xd, wf, bias.to(DEV), g, False, True)
    assert v is not None and torch.equal(y_k, y)
    assert torch.equal(ops.conv3d_bwd_weight(xd, gyd, g, v_cache=v), dw)
    # the direct path on the same inputs (plan switched off) agrees with the Winograd result
    monkeypatch.setenv("DRAM_CONV_ALGO", "1")
    wf1, _ = ops.pack_conv_weight(w.detach().to(DEV), True, False, g)
    y1, _ = ops.conv3d_fwd(xd, wf1, bias.to(DEV), g, False)
    assert rel_l2(y.cpu(), y1.cpu()) < tol


@pytest.mark.parametrize("case", WINO_CASES[:3], ids=[str(c) for c in WINO_CASES[:3]])
def test_winograd_streaming_gemm_equals_tiled_gemm(ops, monkeypatch, case):
    """The persistent streaming form of the Winograd-domain NN GEMM (K, N <= 128: whole B operand in registers, A tiles
    through an LDS ring; the library takes it from 4 096 (point, 64-row tile) items on) accumulates in the order of the
    tiled kernel: forward and data gradient are bit-identical to it, on every tiling -- (N, K) = (64, 64), (128, 64)
    and (64, 128) are its three instantiations."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "2")
    B, D, H, W, Cin, Cout, dil = case
    x = to_ndhwc(rnd(B, Cin, D, H, W, seed=1))
    gy = to_ndhwc(rnd(B, Cout, D, H, W, seed=4))
    w = (rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1).to(DEV)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    for tiling in ("2,2,2", "4,4,4"):
        monkeypatch.setenv("DRAM_WINO_TILING", tiling)
        wf, wb = ops.pack_conv_weight(w, True, True, g)
        res = {}
        for mode in ("0", "2"):
            monkeypatch.setenv("DRAM_NN_STREAM", mode)
            res[mode] = (ops.conv3d_fwd(x, wf, None, g, False)[0], ops.conv3d_bwd_data(gy, wb, g))
        assert torch.equal(res["0"][0], res["2"][0]) and torch.equal(res["0"][1], res["2"][1])
        # (64, 64) has two forms: B fragments loaded directly into registers, two workgroups per CU (the default), and
        # the LDS-image form with a four-stage ring
        monkeypatch.setenv("DRAM_NN_STREAM_DB", "0")
        assert torch.equal(ops.conv3d_fwd(x, wf, None, g, False)[0], res["0"][0])
        assert torch.equal(ops.conv3d_bwd_data(gy, wb, g), res["0"][1])
        monkeypatch.delenv("DRAM_NN_STREAM_DB")
        # ... and a data gradient that says it shares the device with another stream takes the one-workgroup form
        assert torch.equal(ops.conv3d_bwd_data(gy, wb, g, overlapped=True), res["0"][1])
    ref = F.conv3d(rnd(B, Cin, D, H, W, seed=1).double(), w.cpu().double(), None, 1, dil, dil)
    assert rel_l2(to_ncdhw(res["2"][0]).double(), ref) < 3e-5


@pytest.mark.parametrize("case", [(1, 8, 16, 16, 64, 64, 64), (2, 8, 8, 16, 128, 64, 64), (1, 6, 10, 12, 64, 128, 64)], ids=str)
def test_conv_of_two_channel_blocks_equals_conv_of_their_concatenation(ops, monkeypatch, case):
    """crop_concat_5d feeding conv_blocks[0] (reference med3d.py:39-48, :87, :67) without the concatenated tensor: the
    up-sampled half alone (ops.up_fwd == the first Cu channels of upcat_fwd, bit for bit) and the convolution reading the
    two blocks as two sources (ops.conv3d_fwd_cat) give the output, the BatchNorm partial sums and the kept
    Winograd-domain image of the convolution of the materialised concatenation, bit for bit (ragged tiles included)."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "2")
    monkeypatch.setenv("DRAM_WINO_TILING", "4,4,4")
    B, Ds, Hs, Ws, Cu, Cs, Co = case
    src = to_ndhwc(rnd(B, Cu, Ds, Hs, Ws, seed=1))
    skip = to_ndhwc(rnd(B, Cs, 2 * Ds, 2 * Hs, 2 * Ws, seed=2))
    cat = ops.upcat_fwd(src, skip)
    up = ops.up_fwd(src)
    assert torch.equal(up, cat[..., :Cu].contiguous())
    g = ops.ConvGeom(B, 2 * Ds, 2 * Hs, 2 * Ws, Cu + Cs, Co, 3, 1, 1, 1)
    assert ops.conv_cat_ok(g, Cu)
    w = (rnd(Co, Cu + Cs, 3, 3, 3, seed=3) * 0.1).to(DEV)
    bias = rnd(Co, seed=4).to(DEV)
    wf, _ = ops.pack_conv_weight(w, True, False, g)
    y0, s0, v0 = ops.conv3d_fwd_keep(cat, wf, bias, g, True, True)
    y1, s1, v1 = ops.conv3d_fwd_cat(up, skip, wf, bias, g, True, True)
    assert torch.equal(y0, y1) and torch.equal(s0, s1) and torch.equal(v0, v1)
    gy = to_ndhwc(rnd(B, Co, 2 * Ds, 2 * Hs, 2 * Ws, seed=5))
    assert torch.equal(ops.conv3d_bwd_weight(None, gy, g, v_cache=v1), ops.conv3d_bwd_weight(cat, gy, g, v_cache=v0))
    monkeypatch.setenv("DRAM_WINO_TILING", "4,4,2")            # other tilings: the engine materialises the concatenation
    assert not ops.conv_cat_ok(g, Cu)


BNSTAT_CASES = [
    # B, D, H, W, Cin (= channels of the unit in front), Cout, dil, tiling
    (2, 8, 16, 16, 64, 64, 1, "4,4,4"),
    (1, 6, 10, 14, 128, 64, 1, "4,4,4"),       # ragged tiles: out-of-volume outputs must not reach the sums
    (1, 8, 8, 16, 64, 128, 2, "4,4,4"),        # dilation lattice (layer3 / layer4)
    (1, 6, 12, 12, 64, 64, 1, "2,4,4"),        # an unrolled tiling
]


@pytest.mark.parametrize("case", BNSTAT_CASES, ids=[str(c) for c in BNSTAT_CASES])
def test_data_gradient_takes_the_batchnorm_backward_statistics_of_the_unit_in_front(ops, monkeypatch, case):
    """Backward of conv <- relu <- bn (reference med3d.py:121-124, :153-161): the data gradient's output transform also
    writes the rows (sum g, sum g * xhat), g = dx * (y*scale + shift > 0), of the unit whose output the convolution read
    (ops.conv3d_bwd_data_bnstats).  dx is bit-identical to the plain data gradient; the folded sums equal the folded sums
    of the separate pass (ops.bn_bwd_reduce) to fp32 summation-order accuracy."""
    B, D, H, W, Cin, Cout, dil, tiling = case
    monkeypatch.setenv("DRAM_CONV_ALGO", "2")
    monkeypatch.setenv("DRAM_WINO_TILING", tiling)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    assert ops.conv_bwd_bnstats_ok(g)
    w = (rnd(Cout, Cin, 3, 3, 3, seed=1) * 0.1).to(DEV)
    _, wb = ops.pack_conv_weight(w, False, True, g)
    dy = to_ndhwc(rnd(B, Cout, D, H, W, seed=2))
    y = to_ndhwc(rnd(B, Cin, D, H, W, seed=3))
    mean = (rnd(Cin, seed=4) * 0.1).to(DEV)
    invstd = (rnd(Cin, seed=5).abs() + 0.5).to(DEV)
    scale = rnd(Cin, seed=6).to(DEV)                   # both signs: the mask is (y*scale + shift > 0)
    shift = (rnd(Cin, seed=7) * 0.3).to(DEV)
    dx0 = ops.conv3d_bwd_data(dy, wb, g)
    dx1, part1 = ops.conv3d_bwd_data_bnstats(dy, wb, g, y, mean, invstd, scale, shift)
    assert torch.equal(dx0, dx1)
    part0 = ops.bn_bwd_reduce(dx0, None, y, mean, invstd, True, scale, shift)
    s0 = ops.reduce_partials(part0).reshape(2, Cin)
    s1 = ops.reduce_partials(part1).reshape(2, Cin)
    # reference in double from the same dx
    gm = torch.where(torch.addcmul(shift, y, scale) > 0, dx0, torch.zeros_like(dx0)).double()
    ref = torch.stack([gm.sum((0, 1, 2, 3)), (gm * ((y.double() - mean.double()) * invstd.double())).sum((0, 1, 2, 3))])
    denom = ref.abs().max()
    assert ((s1 - ref).abs().max() / denom).item() < 1e-5, ((s1 - ref).abs().max() / denom).item()
    assert ((s0 - ref).abs().max() / denom).item() < 1e-5
    monkeypatch.setenv("DRAM_BWD_BNSTATS", "0")
    assert not ops.conv_bwd_bnstats_ok(g)


PERSIST_CASES = [
    # kind, B, D, H, W, Cin, Cout, dil: Winograd-domain GEMMs whose workgroups walk SEVERAL tiles (> 256 tiles: 216
    # points x 2 M tiles x column tiles) and 1x1x1 convolutions (one point, fused epilogues), 64- and 128-column tiles
    ("wino", 2, 16, 32, 32, 128, 128, 1),      # config 1's layer2: 432 tiles of 256 x 128
    ("wino", 1, 16, 16, 32, 256, 192, 2),      # 216 x 1 x 3 tiles of 64 columns, dilation lattice
    ("c1", 2, 16, 32, 32, 256, 512, 1),        # 1x1x1, 128 x 4 = 512 tiles of 128 columns (K = 256: 8 iterations)
    ("c1", 1, 8, 16, 16, 512, 64, 1),          # 1x1x1, 8 tiles of 64 columns: fewer tiles than CUs
]


@pytest.mark.parametrize("case", PERSIST_CASES, ids=[str(c) for c in PERSIST_CASES])
def test_persistent_nn_gemm_equals_one_tile_gemm(ops, monkeypatch, case):
    """wino_gemm_nn_pers_kernel (workgroups that walk several tiles, the next tile's first stage prefetched under the
    epilogue, accumulators turned through LDS 16 rows at a time) against the one-tile kernel: forward, data gradient
    (with the fused shortcut-gradient epilogue) and the fused BatchNorm partial sums bit for bit."""
    kind, B, D, H, W, Cin, Cout, dil = case
    k = 3 if kind == "wino" else 1
    if kind == "wino":
        monkeypatch.setenv("DRAM_CONV_ALGO", "2")
        monkeypatch.setenv("DRAM_WINO_TILING", "4,4,4")
    else:
        monkeypatch.delenv("DRAM_CONV_ALGO", raising=False)
    monkeypatch.setenv("DRAM_NN_STREAM", "0")
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, k, 1, dil if k == 3 else 0, dil if k == 3 else 1)
    assert ops.conv_algo(g) == (1 if kind == "wino" else 3)
    x = to_ndhwc(rnd(B, Cin, D, H, W, seed=1))
    gy = to_ndhwc(rnd(B, Cout, D, H, W, seed=4))
    add, gate = to_ndhwc(rnd(B, Cin, D, H, W, seed=5)), to_ndhwc(rnd(B, Cin, D, H, W, seed=6))
    w = (rnd(Cout, Cin, k, k, k, seed=2) * 0.1).to(DEV)
    bias = rnd(Cout, seed=3).to(DEV)
    wf, wb = ops.pack_conv_weight(w, True, True, g)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DRAM_NN_PERSIST", mode)
        y, st = ops.conv3d_fwd(x, wf, bias, g, True)
        res[mode] = (y, st, ops.conv3d_bwd_data(gy, wb, g), ops.conv3d_bwd_data(gy, wb, g, add, gate))
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)
    ref = F.conv3d(rnd(B, Cin, D, H, W, seed=1).double(), w.cpu().double(), bias.cpu().double(), 1, dil if k == 3 else 0,
                   dil if k == 3 else 1)
    assert rel_l2(to_ncdhw(res["1"][0]).double(), ref) < 3e-5


@pytest.mark.parametrize("case", [WINO_CASES[0], WINO_CASES[2], WINO_CASES[5], WINO_CASES[7]], ids=str)
def test_winograd_batchnorm_prologue_is_bit_identical(ops, monkeypatch, case):
    """K7(b) (med3d.py:121-124: bn, relu, conv): the F(4,3)^3 input transform applies the producing unit's
    BatchNorm-apply + ReLU on its way in (zero padding outside the volume, ragged / dilated / empty-sub-lattice
    tiles included) -- output, statistics and the cached transformed input are bit-identical to bn_apply followed
    by the plain convolution, and the weight gradient runs from that cached input without the activation tensor."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "2")
    monkeypatch.setenv("DRAM_WINO_TILING", "4,4,4")
    B, D, H, W, Cin, Cout, dil = case
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    assert ops.conv_prologue_ok(g)
    y_pre = to_ndhwc(rnd(B, Cin, D, H, W, seed=1))
    scale = (rnd(Cin, seed=2).abs() + 0.5).to(DEV)
    shift = (rnd(Cin, seed=3) * 0.3).to(DEV)
    w = (rnd(Cout, Cin, 3, 3, 3, seed=4) * 0.1).to(DEV)
    bias = rnd(Cout, seed=5).to(DEV)
    wf, _ = ops.pack_conv_weight(w, True, False, g)
    z = ops.bn_apply(y_pre, scale, shift, None, 1, True)
    y0, s0, v0 = ops.conv3d_fwd_keep(z, wf, bias, g, True, True)
    y1, s1, v1 = ops.conv3d_fwd_keep(y_pre, wf, bias, g, True, True, prologue=(scale, shift))
    assert torch.equal(y0, y1) and torch.equal(s0, s1) and torch.equal(v0, v1)
    gy = to_ndhwc(rnd(B, Cout, D, H, W, seed=6))
    assert torch.equal(ops.conv3d_bwd_weight(None, gy, g, v_cache=v1), ops.conv3d_bwd_weight(z, gy, g, v_cache=v0))
    monkeypatch.setenv("DRAM_WINO_TILING", "4,4,2")            # other tilings: no prologue (the engine materialises z)
    assert not ops.conv_prologue_ok(g)


MATH_TOL = {  # measured rel-L2 vs fp64 (tools/math_check.py): fp32 6.5e-7 / 1.1e-5, bf16x3 9.8e-6 / 1.7e-4, bf16 5.2e-3 / 9e-2
    ("bf16x3", "2,2,2"): 3e-5, ("bf16x3", "4,4,4"): 5e-4, ("bf16", "2,2,2"): 1.5e-2, ("bf16", "4,4,4"): 0.25}


@pytest.mark.parametrize("tiling", ["2,2,2", "4,4,4"], ids=["F222", "F444"])
@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", [WINO_CASES[1], WINO_CASES[4], WINO_CASES[6]], ids=str)
def test_conv3d_winograd_bf16_math_modes(ops, monkeypatch, case, math, tiling):
    """DRAM_MATH = bf16x3 / bf16: the Winograd-domain GEMMs on the bf16 matrix cores (split-bf16 operand
    images written by the transforms; NN form for forward / data gradient, transposed-LDS-read TN form for
    the weight gradient, with and without the cached transformed input) against F.conv3d in fp32."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "2")
    monkeypatch.setenv("DRAM_WINO_TILING", tiling)
    monkeypatch.setenv("DRAM_MATH", math)
    tol = MATH_TOL[(math, tiling)]
    B, D, H, W, Cin, Cout, dil = case
    x = rnd(B, Cin, D, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x, w, bias, 1, dil, dil)
    gy = rnd(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV), True, True, g)
    xd, gyd = to_ndhwc(x.detach()), to_ndhwc(gy)
    y, _, v = ops.conv3d_fwd_keep(xd, wf, bias.to(DEV), g, False, True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < tol
    assert rel_l2(to_ncdhw(ops.conv3d_bwd_data(gyd, wb, g)), gx_ref) < tol
    dw = ops.conv3d_bwd_weight(xd, gyd, g)
    assert rel_l2(dw.cpu(), gw_ref) < tol
    assert torch.equal(ops.conv3d_bwd_weight(xd, gyd, g, v_cache=v), dw)
    if math == "bf16x3":            # the split really carries ~16 mantissa bits: far below one bf16 ulp (4e-3)
        assert rel_l2(to_ncdhw(y), y_ref.detach()) > 1e-7


W2D_CASES = [
    # B, D, H, W, Cin, Cout      (DRAM_CONV_ALGO=3: fused in-plane Winograd wherever applicable)
    (1, 16, 8, 8, 64, 64),
    (2, 9, 11, 13, 32, 64),             # ragged in every axis, one 16-channel pair of chunks
    (1, 20, 16, 24, 128, 64),           # two z tiles, 8 chunks
    (1, 16, 16, 16, 64, 128),           # two N tiles forward, 128 gathered channels backward
    (1, 18, 10, 9, 64, 32),             # 32-column variant
    (1, 33, 8, 8, 96, 96),              # N = 96: 32-column tiles, 3 z tiles
]


@pytest.mark.parametrize("variant", ["1", "2", "3"], ids=["mfma32x32x2", "mfma16x16x4", "mfma32x32x2-2wg"])
@pytest.mark.parametrize("case", W2D_CASES, ids=[str(c) for c in W2D_CASES])
def test_conv3d_fused_inplane_winograd(ops, monkeypatch, case, variant):
    """conv_wino2d_kernel (halo in LDS, A fragments B^T v B formed on the fly, output transform
    folded into the accumulation) against F.conv3d and its autograd: forward with bias + fused BN
    sums, data gradient with the fused shortcut-gradient epilogue."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "3")
    monkeypatch.setenv("DRAM_W2D_V", variant)          # both kernel variants (16-deep / 8-deep tiles)
    tol = 1e-5
    B, D, H, W, Cin, Cout = case
    x = rnd(B, Cin, D, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x, w, bias, 1, 1, 1)
    gy = rnd(*y_ref.shape, seed=4)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, 1, 1)
    assert ops.conv_algo(g) == 2
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV), True, True, g)
    assert wf.shape == (48, Cout, Cin) and wb.shape == (48, Cin, Cout)
    xd, gyd = to_ndhwc(x.detach()), to_ndhwc(gy)
    y, stats = ops.conv3d_fwd(xd, wf, bias.to(DEV), g, True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < tol
    s = ops.reduce_partials(stats).cpu()
    yr = y_ref.detach().double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=2e-5, atol=2e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=2e-5, atol=2e-3)
    y2, st2 = ops.conv3d_fwd(xd, wf, None, g, False)
    assert st2 is None and rel_l2(to_ncdhw(y2), F.conv3d(x, w, None, 1, 1, 1).detach()) < 1e-5
    add = rnd(B, Cin, D, H, W, seed=5)
    gate = rnd(B, Cin, D, H, W, seed=6)
    dx = ops.conv3d_bwd_data(gyd, wb, g)
    assert rel_l2(to_ncdhw(dx), gx_ref) < tol
    dx2 = ops.conv3d_bwd_data(gyd, wb, g, to_ndhwc(add), to_ndhwc(gate))
    assert rel_l2(to_ncdhw(dx2), gx_ref + add * (gate > 0).float()) < tol
    dw = ops.conv3d_bwd_weight(xd, gyd, g)               # in-plane Winograd z-walking weight gradient
    assert rel_l2(dw.cpu(), gw_ref) < 1e-5
    assert torch.equal(dw, ops.conv3d_bwd_weight(xd, gyd, g))      # slabs summed in a fixed order


def test_winograd_linearity_at_scale(ops):
    """BASELINE-sized layer4 conv (512->512, dilation 4 @ 2x16x32x32) on the library's own plan
    (Winograd, F(4,3) on all three axes: rounding error ~1e-5): linearity, a probe of one residue sub-lattice against the CPU op, and bitwise
    reproducibility of the weight gradient (fixed-order slab sum, no atomics)."""
    B, D, H, W, C = 2, 16, 32, 32, 512
    g = ops.ConvGeom(B, D, H, W, C, C, 3, 1, 4, 4)
    assert ops.conv_use_wino(g)
    gen = torch.Generator(device=DEV).manual_seed(0)
    x1 = torch.randn(g.in_shape, device=DEV, generator=gen)
    x2 = torch.randn(g.in_shape, device=DEV, generator=gen)
    w = torch.randn(C, C, 3, 3, 3, device=DEV, generator=gen) * 0.02
    wf, _ = ops.pack_conv_weight(w, True, False, g)
    y1, _ = ops.conv3d_fwd(x1, wf, None, g, False)
    y2, _ = ops.conv3d_fwd(x2, wf, None, g, False)
    y3, _ = ops.conv3d_fwd(0.5 * x1 + x2, wf, None, g, False)
    assert rel_l2((0.5 * y1 + y2).cpu(), y3.cpu()) < 5e-5
    # residue (1, 2, 3) of batch 0 is an ordinary dilation-1 convolution of a 4x8x8 volume
    sub = x1[0, 1::4, 2::4, 3::4].permute(3, 0, 1, 2)[None].cpu()
    ref = F.conv3d(sub, w.cpu(), None, 1, 1)
    assert rel_l2(y1[0, 1::4, 2::4, 3::4].permute(3, 0, 1, 2)[None].cpu(), ref) < 5e-5
    dw1 = ops.conv3d_bwd_weight(x1, y2, g)
    dw2 = ops.conv3d_bwd_weight(x1, y2, g)
    assert torch.equal(dw1, dw2)


def test_conv_linearity_at_scale(ops, monkeypatch):
    """size-independent property at a BASELINE-sized layer (us2.1: 64->64 @ 64x128x128):
    conv(a*x1 + x2) == a*conv(x1) + conv(x2), and a checksum against a strided CPU probe."""
    monkeypatch.setenv("DRAM_CONV_ALGO", "1")
    B, D, H, W, C = 1, 64, 128, 128, 64
    g = ops.ConvGeom(B, D, H, W, C, C, 3, 1, 1, 1)
    gen = torch.Generator(device=DEV).manual_seed(0)
    x1 = torch.randn(g.in_shape, device=DEV, generator=gen)
    x2 = torch.randn(g.in_shape, device=DEV, generator=gen)
    w = torch.randn(C, C, 3, 3, 3, device=DEV, generator=gen) * 0.05
    wf, _ = ops.pack_conv_weight(w)
    y1, _ = ops.conv3d_fwd(x1, wf, None, g, False)
    y2, _ = ops.conv3d_fwd(x2, wf, None, g, False)
    y3, _ = ops.conv3d_fwd(0.5 * x1 + x2, wf, None, g, False)
    assert rel_l2((0.5 * y1 + y2).cpu(), y3.cpu()) < 1e-5
    # probe a slab against the CPU op
    sl = x1[:, 10:16].permute(0, 4, 1, 2, 3).cpu()
    ref = F.conv3d(sl, w.cpu(), None, 1, (0, 1, 1))  # valid in z: output planes 11..14
    assert rel_l2(y1[:, 11:15].permute(0, 4, 1, 2, 3).cpu(), ref) < 1e-5


@pytest.mark.parametrize("shape", [(1, 16, 16, 16), (2, 8, 24, 40), (1, 10, 14, 18)])
def test_stem(ops, shape):
    B, D, H, W = shape
    x = rnd(B, 1, D, H, W, seed=1)
    w = (rnd(64, 1, 7, 7, 7, seed=2) * 0.1).requires_grad_(True)
    y_ref = F.conv3d(x, w, None, 2, 3)
    gy = rnd(*y_ref.shape, seed=3)
    (gw_ref,) = torch.autograd.grad(y_ref, [w], gy)
    xd = x.reshape(B, D, H, W).to(DEV)
    y, stats = ops.stem_fwd(xd, w.detach().to(DEV), True)
    assert rel_l2(to_ncdhw(y), y_ref.detach()) < 2e-6
    s = ops.reduce_partials(stats).cpu()
    assert torch.allclose(s[0], y_ref.detach().double().sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], y_ref.detach().double().square().sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    dw = ops.stem_bwd_weight(xd, to_ndhwc(gy))
    assert rel_l2(dw.cpu(), gw_ref) < 5e-6


@pytest.mark.parametrize("C,res", [(32, None), (64, "id"), (128, "a2"), (256, "a1"), (2048, None)])
def test_batchnorm_relu_residual(ops, C, res):
    B, D, H, W = 2, 4, 6, 5
    y = rnd(B, C, D, H, W, seed=1) * 2 + 0.5
    gamma = (rnd(C, seed=2) * 0.2 + 1).requires_grad_(True)
    beta = (rnd(C, seed=3) * 0.2).requires_grad_(True)
    y.requires_grad_(True)
    if res == "id":
        r = rnd(B, C, D, H, W, seed=4)
        rfull, rs = r, 1
    elif res == "a2":   # stride-2 shortcut A from a wider grid with C/2 channels
        r = rnd(B, C // 2, 2 * D - 1, 2 * H, 2 * W - 1, seed=4)
        rfull, rs = orc.shortcut_a(r, C, 2), 2
    elif res == "a1":   # stride-1 channel-padded shortcut A (ResNet-50 layer1.0)
        r = rnd(B, C // 4, D, H, W, seed=4)
        rfull, rs = orc.shortcut_a(r, C, 1), 1
    else:
        r, rfull, rs = None, 0.0, 1
    out, mean, var = orc.batch_norm_explicit(y, gamma, beta)
    z_ref = F.relu(out + rfull)
    gz = rnd(B, C, D, H, W, seed=5)
    gy_ref, gg_ref, gb_ref = torch.autograd.grad(z_ref, [y, gamma, beta], gz)

    yd = to_ndhwc(y.detach())
    n = B * D * H * W
    yf = yd.reshape(-1, C)
    # statistics path: partial sums -> double reduce -> finalize (+ running stats)
    part = torch.stack([yf.sum(0), (yf * yf).sum(0)])[None].contiguous()
    sums = ops.reduce_partials(part)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean_d, invstd_d, scale, shift = ops.bn_finalize(sums, n, gamma.detach().to(DEV), beta.detach().to(DEV), rm, rv,
                                                     0.1, 1e-5, True)
    assert torch.allclose(mean_d.cpu(), mean.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(invstd_d.cpu(), torch.rsqrt(var.detach() + 1e-5), rtol=1e-5)
    assert torch.allclose(rm.cpu(), 0.1 * mean.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.cpu(), 0.9 + 0.1 * var.detach() * n / (n - 1), rtol=1e-5)
    rd = None if r is None else to_ndhwc(r)
    z = ops.bn_apply(yd, scale, shift, rd, rs, True)
    assert rel_l2(to_ncdhw(z), z_ref.detach()) < 1e-6
    # eval-mode finalize reads the running stats
    m2, is2, _, _ = ops.bn_finalize(None, 1.0, gamma.detach().to(DEV), beta.detach().to(DEV), rm, rv, 0.1, 1e-5, False)
    assert torch.allclose(m2, rm) and torch.allclose(is2.cpu(), torch.rsqrt(rv.cpu() + 1e-5), rtol=1e-6)
    # backward
    gzd = to_ndhwc(gz)
    bp = ops.bn_bwd_reduce(gzd, z, yd, mean_d, invstd_d, True)
    bs = ops.reduce_partials(bp)
    assert rel_l2(bs[1].cpu(), gg_ref) < 1e-5 and rel_l2(bs[0].cpu(), gb_ref) < 1e-5
    dy = ops.bn_bwd_apply(gzd, z, yd, mean_d, invstd_d, gamma.detach().to(DEV), bs, n, True)
    assert rel_l2(to_ncdhw(dy), gy_ref) < 2e-5
    if r is None:      # no residual: the ReLU mask re-derived from y (scale, shift) is bitwise the saved one
        assert torch.equal(ops.bn_bwd_reduce(gzd, None, yd, mean_d, invstd_d, True, scale, shift), bp)
        assert torch.equal(ops.bn_bwd_apply(gzd, None, yd, mean_d, invstd_d, gamma.detach().to(DEV), bs, n, True,
                                            scale, shift), dy)
    dy2, cp = ops.bn_bwd_apply(gzd, z, yd, mean_d, invstd_d, gamma.detach().to(DEV), bs, n, True, want_colsum=True)
    assert torch.equal(dy2, dy)
    if cp is not None:     # column sums of dy on the way (bias gradient of a convolution in front)
        ref = dy.double().reshape(-1, C).sum(0)
        assert float((ops.reduce_partials(cp)[0] - ref).abs().max()) <= 1e-5 * float(dy.abs().sum() / C) + 1e-6
    else:
        assert 256 % (C // 4) != 0
    cs = ops.reduce_partials(ops.colsum(gzd))[0]
    assert rel_l2(cs.cpu(), gz.double().sum((0, 2, 3, 4))) < 1e-5


def test_add(ops):
    a, b = rnd(1237, seed=1).to(DEV), rnd(1237, seed=2).to(DEV)
    assert torch.equal(ops.add(a, b), a + b)


@pytest.mark.parametrize("shape", [(1, 8, 8, 8, 64), (2, 6, 10, 7, 64), (1, 5, 5, 5, 8)])
def test_maxpool(ops, shape):
    B, D, H, W, C = shape
    x = F.relu(rnd(B, C, D, H, W, seed=1)).requires_grad_(True)   # post-ReLU: ties at 0 like the stem
    y_ref = F.max_pool3d(x, 3, 2, 1)
    gy = rnd(*y_ref.shape, seed=2)
    (gx_ref,) = torch.autograd.grad(y_ref, [x], gy)
    xd = to_ndhwc(x.detach())
    y, am = ops.maxpool_fwd(xd)
    assert torch.equal(to_ncdhw(y), y_ref.detach())
    addt = rnd(B, C, D, H, W, seed=3)
    dx = ops.maxpool_bwd(to_ndhwc(gy), am, tuple(xd.shape), to_ndhwc(addt))
    assert rel_l2(to_ncdhw(dx), gx_ref + addt) < 1e-6
    if C % 4 == 0:      # the add operand as a channel slice of a wider tensor, read in place
        wide = torch.cat([torch.zeros_like(to_ndhwc(addt))[..., :4], to_ndhwc(addt)], dim=-1).contiguous()
        assert torch.equal(ops.maxpool_bwd(to_ndhwc(gy), am, tuple(xd.shape), wide[..., 4:]), dx)


@pytest.mark.parametrize("shape", [(1, 8, 8, 8, 64), (2, 6, 10, 7, 64), (1, 5, 7, 9, 8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stem_bn_relu_maxpool_one_pass(ops, shape, dtype):
    """K9 (med3d.py:272-275): BatchNorm-apply + ReLU + max-pool of the stem in ONE pass over the pre-BN tensor is
    bit-identical to bn_apply followed by maxpool_fwd -- z, pooled values and taps; odd extents included."""
    B, D, H, W, C = shape
    y = rnd(B, D, H, W, C, seed=1).to(DEV).to(dtype)
    scale = (rnd(C, seed=2).abs() + 0.5).to(DEV)
    shift = rnd(C, seed=3).to(DEV) * 0.3
    z_ref = ops.bn_apply(y, scale, shift, None, 1, True)
    p_ref, a_ref = ops.maxpool_fwd(z_ref)
    z, p, a = ops.bn_maxpool_fwd(y, scale, shift)
    assert torch.equal(z, z_ref) and torch.equal(p, p_ref) and torch.equal(a, a_ref)


@pytest.mark.parametrize("case", [(1, 3, 4, 5, 8, 6, 8, 10, 4), (2, 2, 4, 4, 16, 5, 9, 8, 8), (1, 4, 4, 4, 64, 8, 8, 8, 64)])
def test_upcat(ops, case):
    B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck = case
    src = rnd(B, Cu, Ds, Hs, Ws, seed=1).requires_grad_(True)
    skip = rnd(B, Ck, Dk, Hk, Wk, seed=2).requires_grad_(True)
    cat_ref = orc.crop_concat(orc.upsample2_trilinear(src), skip)
    g = rnd(*cat_ref.shape, seed=3)
    gs_ref, gk_ref = torch.autograd.grad(cat_ref, [src, skip], g)
    cat = ops.upcat_fwd(to_ndhwc(src.detach()), to_ndhwc(skip.detach()))
    assert rel_l2(to_ncdhw(cat), cat_ref.detach()) < 1e-6
    dsrc, dskip = ops.upcat_bwd(to_ndhwc(g), (B, Ds, Hs, Ws, Cu), (B, Dk, Hk, Wk, Ck))
    assert rel_l2(to_ncdhw(dsrc), gs_ref) < 1e-5
    assert torch.equal(to_ncdhw(dskip), gk_ref)


def test_upcat_tiled_forward(ops, monkeypatch):
    """>= 512 output blocks and Cu % 64 == 0: the LDS-tiled forward kernel (ragged extents, cropped skip, two
    64-channel blocks) against the oracle and against the untiled kernel (same corner order; fma contraction may
    differ between the two kernels, the copied skip channels are identical)."""
    B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck = 1, 33, 30, 35, 128, 67, 62, 70, 64
    src, skip = rnd(B, Cu, Ds, Hs, Ws, seed=1), rnd(B, Ck, Dk, Hk, Wk, seed=2)
    cat_ref = orc.crop_concat(orc.upsample2_trilinear(src), skip)
    cat = ops.upcat_fwd(to_ndhwc(src), to_ndhwc(skip))
    assert rel_l2(to_ncdhw(cat), cat_ref) < 1e-6
    monkeypatch.setenv("DRAM_UPCAT_UNTILED", "1")
    cat2 = ops.upcat_fwd(to_ndhwc(src), to_ndhwc(skip))
    assert rel_l2(cat2.cpu(), cat.cpu()) < 2e-6 and torch.equal(cat2[..., Cu:], cat[..., Cu:])


UPMIX_CASES = [
    # B, Ds, Hs, Ws, Cu, Cs, Co      (us1.0: conv3x3x3(concat(up(src), skip)) -- med3d.py:83-87 / :67)
    (1, 2, 4, 4, 512, 64, 64),        # the golden fixtures' geometry (16x32x32 input): M = 32, direct 1x1x1 kernel
    (2, 4, 4, 8, 128, 64, 64),        # M = 256: the batched-GEMM 1x1x1 kernels
    (1, 3, 5, 2, 64, 32, 32),         # odd / tiny low-resolution extents (scale 2/5, 4/9, 1/3), 32 output channels
    (1, 1, 2, 3, 64, 32, 64),         # Ds = 1: scale 0 along z
]


@pytest.mark.parametrize("case", UPMIX_CASES, ids=[str(c) for c in UPMIX_CASES])
def test_upmix_lowres_mixing_equals_conv_of_upsampled_concat(ops, case):
    """The first decoder convolution without the up-sampled tensor (csrc/upmix.hip): low-resolution GEMM + separable
    tap/trilinear gather + skip convolution against the fp64 convolution of concat(upsample(src), skip) -- forward
    (rel-L2 <= 1e-5), BatchNorm partial sums, both data gradients and the merged weight gradient (<= 1e-5), and the
    gather passes as exact transposes of each other."""
    B, Ds, Hs, Ws, Cu, Cs, Co = case
    src = rnd(B, Cu, Ds, Hs, Ws, seed=1).double().requires_grad_(True)
    skip = rnd(B, Cs, 2 * Ds, 2 * Hs, 2 * Ws, seed=2).double().requires_grad_(True)
    w = (rnd(Co, Cu + Cs, 3, 3, 3, seed=3) * 0.05).double().requires_grad_(True)
    bias = rnd(Co, seed=4).double()
    up = F.interpolate(src, scale_factor=2, mode="trilinear", align_corners=True)
    y_ref = F.conv3d(torch.cat([up, skip], 1), w, bias, 1, 1)
    gy = rnd(*y_ref.shape, seed=5).double()
    gsrc_ref, gskip_ref, gw_ref = torch.autograd.grad(y_ref, [src, skip, w], gy)

    wd = w.detach().float().to(DEV)
    wlo, ws = ops.upmix_split_weight(wd, Cu)
    assert torch.equal(wlo.reshape(27, Co, Cu).cpu(), w.detach().float()[:, :Cu].reshape(Co, Cu, 27).permute(2, 0, 1))
    assert torch.equal(ws.cpu(), w.detach().float()[:, Cu:])
    g_lo = ops.ConvGeom(B, Ds, Hs, Ws, Cu, 27 * Co, 1, 1, 0, 1)
    g_s = ops.ConvGeom(B, 2 * Ds, 2 * Hs, 2 * Ws, Cs, Co, 3, 1, 1, 1)
    wf_lo, wb_lo = ops.pack_conv_weight(wlo, True, True, g_lo)
    wf_s, wb_s = ops.pack_conv_weight(ws, True, True, g_s)
    a = to_ndhwc(src.detach().float())
    sk = to_ndhwc(skip.detach().float())
    b, _ = ops.conv3d_fwd(a, wf_lo, None, g_lo, False)
    ysk, _ = ops.conv3d_fwd(sk, wf_s, bias.float().to(DEV), g_s, False)
    y, stats = ops.upmix_gather_fwd(b, ysk, Co, True)
    assert y.data_ptr() == ysk.data_ptr()                      # in place
    assert rel_l2(to_ncdhw(y).double(), y_ref.detach()) < 1e-5
    s = ops.reduce_partials(stats).cpu()
    yd = to_ncdhw(y).double()
    assert torch.allclose(s[0], yd.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yd * yd).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)

    gyd = to_ndhwc(gy.float())
    h = ops.upmix_gather_bwd(gyd)
    # exact transposes: <gather(b), gy> == <b, gather^T(gy)>
    y0, _ = ops.upmix_gather_fwd(b, None, Co, False)
    lhs = float((y0.double() * gyd.double()).sum())
    rhs = float((b.double() * h.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    dsrc = ops.conv3d_bwd_data(h, wb_lo, g_lo)
    dskip = ops.conv3d_bwd_data(gyd, wb_s, g_s)
    assert rel_l2(to_ncdhw(dsrc).double(), gsrc_ref) < 1e-5
    assert rel_l2(to_ncdhw(dskip).double(), gskip_ref) < 1e-5
    if Cu % 64 == 0 and Cs % 64 == 0:
        dw = ops.upmix_merge_wgrad(ops.conv3d_bwd_weight(a, h, g_lo), ops.conv3d_bwd_weight(sk, gyd, g_s))
        assert rel_l2(dw.cpu().double(), gw_ref) < 1e-5


@pytest.mark.parametrize("mode", ["cls", "reg", "reg_nolungs"])
def test_head(ops, mode):
    B, D, H, W = 2, 4, 6, 5
    x = rnd(B, 32, D, H, W, seed=1).requires_grad_(True)
    n = (6, 3) if mode == "cls" else (1, 1)
    NO = sum(n)
    w = (rnd(NO, 32, 1, 1, 1, seed=2) * 0.3).requires_grad_(True)
    b = (rnd(NO, seed=3) * 0.1).requires_grad_(True)
    lungs = (torch.rand(B, 1, 2 * D, 2 * H, 2 * W, generator=torch.Generator().manual_seed(4)) > 0.4).float()
    pre = F.conv3d(x, w, b)
    if mode == "cls":
        dense_ref = pre
        outs_ref = dense_ref.mean((2, 3, 4))
    else:
        dense_ref = torch.sigmoid(pre)
        lg = torch.ones(B, 1, D, H, W) if mode == "reg_nolungs" else F.interpolate(lungs, (D, H, W), mode="nearest")
        outs_ref = (dense_ref * lg).flatten(2).sum(-1) / lg.flatten(2).sum(-1)
    go = rnd(B, NO, seed=5)
    gd = rnd(B, NO, D, H, W, seed=6) * 0.01
    loss = (outs_ref * go).sum() + (dense_ref * gd).sum()
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(loss, [x, w, b])

    sig = mode != "cls"
    lun = lungs.reshape(B, 2 * D, 2 * H, 2 * W).to(DEV) if mode == "reg" else None
    xd = to_ndhwc(x.detach())
    wd = w.detach().reshape(NO, 32).to(DEV)
    dense, partial = ops.head_fwd(xd, wd, b.detach().to(DEV), lun, sig)
    assert rel_l2(dense.cpu(), dense_ref.detach()) < 1e-6
    sums = partial.sum(1)
    outs = sums[:, :NO] / sums[:, NO:]
    assert torch.allclose(outs.cpu(), outs_ref.detach(), rtol=1e-5, atol=1e-6)
    gpool = (go.to(DEV) / sums[:, NO:]).contiguous()
    dx, wpart = ops.head_bwd(xd, wd, dense if sig else None, gd.to(DEV), gpool, lun, sig)
    assert rel_l2(to_ncdhw(dx), gx_ref) < 1e-5
    wg = ops.reduce_partials(wpart.reshape(wpart.shape[0], 1, NO * 33)).reshape(NO, 33).cpu()
    assert rel_l2(wg[:, :32], gw_ref.reshape(NO, 32)) < 1e-5
    assert rel_l2(wg[:, 32], gb_ref) < 1e-5


@pytest.mark.parametrize("B,binary", [(2, [1.0, 0.0]), (3, [1.0, 1.0, 1.0])])
def test_segloss(ops, B, binary):
    D, H, W = 4, 8, 6
    gen = torch.Generator().manual_seed(7)
    cle = torch.rand(B, 1, D, H, W, generator=gen).requires_grad_(True)
    pse = torch.rand(B, 1, D, H, W, generator=gen).requires_grad_(True)
    lungs = (torch.rand(B, 1, 2 * D, 2 * H, 2 * W, generator=gen) > 0.4).float()
    ems = (torch.rand(B, 1, 2 * D, 2 * H, 2 * W, generator=gen) > 0.7).float() * lungs
    bt = torch.tensor(binary)
    seg_labels = F.interpolate(ems * bt.view(B, 1, 1, 1, 1), (D, H, W), mode="nearest")
    lung_labels = F.interpolate(lungs, (D, H, W), mode="nearest")
    mul_ref, seg_ref = orc.segmentation_loss(cle, pse, seg_labels, lung_labels)
    (2.0 * mul_ref + seg_ref).backward()

    c4, p4 = cle.detach().reshape(B, D, H, W).to(DEV), pse.detach().reshape(B, D, H, W).to(DEV)
    l4, e4 = lungs.reshape(B, 2 * D, 2 * H, 2 * W).to(DEV), ems.reshape(B, 2 * D, 2 * H, 2 * W).to(DEV)
    part = ops.segloss_fwd(c4, p4, l4, e4, bt.to(DEV))
    s = part.double().sum(0).cpu()
    N = B * D * H * W
    st, A1, A0, I, S1, S2 = [float(v) for v in s]
    alpha = min(max(1.0 - st / B, 0.3), 0.7)
    sw = alpha * st + (1 - alpha) * (N - st)
    seg = (alpha * A1 + (1 - alpha) * A0) / sw
    den = S1 + S2 + 1e-7
    mul = (2 * I + 1e-7) / den
    assert abs(mul - float(mul_ref)) < 1e-5 and abs(seg - float(seg_ref)) < 1e-5 * max(1, abs(float(seg_ref)))
    coef = torch.tensor([2.0 * 2 / den, 2.0 * (2 * I + 1e-7) / den ** 2, alpha / sw, (1 - alpha) / sw, 0, 0, 0, 0],
                        dtype=torch.float32, device=DEV)
    gc, gp = ops.segloss_bwd(c4, p4, l4, e4, bt.to(DEV), coef)
    assert rel_l2(gc.cpu().reshape(cle.shape), cle.grad) < 1e-5
    assert rel_l2(gp.cpu().reshape(pse.shape), pse.grad) < 1e-5


def test_regression_train_loss_tail_in_one_launch(ops):
    """models.reg_train_loss (seg-loss pass + ONE tail kernel) against the oracle's restatement of models.py:549-574:
    every component, the total, d loss / d dense maps and d loss / d scores (fp64 autograd of the oracle), with
    labels that cover the empty band of class 0; a label outside the band table must poison the loss."""
    from bodyct_dram_emph_subtype_amd import models
    B, D, H, W = 5, 4, 8, 6
    gen = torch.Generator().manual_seed(11)
    cle = torch.rand(B, 1, D, H, W, generator=gen) * 0.6
    pse = torch.rand(B, 1, D, H, W, generator=gen) * 0.6
    reg = [torch.rand(B, generator=gen) * 0.8 + 0.05 for _ in range(2)]
    lungs = (torch.rand(B, 1, 2 * D, 2 * H, 2 * W, generator=gen) > 0.4).float()
    ems = (torch.rand(B, 1, 2 * D, 2 * H, 2 * W, generator=gen) > 0.7).float() * lungs
    cl, pl = torch.tensor([0, 5, 2, 0, 3]), torch.tensor([0, 2, 1, 1, 0])
    cw, pw = torch.rand(B, generator=gen) + 0.1, torch.rand(B, generator=gen) + 0.1
    leaf = [t.double().requires_grad_(True) for t in (cle, pse, reg[0], reg[1])]
    l_ref, parts_ref = orc.reg_train_loss(leaf[:2], leaf[2:], lungs.double(), ems.double(), cl, pl, cw.double(), pw.double())
    l_ref.backward()

    dev = [t.to(DEV).requires_grad_(True) for t in (cle, pse, reg[0], reg[1])]
    loss, parts = models.reg_train_loss(dev[:2], dev[2:], lungs.to(DEV), ems.to(DEV), cl.to(DEV), pl.to(DEV),
                                        cw.to(DEV), pw.to(DEV))
    assert abs(float(loss) - float(l_ref)) < 1e-5 * max(1.0, abs(float(l_ref)))
    for k, v in parts_ref.items():
        assert abs(float(parts[k]) - float(v)) < 1e-5 * max(1.0, abs(float(v))), k
        assert not parts[k].requires_grad
    (3.0 * loss).backward()                                   # an upstream factor reaches every gradient
    ok = ((cle + pse).double() - 1.0).abs() > 1e-3            # off the clamp kink of models.py:527
    for i in (0, 1):
        assert rel_l2(dev[i].grad.cpu()[ok], 3.0 * leaf[i].grad[ok]) < 1e-5, i
    for i in (2, 3):
        assert rel_l2(dev[i].grad.cpu(), 3.0 * leaf[i].grad) < 1e-5, i
    bad, _ = models.reg_train_loss(dev[:2], dev[2:], lungs.to(DEV), ems.to(DEV), torch.full((B,), 6).to(DEV),
                                   pl.to(DEV), cw.to(DEV), pw.to(DEV))
    assert torch.isnan(bad).item()


def test_upproject(ops):
    B, D, H, W = 2, 4, 6, 5
    dense = torch.rand(B, 1, D, H, W, generator=torch.Generator().manual_seed(1))
    size = (8, 12, 10)
    ess = (torch.rand(B, 1, *size, generator=torch.Generator().manual_seed(2)) > 0.5).float()
    lungs = (torch.rand(B, 1, *size, generator=torch.Generator().manual_seed(3)) > 0.3).float()
    up_ref, pct_ref = orc.predict_upproject(dense, size, ess, lungs)
    out, partial = ops.upproject(dense.reshape(B, D, H, W).to(DEV), ess.reshape(B, *size).to(DEV), size)
    assert rel_l2(out.cpu().reshape(up_ref.shape), up_ref) < 1e-6
    pct = partial.sum(1).cpu() / lungs.sum()
    assert torch.allclose(pct, pct_ref, rtol=1e-5)
