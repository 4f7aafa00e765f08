"""GPU: the bf16-STORAGE path (reference: Lightning `--precision bf16`, train.py:46 -> torch.autocast(bfloat16);
BASELINE configs[2] / [4]) through the C ABI.

Contract of the path (include/dram_hip.h, "bf16 STORAGE path"): activation tensors are bf16 in HBM, every product
is a product of bf16 operands accumulated in fp32, element-wise math is fp32, ONE rounding (nearest even) per store;
statistics, parameters, weight gradients and the dense head outputs are fp32.

Tolerances, stated up front (SURVEY.md section 7 prescribes stating them; bf16 has 8 significant bits, 2^-9 = 2e-3
per rounding):
  * convolution kernels vs the fp64 convolution of the SAME bf16 operands: output relative L2 <= 2e-3 (its one output
    rounding: 2^-9 / sqrt(3) = 1.1e-3 expected); weight gradient (fp32 result) <= 2e-5.
  * element-wise kernels: BIT-IDENTICAL to the fp32 kernel of the same op run on the up-cast operands and rounded
    once -- same arithmetic, same order, only the storage type differs.
  * whole network vs the fp32 oracle: pooled scores / logits max-relative <= 1e-3 (ResNet-18; 5e-3 for the
    ResNet-50 fixture, whose few-hundred-voxel stride-8 stages make every BatchNorm statistic a noisy estimate), dense
    maps relative L2 <= 2e-2 at the fixtures (or 1.2 x the reference autocast's own distance where that is larger) and
    <= 4e-2 at 2x128x256x256; gradients (fp32 tensors) vs the fp64 oracle pinned to the bf16 forward's own ReLU /
    max-pool decisions relative L2 <= 1e-1 per tensor (measured 2-8e-2: every activation gradient is rounded to 8
    bits once per layer and BatchNorm's backward subtracts two nearly equal sums of them).  For scale, the
    reference's OWN arithmetic under CPU autocast(bfloat16) sits 4e-4 ... 5e-3 (pooled), 1.2-1.9e-2 (dense) and
    10-33 % (gradients, decisions free) from its fp32 self on the same fixtures (printed by the test); on the dense
    maps the HIP path must be no further from fp32 than 1.5 x that.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import med3d_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from bodyct_dram_emph_subtype_amd import ops as o
    import bodyct_dram_emph_subtype_amd as pkg
    pkg.load_library()
    return o


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def r16(t):
    """values representable in bf16, kept as float32"""
    return t.to(BF).float()


def nd(t):   # NCDHW cpu fp32 (bf16-representable) -> NDHWC gpu bf16
    return t.permute(0, 2, 3, 4, 1).contiguous().to(DEV).to(BF)


def nc(t):   # NDHWC gpu -> NCDHW cpu fp32
    return t.float().permute(0, 4, 1, 2, 3).contiguous().cpu()


CASES = [
    # B, D, H, W, Cin, Cout, dil
    (1, 8, 8, 8, 64, 64, 1),
    (2, 7, 10, 9, 64, 128, 1),          # ragged extents
    (1, 9, 12, 10, 128, 64, 2),         # dilation lattice, ragged residues
    (1, 8, 16, 16, 256, 256, 4),        # layer3/4-like
    (1, 6, 9, 11, 64, 32, 1),           # us3: 32 output columns (NB = 1); its data gradient has ONE 32-channel chunk
    (1, 4, 8, 8, 576, 64, 1),           # us1.0: 18 chunks
    (2, 4, 5, 6, 32, 96, 1),            # channel counts that are only multiples of 32
    (1, 16, 16, 24, 64, 64, 1),         # 16 planes: the 8-wave (512-voxel) tiles, a long z walk
    (2, 19, 9, 8, 128, 64, 2),          # ragged depth on a dilation lattice, two ci blocks
]
# kernel variants: forward / data-gradient tile depth (4 or 8 waves), weight gradient z-walking or tiled
VARIANTS = [("", ""), ("4", "tile"), ("8", "zwalk")]


def test_bf16_weight_packing_tiles_and_work_list(ops):
    """The LDS-turned packing (csrc/conv_bf16.hip pack_tile_bf16): full and ragged 32 x 32 tiles at 27 taps, the 864-wide
    tiles of the 1x1x1 weights, an odd tap count; wf / wb are the round-to-nearest-even bf16 images of the permuted /
    tap-flipped fp32 weight, and the one-launch work list writes exactly what the per-weight launches write."""
    from bodyct_dram_emph_subtype_amd import _lib
    shapes = [(64, 64, 3), (32, 96, 3), (40, 36, 3), (128, 1728, 1), (70, 33, 1), (8, 8, 2)]
    ws = [(rnd(co, ci, k, k, k, seed=10 + i) * 0.3).to(DEV) for i, (co, ci, k) in enumerate(shapes)]
    L = _lib.load()
    singles = []
    for w, (co, ci, k) in zip(ws, shapes):
        taps = k ** 3
        wf = torch.empty((taps, co, ci), device=DEV, dtype=BF)
        wb = torch.empty((taps, ci, co), device=DEV, dtype=BF)
        rc = L.dram_pack_conv_weight_bf16(w.data_ptr(), wf.data_ptr(), wb.data_ptr(), co, ci, taps,
                                          torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = w.reshape(co, ci, taps).to(BF)                      # RNE
        assert torch.equal(wf, ref.permute(2, 0, 1)), (co, ci, k)
        assert torch.equal(wb, ref.flip(2).permute(2, 1, 0)), (co, ci, k)
        singles.append((wf, wb))
    multi = ops.pack_conv_weights_bf16_multi(ws)
    for (wf, wb), (mf, mb) in zip(singles, multi):
        assert torch.equal(wf, mf) and torch.equal(wb, mb)
    assert L.dram_pack_conv_weight_bf16_tiles(4, 4, 1000) < 0      # more taps than a tile holds: refused


@pytest.mark.parametrize("nw,wg", VARIANTS, ids=["plan", "nw4-tile", "nw8-zwalk"])
@pytest.mark.parametrize("case", CASES, ids=str)
def test_conv3_bf16_fwd_dgrad_wgrad(ops, monkeypatch, case, nw, wg):
    if nw:
        monkeypatch.setenv("DRAM_BF16_NW", nw)
    if wg:
        monkeypatch.setenv("DRAM_BF16_WGRAD", wg)
    B, D, H, W, Cin, Cout, dil = case
    x = r16(rnd(B, Cin, D, H, W, seed=1)).requires_grad_(True)
    w32 = rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1
    w = r16(w32).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x.double(), w.double(), bias.double(), 1, dil, dil)
    gy = r16(rnd(*y_ref.shape, seed=4))
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy.double())
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
    assert ops.conv_plan(g).bf16
    wf, wb = ops.pack_conv_weight(w32.to(DEV), True, True, g, BF)
    assert wf.dtype == BF and tuple(wf.shape) == (27, Cout, Cin) and tuple(wb.shape) == (27, Cin, Cout)
    assert torch.equal(wf.float().cpu(), w.detach().permute(2, 3, 4, 0, 1).reshape(27, Cout, Cin))     # RNE packing
    y, stats, _ = ops.conv3d_fwd_keep(nd(x.detach()), wf, bias.to(DEV), g, True, False)
    assert y.dtype == BF
    assert rel_l2(nc(y), y_ref.detach()) < 2e-3
    # the BatchNorm sums are those of the values the next pass reads: the rounded outputs
    s = ops.reduce_partials(stats).cpu()
    yr = nc(y).double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    dx = ops.conv3d_bwd_data(nd(gy), wb, g)
    assert dx.dtype == BF and rel_l2(nc(dx), gx_ref) < 2e-3
    add = r16(rnd(B, Cin, D, H, W, seed=5))
    gate = r16(rnd(B, Cin, D, H, W, seed=6))
    dx2 = ops.conv3d_bwd_data(nd(gy), wb, g, nd(add), nd(gate))
    assert rel_l2(nc(dx2), gx_ref + (add * (gate > 0).float()).double()) < 2e-3
    dw = ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g)
    assert dw.dtype == torch.float32 and rel_l2(dw.cpu(), gw_ref) < 2e-5
    # deterministic (fixed-order slab reduce, no atomics)
    assert torch.equal(dw, ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g))


@pytest.mark.parametrize("case", [(2, 4, 6, 5, 128, 64), (1, 8, 8, 8, 64, 256), (1, 5, 6, 7, 256, 128), (1, 16, 16, 16, 512, 64),
                                  (1, 8, 8, 9, 192, 320),      # 128 x 128 weight-gradient blocks, ragged in both channel counts
                                  (2, 8, 16, 16, 256, 384)], ids=str)
def test_conv1x1_bf16_gemm(ops, case):
    """the 1x1x1 convolutions of the Bottleneck blocks as bf16 GEMMs over the flat voxel index (gemm1_bf16_kernel,
    wgrad1_bf16_kernel): ragged row tiles, 64- and 128-column tiles, statistics, the shortcut-gradient epilogue."""
    B, D, H, W, Cin, Cout = case
    x = r16(rnd(B, Cin, D, H, W, seed=1)).requires_grad_(True)
    w32 = rnd(Cout, Cin, 1, 1, 1, seed=2) * 0.1
    w = r16(w32).requires_grad_(True)
    y_ref = F.conv3d(x.double(), w.double())
    gy = r16(rnd(*y_ref.shape, seed=4))
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy.double())
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 1, 1, 0, 1)
    assert ops.conv_plan(g).bf16
    wf, wb = ops.pack_conv_weight(w32.to(DEV), True, True, g, BF)
    assert wf.dtype == BF and tuple(wf.shape) == (1, Cout, Cin) and tuple(wb.shape) == (1, Cin, Cout)
    y, stats, _ = ops.conv3d_fwd_keep(nd(x.detach()), wf, None, g, True, False)
    assert y.dtype == BF and rel_l2(nc(y), y_ref.detach()) < 2e-3
    s = ops.reduce_partials(stats).cpu()
    yr = nc(y).double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    dx = ops.conv3d_bwd_data(nd(gy), wb, g)
    assert dx.dtype == BF and rel_l2(nc(dx), gx_ref) < 2e-3
    add, gate = r16(rnd(B, Cin, D, H, W, seed=5)), r16(rnd(B, Cin, D, H, W, seed=6))
    dx2 = ops.conv3d_bwd_data(nd(gy), wb, g, nd(add), nd(gate))
    assert rel_l2(nc(dx2), gx_ref + (add * (gate > 0).float()).double()) < 2e-3
    dw = ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g)
    assert dw.dtype == torch.float32 and rel_l2(dw.cpu(), gw_ref) < 2e-5
    assert torch.equal(dw, ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g))


@pytest.mark.parametrize("case", [(1, 8, 12, 10, 64, 128), (2, 6, 8, 8, 128, 128), (1, 4, 8, 6, 32, 64)], ids=str)
def test_conv_bf16_stride2_as_space_to_depth(ops, case):
    """The stride-2 3x3x3 convolution (one per network) on the bf16 kernels: stride 1 over the space-to-depth tensor
    with the embedded weights (dram_s2d_bf16 / dram_s2_embed_weight / dram_d2s_bf16 / dram_s2_extract_wgrad).  Same
    bars as the stride-1 kernels, against the fp64 convolution of the same bf16 operands."""
    B, D, H, W, Cin, Cout = case
    x = r16(rnd(B, Cin, D, H, W, seed=1)).requires_grad_(True)
    w32 = rnd(Cout, Cin, 3, 3, 3, seed=2) * 0.1
    w = r16(w32).requires_grad_(True)
    bias = rnd(Cout, seed=3)
    y_ref = F.conv3d(x.double(), w.double(), bias.double(), 2, 1, 1)
    gy = r16(rnd(*y_ref.shape, seed=4))
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy.double())
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 2, 1, 1)
    g8 = ops.s2_geom(g)
    assert not ops.conv_plan(g).bf16 and g8 is not None and g8.Cin == 8 * Cin
    wf, wb = ops.pack_conv_weight(w32.to(DEV), True, True, g, BF)
    assert wf.dtype == BF and tuple(wf.shape) == (27, Cout, 8 * Cin) and tuple(wb.shape) == (27, 8 * Cin, Cout)
    # 27 of the 216 (parity, offset) slots of a (co, ci) pair carry a tap
    assert int((wf.float() != 0).sum()) <= 27 * Cout * Cin
    y, stats, _ = ops.conv3d_fwd_keep(nd(x.detach()), wf, bias.to(DEV), g, True, False)
    assert y.dtype == BF and tuple(y.shape) == tuple(g.out_shape) and rel_l2(nc(y), y_ref.detach()) < 2e-3
    s = ops.reduce_partials(stats).cpu()
    yr = nc(y).double()
    assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
    dx = ops.conv3d_bwd_data(nd(gy), wb, g)
    assert dx.dtype == BF and tuple(dx.shape) == tuple(g.in_shape) and rel_l2(nc(dx), gx_ref) < 2e-3
    add, gate = r16(rnd(B, Cin, D, H, W, seed=5)), r16(rnd(B, Cin, D, H, W, seed=6))
    # (the shortcut-gradient epilogue is applied by the depth-to-space pass on the ROUNDED gradient: two roundings)
    dx2 = ops.conv3d_bwd_data(nd(gy), wb, g, nd(add), nd(gate))
    assert rel_l2(nc(dx2), gx_ref + (add * (gate > 0).float()).double()) < 3e-3
    dx3 = ops.conv3d_bwd_data(nd(gy), wb, g, nd(add), None)
    assert rel_l2(nc(dx3), gx_ref + add.double()) < 3e-3
    dw = ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin, 3, 3, 3) and rel_l2(dw.cpu(), gw_ref) < 2e-5
    assert torch.equal(dw, ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g))


@pytest.mark.parametrize("case", [(1, 8, 12, 10, 64, 128, 3, 2, 1, "0"), (2, 6, 8, 8, 128, 128, 3, 2, 1, "0"),
                                  (1, 7, 8, 9, 64, 64, 3, 2, 1, "1")], ids=str)
def test_conv_bf16_fallback_geometries(ops, case, monkeypatch):
    """Outside the bf16 kernels (stride 2 with the space-to-depth form switched off, or with odd extents) -> fp32
    kernels around cast passes, bf16 in / bf16 out."""
    B, D, H, W, Cin, Cout, k, stride, dil, s2 = case
    monkeypatch.setenv("DRAM_BF16_S2", s2)
    pad = dil * (k - 1) // 2
    x = r16(rnd(B, Cin, D, H, W, seed=1)).requires_grad_(True)
    w = (rnd(Cout, Cin, k, k, k, seed=2) * 0.1).requires_grad_(True)
    y_ref = F.conv3d(x, w, None, stride, pad, dil)
    gy = r16(rnd(*y_ref.shape, seed=4))
    gx_ref, gw_ref = torch.autograd.grad(y_ref, [x, w], gy)
    g = ops.ConvGeom(B, D, H, W, Cin, Cout, k, stride, pad, dil)
    assert not ops.conv_plan(g).bf16 and ops.s2_geom(g) is None
    wf, wb = ops.pack_conv_weight(w.detach().to(DEV), True, True, g, BF)
    assert wf.dtype == torch.float32
    y, stats, _ = ops.conv3d_fwd_keep(nd(x.detach()), wf, None, g, True, False)
    assert y.dtype == BF and rel_l2(nc(y), y_ref.detach()) < 2e-3 and stats is not None
    dx = ops.conv3d_bwd_data(nd(gy), wb, g)
    assert dx.dtype == BF and rel_l2(nc(dx), gx_ref) < 2e-3
    dw = ops.conv3d_bwd_weight(nd(x.detach()), nd(gy), g)
    assert rel_l2(dw.cpu(), gw_ref) < 2e-5


def test_stem_bf16_matrix_core_kernels(ops):
    """stem_bf16.hip: the 7x7x7 stride-2 stem on the bf16 matrix cores (input and weights rounded to bf16, fp32
    accumulation) against the fp64 convolution of the same rounded operands: forward (bf16 store: 2e-3) with its
    BatchNorm sums, weight gradient (fp32: 2e-5), ragged extents."""
    for B, D, H, W in ((1, 16, 24, 24), (2, 9, 20, 35)):
        x = r16(rnd(B, 1, D, H, W, seed=1))
        w32 = rnd(64, 1, 7, 7, 7, seed=2) * 0.05
        w = r16(w32).requires_grad_(True)
        y_ref = F.conv3d(x.double(), w.double(), None, 2, 3)
        gy = r16(rnd(*y_ref.shape, seed=3))
        (gw_ref,) = torch.autograd.grad(y_ref, [w], gy.double())
        xd = x[:, 0].contiguous().to(DEV)
        y, stats = ops.stem_fwd(xd, w32.to(DEV), True, BF)
        assert y.dtype == BF and rel_l2(nc(y), y_ref.detach()) < 2e-3
        s = ops.reduce_partials(stats).cpu()
        yr = nc(y).double()
        assert torch.allclose(s[0], yr.sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
        assert torch.allclose(s[1], (yr * yr).sum((0, 2, 3, 4)), rtol=1e-5, atol=1e-3)
        dw = ops.stem_bwd_weight(xd, nd(gy))
        assert rel_l2(dw.cpu(), gw_ref) < 2e-5
        assert torch.equal(dw, ops.stem_bwd_weight(xd, nd(gy)))


def test_elementwise_bf16_kernels_equal_fp32_kernels_on_upcast_operands(ops, monkeypatch):
    monkeypatch.setenv("DRAM_STEM_BF16", "0")      # the fp32-MFMA stem with a bf16 store (the bf16-MFMA stem: test above)
    """Every element-wise kernel of the bf16 path is the SAME template as its fp32 form with a different storage
    type: run on bf16 tensors it must give, bit for bit, the fp32 kernel's result on the up-cast tensors rounded
    once to bf16 (fp32 outputs -- partial sums, dense maps -- must be bit-identical outright)."""
    B, D, H, W, C = 2, 6, 8, 10, 64
    y = rnd(B, D, H, W, C, seed=1).to(DEV).to(BF)
    res = rnd(B, D, H, W, C, seed=2).to(DEV).to(BF)
    dz = rnd(B, D, H, W, C, seed=3).to(DEV).to(BF)
    scale, shift = (rnd(C, seed=4).abs() + 0.5).to(DEV), rnd(C, seed=5).to(DEV)
    mean, invstd, gamma = rnd(C, seed=6).to(DEV), (rnd(C, seed=7).abs() + 0.5).to(DEV), rnd(C, seed=8).to(DEV)
    f = lambda t: t.float()
    # BN apply: plain, identity residual, shortcut A (stride 2, 32 of 64 channels)
    assert torch.equal(ops.bn_apply(y, scale, shift, None, 1, True), ops.bn_apply(f(y), scale, shift, None, 1, True).to(BF))
    assert torch.equal(ops.bn_apply(y, scale, shift, res, 1, True), ops.bn_apply(f(y), scale, shift, f(res), 1, True).to(BF))
    big = rnd(B, 2 * D, 2 * H, 2 * W, 32, seed=9).to(DEV).to(BF)
    assert torch.equal(ops.bn_apply(y, scale, shift, big, 2, True), ops.bn_apply(f(y), scale, shift, f(big), 2, True).to(BF))
    # BN backward: reduce (fp32 partials), apply (+ column sums)
    z = ops.bn_apply(y, scale, shift, res, 1, True)
    for zz, sc, sh in ((z, None, None), (None, scale, shift)):
        p16 = ops.bn_bwd_reduce(dz, zz, y, mean, invstd, True, sc, sh)
        p32 = ops.bn_bwd_reduce(f(dz), None if zz is None else f(zz), f(y), mean, invstd, True, sc, sh)
        assert torch.equal(p16, p32)
        sums = ops.reduce_partials(p16)
        d16, c16 = ops.bn_bwd_apply(dz, zz, y, mean, invstd, gamma, sums, float(B * D * H * W), True, sc, sh, want_colsum=True)
        d32, c32 = ops.bn_bwd_apply(f(dz), None if zz is None else f(zz), f(y), mean, invstd, gamma, sums, float(B * D * H * W),
                                    True, sc, sh, want_colsum=True)
        assert torch.equal(d16, d32.to(BF)) and torch.equal(c16, c32)
    assert torch.equal(ops.colsum(dz), ops.colsum(f(dz)))
    # max pool forward / backward (+ skip-gradient add)
    p16, a16 = ops.maxpool_fwd(y)
    p32, a32 = ops.maxpool_fwd(f(y))
    assert torch.equal(p16, p32.to(BF)) and torch.equal(a16, a32)
    gp = rnd(*p16.shape, seed=10).to(DEV).to(BF)
    assert torch.equal(ops.maxpool_bwd(gp, a16, tuple(y.shape), res), ops.maxpool_bwd(f(gp), a32, tuple(y.shape), f(res)).to(BF))
    # upsample + concat forward (tiled and untiled forms) / backward
    # (channel counts that are not multiples of 8 take the 4-channel-per-thread form of the bf16 kernels)
    y12 = rnd(1, 5, 6, 7, 12, seed=20).to(DEV).to(BF)
    q16, b16 = ops.maxpool_fwd(y12)
    q32, b32 = ops.maxpool_fwd(f(y12))
    assert torch.equal(q16, q32.to(BF)) and torch.equal(b16, b32)
    gq = rnd(*q16.shape, seed=21).to(DEV).to(BF)
    assert torch.equal(ops.maxpool_bwd(gq, b16, tuple(y12.shape), None), ops.maxpool_bwd(f(gq), b32, tuple(y12.shape), None).to(BF))
    for sshape, kshape in (((1, 3, 4, 5, 64), (1, 7, 9, 10, 32)), ((1, 32, 32, 32, 64), (1, 64, 64, 64, 64)),
                           ((1, 3, 4, 5, 12), (1, 7, 9, 10, 20))):
        src, skip = rnd(*sshape, seed=11).to(DEV).to(BF), rnd(*kshape, seed=12).to(DEV).to(BF)
        c16 = ops.upcat_fwd(src, skip)
        assert torch.equal(c16, ops.upcat_fwd(f(src), f(skip)).to(BF))
        gc = rnd(*c16.shape, seed=13).to(DEV).to(BF)
        s16, k16 = ops.upcat_bwd(gc, sshape, kshape)
        s32, k32 = ops.upcat_bwd(f(gc), sshape, kshape)
        assert torch.equal(s16, s32.to(BF)) and torch.equal(k16, k32.to(BF))
    # heads: dense maps / pooling sums are fp32 on both paths
    xh = rnd(2, 4, 6, 8, 32, seed=14).to(DEV).to(BF)
    lungs = (rnd(2, 8, 12, 16, seed=15) > 0).float().to(DEV)
    for NO, sig in ((2, True), (9, False)):
        hw, hb = rnd(NO, 32, seed=16).to(DEV), rnd(NO, seed=17).to(DEV)
        d16, q16 = ops.head_fwd(xh, hw, hb, lungs if sig else None, sig)
        d32, q32 = ops.head_fwd(f(xh), hw, hb, lungs if sig else None, sig)
        # (the two instantiations contract their 32-term dot products into fmas in different orders: last-bit
        # differences in fp32, hence at most one bf16 ulp on a few elements after the store rounding)
        assert torch.allclose(d16, d32, rtol=1e-4, atol=1e-6) and torch.allclose(q16, q32, rtol=1e-4, atol=1e-4)
        gd, gpool = rnd(*d16.shape, seed=18).to(DEV), rnd(2, NO, seed=19).to(DEV)
        x16, w16 = ops.head_bwd(xh, hw, d16 if sig else None, gd, gpool, lungs if sig else None, sig)
        x32, w32 = ops.head_bwd(f(xh), hw, d16 if sig else None, gd, gpool, lungs if sig else None, sig)
        assert rel_l2(x16.float().cpu(), x32.cpu()) < 3e-3 and float((x16.float() - x32).abs().max()) <= 2 ** -7 * float(x32.abs().max())
        assert torch.allclose(w16, w32, rtol=1e-4, atol=1e-5)
    # stem: fp32 arithmetic, bf16 store; weight gradient from a bf16 dy
    xs, ws = rnd(1, 16, 24, 24, seed=20).to(DEV), (rnd(64, 1, 7, 7, 7, seed=21) * 0.05).to(DEV)
    y16, st16 = ops.stem_fwd(xs, ws, True, BF)
    y32, st32 = ops.stem_fwd(xs, ws, True)
    assert torch.equal(y16, y32.to(BF)) and torch.equal(st16, st32)
    gy = rnd(*y16.shape, seed=22).to(DEV).to(BF)
    assert torch.equal(ops.stem_bwd_weight(xs, gy), ops.stem_bwd_weight(xs, f(gy)))
    # casts round to nearest even and are exact back
    t = rnd(1000, seed=23).to(DEV)
    assert torch.equal(ops.cast(t, BF), t.to(BF)) and torch.equal(ops.cast(t.to(BF), torch.float32), t.to(BF).float())


def _build(factory, seed):
    from bodyct_dram_emph_subtype_amd import med3d
    torch.manual_seed(seed)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    return getattr(med3d, factory)(**kw)


def _loss(dense, outs, cls):
    if cls:
        return outs[0].float().square().sum() + outs[1].float().sum()
    return outs[0].float().sum() * 0.7 - outs[1].float().sum() * 1.3 + 0.1 * (dense[0].float() * dense[1].float()).mean()


@pytest.mark.parametrize("factory,shape,mode", [("resnet18segreg", (2, 1, 32, 64, 64), "attr"),
                                                ("resnet18segcls", (1, 1, 24, 48, 40), "autocast"),
                                                ("resnet50segreg", (1, 1, 32, 64, 96), "attr")])
def test_network_train_step_bf16_storage(factory, shape, mode):
    """One train step with bf16 activations (selected by module.storage_dtype, or -- as Lightning's `--precision bf16`
    does -- by running inside torch.autocast(bfloat16)) against the fp32 oracle, the reference's own autocast
    arithmetic, and the fp64 oracle pinned to the bf16 forward's decisions."""
    from bodyct_dram_emph_subtype_amd.engine import forward_decisions
    cls = factory.endswith("cls")
    m = _build(factory, 3)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(*shape, generator=g)
    lungs = (torch.rand(*shape, generator=g) > 0.3).float()
    m = m.to(DEV).train()
    if mode == "attr":
        m.storage_dtype = BF
        dense, outs = m(x.to(DEV), lungs.to(DEV))
    else:
        with torch.autocast("cuda", dtype=BF):
            dense, outs = m(x.to(DEV), lungs.to(DEV))
    saved = dense[0].grad_fn.saved_state
    assert saved["xs"].dtype == BF and saved["xup3"].dtype == BF and dense[0].dtype == torch.float32
    pins = {k: v.cpu() for k, v in forward_decisions(saved).items()}
    _loss(dense, outs, cls).backward()
    torch.cuda.synchronize()
    # forward quantities vs the fp32 oracle, next to the reference arithmetic under CPU autocast(bfloat16)
    with torch.no_grad():
        d32, o32 = orc.forward(dict(sd0), x, lungs, factory, train=True)
        with torch.autocast("cpu", dtype=BF):
            dac, oac = orc.forward(dict(sd0), x, lungs, factory, train=True)
    for a, b, r in zip(outs, o32, oac):
        e = float((a.detach().cpu() - b).abs().max() / b.abs().max())
        e_ref = float((r.float() - b).abs().max() / b.abs().max())
        print(f"[{factory} bf16] pooled: hip {e:.2e}, reference autocast {e_ref:.2e}")
        assert e <= (5e-3 if factory.startswith("resnet50") else 1e-3)
    for a, b, r in zip(dense, d32, dac):
        e, e_ref = rel_l2(a.detach().cpu(), b), rel_l2(r.float(), b)
        print(f"[{factory} bf16] dense: hip {e:.2e}, reference autocast {e_ref:.2e}")
        # (the 16x64x64 ResNet-50 fixture: 1.5e-1 for the reference's autocast arithmetic too -- its stride-8 stages hold
        # 128 voxels, every BatchNorm statistic there is a 128-sample estimate of bf16-rounded values)
        assert e <= max(2e-2, 1.2 * e_ref)
        # ABSOLUTE bar next to the relative one for ResNet-18.  The ResNet-50 fixture is a SMOKE case in bf16 (its
        # stride-8 stages hold 384 voxels: an absolute bar on it would have to sit at 2.8e-1 and hold nothing); the
        # absolute ResNet-50 bars are test_resnet50_bf16_mid_size_vs_oracle's, at a size where BatchNorm is conditioned
        if not factory.startswith("resnet50"):
            assert e <= 2e-2
    # BN running statistics follow the same batch statistics
    ns = {}
    orc.forward(dict(sd0), x, lungs, factory, train=True, new_stats=ns)
    sd = m.state_dict()
    for k in ("bn1.running_mean", "bn1.running_var", "us3.1.running_var"):
        assert np.allclose(sd[k].cpu().numpy(), ns[k].numpy(), rtol=2e-2, atol=2e-3), k
    # gradients (fp32 tensors) vs the fp64 oracle on the bf16 forward's own decisions; next to it the distance of the
    # reference's own autocast(bfloat16) gradients from its fp32 gradients (decisions free): the bar is 1e-1 per
    # tensor, or 1.5 x that reference distance where bf16 itself is worse on the fixture (ResNet-50 at 16x64x64)
    lv = {k: (v.clone().double().requires_grad_(True) if k in names else (v.clone().double() if v.is_floating_point() else v.clone()))
          for k, v in sd0.items()}
    d, o = orc.forward(lv, x.double(), lungs.double(), factory, train=True, pins=pins)
    _loss(d, o, cls).backward()

    def free_grads(autocast):
        leaves = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd0.items()}
        if autocast:
            with torch.autocast("cpu", dtype=BF):
                dd, oo = orc.forward(leaves, x, lungs, factory, train=True)
        else:
            dd, oo = orc.forward(leaves, x, lungs, factory, train=True)
        _loss(dd, oo, cls).backward()
        return {n: leaves[n].grad for n in names}
    gref32, grefac = free_grads(False), free_grads(True)
    worst = (0.0, 0.0, "")
    for n, p in m.named_parameters():
        assert p.grad.dtype == torch.float32
        if n.endswith(".0.bias") and n.startswith("us"):
            continue
        e, e_ref = rel_l2(p.grad.cpu(), lv[n].grad), rel_l2(grefac[n], gref32[n])
        if e_ref != e_ref:          # the reference's autocast backward overflowed to NaN on this tensor
            e_ref = float("inf")
        worst = max(worst, (e, e_ref, n))
        # (2x the reference autocast's own distance where that is larger: every convolution, the strided one included,
        # multiplies bf16 operands here and every activation is ROUNDED to bf16 between layers, which the reference's
        # CPU autocast -- fp32 BatchNorm outputs -- does not do; 1.5x held while the strided convolution ran in fp32)
        assert e <= max(1e-1, 2.0 * e_ref), f"{n}: bf16-storage gradient vs decision-pinned fp64 oracle {e:.2e} (reference autocast vs its fp32 self: {e_ref:.2e})"
        # absolute cap for ResNet-18: 1.5 x the worst measured tensor (9.3e-2); ResNet-50 at this size: smoke (see above)
        if not factory.startswith("resnet50"):
            assert e <= 1.4e-1, f"{n}: {e:.2e} above the absolute bar"
    print(f"[{factory} bf16] worst gradient vs decision-pinned fp64 oracle (hip, reference-autocast-vs-fp32, tensor): {worst}")


# ResNet-50, bf16 storage, 1x64x128x128, END TO END against the storage-aware fp64 oracle (measured, round 5: pooled scores
# 2.8e-3 / 9.7e-4, dRAM volumes 9.3e-2 / 1.28e-1, worst gradient 2.2e-1 (bn1.bias), median 1.1e-1; against the fp32 oracle
# the same run sits at 2.0e-1 / 2.8e-1 -- exactly where the reference's own autocast arithmetic sits, 1.98e-1 / 2.79e-1).
# Config 2's bars (1e-3 / 4e-2) are not attainable end to end by ANY bf16 arithmetic on this model: a rounding TIE that
# falls differently in two correct implementations is amplified like any other rounding.  The un-amplified statement is
# test_resnet50_bf16_every_unit_vs_oracle_teacher_forced (one rounding per unit); these are the end-to-end caps.
R50_MID_POOLED_BAR = 4e-3    # max-relative, pooled regression scores
R50_MID_DENSE_BAR = 1.6e-1   # relative L2, dRAM volumes (1.25 x measured; half the distance to the fp32 oracle)
R50_MID_GRAD_BAR = 3e-1      # per-tensor relative L2 of the parameter gradients (1.35 x measured)


def test_resnet50_bf16_mid_size_vs_oracle():
    """ResNet-50 + dRAM head (the reference's default model, reference train.py:22) in bf16 storage (Lightning
    `--precision bf16`, train.py:46; BASELINE configs[4]'s precision) at 1x64x128x128 -- stride-8 stages of 8x16x16 =
    2,048 voxels, so every BatchNorm has a conditioned statistic -- with ABSOLUTE bars.

    Yardstick.  A randomly initialised ResNet-50 amplifies ANY bf16 rounding by ~100x through its 54 BatchNorm layers:
    against the fp32 oracle the dRAM volumes sit 2.0e-1 / 2.8e-1 (relative L2) away -- and the reference's own arithmetic
    under CPU autocast(bfloat16) sits 2.0e-1 / 2.8e-1 away as well (printed below), so no bar against the fp32 forward
    can hold anything for this model.  The oracle is therefore evaluated in fp64 WITH the build's storage roundings
    (oracle.forward(storage=bfloat16): input, convolution weights, every stored activation rounded once to bf16; statistics,
    affine parameters and heads exact) on the HIP forward's own ReLU / max-pool decisions: what is left between the two
    is accumulation order, bf16 rounding ties, the other rounding points of us1's first convolution (csrc/upmix.hip) and
    -- for the gradients -- the bf16 rounding of the activation GRADIENTS, which the oracle's backward does not do; all of
    it amplified by the network.  Bars: R50_MID_* above (absolute)."""
    from bodyct_dram_emph_subtype_amd.engine import forward_decisions
    factory = "resnet50segreg"
    m = _build(factory, 5)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    g = torch.Generator().manual_seed(11)
    shape = (1, 1, 64, 128, 128)
    x = torch.randn(*shape, generator=g)
    zz, yy, xx = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 128), torch.linspace(-1, 1, 128), indexing="ij")
    lungs = (((zz / 0.8) ** 2 + (yy / 0.7) ** 2 + (xx / 0.8) ** 2) <= 1.0).float()[None, None]
    m = m.to(DEV).train()
    m.storage_dtype = BF
    dense, outs = m(x.to(DEV), lungs.to(DEV))
    saved = dense[0].grad_fn.saved_state
    assert saved["xs"].dtype == BF and saved["xup3"].dtype == BF
    pins = {k: v.cpu() for k, v in forward_decisions(saved).items()}
    _loss(dense, outs, False).backward()
    torch.cuda.synchronize()
    # context: the fp32 oracle, and the reference's own arithmetic under `--precision bf16` (CPU autocast)
    with torch.no_grad():
        d32, o32 = orc.forward(dict(sd0), x, lungs, factory, train=True)
        with torch.autocast("cpu", dtype=BF):
            dac, oac = orc.forward(dict(sd0), x, lungs, factory, train=True)
    print(f"[resnet50 bf16 1x64x128x128] vs the fp32 oracle: pooled scores "
          f"{[float((a.detach().cpu() - b).abs().max() / b.abs().max()) for a, b in zip(outs, o32)]} (reference autocast: "
          f"{[float((r.float() - b).abs().max() / b.abs().max()) for r, b in zip(oac, o32)]}); dRAM volumes relative L2 "
          f"{[rel_l2(a.detach().cpu(), b) for a, b in zip(dense, d32)]} (reference autocast: "
          f"{[rel_l2(r.float(), b) for r, b in zip(dac, d32)]})")

    def pinned(storage):
        lv = {k: (v.clone().double().requires_grad_(True) if k in names else (v.clone().double() if v.is_floating_point() else v.clone()))
              for k, v in sd0.items()}
        d, o = orc.forward(lv, x.double(), lungs.double(), factory, train=True, pins=pins, storage=storage)
        _loss(d, o, False).backward()
        return [t.detach() for t in d], [t.detach() for t in o], {n: lv[n].grad for n in names}
    d_st, o_st, g_st = pinned(BF)            # THE yardstick: fp64 arithmetic, the build's storage roundings
    _, _, g_plain = pinned(None)             # (context: the fp64 oracle without them)
    e_pool = [float((a.detach().cpu().double() - b).abs().max() / b.abs().max()) for a, b in zip(outs, o_st)]
    e_dense = [rel_l2(a.detach().cpu(), b) for a, b in zip(dense, d_st)]
    errs, errs_plain = [], []
    for n, p in m.named_parameters():
        if n.endswith(".0.bias") and n.startswith("us"):
            continue
        errs.append((rel_l2(p.grad.cpu(), g_st[n]), n))
        errs_plain.append((rel_l2(p.grad.cpu(), g_plain[n]), n))
    errs.sort(reverse=True)
    errs_plain.sort(reverse=True)
    print(f"[resnet50 bf16 1x64x128x128] vs the storage-aware fp64 oracle on the same decisions: pooled scores {e_pool}, "
          f"dRAM volumes relative L2 {e_dense}; gradients worst {errs[:3]}, median {errs[len(errs) // 2][0]:.2e} "
          f"(against the fp64 oracle WITHOUT the storage roundings: worst {errs_plain[0]}, median {errs_plain[len(errs_plain) // 2][0]:.2e})")
    assert max(e_pool) <= R50_MID_POOLED_BAR and max(e_dense) <= R50_MID_DENSE_BAR
    assert errs[0][0] <= R50_MID_GRAD_BAR, f"{errs[0][1]}: {errs[0][0]:.2e}"


def _ncdhw(t):
    return t.detach().float().permute(0, 4, 1, 2, 3).contiguous().cpu().double()


def test_resnet50_bf16_every_unit_vs_oracle_teacher_forced():
    """The end-to-end comparison above carries the network's own amplification of a rounding (ResNet-50 at a random
    initialisation: ~100x).  This one does not: every convolution + BatchNorm (+ residual) + ReLU unit of the SAME
    1x64x128x128 bf16 forward is checked on its own -- the oracle's arithmetic (fp64 F.conv3d on the bf16-rounded
    weights, batch statistics of the rounded output, affine + residual + ReLU, one rounding per stored tensor: reference
    med3d.py:121-124, 153-159, 164-184, 85-89) applied to the INPUT THE HIP UNIT ITSELF READ (its saved bf16 tensor) must
    reproduce the unit's stored pre-BatchNorm output, its batch mean / inverse standard deviation and its stored
    activation.  What may differ is a bf16 rounding tie (one ulp = 2^-8 relative on an element whose exact value lies
    within accumulation error of a rounding boundary): relative L2 <= UNIT_TOL = 2e-4 per tensor, statistics <= 1e-6.
    Covers all 16 Bottleneck blocks (1x1x1, 3x3x3 dilated, the stride-2 convolution through space-to-depth, shortcut-A
    and identity residuals), both decoder blocks (us1's first convolution runs WITHOUT the up-sampled tensor --
    csrc/upmix.hip, other rounding points: UPMIX_TOL), us3 and the stem."""
    import torch.nn.functional as F
    factory = "resnet50segreg"
    m = _build(factory, 5)
    sd0 = {k: v.clone().double() if v.is_floating_point() else v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 1, 64, 128, 128, generator=g)
    m = m.to(DEV).train()
    m.storage_dtype = BF
    dense, outs = m(x.to(DEV), None)
    saved = dense[0].grad_fn.saved_state
    # measured (round 5): pre-BatchNorm outputs 4.7e-5, activations 2.6e-5, statistics 6.4e-8; the upmix unit 2.7e-3
    UNIT_TOL, UPMIX_TOL, STAT_TOL = 2e-4, 8e-3, 1e-6

    def qd(t):
        return t.to(BF).double()

    worst = {"y": (0.0, ""), "y-upmix": (0.0, ""), "z": (0.0, ""), "stat": (0.0, "")}

    def check(kind, name, got, ref, tol):
        e = rel_l2(got, ref)
        worst[kind] = max(worst[kind], (e, name))
        assert e <= tol, f"{name} [{kind}]: {e:.2e} > {tol:g}"

    def unit(c, xin, res=None, z=None):
        """conv -> y (stored), batch statistics, BN + residual + ReLU -> z (stored by the unit, or -- a unit whose
        BatchNorm-apply was left to its consumer -- the tensor that consumer materialised and read)"""
        geo = c["g"]
        w = qd(sd0[c["w"]])
        b = sd0[c["b"]] if c["b"] else None
        y_ref = qd(F.conv3d(xin, w, b, geo.stride, geo.pad, geo.dil))
        y = _ncdhw(c["y"])
        check("y", c["w"], y, y_ref, UNIT_TOL)
        bn_z(c, y, res, z)

    def bn_z(c, y, res, z=None):
        mean, var = y.mean((0, 2, 3, 4)), y.var((0, 2, 3, 4), unbiased=False)
        check("stat", c["bn"] + ".mean", c["mean"].cpu().double(), mean, STAT_TOL)
        check("stat", c["bn"] + ".invstd", c["invstd"].cpu().double(), torch.rsqrt(var + 1e-5), STAT_TOL)
        z = c.get("z") if c.get("z") is not None else z
        assert z is not None, c["bn"]
        sh = (1, -1, 1, 1, 1)
        zz = (y - mean.view(sh)) * torch.rsqrt(var.view(sh) + 1e-5) * sd0[c["bn"] + ".weight"].view(sh) + sd0[c["bn"] + ".bias"].view(sh)
        if res is not None:
            zz = zz + res
        check("z", c["bn"], _ncdhw(z), qd(torch.relu(zz)), UNIT_TOL)

    # stem (reference med3d.py:371-373)
    y0 = _ncdhw(saved["y0"])
    check("y", "conv1.weight", y0, qd(F.conv3d(qd(x), qd(sd0["conv1.weight"]), None, 2, 3)), UNIT_TOL)
    bn_z(dict(bn="bn1", mean=saved["mean0"], invstd=saved["invstd0"], z=saved["xs"]), y0, None)
    # Bottleneck blocks (med3d.py:164-184)
    for c1, c2, c3, has_ds in saved["blocks"]:
        xin = _ncdhw(c1["x"])
        unit(c1, xin, z=c2["x"])
        unit(c2, _ncdhw(c2["x"]), z=c3["x"])
        if has_ds:
            res = orc.shortcut_a(xin, c3["g"].Cout, c2["g"].stride)
        else:
            res = xin
        unit(c3, _ncdhw(c3["x"]), res)
    # decoder (med3d.py:85-89): us1 without the up-sampled tensor, us2 on the materialised concat, us3
    for key in ("cu1", "cu2"):
        ca, cb = saved[key][:2]
        if ca.get("kind") == "upmix":
            cat = orc.crop_concat(qd(orc.upsample2_trilinear(_ncdhw(ca["src"]))), _ncdhw(ca["skip"]))
            y_ref = qd(F.conv3d(cat, qd(sd0[ca["w"]]), sd0[ca["b"]], 1, 1))
            ya = _ncdhw(ca["y"])
            check("y-upmix", ca["w"] + " (upmix)", ya, y_ref, UPMIX_TOL)
            bn_z(ca, ya, None, cb["x"])
        else:
            unit(ca, _ncdhw(ca["x"]), z=cb["x"])
        unit(cb, _ncdhw(cb["x"]), z=(saved["cu2"][0]["src"] if (key == "cu1" and saved["cu2"][0].get("kind") == "upmix") else None)
             if cb.get("z") is None else None)
    unit(saved["cu3"], _ncdhw(saved["cu3"]["x"]), z=saved["xup3"])
    print(f"[resnet50 bf16 1x64x128x128, every unit on its own stored input] worst pre-BN output {worst['y']} (us1's first convolution, other rounding points: {worst['y-upmix']}), "
          f"worst activation {worst['z']}, worst statistic {worst['stat']}")


def test_bf16_eval_forward_and_train_steps_run_the_optimizer():
    """eval-mode forward (packed-weight cache keyed by storage type) and three optimizer steps: the loss of a fixed
    batch goes down, parameters stay fp32."""
    from bodyct_dram_emph_subtype_amd.models import cls_train_loss
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    m = _build("resnet18segcls", 5).to(DEV)
    m.storage_dtype = BF
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 1, 16, 32, 32, generator=g).to(DEV)
    lungs = (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.3).float().to(DEV)
    cle, pse = torch.tensor([1, 4]).to(DEV), torch.tensor([0, 2]).to(DEV)
    cw, pw = torch.full((6,), 1 / 6, device=DEV), torch.full((3,), 1 / 3, device=DEV)
    m.eval()
    with torch.no_grad():
        d_bf, o_bf = m(x, lungs)
        m.storage_dtype = torch.float32
        d_32, o_32 = m(x, lungs)
        m.storage_dtype = BF
    assert float((o_bf[0] - o_32[0]).abs().max() / o_32[0].abs().max()) < 5e-3
    m.train()
    opt = FusedAdam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = cls_train_loss(m(x, lungs)[1], cle, pse, cw, pw)[0]
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses
    assert all(p.dtype == torch.float32 for p in m.parameters())


@pytest.mark.slow
def test_config2_as_specified_full_size_bf16_vs_fp32_path():
    """BASELINE configs[2] AS SPECIFIED: resnet18segreg + dRAM loss, batch 2, 1x128x256x256, bf16 storage -- one full
    train step against the fp32 path of the same library on the same inputs (which tests/test_network_gpu.py holds
    to the fp64 oracle at this size; a CPU oracle run of this batch costs minutes).  Pooled scores max-relative
    <= 1e-3, dRAM volumes relative L2 <= 4e-2 (measured 3.0e-2: ~40 roundings to 8 bits along the deepest path),
    dRAM train loss <= 2e-3, BatchNorm running statistics <= 1e-2.  Gradients: the dRAM loss at a random
    initialisation sits on the clamp kink of models.py:527 and is ill-conditioned in the reference's own fp32
    arithmetic (tests/test_network_gpu.py::_dram_loss_checks: its end-to-end gradient is 130 % from fp64), so the
    backward pass is compared through a SMOOTH objective over scores and dRAM volumes: per-tensor relative L2
    <= 0.5 against the fp32 path (measured: up to 0.37 at conv1.weight, the end of the longest chain; the reference's
    own autocast arithmetic sits 0.27-0.41 from its fp32 self on the fixtures above) -- the two runs take different ReLU / max-pool decisions wherever an activation is
    within bf16 rounding of zero, so this is a sanity bar (the decision-pinned comparisons above are the parity
    bars); the step must reproduce itself bit for bit."""
    from bodyct_dram_emph_subtype_amd import med3d, models
    torch.manual_seed(0)
    m = med3d.resnet18segreg().to(DEV).train()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator(device=DEV).manual_seed(1234)
    B, D, H, W = 2, 128, 256, 256
    x = torch.randn(B, 1, D, H, W, device=DEV, generator=g)
    z = (torch.arange(D, device=DEV).float() - (D - 1) / 2) / (0.4 * D)
    y = (torch.arange(H, device=DEV).float() - (H - 1) / 2) / (0.35 * H)
    xx = (torch.arange(W, device=DEV).float() - (W - 1) / 2) / (0.4 * W)
    lungs = ((z[:, None, None] ** 2 + y[None, :, None] ** 2 + xx[None, None, :] ** 2) <= 1.0).float()[None, None].expand(B, 1, D, H, W).contiguous()
    ems = ((x < -1.0).float() * lungs)
    cle, pse = torch.tensor([4, 1], device=DEV), torch.tensor([0, 2], device=DEV)
    cw, pw = torch.tensor([0.3, 0.2], device=DEV), torch.tensor([0.6, 0.1], device=DEV)

    def step(storage):
        m.load_state_dict(sd0)
        m.storage_dtype = storage
        m.zero_grad(set_to_none=True)
        dense, outs = m(x, lungs)
        loss, _ = models.reg_train_loss(dense, outs, lungs, ems, cle, pse, cw, pw)
        (outs[0].sum() * 0.7 - outs[1].sum() * 1.3 + 0.1 * (dense[0] * dense[1]).mean()).backward()
        torch.cuda.synchronize()
        return ([d.detach().clone() for d in dense], [o.detach().clone() for o in outs], float(loss),
                {n: p.grad.clone() for n, p in m.named_parameters()},
                {k: v.clone() for k, v in m.state_dict().items() if "running" in k})

    d32, o32, l32, g32, s32 = step(torch.float32)
    d16, o16, l16, g16, s16 = step(BF)
    for a, b in zip(o16, o32):
        assert float((a - b).abs().max() / b.abs().max()) <= 1e-3
    for a, b in zip(d16, d32):
        assert rel_l2(a.cpu(), b.cpu()) <= 4e-2
    # ... and against the ORACLE's fp32 forward on the same inputs (the reference's arithmetic, not this library's):
    # pooled scores max-relative <= 1e-3, dRAM volumes relative L2 <= 4e-2 -- the bars the fixtures use
    with torch.no_grad():
        d_or, o_or = orc.forward({k: v.cpu() for k, v in sd0.items()}, x.cpu(), lungs.cpu(), "resnet18segreg", train=True)
    e_o = max(float((a.cpu() - b).abs().max() / b.abs().max()) for a, b in zip(o16, o_or))
    e_d = max(rel_l2(a.cpu(), b) for a, b in zip(d16, d_or))
    print(f"[config 2 as specified, bf16] vs the fp32 ORACLE forward: pooled scores {e_o:.2e} (bar 1e-3), dRAM volumes {e_d:.2e} (bar 4e-2)")
    assert e_o <= 1e-3 and e_d <= 4e-2
    del d_or, o_or
    assert abs(l16 - l32) <= 2e-3 * abs(l32)
    for k in s32:
        assert float((s16[k] - s32[k]).abs().max()) <= 1e-2 * float(s32[k].abs().max()) + 1e-4, k
    worst = (0.0, "")
    for n in g32:
        if n.endswith(".0.bias") and n.startswith("us"):
            continue
        e = rel_l2(g16[n].cpu(), g32[n].cpu())
        worst = max(worst, (e, n))
        assert e <= 0.5, f"{n}: bf16-storage vs fp32 path gradient {e:.2e}"
    _, _, l16b, g16b, _ = step(BF)
    assert l16b == l16 and all(torch.equal(g16[n], g16b[n]) for n in g16)
    print(f"[config 2 as specified, bf16] loss {l16:.6f} (fp32 path {l32:.6f}); worst gradient distance to the fp32 path {worst}")


def test_graphed_bf16_train_step_equals_eager():
    """graph.GraphedTrainStep on the bf16 storage path: three replays of the captured step leave the weights, running
    statistics and Adam state bit-identical to three eager steps (buffer-descriptor DMA, rings and slabs are all
    plain stream work with host-side arguments fixed at capture)."""
    from bodyct_dram_emph_subtype_amd import med3d
    from bodyct_dram_emph_subtype_amd.graph import GraphedTrainStep
    from bodyct_dram_emph_subtype_amd.models import cls_train_loss
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    batch = (torch.randn(2, 1, 16, 32, 32, generator=g).to(DEV), (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.3).float().to(DEV),
             torch.randint(0, 6, (2,), generator=g).to(DEV), torch.randint(0, 3, (2,), generator=g).to(DEV))
    cw, pw = torch.full((6,), 1 / 6, device=DEV), torch.full((3,), 1 / 3, device=DEV)

    def churn():
        """More than eight OTHER sets of weights through the one-launch packing (its cache of device work lists holds
        eight) and scratch requests from another capture-free stream, then allocations that recycle whatever was
        freed, filled with ones: a captured graph whose work list or scratch had been given up would now pack garbage."""
        from bodyct_dram_emph_subtype_amd import ops
        with ops.launch_scope(DEV):
            for i in range(10):
                ws = [torch.randn(32 + 32 * (i % 3), 32, 3, 3, 3, device=DEV) for _ in range(2 + i % 2)]
                assert ops.pack_conv_weights_bf16_multi(ws) is not None
        torch.cuda.synchronize()
        junk = [torch.full((n,), 0xFF, device=DEV, dtype=torch.uint8) for n in (512, 4096, 32768, 1 << 18, 1 << 21) for _ in range(8)]
        torch.cuda.synchronize()
        del junk

    def run(graphed):
        torch.manual_seed(11)
        m = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV).train()
        m.storage_dtype = BF
        opt = FusedAdam(m.parameters(), lr=1e-3, capturable=True)
        loss_fn = lambda i, l, c, p: cls_train_loss(m(i, l)[1], c, p, cw, pw)[0]      # noqa: E731
        if graphed:
            step = GraphedTrainStep(m, opt, loss_fn, batch, warmup=2)
            assert step._keep, "the capture kept no scratch buffer / work list alive"
            churn()
        else:
            def step(*b):
                opt.zero_grad(set_to_none=True)
                loss = loss_fn(*b)
                loss.backward()
                opt.step()
                return loss.detach()
            step(*batch); step(*batch)
        for _ in range(3):
            loss = step(*batch).clone()
        torch.cuda.synchronize()
        return float(loss), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    lg, sg = run(True)
    le, se = run(False)
    assert lg == le
    for k in se:
        assert torch.equal(sg[k], se[k]), k
