"""GPU parity tests: every HIP op (called through the C ABI via hp_vae_gan_amd.ops) against the CPU oracle on the
same seeded inputs, plus the reference-generated golden block fixtures.  Tolerance 1e-3 relative (north_star);
fp32 MFMA == fmaf chain, so the observed error is ~1e-6."""
import pytest
import torch

from helpers import RTOL, assert_close, bn_bias_atol, load_golden, opt_from
from oracle import hpvg_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    import hp_vae_gan_amd as hp  # noqa: F401
    from hp_vae_gan_amd import ops as _ops
    return _ops


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # (B, Cin, Cout, spatial)
    (2, 3, 64, (5, 7, 9)), (2, 64, 64, (3, 5, 6)), (1, 64, 64, (4, 18, 33)), (2, 64, 3, (3, 6, 5)), (2, 64, 1, (2, 5, 7)),
    (1, 128, 64, (2, 4, 5)), (1, 64, 128, (2, 5, 4)), (1, 8, 8, (1, 1, 1)), (1, 16, 24, (3, 4, 7)), (2, 3, 8, (4, 23, 41)),
    (1, 64, 64, (5, 36, 65)), (1, 64, 64, (2, 9, 130)), (1, 5, 70, (2, 3, 300)),
    (2, 3, 64, (11, 13)), (2, 64, 64, (9, 10)), (2, 64, 64, (24, 33)), (1, 64, 3, (30, 41)), (1, 64, 1, (7, 12)),
    (1, 128, 64, (6, 5)), (1, 64, 128, (48, 65)), (1, 64, 64, (96, 129)),
]


@pytest.mark.parametrize("B,Cin,Cout,sp", CONV_CASES)
def test_conv_fwd_bwd(ops, B, Cin, Cout, sp):
    nd = len(sp)
    x = _rand(B, Cin, *sp, seed=1).requires_grad_(True)
    w = _rand(Cout, Cin, *([3] * nd), seed=2, scale=0.1).requires_grad_(True)
    b = _rand(Cout, seed=3).requires_grad_(True)
    gy = _rand(B, Cout, *sp, seed=4)
    y = O.conv(x, w, b)
    dx, dw, db = torch.autograd.grad(y, [x, w, b], gy)

    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = ops.Conv.apply(xd, wd, bd, False)
    assert_close(yd, y, RTOL, "conv.y")
    dxd, dwd, dbd = torch.autograd.grad(yd, [xd, wd, bd], gy.to(DEV))
    assert_close(dxd, dx, RTOL, "conv.dx")
    assert_close(dwd, dw, RTOL, "conv.dw")
    assert_close(dbd, db, RTOL, "conv.db")


@pytest.mark.parametrize("B,Cin,Cout,sp", [
    (2, 64, 64, (7, 72, 129)),   # 560 tiles: one data-parallel round + 48 stream-K tiles cut over 512 workgroups
    (2, 64, 64, (5, 45, 81)),    # every workgroup owns a fraction of a tile (all tiles finished by the fix-up kernel)
    (1, 128, 70, (3, 30, 50)),   # 16 channel chunks, ragged Cout
    (2, 64, 64, (200, 300)),     # 2-D, several rounds
    (2, 24, 3, (3, 64, 100)),    # narrow output, 3 chunks
])
def test_conv_stream_k_schedules_and_plain_launch(ops, B, Cin, Cout, sp):
    """The stream-K schedule (workspace given) and the one-workgroup-per-tile launch (ws = NULL) against the oracle, on
    shapes whose tiles are split across workgroups; the two launches agree to fp32 summation order."""
    import ctypes
    from hp_vae_gan_amd import lib as hplib
    nd = len(sp)
    x = _rand(B, Cin, *sp, seed=11)
    w = _rand(Cout, Cin, *([3] * nd), seed=12, scale=0.05)
    b = _rand(Cout, seed=13)
    want = O.conv(x, w, b)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y_sk = ops.conv_fwd_raw(xd, wd, bd)
    assert_close(y_sk, want, RTOL, "conv.streamk")
    Bq, C, T, H, W = ops.geom(xd)
    KT = 3 if nd == 3 else 1
    assert (hplib.call("hpvg_conv_fwd_ws_bytes", Bq, Cin, Cout, T, H, W, KT) > 0) == (Cout > 4)  # narrow outputs: direct kernel
    wp = ops.pack_weight(wd, False)
    y_pl = torch.full_like(y_sk, float("nan"))
    hplib.call("hpvg_conv_fwd_f32", hplib.ptr(xd), hplib.ptr(wp), hplib.ptr(bd), None, None, 0, hplib.ptr(y_pl), 0, None, None,
               ctypes.c_size_t(0), Bq, Cin, Cout, T, H, W, KT, hplib.stream())
    assert_close(y_pl, want, RTOL, "conv.plain")
    assert_close(y_pl, y_sk, 1e-5, "conv.plain-vs-streamk")


@pytest.fixture
def wino_mode():
    """Switch the Winograd path (hpvg_conv_wino_config) for one test; restored afterwards."""
    from hp_vae_gan_amd import lib as hplib
    lib = hplib.load()
    prev = lib.hpvg_conv_wino_config(-1, -1)

    def set_mode(m):
        assert lib.hpvg_conv_wino_config(m, -1) == m
    yield set_mode
    lib.hpvg_conv_wino_config(prev, -1)


@pytest.mark.parametrize("B,Cin,Cout,sp", [
    (2, 64, 64, (3, 5, 6)), (1, 64, 64, (4, 18, 33)), (1, 64, 128, (2, 5, 4)), (1, 128, 70, (3, 30, 50)), (1, 16, 40, (3, 4, 7)),
    (1, 12, 64, (2, 7, 1)), (1, 64, 64, (2, 9, 130)), (2, 64, 64, (7, 72, 129)), (2, 64, 64, (5, 45, 81)), (1, 64, 64, (1, 1, 1)),
    (2, 64, 64, (9, 10)), (2, 64, 64, (24, 33)), (1, 64, 128, (48, 65)), (2, 64, 64, (200, 300)),
    # even widths run the rows-as-in-memory staging (16-byte LDS-DMA pieces, border factors in the input transform):
    # H*W = 2 (mod 4) exercises the dword piece for the group cut by the plane end; W = 2: every pair is at both borders
    (2, 64, 64, (3, 57, 102)), (1, 64, 64, (7, 91, 162)), (2, 16, 40, (2, 3, 2)), (1, 64, 64, (2, 1, 258)), (2, 64, 64, (45, 82)),
    (1, 64, 64, (3, 400)),   # too wide for one 16-byte group per lane: back to the halo'd bands
])
def test_conv_winograd_kernel_against_direct_kernel_and_oracle(ops, wino_mode, B, Cin, Cout, sp):
    """conv_wino_kernel (F(2,3) along W; every eligible launch forced onto it) against the oracle and against the direct
    kernel on the same inputs: forward, backward-data (flipped pack), LeakyReLU epilogue + 1-bit mask output, masked
    epilogues (1-bit and fp32 masks), stream-K and one-workgroup-per-tile launches; odd and tiny widths, ragged channel
    counts, two output-channel groups, cut tiles finished by conv_wino_fixup_kernel."""
    import ctypes
    from hp_vae_gan_amd import lib as hplib
    nd = len(sp)
    x = _rand(B, Cin, *sp, seed=21)
    w = _rand(Cout, Cin, *([3] * nd), seed=22, scale=0.05)
    b = _rand(Cout, seed=23)
    gy = _rand(B, Cout, *sp, seed=24)
    want = O.conv(x, w, b)
    xr = x.clone().requires_grad_(True)
    (want_dx,) = torch.autograd.grad(O.conv(xr, w, None), xr, gy)
    act = O.leaky_relu(want)
    xd, wd, bd, gyd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    res = {}
    # 3 / 4: every eligible launch on the Winograd kernel with the rows-as-in-memory staging (16-byte pieces; even widths
    # that fit, else it falls back to the bands) / with the halo'd bands; 0: the direct kernel
    for mode in (3, 4, 0):
        wino_mode(mode)
        y = ops.conv_fwd_raw(xd, wd, bd)
        dx = ops.conv_fwd_raw(gyd, wd, None, flip=True)
        ya, bits = ops.conv_fwd_raw(xd, wd, bd, out_lrelu=True, want_bits=True)
        # the layer below's leaky_relu_backward in this layer's backward-data epilogue: mask = sign of x (any tensor of dx's shape)
        xbits_src, xbits = ops.conv_fwd_raw(gyd, wd, None, flip=True, out_lrelu=True, want_bits=True) if Cin > 4 else (None, None)
        dxm_bits = ops.conv_fwd_raw(gyd, wd, None, flip=True, mask_bits=xbits)
        dxm_f32 = ops.conv_fwd_raw(gyd, wd, None, flip=True, out_mask=xd)
        Bq, C, T, H, W = ops.geom(xd)
        KT = 3 if nd == 3 else 1
        wp = ops.pack_weight(wd, False)
        y_pl = torch.full_like(y, float("nan"))
        hplib.call("hpvg_conv_fwd_f32", hplib.ptr(xd), hplib.ptr(wp), hplib.ptr(bd), None, None, 0, hplib.ptr(y_pl), 0, None, None,
                   ctypes.c_size_t(0), Bq, Cin, Cout, T, H, W, KT, hplib.stream())
        res[mode] = dict(y=y, dx=dx, ya=ya, bits=bits, dxm_bits=dxm_bits, dxm_f32=dxm_f32, y_pl=y_pl, xbits_src=xbits_src)
    for mode, r in res.items():
        tag = {3: "wino.rows.", 4: "wino.bands.", 0: "direct."}[mode]
        assert_close(r["y"], want, RTOL, tag + "y")
        assert_close(r["y_pl"], want, RTOL, tag + "y.plain-launch")
        assert_close(r["dx"], want_dx, RTOL, tag + "dx")
        assert_close(r["ya"], act, RTOL, tag + "lrelu")
        assert_close(r["dxm_f32"], want_dx * torch.where(x > 0, 1.0, 0.2), RTOL, tag + "dx.mask_f32")
        # the 1-bit mask this launch read is the sign pattern of its own producer launch
        m = torch.where(r["xbits_src"] > 0, 1.0, 0.2)
        assert_close(r["dxm_bits"], r["dx"] * m, 1e-6, tag + "dx.mask_bits")
    # Winograd against the direct kernel: rounding only (measured ~1e-6 of the output scale)
    far = (want.abs() > 1e-4 * float(want.abs().max())).to(DEV)
    cw = hplib.call("hpvg_conv_mask_words", B, Cout, *((ops.geom(xd))[2:]))
    mt = (Cout + 31) // 32
    for wm in (3, 4):
        for k in ("y", "dx", "ya", "y_pl", "dxm_f32"):
            assert_close(res[wm][k], res[0][k], 2e-5, "wino%d-vs-direct.%s" % (wm, k))
        # the sign bits agree wherever the activation is not within rounding of zero
        sa, sb = res[wm]["ya"] > 0, res[0]["ya"] > 0
        assert bool(((sa == sb) | ~far).all())
        assert res[wm]["bits"].numel() == cw
        # bits written by the Winograd epilogue decode to the sign of its own output
        bw = res[wm]["bits"].view(B, mt, -1).cpu()
        yw = res[wm]["ya"].reshape(B, Cout, -1).cpu()
        for c in (0, Cout // 2, Cout - 1):
            got = (bw[:, c // 32, :] >> (c % 32)) & 1
            assert bool((got.bool() == (yw[:, c] > 0)).all()), "bits of channel %d" % c


@pytest.mark.parametrize("B,Cin,Cout,sp", [
    (1, 64, 64, (2, 4, 6)), (2, 64, 64, (3, 6, 8)), (1, 64, 128, (2, 5, 4)), (1, 128, 70, (3, 30, 50)), (1, 10, 40, (3, 4, 4)),
    (2, 64, 64, (5, 36, 130)), (1, 64, 64, (2, 9, 256)), (1, 12, 64, (2, 2, 2)), (2, 64, 64, (3, 57, 104)), (1, 64, 64, (14, 23, 40)),
    # H*W = 2 (mod 4): the 16-byte group at the end of a plane is patched in by the tile that stages it (stage_tail); with
    # several tiles per plane, tiny planes, the last plane of the tensor (nothing may be read past it) and an odd height
    (1, 64, 64, (3, 57, 102)), (2, 16, 40, (2, 3, 2)), (1, 64, 64, (5, 7, 6)), (2, 64, 64, (2, 91, 162)), (1, 64, 64, (3, 1, 2)),
    # odd W (round 3: the ODD instance - dword LDS reads, a third border factor, one-column last quads, dword mask words):
    # H * W = 1, 3 (mod 4) (the plane-end patch copies 1 or 3 floats) and even H, several tiles per plane, W = 3, ragged channels
    (1, 64, 64, (2, 5, 7)), (2, 64, 64, (3, 36, 65)), (1, 64, 64, (2, 9, 129)), (1, 24, 40, (2, 3, 3)), (1, 64, 64, (3, 7, 5)),
    (2, 64, 64, (2, 5, 9)), (1, 64, 64, (4, 18, 33)),
    # a tall plane (256 000 floats: just below the kernel's zero plane of 2^18): the time planes outside the clip are staged from
    # that zero plane with the tile's own lane offsets - up to 1 MB into it
    (1, 8, 40, (2, 1000, 256)),
])
def test_conv_winograd_two_axis_kernel_against_direct_kernel_and_oracle(ops, wino_mode, B, Cin, Cout, sp):
    """conv_wino2d_kernel (F(2x2, 3x3) over H and W, one workgroup per CU, software-pipelined with hand-counted waits;
    forced wherever it can run: 3x3x3, even W, H*W a multiple of 4) against the oracle and the direct kernel: forward,
    backward-data, LeakyReLU + 1-bit mask output, masked epilogues; odd H (half a quad row), W = 2, two tiles per quad
    row, ragged channel counts (sub-chunks past Cin, an odd number of output-channel tiles), several tiles and sub-chunks per
    workgroup (the pipeline's steady state and its wrap to the next tile)."""
    from hp_vae_gan_amd import lib as hplib
    x = _rand(B, Cin, *sp, seed=41)
    w = _rand(Cout, Cin, 3, 3, 3, seed=42, scale=0.05)
    b = _rand(Cout, seed=43)
    gy = _rand(B, Cout, *sp, seed=44)
    want = O.conv(x, w, b)
    xr = x.clone().requires_grad_(True)
    (want_dx,) = torch.autograd.grad(O.conv(xr, w, None), xr, gy)
    act = O.leaky_relu(want)
    xd, wd, bd, gyd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    res = {}
    for mode in (5, 0):
        wino_mode(mode)
        y = ops.conv_fwd_raw(xd, wd, bd)
        dx = ops.conv_fwd_raw(gyd, wd, None, flip=True)
        ya, bits = ops.conv_fwd_raw(xd, wd, bd, out_lrelu=True, want_bits=True)
        xbits_src, xbits = ops.conv_fwd_raw(gyd, wd, None, flip=True, out_lrelu=True, want_bits=True)
        dxm_bits = ops.conv_fwd_raw(gyd, wd, None, flip=True, mask_bits=xbits)
        dxm_f32 = ops.conv_fwd_raw(gyd, wd, None, flip=True, out_mask=xd)
        res[mode] = dict(y=y, dx=dx, ya=ya, bits=bits, dxm_bits=dxm_bits, dxm_f32=dxm_f32, xbits_src=xbits_src)
    for mode, r in res.items():
        tag = "wino2d." if mode == 5 else "direct."
        assert_close(r["y"], want, RTOL, tag + "y")
        assert_close(r["dx"], want_dx, RTOL, tag + "dx")
        assert_close(r["ya"], act, RTOL, tag + "lrelu")
        assert_close(r["dxm_f32"], want_dx * torch.where(x > 0, 1.0, 0.2), RTOL, tag + "dx.mask_f32")
        m = torch.where(r["xbits_src"] > 0, 1.0, 0.2)
        assert_close(r["dxm_bits"], r["dx"] * m, 1e-6, tag + "dx.mask_bits")
    for k in ("y", "dx", "ya", "dxm_f32"):
        assert_close(res[5][k], res[0][k], 3e-5, "wino2d-vs-direct." + k)
    mt = (Cout + 31) // 32
    bw = res[5]["bits"].view(B, mt, -1).cpu()
    yw = res[5]["ya"].reshape(B, Cout, -1).cpu()
    for c in (0, Cout // 2, Cout - 1):
        got = (bw[:, c // 32, :] >> (c % 32)) & 1
        assert bool((got.bool() == (yw[:, c] > 0)).all()), "bits of channel %d" % c


@pytest.mark.parametrize("B,Cin,Cout,sp", [
    (2, 64, 64, (3, 5, 6)), (1, 64, 64, (4, 18, 33)), (1, 64, 128, (2, 5, 4)), (1, 128, 70, (3, 30, 50)), (1, 16, 24, (3, 4, 7)),
    (1, 8, 8, (1, 1, 1)), (1, 12, 64, (2, 7, 1)), (1, 64, 64, (2, 9, 130)), (2, 64, 64, (7, 72, 129)), (1, 5, 70, (2, 3, 300)),
    (2, 64, 64, (9, 10)), (2, 64, 64, (24, 33)), (1, 64, 128, (48, 65)), (2, 64, 64, (200, 300)),
    # widths that are multiples of 4: the 16-byte staging form (bands cut ragged, one band, rows past the image, channel
    # blocks that are not full, several pieces per row) - mode 3 runs the 4-byte form on the same inputs, mode 4 the 16-byte
    # form on four waves (mode 2: eight)
    (1, 64, 64, (3, 7, 8)), (2, 64, 64, (2, 9, 36)), (1, 70, 130, (2, 5, 52)), (1, 64, 64, (2, 20, 204)), (2, 64, 64, (31, 256)),
    (1, 24, 40, (3, 3, 4)), (1, 64, 64, (1, 50, 100)),
    # (mode 5: the two-axis kernel on every W % 4 == 0 shape above - odd H, H = 1, tiles padded to an even number of K steps,
    # several pieces per row, ragged bands and channel blocks - and on these: one tile exactly, a single row, 2-D one quad row)
    (1, 64, 64, (2, 6, 16)), (2, 64, 64, (3, 1, 4)), (1, 64, 64, (2, 12)), (2, 70, 40, (2, 13, 24)),
    # W = 2 (mod 4) (also (3, 5, 6), (3, 30, 50), (2, 9, 130), (9, 10) above): the group that straddles the right border is
    # patched - a last band of two columns (162 = 10 x 16 + 2: the band before it has the straddler in its halo), W = 2, 2-D
    (1, 64, 64, (2, 13, 162)), (2, 64, 64, (2, 7, 2)), (1, 72, 64, (3, 4, 18)), (2, 64, 64, (21, 38)),
])
def test_conv_weight_gradient_winograd_kernel_against_direct_kernels_and_oracle(ops, B, Cin, Cout, sp):
    """conv_wgradw_kernel (transposed F(2,3) along W, output transform in its reduce kernel; forced for every wide layer)
    against the oracle and against the direct weight-gradient kernels on the same inputs: overwrite and accumulate forms,
    odd / tiny widths, ragged channel counts, several 64-channel blocks, 2-D and 3-D."""
    from hp_vae_gan_amd import lib as hplib
    lib = hplib.load()
    nd = len(sp)
    x = _rand(B, Cin, *sp, seed=31)
    gy = _rand(B, Cout, *sp, seed=32)
    w = _rand(Cout, Cin, *([3] * nd), seed=33, scale=0.1).requires_grad_(True)
    (want,) = torch.autograd.grad(O.conv(x, w, None), w, gy)
    xd, gyd = x.to(DEV), gy.to(DEV)
    base = _rand(*w.shape, seed=34).to(DEV)
    prev = lib.hpvg_conv_bwd_weight_wino_config(-1)
    res = {}
    try:
        want_db = gy.sum(dim=[0] + list(range(2, 2 + nd)))
        bbase = _rand(Cout, seed=35).to(DEV)
        for mode in (2, 3, 4, 5, 0):
            assert lib.hpvg_conv_bwd_weight_wino_config(mode) == mode
            if Cin > 4 and Cout > 4:   # mode 5 = the two-axis kernel on every wide layer
                kind = lib.hpvg_conv_bwd_weight_kernel_kind(B, Cin, Cout, sp[0] if nd == 3 else 1, sp[-2], sp[-1], 3 if nd == 3 else 1)
                assert kind == (3 if mode == 5 else (2 if mode >= 2 else kind)) and (mode or kind in (0, 1))
            dw = ops.conv_bwd_weight_raw(gyd, xd, w.shape)
            acc = base.clone()
            assert ops.conv_bwd_weight_raw(gyd, xd, w.shape, into=acc) is None
            res[mode] = (dw, acc)
            # weight AND bias gradient from one launch (the Winograd kernel's centre-tap workgroups sum dY on the side)
            accw, accb = base.clone(), bbase.clone()
            fused = ops.conv_bwd_weight_bias_raw(gyd, xd, w.shape, accw, accb)
            assert fused == (mode >= 2 and Cin > 4 and Cout > 4)
            if fused:
                assert_close(accw - base, want, RTOL, "wino.fused.dw", atol=1e-5 * float(base.abs().max()))
                assert_close(accb - bbase, want_db, 1e-5, "wino.fused.db", atol=2e-6 * float(bbase.abs().max()))
                assert torch.equal(accw, acc)      # the same weight-gradient launch, bit for bit
    finally:
        lib.hpvg_conv_bwd_weight_wino_config(prev)
    for mode, (dw, acc) in res.items():
        tag = {2: "wino.", 3: "wino4.", 4: "wino16x4waves.", 5: "wino2axis.", 0: "direct."}[mode]
        assert_close(dw, want, RTOL, tag + "dw")
        assert_close(acc - base, want, RTOL, tag + "dw.accumulate", atol=1e-5 * float(base.abs().max()))
    assert_close(res[2][0], res[0][0], 3e-5, "wino-vs-direct.dw")
    assert_close(res[3][0], res[0][0], 3e-5, "wino4-vs-direct.dw")
    assert_close(res[4][0], res[0][0], 3e-5, "wino16x4waves-vs-direct.dw")
    assert_close(res[5][0], res[0][0], 3e-5, "wino2axis-vs-direct.dw")


@pytest.mark.parametrize("B,Cin,Cout,sp", [(2, 3, 64, (4, 6, 7)), (1, 64, 64, (3, 4, 7)), (2, 64, 64, (7, 12))])
def test_conv_lrelu_epilogue_and_affine_prologue(ops, B, Cin, Cout, sp):
    nd = len(sp)
    x = _rand(B, Cin, *sp, seed=5)
    w = _rand(Cout, Cin, *([3] * nd), seed=6, scale=0.1)
    b = _rand(Cout, seed=7)
    sc, sh = _rand(Cin, seed=8).abs() + 0.5, _rand(Cin, seed=9)
    shape = (1, -1) + (1,) * nd
    want = O.leaky_relu(O.conv(O.leaky_relu(x * sc.view(shape) + sh.view(shape)), w, b))
    got = ops.conv_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), out_lrelu=True, in_affine=(sc.to(DEV), sh.to(DEV)), in_lrelu=True)
    assert_close(got, want, RTOL, "conv.fused")
    want2 = O.conv(x * sc.view(shape) + sh.view(shape), w, None)
    got2 = ops.conv_fwd_raw(x.to(DEV), w.to(DEV), None, in_affine=(sc.to(DEV), sh.to(DEV)), in_lrelu=False)
    assert_close(got2, want2, RTOL, "conv.affine_only")


def test_conv_double_backward_closed_set(ops):
    """d/dw and d/dx of <conv_bwd_data(dy, w), g> and of <conv_bwd_weight(dy, x), gw> against torch autograd."""
    x = _rand(2, 8, 3, 5, 6, seed=10).requires_grad_(True)
    w = _rand(12, 8, 3, 3, 3, seed=11, scale=0.2).requires_grad_(True)
    gy = _rand(2, 12, 3, 5, 6, seed=12).requires_grad_(True)
    y = O.conv(x, w)
    dx, dw = torch.autograd.grad(y, [x, w], gy, create_graph=True)
    loss = (dx ** 2).sum() + (dw ** 3).sum()
    want = torch.autograd.grad(loss, [x, w, gy])

    xd, wd, gyd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, gy))
    yd = ops.Conv.apply(xd, wd, None, False)
    dxd, dwd = torch.autograd.grad(yd, [xd, wd], gyd, create_graph=True)
    assert_close(dxd, dx, RTOL, "dx")
    assert_close(dwd, dw, RTOL, "dw")
    lossd = (dxd ** 2).sum() + (dwd ** 3).sum()
    got = torch.autograd.grad(lossd, [xd, wd, gyd])
    for g, r, n in zip(got, want, ("d/dx", "d/dw", "d/dgy")):
        assert_close(g, r, RTOL, "double." + n)


@pytest.mark.parametrize("shape,chans", [((2, 3, 3, 6, 7), (8, 12, 1)), ((1, 3, 9, 11), (64, 64, 3)), ((2, 5, 4, 18, 33), (64, 70, 64))])
def test_activated_conv_chain_with_consumer_side_masks(ops, shape, chans):
    """A chain of conv + LeakyReLU layers (the critic / encoder pattern) whose LeakyReLU backward is applied by the CONSUMER
    conv's backward-data epilogue (ops.Conv in_act / mask_by_consumer, kernel out_mask): first-order gradients and the
    gradient-penalty style double backward against the oracle's plain autograd over the same chain; also covers the
    stream-K fix-up path and the narrow-output kernel (Cout <= 4)."""
    nd = len(shape) - 2
    x = _rand(*shape, seed=60).requires_grad_(True)
    cin = shape[1]
    ws, bs = [], []
    for i, co in enumerate(chans):
        ws.append(_rand(co, cin, *([3] * nd), seed=61 + i, scale=0.3).requires_grad_(True))
        bs.append(_rand(co, seed=71 + i).requires_grad_(True))
        cin = co

    def chain_ref(x):
        h = x
        for i, (w, b) in enumerate(zip(ws, bs)):
            h = O.conv(h, w, b)
            if i < len(ws) - 1:
                h = O.leaky_relu(h)
        return h
    y = chain_ref(x)
    gy = _rand(*y.shape, seed=80)
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    pen = ((gx.norm(2, dim=1) - 1) ** 2).mean()
    want2 = torch.autograd.grad(pen, ws, retain_graph=True)
    want1 = torch.autograd.grad(y, [x] + ws + bs, gy)

    xd = x.detach().to(DEV).requires_grad_(True)
    wd = [w.detach().to(DEV).requires_grad_(True) for w in ws]
    bd = [b.detach().to(DEV).requires_grad_(True) for b in bs]

    def chain_hip(x):
        h = x
        n = len(wd)
        for i, (w, b) in enumerate(zip(wd, bd)):
            h = ops.Conv.apply(h, w, b, i < n - 1, i > 0, i < n - 1)
        return h
    yd = chain_hip(xd)
    assert_close(yd, y, RTOL, "chain.y")
    got1 = torch.autograd.grad(yd, [xd] + wd + bd, gy.to(DEV), retain_graph=True)
    for g, r, n in zip(got1, want1, ["dx"] + ["dw%d" % i for i in range(len(ws))] + ["db%d" % i for i in range(len(bs))]):
        assert_close(g, r, RTOL, "chain." + n, atol=1e-6)
    with ops.inputs_only():
        (gxd,) = torch.autograd.grad(yd, xd, gy.to(DEV), create_graph=True)
    assert_close(gxd, gx, RTOL, "chain.gx")
    pend = ops.GradPenalty.apply(gxd, 1.0)
    assert_close(pend, pen, RTOL, "chain.penalty")
    got2 = torch.autograd.grad(pend, wd)
    for g, r, i in zip(got2, want2, range(len(ws))):
        assert_close(g, r, RTOL, "chain.double.dw%d" % i, atol=1e-7)


@pytest.mark.parametrize("shape", [(2, 8, 3, 5, 7), (2, 64, 4, 18, 33), (1, 16, 37, 41), (2, 64, 1, 1, 3)])
def test_bn_act(ops, shape):
    C = shape[1]
    r = (_rand(*shape, seed=20) * 1.7 + 0.3).requires_grad_(True)
    gamma = (_rand(C, seed=21).abs() + 0.5).requires_grad_(True)
    beta = _rand(C, seed=22).requires_grad_(True)
    rm, rv = _rand(C, seed=23), _rand(C, seed=24).abs() + 0.5
    gh = _rand(*shape, seed=25)
    rm_o, rv_o = rm.clone(), rv.clone()
    h = O.leaky_relu(O.batch_norm_train(r, gamma, beta, rm_o, rv_o))
    dr, dg, db = torch.autograd.grad(h, [r, gamma, beta], gh)

    rd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (r, gamma, beta))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    hd = ops.BNAct.apply(rd, gd, bd, rmd, rvd, 0.1, 1e-5, True)
    assert_close(hd, h, RTOL, "bn.h")
    assert_close(rmd, rm_o, RTOL, "bn.running_mean")
    assert_close(rvd, rv_o, RTOL, "bn.running_var")
    drd, dgd, dbd = torch.autograd.grad(hd, [rd, gd, bd], gh.to(DEV))
    assert_close(drd, dr, RTOL, "bn.dr")
    assert_close(dgd, dg, RTOL, "bn.dgamma")
    assert_close(dbd, db, RTOL, "bn.dbeta")


@pytest.mark.parametrize("shape,lrelu", [((2, 8, 3, 5, 7), True), ((2, 64, 4, 18, 33), True), ((1, 16, 37, 41), False), ((3, 5, 2, 3, 4), True)])
def test_bn_act_double_backward(ops, shape, lrelu):
    """Second-order BatchNorm (+ LeakyReLU): what the gradient penalty of the baselines' BatchNorm critic differentiates
    (networks_3d.py:184-210, modules/utils.py:14-18).  d/dr, d/dgamma and d/d(dh) of a functional of the first-order input
    gradient, against torch autograd over the oracle's written-out BatchNorm."""
    C = shape[1]
    r = (_rand(*shape, seed=120) * 1.3 + 0.2).requires_grad_(True)
    gamma = (_rand(C, seed=121).abs() + 0.5).requires_grad_(True)
    beta = _rand(C, seed=122).requires_grad_(True)
    gh = _rand(*shape, seed=123).requires_grad_(True)
    wgt = _rand(*shape, seed=124)

    def fwd(r, gamma, beta):
        y = O.batch_norm_train(r, gamma, beta, torch.zeros(C), torch.ones(C))
        return O.leaky_relu(y) if lrelu else y
    h = fwd(r, gamma, beta)
    (dr,) = torch.autograd.grad(h, r, gh, create_graph=True)
    loss = (dr * wgt).sum() + 0.5 * (dr ** 2).sum()
    want = torch.autograd.grad(loss, [r, gamma, gh])

    rd, gd, bd, ghd = (t.detach().to(DEV).requires_grad_(True) for t in (r, gamma, beta, gh))
    hd = ops.BNAct.apply(rd, gd, bd, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), 0.1, 1e-5, lrelu)
    assert_close(hd, h, RTOL, "bn2.h")
    (drd,) = torch.autograd.grad(hd, rd, ghd, create_graph=True)
    assert_close(drd, dr, RTOL, "bn2.dr")
    lossd = (drd * wgt.to(DEV)).sum() + 0.5 * (drd ** 2).sum()
    got = torch.autograd.grad(lossd, [rd, gd, ghd])
    for g, w_, n in zip(got, want, ("d/dr", "d/dgamma", "d/d(dh)")):
        assert_close(g, w_, RTOL, "bn2." + n, atol=1e-6)


def test_bn_act_batch_split_equals_whole_batch(ops):
    """Multi-GPU BatchNorm: two 'ranks' hold one sample each; with the per-channel sums exchanged (here: added by hand, as
    the all-reduce would) outputs, running statistics, dr and the summed parameter gradients equal BNAct on the batch."""
    B, C, sp = 2, 24, (3, 7, 9)
    r = _rand(B, C, *sp, seed=31).to(DEV)
    dh = _rand(B, C, *sp, seed=32).to(DEV)
    gamma = (_rand(C, seed=33) * 0.2 + 1).to(DEV)
    beta = (_rand(C, seed=34) * 0.1).to(DEV)
    rm0, rv0 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    g1, b1 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    r1 = r.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    h = ops.BNAct.apply(r1, g1, b1, rm, rv, 0.1, 1e-5, True)
    dr, dg, db = torch.autograd.grad(h, [r1, g1, b1], dh)

    # two ranks executed one after the other.  The "all-reduce" adds the partner's rank-local tensor: the forward pair
    # comes from the raw entry point, the backward pair from a first run of the partner (it only depends on the partner's
    # samples and on the global statistics, which are already right in that run).
    import ctypes
    from hp_vae_gan_amd import lib as hplib
    S = sp[0] * sp[1] * sp[2]
    fwd_local = {}
    for rank in (0, 1):
        x = r[rank:rank + 1].contiguous()
        ws = ops.workspace(hplib.call("hpvg_bn_ws_bytes", C), x.device)
        sums = torch.empty(C, 2, dtype=torch.float64, device=DEV)
        hplib.call("hpvg_bn_sums_f32", hplib.ptr(x), hplib.ptr(sums), hplib.ptr(ws), ctypes.c_size_t(ws.numel()), 1, C,
                   ctypes.c_long(S), hplib.stream())
        fwd_local[rank] = sums

    def run(rank, partner_bwd):
        calls = []

        def allreduce(t):
            calls.append(t.clone())
            t.add_(fwd_local[1 - rank] if len(calls) == 1 else partner_bwd)

        g2, b2 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        r2 = r[rank:rank + 1].clone().requires_grad_(True)
        rm2, rv2 = rm0.clone(), rv0.clone()
        h2 = ops.BNActSync.apply(r2, g2, b2, rm2, rv2, 0.1, 1e-5, True, allreduce, 2)
        grads = torch.autograd.grad(h2, [r2, g2, b2], dh[rank:rank + 1])
        return (h2, rm2, rv2) + grads, calls[1]

    bwd_local = {rank: run(rank, torch.zeros(C, 2, dtype=torch.float64, device=DEV))[1] for rank in (0, 1)}
    outs = {rank: run(rank, bwd_local[1 - rank])[0] for rank in (0, 1)}
    for rank in (0, 1):
        h2, rm2, rv2, dr2, dg2, db2 = outs[rank]
        assert_close(h2, h[rank:rank + 1], 1e-6, "syncbn.h")
        assert_close(rm2, rm, 1e-6, "syncbn.running_mean")
        assert_close(rv2, rv, 1e-6, "syncbn.running_var")
        assert_close(dr2, dr[rank:rank + 1], 1e-5, "syncbn.dr")
    assert_close(outs[0][4] + outs[1][4], dg, 1e-5, "syncbn.dgamma (sum of shares)")
    assert_close(outs[0][5] + outs[1][5], db, 1e-5, "syncbn.dbeta (sum of shares)")


@pytest.mark.parametrize("Co,Ci,nd", [(64, 3, 3), (64, 64, 3), (24, 16, 3), (64, 64, 2), (1, 64, 3)])
def test_spectral_norm_weight(ops, Co, Ci, nd):
    w = _rand(Co, Ci, *([3] * nd), seed=30, scale=0.1).requires_grad_(True)
    u = torch.nn.functional.normalize(_rand(Co, seed=31), dim=0)
    v = torch.nn.functional.normalize(_rand(Ci * 3 ** nd, seed=32), dim=0)
    gw = _rand(*w.shape, seed=33)
    uo, vo = u.clone(), v.clone()
    we = O.spectral_norm_weight(w, uo, vo, True)
    (dw,) = torch.autograd.grad(we, w, gw)
    wd = w.detach().to(DEV).requires_grad_(True)
    ud, vd = u.to(DEV), v.to(DEV)
    wed = ops.SpectralNormWeight.apply(wd, ud, vd, True, 1e-12)
    assert_close(wed, we, RTOL, "sn.w")
    assert_close(ud, uo, RTOL, "sn.u")
    assert_close(vd, vo, RTOL, "sn.v")
    (dwd,) = torch.autograd.grad(wed, wd, gw.to(DEV))
    assert_close(dwd, dw, RTOL, "sn.dw")
    # eval mode: no power iteration
    we2 = O.spectral_norm_weight(w, uo, vo, False)
    wed2 = ops.SpectralNormWeight.apply(wd, ud, vd, False, 1e-12)
    assert_close(wed2, we2, RTOL, "sn.w.eval")


@pytest.mark.parametrize("in_shape,size", [((2, 3, 4, 18, 33), (4, 23, 41)), ((1, 3, 5, 57, 102), (7, 72, 129)),
                                           ((2, 3, 7, 20, 31), (13, 29, 40)), ((2, 3, 24, 33), (30, 41)), ((1, 2, 1, 5, 5), (1, 9, 3)),
                                           ((1, 1, 3, 4, 5), (1, 1, 1)), ((2, 3, 9, 40, 70), (4, 21, 33)),
                                           ((1, 2, 2, 3, 4), (7, 19, 33))])
def test_upsample(ops, in_shape, size):
    x = _rand(*in_shape, seed=40).requires_grad_(True)
    y = O.resize_linear_ac(x, size)
    gy = _rand(*y.shape, seed=41)
    (dx,) = torch.autograd.grad(y, x, gy, retain_graph=True)
    xd = x.detach().to(DEV).requires_grad_(True)
    yd = ops.UpsampleAC.apply(xd, tuple(size), None, 0.0)
    assert_close(yd, y, RTOL, "up.y")
    (dxd,) = torch.autograd.grad(yd, xd, gy.to(DEV))
    assert_close(dxd, dx, RTOL, "up.dx")
    noise = _rand(*y.shape, seed=42)
    yd1, ydn = ops.UpsampleAC.apply(xd, tuple(size), noise.to(DEV), 0.37)
    assert_close(yd1, y, RTOL, "up.y(noisy call)")
    assert_close(ydn, y + 0.37 * noise, RTOL, "up.yn")
    g2 = _rand(*y.shape, seed=43)
    (dxd2,) = torch.autograd.grad([yd1, ydn], xd, [gy.to(DEV), g2.to(DEV)], retain_graph=True)
    (dx2,) = torch.autograd.grad(y, x, gy + g2)
    assert_close(dxd2, dx2, RTOL, "up.dx(two outputs)")
    # the backward is a gather in a fixed order (no float atomics): bitwise reproducible
    (again,) = torch.autograd.grad([yd1, ydn], xd, [gy.to(DEV), g2.to(DEV)])
    assert torch.equal(again, dxd2)


def test_pointwise_and_losses(ops):
    a = _rand(2, 3, 5, 17, 19, seed=50).requires_grad_(True)
    b = _rand(2, 3, 5, 17, 19, seed=51).requires_grad_(True)
    g = _rand(2, 3, 5, 17, 19, seed=52)
    ad, bd = (t.detach().to(DEV).requires_grad_(True) for t in (a, b))
    # tanh(x + res)
    y = torch.tanh(a + b)
    da, db = torch.autograd.grad(y, [a, b], g)
    yd = ops.TanhRes.apply(ad, bd)
    dad, dbd = torch.autograd.grad(yd, [ad, bd], g.to(DEV))
    assert_close(yd, y, RTOL, "tanhres.y"); assert_close(dad, da, RTOL, "tanhres.da"); assert_close(dbd, db, RTOL, "tanhres.db")
    y = torch.tanh(a)
    (da,) = torch.autograd.grad(y, a, g)
    yd = ops.TanhRes.apply(ad, None)
    (dad,) = torch.autograd.grad(yd, ad, g.to(DEV))
    assert_close(yd, y, RTOL, "tanh.y"); assert_close(dad, da, RTOL, "tanh.da")
    # mse
    m = O.mse(a, b)
    da, db = torch.autograd.grad(m, [a, b])
    md = ops.MSE.apply(ad, bd)
    dad, dbd = torch.autograd.grad(md, [ad, bd])
    assert_close(md, m, RTOL, "mse"); assert_close(dad, da, RTOL, "mse.da"); assert_close(dbd, db, RTOL, "mse.db")
    # signed mean
    m = -a.mean() * 1.0
    (da,) = torch.autograd.grad(m, a)
    md = ops.MeanScaled.apply(ad, -1.0)
    (dad,) = torch.autograd.grad(md, ad)
    assert_close(md, m, RTOL, "mean"); assert_close(dad, da, RTOL, "mean.da")
    # KL + reparameterize
    mu = _rand(2, 128, 4, 18, 33, seed=53).requires_grad_(True)
    lv = (_rand(2, 128, 4, 18, 33, seed=54) * 0.3).requires_grad_(True)
    eps = _rand(2, 128, 4, 18, 33, seed=55)
    gz = _rand(2, 128, 4, 18, 33, seed=56)
    mud, lvd = (t.detach().to(DEV).requires_grad_(True) for t in (mu, lv))
    kl = O.kl_criterion(mu, lv)
    dmu, dlv = torch.autograd.grad(kl, [mu, lv])
    kld = ops.KL.apply(mud, lvd)
    dmud, dlvd = torch.autograd.grad(kld, [mud, lvd])
    assert_close(kld, kl, RTOL, "kl"); assert_close(dmud, dmu, RTOL, "kl.dmu"); assert_close(dlvd, dlv, RTOL, "kl.dlv")
    z = eps * torch.exp(0.5 * lv) + mu
    dmu, dlv = torch.autograd.grad(z, [mu, lv], gz)
    zd = ops.Reparam.apply(mud, lvd, eps.to(DEV))
    dmud, dlvd = torch.autograd.grad(zd, [mud, lvd], gz.to(DEV))
    assert_close(zd, z, RTOL, "reparam.z"); assert_close(dmud, dmu, RTOL, "reparam.dmu"); assert_close(dlvd, dlv, RTOL, "reparam.dlv")
    # gradient-penalty norm term and lerp
    gg = _rand(2, 3, 5, 17, 19, seed=57).requires_grad_(True)
    gp = ((gg.norm(2, dim=1) - 1) ** 2).mean() * 0.1
    (dgg,) = torch.autograd.grad(gp, gg)
    ggd = gg.detach().to(DEV).requires_grad_(True)
    gpd = ops.GradPenalty.apply(ggd, 0.1)
    (dggd,) = torch.autograd.grad(gpd, ggd)
    assert_close(gpd, gp, RTOL, "gp"); assert_close(dggd, dgg, RTOL, "gp.dg")
    al = torch.tensor([0.3125])
    assert_close(ops.lerp(ad, bd, al.to(DEV)), 0.3125 * a + (1 - 0.3125) * b, RTOL, "lerp")


def test_adam_and_clip(ops):
    n = 100003
    p, g = _rand(n, seed=60), _rand(n, seed=61) * 3
    total, coef = O.clip_grad_norm([g_ := g.clone()], 5.0)
    gd = g.to(DEV)
    info = torch.zeros(2, device=DEV)
    ops.clip_scale_(gd, ops.sqsum(gd), 5.0, info)
    assert_close(gd, g_, RTOL, "clip.g")
    assert_close(info.cpu(), torch.stack([coef, total]), RTOL, "clip.info")
    pd, md, vd = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    po, state = p.clone(), {}
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    for t in range(1, 4):
        O.adam_step(po, g_ * t, state, 5e-4, 0.5)
        ops.counter_inc_(cnt)
        # steps 1-2 take the count from the host argument, step 3 from device memory (the hipGraph-safe path)
        ops.adam_step_(pd, (gd * t).contiguous(), md, vd, 5e-4, 0.5, 0.999, 1e-8, t if t < 3 else 999, None if t < 3 else cnt)
    assert_close(pd, po, 1e-5, "adam.p")
    assert_close(md, state["m"], RTOL, "adam.m")
    assert_close(vd, state["v"], RTOL, "adam.v")


# ------------------------------------------------------------------------------------------------ golden block fixtures
def _module_for(name):
    from hp_vae_gan_amd.modules import networks_2d as n2, networks_3d as n3, _nets
    cin_cout = [int(t) for t in name.split("_")[1:3]]
    if name.startswith("convblock3dsn"):
        return n3.ConvBlock3DSN(*cin_cout, 3, 1, 1)
    if name.startswith("convblock2dsn"):
        return n2.ConvBlock2DSN(*cin_cout, 3, 1, 1)
    if name.startswith("convblock3d"):
        return n3.ConvBlock3D(*cin_cout, 3, 1, 1, bn=not name.endswith("plain"), act=None if name.endswith("plain") else "lrelu")
    if name.startswith("convblock2d"):
        return n2.ConvBlock2D(*cin_cout, 3, 1, 1)
    if name.startswith("tail3d"):
        return _nets.Conv(3, *cin_cout)
    raise KeyError(name)


@pytest.mark.parametrize("name", ["convblock3d_3_8", "convblock3d_64_64", "convblock3d_128_8", "convblock3d_8_128_plain",
                                  "convblock3dsn_3_64", "convblock3dsn_16_24", "convblock2d_3_64", "convblock2d_64_64",
                                  "convblock2dsn_64_64", "tail3d_64_3", "tail3d_64_1"])
def test_golden_blocks(ops, name):
    fx = load_golden("ops.pt")[name]
    blk = _module_for(name)
    blk.load_state_dict(fx["sd_before"])
    blk.to(DEV)
    x = fx["x"].to(DEV).requires_grad_(True)
    y = blk(x)
    assert_close(y, fx["y"], RTOL, name + ".y")
    params = dict(blk.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), fx["gy"].to(DEV))
    assert_close(grads[0], fx["dx"], RTOL, name + ".dx")
    for (k, _), g in zip(params.items(), grads[1:]):
        assert_close(g, fx["dparams"][k], RTOL, name + ".d" + k, atol=bn_bias_atol("blk." + k, {"blk." + kk: vv for kk, vv in fx["dparams"].items()}, 1e-6))
    sd = blk.state_dict()
    for k, v in fx["sd_after"].items():
        assert_close(sd[k].float(), v.float(), RTOL, name + ".after." + k)


def test_golden_gradient_penalty(ops):
    """calc_gradient_penalty through WDiscriminator3D: value and the second-order gradients on every D parameter."""
    from hp_vae_gan_amd.modules import networks_3d as n3
    from hp_vae_gan_amd.modules.utils import calc_gradient_penalty
    fx = load_golden("ops.pt")["gp3d"]
    opt = opt_from(dict(fx["opt"]))
    D = n3.WDiscriminator3D(opt)
    D.load_state_dict(fx["D_before"])
    D.to(DEV)
    gp = calc_gradient_penalty(D, fx["real"].to(DEV), fx["fake"].to(DEV), 0.1, DEV, alpha=fx["alpha"])
    assert_close(gp, fx["gp"], RTOL, "gp")
    gp.backward()
    for n, p in D.named_parameters():
        ref = fx["grads"][n]
        if ref is None or float(ref.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) <= 1e-9, n
        else:
            assert_close(p.grad, ref, RTOL, "gp.grad." + n)
    sd = D.state_dict()
    for k, v in fx["D_after"].items():
        if k.endswith(("weight_u", "weight_v")):
            assert_close(sd[k], v, RTOL, "gp.after." + k)


def test_cpu_tensor_fails_loudly(ops):
    with pytest.raises(RuntimeError):
        ops.Conv.apply(torch.zeros(1, 3, 4, 4), torch.zeros(8, 3, 3, 3), None, False)


def test_packed_weight_cache_follows_the_weight(ops):
    """ops.pack_weight keeps the MFMA-order copy of a weight between optimizer steps: a second launch with the same
    weight reuses it, an in-place torch update (version counter), an Adam step (raw kernel -> weights_changed) and a new
    tensor at a recycled address all get a fresh pack."""
    x = _rand(1, 64, 3, 6, 7, seed=1).to(DEV)
    w = _rand(64, 64, 3, 3, 3, seed=2, scale=0.05).to(DEV)
    ops.weights_changed()
    y0 = ops.conv_fwd_raw(x, w, None)
    wp0 = ops.pack_weight(w, False)
    assert ops.pack_weight(w, False) is wp0 and ops.pack_weight(w, True) is not wp0
    assert_close(y0, O.conv(x.cpu(), w.cpu()), RTOL, "cache.first")
    w.mul_(2.0)                                                     # torch in-place: version moves
    assert ops.pack_weight(w, False) is not wp0
    assert_close(ops.conv_fwd_raw(x, w, None), O.conv(x.cpu(), w.cpu()), RTOL, "cache.after-inplace")
    g, m, v = torch.ones_like(w), torch.zeros_like(w), torch.zeros_like(w)
    ops.adam_step_(w.view(-1), g.view(-1), m.view(-1), v.view(-1), 0.05, 0.5, 0.999, 1e-8, 1)   # raw kernel: no version bump
    assert_close(ops.conv_fwd_raw(x, w, None), O.conv(x.cpu(), w.cpu()), RTOL, "cache.after-adam")
    addr = w.data_ptr()
    del w, wp0
    w2 = _rand(64, 64, 3, 3, 3, seed=3, scale=0.05).to(DEV)         # may or may not land on the old address
    assert_close(ops.conv_fwd_raw(x, w2, None), O.conv(x.cpu(), w2.cpu()), RTOL, "cache.new-tensor(%s)" % (w2.data_ptr() == addr))


def test_full_size_conv_family_properties(ops):
    """BASELINE configs[2] finest level (B=2, 64->64, 13 x 144 x 256: 958 464 voxels), where the oracle takes minutes: the
    three conv kernels are tied together by size-independent properties instead -
      * locality: any crop of the output equals the oracle's conv of the matching input crop (+1 voxel of context);
      * adjointness: <conv(x, w), dy> = <x, bwd_data(dy, w)> = <w, bwd_weight(dy, x)>  (one number, three kernels);
      * linearity of the forward kernel in its input."""
    B, C, T, H, W = 2, 64, 13, 144, 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, C, T, H, W, device=DEV, generator=g)
    dy = torch.randn(B, C, T, H, W, device=DEV, generator=g)
    w = _rand(C, C, 3, 3, 3, seed=21, scale=0.03).to(DEV)
    b = _rand(C, seed=22).to(DEV)
    y = ops.conv_fwd_raw(x, w, b)
    # locality: three crops (a corner with zero padding on three sides, an interior block, the far corner)
    for (t0, t1, h0, h1, w0, w1) in [(0, 3, 0, 6, 0, 9), (5, 8, 70, 77, 120, 131), (10, 13, 138, 144, 247, 256)]:
        ts, hs, ws_ = max(t0 - 1, 0), max(h0 - 1, 0), max(w0 - 1, 0)
        te, he, we = min(t1 + 1, T), min(h1 + 1, H), min(w1 + 1, W)
        want = O.conv(x[1:2, :, ts:te, hs:he, ws_:we].cpu(), w.cpu(), b.cpu())
        want = want[:, :, t0 - ts:t0 - ts + (t1 - t0), h0 - hs:h0 - hs + (h1 - h0), w0 - ws_:w0 - ws_ + (w1 - w0)]
        # a crop edge that is not an image edge saw real neighbours in the full conv, zero padding in the cropped one:
        # the +1 context above makes every compared voxel see identical inputs
        assert_close(y[1:2, :, t0:t1, h0:h1, w0:w1], want, RTOL, "fullsize.crop(%d,%d,%d)" % (t0, h0, w0))
    # adjointness (bias-free)
    y0 = ops.conv_fwd_raw(x, w, None)
    dx = ops.conv_fwd_raw(dy, w, None, flip=True)
    dw = ops.conv_bwd_weight_raw(dy, x, w.shape)
    a1 = float((y0.double() * dy.double()).sum())
    a2 = float((x.double() * dx.double()).sum())
    a3 = float((w.double() * dw.double()).sum())
    scale = float(y0.double().norm() * dy.double().norm())
    assert abs(a1 - a2) <= 1e-5 * scale and abs(a1 - a3) <= 1e-5 * scale, (a1, a2, a3, scale)
    # linearity
    x2 = torch.randn(B, C, T, H, W, device=DEV, generator=g)
    lhs = ops.conv_fwd_raw(1.5 * x + x2, w, None)
    rhs = 1.5 * y0 + ops.conv_fwd_raw(x2, w, None)
    assert_close(lhs, rhs, 1e-4, "fullsize.linearity")


def test_full_size_weight_gradient_by_supports_and_kernels(ops):
    """The weight gradient at the BASELINE configs[2] finest level (B=2, 64->64, 13 x 144 x 256), where the oracle takes minutes:
      * a dY that is zero outside a small block gives the oracle's weight (and bias) gradient of the matching crop of X (+1 voxel
        of context) - three supports: the near corner (zero padding on three sides), an interior block that straddles tile and
        band borders, the far corner (the last 16-byte groups of the tensor);
      * on dense inputs the Winograd kernel in each of its forms (16-byte staging on eight and on four waves, 4-byte staging)
        agrees with the direct kernels element by element, and the fused bias gradient with a plain sum."""
    from hp_vae_gan_amd import lib as hplib
    lib = hplib.load()
    B, C, T, H, W = 2, 64, 13, 144, 256
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(B, C, T, H, W, device=DEV, generator=g)
    wshape = (C, C, 3, 3, 3)
    prev = lib.hpvg_conv_bwd_weight_wino_config(-1)
    try:
        for smode in (2, 5):    # the one-axis and the two-axis Winograd kernel (5: conv_wgradw2_kernel, W = 256)
            assert lib.hpvg_conv_bwd_weight_wino_config(smode) == smode
            assert lib.hpvg_conv_bwd_weight_kernel_kind(B, C, C, T, H, W, 3) == (3 if smode == 5 else 2)
            for (t0, t1, h0, h1, w0, w1) in [(0, 2, 0, 5, 0, 7), (5, 8, 68, 75, 125, 135), (11, 13, 139, 144, 249, 256)]:
                dy = torch.zeros(B, C, T, H, W, device=DEV)
                blk = torch.randn(C, t1 - t0, h1 - h0, w1 - w0, device=DEV, generator=g)
                dy[1, :, t0:t1, h0:h1, w0:w1] = blk
                dw, db = torch.zeros(wshape, device=DEV), torch.zeros(C, device=DEV)
                assert ops.conv_bwd_weight_bias_raw(dy, x, wshape, dw, db)
                ts, hs, ws_ = max(t0 - 1, 0), max(h0 - 1, 0), max(w0 - 1, 0)
                te, he, we = min(t1 + 1, T), min(h1 + 1, H), min(w1 + 1, W)
                xc = x[1:2, :, ts:te, hs:he, ws_:we].cpu()
                dyc = torch.zeros(1, C, te - ts, he - hs, we - ws_)
                dyc[0, :, t0 - ts:t1 - ts, h0 - hs:h1 - hs, w0 - ws_:w1 - ws_] = blk.cpu()
                wz = torch.zeros(wshape, requires_grad=True)
                (want,) = torch.autograd.grad(O.conv(xc, wz, None), wz, dyc)
                assert_close(dw, want, RTOL, "fullsize.wgrad.mode%d.support(%d,%d,%d)" % (smode, t0, h0, w0))
                assert_close(db, blk.cpu().sum(dim=(1, 2, 3)), 1e-5, "fullsize.bgrad.mode%d.support(%d,%d,%d)" % (smode, t0, h0, w0))
        dy = torch.randn(B, C, T, H, W, device=DEV, generator=g)
        res = {}
        for mode in (2, 4, 3, 5, 0):
            assert lib.hpvg_conv_bwd_weight_wino_config(mode) == mode
            res[mode] = ops.conv_bwd_weight_raw(dy, x, wshape)
        for mode in (2, 4, 3, 5):
            assert_close(res[mode], res[0], 3e-5, "fullsize.wgrad.mode%d-vs-direct" % mode)
        for mode in (2, 5):
            assert lib.hpvg_conv_bwd_weight_wino_config(mode) == mode
            dw, db = torch.zeros(wshape, device=DEV), torch.zeros(C, device=DEV)
            assert ops.conv_bwd_weight_bias_raw(dy, x, wshape, dw, db)
            assert torch.equal(dw, res[mode])
            assert_close(db, dy.double().sum(dim=(0, 2, 3, 4)).float(), 1e-5, "fullsize.bgrad", atol=2e-2)
    finally:
        lib.hpvg_conv_bwd_weight_wino_config(prev)


def test_full_size_batchnorm_properties(ops):
    """Finest-level BatchNorm + LeakyReLU (B=2, 64 ch, 13 x 144 x 256): before the activation the output has per-channel
    mean beta and variance gamma^2 (eps-corrected), whatever the input; dbeta / dgamma equal the direct sums."""
    B, C, T, H, W = 2, 64, 13, 144, 256
    g = torch.Generator(device=DEV).manual_seed(6)
    r = (torch.randn(B, C, T, H, W, device=DEV, generator=g) * 3.0 + 0.7).requires_grad_(True)
    gamma = (torch.rand(C, device=DEV, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, device=DEV, generator=g).requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    z = ops.BNAct.apply(r, gamma, beta, rm, rv, 0.1, 1e-5, False)
    zm = z.double().mean(dim=(0, 2, 3, 4))
    zv = z.double().var(dim=(0, 2, 3, 4), unbiased=False)
    assert_close(zm.float(), beta.detach(), 1e-4, "fullsize.bn.mean", atol=1e-5)
    var_in = r.detach().double().var(dim=(0, 2, 3, 4), unbiased=False)
    assert_close(zv.float(), (gamma.detach().double() ** 2 * var_in / (var_in + 1e-5)).float(), 1e-4, "fullsize.bn.var")
    dh = torch.randn(B, C, T, H, W, device=DEV, generator=g)
    z.backward(dh)
    assert_close(beta.grad, dh.double().sum(dim=(0, 2, 3, 4)).float(), 1e-4, "fullsize.bn.dbeta", atol=1e-2)
    xhat = (r.detach().double() - r.detach().double().mean(dim=(0, 2, 3, 4), keepdim=True)) / torch.sqrt(var_in + 1e-5).view(1, -1, 1, 1, 1)
    assert_close(gamma.grad, (dh.double() * xhat).sum(dim=(0, 2, 3, 4)).float(), 1e-4, "fullsize.bn.dgamma", atol=1e-2)
    # dr sums to zero per channel and is orthogonal to xhat (the two projections BatchNorm's backward removes)
    dr = r.grad.double()
    n = B * T * H * W
    assert float(dr.sum(dim=(0, 2, 3, 4)).abs().max()) <= 1e-6 * n
    assert float((dr * xhat).sum(dim=(0, 2, 3, 4)).abs().max()) <= 1e-6 * n


@pytest.mark.parametrize("Cin,Cout", [(64, 3), (3, 64), (64, 1), (128, 64)])
def test_full_size_head_and_tail_convs_by_crops(ops, Cin, Cout):
    """The other layer shapes of the path at the finest level's size (tails 64->3 / 64->1 on the narrow-output kernel,
    head 3->64, decoder head 128->64 at its own largest size): output crops against the oracle, forward and backward-data."""
    B, T, H, W = (2, 13, 144, 256) if Cin != 128 else (2, 4, 18, 33)
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(B, Cin, T, H, W, device=DEV, generator=g)
    dy = torch.randn(B, Cout, T, H, W, device=DEV, generator=g)
    w = _rand(Cout, Cin, 3, 3, 3, seed=23, scale=0.05).to(DEV)
    b = _rand(Cout, seed=24).to(DEV)
    y = ops.conv_fwd_raw(x, w, b)
    dx = ops.conv_fwd_raw(dy, w, None, flip=True)
    wf = w.cpu().flip(2, 3, 4).transpose(0, 1).contiguous()     # backward-data = conv with the flipped, transposed weight
    crops = [(0, 3, 0, 6, 0, 9), (T - 3, T, H - 6, H, W - 9, W)] + ([(5, 8, 70, 77, 120, 131)] if T > 8 else [])
    for (t0, t1, h0, h1, w0, w1) in crops:
        ts, hs, ws_ = max(t0 - 1, 0), max(h0 - 1, 0), max(w0 - 1, 0)
        te, he, we = min(t1 + 1, T), min(h1 + 1, H), min(w1 + 1, W)
        sl = (slice(0, 1), slice(None), slice(ts, te), slice(hs, he), slice(ws_, we))
        inner = (slice(None), slice(None), slice(t0 - ts, t0 - ts + t1 - t0), slice(h0 - hs, h0 - hs + h1 - h0),
                 slice(w0 - ws_, w0 - ws_ + w1 - w0))
        assert_close(y[0:1, :, t0:t1, h0:h1, w0:w1], O.conv(x[sl].cpu(), w.cpu(), b.cpu())[inner], RTOL, "crop.fwd")
        assert_close(dx[0:1, :, t0:t1, h0:h1, w0:w1], O.conv(dy[sl].cpu(), wf)[inner], RTOL, "crop.bwd_data")


def test_device_noise_kernel(ops):
    """hpvg_normal_f32 (Philox4x32-10 + Box-Muller): N(0,1) moments, no serial correlation, reproducible per
    (seed, iteration, call), different across calls / iterations / seeds; and the resize kernel that generates the level
    noise itself adds exactly that stream (networks_3d.py:395-400), to the rand half only when asked."""
    import hp_vae_gan_amd.utils as hu
    dev = torch.device(DEV)
    torch.manual_seed(1234)
    ops.rng_next_iteration(dev)
    a = hu.generate_noise(size=[2, 3, 13, 144, 256], device=dev)          # 2.9 M values
    x = a.double().flatten()
    n = x.numel()
    assert abs(float(x.mean())) < 4.0 / n ** 0.5
    assert abs(float(x.var()) - 1.0) < 5.0 * (2.0 / n) ** 0.5
    assert abs(float((x ** 3).mean())) < 5.0 * (15.0 / n) ** 0.5
    assert abs(float((x ** 4).mean()) - 3.0) < 5.0 * (96.0 / n) ** 0.5
    assert abs(float((x[:-1] * x[1:]).mean())) < 4.0 / n ** 0.5 and abs(float((x[:-4] * x[4:]).mean())) < 4.0 / n ** 0.5
    assert float(x.abs().max()) < 7.0 and bool(torch.isfinite(x).all())
    b = hu.generate_noise(ref=a)                                            # next call of the same iteration
    assert abs(float((a.double() * b.double()).mean())) < 4.0 / n ** 0.5 and not torch.equal(a, b)
    # same (seed, iteration, call) -> same stream; another iteration -> another stream
    st = ops._rng(dev)
    it0 = int(st.iter_dev.item())
    st.call = 0
    a2 = hu.generate_noise(ref=a)
    assert torch.equal(a2, a)
    ops.rng_next_iteration(dev)
    assert int(st.iter_dev.item()) == it0 + 1 and st.call == 0
    c = hu.generate_noise(ref=a)
    assert not torch.equal(c, a) and abs(float((a.double() * c.double()).mean())) < 4.0 / n ** 0.5
    torch.manual_seed(99)
    st.call = 0
    d = hu.generate_noise(ref=a)
    assert not torch.equal(d, c)
    # fused: resize + in-kernel noise == resize, then + amp * (the same stream written out)
    xin = _rand(4, 3, 4, 18, 33, seed=7).to(DEV)
    size = (4, 23, 41)
    st.call = 5
    up, upn = ops.UpsampleACNoise.apply(xin, size, 0.37, 2)
    st.call = 5
    nz = hu.generate_noise(ref=up)
    plain = ops.UpsampleAC.apply(xin, size, None, 0.0)
    assert_close(up, plain, 1e-6, "fused resize")                          # (two kernels: fma contraction may differ in the last bit)
    assert torch.equal(upn[:2], up[:2])                                     # the rec half of a merged pass gets no noise
    assert_close(upn[2:], plain[2:] + 0.37 * nz[2:], 1e-6, "fused level noise")


def test_wgrad_all_taps_kernel_on_small_shapes():
    """conv_wgrad3_kernel (all 27 taps of a 64 x 32 channel block in one workgroup, sliding window over t) is picked by size
    (>= 64 tiles per workgroup: the BASELINE stage-9 shapes, covered by test_full_size_conv_family_properties); here a child
    process forces it (HPVG_WGRAD3=2, read once at library load) onto odd small shapes against the oracle."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, HPVG_WGRAD3="2")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "wgrad3_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "wgrad3 ok" in r.stdout


@pytest.mark.gpu
def test_two_axis_weight_gradient_time_major_tile_walk():
    """conv_wgradw2_kernel walks its tiles plane-major by default; HPVG_WG2_ORDER=0 (read once per process) selects the time-major
    walk, whose tile coordinates advance with another carry chain.  A child process with that switch: several tiles per
    workgroup in every direction (bands x band rows x planes x samples), against the direct kernel."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import hp_vae_gan_amd
from hp_vae_gan_amd import ops, lib as hplib
lib = hplib.load()
torch.manual_seed(5)
for B, C, sp in ((2, 64, (9, 40, 70)), (3, 64, (5, 33, 52)), (2, 64, (48, 160))):
    x = torch.randn(B, C, *sp, device="cuda"); dy = torch.randn(B, C, *sp, device="cuda")
    ws = (C, C) + (3,) * len(sp)
    out = {}
    for mode in (0, 5):
        lib.hpvg_conv_bwd_weight_wino_config(mode)
        out[mode] = ops.conv_bwd_weight_raw(dy, x, ws)
    err = float((out[5] - out[0]).abs().max() / out[0].abs().max())
    kind = lib.hpvg_conv_bwd_weight_kernel_kind(B, C, C, sp[0] if len(sp) == 3 else 1, sp[-2], sp[-1], 3 if len(sp) == 3 else 1)
    print("shape", B, sp, "kind", kind, "rel err", err)
    assert kind == 3 and err < 2e-5, (B, sp, kind, err)
print("time-major ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HPVG_WG2_ORDER="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "time-major ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_first_generation_two_axis_conv_kernel_still_matches():
    """HPVG_WINO2R=0 (read once per process) puts the two-axis convs back on round 2's conv_wino2d_kernel (every wave transforms
    whole patches; even widths only), which shares the weight fragments, the epilogue's position-major 1-bit mask words and the
    launch code with conv_wino2r_kernel.  A child process with that switch: forward, LeakyReLU + bit words, bit-masked and
    float-masked backward-data against the direct kernel."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import hp_vae_gan_amd
from hp_vae_gan_amd import ops, lib as hplib
lib = hplib.load()
torch.manual_seed(6)
for B, Cin, Cout, sp in ((2, 64, 64, (3, 36, 64)), (1, 24, 70, (2, 9, 130)), (1, 64, 64, (5, 57, 102))):
    x = torch.randn(B, Cin, *sp, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda"); gy = torch.randn(B, Cout, *sp, device="cuda")
    res = {}
    for mode in (5, 0):
        lib.hpvg_conv_wino_config(mode, -1)
        y = ops.conv_fwd_raw(x, w, b)
        ya, bits = ops.conv_fwd_raw(x, w, b, out_lrelu=True, want_bits=True)
        src, xbits = ops.conv_fwd_raw(gy, w, None, flip=True, out_lrelu=True, want_bits=True)
        dxm = ops.conv_fwd_raw(gy, w, None, flip=True, mask_bits=xbits)
        dxf = ops.conv_fwd_raw(gy, w, None, flip=True, out_mask=x)
        res[mode] = (y, ya, dxm, dxf, src)
    for k in range(4):
        a, r = res[5][k], res[0][k]
        err = float((a - r).abs().max() / r.abs().max())
        print("shape", B, Cin, Cout, sp, "output", k, "rel err", err)
        assert err < 3e-5, (sp, k, err)
    # the bit-masked result is the plain backward-data times the mask its own producer launch wrote
    m = torch.where(res[5][4] > 0, 1.0, 0.2)
    plain = ops.conv_fwd_raw(gy, w, None, flip=True)
    lib.hpvg_conv_wino_config(5, -1)
    plain5 = ops.conv_fwd_raw(gy, w, None, flip=True)
    assert float((res[5][2] - plain5 * m).abs().max() / plain5.abs().max()) < 1e-6
print("first generation ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HPVG_WINO2R="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "first generation ok" in r.stdout, r.stdout + r.stderr
