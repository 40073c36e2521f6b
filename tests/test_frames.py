"""Data front-end: the numpy restatement against hand-computed cv2.INTER_LINEAR answers (CPU)
and the HIP kernel / dataset classes against the restatement (GPU).  cv2 is absent from the image: parity with the
reference's own resizer stays UNPINNED; the uint8 path of both sides is OpenCV's published integer algorithm (11-bit
fixed-point weights), so HIP == restatement is tested with torch.equal - no one-level tolerance."""
import types

import numpy as np
import pytest
import torch

from oracle import frames as OF


def test_restatement_known_answers():
    # cv2.resize(np.array([[0, 100]], np.uint8), (4, 1), interpolation=cv2.INTER_LINEAR) -> [[0, 25, 75, 100]]
    img = np.array([[[0, 0, 0], [100, 100, 100]]], dtype=np.uint8)
    assert OF.resize_linear_cv(img, 1, 4)[0, :, 0].tolist() == [0.0, 25.0, 75.0, 100.0]
    # 2x decimation samples the midpoints: (0+10)/2, (20+30)/2
    img = np.array([[[0] * 3, [10] * 3, [20] * 3, [30] * 3]], dtype=np.uint8)
    assert OF.resize_linear_cv(img, 1, 2)[0, :, 0].tolist() == [5.0, 25.0]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(7, 9, 3), dtype=np.uint8)
    assert np.array_equal(OF.resize_linear_cv(img, 7, 9), img.astype(np.float64))            # identity
    const = np.full((5, 6, 3), 77, dtype=np.uint8)
    assert np.all(OF.resize_linear_cv(const, 11, 4) == 77.0)                                  # partition of unity
    ramp = np.tile((np.arange(16, dtype=np.uint8) * 8)[None, :, None], (4, 1, 3))
    up = OF.resize_linear_cv(ramp, 4, 32, quantize=False)[0, :, 0]
    want = (np.arange(32) + 0.5) * 0.5 - 0.5
    assert np.allclose(up[1:-1], 8 * want[1:-1])                                             # linear ramps are reproduced
    clip = OF.clip_tensor(np.stack([img, img[::-1]]), 0, 1, 2, 7, 9, hflip=True)
    assert clip.shape == (3, 2, 7, 9) and clip.min() >= -1 and clip.max() <= 1
    assert np.allclose(clip[:, 0], np.transpose((img[:, ::-1] / 255.0 - 0.5) / 0.5, (2, 0, 1)), atol=2e-7)   # (float32 pipeline)


def test_restatement_is_fixed_point_where_that_differs_from_exact_bilinear():
    """[0, 255] stretched to five samples, worked by hand with 11-bit weights: x = 3 sits at f = 0.9 -> a1 = rint(1843.2) = 1843,
    S = 255 * 1843 = 469965, ((2048 * (S >> 4)) >> 16) = 917, (917 + 2) >> 2 = 229 - the exact bilinear value is 229.5, which
    round-half-up (the float path) puts at 230.  x = 1: a1 = rint(204.8) = 205 -> 26 (exact 25.5 -> 26: equal)."""
    img = np.array([[[0, 0, 0], [255, 255, 255]]], dtype=np.uint8)
    got = OF.resize_linear_cv(img, 1, 5)[0, :, 0].tolist()
    assert got == [0.0, 26.0, 128.0, 229.0, 255.0]
    exact = np.floor(OF.resize_linear_cv(img, 1, 5, quantize=False)[0, :, 0] + 0.5).tolist()
    assert exact == [0.0, 26.0, 128.0, 230.0, 255.0]
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    a = OF.resize_linear_cv(img, 23, 41)
    b = np.floor(OF.resize_linear_cv(img, 23, 41, quantize=False) + 0.5)
    assert np.abs(a - b).max() == 1.0 and 0.02 < (a != b).mean() < 0.3     # one level apart on a tenth of the pixels
    assert a.min() >= 0 and a.max() <= 255
    # rows are clamped with their weights kept, columns by zeroing the fraction: a one-pixel image stays itself at any size
    one = np.array([[[9, 99, 199]]], dtype=np.uint8)
    assert np.array_equal(OF.resize_linear_cv(one, 4, 3), np.tile(one.astype(np.float64), (4, 3, 1)))


def test_load_frames_and_stage_size(tmp_path):
    import hp_vae_gan_amd  # noqa: F401
    from hp_vae_gan_amd import datasets as D
    arr = np.random.default_rng(1).integers(0, 256, size=(5, 12, 16, 3), dtype=np.uint8)
    np.save(tmp_path / "clip.npy", arr)
    assert np.array_equal(D.load_frames(str(tmp_path / "clip.npy")), arr)
    from PIL import Image
    d = tmp_path / "frames"
    d.mkdir()
    for i in range(3):
        Image.fromarray(arr[i]).save(d / ("f%03d.png" % i))
    assert np.array_equal(D.load_frames(str(d)), arr[:3])
    with pytest.raises(NotImplementedError):
        D.load_frames(str(tmp_path / "clip.mp4"))
    import json, os
    tab = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tables.json")))[0]  # reference-generated geometry
    opt = types.SimpleNamespace(scale_factor=tab["scale_factor"], stop_scale=tab["stop_scale"], img_size=tab["img_size"], ar=tab["ar"])
    for lv in tab["levels"]:
        assert D._stage_size(opt, lv["index"]) == [lv["h"], lv["w"]]


def _check(got, want, quantize):
    if not quantize:
        assert np.abs(got.double().cpu().numpy() - want).max() < 1e-5  # fp32 tap weights: ~1e-3 of a uint8 level
        return
    # the uint8 path is integer arithmetic on both sides (and the same float32 operations after it): bit for bit
    assert want.dtype == np.float32
    assert torch.equal(got.cpu(), torch.from_numpy(want))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(18, 33), (23, 41), (37, 53), (50, 70), (1, 1), (74, 106), (111, 160), (36, 53), (9, 13)])
@pytest.mark.parametrize("quantize", [True, False])
def test_hip_clip_matches_restatement(h, w, quantize):
    import hp_vae_gan_amd  # noqa: F401
    from hp_vae_gan_amd import datasets as D
    frames = np.random.default_rng(2).integers(0, 256, size=(9, 37, 53, 3), dtype=np.uint8)
    store = D._DeviceFrames(frames, "cuda")
    for first, step, count, hflip in [(0, 1, 9, False), (2, 3, 3, True), (8, 1, 1, False)]:
        got = store.clip(first, step, count, h, w, hflip, quantize)
        _check(got, OF.clip_tensor(frames, first, step, count, h, w, hflip, quantize), quantize)
    from hp_vae_gan_amd import lib as hplib
    out = torch.empty(3, 2, h, w, device="cuda")
    with pytest.raises(RuntimeError):  # window runs past the clip
        hplib.call("hpvg_frames_resize_norm_u8_f32", hplib.ptr(store.dev), hplib.ptr(out), 9, 37, 53, 8, 1, 2, h, w, 0, 1, hplib.stream())


@pytest.mark.gpu
def test_hip_video_and_image_datasets():
    import hp_vae_gan_amd  # noqa: F401
    from hp_vae_gan_amd import datasets as D
    from hp_vae_gan_amd import utils as hu
    frames = np.random.default_rng(3).integers(0, 256, size=(13, 72, 128, 3), dtype=np.uint8)
    opt = types.SimpleNamespace(frames=frames, sampling_rates=[4, 3, 2, 1], scale_factor_init=0.75, img_size=64, min_size=16,
                                max_size=64, hflip=False, data_rep=2, scale_idx=0, fps_index=0, device="cuda", max_frames=13)
    hu.adjust_scales2image(opt.img_size, opt)  # as the trainer does before it builds the dataset (train_video.py:339)
    ds = D.SingleVideoDataset(opt)
    ds.generate_frames(0)
    assert opt.fps_lcm == 12 and opt.ar == 72 / 128 and len(ds) == (13 - 12) * 2
    x = ds[1]  # idx wraps: 1 % 1 == 0; every = 4 -> frames 0, 4, 8, 12
    h0, w0 = D._stage_size(opt, 0)
    assert tuple(x.shape) == (3, 4, h0, w0)
    _check(x, OF.clip_tensor(frames, 0, 4, 4, h0, w0), True)
    opt.scale_idx, opt.fps_index = 2, 2
    ds.generate_frames(2)
    xs, x0 = ds[0]  # every = 2 -> 7 frames at stage 2; stage-0 clip keeps sampling_rates[0]
    h2, w2 = D._stage_size(opt, 2)
    assert tuple(xs.shape) == (3, 7, h2, w2) and tuple(x0.shape) == (3, 4, h0, w0)
    _check(xs, OF.clip_tensor(frames, 0, 2, 7, h2, w2), True)
    iopt = types.SimpleNamespace(frames=frames[:1], scale_factor=opt.scale_factor, stop_scale=opt.stop_scale, img_size=64,
                                 hflip=False, scale_idx=1, device="cuda")
    ids = D.SingleImageDataset(iopt)
    a, b = ids[0]
    h1, w1 = D._stage_size(iopt, 1)
    assert tuple(a.shape) == (3, h1, w1) and tuple(b.shape) == (3, h0, w0)
    _check(a, OF.clip_tensor(frames, 0, 1, 1, h1, w1)[:, 0], True)
