"""CPU restatement (numpy, float64) of the data front-end: TEST INFRASTRUCTURE ONLY (tests/, smoke, bench baseline).

Follows the reference: datasets/generate_frames.py:44-46 (cv2.resize INTER_LINEAR per frame), datasets/video.py:53-82
(window frames[idx:idx+fps_lcm+1:every], /255, hflip, normalize(0.5, 0.5), permute C,T,H,W), datasets/image.py:20-49.
cv2 / kornia are not installed in the build image, so this restatement is NOT pinned against the reference's own
resizer.  PARITY UNPINNED for this module.  The quantised path (what cv2.resize returns for uint8
frames) restates the published algorithm of the third-party dependency - OpenCV's imgproc resize, INTER_LINEAR on 8-bit images
(the reference pins no OpenCV version: env.sh installs `opencv` from conda) - in INTEGER arithmetic, so that the HIP kernel can
be held to it bit for bit: 11-bit fixed-point tap weights (INTER_RESIZE_COEF_SCALE = 2048, rounded half-to-even from float32
fractions), a horizontal pass into 32-bit integers, a vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
import numpy as np


def _cv_taps(D, S):
    """Tap positions and float32 fractions of cv2's linear resize along one axis (resizeGeneric setup): scale = 1 / (D / S) in
    double, f = float32((d + 0.5) * scale - 0.5), s = floor(f), fraction = f - s in float32."""
    scale = 1.0 / (float(D) / float(S))
    f = ((np.arange(D, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    fr = (f - s0.astype(np.float32)).astype(np.float32)
    return s0, fr


def _cv_coef(fr):
    """(1 - f, f) * 2048 rounded half-to-even to short, as saturate_cast<short>(float) does (cvRound)."""
    one = np.float32(1.0)
    c0 = np.rint(((one - fr) * np.float32(2048.0)).astype(np.float32)).astype(np.int64)
    c1 = np.rint((fr * np.float32(2048.0)).astype(np.float32)).astype(np.int64)
    return c0, c1


def resize_linear_cv_u8(img, h, w):
    """img [H, W, C] uint8 -> [h, w, C] int64 uint8 levels: OpenCV's 8-bit INTER_LINEAR in integer arithmetic."""
    H, W = img.shape[:2]
    src = img.astype(np.int64)
    sx, fx = _cv_taps(w, W)
    lo = sx < 0                      # x border: the fraction is zeroed (resizeGeneric: fx = 0, sx = 0 / sx = W - 1)
    sx[lo] = 0
    fx[lo] = 0.0
    hi = sx >= W - 1
    sx[hi] = W - 1
    fx[hi] = 0.0
    a0, a1 = _cv_coef(fx)
    sx1 = np.minimum(sx + 1, W - 1)
    sy, fy = _cv_taps(h, H)
    b0, b1 = _cv_coef(fy)            # y border: the ROWS are clamped, the weights stay (VResize reads clip(sy + k, 0, H - 1))
    r0 = np.clip(sy, 0, H - 1)
    r1 = np.clip(sy + 1, 0, H - 1)
    hrow = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]       # [H, w, C], value * 2048
    S0, S1 = hrow[r0], hrow[r1]
    return (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2


def resize_linear_cv(img, h, w, quantize=True):
    """img [H, W, C] uint8 -> [h, w, C] float64.  quantize (what cv2.resize returns for uint8 input): the integer algorithm
    above; else the exact bilinear value on cv2.INTER_LINEAR's geometry src = (dst + 0.5) * (S/D) - 0.5, taps clamped."""
    if quantize:
        return resize_linear_cv_u8(img, h, w).astype(np.float64)
    H, W = img.shape[:2]
    src = img.astype(np.float64)

    def taps(D, S):
        f = (np.arange(D, dtype=np.float64) + 0.5) * (np.float32(S) / np.float32(D)).astype(np.float64) - 0.5
        i0 = np.floor(f).astype(np.int64)
        fr = f - i0
        lo = i0 < 0
        i0[lo] = 0
        fr[lo] = 0.0
        i1 = i0 + 1
        hi = i0 >= S - 1
        i0[hi] = S - 1
        i1[hi] = S - 1
        fr[hi] = 0.0
        return i0, i1, fr

    y0, y1, fy = taps(h, H)
    x0, x1, fx = taps(w, W)
    top = src[y0][:, x0] + fx[None, :, None] * (src[y0][:, x1] - src[y0][:, x0])
    bot = src[y1][:, x0] + fx[None, :, None] * (src[y1][:, x1] - src[y1][:, x0])
    out = top + fy[:, None, None] * (bot - top)
    return out


def clip_tensor(frames, first, step, count, h, w, hflip=False, quantize=True):
    """frames [N, H, W, 3] uint8 -> [3, count, h, w]: resize, /255, hflip, normalize(0.5, 0.5), permute.  quantize: float32,
    with the float32 operations of the reference's tensor pipeline (uint8 level / 255, then (x - 0.5) / 0.5) - bit-comparable
    with the HIP kernel; else float64."""
    out = np.empty((3, count, h, w), dtype=np.float32 if quantize else np.float64)
    for k in range(count):
        r = resize_linear_cv(frames[first + k * step], h, w, quantize)
        if hflip:
            r = r[:, ::-1]
        if quantize:
            v = (r.astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
        else:
            v = (r / 255.0 - 0.5) / 0.5
        out[:, k] = np.transpose(v, (2, 0, 1))
    return out
