"""CPU restatement (numpy, float64) of the data front-end: TEST INFRASTRUCTURE ONLY (tests/, smoke, bench baseline).

Follows the reference: datasets/generate_frames.py:44-46 (cv2.resize INTER_LINEAR per frame), datasets/video.py:53-82
(window frames[idx:idx+fps_lcm+1:every], /255, hflip, normalize(0.5, 0.5), permute C,T,H,W), datasets/image.py:20-49.
cv2 / kornia are not installed in the build image, so this restatement is NOT pinned against the reference's own
resizer: cv2 evaluates uint8 INTER_LINEAR with 11-bit fixed-point weights and can differ by one uint8 level from the
exact bilinear value computed here.  PARITY UNPINNED for this module."""
import numpy as np


def resize_linear_cv(img, h, w, quantize=True):
    """img [H, W, C] uint8 -> [h, w, C]; cv2.INTER_LINEAR geometry: src = (dst + 0.5) * (S/D) - 0.5, taps clamped."""
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
    return np.floor(out + 0.5) if quantize else out


def clip_tensor(frames, first, step, count, h, w, hflip=False, quantize=True):
    """frames [N, H, W, 3] uint8 -> float64 [3, count, h, w]: resize, /255, hflip, normalize(0.5, 0.5), permute."""
    out = np.empty((3, count, h, w), dtype=np.float64)
    for k in range(count):
        r = resize_linear_cv(frames[first + k * step], h, w, quantize)
        if hflip:
            r = r[:, ::-1]
        out[:, k] = np.transpose((r / 255.0 - 0.5) / 0.5, (2, 0, 1))
    return out
