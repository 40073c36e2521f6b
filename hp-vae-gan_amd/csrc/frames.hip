// Data front-end of the train step: the clip's frames go from full-resolution uint8 RGB to the
// normalised fp32 C,T,H,W tensor of a pyramid stage in one pass on the device.
//
// Replaces (reference, /root/reference): datasets/generate_frames.py:44-46 (cv2.resize INTER_LINEAR per frame),
// datasets/video.py:53-66,69-82 (temporal window frames[idx:idx+fps_lcm+1:every], /255, K.hflip, K.normalize(0.5, 0.5),
// permute to C,T,H,W) and datasets/image.py:20-49 (the same for one image).
//
// Resize = cv2's INTER_LINEAR geometry (pixel centres: src = (dst + 0.5) * (S/D) - 0.5, taps clamped to the image).
// cv2 evaluates uint8 images with 11-bit fixed-point weights; this kernel uses fp32 weights and rounds to the nearest
// uint8 level when `quantize` is set (cv2.resize returns uint8), so single pixels can differ from cv2 by one level.
// cv2 is not installed in the build image: PARITY UNPINNED against the reference's decoder/resizer.
#include "hpvg_common.h"
#include "hpvg.h"

namespace {

__global__ __launch_bounds__(256) void frames_resize_norm_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int H,
                                                                  int W, int first, int step, int count, int h, int w,
                                                                  float sy, float sx, int hflip, int quantize) {
  const long n = (long)count * h * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int k = (int)(i / ((long)w * h));
    const int xs = hflip ? (w - 1 - x) : x;  // K.hflip acts on the resized frame
    float fy = ((float)y + 0.5f) * sy - 0.5f;
    float fx = ((float)xs + 0.5f) * sx - 0.5f;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    fy -= (float)y0;
    fx -= (float)x0;
    if (y0 < 0) { y0 = 0; fy = 0.f; }
    if (x0 < 0) { x0 = 0; fx = 0.f; }
    int y1 = y0 + 1, x1 = x0 + 1;
    if (y0 >= H - 1) { y0 = H - 1; y1 = H - 1; fy = 0.f; }
    if (x0 >= W - 1) { x0 = W - 1; x1 = W - 1; fx = 0.f; }
    const unsigned char* f = src + (long)(first + k * step) * H * W * 3;
    const unsigned char* p00 = f + ((long)y0 * W + x0) * 3;
    const unsigned char* p01 = f + ((long)y0 * W + x1) * 3;
    const unsigned char* p10 = f + ((long)y1 * W + x0) * 3;
    const unsigned char* p11 = f + ((long)y1 * W + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = (float)p00[c] + fx * ((float)p01[c] - (float)p00[c]);
      const float bot = (float)p10[c] + fx * ((float)p11[c] - (float)p10[c]);
      float v = top + fy * (bot - top);
      if (quantize) v = floorf(v + 0.5f);
      // /255 then K.normalize(mean 0.5, std 0.5)
      dst[((long)c * count + k) * h * w + (long)y * w + x] = (v / 255.f - 0.5f) / 0.5f;
    }
  }
}

}  // namespace

extern "C" {

// src: N frames [N][H][W][3] uint8 RGB (device).  dst: [3][count][h][w] fp32 = normalize(resize(frame[first + k*step])).
int hpvg_frames_resize_norm_u8_f32(const unsigned char* src, float* dst, int N, int H, int W, int first, int step, int count, int h,
                                   int w, int hflip, int quantize, void* stream) {
  if (!src || !dst || N < 1 || H < 1 || W < 1 || h < 1 || w < 1 || count < 1 || step < 1 || first < 0) return HPVG_ERR_ARG;
  if ((long)first + (long)(count - 1) * step >= N) return HPVG_ERR_ARG;
  const long n = (long)count * h * w;
  long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(frames_resize_norm_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, first, step,
                     count, h, w, (float)H / (float)h, (float)W / (float)w, hflip, quantize);
  return hpvg_launch_status();
}

}  // extern "C"
