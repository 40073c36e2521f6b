// Data front-end of the train step: the clip's frames go from full-resolution uint8 RGB to the
// normalised fp32 C,T,H,W tensor of a pyramid stage in one pass on the device.
//
// Replaces (reference, /root/reference): datasets/generate_frames.py:44-46 (cv2.resize INTER_LINEAR per frame),
// datasets/video.py:53-66,69-82 (temporal window frames[idx:idx+fps_lcm+1:every], /255, K.hflip, K.normalize(0.5, 0.5),
// permute to C,T,H,W) and datasets/image.py:20-49 (the same for one image).
//
// Resize = cv2's INTER_LINEAR.  With `quantize` (what the reference gets: cv2.resize of uint8 frames returns uint8) the kernel
// restates OpenCV's 8-bit path in INTEGER arithmetic, exactly as oracle/frames.py does: positions f = float((d + 0.5) * scale -
// 0.5) with scale = 1 / (D / S) in double, 11-bit fixed-point tap weights rounded half-to-even from the float32 fractions
// (INTER_RESIZE_COEF_SCALE = 2048), the x border by zeroing the fraction and the y border by clamping the rows, a horizontal
// pass in 32-bit integers and the vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 - HIP output ==
// oracle output bit for bit (tests/test_frames.py, torch.equal).  Without `quantize`: the exact bilinear value in fp32.
// cv2 is not installed in the build image: PARITY UNPINNED against the reference's decoder/resizer.
#include "hpvg_common.h"
#include "hpvg.h"

namespace {

// tap position and float32 fraction of output index d along an axis of S source / D destination samples (cv2 resizeGeneric)
__device__ __forceinline__ void cv_tap(int d, double scale, int& s0, float& fr) {
  const float f = (float)(((double)d + 0.5) * scale - 0.5);
  s0 = (int)floorf(f);
  fr = f - (float)s0;
}

__global__ __launch_bounds__(256) void frames_resize_norm_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int H,
                                                                  int W, int first, int step, int count, int h, int w,
                                                                  double scale_y, double scale_x, float sy, float sx, int hflip,
                                                                  int quantize) {
  const long n = (long)count * h * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int x = (int)(i % w);
    const int y = (int)((i / w) % h);
    const int k = (int)(i / ((long)w * h));
    const int xs = hflip ? (w - 1 - x) : x;  // K.hflip acts on the resized frame
    const unsigned char* f = src + (long)(first + k * step) * H * W * 3;
    if (quantize) {
      int x0, y0;
      float fx, fy;
      cv_tap(xs, scale_x, x0, fx);
      cv_tap(y, scale_y, y0, fy);
      if (x0 < 0) { x0 = 0; fx = 0.f; }
      if (x0 >= W - 1) { x0 = W - 1; fx = 0.f; }
      const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
      const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
      const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
      const int r0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
      const int r1 = y0 + 1 < 0 ? 0 : (y0 + 1 > H - 1 ? H - 1 : y0 + 1);
      const unsigned char* p00 = f + ((long)r0 * W + x0) * 3;
      const unsigned char* p01 = f + ((long)r0 * W + x1) * 3;
      const unsigned char* p10 = f + ((long)r1 * W + x0) * 3;
      const unsigned char* p11 = f + ((long)r1 * W + x1) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = (int)p00[c] * a0 + (int)p01[c] * a1;
        const int S1 = (int)p10[c] * a0 + (int)p11[c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        // /255 then K.normalize(mean 0.5, std 0.5)
        dst[((long)c * count + k) * h * w + (long)y * w + x] = ((float)v / 255.f - 0.5f) / 0.5f;
      }
      continue;
    }
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
    const unsigned char* p00 = f + ((long)y0 * W + x0) * 3;
    const unsigned char* p01 = f + ((long)y0 * W + x1) * 3;
    const unsigned char* p10 = f + ((long)y1 * W + x0) * 3;
    const unsigned char* p11 = f + ((long)y1 * W + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = (float)p00[c] + fx * ((float)p01[c] - (float)p00[c]);
      const float bot = (float)p10[c] + fx * ((float)p11[c] - (float)p10[c]);
      const float v = top + fy * (bot - top);
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
                     count, h, w, 1.0 / ((double)h / (double)H), 1.0 / ((double)w / (double)W), (float)H / (float)h,
                     (float)W / (float)w, hflip, quantize);
  return hpvg_launch_status();
}

}  // extern "C"
