/* CPU ORACLE (test infrastructure only): plain-C definition of the convolution on the HP-VAE-GAN hot path.
 * nn.Conv3d / nn.Conv2d with kernel 3, stride 1, zero padding 1 (cross-correlation, bias added), NCDHW fp32,
 * accumulated in double.  Reference call sites: modules/networks_3d.py:51,63,175,341,362.
 * Built by __graft_entry__.build() into oracle/libconv_direct.so; used by tests to pin oracle/hpvg_oracle.py's `conv`
 * (which calls torch's CPU convolution) on small cases.  2-D convs: T = 1, KT = 1. */
#include <stddef.h>

void hpvg_oracle_conv_direct(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int T,
                             int H, int W, int KT) {
  const int pt = KT == 3 ? 1 : 0;
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < Cout; ++o)
      for (int t = 0; t < T; ++t)
        for (int h = 0; h < H; ++h)
          for (int ww = 0; ww < W; ++ww) {
            double acc = bias ? (double)bias[o] : 0.0;
            for (int c = 0; c < Cin; ++c)
              for (int dt = 0; dt < KT; ++dt) {
                const int tt = t + dt - pt;
                if (tt < 0 || tt >= T) continue;
                for (int dh = 0; dh < 3; ++dh) {
                  const int hh = h + dh - 1;
                  if (hh < 0 || hh >= H) continue;
                  for (int dw = 0; dw < 3; ++dw) {
                    const int wx = ww + dw - 1;
                    if (wx < 0 || wx >= W) continue;
                    acc += (double)w[(((size_t)o * Cin + c) * KT + dt) * 9 + dh * 3 + dw] *
                           (double)x[(((size_t)b * Cin + c) * T + tt) * H * W + (size_t)hh * W + wx];
                  }
                }
              }
            y[(((size_t)b * Cout + o) * T + t) * H * W + (size_t)h * W + ww] = (float)acc;
          }
}
