// HBM-bound elementwise / reduction kernels of the HP-VAE-GAN train step (gfx950, wave64).
// Each kernel cites the reference statement it replaces (paths relative to /root/reference).
// Layout: activations are NCDHW / NCHW contiguous fp32; S = T*H*W (or H*W) is the per-channel extent.
// Reductions accumulate short fp32 runs per thread, then combine in fp64 in a fixed order
// (two-stage: per-block partials -> one finishing block), so results are run-to-run identical.
#include "hpvg_common.h"
#include "hpvg.h"

namespace {

constexpr int RED_BLOCKS_MAX = 1024;

__device__ __forceinline__ float ld(const float* p, long i) { return p[i]; }

// ------------------------------------------------------------------ generic scalar reductions
enum RedOp { RED_SUM = 0, RED_SQ = 1, RED_SQDIFF = 2, RED_KL = 3 };

template <int OP>
__device__ __forceinline__ float red_term(const float* a, const float* b, long i) {
  if (OP == RED_SUM) return a[i];
  if (OP == RED_SQ) { const float v = a[i]; return v * v; }
  if (OP == RED_SQDIFF) { const float d = a[i] - b[i]; return d * d; }
  // RED_KL: -0.5*(1 + logvar - mu^2 - exp(logvar)), a = mu, b = logvar   (modules/losses.py:8)
  const float mu = a[i], lv = b[i];
  return -0.5f * (1.f + lv - mu * mu - expf(lv));
}

template <int OP>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                              double* __restrict__ part) {
  __shared__ double sh[4];
  double acc = 0.0;
  float loc = 0.f;
  int cnt = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    loc += red_term<OP>(a, b, i);
    if (++cnt == 32) { acc += loc; loc = 0.f; cnt = 0; }
  }
  acc += loc;
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// out[0] = scale * sum(part[0..np))   (single block)
__global__ __launch_bounds__(256) void reduce_finish_kernel(const double* __restrict__ part, int np, double scale,
                                                             float* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) acc += part[i];
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) out[0] = (float)(tot * scale);
}

template <int OP>
int reduce_scalar(const float* a, const float* b, long n, double scale, float* out, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!a || !out || !ws || n < 1) return HPVG_ERR_ARG;
  int nb = hpvg_cdiv(n, 256 * 16);
  if (nb > RED_BLOCKS_MAX) nb = RED_BLOCKS_MAX;
  if (nb < 1) nb = 1;
  if (ws_bytes < (size_t)nb * sizeof(double)) return HPVG_ERR_WORKSPACE;
  hipLaunchKernelGGL(reduce_partial_kernel<OP>, dim3(nb), dim3(256), 0, s, a, b, n, (double*)ws);
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nb, scale, out);
  return hpvg_launch_status();
}

// ------------------------------------------------------------------ BatchNorm (train mode)
// per-channel (sum, sumsq) partials: grid (nsplit, C).  V floats per lane and load (rows and chunks V-aligned): the
// dword version ran at 1.9 TB/s on 245 MB tensors, 16-byte loads reach the streaming kernels' ~5 TB/s.
template <int V>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, int B, int C, long S, int nsplit,
                                                                double* __restrict__ part) {
  // grid.y = C * G: G independent groups of B samples each, back to back in x (G = 1: plain BatchNorm); partials [g*C + c]
  typedef typename HpvgVec<V>::type Vec;
  __shared__ double sh[4];
  const int cg = blockIdx.y, k = blockIdx.x;
  const int g = cg / C, c = cg - g * C;
  x += (long)g * B * C * S;
  const long SV = S / V;
  const long chunk = (SV + nsplit - 1) / nsplit;
  const long lo = (long)k * chunk, hi = (lo + chunk < SV) ? lo + chunk : SV;
  double a1 = 0.0, a2 = 0.0;
  for (int b = 0; b < B; ++b) {
    const Vec* p = reinterpret_cast<const Vec*>(x + ((long)b * C + c) * S);
    float s1 = 0.f, s2 = 0.f;
    int cnt = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const Vec v = p[i];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float f = hpvg_vget<V>(v, e);
        s1 += f;
        s2 += f * f;
      }
      if (++cnt == 32 / V) { a1 += s1; a2 += s2; s1 = 0.f; s2 = 0.f; cnt = 0; }
    }
    a1 += s1;
    a2 += s2;
  }
  const double t1 = hpvg_block_sum_d(a1, sh);
  const double t2 = hpvg_block_sum_d(a2, sh);
  if (threadIdx.x == 0) {
    part[((long)cg * nsplit + k) * 2 + 0] = t1;
    part[((long)cg * nsplit + k) * 2 + 1] = t2;
  }
}

// mean / biased var -> invstd, fused affine (scale, shift), running-stat update (unbiased var, momentum)
// reference: nn.BatchNorm3d in ConvBlock3D, modules/networks_3d.py:54 (train mode on every forward)
__global__ void bn_finalize_kernel(const double* __restrict__ part, int nsplit, int C, double count, float eps, float momentum,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                   float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    s1 += part[((long)c * nsplit + k) * 2 + 0];
    s2 += part[((long)c * nsplit + k) * 2 + 1];
  }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const float sc = g * invstd;
  scale_out[c] = sc;
  shift_out[c] = bt - (float)mean * sc;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// sums[c][j] = sum_k part[c][k][j]: the rank-local per-channel pair handed to the all-reduce of a batch-split BatchNorm
__global__ void bn_sum_partials_kernel(const double* __restrict__ part, int nsplit, int C, double* __restrict__ sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    s1 += part[((long)c * nsplit + k) * 2 + 0];
    s2 += part[((long)c * nsplit + k) * 2 + 1];
  }
  sums[2 * c] = s1;
  sums[2 * c + 1] = s2;
}

// Train-mode BatchNorm apply with the statistics finalize folded in: every workgroup rebuilds ITS channel's (mean, invstd,
// scale, shift) from the nsplit partial pairs (one wave, fixed shuffle tree: the same bits in every workgroup); the
// workgroup (x = 0, b = 0) of a channel also publishes them and updates the running statistics.  Saves the finalize launch.
__global__ __launch_bounds__(256) void bn_apply_fused_kernel(const float* __restrict__ x, const double* __restrict__ part, int nsplit,
                                                              double count, float eps, float momentum,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                              float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                              float* __restrict__ y, int C, long S, int lrelu, int Bg, int G,
                                                              int gstride) {
  // G groups of Bg samples (statistics per group; `count` = elements per channel of ONE group; the *_out arrays of group g
  // start gstride floats after those of group g-1).  Running statistics: one workgroup per channel applies the groups'
  // updates one after the other, in order - what G separate BatchNorm calls would do.
  __shared__ float s_sc, s_sh;
  const int bc = blockIdx.y;
  const int b = bc / C, c = bc - b * C;
  const int g = b / Bg;
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;
    const long pc = (long)g * C + c;
    double s1 = k < nsplit ? part[(pc * nsplit + k) * 2 + 0] : 0.0;
    double s2 = k < nsplit ? part[(pc * nsplit + k) * 2 + 1] : 0.0;
    s1 = hpvg_wave_sum_d(s1);
    s2 = hpvg_wave_sum_d(s2);
    if (k == 0) {
      const double mean = s1 / count;
      double var = s2 / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
      const float sc = gm * invstd;
      const float sh = bt - (float)mean * sc;
      s_sc = sc;
      s_sh = sh;
      if (blockIdx.x == 0 && b == g * Bg) {
        const long o = (long)g * gstride + c;
        mean_out[o] = (float)mean;
        invstd_out[o] = invstd;
        scale_out[o] = sc;
        shift_out[o] = sh;
      }
    }
    if (running_mean && blockIdx.x == 0 && b == 0) {
      for (int gg = 0; gg < G; ++gg) {
        const long qc = (long)gg * C + c;
        double r1 = k < nsplit ? part[(qc * nsplit + k) * 2 + 0] : 0.0;
        double r2 = k < nsplit ? part[(qc * nsplit + k) * 2 + 1] : 0.0;
        r1 = hpvg_wave_sum_d(r1);
        r2 = hpvg_wave_sum_d(r2);
        if (k == 0) {
          const double mean = r1 / count;
          double var = r2 / count - mean * mean;
          if (var < 0.0) var = 0.0;
          const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
      }
    }
  }
  __syncthreads();
  const float sc = s_sc, sh = s_sh;
  const float* xp = x + (long)bc * S;
  float* yp = y + (long)bc * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
    float v = xp[i] * sc + sh;
    if (lrelu) v = hpvg_lrelu(v);
    yp[i] = v;
  }
}

// y = lrelu?(scale[c]*x + shift[c])      grid (blocks over S, B*C)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ y, int C, long S,
                                                          int lrelu) {
  const int bc = blockIdx.y;
  const int c = bc % C;
  const float sc = scale[c], sh = shift[c];
  const float* xp = x + (long)bc * S;
  float* yp = y + (long)bc * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
    float v = xp[i] * sc + sh;
    if (lrelu) v = hpvg_lrelu(v);
    yp[i] = v;
  }
}

// BN+LeakyReLU backward, reduction pass: per channel  s1 = sum dz, s2 = sum dz*xhat
//   z = scale*r + shift, dz = dh * (z > 0 ? 1 : 0.2), xhat = (r - mean)*invstd
template <int V>
__global__ __launch_bounds__(256) void bn_lrelu_bwd_reduce_kernel(const float* __restrict__ dh, const float* __restrict__ r,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   int B, int C, long S, int nsplit, int lrelu,
                                                                   double* __restrict__ part, int gstride) {
  // grid.y = C * G (groups as in bn_stats_partial_kernel; the statistics of group g start g * gstride floats further)
  typedef typename HpvgVec<V>::type Vec;
  __shared__ double sh[4];
  const int cg = blockIdx.y, k = blockIdx.x;
  const int g = cg / C, c = cg - g * C;
  dh += (long)g * B * C * S;
  r += (long)g * B * C * S;
  const long so = (long)g * gstride + c;
  const long SV = S / V;
  const long chunk = (SV + nsplit - 1) / nsplit;
  const long lo = (long)k * chunk, hi = (lo + chunk < SV) ? lo + chunk : SV;
  const float mu = mean[so], is = invstd[so], sc = scale[so], sf = shift[so];
  double a1 = 0.0, a2 = 0.0;
  for (int b = 0; b < B; ++b) {
    const Vec* dp = reinterpret_cast<const Vec*>(dh + ((long)b * C + c) * S);
    const Vec* rp = reinterpret_cast<const Vec*>(r + ((long)b * C + c) * S);
    float s1 = 0.f, s2 = 0.f;
    int cnt = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const Vec rv4 = rp[i];
      const Vec dz4 = dp[i];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float rv = hpvg_vget<V>(rv4, e);
        float dz = hpvg_vget<V>(dz4, e);
        if (lrelu && !(rv * sc + sf > 0.f)) dz *= HPVG_LRELU_SLOPE;
        s1 += dz;
        s2 += dz * ((rv - mu) * is);
      }
      if (++cnt == 32 / V) { a1 += s1; a2 += s2; s1 = 0.f; s2 = 0.f; cnt = 0; }
    }
    a1 += s1;
    a2 += s2;
  }
  const double t1 = hpvg_block_sum_d(a1, sh);
  const double t2 = hpvg_block_sum_d(a2, sh);
  if (threadIdx.x == 0) {
    part[((long)cg * nsplit + k) * 2 + 0] = t1;
    part[((long)cg * nsplit + k) * 2 + 1] = t2;
  }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int nsplit, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ sums, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    s1 += part[((long)c * nsplit + k) * 2 + 0];
    s2 += part[((long)c * nsplit + k) * 2 + 1];
  }
  dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
  dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
  sums[2 * c] = (float)s1;
  sums[2 * c + 1] = (float)s2;
}

// dr = gamma*invstd * (dz - s1/N - xhat*s2/N)
__global__ __launch_bounds__(256) void bn_lrelu_bwd_apply_kernel(const float* __restrict__ dh, const float* __restrict__ r,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ sums, float* __restrict__ dr, int C,
                                                                  long S, float inv_count, int lrelu) {
  const int bc = blockIdx.y;
  const int c = bc % C;
  const float mu = mean[c], is = invstd[c], sc = scale[c], sf = shift[c];
  const float m1 = sums[2 * c] * inv_count, m2 = sums[2 * c + 1] * inv_count;
  const float* dp = dh + (long)bc * S;
  const float* rp = r + (long)bc * S;
  float* op = dr + (long)bc * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
    const float rv = rp[i];
    float dz = dp[i];
    if (lrelu && !(rv * sc + sf > 0.f)) dz *= HPVG_LRELU_SLOPE;
    const float xh = (rv - mu) * is;
    op[i] = sc * (dz - m1 - xh * m2);
  }
}

// The same with the finalize folded in (single-GPU path): every workgroup sums its channel's nsplit partial pairs itself;
// workgroup (x = 0, b = 0) of a channel writes dbeta / dgamma (+= when accumulate).
__global__ __launch_bounds__(256) void bn_lrelu_bwd_apply_fused_kernel(const float* __restrict__ dh, const float* __restrict__ r,
                                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                                        const double* __restrict__ part, int nsplit,
                                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                        int accumulate, float* __restrict__ dr, int C, long S,
                                                                        float inv_count, int lrelu, int Bg, int G, int gstride) {
  __shared__ float s_m1, s_m2;
  const int bc = blockIdx.y;
  const int b = bc / C, c = bc - b * C;
  const int g = b / Bg;
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;
    const long pc = (long)g * C + c;
    double s1 = k < nsplit ? part[(pc * nsplit + k) * 2 + 0] : 0.0;
    double s2 = k < nsplit ? part[(pc * nsplit + k) * 2 + 1] : 0.0;
    s1 = hpvg_wave_sum_d(s1);
    s2 = hpvg_wave_sum_d(s2);
    if (k == 0) {
      s_m1 = (float)s1 * inv_count;
      s_m2 = (float)s2 * inv_count;
    }
    if (blockIdx.x == 0 && b == 0) {  // dbeta / dgamma: the groups' sums added up in order by one workgroup per channel
      float tb = 0.f, tg = 0.f;
      for (int gg = 0; gg < G; ++gg) {
        const long qc = (long)gg * C + c;
        double r1 = k < nsplit ? part[(qc * nsplit + k) * 2 + 0] : 0.0;
        double r2 = k < nsplit ? part[(qc * nsplit + k) * 2 + 1] : 0.0;
        r1 = hpvg_wave_sum_d(r1);
        r2 = hpvg_wave_sum_d(r2);
        tb += (float)r1;
        tg += (float)r2;
      }
      if (k == 0) {
        dbeta[c] = accumulate ? dbeta[c] + tb : tb;
        dgamma[c] = accumulate ? dgamma[c] + tg : tg;
      }
    }
  }
  __syncthreads();
  const long so = (long)g * gstride + c;
  const float mu = mean[so], is = invstd[so], sc = scale[so], sf = shift[so];
  const float m1 = s_m1, m2 = s_m2;
  const float* dp = dh + (long)bc * S;
  const float* rp = r + (long)bc * S;
  float* op = dr + (long)bc * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
    const float rv = rp[i];
    float dz = dp[i];
    if (lrelu && !(rv * sc + sf > 0.f)) dz *= HPVG_LRELU_SLOPE;
    const float xh = (rv - mu) * is;
    op[i] = sc * (dz - m1 - xh * m2);
  }
}

// ------------------------------------------------------------------ BatchNorm + LeakyReLU, SECOND order
// The gradient penalty of a critic with BatchNorm (WDiscriminatorBaselines, networks_3d.py:184-210; modules/utils.py:14-18)
// differentiates the first-order backward  dr = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)),  dz = dh*lrelu'(z),
// once more.  With G = dL/d(dr), N elements per channel and per-channel sums  Sd = sum dz, Sdx = sum dz*xhat, SG = sum G,
// SGx = sum G*xhat, SdG = sum dz*G  (torch: batchnorm_double_backward, ggG = ggB = none; lrelu'' = 0):
//   dL/d(dh)  = lrelu'(z) * gamma*invstd * (G - SG/N - xhat*SGx/N)
//   dL/d(r)   = gamma*invstd^2 * ( xhat*(SG*Sd/N - SdG + 3*Sdx*SGx/N)/N + (SGx/N)*(Sd/N - dz) + (Sdx/N)*(SG/N - G) )
//   dL/dgamma = invstd * (SdG - Sd*SG/N - Sdx*SGx/N)
template <int V>
__global__ __launch_bounds__(256) void bn_lrelu_bwd2_reduce_kernel(const float* __restrict__ dh, const float* __restrict__ G,
                                                                    const float* __restrict__ r, const float* __restrict__ mean,
                                                                    const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, int B, int C, long S, int nsplit,
                                                                    int lrelu, double* __restrict__ part) {
  typedef typename HpvgVec<V>::type Vec;
  __shared__ double sh[4];
  const int c = blockIdx.y, k = blockIdx.x;
  const long SV = S / V;
  const long chunk = (SV + nsplit - 1) / nsplit;
  const long lo = (long)k * chunk, hi = (lo + chunk < SV) ? lo + chunk : SV;
  const float mu = mean[c], is = invstd[c], sc = scale[c], sf = shift[c];
  double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = 0; b < B; ++b) {
    const Vec* dp = reinterpret_cast<const Vec*>(dh + ((long)b * C + c) * S);
    const Vec* gp = reinterpret_cast<const Vec*>(G + ((long)b * C + c) * S);
    const Vec* rp = reinterpret_cast<const Vec*>(r + ((long)b * C + c) * S);
    float f[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const Vec rv4 = rp[i], dz4 = dp[i], g4 = gp[i];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float rv = hpvg_vget<V>(rv4, e);
        float dz = hpvg_vget<V>(dz4, e);
        const float g = hpvg_vget<V>(g4, e);
        if (lrelu && !(rv * sc + sf > 0.f)) dz *= HPVG_LRELU_SLOPE;
        const float xh = (rv - mu) * is;
        f[0] += dz; f[1] += dz * xh; f[2] += g; f[3] += g * xh; f[4] += dz * g;
      }
      if (++cnt == 32 / V) {
#pragma unroll
        for (int q = 0; q < 5; ++q) { a[q] += f[q]; f[q] = 0.f; }
        cnt = 0;
      }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) a[q] += f[q];
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const double t = hpvg_block_sum_d(a[q], sh);
    if (threadIdx.x == 0) part[((long)c * nsplit + k) * 5 + q] = t;
  }
}

// per channel: the five sums -> constants of the apply pass (Sd/N, Sdx/N, SG/N, SGx/N, all_sub/N) and dgamma
__global__ void bn_bwd2_finalize_kernel(const double* __restrict__ part, int nsplit, int C, double count,
                                        const float* __restrict__ invstd, float* __restrict__ consts, float* __restrict__ dgamma,
                                        int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < nsplit; ++k)
    for (int q = 0; q < 5; ++q) t[q] += part[((long)c * nsplit + k) * 5 + q];
  const double Sd = t[0], Sdx = t[1], SG = t[2], SGx = t[3], SdG = t[4];
  const double inv = 1.0 / count;
  consts[5 * c + 0] = (float)(Sd * inv);
  consts[5 * c + 1] = (float)(Sdx * inv);
  consts[5 * c + 2] = (float)(SG * inv);
  consts[5 * c + 3] = (float)(SGx * inv);
  consts[5 * c + 4] = (float)((SG * Sd * inv - SdG + 3.0 * Sdx * SGx * inv) * inv);
  if (dgamma) {
    const float v = (float)((double)invstd[c] * (SdG - Sd * SG * inv - Sdx * SGx * inv));
    dgamma[c] = accumulate ? dgamma[c] + v : v;
  }
}

__global__ __launch_bounds__(256) void bn_lrelu_bwd2_apply_kernel(const float* __restrict__ dh, const float* __restrict__ G,
                                                                   const float* __restrict__ r, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ consts,
                                                                   float* __restrict__ g_dh, float* __restrict__ g_r, int C, long S,
                                                                   int lrelu) {
  const int bc = blockIdx.y;
  const int c = bc % C;
  const float mu = mean[c], is = invstd[c], sc = scale[c], sf = shift[c];
  const float mSd = consts[5 * c], mSdx = consts[5 * c + 1], mSG = consts[5 * c + 2], mSGx = consts[5 * c + 3], asub = consts[5 * c + 4];
  const float* dp = dh + (long)bc * S;
  const float* gp = G + (long)bc * S;
  const float* rp = r + (long)bc * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < S; i += (long)gridDim.x * 256) {
    const float rv = rp[i], g = gp[i];
    float dz = dp[i];
    const float m = (lrelu && !(rv * sc + sf > 0.f)) ? HPVG_LRELU_SLOPE : 1.f;
    dz *= m;
    const float xh = (rv - mu) * is;
    if (g_dh) g_dh[(long)bc * S + i] = m * sc * (g - mSG - xh * mSGx);
    if (g_r) g_r[(long)bc * S + i] = sc * is * (xh * asub + mSGx * (mSd - dz) + mSdx * (mSG - g));
  }
}

// ------------------------------------------------------------------ pointwise
// out = dy * (h > 0 ? 1 : 0.2)    (leaky_relu_backward on the in-place activated tensor)
__global__ __launch_bounds__(256) void lrelu_mask_mul_kernel(const float* __restrict__ dy, const float* __restrict__ h,
                                                              float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = h[i] > 0.f ? dy[i] : HPVG_LRELU_SLOPE * dy[i];
}

// y = tanh(x + res)   (res optional)   networks_3d.py:377,404
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = x[i];
    if (res) v += res[i];
    y[i] = tanhf(v);
  }
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float t = y[i];
    dx[i] = dy[i] * (1.f - t * t);
  }
}

// z = eps*exp(0.5*logvar) + mu   networks_3d.py:31-33
__global__ __launch_bounds__(256) void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                           const float* __restrict__ eps, float* __restrict__ z, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    z[i] = eps[i] * expf(0.5f * lv[i]) + mu[i];
}
// dlogvar = dz * eps * 0.5*exp(0.5*logvar)   (dmu = dz)
__global__ __launch_bounds__(256) void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ lv,
                                                           const float* __restrict__ eps, float* __restrict__ dlv, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dlv[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * lv[i]);
}
// KL backward: dmu = g*mu/n ; dlogvar = g*(-0.5)*(1-exp(lv))/n
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ g, const float* __restrict__ mu,
                                                      const float* __restrict__ lv, float* __restrict__ dmu,
                                                      float* __restrict__ dlv, long n) {
  const float gs = g[0] / (float)n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    dmu[i] = gs * mu[i];
    dlv[i] = gs * (-0.5f) * (1.f - expf(lv[i]));
  }
}
// MSE backward: da = g*2*(a-b)/n
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                       const float* __restrict__ b, float* __restrict__ da, long n) {
  const float gs = 2.f * g[0] / (float)n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) da[i] = gs * (a[i] - b[i]);
}
// out[i] = g[0]*coef
__global__ __launch_bounds__(256) void fill_scaled_kernel(const float* __restrict__ g, float coef, float* __restrict__ out,
                                                           long n) {
  const float v = g[0] * coef;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = 0.f;
}
// out = a + b
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] + b[i];
}
// out = x / s[0]   (spectral norm: weight = weight_orig / sigma)
__global__ __launch_bounds__(256) void div_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                          float* __restrict__ out, long n) {
  const float d = s[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] / d;
}
// out = alpha*a + (1-alpha)*b, alpha a device scalar    modules/utils.py:9
__global__ __launch_bounds__(256) void lerp_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ alpha, float* __restrict__ out, long n) {
  const float al = alpha[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = al * a[i] + (1.f - al) * b[i];
}

// ------------------------------------------------------------------ gradient penalty   modules/utils.py:18
// per (b, voxel): nrm = ||g[b,:,s]||_2 over C channels; partial sum of (nrm-1)^2
__global__ __launch_bounds__(256) void gp_partial_kernel(const float* __restrict__ g, int B, int C, long S,
                                                          double* __restrict__ part) {
  __shared__ double sh[4];
  double acc = 0.0;
  const long n = (long)B * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / S, s = i - b * S;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = g[((long)b * C + c) * S + s];
      ss += v * v;
    }
    const float d = sqrtf(ss) - 1.f;
    acc += (double)(d * d);
  }
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// dg[b,c,s] = gout * lambda/(B*S) * 2*(nrm-1)/nrm * g[b,c,s]      (0 where nrm == 0)
__global__ __launch_bounds__(256) void gp_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ g,
                                                      float* __restrict__ dg, int B, int C, long S, float lambda) {
  const long n = (long)B * S;
  const float k = gout[0] * lambda * 2.f / (float)n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / S, s = i - b * S;
    float ss = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = g[((long)b * C + c) * S + s];
      ss += v * v;
    }
    const float nrm = sqrtf(ss);
    const float f = nrm > 0.f ? k * (nrm - 1.f) / nrm : 0.f;
    for (int c = 0; c < C; ++c) {
      const long o = ((long)b * C + c) * S + s;
      dg[o] = f * g[o];
    }
  }
}

// ------------------------------------------------------------------ N(0,1) noise: Philox4x32-10 + Box-Muller
// utils/images.py:49 (zeros(...).normal_(0, 1)) and networks_3d.py:32 (the reparameterisation eps).  Counter-based, so a
// draw needs no generator state in memory: element quadruple q of call `call_id` in iteration iter[0] under `seed` is
// philox(counter = (q_lo, q_hi, call_id, iter[0]), key = seed).  `iter` lives on the device and is bumped once per train
// iteration by hpvg_counter_inc_i32 - a replayed hipGraph draws fresh noise without any launch argument changing.
struct Philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// four N(0,1) values from one Philox block (two Box-Muller pairs; uniforms in (0, 1) from the top 24 bits)
__device__ __forceinline__ void philox_normal4(long q, unsigned call_id, unsigned iter, unsigned long long seed, float (&n)[4]) {
  const Philox4 r = philox4x32_10((unsigned)q, (unsigned)((unsigned long long)q >> 32), call_id, iter, (unsigned)seed,
                                  (unsigned)(seed >> 32));
  const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.f / 16777216.f), u1 = ((float)(r.y >> 8) + 0.5f) * (1.f / 16777216.f);
  const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.f / 16777216.f), u3 = ((float)(r.w >> 8) + 0.5f) * (1.f / 16777216.f);
  const float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
  float sa, ca, sb, cb;
  sincosf(6.283185307179586f * u1, &sa, &ca);
  sincosf(6.283185307179586f * u3, &sb, &cb);
  n[0] = ra * ca; n[1] = ra * sa; n[2] = rb * cb; n[3] = rb * sb;
}
__global__ __launch_bounds__(256) void normal_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned call_id,
                                                      const int* __restrict__ iter) {
  const unsigned it = iter ? (unsigned)iter[0] : 0u;
  const long nq = (n + 3) / 4;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    float v[4];
    philox_normal4(q, call_id, it, seed, v);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < n) out[4 * q + e] = v[e];
  }
}

// U(0,1) from the same Philox stream (reparameterize_bern's eps, networks_3d.py:40)
__global__ __launch_bounds__(256) void uniform_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned call_id,
                                                       const int* __restrict__ iter) {
  const unsigned it = iter ? (unsigned)iter[0] : 0u;
  const long nq = (n + 3) / 4;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    const Philox4 r = philox4x32_10((unsigned)q, (unsigned)((unsigned long long)q >> 32), call_id, it, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < n) out[4 * q + e] = (float)(w[e] >> 8) * (1.f / 16777216.f);   // [0, 1) like Tensor.uniform_()
  }
}

// ------------------------------------------------------------------ tri/bi-linear resize, align_corners=True
// utils/images.py:13,17,24 (F.interpolate(..., align_corners=True)).  src = dst*(in-1)/(out-1) in fp32.
struct Lin { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lin lin_coef(int o, int in, float scale) {
  const float src = scale * (float)o;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
  const float w1 = src - (float)i0;
  return Lin{i0, i1, 1.f - w1, w1};
}

// y[bc][to][ho][wo]; optional fused noise injection: yn = y + amp*noise   (networks_3d.py:399-400)
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ noise, float amp, float* __restrict__ yn,
                                                            long BC, int Ti, int Hi, int Wi, int To, int Ho, int Wo, float st,
                                                            float sh, float sw) {
  const long n = BC * To * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long r = i;
    const int wo = r % Wo; r /= Wo;
    const int ho = r % Ho; r /= Ho;
    const int to = r % To; r /= To;
    const float* xp = x + r * ((long)Ti * Hi * Wi);
    const Lin lt = lin_coef(to, Ti, st), lh = lin_coef(ho, Hi, sh), lw = lin_coef(wo, Wi, sw);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ti = a ? lt.i1 : lt.i0;
      const float wt = a ? lt.w1 : lt.w0;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int hi = b ? lh.i1 : lh.i0;
        const float wh = b ? lh.w1 : lh.w0;
        const float* row = xp + ((long)ti * Hi + hi) * Wi;
        acc += wt * wh * (lw.w0 * row[lw.i0] + lw.w1 * row[lw.i1]);
      }
    }
    y[i] = acc;
    if (yn) yn[i] = acc + amp * noise[i];
  }
}
// The same with the level noise GENERATED in the kernel (networks_3d.py:395-400: up + noise * amp with noise ~ N(0, 1)): the
// noise tensor never exists in memory.  Samples (bc / C) below `first_noisy` get no noise (the reconstruction half of a
// merged generator pass).  Element i of the output uses value i % 4 of Philox block i / 4 - what normal_kernel writes.
__global__ __launch_bounds__(256) void upsample_noise_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  float* __restrict__ yn, float amp, long BC, int C, int first_noisy,
                                                                  int Ti, int Hi, int Wi, int To, int Ho, int Wo, float st, float sh,
                                                                  float sw, unsigned long long seed, unsigned call_id,
                                                                  const int* __restrict__ iter) {
  const unsigned it = iter ? (unsigned)iter[0] : 0u;
  const long n = BC * To * Ho * Wo;
  const long nq = (n + 3) / 4;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    float nz[4];
    philox_normal4(q, call_id, it, seed, nz);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long i = 4 * q + e;
      if (i >= n) break;
      long r = i;
      const int wo = r % Wo; r /= Wo;
      const int ho = r % Ho; r /= Ho;
      const int to = r % To; r /= To;
      const float* xp = x + r * ((long)Ti * Hi * Wi);
      const Lin lt = lin_coef(to, Ti, st), lh = lin_coef(ho, Hi, sh), lw = lin_coef(wo, Wi, sw);
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ti = a ? lt.i1 : lt.i0;
        const float wt = a ? lt.w1 : lt.w0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int hi = b ? lh.i1 : lh.i0;
          const float wh = b ? lh.w1 : lh.w0;
          const float* row = xp + ((long)ti * Hi + hi) * Wi;
          acc += wt * wh * (lw.w0 * row[lw.i0] + lw.w1 * row[lw.i1]);
        }
      }
      y[i] = acc;
      yn[i] = (int)(r / C) >= first_noisy ? acc + amp * nz[e] : acc;
    }
  }
}

// Backward of the resize as a GATHER: one thread per INPUT voxel sums, in a fixed order, the contributions of the output
// voxels whose interpolation touches it (no float atomics, no pre-zeroed buffer: bitwise reproducible).  Per axis the
// candidate outputs of input index i are those with src = scale*o in (i-1, i+1); the exact membership test re-evaluates
// lin_coef(o) - the forward's own fp32 expression - so the weights are the forward's weights bit for bit.
// dy2 (nullable): a second gradient of the same shape (the `up` and `up + amp*noise` outputs of one resize both carry
// gradients into it), added on load.
__device__ __forceinline__ void gather_range(int i, int out, float scale, int& lo, int& hi) {
  if (!(scale > 0.f)) { lo = 0; hi = out - 1; return; }   // out == 1 (or in == 1): every output reads input 0 (and 1 with weight 0)
  const float inv = 1.f / scale;
  lo = (int)floorf((float)(i - 1) * inv) - 1;
  hi = (int)ceilf((float)(i + 1) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}
__device__ __forceinline__ float gather_weight(int o, int i, int in, float scale) {
  const Lin l = lin_coef(o, in, scale);
  float w = 0.f;
  if (l.i0 == i) w += l.w0;
  if (l.i1 == i) w += l.w1;
  return w;
}
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dy2,
                                                            float* __restrict__ dx, long BC, int Ti, int Hi, int Wi, int To, int Ho,
                                                            int Wo, float st, float sh, float sw) {
  const long n = BC * Ti * Hi * Wi;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long r = i;
    const int wi = r % Wi; r /= Wi;
    const int hi = r % Hi; r /= Hi;
    const int ti = r % Ti; r /= Ti;
    const float* gp = dy + r * ((long)To * Ho * Wo);
    const float* gp2 = dy2 ? dy2 + r * ((long)To * Ho * Wo) : nullptr;
    int t0, t1, h0, h1, w0, w1;
    gather_range(ti, To, st, t0, t1);
    gather_range(hi, Ho, sh, h0, h1);
    gather_range(wi, Wo, sw, w0, w1);
    float acc = 0.f;
    for (int to = t0; to <= t1; ++to) {
      const float wt = gather_weight(to, ti, Ti, st);
      if (wt == 0.f) continue;
      for (int ho = h0; ho <= h1; ++ho) {
        const float wh = gather_weight(ho, hi, Hi, sh);
        if (wh == 0.f) continue;
        const long row = ((long)to * Ho + ho) * Wo;
        float racc = 0.f;
        for (int wo = w0; wo <= w1; ++wo) {
          const float ww = gather_weight(wo, wi, Wi, sw);
          if (ww == 0.f) continue;
          float g = gp[row + wo];
          if (gp2) g += gp2[row + wo];
          racc += ww * g;
        }
        acc += wt * wh * racc;
      }
    }
    dx[i] = acc;
  }
}

// ------------------------------------------------------------------ spectral norm (torch hook semantics, 1 power iteration)
// networks_3d.py:63 nn.utils.spectral_norm: v <- normalize(W^T u), u <- normalize(W v), sigma = u^T W v
// single workgroup of 1024 threads; W is [Co][K] row-major (K = Cin*taps).
__device__ __forceinline__ double block1024_sum(double v, double* sh) {
  v = hpvg_wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < 16; ++i) t += sh[i];
  return t;
}

__device__ __forceinline__ void sn_power_iter_body(const float* __restrict__ w, float* __restrict__ u,
                                                              float* __restrict__ v, float* __restrict__ sigma_out,
                                                              float* __restrict__ inv_sigma_out, int Co, int K, int do_iter,
                                                              float eps, float* __restrict__ wv_ws, float* __restrict__ uv_copy,
                                                              float* __restrict__ w_eff) {
  // The whole power iteration is one dependent chain on ONE workgroup (442 KB of weights): what it costs is memory
  // latency, so the two matrix-vector products read W as float4 with every thread's loads independent of each other.
  __shared__ double sh[16];
  __shared__ float su[1024];
  __shared__ float4 part[1024];
  __shared__ float4 sv4[1024];
  __shared__ float s_sigma;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int K4 = K >> 2;
  const bool vec = (K & 3) == 0 && K4 <= 1024 && (((size_t)w | (size_t)v) & 15) == 0;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int o = tid; o < Co; o += 1024) su[o] = u[o];
  __syncthreads();
  if (do_iter) {
    // v = normalize(W^T u)
    double nrm = 0.0;
    if (vec) {
      // thread = (float4 column c, row slice s): S row slices share a column and meet in LDS
      const int S = K4 <= 512 ? min(1024 / K4, 8) : 1;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tid < S * K4) {
        const int c = tid % K4, s0 = tid / K4;
#pragma unroll 8
        for (int o = s0; o < Co; o += S) {
          const float4 x = w4[(long)o * K4 + c];
          const float uo = su[o];
          acc.x += x.x * uo; acc.y += x.y * uo; acc.z += x.z * uo; acc.w += x.w * uo;
        }
      }
      part[tid] = acc;
      __syncthreads();
      if (tid < K4) {
        float4 t = part[tid];
        for (int s1 = 1; s1 < S; ++s1) {
          const float4 q = part[s1 * K4 + tid];
          t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
        }
        sv4[tid] = t;
        nrm = (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
      }
    } else {
      for (int k = tid; k < K; k += 1024) {
        float acc = 0.f;
        for (int o = 0; o < Co; ++o) acc += w[(long)o * K + k] * su[o];
        v[k] = acc;
        nrm += (double)acc * acc;
      }
    }
    const double tot = block1024_sum(nrm, sh);
    const float den = fmaxf((float)sqrt(tot), eps);
    __syncthreads();
    if (vec) {
      if (tid < K4) {
        float4 t = sv4[tid];
        t.x /= den; t.y /= den; t.z /= den; t.w /= den;
        sv4[tid] = t;
        reinterpret_cast<float4*>(v)[tid] = t;
      }
    } else {
      for (int k = tid; k < K; k += 1024) v[k] = v[k] / den;
    }
    __syncthreads();
  } else if (vec) {
    if (tid < K4) sv4[tid] = reinterpret_cast<const float4*>(v)[tid];
    __syncthreads();
  }
  // wv = W v  (one wave per row, rows strided by 16)
  if (vec) {
    for (int o = wave; o < Co; o += 16) {
      float acc = 0.f;
#pragma unroll 7
      for (int c = lane; c < K4; c += 64) {
        const float4 x = w4[(long)o * K4 + c];
        const float4 y = sv4[c];
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
      acc = hpvg_wave_sum(acc);
      if (lane == 0) wv_ws[o] = acc;
    }
  } else {
    for (int o = wave; o < Co; o += 16) {
      float acc = 0.f;
      for (int k = lane; k < K; k += 64) acc += w[(long)o * K + k] * v[k];
      acc = hpvg_wave_sum(acc);
      if (lane == 0) wv_ws[o] = acc;
    }
  }
  __syncthreads();
  if (do_iter) {
    double nrm = 0.0;
    for (int o = tid; o < Co; o += 1024) nrm += (double)wv_ws[o] * wv_ws[o];
    const double tot = block1024_sum(nrm, sh);
    const float den = fmaxf((float)sqrt(tot), eps);
    for (int o = tid; o < Co; o += 1024) {
      const float un = wv_ws[o] / den;
      su[o] = un;
      u[o] = un;
    }
    __syncthreads();
  }
  double sg = 0.0;
  for (int o = tid; o < Co; o += 1024) sg += (double)su[o] * wv_ws[o];
  const double sig = block1024_sum(sg, sh);
  if (tid == 0) {
    sigma_out[0] = (float)sig;
    inv_sigma_out[0] = (float)(1.0 / sig);
    s_sigma = (float)sig;
  }
  if (uv_copy) {  // the (u, v) this sigma belongs to, for the backward (later forwards overwrite the buffers)
    for (int o = tid; o < Co; o += 1024) uv_copy[o] = su[o];
    if (vec) {
      for (int c = tid; c < K4; c += 1024) {
        const float4 t = sv4[c];
        float* d = uv_copy + Co + 4 * c;
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
      }
    } else {
      for (int k = tid; k < K; k += 1024) uv_copy[Co + k] = v[k];
    }
  }
  if (w_eff) {  // weight = weight_orig / sigma (the batched launch folds the division in: one workgroup per layer)
    __syncthreads();
    const float d = s_sigma;
    const long n = (long)Co * K;
    if (vec) {
      const long n4 = n >> 2;
      float4* o4 = reinterpret_cast<float4*>(w_eff);
#pragma unroll 4
      for (long i = tid; i < n4; i += 1024) {
        const float4 x = w4[i];
        o4[i] = make_float4(x.x / d, x.y / d, x.z / d, x.w / d);
      }
    } else {
      for (long i = tid; i < n; i += 1024) w_eff[i] = w[i] / d;
    }
  }
}

__global__ __launch_bounds__(1024) void sn_power_iter_kernel(const float* __restrict__ w, float* __restrict__ u,
                                                              float* __restrict__ v, float* __restrict__ sigma_out,
                                                              float* __restrict__ inv_sigma_out, int Co, int K, int do_iter,
                                                              float eps, float* __restrict__ wv_ws, float* __restrict__ uv_copy) {
  sn_power_iter_body(w, u, v, sigma_out, inv_sigma_out, Co, K, do_iter, eps, wv_ws, uv_copy, nullptr);
}

// All spectral-norm layers of a network in ONE launch, one workgroup per layer (they are independent): the six power
// iterations of a discriminator forward run side by side instead of as six ~19 us single-workgroup launches in a row,
// and each workgroup also writes weight = weight_orig / sigma.
struct SnBatchArgs {
  const float* w[HPVG_SN_BATCH_MAX];
  float* u[HPVG_SN_BATCH_MAX];
  float* v[HPVG_SN_BATCH_MAX];
  float* sig[HPVG_SN_BATCH_MAX];      // sigma, 1/sigma (2 floats)
  float* uv_copy[HPVG_SN_BATCH_MAX];
  float* w_eff[HPVG_SN_BATCH_MAX];
  float* wv_ws[HPVG_SN_BATCH_MAX];
  int Co[HPVG_SN_BATCH_MAX], K[HPVG_SN_BATCH_MAX];
};
__global__ __launch_bounds__(1024) void sn_power_iter_batch_kernel(const SnBatchArgs a, int do_iter, float eps) {
  const int i = blockIdx.x;
  sn_power_iter_body(a.w[i], a.u[i], a.v[i], a.sig[i], a.sig[i] + 1, a.Co[i], a.K[i], do_iter, eps, a.wv_ws[i], a.uv_copy[i],
                     a.w_eff[i]);
}

// backward of W = W_orig / sigma, sigma = u^T W_orig v (u, v constants):
//   dW_orig[o][k] (+)= dW[o][k]/sigma - (sum(dW .* W_orig)/sigma^2) * u[o] v[k]
// Two launches over G workgroups (4096 elements each): partial dots -> ws (doubles), then every workgroup sums the G
// partials in the same fixed order and applies its chunk.  (One workgroup for everything took 40-70 us on 442 KB.)
constexpr int SN_CHUNK = 4096;
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const float* __restrict__ dweff, const float* __restrict__ worig, long n,
                                                          int vec, double* __restrict__ part) {
  __shared__ double sh[4];
  const long lo = (long)blockIdx.x * SN_CHUNK;
  const long hi = lo + SN_CHUNK < n ? lo + SN_CHUNK : n;
  double acc = 0.0;
  if (vec) {
    const float4* a4 = reinterpret_cast<const float4*>(dweff);
    const float4* b4 = reinterpret_cast<const float4*>(worig);
#pragma unroll 4
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256) {
      const float4 a = a4[i], b = b4[i];
      acc += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) acc += (double)dweff[i] * worig[i];
  }
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ dweff, const float* __restrict__ u,
                                                            const float* __restrict__ v, const float* __restrict__ sigma,
                                                            const double* __restrict__ part, int nparts,
                                                            float* __restrict__ dworig, long n, int K, int vec, int accumulate) {
  double dot = 0.0;
  for (int g = 0; g < nparts; ++g) dot += part[g];  // same order in every workgroup
  const float sg = sigma[0];
  const float coef = (float)(dot / ((double)sg * sg));
  const long lo = (long)blockIdx.x * SN_CHUNK;
  const long hi = lo + SN_CHUNK < n ? lo + SN_CHUNK : n;
  if (vec) {
    const float4* a4 = reinterpret_cast<const float4*>(dweff);
    const float4* v4 = reinterpret_cast<const float4*>(v);
    float4* o4 = reinterpret_cast<float4*>(dworig);
    const int K4 = K >> 2;
#pragma unroll 4
    for (long i = (lo >> 2) + threadIdx.x; i < (hi >> 2); i += 256) {
      const int o = (int)(i / K4), c = (int)(i - (long)o * K4);
      const float4 a = a4[i], y = v4[c];
      const float cu = coef * u[o];
      float4 t = make_float4(a.x / sg - cu * y.x, a.y / sg - cu * y.y, a.z / sg - cu * y.z, a.w / sg - cu * y.w);
      if (accumulate) {
        const float4 p = o4[i];
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      o4[i] = t;
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const int o = (int)(i / K), k = (int)(i - (long)o * K);
      const float t = dweff[i] / sg - coef * u[o] * v[k];
      dworig[i] = accumulate ? dworig[i] + t : t;
    }
  }
}

// ------------------------------------------------------------------ optimizer   train_video.py:88,201-202
// g *= min(1, max_norm/(sqrt(sqsum)+1e-6))   torch.nn.utils.clip_grad_norm_
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, long n, const float* __restrict__ sqsum,
                                                          float max_norm, float* __restrict__ coef_out) {
  const float total = sqrtf(sqsum[0]);
  float coef = max_norm / (total + 1e-6f);
  if (coef > 1.f) coef = 1.f;
  if (coef_out && blockIdx.x == 0 && threadIdx.x == 0) { coef_out[0] = coef; coef_out[1] = total; }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) g[i] *= coef;
}
// torch.optim.Adam (amsgrad=False, weight_decay=0).  The step count comes either from the host (`step`) or, when
// step_dev != nullptr, from device memory (hipGraph replays: the captured launch must not bake the count in).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                                                    int step, const int* __restrict__ step_dev) {
  const int t = step_dev ? step_dev[0] : step;
  const float bc1 = 1.f - powf(beta1, (float)t);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)t));
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}
__global__ void counter_inc_kernel(int* c) { if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1; }

inline int ew_blocks(long n) {
  long nb = (n + 256 * 4 - 1) / (256 * 4);
  if (nb < 1) nb = 1;
  if (nb > 4096) nb = 4096;
  return (int)nb;
}
// BatchNorm apply kernels fold the finalize in below this many elements (a few rounds of 4096-element workgroups); above, the extra
// launch is cheaper than the per-workgroup prologue (A/B on stages 3-7: backward +0.4 %, forward neutral below the bound; fused everywhere: -1 % at stages >= 6)
constexpr long HPVG_BN_FUSE_MAX = 1L << 25;
inline int bn_nsplit(int B, int C, long S) {
  // enough blocks to fill the chip (>= ~1024) but at least ~2048 elements per block
  long want = (1024 + C - 1) / C;
  long maxs = (S * B + 2047) / 2048;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int)want;
}

}  // namespace

extern "C" {

size_t hpvg_reduce_ws_bytes() { return (size_t)RED_BLOCKS_MAX * sizeof(double); }
size_t hpvg_bn_ws_bytes(int C) { return (size_t)C * 64 * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float); }

// mean / invstd / (scale, shift) of BatchNorm in train mode + running-stat update
int hpvg_bn_train_stats_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, void* ws,
                            size_t ws_bytes, int B, int C, long S, void* stream) {
  if (!x || !mean || !invstd || !scale || !shift || !ws || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_bn_ws_bytes(C)) return HPVG_ERR_WORKSPACE;
  const int ns = bn_nsplit(B, C, S);
  hipStream_t s = (hipStream_t)stream;
  switch (hpvg_vec_width(x, S)) {
    case 4: hipLaunchKernelGGL(bn_stats_partial_kernel<4>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws); break;
    case 2: hipLaunchKernelGGL(bn_stats_partial_kernel<2>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws); break;
    default: hipLaunchKernelGGL(bn_stats_partial_kernel<1>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)ws, ns, C, (double)B * (double)S,
                     eps, momentum, gamma, beta, running_mean, running_var, mean, invstd, scale, shift);
  return hpvg_launch_status();
}

// h = lrelu?(BatchNorm_train(x)) in two launches: partial sums, then the apply kernel that finalizes its own channel
// (statistics out, running-stat update) - hpvg_bn_train_stats_f32 + hpvg_affine_act_f32 without the finalize launch.
int hpvg_bn_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, float* y, int lrelu,
                          int groups, void* ws, size_t ws_bytes, int B, int C, long S, void* stream) {
  // groups > 1: the batch is `groups` independent passes of B / groups samples, normalised separately (running statistics
  // updated once per group, in order); the statistics arrays of group g start 4 * C floats after those of group g - 1
  // (layout [groups][mean, invstd, scale, shift][C])
  if (!x || !y || !mean || !invstd || !scale || !shift || !ws || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  if (groups < 1 || B % groups) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_bn_ws_bytes(C * groups)) return HPVG_ERR_WORKSPACE;
  const int Bg = B / groups;
  const int gstride = 4 * C;
  const int ns = bn_nsplit(Bg, C, S);
  hipStream_t s = (hipStream_t)stream;
  if ((long)B * C * S > HPVG_BN_FUSE_MAX) {
    // large tensors: tens of thousands of apply workgroups would each repeat the finalize prologue - three launches (per
    // group) win
    for (int g = 0; g < groups; ++g) {
      const float* xg = x + (long)g * Bg * C * S;
      float* yg = y + (long)g * Bg * C * S;
      switch (hpvg_vec_width(xg, S)) {
        case 4: hipLaunchKernelGGL(bn_stats_partial_kernel<4>, dim3(ns, C), dim3(256), 0, s, xg, Bg, C, S, ns, (double*)ws); break;
        case 2: hipLaunchKernelGGL(bn_stats_partial_kernel<2>, dim3(ns, C), dim3(256), 0, s, xg, Bg, C, S, ns, (double*)ws); break;
        default: hipLaunchKernelGGL(bn_stats_partial_kernel<1>, dim3(ns, C), dim3(256), 0, s, xg, Bg, C, S, ns, (double*)ws);
      }
      hipLaunchKernelGGL(bn_finalize_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)ws, ns, C, (double)Bg * (double)S,
                         eps, momentum, gamma, beta, running_mean, running_var, mean + g * gstride, invstd + g * gstride,
                         scale + g * gstride, shift + g * gstride);
      int nbx = hpvg_cdiv(S, 256 * 4);
      if (nbx > 1024) nbx = 1024;
      hipLaunchKernelGGL(affine_act_kernel, dim3(nbx, Bg * C), dim3(256), 0, s, xg, (const float*)(scale + g * gstride),
                         (const float*)(shift + g * gstride), yg, C, S, lrelu);
    }
    return hpvg_launch_status();
  }
  const long gsz = (long)Bg * C * S;
  int vw = 4;
  for (int g = 0; g < groups; ++g) { const int v = hpvg_vec_width(x + g * gsz, S); if (v < vw) vw = v; }
  switch (vw) {
    case 4: hipLaunchKernelGGL(bn_stats_partial_kernel<4>, dim3(ns, C * groups), dim3(256), 0, s, x, Bg, C, S, ns, (double*)ws); break;
    case 2: hipLaunchKernelGGL(bn_stats_partial_kernel<2>, dim3(ns, C * groups), dim3(256), 0, s, x, Bg, C, S, ns, (double*)ws); break;
    default: hipLaunchKernelGGL(bn_stats_partial_kernel<1>, dim3(ns, C * groups), dim3(256), 0, s, x, Bg, C, S, ns, (double*)ws);
  }
  const int nbx = hpvg_cdiv(S, 256 * 16);
  hipLaunchKernelGGL(bn_apply_fused_kernel, dim3(nbx, B * C), dim3(256), 0, s, x, (const double*)ws, ns, (double)Bg * (double)S, eps,
                     momentum, gamma, beta, running_mean, running_var, mean, invstd, scale, shift, y, C, S, lrelu, Bg, groups,
                     gstride);
  return hpvg_launch_status();
}

// ---- BatchNorm with the batch split over ranks (multi-GPU): rank-local sums -> caller all-reduces -> finalize / apply
// sums[c] = (sum x, sum x^2) over this rank's part of the batch, double
int hpvg_bn_sums_f32(const float* x, double* sums, void* ws, size_t ws_bytes, int B, int C, long S, void* stream) {
  if (!x || !sums || !ws || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_bn_ws_bytes(C)) return HPVG_ERR_WORKSPACE;
  const int ns = bn_nsplit(B, C, S);
  hipStream_t s = (hipStream_t)stream;
  switch (hpvg_vec_width(x, S)) {
    case 4: hipLaunchKernelGGL(bn_stats_partial_kernel<4>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws); break;
    case 2: hipLaunchKernelGGL(bn_stats_partial_kernel<2>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws); break;
    default: hipLaunchKernelGGL(bn_stats_partial_kernel<1>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws);
  }
  hipLaunchKernelGGL(bn_sum_partials_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)ws, ns, C, sums);
  return hpvg_launch_status();
}
// statistics from (all-reduced) sums over `count` elements per channel; same outputs as hpvg_bn_train_stats_f32
int hpvg_bn_finalize_f32(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                         int C, void* stream) {
  if (!sums || !mean || !invstd || !scale || !shift || C < 1 || !(count >= 1.0)) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, 1, C, count, eps, momentum,
                     gamma, beta, running_mean, running_var, mean, invstd, scale, shift);
  return hpvg_launch_status();
}
// sums[c] = (sum dz, sum dz*xhat) over this rank's part of the batch (dz = dh * LeakyReLU'(z)), double
int hpvg_bn_act_bwd_sums_f32(const float* dh, const float* r, const float* mean, const float* invstd, const float* scale,
                             const float* shift, int lrelu, double* sums, void* ws, size_t ws_bytes, int B, int C, long S,
                             void* stream) {
  if (!dh || !r || !mean || !invstd || !scale || !shift || !sums || !ws) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_bn_ws_bytes(C)) return HPVG_ERR_WORKSPACE;
  const int ns = bn_nsplit(B, C, S);
  hipStream_t s = (hipStream_t)stream;
  {
    const int vw = hpvg_vec_width(dh, S) < hpvg_vec_width(r, S) ? hpvg_vec_width(dh, S) : hpvg_vec_width(r, S);
    if (vw == 4) hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<4>, dim3(ns, C), dim3(256), 0, s, dh, r, mean, invstd, scale, shift, B, C, S, ns, lrelu, (double*)ws, 0);
    else if (vw == 2) hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<2>, dim3(ns, C), dim3(256), 0, s, dh, r, mean, invstd, scale, shift, B, C, S, ns, lrelu, (double*)ws, 0);
    else hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<1>, dim3(ns, C), dim3(256), 0, s, dh, r, mean, invstd, scale, shift, B, C, S, ns, lrelu, (double*)ws, 0);
  }
  hipLaunchKernelGGL(bn_sum_partials_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)ws, ns, C, sums);
  return hpvg_launch_status();
}
// dr from the GLOBAL sums (float pairs) and 1/(global element count per channel)
int hpvg_bn_act_bwd_apply_f32(const float* dh, const float* r, const float* mean, const float* invstd, const float* scale,
                              const float* shift, int lrelu, const float* sums, float inv_count, float* dr, int B, int C, long S,
                              void* stream) {
  if (!dh || !r || !mean || !invstd || !scale || !shift || !sums || !dr || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  int nbx = hpvg_cdiv(S, 256 * 4);
  if (nbx > 1024) nbx = 1024;
  hipLaunchKernelGGL(bn_lrelu_bwd_apply_kernel, dim3(nbx, B * C), dim3(256), 0, (hipStream_t)stream, dh, r, mean, invstd, scale,
                     shift, sums, dr, C, S, inv_count, lrelu);
  return hpvg_launch_status();
}

int hpvg_affine_act_f32(const float* x, const float* scale, const float* shift, float* y, int lrelu, int B, int C, long S,
                        void* stream) {
  if (!x || !scale || !shift || !y || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  int nbx = hpvg_cdiv(S, 256 * 4);
  if (nbx > 1024) nbx = 1024;
  hipLaunchKernelGGL(affine_act_kernel, dim3(nbx, B * C), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, C, S, lrelu);
  return hpvg_launch_status();
}

// dr, dgamma, dbeta of  h = lrelu?(BN_train(r))  given dh
int hpvg_bn_act_bwd_f32(const float* dh, const float* r, const float* mean, const float* invstd, const float* scale,
                        const float* shift, int lrelu, int groups, float* dr, float* dgamma, float* dbeta, int accumulate, void* ws,
                        size_t ws_bytes, int B, int C, long S, void* stream) {
  // groups: as in hpvg_bn_train_fwd_f32 (statistics layout [groups][4][C]); dgamma / dbeta are summed over the groups
  if (!dh || !r || !mean || !invstd || !scale || !shift || !dr || !dgamma || !dbeta || !ws) return HPVG_ERR_ARG;
  if (groups < 1 || B % groups) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_bn_ws_bytes(C * groups)) return HPVG_ERR_WORKSPACE;
  const int Bg = B / groups;
  const int gstride = 4 * C;
  const int ns = bn_nsplit(Bg, C, S);
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)ws;
  const long gsz = (long)Bg * C * S;
  if ((long)B * C * S > HPVG_BN_FUSE_MAX) {
    float* sums = (float*)((char*)ws + (size_t)C * groups * 64 * 2 * sizeof(double));
    for (int g = 0; g < groups; ++g) {
      const float* dhg = dh + g * gsz;
      const float* rg = r + g * gsz;
      const float *mg = mean + g * gstride, *ig = invstd + g * gstride, *sg = scale + g * gstride, *fg = shift + g * gstride;
      const int vw = hpvg_vec_width(dhg, S) < hpvg_vec_width(rg, S) ? hpvg_vec_width(dhg, S) : hpvg_vec_width(rg, S);
      if (vw == 4) hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<4>, dim3(ns, C), dim3(256), 0, s, dhg, rg, mg, ig, sg, fg, Bg, C, S, ns, lrelu, part, 0);
      else if (vw == 2) hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<2>, dim3(ns, C), dim3(256), 0, s, dhg, rg, mg, ig, sg, fg, Bg, C, S, ns, lrelu, part, 0);
      else hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<1>, dim3(ns, C), dim3(256), 0, s, dhg, rg, mg, ig, sg, fg, Bg, C, S, ns, lrelu, part, 0);
      hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)part, ns, C, dgamma, dbeta,
                         sums, (accumulate || g > 0) ? 1 : 0);
      int nbx = hpvg_cdiv(S, 256 * 4);
      if (nbx > 1024) nbx = 1024;
      hipLaunchKernelGGL(bn_lrelu_bwd_apply_kernel, dim3(nbx, Bg * C), dim3(256), 0, s, dhg, rg, mg, ig, sg, fg,
                         (const float*)sums, dr + g * gsz, C, S, (float)(1.0 / ((double)Bg * (double)S)), lrelu);
    }
    return hpvg_launch_status();
  }
  int vw = 4;
  for (int g = 0; g < groups; ++g) {
    const int v = hpvg_vec_width(dh + g * gsz, S) < hpvg_vec_width(r + g * gsz, S) ? hpvg_vec_width(dh + g * gsz, S) : hpvg_vec_width(r + g * gsz, S);
    if (v < vw) vw = v;
  }
  if (vw == 4) hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<4>, dim3(ns, C * groups), dim3(256), 0, s, dh, r, mean, invstd, scale, shift, Bg, C, S, ns, lrelu, part, gstride);
  else if (vw == 2) hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<2>, dim3(ns, C * groups), dim3(256), 0, s, dh, r, mean, invstd, scale, shift, Bg, C, S, ns, lrelu, part, gstride);
  else hipLaunchKernelGGL(bn_lrelu_bwd_reduce_kernel<1>, dim3(ns, C * groups), dim3(256), 0, s, dh, r, mean, invstd, scale, shift, Bg, C, S, ns, lrelu, part, gstride);
  const int nbx = hpvg_cdiv(S, 256 * 16);
  hipLaunchKernelGGL(bn_lrelu_bwd_apply_fused_kernel, dim3(nbx, B * C), dim3(256), 0, s, dh, r, mean, invstd, scale, shift,
                     (const double*)part, ns, dgamma, dbeta, accumulate, dr, C, S, (float)(1.0 / ((double)Bg * (double)S)), lrelu, Bg,
                     groups, gstride);
  return hpvg_launch_status();
}

// second-order backward of h = LeakyReLU_opt(BN_train(r)) (see bn_lrelu_bwd2_reduce_kernel): g = dL/d(dr) of the first-order
// backward; outputs (each optional): g_dh = dL/d(dh), g_r = dL/d(r), g_gamma = dL/d(gamma) (+= when accumulate).
size_t hpvg_bn_bwd2_ws_bytes(int C) { return (size_t)C * 64 * 5 * sizeof(double) + (size_t)C * 5 * sizeof(float); }
int hpvg_bn_act_bwd2_f32(const float* dh, const float* g, const float* r, const float* mean, const float* invstd, const float* scale,
                         const float* shift, int lrelu, float* g_dh, float* g_r, float* g_gamma, int accumulate, void* ws,
                         size_t ws_bytes, int B, int C, long S, void* stream) {
  if (!dh || !g || !r || !mean || !invstd || !scale || !shift || !ws || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_bn_bwd2_ws_bytes(C)) return HPVG_ERR_WORKSPACE;
  const int ns = bn_nsplit(B, C, S);
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)ws;
  float* consts = (float*)((char*)ws + (size_t)C * 64 * 5 * sizeof(double));
  int vw = hpvg_vec_width(dh, S);
  if (hpvg_vec_width(g, S) < vw) vw = hpvg_vec_width(g, S);
  if (hpvg_vec_width(r, S) < vw) vw = hpvg_vec_width(r, S);
  if (vw == 4) hipLaunchKernelGGL(bn_lrelu_bwd2_reduce_kernel<4>, dim3(ns, C), dim3(256), 0, s, dh, g, r, mean, invstd, scale, shift, B, C, S, ns, lrelu, part);
  else if (vw == 2) hipLaunchKernelGGL(bn_lrelu_bwd2_reduce_kernel<2>, dim3(ns, C), dim3(256), 0, s, dh, g, r, mean, invstd, scale, shift, B, C, S, ns, lrelu, part);
  else hipLaunchKernelGGL(bn_lrelu_bwd2_reduce_kernel<1>, dim3(ns, C), dim3(256), 0, s, dh, g, r, mean, invstd, scale, shift, B, C, S, ns, lrelu, part);
  hipLaunchKernelGGL(bn_bwd2_finalize_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)part, ns, C, (double)B * (double)S,
                     invstd, consts, g_gamma, accumulate);
  if (g_dh || g_r) {
    int nbx = hpvg_cdiv(S, 256 * 4);
    if (nbx > 1024) nbx = 1024;
    hipLaunchKernelGGL(bn_lrelu_bwd2_apply_kernel, dim3(nbx, B * C), dim3(256), 0, s, dh, g, r, mean, invstd, scale, shift,
                       (const float*)consts, g_dh, g_r, C, S, lrelu);
  }
  return hpvg_launch_status();
}

int hpvg_lrelu_mask_mul_f32(const float* dy, const float* h, float* out, long n, void* stream) {
  if (!dy || !h || !out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(lrelu_mask_mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, h, out, n);
  return hpvg_launch_status();
}

int hpvg_tanh_fwd_f32(const float* x, const float* res, float* y, long n, void* stream) {
  if (!x || !y || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, res, y, n);
  return hpvg_launch_status();
}
int hpvg_tanh_bwd_f32(const float* dy, const float* y, float* dx, long n, void* stream) {
  if (!dy || !y || !dx || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
  return hpvg_launch_status();
}

int hpvg_reparam_fwd_f32(const float* mu, const float* logvar, const float* eps, float* z, long n, void* stream) {
  if (!mu || !logvar || !eps || !z || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, z, n);
  return hpvg_launch_status();
}
int hpvg_reparam_bwd_f32(const float* dz, const float* logvar, const float* eps, float* dlogvar, long n, void* stream) {
  if (!dz || !logvar || !eps || !dlogvar || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dz, logvar, eps, dlogvar, n);
  return hpvg_launch_status();
}

// out[0] = mean(-0.5*(1 + logvar - mu^2 - exp(logvar)))   modules/losses.py:7-9
int hpvg_kl_fwd_f32(const float* mu, const float* logvar, float* out, void* ws, size_t ws_bytes, long n, void* stream) {
  return reduce_scalar<RED_KL>(mu, logvar, n, 1.0 / (double)n, out, ws, ws_bytes, (hipStream_t)stream);
}
int hpvg_kl_bwd_f32(const float* gout, const float* mu, const float* logvar, float* dmu, float* dlogvar, long n, void* stream) {
  if (!gout || !mu || !logvar || !dmu || !dlogvar || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, gout, mu, logvar, dmu, dlogvar, n);
  return hpvg_launch_status();
}
// out[0] = mean((a-b)^2)   nn.MSELoss, train_video.py:355
int hpvg_mse_fwd_f32(const float* a, const float* b, float* out, void* ws, size_t ws_bytes, long n, void* stream) {
  if (!b) return HPVG_ERR_ARG;
  return reduce_scalar<RED_SQDIFF>(a, b, n, 1.0 / (double)n, out, ws, ws_bytes, (hipStream_t)stream);
}
int hpvg_mse_bwd_f32(const float* gout, const float* a, const float* b, float* da, long n, void* stream) {
  if (!gout || !a || !b || !da || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, gout, a, b, da, n);
  return hpvg_launch_status();
}
// out[0] = scale * sum(x)     (scale = +-1/n for the WGAN terms, train_video.py:170,178,194)
int hpvg_sum_scaled_f32(const float* x, float* out, double scale, void* ws, size_t ws_bytes, long n, void* stream) {
  return reduce_scalar<RED_SUM>(x, nullptr, n, scale, out, ws, ws_bytes, (hipStream_t)stream);
}
// out[0] = sum(x^2)
int hpvg_sqsum_f32(const float* x, float* out, void* ws, size_t ws_bytes, long n, void* stream) {
  return reduce_scalar<RED_SQ>(x, nullptr, n, 1.0, out, ws, ws_bytes, (hipStream_t)stream);
}
// out[i] = gout[0]*coef
int hpvg_fill_scaled_f32(const float* gout, float coef, float* out, long n, void* stream) {
  if (!gout || !out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(fill_scaled_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, gout, coef, out, n);
  return hpvg_launch_status();
}
// dst = src (a KERNEL copy: hipMemcpyAsync becomes a memcpy node in a captured hipGraph - see hpvg_upsample_linear_ac_bwd_f32);
// src == NULL: dst = 0
int hpvg_copy_f32(const float* src, float* dst, long n, void* stream) {
  if (!dst || n < 1) return HPVG_ERR_ARG;
  if (src) hipLaunchKernelGGL(copy_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  else hipLaunchKernelGGL(zero_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dst, n);
  return hpvg_launch_status();
}
// out = a + b
int hpvg_add_f32(const float* a, const float* b, float* out, long n, void* stream) {
  if (!a || !b || !out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  return hpvg_launch_status();
}
// out = x / s[0]   (s: device scalar)
int hpvg_div_scalar_f32(const float* x, const float* s, float* out, long n, void* stream) {
  if (!x || !s || !out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(div_scalar_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, s, out, n);
  return hpvg_launch_status();
}
// out = alpha*a + (1-alpha)*b   (alpha: device scalar)
int hpvg_lerp_f32(const float* a, const float* b, const float* alpha, float* out, long n, void* stream) {
  if (!a || !b || !alpha || !out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(lerp_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, out, n);
  return hpvg_launch_status();
}

// out[0] = lambda * mean_{b,s} (||g[b,:,s]||_2 - 1)^2
int hpvg_gp_fwd_f32(const float* g, float* out, float lambda, void* ws, size_t ws_bytes, int B, int C, long S, void* stream) {
  if (!g || !out || !ws || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  const long n = (long)B * S;
  int nb = hpvg_cdiv(n, 256 * 8);
  if (nb > RED_BLOCKS_MAX) nb = RED_BLOCKS_MAX;
  if (ws_bytes < (size_t)nb * sizeof(double)) return HPVG_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gp_partial_kernel, dim3(nb), dim3(256), 0, s, g, B, C, S, (double*)ws);
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, nb, (double)lambda / (double)n, out);
  return hpvg_launch_status();
}
int hpvg_gp_bwd_f32(const float* gout, const float* g, float* dg, float lambda, int B, int C, long S, void* stream) {
  if (!gout || !g || !dg || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(ew_blocks((long)B * S)), dim3(256), 0, (hipStream_t)stream, gout, g, dg, B, C, S,
                     lambda);
  return hpvg_launch_status();
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// y = resize(x) to (To,Ho,Wo), align_corners=True; if noise: yn = y + amp*noise
int hpvg_upsample_linear_ac_f32(const float* x, float* y, const float* noise, float amp, float* yn, long BC, int Ti, int Hi,
                                int Wi, int To, int Ho, int Wo, void* stream) {
  if (!x || !y || BC < 1 || Ti < 1 || Hi < 1 || Wi < 1 || To < 1 || Ho < 1 || Wo < 1) return HPVG_ERR_ARG;
  if ((noise == nullptr) != (yn == nullptr)) return HPVG_ERR_ARG;
  const long n = BC * To * Ho * Wo;
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, noise, amp, yn, BC, Ti,
                     Hi, Wi, To, Ho, Wo, ac_scale(Ti, To), ac_scale(Hi, Ho), ac_scale(Wi, Wo));
  return hpvg_launch_status();
}
// N(0,1) fill (Philox4x32-10 + Box-Muller): out[i] = value (i % 4) of block i / 4 of stream (seed, call_id, iter[0]); iter may be NULL (0)
int hpvg_normal_f32(float* out, long n, unsigned long long seed, unsigned call_id, const int* iter, void* stream) {
  if (!out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(normal_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, call_id, iter);
  return hpvg_launch_status();
}
int hpvg_uniform_f32(float* out, long n, unsigned long long seed, unsigned call_id, const int* iter, void* stream) {
  if (!out || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(uniform_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out, n, seed, call_id, iter);
  return hpvg_launch_status();
}
// y = resize(x); yn = y + amp * N(0,1) with the noise of hpvg_normal_f32(stream (seed, call_id, iter)) generated in the kernel;
// samples b < first_noisy get yn = y.  C = channels per sample (BC = batch * C).
int hpvg_upsample_linear_ac_noise_f32(const float* x, float* y, float* yn, float amp, long BC, int C, int first_noisy, int Ti, int Hi,
                                      int Wi, int To, int Ho, int Wo, unsigned long long seed, unsigned call_id, const int* iter,
                                      void* stream) {
  if (!x || !y || !yn || BC < 1 || C < 1 || Ti < 1 || Hi < 1 || Wi < 1 || To < 1 || Ho < 1 || Wo < 1) return HPVG_ERR_ARG;
  const long n = BC * To * Ho * Wo;
  hipLaunchKernelGGL(upsample_noise_fwd_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, yn, amp, BC, C,
                     first_noisy, Ti, Hi, Wi, To, Ho, Wo, ac_scale(Ti, To), ac_scale(Hi, Ho), ac_scale(Wi, Wo), seed, call_id, iter);
  return hpvg_launch_status();
}
int hpvg_upsample_linear_ac_bwd_f32(const float* dy, const float* dy2, float* dx, long BC, int Ti, int Hi, int Wi, int To, int Ho,
                                    int Wo, void* stream) {
  if (!dy || !dx || BC < 1 || Ti < 1 || Hi < 1 || Wi < 1 || To < 1 || Ho < 1 || Wo < 1) return HPVG_ERR_ARG;
  // a gather over the input voxels (upsample_bwd_kernel): every dx element is written exactly once, so there is no zero
  // fill (an earlier scatter version needed one, and a hipMemsetAsync NODE inside a captured hipGraph is not reliably
  // ordered against the kernel nodes around it on this runtime, ROCm 7.2: DESIGN.md section 4) and no float atomics
  const long nin = BC * Ti * Hi * Wi;
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ew_blocks(nin)), dim3(256), 0, (hipStream_t)stream, dy, dy2, dx, BC, Ti, Hi, Wi, To,
                     Ho, Wo, ac_scale(Ti, To), ac_scale(Hi, Ho), ac_scale(Wi, Wo));
  return hpvg_launch_status();
}

// spectral norm: optional power iteration (updates u, v in place), sigma and 1/sigma (device scalars)
// ws: Co floats
int hpvg_sn_power_iter_f32(const float* w, float* u, float* v, float* sigma, float* inv_sigma, float* uv_copy, int Co, int K,
                           int do_iter, float eps, void* ws, size_t ws_bytes, void* stream) {
  if (!w || !u || !v || !sigma || !inv_sigma || !ws || Co < 1 || Co > 1024 || K < 1) return HPVG_ERR_ARG;
  if (ws_bytes < (size_t)Co * sizeof(float)) return HPVG_ERR_WORKSPACE;
  hipLaunchKernelGGL(sn_power_iter_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, u, v, sigma, inv_sigma, Co, K, do_iter,
                     eps, (float*)ws, uv_copy);
  return hpvg_launch_status();
}
// batched over layers: grid (chunks of the largest layer, n layers)
struct SnBwdBatchArgs {
  const float* dweff[HPVG_SN_BATCH_MAX];
  const float* worig[HPVG_SN_BATCH_MAX];
  const float* uv[HPVG_SN_BATCH_MAX];      // u (Co) then v (K)
  const float* sigma[HPVG_SN_BATCH_MAX];
  float* dworig[HPVG_SN_BATCH_MAX];
  double* part[HPVG_SN_BATCH_MAX];
  int Co[HPVG_SN_BATCH_MAX], K[HPVG_SN_BATCH_MAX], accumulate[HPVG_SN_BATCH_MAX], vec[HPVG_SN_BATCH_MAX];
};
__global__ __launch_bounds__(256) void sn_bwd_dot_batch_kernel(const SnBwdBatchArgs a) {
  __shared__ double sh[4];
  const int i = blockIdx.y;
  const long n = (long)a.Co[i] * a.K[i];
  const long lo = (long)blockIdx.x * SN_CHUNK;
  if (lo >= n) return;
  const long hi = lo + SN_CHUNK < n ? lo + SN_CHUNK : n;
  const float* dweff = a.dweff[i];
  const float* worig = a.worig[i];
  double acc = 0.0;
  if (a.vec[i]) {
    const float4* a4 = reinterpret_cast<const float4*>(dweff);
    const float4* b4 = reinterpret_cast<const float4*>(worig);
#pragma unroll 4
    for (long j = (lo >> 2) + threadIdx.x; j < (hi >> 2); j += 256) {
      const float4 x = a4[j], y = b4[j];
      acc += (double)x.x * y.x + (double)x.y * y.y + (double)x.z * y.z + (double)x.w * y.w;
    }
  } else {
    for (long j = lo + threadIdx.x; j < hi; j += 256) acc += (double)dweff[j] * worig[j];
  }
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) a.part[i][blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void sn_bwd_apply_batch_kernel(const SnBwdBatchArgs a) {
  const int i = blockIdx.y;
  const int Co = a.Co[i], K = a.K[i];
  const long n = (long)Co * K;
  const long lo = (long)blockIdx.x * SN_CHUNK;
  if (lo >= n) return;
  const long hi = lo + SN_CHUNK < n ? lo + SN_CHUNK : n;
  const int nparts = (int)((n + SN_CHUNK - 1) / SN_CHUNK);
  double dot = 0.0;
  for (int g = 0; g < nparts; ++g) dot += a.part[i][g];
  const float sg = a.sigma[i][0];
  const float coef = (float)(dot / ((double)sg * sg));
  const float* dweff = a.dweff[i];
  const float* u = a.uv[i];
  const float* v = a.uv[i] + Co;
  float* dworig = a.dworig[i];
  const int accumulate = a.accumulate[i];
  if (a.vec[i]) {
    const float4* a4 = reinterpret_cast<const float4*>(dweff);
    const float4* v4 = reinterpret_cast<const float4*>(v);
    float4* o4 = reinterpret_cast<float4*>(dworig);
    const int K4 = K >> 2;
#pragma unroll 4
    for (long j = (lo >> 2) + threadIdx.x; j < (hi >> 2); j += 256) {
      const int o = (int)(j / K4), c = (int)(j - (long)o * K4);
      const float4 x = a4[j], y = v4[c];
      const float cu = coef * u[o];
      float4 t = make_float4(x.x / sg - cu * y.x, x.y / sg - cu * y.y, x.z / sg - cu * y.z, x.w / sg - cu * y.w);
      if (accumulate) {
        const float4 p = o4[j];
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      o4[j] = t;
    }
  } else {
    for (long j = lo + threadIdx.x; j < hi; j += 256) {
      const int o = (int)(j / K), k = (int)(j - (long)o * K);
      const float t = dweff[j] / sg - coef * u[o] * v[k];
      dworig[j] = accumulate ? dworig[j] + t : t;
    }
  }
}

size_t hpvg_sn_bwd_ws_bytes(int Co, int K) { return (size_t)hpvg_cdiv((long)Co * K, SN_CHUNK) * sizeof(double); }
// n <= HPVG_SN_BATCH_MAX layers at once; arrays of n device pointers / sizes on the HOST (copied into the kernel arguments)
// sig[i]: 2 floats (sigma, 1/sigma); uv_copy[i]: Co+K floats or NULL; w_eff[i]: Co*K floats; ws: sum(Co) floats
int hpvg_sn_power_iter_batch_f32(int n, const float* const* w, float* const* u, float* const* v, float* const* sig,
                                 float* const* uv_copy, float* const* w_eff, const int* Co, const int* K, int do_iter, float eps,
                                 void* ws, size_t ws_bytes, void* stream) {
  if (n < 1 || n > HPVG_SN_BATCH_MAX || !w || !u || !v || !sig || !w_eff || !Co || !K || !ws) return HPVG_ERR_ARG;
  SnBatchArgs a;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    if (!w[i] || !u[i] || !v[i] || !sig[i] || !w_eff[i] || Co[i] < 1 || Co[i] > 1024 || K[i] < 1) return HPVG_ERR_ARG;
    a.w[i] = w[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.sig[i] = sig[i]; a.uv_copy[i] = uv_copy ? uv_copy[i] : nullptr;
    a.w_eff[i] = w_eff[i]; a.Co[i] = Co[i]; a.K[i] = K[i];
    a.wv_ws[i] = (float*)ws + off;
    off += (size_t)Co[i];
  }
  if (ws_bytes < off * sizeof(float)) return HPVG_ERR_WORKSPACE;
  hipLaunchKernelGGL(sn_power_iter_batch_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, a, do_iter, eps);
  return hpvg_launch_status();
}
// the backward of n <= HPVG_SN_BATCH_MAX layers in two launches; host arrays as in hpvg_sn_power_iter_batch_f32; uv[i] = the
// (u, v) copy of the forward (Co + K floats); ws: sum over layers of hpvg_sn_bwd_ws_bytes
int hpvg_sn_bwd_batch_f32(int n, const float* const* dweff, const float* const* worig, const float* const* uv,
                          const float* const* sigma, float* const* dworig, const int* accumulate, const int* Co, const int* K,
                          void* ws, size_t ws_bytes, void* stream) {
  if (n < 1 || n > HPVG_SN_BATCH_MAX || !dweff || !worig || !uv || !sigma || !dworig || !accumulate || !Co || !K || !ws)
    return HPVG_ERR_ARG;
  SnBwdBatchArgs a;
  size_t off = 0;
  int gmax = 1;
  for (int i = 0; i < n; ++i) {
    if (!dweff[i] || !worig[i] || !uv[i] || !sigma[i] || !dworig[i] || Co[i] < 1 || K[i] < 1) return HPVG_ERR_ARG;
    const int G = (int)hpvg_cdiv((long)Co[i] * K[i], SN_CHUNK);
    if (G > gmax) gmax = G;
    a.dweff[i] = dweff[i]; a.worig[i] = worig[i]; a.uv[i] = uv[i]; a.sigma[i] = sigma[i]; a.dworig[i] = dworig[i];
    a.Co[i] = Co[i]; a.K[i] = K[i]; a.accumulate[i] = accumulate[i];
    a.vec[i] = (K[i] & 3) == 0 && (Co[i] & 3) == 0 &&
               (((size_t)dweff[i] | (size_t)worig[i] | (size_t)uv[i] | (size_t)dworig[i]) & 15) == 0;
    a.part[i] = (double*)((char*)ws + off);
    off += (size_t)G * sizeof(double);
  }
  if (ws_bytes < off) return HPVG_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_bwd_dot_batch_kernel, dim3(gmax, n), dim3(256), 0, s, a);
  hipLaunchKernelGGL(sn_bwd_apply_batch_kernel, dim3(gmax, n), dim3(256), 0, s, a);
  return hpvg_launch_status();
}
int hpvg_sn_bwd_f32(const float* dweff, const float* worig, const float* u, const float* v, const float* sigma, float* dworig,
                    int accumulate, void* ws, size_t ws_bytes, int Co, int K, void* stream) {
  if (!dweff || !worig || !u || !v || !sigma || !dworig || !ws || Co < 1 || K < 1) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_sn_bwd_ws_bytes(Co, K)) return HPVG_ERR_WORKSPACE;
  const long n = (long)Co * K;
  const int G = (int)hpvg_cdiv(n, SN_CHUNK);
  const int vec = (K & 3) == 0 && (((size_t)dweff | (size_t)worig | (size_t)v | (size_t)dworig) & 15) == 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(G), dim3(256), 0, s, dweff, worig, n, vec, (double*)ws);
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(G), dim3(256), 0, s, dweff, u, v, sigma, (const double*)ws, G, dworig, n, K, vec,
                     accumulate);
  return hpvg_launch_status();
}

// g *= min(1, max_norm/(sqrt(sqsum[0]) + 1e-6)); coef_out (optional, 2 floats) = {coef, total_norm}
int hpvg_clip_scale_f32(float* g, long n, const float* sqsum, float max_norm, float* coef_out, void* stream) {
  if (!g || !sqsum || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(clip_scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, n, sqsum, max_norm, coef_out);
  return hpvg_launch_status();
}
// one Adam step over a flat range; step >= 1 is the 1-based step count, or (step_dev != NULL) read from device memory
int hpvg_adam_step_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                       int step, const int* step_dev, void* stream) {
  if (!p || !g || !m || !v || n < 1 || (!step_dev && step < 1)) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     step, step_dev);
  return hpvg_launch_status();
}
// counter[0] += 1 (device-resident optimizer step count)
int hpvg_counter_inc_i32(int* counter, void* stream) {
  if (!counter) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter);
  return hpvg_launch_status();
}

}  // extern "C"
