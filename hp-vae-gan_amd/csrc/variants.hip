// Kernels of the reference's variant models (no trainer of the reference drives them; built for completeness of the module
// surface): Encode3DVAE_nb / Encode2DVAE_nb (modules/networks_3d.py:110-138, networks_2d.py:115-143: sigmoid gate on the
// features, global average pooling of mu / logvar), GeneratorVAE_nb (networks_3d.py:409-485: latent = normal code x relaxed
// Bernoulli map), reparameterize_bern (networks_3d.py:38-45), kl_bern_criterion (modules/losses.py:12-14).
// All tensors fp32 contiguous [B][C][S] (S = T*H*W); these layers only exist at the coarsest scale (S <= a few thousand),
// so the kernels are simple: one thread per (b, s) column or one workgroup per (b, c) row.
#include "hpvg_common.h"
#include "hpvg.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// bern[b][s] = sigmoid(logit[b][s]);  out[b][c][s] = bern[b][s] * f[b][c][s]       (networks_3d.py:131-133)
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ f, const float* __restrict__ logit,
                                                        float* __restrict__ out, float* __restrict__ bern, int B, int C, long S) {
  const long n = (long)B * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / S, s = i - b * S;
    const float g = sigmoidf_(logit[i]);
    bern[i] = g;
    for (int c = 0; c < C; ++c) out[(b * C + c) * S + s] = g * f[(b * C + c) * S + s];
  }
}
// df = dout * bern;  dlogit = (sum_c dout*f + dbern_ext) * bern * (1 - bern)     (dbern_ext: gradient arriving at bern itself)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ f,
                                                        const float* __restrict__ bern, const float* __restrict__ dbern_ext,
                                                        float* __restrict__ df, float* __restrict__ dlogit, int B, int C, long S) {
  const long n = (long)B * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / S, s = i - b * S;
    const float g = bern[i];
    float acc = dbern_ext ? dbern_ext[i] : 0.f;
    for (int c = 0; c < C; ++c) {
      const long o = (b * C + c) * S + s;
      const float d = dout ? dout[o] : 0.f;
      acc += d * f[o];
      if (df) df[o] = d * g;
    }
    if (dlogit) dlogit[i] = acc * g * (1.f - g);
  }
}

// out[b][c] = scale * sum_s x[b][c][s] * (w ? w[b][s] : 1)     one workgroup per (b, c)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out,
                                                      int C, long S, float scale) {
  __shared__ double sh[4];
  const int bc = blockIdx.x;
  const int b = bc / C;
  const float* xp = x + (long)bc * S;
  const float* wp = w ? w + (long)b * S : nullptr;
  double acc = 0.0;
  for (long s = threadIdx.x; s < S; s += 256) acc += (double)(wp ? xp[s] * wp[s] : xp[s]);
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) out[bc] = (float)(tot * scale);
}
// out[b][c][s] = g[b][c] * (w ? w[b][s] : scale)      (outer product zn x zb; backward of the average pool with w = NULL)
__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out,
                                                     int C, long S, float scale, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long bc = i / S, s = i - bc * S;
    out[i] = g[bc] * (w ? w[(bc / C) * S + s] : scale);
  }
}
// out[b][s] = sum_c a[b][c][s] * v[b][c]       (gradient of the outer product w.r.t. the map)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, const float* __restrict__ v, float* __restrict__ out,
                                                      int B, int C, long S) {
  const long n = (long)B * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / S, s = i - b * S;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += a[(b * C + c) * S + s] * v[b * C + c];
    out[i] = acc;
  }
}

// z = log(x + 1e-20) - log(-log(eps + 1e-20) + 1e-20), eps ~ U(0,1)     (networks_3d.py:41-42);  dx = dz / (x + 1e-20)
__global__ __launch_bounds__(256) void reparam_bern_fwd_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                                float* __restrict__ z, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    z[i] = logf(x[i] + 1e-20f) - logf(-logf(eps[i] + 1e-20f) + 1e-20f);
}
__global__ __launch_bounds__(256) void reparam_bern_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                float* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = dz[i] / (x[i] + 1e-20f);
}

// kl_bern: mean( x*(log(x+1e-20) - log 0.5) + (1-x)*(log(1-x+1e-20) - log 0.5) )    (modules/losses.py:12-14)
__global__ __launch_bounds__(256) void kl_bern_partial_kernel(const float* __restrict__ x, long n, double* __restrict__ part) {
  __shared__ double sh[4];
  double acc = 0.0;
  const float l5 = -0.6931471805599453f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    acc += (double)(v * (logf(v + 1e-20f) - l5) + (1.f - v) * (logf(1.f - v + 1e-20f) - l5));
  }
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void sum_finish_kernel(const double* __restrict__ part, int np, double scale, float* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) acc += part[i];
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) out[0] = (float)(tot * scale);
}
// d/dx = log(x+1e-20) + x/(x+1e-20) - log(1-x+1e-20) - (1-x)/(1-x+1e-20), times gout/n
__global__ __launch_bounds__(256) void kl_bern_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                           float* __restrict__ dx, long n) {
  const float k = gout[0] / (float)n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    dx[i] = k * (logf(v + 1e-20f) + v / (v + 1e-20f) - logf(1.f - v + 1e-20f) - (1.f - v) / (1.f - v + 1e-20f));
  }
}

inline int blocks_for(long n) {
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int hpvg_gate_fwd_f32(const float* f, const float* logit, float* out, float* bern, int B, int C, long S, void* stream) {
  if (!f || !logit || !out || !bern || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(blocks_for((long)B * S)), dim3(256), 0, (hipStream_t)stream, f, logit, out, bern, B, C, S);
  return hpvg_launch_status();
}
int hpvg_gate_bwd_f32(const float* dout, const float* f, const float* bern, const float* dbern_ext, float* df, float* dlogit, int B,
                      int C, long S, void* stream) {
  if (!f || !bern || B < 1 || C < 1 || S < 1 || (!dout && !dbern_ext)) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(blocks_for((long)B * S)), dim3(256), 0, (hipStream_t)stream, dout, f, bern, dbern_ext, df,
                     dlogit, B, C, S);
  return hpvg_launch_status();
}
int hpvg_rowsum_f32(const float* x, const float* w, float* out, float scale, int B, int C, long S, void* stream) {
  if (!x || !out || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(rowsum_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, x, w, out, C, S, scale);
  return hpvg_launch_status();
}
int hpvg_outer_f32(const float* g, const float* w, float* out, float scale, int B, int C, long S, void* stream) {
  if (!g || !out || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  const long n = (long)B * C * S;
  hipLaunchKernelGGL(outer_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, g, w, out, C, S, scale, n);
  return hpvg_launch_status();
}
int hpvg_colsum_f32(const float* a, const float* v, float* out, int B, int C, long S, void* stream) {
  if (!a || !v || !out || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks_for((long)B * S)), dim3(256), 0, (hipStream_t)stream, a, v, out, B, C, S);
  return hpvg_launch_status();
}
int hpvg_reparam_bern_fwd_f32(const float* x, const float* eps, float* z, long n, void* stream) {
  if (!x || !eps || !z || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(reparam_bern_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, eps, z, n);
  return hpvg_launch_status();
}
int hpvg_reparam_bern_bwd_f32(const float* dz, const float* x, float* dx, long n, void* stream) {
  if (!dz || !x || !dx || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(reparam_bern_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dz, x, dx, n);
  return hpvg_launch_status();
}
int hpvg_kl_bern_fwd_f32(const float* x, float* out, void* ws, size_t ws_bytes, long n, void* stream) {
  if (!x || !out || !ws || n < 1) return HPVG_ERR_ARG;
  int nb = blocks_for(n);
  if (nb > 1024) nb = 1024;
  if (ws_bytes < (size_t)nb * sizeof(double)) return HPVG_ERR_WORKSPACE;
  hipLaunchKernelGGL(kl_bern_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, (double*)ws);
  hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, nb, 1.0 / (double)n, out);
  return hpvg_launch_status();
}
int hpvg_kl_bern_bwd_f32(const float* gout, const float* x, float* dx, long n, void* stream) {
  if (!gout || !x || !dx || n < 1) return HPVG_ERR_ARG;
  hipLaunchKernelGGL(kl_bern_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, gout, x, dx, n);
  return hpvg_launch_status();
}

}  // extern "C"
