// hp-vae-gan_amd — shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
// Everything here is written for MI355X only: 64-wide wavefronts, 160 KiB LDS per CU,
// 256 CUs in 8 XCDs.  No other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define HPVG_OK 0
#define HPVG_ERR_ARG (-1)
#define HPVG_ERR_WORKSPACE (-2)
#define HPVG_ERR_UNSUPPORTED (-3)
#define HPVG_ERR_LAUNCH (-4)

#define HPVG_WAVE 64
#define HPVG_NUM_CU 256
#define HPVG_NUM_XCD 8

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LeakyReLU slope used everywhere on the path (reference: modules/networks_3d.py:21).
#define HPVG_LRELU_SLOPE 0.2f

static inline int hpvg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? HPVG_OK : HPVG_ERR_LAUNCH;
}

static inline int hpvg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// widest per-lane load (floats) the per-channel reduction kernels may use on rows of S floats starting at p: every row
// start ((b*C + c)*S) and every chunk start must stay aligned to it
static inline int hpvg_vec_width(const void* p, long S) {
  const uintptr_t a = (uintptr_t)p;
  if ((a & 15) == 0 && (S & 3) == 0) return 4;
  if ((a & 7) == 0 && (S & 1) == 0) return 2;
  return 1;
}

#ifdef __HIPCC__
template <int V> struct HpvgVec;
template <> struct HpvgVec<4> { typedef f32x4 type; };
template <> struct HpvgVec<2> { typedef f32x2 type; };
template <> struct HpvgVec<1> { typedef float type; };
template <int V> __device__ __forceinline__ float hpvg_vget(const typename HpvgVec<V>::type& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float hpvg_vget<1>(const float& v, int) { return v; }
__device__ __forceinline__ float hpvg_lrelu(float v) { return v > 0.f ? v : HPVG_LRELU_SLOPE * v; }

// Blocks are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Map the
// hardware block id onto a logical id so that each XCD owns one contiguous chunk of the
// logical range (neighbouring tiles share halo lines in that XCD's L2).  Bijective for any n.
__device__ __forceinline__ int hpvg_xcd_remap(int bid, int n) {
  const int q = n / HPVG_NUM_XCD, r = n % HPVG_NUM_XCD;
  const int xcd = bid % HPVG_NUM_XCD, idx = bid / HPVG_NUM_XCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// wave64 all-lanes sum (every lane ends with the total)
__device__ __forceinline__ float hpvg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double hpvg_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of doubles for blockDim.x == 256 (4 waves).  `sh` needs 4 doubles.
__device__ __forceinline__ double hpvg_block_sum_d(double v, double* sh) {
  v = hpvg_wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
#endif
