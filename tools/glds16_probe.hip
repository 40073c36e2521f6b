// Probe (development): is a global_load_lds_dwordx4 whose per-lane GLOBAL address is only 4- or 8-byte aligned legal and
// correct on gfx950, and what does it cost against the dword form?  usage: hipcc --offload-arch=gfx950 -O3 tools/glds16_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int W16>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ dst, int off, int reps, int loff) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  float acc = 0.f;
  for (int r = 0; r < reps; ++r) {
    const float* s = src + off + (size_t)blockIdx.x * 4096 + (r & 7) * 1024;
    if (W16) {
      // 256 lanes x 16 B = 4 KB per round; 1 round
      __builtin_amdgcn_global_load_lds((gptr_t)(s + tid * 4), (lptr_t)(lds + loff + wave * 256), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(s + j * 256 + tid), (lptr_t)(lds + loff + j * 256 + wave * 64), 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    acc += lds[(tid * 7 + r) & 1023];
    __syncthreads();
  }
  // last round's image out (r = reps-1)
  const float* s = src + off + (size_t)blockIdx.x * 4096 + ((reps - 1) & 7) * 1024;
  (void)s;
  for (int i = tid; i < 1024; i += 256) dst[(size_t)blockIdx.x * 1024 + i] = lds[loff + i];
  if (acc == 12345.678f) dst[0] = acc;
}

int main() {
  const int NB = 1024;
  std::vector<float> h((size_t)NB * 4096 + 8192 + 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 100003) * 0.5f;
  float *d, *o;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&o, (size_t)NB * 1024 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> out((size_t)NB * 1024);
  for (int loff = 0; loff < 4; ++loff)
  for (int w16 = 1; w16 < 2; ++w16)
    for (int off = 0; off < 4; off += 3) {
      hipMemset(o, 0, out.size() * 4);
      const int reps = 1;
      if (w16) hipLaunchKernelGGL(probe<1>, dim3(NB), dim3(256), 4096 + 64, 0, d, o, off, reps, loff);
      else hipLaunchKernelGGL(probe<0>, dim3(NB), dim3(256), 4096 + 64, 0, d, o, off, reps, loff);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(out.data(), o, out.size() * 4, hipMemcpyDeviceToHost);
      size_t bad = 0;
      for (int b = 0; b < NB; ++b)
        for (int i = 0; i < 1024; ++i)
          if (out[(size_t)b * 1024 + i] != h[(size_t)off + (size_t)b * 4096 + i]) ++bad;
      printf("width %2d B, source offset %d floats, LDS offset %d floats: %s, mismatches %zu\n", w16 ? 16 : 4, off, loff, hipGetErrorString(e), bad);
      // timing
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      const int R = 2000;
      hipEventRecord(e0);
      if (w16) hipLaunchKernelGGL(probe<1>, dim3(NB), dim3(256), 4096 + 64, 0, d, o, off, R, loff);
      else hipLaunchKernelGGL(probe<0>, dim3(NB), dim3(256), 4096 + 64, 0, d, o, off, R, loff);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("   %d rounds of 4 KB per workgroup: %.3f ms  (%.1f ns per round)\n", R, ms, ms * 1e6 / R / (NB / 1024.0) );
    }
  return 0;
}
