// Development microbenchmark: sustained issue rate of v_mfma_f32_32x32x2_f32 on gfx950 (cycles per instruction per SIMD)
// with 1 or 2 waves per SIMD and 4..8 independent accumulators.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
  const int nwg = 256 * wgs_per_cu;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, nwg * 256 * sizeof(float));
  (void)hipMalloc(&cyc, nwg * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  mfma_loop<NACC><<<nwg, 256, wgs_per_cu == 1 ? 100 * 1024 : 60 * 1024>>>(out, iters, cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  mfma_loop<NACC><<<nwg, 256, wgs_per_cu == 1 ? 100 * 1024 : 60 * 1024>>>(out, iters, cyc);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4096];
  (void)hipMemcpy(h, cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < nwg; ++i) avg += (double)h[i];
  avg /= nwg;
  const double n_mfma = (double)iters * NACC;  // per wave
  const double flops = (double)nwg * 4 * n_mfma * 4096.0;
  printf("NACC %d  %d WG/CU: %.3f ms  %.1f TFLOP/s   cycles per MFMA per wave %.2f (per SIMD %.2f)  clock %.3f GHz\n", NACC,
         wgs_per_cu, ms, flops / ms / 1e9, avg / n_mfma, avg / n_mfma / wgs_per_cu, avg / (ms * 1e6));
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<4>(1, 20000);
  run<8>(1, 10000);
  run<4>(2, 20000);
  run<8>(2, 10000);
  return 0;
}
