// Development microbenchmark: what fits into the shadow of a v_mfma_f32_32x32x2_f32 (64 cycles on the matrix pipe) on gfx950 -
// one wave per SIMD, 16 independent accumulators, NF fillers of one kind between consecutive MFMAs (inline asm: the compiler
// neither moves nor merges them).  Prints cycles per MFMA.  hipcc --offload-arch=gfx950 -O3 tools/mfma_fillers.hip -o tools/mfma_fillers.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_add_f32   1: v_pk_add_f32   2: v_mov_b32   3: ds_read_b64   4: ds_read2_b64   5: s_add_u32 (scalar)   6: v_fma_f32
template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void loop(float* out, int iters, unsigned long long* cyc) {
  __shared__ float lds[4096];
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
  float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  f32x2 p[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
  f32x2 r2[2];
  float r4[4];
  unsigned sa = 1;
  const unsigned laddr = (unsigned)(threadIdx.x & 63) * 8u;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f[k & 7]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[k & 3]) : "v"(p[(k + 1) & 3]), "v"(p[(k + 2) & 3]));
        if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
        if (KIND == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[k & 1]) : "v"(laddr));
        if (KIND == 4) asm volatile("ds_read2_b64 %0, %1 offset0:2 offset1:3" : "=v"(*(f32x2(*)[2])r4) : "v"(laddr));
        if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa));
        if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
      }
      if (KIND == 3 || KIND == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int k = 0; k < 8; ++k) s += f[k];
  for (int k = 0; k < 4; ++k) s += p[k][0] + p[k][1];
  s += r2[0][0] + r2[1][1] + r4[0] + r4[3] + (float)sa;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NF>
void run(const char* name) {
  const int nwg = 256, iters = 2000;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, nwg * 256 * sizeof(float));
  (void)hipMalloc(&cyc, nwg * sizeof(unsigned long long));
  loop<KIND, NF><<<nwg, 256>>>(out, iters, cyc);
  (void)hipDeviceSynchronize();
  loop<KIND, NF><<<nwg, 256>>>(out, iters, cyc);
  (void)hipDeviceSynchronize();
  unsigned long long h[256];
  (void)hipMemcpy(h, cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < nwg; ++i) avg += (double)h[i];
  avg /= nwg;
  printf("%-14s x %2d per gap: %.1f cycles per MFMA\n", name, NF, avg / ((double)iters * 16));
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<0, 0>("none");
  run<0, 4>("v_add_f32"); run<0, 8>("v_add_f32"); run<0, 12>("v_add_f32"); run<0, 14>("v_add_f32"); run<0, 16>("v_add_f32"); run<0, 24>("v_add_f32");
  run<6, 8>("v_fma_f32"); run<6, 14>("v_fma_f32");
  run<1, 2>("v_pk_add_f32"); run<1, 4>("v_pk_add_f32"); run<1, 6>("v_pk_add_f32"); run<1, 8>("v_pk_add_f32");
  run<2, 8>("v_mov_b32"); run<2, 14>("v_mov_b32");
  run<3, 1>("ds_read_b64"); run<3, 2>("ds_read_b64"); run<3, 4>("ds_read_b64"); run<3, 6>("ds_read_b64");
  run<4, 1>("ds_read2_b64"); run<4, 2>("ds_read2_b64"); run<4, 4>("ds_read2_b64");
  run<5, 8>("s_add_u32"); run<5, 16>("s_add_u32"); run<5, 32>("s_add_u32");
  return 0;
}
