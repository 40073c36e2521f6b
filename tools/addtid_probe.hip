// Probe (development): a MUBUF load whose per-lane address comes from the buffer resource's ADD_TID_ENABLE (address = base + soffset +
// lane * stride) needs no address VGPR.  Is it correct on gfx950 (to registers and as LDS-DMA), and what does it cost between
// fp32 MFMAs against global_load_dwordx4 (tools/mfma_fillers.hip: ~40 cycles)?
// hipcc --offload-arch=gfx950 -O3 tools/addtid_probe.hip -o tools/addtid_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned stride, unsigned word3) {
  const unsigned long long b = (unsigned long long)p;
  u32x4 r;
  r[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu)) | (stride << 16);
  r[2] = 0xffffffffu;
  r[3] = word3;
  return r;
}

// mode 0: to registers; 1: LDS-DMA
__global__ __launch_bounds__(64) void check(const float* src, float* dst, unsigned word3, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.f;
  __syncthreads();
  const u32x4 rs = make_rsrc(src, 16, word3);
  if (mode == 0) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, off, %1, 0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "s"(rs) : "memory");
    for (int e = 0; e < 4; ++e) dst[threadIdx.x * 4 + e] = v[e];
  } else {
    const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, 0 lds\n\ts_waitcnt vmcnt(0)" : : "s"(rs), "s"(l) : "memory", "m0");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
  }
}

// Out-of-range lanes of a raw-buffer LDS-DMA (offen): does LDS receive zeros, or nothing?  Odd lanes get an offset past
// num_records; soffset carries part of the address (is it range-checked?).
__global__ __launch_bounds__(64) void check_oob(const float* src, float* dst, unsigned nrec, unsigned soff, unsigned oobv) {
  __shared__ __attribute__((aligned(16))) float lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -1.f;
  __syncthreads();
  u32x4 rs = make_rsrc(src, 0, 0x00027000u);
  rs[2] = nrec;
  const unsigned voff = (threadIdx.x & 1) ? oobv : threadIdx.x * 16;
  const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)soff);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %0, %2 offen lds\n\ts_waitcnt vmcnt(0)" : : "s"(rs), "s"(l), "s"(so), "v"(voff) : "memory", "m0");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
}

// LDS destination past 64 KB (the m0 field): lane i of the piece must land at lds + off + 16 i
__global__ __launch_bounds__(64) void check_far(const float* src, float* dst, unsigned off_floats, unsigned total_floats) {
  extern __shared__ __attribute__((aligned(16))) float dl[];
  for (unsigned i = threadIdx.x; i < total_floats; i += 64) dl[i] = -1.f;
  __syncthreads();
  u32x4 rs = make_rsrc(src, 0, 0x00027000u);
  rs[2] = 4096u;
  const unsigned voff = threadIdx.x * 16;
  const unsigned l = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(dl + off_floats);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %0, %2 offen lds\n\ts_waitcnt vmcnt(0)" : : "s"(rs), "s"(l), "s"(0u), "v"(voff) : "memory", "m0");
  __syncthreads();
  // where did it land?  report the first float index that is no longer -1 and how many floats at the wanted place are right
  int first = -1, right = 0;
  if (threadIdx.x == 0) {
    for (unsigned i = 0; i < total_floats; ++i)
      if (dl[i] != -1.f) { first = (int)i; break; }
    for (int i = 0; i < 256; ++i) right += dl[off_floats + i] == src[i];
    dst[0] = (float)first;
    dst[1] = (float)right;
  }
}

// cost between MFMAs: KIND 0 global_load_dwordx4 (voff + scalar base), 1 buffer_load_dwordx4 add_tid to registers, 2 buffer add_tid lds,
// 3 buffer_load_dwordx4 offen (VGPR offset) to registers, 4 none
template <int KIND>
__global__ __launch_bounds__(256, 1) void loop(float* out, int iters, unsigned long long* cyc, const float* src, unsigned word3) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
  f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float* mysrc = src + (size_t)blockIdx.x * 4096 + wv * 1024;
  const u32x4 rs = make_rsrc(mysrc, 16, word3);
  const u32x4 rs0 = make_rsrc(mysrc, 0, 0x00027000u);
  const unsigned long long bb = (unsigned long long)mysrc;
  const unsigned long long base = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb) |
                                  ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bb >> 32)) << 32);
  const unsigned voff = (threadIdx.x & 63) * 16;
  const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + wv * 1024u));
  unsigned long long mask = 0x3ffffffffull;
  asm volatile("" : "+s"(mask));
  unsigned vx = threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      if ((i & 3) == 1) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(((it * 4 + (i >> 2)) & 3) * 1024);
        if (KIND == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r4) : "v"(voff), "s"(base) : "memory");
        if (KIND == 1) asm volatile("buffer_load_dwordx4 %0, off, %1, %2" : "=v"(r4) : "s"(rs), "s"(so) : "memory");
        if (KIND == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, %2 lds" : : "s"(rs), "s"(l), "s"(so) : "memory", "m0");
        if (KIND == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r4) : "v"(voff), "s"(rs0), "s"(so) : "memory");
        if (KIND == 5) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(l), "s"(base) : "memory", "m0");
        if (KIND == 6) asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\ts_mov_b64 exec, -1" : : "v"(voff), "s"(l), "s"(base), "s"(mask) : "memory", "m0");
        if (KIND == 7) {
          unsigned long long keep_exec;
          unsigned keep_m0;
          asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 %1, m0\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                       : "=&s"(keep_exec), "=&s"(keep_m0) : "v"(voff), "s"(mask), "s"(l), "s"(base) : "memory");
        }
        if (KIND == 8) asm volatile("s_mov_b64 exec, %0\n\ts_mov_b64 exec, -1" : : "s"(mask) : "memory");
        if (KIND == 9) asm volatile("s_mov_b32 m0, %0" : : "s"(l) : "memory", "m0");
        if (KIND == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vx) : "v"(voff));
        if (KIND != 4 && KIND < 8) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  s += r4[0] + r4[3] + lds[threadIdx.x] + (float)vx;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, const float* src, unsigned word3) {
  const int nwg = 256, iters = 2000;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, nwg * 256 * sizeof(float));
  (void)hipMalloc(&cyc, nwg * sizeof(unsigned long long));
  for (int r = 0; r < 2; ++r) {
    loop<KIND><<<nwg, 256>>>(out, iters, cyc, src, word3);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[256];
  (void)hipMemcpy(h, cyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < nwg; ++i) avg += (double)h[i];
  avg /= nwg;
  printf("%-36s: %.1f cycles per MFMA (one load per 4 MFMAs: %.0f per load)\n", name, avg / ((double)iters * 16), (avg / ((double)iters * 16) - 64.0) * 4);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  const size_t n = (size_t)256 * 4096 + 8192;
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)i;
  float *src, *dst;
  (void)hipMalloc(&src, n * 4);
  (void)hipMalloc(&dst, 1024 * 4);
  (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  const unsigned words[] = {0x00827000u, 0x00807000u, 0x00800000u};
  unsigned good = 0;
  for (unsigned w3 : words)
    for (int mode = 0; mode < 2; ++mode) {
      (void)hipMemset(dst, 0xff, 1024 * 4);
      check<<<1, 64>>>(src + 64, dst, w3, mode);
      hipError_t e = hipDeviceSynchronize();
      float o[256];
      (void)hipMemcpy(o, dst, sizeof(o), hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < 256; ++i) bad += o[i] != (float)(64 + i);
      printf("word3 %08x %s: %s, %d of 256 floats wrong (first: %g %g %g %g %g)\n", w3, mode ? "lds" : "regs", hipGetErrorString(e), bad, o[0], o[1], o[4], o[5], o[8]);
      if (!bad && !good) good = w3;
    }
  {
    struct { unsigned nrec, soff, oobv; const char* what; } cases[] = {
        {4096u, 0u, 0xffffff00u, "num_records 4096, soffset 0, odd lanes at 0xffffff00"},
        {4096u, 0u, 4096u, "num_records 4096, soffset 0, odd lanes at 4096 (just past)"},
        {4096u, 256u, 0xffffff00u, "num_records 4096, soffset 256 B"},
        {1024u, 2048u, 0xffffff00u, "num_records 1024 (lanes >= 64 floats... all in range by voff), soffset 2048 B"},
        {0u, 0u, 0xffffff00u, "num_records 0 (every lane out of range)"}};
    for (auto& c : cases) {
      (void)hipMemset(dst, 0xff, 1024 * 4);
      check_oob<<<1, 64>>>(src + 64, dst, c.nrec, c.soff, c.oobv);
      hipError_t e = hipDeviceSynchronize();
      float o[256];
      (void)hipMemcpy(o, dst, sizeof(o), hipMemcpyDeviceToHost);
      int even_ok = 0, odd_zero = 0, odd_kept = 0, even_zero = 0;
      for (int ln = 0; ln < 64; ++ln)
        for (int e4 = 0; e4 < 4; ++e4) {
          const float v = o[ln * 4 + e4], want = (float)(64 + c.soff / 4 + ln * 4 + e4);
          if (ln & 1) { odd_zero += v == 0.f; odd_kept += v == -1.f; }
          else { even_ok += v == want; even_zero += v == 0.f; }
        }
      printf("oob: %s: %s; even lanes right %d / 128 (zero %d), odd lanes zero %d / 128, left untouched %d\n", c.what, hipGetErrorString(e), even_ok, even_zero, odd_zero, odd_kept);
    }
  }
  {
    (void)hipFuncSetAttribute((const void*)check_far, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned total = 160 * 1024 / 4;
    for (unsigned off : {1024u, 16000u, 16384u, 20000u, 36000u, 40000u}) {
      check_far<<<1, 64, 160 * 1024>>>(src, dst, off, total);
      hipError_t e = hipDeviceSynchronize();
      float o[2];
      (void)hipMemcpy(o, dst, sizeof(o), hipMemcpyDeviceToHost);
      printf("far: destination float %u (byte %u): %s; first float written %d, right at the wanted place %d / 256\n", off, off * 4, hipGetErrorString(e), (int)o[0], (int)o[1]);
    }
  }
  if (!good) { printf("no descriptor form gave lane * 16 addressing\n"); good = 0x00827000u; }
  run<4>("none", src, good);
  run<0>("global_load_dwordx4 voff + sbase", src, good);
  run<3>("buffer_load_dwordx4 offen", src, good);
  run<1>("buffer_load_dwordx4 add_tid", src, good);
  run<2>("buffer_load_dwordx4 add_tid lds", src, good);
  run<5>("global_load_lds_dwordx4 + m0", src, good);
  run<6>("  + exec mask, exec back by -1", src, good);
  run<7>("  + exec and m0 saved / restored", src, good);
  run<8>("exec written twice alone", src, good);
  run<9>("m0 written alone", src, good);
  run<10>("one v_add_u32 alone", src, good);
  return 0;
}
