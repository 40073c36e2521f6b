// Convolution backward-weight (and bias gradient) for the 3x3 / 3x3x3 "same" convs, on the
// gfx950 fp32 matrix cores.
//
// Replaces (reference): the weight-gradient half of aten::convolution_backward reached from
//   total_loss.backward() / errD_total.backward()      train_video.py:182,200  train_image.py:193,215
// and, in the gradient-penalty double backward (modules/utils.py:14-18), the derivative of a
// backward-data conv with respect to its weight (same contraction, operands swapped by the caller).
//
//   dW[o][c][dt][dh][dw] = sum_{b,t,h,w} dY[b][o][t][h][w] * f(X)[b][c][t+dt-pt][h+dh-1][w+dw-1]
//
// GEMM view per (dt): M = o (32 per tile), N = c (32 per tile), K = output positions, two per
// v_mfma_f32_32x32x2_f32.  A workgroup owns one dt, one 64x64 (o,c) block (one 32x32 sub-block per
// wave, 9 in-plane taps -> 9 accumulator tiles = 144 VGPRs) and walks a strided list of spatial
// tiles, keeping the accumulators in registers; it writes ONE partial slab at the end, and a second
// kernel sums the slabs in a fixed order (bitwise reproducible - no float atomics).
// The dY tile and the halo'd X tile of one time plane are staged in LDS with odd row strides so
// that the 32 lanes of a half wave (32 different channels, same position) hit 32 different banks.
#include "hpvg_common.h"
#include "hpvg.h"
#include <stdlib.h>

namespace {

// source of zero padding for the LDS-DMA staging (out-of-image lanes read this word with stride 0)
__device__ const float g_wzero[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct WgradArgs {
  const float* dy;
  const float* x;
  const float* in_scale;
  const float* in_shift;
  float* part;
  int B, Cin, Cout, T, H, W;
  int Th, Tw, RS, DS, XS, QK, nth, ntw, S, ncb, nob;
  int S0;  // KT == 3: persistent slots of the outer time taps (dt = 0, 2); the centre tap has S (>= S0)
  int in_lrelu;
  float* bpart;  // conv_wgradw_kernel only (else unused): [S][nob][64] per-slot sums of dY over the slot's tiles - the bias gradient
  int order;     // conv_wgradw2_kernel only: the walk over the tiles (0 time-major, 1 plane-major)
};


// Persistent, software-pipelined: one workgroup per CU (1 wave per SIMD), two LDS tile buffers; while the MFMA
// K-loop of tile i runs out of buffer i&1, tile i+1 is register-staged into the other buffer in 8-channel batches
// issued between K-loop segments (loads of batch b are in flight during segment b and written to LDS after it).
template <int KT, int NJD, int NJX>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int oblk = wave >> 1, cblk = wave & 1;
  // Workgroup ids: time tap fastest, then the persistent slot, inside the contiguous id range hpvg_xcd_remap gives each
  // XCD.  The KT workgroups of a slot walk the SAME tile list (time-major, below), so at any moment they hold the same
  // dY tile and the X planes t-1, t, t+1, which the neighbouring slots (tiles t-1, t+1 of the same spatial position)
  // need as well: each plane is fetched from HBM once and hits in the XCD's L2 for its other readers.
  // The outer taps (dt = 0, 2) have no input plane for 1 of the T output planes, i.e. (T-1)/T of the centre tap's work:
  // they get S0 < S slots, so that every workgroup of the launch ends at the same time (with equal slot counts the
  // outer-tap workgroups idled 1/T of the launch: 8 % at T = 13, 25 % at T = 4).  Ids: slots < S0 carry all three taps
  // (tap fastest), the centre tap's extra slots follow.
  const int nz = a.nob * a.ncb;
  const int L = hpvg_xcd_remap(blockIdx.x, gridDim.x);
  const int Stot = KT == 3 ? 2 * a.S0 + a.S : a.S;
  const int idx = L % Stot;
  const int z = L / Stot;
  int dt = 0, slot = idx, nslot = a.S;
  if (KT == 3) {
    if (idx < 3 * a.S0) {
      dt = idx % 3;
      slot = idx / 3;
      nslot = dt == 1 ? a.S : a.S0;
    } else {
      dt = 1;
      slot = a.S0 + idx - 3 * a.S0;
    }
  }
  const int ob = z / a.ncb, cb = z % a.ncb;
  const int RS = a.RS, DS = a.DS, XS = a.XS;
  const int BUF = 64 * (DS + XS);
  const long HW = (long)a.H * a.W;
  const long cstride = (long)a.T * HW;
  const bool active = (ob * 64 + oblk * 32 < a.Cout) && (cb * 64 + cblk * 32 < a.Cin);
  int no = a.Cout - ob * 64; if (no > 64) no = 64;   // channels present in this 64-block
  int nc = a.Cin - cb * 64;  if (nc > 64) nc = 64;

  // every time tap walks all B*T*nth*ntw tiles; where the input plane t + dt - pt does not exist the X rows are read
  // from the zero word (those products are zero; the taps 0 and KT-1 have that slack: their lists used to be shorter)
  const int pt = (KT == 3 ? 1 : 0);
  const int ntiles = a.B * a.T * a.nth * a.ntw;

  f32x16 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

  // zero both buffers once: rows of absent channels are never written afterwards
  for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 0.f;

  // ---- staging of the NEXT tile by LDS-DMA (global_load_lds_dword: no VGPR round trip, no ds_write pass).
  // One wave instruction moves 64 consecutive positions of one channel row; lanes past the row end are masked
  // off, out-of-image positions read a global zero word instead.  The DMA instructions are issued BETWEEN the
  // MFMAs of the running K loop (one channel per 6 MFMAs), so they ride in the matrix pipe's shadow.
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // per-lane staging state: source pointer of the current channel row (or the zero word) and its per-channel byte
  // stride (0 for padded positions), one pair per 256-position slot; lanes past the row end are masked off
  const char* dptr[NJD];
  const char* xptr[NJX];
  unsigned dstr[NJD], xstr[NJX];
  bool dln[NJD], xln[NJX];
#pragma unroll
  for (int j = 0; j < NJD; ++j) dln[j] = j * 256 + tid < DS;
#pragma unroll
  for (int j = 0; j < NJX; ++j) xln[j] = j * 256 + tid < XS;
  const unsigned cbytes = (unsigned)(cstride * 4);
  auto setup = [&](int tile) {
    const int t = tile % a.T;  // time-major tile order
    int r = tile / a.T;
    const int tw_i = r % a.ntw;
    r /= a.ntw;
    const int th_i = r % a.nth;
    const int b = r / a.nth;
    const int tt = t + dt - pt;
    const bool tok = tt >= 0 && tt < a.T;
    const int h0 = th_i * a.Th, w0 = tw_i * a.Tw;
    const float* dyb = a.dy + (((long)b * a.Cout + ob * 64) * a.T + t) * HW;
    const float* xb = a.x + (((long)b * a.Cin + cb * 64) * a.T + (tok ? tt : 0)) * HW;
#pragma unroll
    for (int j = 0; j < NJD; ++j) {
      const int p = j * 256 + tid;
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh, gw = w0 + ww;
      const bool ok = hh < a.Th && ww < a.Tw && gh < a.H && gw < a.W;
      dptr[j] = ok ? (const char*)(dyb + gh * a.W + gw) : (const char*)g_wzero;
      dstr[j] = ok ? cbytes : 0u;
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int p = j * 256 + tid;
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh - 1, gw = w0 + ww - 1;
      const bool ok = tok && hh < a.Th + 2 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      xptr[j] = ok ? (const char*)(xb + gh * a.W + gw) : (const char*)g_wzero;
      xstr[j] = ok ? cbytes : 0u;
    }
  };
  // stage the next channel row pair (channels are staged strictly in order 0,1,2,...): dY row c and X row c
  float* dma_d = lds;   // LDS destinations of this wave for the channel being staged
  float* dma_x = lds;
  auto dma_begin = [&](float* buf) {
    dma_d = buf + wave * 64;
    dma_x = buf + 64 * DS + wave * 64;
  };
  auto dma_channel = [&](int c) {
    if (c < no) {
#pragma unroll
      for (int j = 0; j < NJD; ++j) {
        if (dln[j]) __builtin_amdgcn_global_load_lds((gptr_t)dptr[j], (lptr_t)(dma_d + j * 256), 4, 0, 0);
        dptr[j] += dstr[j];
      }
    }
    if (c < nc) {
#pragma unroll
      for (int j = 0; j < NJX; ++j) {
        if (xln[j]) __builtin_amdgcn_global_load_lds((gptr_t)xptr[j], (lptr_t)(dma_x + j * 256), 4, 0, 0);
        xptr[j] += xstr[j];
      }
    }
    dma_d += DS;
    dma_x += XS;
  };

  int tile = slot;
  __syncthreads();  // zero fill done
  if (tile < ntiles) {
    setup(tile);
    dma_begin(lds);
    for (int c = 0; c < 64; ++c) dma_channel(c);
  }
  __syncthreads();  // (waits for the DMA: pending LDS-DMA counts on vmcnt)

  const int nsteps = a.QK >> 2;  // K-loop iterations (4 positions = 2 MFMA k-steps each)
  int cur = 0;
  for (; tile < ntiles; tile += nslot) {
    const int next = tile + nslot;
    const bool have_next = next < ntiles;
    float* bufc = lds + cur * BUF;
    int cnext = 64;                // next channel to stage (64 = nothing left)
    if (have_next) {
      setup(next);
      dma_begin(lds + (cur ^ 1) * BUF);
      cnext = 0;
    }
    // a tile whose input plane t + dt - pt lies outside the clip contributes nothing: no MFMAs, only the staging of the
    // next tile (uniform per workgroup)
    const int tt_cur = tile % a.T + dt - pt;
    if (active && tt_cur >= 0 && tt_cur < a.T) {
      const float* dl = bufc + (oblk * 32 + l31) * DS + half;
      const float* xl = bufc + 64 * DS + (cblk * 32 + l31) * XS + half;
      // two register sets: the LDS reads of step st+1 are issued before the MFMAs of step st (1 wave per SIMD:
      // nothing else hides the ds_read latency)
      float pa0, pa1, pb0[9], pb1[9], qa0, qa1, qb0[9], qb1[9];
#define WG_LOAD(A0, A1, B0, B1, ST)                                   \
  {                                                                   \
    const int q0_ = (ST) * 4;                                         \
    A0 = dl[q0_];                                                     \
    A1 = dl[q0_ + 2];                                                 \
    _Pragma("unroll") for (int dh = 0; dh < 3; ++dh)                  \
        _Pragma("unroll") for (int dw = 0; dw < 3; ++dw) {            \
      B0[dh * 3 + dw] = xl[q0_ + dh * RS + dw];                       \
      B1[dh * 3 + dw] = xl[q0_ + 2 + dh * RS + dw];                   \
    }                                                                 \
  }
// 18 MFMAs with three channels of DMA staging slotted in between (rides in the matrix pipe's shadow)
#define WG_MMA(A0, A1, B0, B1)                                                                                     \
  {                                                                                                                \
    _Pragma("unroll") for (int k = 0; k < 6; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0[k], acc[k], 0, 0, 0); \
    if (cnext < 64) { dma_channel(cnext); ++cnext; }                                                               \
    _Pragma("unroll") for (int k = 6; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0[k], acc[k], 0, 0, 0); \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1[k], acc[k], 0, 0, 0); \
    if (cnext < 64) { dma_channel(cnext); ++cnext; }                                                               \
    _Pragma("unroll") for (int k = 3; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1[k], acc[k], 0, 0, 0); \
    if (cnext < 64) { dma_channel(cnext); ++cnext; }                                                               \
  }
      int st = 0;
      if (st < nsteps) WG_LOAD(pa0, pa1, pb0, pb1, st);
      for (; st + 1 < nsteps; st += 2) {
        WG_LOAD(qa0, qa1, qb0, qb1, st + 1);
        WG_MMA(pa0, pa1, pb0, pb1);
        if (st + 2 < nsteps) WG_LOAD(pa0, pa1, pb0, pb1, st + 2);
        WG_MMA(qa0, qa1, qb0, qb1);
      }
      if (st < nsteps) WG_MMA(pa0, pa1, pb0, pb1);
#undef WG_LOAD
#undef WG_MMA
    }
    while (cnext < 64) { dma_channel(cnext); ++cnext; }  // whatever did not fit into the K loop (short loops, idle waves)
    __syncthreads();  // next buffer complete (the barrier's fence waits for the pending LDS-DMA), current one free
    cur ^= 1;
  }

  // ---- partial slab: part[s][dt][z][tap9][o64][c64]
  float* pp = a.part + ((((long)slot * KT + dt) * nz + z) * 9) * 4096;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = oblk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      pp[(long)k * 4096 + row * 64 + cblk * 32 + l31] = active ? acc[k][e] : 0.f;
    }
}

#include "conv_wgrad_wino.inl"
#include "conv_wgrad_wino2.inl"

// ------------------------------------------------------------------------------------------
// Backward-weight of the 3x3x3 conv, second generation: ONE workgroup owns all 27 taps of a 64(o) x 32(c) block.
// conv_wgrad_kernel gives each time tap its own workgroup, so a dY tile is staged by three workgroups and an X plane by
// three (2.37 GB of HBM traffic per launch against 0.49 GB algorithmic, the sharing left to the L2 and to keeping the taps
// in step) and the outer taps have less work than the centre one.  Here the sharing is structural:
//   * a workgroup walks a contiguous range of the time-major tile list (spatial tile, then t): moving from t to t+1 it
//     stages ONE new X plane (t+2) into a ring of four plane slots and the next dY tile, and runs 27 taps on them - a third
//     of the staging instructions per MFMA; planes -1 and T are a shared row of zeros in LDS (no staging, no code variants);
//   * 64 x 32 x 27 outputs = 54 accumulator tiles over 4 waves (one per SIMD, 512 registers each): wave (oblk, th) holds
//     taps th*14 .. th*14+12 of its 32 output channels and HALF of the centre tap 13 (th = 0 takes the even K steps, th = 1
//     the odd ones; the two halves are separate slab entries, summed by the reduce kernel) - 13.5 tiles each, no imbalance;
//   * the partial slab is part[slot][z][28][64][32], summed in slot order by conv_wgrad3_reduce_kernel (reproducible).
struct Wgrad3Args {
  const float* dy;
  const float* x;
  float* part;
  int B, Cin, Cout, T, H, W;
  int Th, Tw, RS, DS, XS, QK, nth, ntw, S, ncb, nob;   // ncb: 32-channel blocks of Cin
  int ntiles;                                          // B * nth * ntw * T
};

// One channel row of a staging stream by LDS-DMA, branch-free: `mask` (wave-uniform) switches off the lanes past the row
// end through EXEC inside the statement, so the piece is straight-line code that fits in the shadow of ONE MFMA (a
// compiler-generated `if (lane < n)` around the builtin is a saveexec + branch pair, and a staging piece with branches in
// it only starts after the last MFMA in front of it has issued).  The compiler does not see the load: every barrier that
// publishes staged data is preceded by an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void wg3_dma_row(const char* src, unsigned lds_addr, unsigned long long mask) {
  unsigned long long keep_exec;
  unsigned keep_m0;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b32 %1, m0\n\t"
      "s_mov_b64 exec, %3\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %2, off\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(keep_exec), "=&s"(keep_m0)
      : "v"(src), "s"(mask), "s"(lds_addr)
      : "memory");
}

__global__ __launch_bounds__(256, 1) void conv_wgrad3_kernel(const Wgrad3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int oblk = wave & 1, th = wave >> 1;
  // ids: channel-block pair fastest inside the contiguous id range an XCD owns, so the workgroups that stage the same dY
  // tiles (same slot, other input-channel half) share that XCD's L2
  const int nz = a.nob * a.ncb;
  const int L = hpvg_xcd_remap(blockIdx.x, gridDim.x);
  const int z = L % nz, slot = L / nz;
  const int ob = z / a.ncb, cb = z % a.ncb;
  const int RS = a.RS, DS = a.DS, XS = a.XS;
  const long HW = (long)a.H * a.W;
  const long cstride = (long)a.T * HW;
  int no = a.Cout - ob * 64; if (no > 64) no = 64;   // channels present in this block
  int nc = a.Cin - cb * 32;  if (nc > 32) nc = 32;
  const bool active = oblk * 32 < no;

  const int DYB = 0;                       // two dY tiles   [64][DS]        (offsets in floats)
  const int XR = 2 * 64 * DS;              // four X planes  [32][XS]
  const int ZR = XR + 4 * 32 * XS;         // XS zeros: the planes outside the clip
  for (int i = tid; i < ZR + XS; i += 256) lds[i] = 0.f;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;

  f32x16 acc[14];
#pragma unroll
  for (int k = 0; k < 14; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

  // ---- staging streams of the tile being staged: dY (64 channel rows), X plane A, X plane B (32 rows each).  Per lane: a
  // source pointer (the zero word for padded positions) and its per-channel byte stride (0 for padded positions); per
  // wave: the EXEC mask of the lanes inside the row and the LDS byte address of this wave's 64-position piece of the row.
  // A row past the block's last channel repeats that channel (finite data in a row whose outputs the reduce kernel drops).
  auto rowmask = [&](int len) __attribute__((always_inline)) -> unsigned long long {
    int n = len - wave * 64;
    n = n < 0 ? 0 : (n > 64 ? 64 : n);
    return n == 64 ? ~0ull : ((1ull << n) - 1ull);
  };
  const unsigned long long dmask = rowmask(DS), xmask = rowmask(XS);
  const unsigned cbytes = (unsigned)(cstride * 4);
  const char* dp;  unsigned dst;          // dY stream
  const char* ap;  const char* bp; unsigned xst;   // planes A, B (same lane geometry)
  unsigned d_lds = 0, a_lds = 0, b_lds = 0;        // LDS byte addresses of the next row
  int d_row = 0, a_row = 0, b_row = 0;
  auto setup = [&](int g, int pa, int pb) __attribute__((always_inline)) {
    const int t = g % a.T;
    int r = g / a.T;
    const int tw_i = r % a.ntw;
    r /= a.ntw;
    const int th_i = r % a.nth;
    const int b = r / a.nth;
    const int h0 = th_i * a.Th, w0 = tw_i * a.Tw;
    const float* dyp = a.dy + (((long)b * a.Cout + ob * 64) * a.T + t) * HW;
    const float* xp0 = a.x + ((long)b * a.Cin + cb * 32) * a.T * HW;
    {
      const int p = tid;
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh, gw = w0 + ww;
      const bool ok = hh < a.Th && ww < a.Tw && gh < a.H && gw < a.W;
      dp = ok ? (const char*)(dyp + gh * a.W + gw) : (const char*)g_wzero;
      dst = ok ? cbytes : 0u;
    }
    {
      const int p = tid;
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh - 1, gw = w0 + ww - 1;
      const bool ok = hh < a.Th + 2 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      const long off = (long)gh * a.W + gw;
      ap = (ok && pa >= 0) ? (const char*)(xp0 + pa * HW + off) : (const char*)g_wzero;
      bp = (ok && pb >= 0) ? (const char*)(xp0 + pb * HW + off) : (const char*)g_wzero;
      xst = ok ? cbytes : 0u;
    }
    d_row = a_row = b_row = 0;
  };
#define W3_STAGE_D { wg3_dma_row(dp, d_lds, dmask); ++d_row; dp += (d_row < no ? dst : 0u); d_lds += DS * 4; }
#define W3_STAGE_A { wg3_dma_row(ap, a_lds, xmask); ++a_row; ap += (a_row < nc ? xst : 0u); a_lds += XS * 4; }
#define W3_STAGE_B { wg3_dma_row(bp, b_lds, xmask); ++b_row; bp += (b_row < nc ? xst : 0u); b_lds += XS * 4; }
#define W3_STAGE_N {}

  const int g_lo = (int)((long)slot * a.ntiles / a.S), g_hi = (int)((long)(slot + 1) * a.ntiles / a.S);
  int ridx[3] = {-1, -1, -1};   // ring slots of the planes t-1, t, t+1 of the current tile (-1: outside the clip)
  int cur = 0;
  __syncthreads();  // zero fill done
  if (g_lo < g_hi) {
    // first tile of the range: everything at once (planes t and t+1 as A / B, then t-1 alone)
    const int t = g_lo % a.T;
    ridx[1] = 0;
    ridx[2] = t + 1 < a.T ? 1 : -1;
    ridx[0] = t > 0 ? 2 : -1;
    setup(g_lo, t, t + 1 < a.T ? t + 1 : -1);
    d_lds = lds0 + (DYB + wave * 64) * 4;
    a_lds = lds0 + (XR + 0 * 32 * XS + wave * 64) * 4;
    b_lds = lds0 + (XR + 1 * 32 * XS + wave * 64) * 4;
#pragma unroll 1
    for (int k = 0; k < 64; ++k) W3_STAGE_D
#pragma unroll 1
    for (int k = 0; k < 32; ++k) W3_STAGE_A
    if (t + 1 < a.T) {
#pragma unroll 1
      for (int k = 0; k < 32; ++k) W3_STAGE_B
    }
    if (t > 0) {
      setup(g_lo, t - 1, -1);
      a_lds = lds0 + (XR + 2 * 32 * XS + wave * 64) * 4;
#pragma unroll 1
      for (int k = 0; k < 32; ++k) W3_STAGE_A
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMAs above are invisible to the compiler
  __syncthreads();

  const int nsteps = a.QK >> 2;  // K-loop iterations (4 positions = 2 MFMA k-steps each)
  for (int g = g_lo; g < g_hi; ++g) {
    const int t = g % a.T;
    int rows_d = 0, rows_a = 0, rows_b = 0;   // rows of the next tile's streams still to issue (multiples of 8)
    int rnext[3] = {-1, -1, -1};
    if (g + 1 < g_hi) {
      // the two lowest ring slots not used by the current tile (it holds at most three of the four)
      unsigned freem = 15u;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (ridx[d] >= 0) freem &= ~(1u << ridx[d]);
      const int fr0 = __builtin_ctz(freem);
      const int fr1 = __builtin_ctz((freem & (freem - 1u)) | 16u);
      int pa = -1, pb = -1, sa = 0, sb = 0;
      if (t + 1 < a.T) {           // same spatial tile, next plane: slide the window
        rnext[0] = ridx[1];
        rnext[1] = ridx[2];
        if (t + 2 < a.T) { pa = t + 2; rnext[2] = fr0; sa = fr0; }
      } else {                     // next spatial tile starts at t = 0: planes 0 and 1 (the current tile holds two slots)
        pa = 0;
        rnext[1] = fr0; sa = fr0;
        if (a.T > 1) { pb = 1; rnext[2] = fr1; sb = fr1; }
      }
      setup(g + 1, pa, pb);
      d_lds = lds0 + (DYB + (cur ^ 1) * 64 * DS + wave * 64) * 4;
      a_lds = lds0 + (XR + sa * 32 * XS + wave * 64) * 4;
      b_lds = lds0 + (XR + sb * 32 * XS + wave * 64) * 4;
      rows_d = 64;
      rows_a = pa >= 0 ? 32 : 0;
      rows_b = pb >= 0 ? 32 : 0;
    }
    if (active) {
      // LDS offsets (in floats) of this lane's operand rows.  One code path for both tap halves: the 13 taps of a wave are
      // a table of per-lane offsets (plane of the tap's dt, row dh, column dw), picked by th once per tile.
      const int dlo = DYB + cur * 64 * DS + (oblk * 32 + l31) * DS + half;
      const int zo = ZR + half;
      const int xb0 = ridx[0] >= 0 ? XR + ridx[0] * 32 * XS + l31 * XS + half : zo;
      const int xb1 = ridx[1] >= 0 ? XR + ridx[1] * 32 * XS + l31 * XS + half : zo;
      const int xb2 = ridx[2] >= 0 ? XR + ridx[2] * 32 * XS + l31 * XS + half : zo;
#define W3_TAPOFS(TAP) (((TAP) / 9 == 0 ? xb0 : ((TAP) / 9 == 1 ? xb1 : xb2)) + (((TAP) % 9) / 3) * RS + (TAP) % 3)
      int xo[13];
#pragma unroll
      for (int k = 0; k < 13; ++k) xo[k] = th == 0 ? W3_TAPOFS(k) : W3_TAPOFS(14 + k);
#undef W3_TAPOFS
      const int xso = xb1 + RS + 1 + th * 4;   // the shared centre tap: th = 0 takes the even K steps, th = 1 the odd ones
      const int dso = dlo + th * 4;
      // two register sets: the LDS reads of step st+1 are issued before the MFMAs of step st (1 wave per SIMD: nothing
      // else hides the ds_read latency).  Set p runs the even steps, set q the odd ones; the centre tap's half (acc[13])
      // rides with p: its operands sa / sb are those of step st + th.
      float pa0, pa1, pb0[13], pb1[13], qa0, qa1, qb0[13], qb1[13], sa0, sa1, sb0, sb1;
#define W3_LOAD(A0, A1, B0, B1, ST)                                   \
  {                                                                   \
    const int q0_ = (ST) * 4;                                         \
    A0 = lds[dlo + q0_];                                              \
    A1 = lds[dlo + q0_ + 2];                                          \
    _Pragma("unroll") for (int k = 0; k < 13; ++k) {                  \
      B0[k] = lds[xo[k] + q0_];                                       \
      B1[k] = lds[xo[k] + q0_ + 2];                                   \
    }                                                                 \
  }
#define W3_LOADS(ST)                                                  \
  {                                                                   \
    const int q0_ = (ST) * 4;                                         \
    sa0 = lds[dso + q0_];                                             \
    sa1 = lds[dso + q0_ + 2];                                         \
    sb0 = lds[xso + q0_];                                             \
    sb1 = lds[xso + q0_ + 2];                                         \
  }
// 26 (28) MFMAs with four rows of DMA staging between them (each row piece is straight-line: one MFMA's shadow)
#define W3_MMA(A0, A1, B0, B1, SH, STG)                                                                             \
  {                                                                                                                 \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0[k], acc[k], 0, 0, 0);  \
    STG                                                                                                             \
    _Pragma("unroll") for (int k = 4; k < 13; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0[k], acc[k], 0, 0, 0); \
    if (SH) acc[13] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa0, sb0, acc[13], 0, 0, 0);                             \
    STG                                                                                                             \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1[k], acc[k], 0, 0, 0);  \
    STG                                                                                                             \
    _Pragma("unroll") for (int k = 4; k < 13; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1[k], acc[k], 0, 0, 0); \
    if (SH) acc[13] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa1, sb1, acc[13], 0, 0, 0);                             \
    STG                                                                                                             \
  }
// two K steps (sets p and q) with eight rows of one staging stream
#define W3_BODY(STG)                                                                    \
  {                                                                                     \
    W3_LOAD(qa0, qa1, qb0, qb1, st + 1);                                                \
    W3_MMA(pa0, pa1, pb0, pb1, true, STG);                                              \
    if (st + 2 < nsteps) { W3_LOAD(pa0, pa1, pb0, pb1, st + 2); W3_LOADS(st + 2); }     \
    W3_MMA(qa0, qa1, qb0, qb1, false, STG);                                             \
  }
      int st = 0;
      if (st < nsteps) { W3_LOAD(pa0, pa1, pb0, pb1, st); W3_LOADS(st); }
      // the K loop in phases, one per staging stream (the stream is then compile-time in the loop body)
      for (; st + 1 < nsteps && rows_d > 0; st += 2, rows_d -= 8) W3_BODY(W3_STAGE_D)
      for (; st + 1 < nsteps && rows_a > 0; st += 2, rows_a -= 8) W3_BODY(W3_STAGE_A)
      for (; st + 1 < nsteps && rows_b > 0; st += 2, rows_b -= 8) W3_BODY(W3_STAGE_B)
      for (; st + 1 < nsteps; st += 2) W3_BODY(W3_STAGE_N)
      if (st < nsteps) {
        // odd step count: the last step is an even one - th = 0's half of the centre tap; th = 1's would be step nsteps
        W3_MMA(pa0, pa1, pb0, pb1, false, W3_STAGE_N);
        if (th == 0) {
          acc[13] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa0, sb0, acc[13], 0, 0, 0);
          acc[13] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa1, sb1, acc[13], 0, 0, 0);
        }
      }
#undef W3_BODY
#undef W3_MMA
#undef W3_LOADS
#undef W3_LOAD
    }
    // whatever did not fit into the K loop (short loops, idle waves)
#pragma unroll 1
    for (; rows_d > 0; --rows_d) W3_STAGE_D
#pragma unroll 1
    for (; rows_a > 0; --rows_a) W3_STAGE_A
#pragma unroll 1
    for (; rows_b > 0; --rows_b) W3_STAGE_B
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): next tile landed (this wave's share) ...
    __syncthreads();                     // ... everybody's has, and the current buffers are free
    cur ^= 1;
    ridx[0] = rnext[0]; ridx[1] = rnext[1]; ridx[2] = rnext[2];
  }
#undef W3_STAGE_D
#undef W3_STAGE_A
#undef W3_STAGE_B
#undef W3_STAGE_N

  // ---- partial slab: part[slot][z][28][o64][c32]; entry th*14 + k = tap th*14 + k (k < 13), entries 13 / 27 = the two
  // halves of the centre tap
  float* pp = a.part + ((long)slot * nz + z) * 28 * 2048;
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    const int entry = k < 13 ? th * 14 + k : (th == 0 ? 13 : 27);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = oblk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      pp[(long)entry * 2048 + row * 32 + l31] = active ? acc[k][e] : 0.f;
    }
  }
}

// dW[o][c][tap27] = sum_slot part[slot][z][tap][o%64][c%32] (+ the second half of the centre tap).  A block is 256 elements
// x 4 slot groups (group g sums the slots [g*S/4, (g+1)*S/4) in order, two chains), the four group sums are added in
// group order through LDS: reproducible, and four times the loads in flight of a one-thread-per-element loop (128 slabs a
// quarter of a megabyte apart: that loop was latency-bound at 1.5 TB/s).
__global__ __launch_bounds__(1024) void conv_wgrad3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S,
                                                                  int nob, int ncb, int Cout, int Cin, int accumulate) {
  __shared__ float sm[4][256];
  const int g = threadIdx.y;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int nz = nob * ncb;
  const long total = (long)nz * 27 * 2048;
  long r = idx < total ? idx : total - 1;
  const int c32 = r % 32; r /= 32;
  const int o64 = r % 64; r /= 64;
  const int tap = r % 27; r /= 27;
  const int z = (int)r;
  const int o = (z / ncb) * 64 + o64, c = (z % ncb) * 32 + c32;
  const bool live = idx < total && o < Cout && c < Cin;
  const long per_s = (long)nz * 28 * 2048;
  const float* p0 = part + ((long)z * 28 + tap) * 2048 + o64 * 32 + c32;
  float s0 = 0.f, s1 = 0.f;
  if (live) {
    int sl = (int)((long)g * S / 4);
    const int hi = (int)((long)(g + 1) * S / 4);
    if (tap == 13) {
      for (; sl < hi; ++sl) {
        s0 += p0[(long)sl * per_s];
        s1 += p0[(long)sl * per_s + 14 * 2048];
      }
    } else {
      for (; sl + 2 <= hi; sl += 2) {
        s0 += p0[(long)sl * per_s];
        s1 += p0[(long)(sl + 1) * per_s];
      }
      if (sl < hi) s0 += p0[(long)sl * per_s];
    }
  }
  sm[g][threadIdx.x] = s0 + s1;
  __syncthreads();
  if (g == 0 && live) {
    const float tot = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    float* dst = dw + ((long)o * Cin + c) * 27 + tap;
    *dst = accumulate ? *dst + tot : tot;
  }
}

// ------------------------------------------------------------------------------------------
// Narrow backward-weight: one side of the layer has <= 4 channels (heads 3->64, tails 64->3 / 64->1).
//   R[cw][cn][tap] = sum_pos wide[cw][pos] * narrow[cn][pos + off(tap)]          (all KT*9 taps at once)
// head (Cin <= 4):  wide = dY (Cout ch), narrow = X          ->  dW[o=cw][c=cn][tap]        = R
// tail (Cout <= 4): wide = X (Cin ch),  narrow = dY (halo'd) ->  dW[o=cn][c=cw][ntaps-1-tap] = R
// GEMM: M = wide channels (2 tiles of 32 per 64-block), N = (cn, tap) pairs (<= 108 -> up to 4 tiles of 32, read
// from the halo'd narrow tile through a per-lane offset), K = positions of the tile, split over the 4 waves.
// The wide operand is streamed exactly once (HBM bound); staging is LDS-DMA, double buffered, like conv_wgrad_kernel.
struct NarrowArgs {
  const float* wide;
  const float* narrow;
  float* part;
  int B, CW, CN, T, H, W;
  int Th, Tw, RS, DS, XPL, QK, nth, ntw, S, ntiles;
};

template <int KT, int NT>
__global__ __launch_bounds__(256, 1) void conv_wgrad_narrow_kernel(const NarrowArgs a) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int TAPS = KT * 9;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wb = blockIdx.y;  // 64-channel block of the wide operand
  const int RS = a.RS, DS = a.DS, XPL = a.XPL;
  const int NROW = a.CN * KT;           // narrow planes
  const int BUF = 64 * DS + NROW * XPL;
  const long HW = (long)a.H * a.W;
  const long cstride = (long)a.T * HW;
  const unsigned cbytes = (unsigned)(cstride * 4);
  const int pt = (KT == 3 ? 1 : 0);
  int nw = a.CW - wb * 64; if (nw > 64) nw = 64;

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 0.f;

  // per-lane offset of column j = n*32 + (lane&31) -> (cn, tap) inside the halo'd narrow tile
  int loff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int j = n * 32 + l31;
    int o = 0;
    if (j < a.CN * TAPS) {
      const int cn = j / TAPS, tap = j - cn * TAPS;
      const int dt = tap / 9, r = tap - dt * 9, dh = r / 3, dw = r - dh * 3;
      o = (cn * KT + dt) * XPL + dh * RS + dw;
    }
    loff[n] = o;
  }

  // staging state (2 slots of 256 positions for the wide rows, 2 for the narrow planes)
  const char* wptr[2];
  unsigned wstr[2];
  int noff[2];
  bool nok[2];
  const bool wln[2] = {tid < DS, 256 + tid < DS};
  const bool nln[2] = {tid < XPL, 256 + tid < XPL};
  const float* nbase = a.narrow;
  int st_t = 0;
  auto setup = [&](int tile) {
    const int tw_i = tile % a.ntw;
    int r = tile / a.ntw;
    const int th_i = r % a.nth;
    r /= a.nth;
    const int t = r % a.T;
    const int b = r / a.T;
    const int h0 = th_i * a.Th, w0 = tw_i * a.Tw;
    const float* wbp = a.wide + (((long)b * a.CW + wb * 64) * a.T + t) * HW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = j * 256 + tid;
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh, gw = w0 + ww;
      const bool ok = hh < a.Th && ww < a.Tw && gh < a.H && gw < a.W;
      wptr[j] = ok ? (const char*)(wbp + gh * a.W + gw) : (const char*)g_wzero;
      wstr[j] = ok ? cbytes : 0u;
      const int gh2 = h0 + hh - 1, gw2 = w0 + ww - 1;
      nok[j] = hh < a.Th + 2 && gh2 >= 0 && gh2 < a.H && gw2 >= 0 && gw2 < a.W;
      noff[j] = nok[j] ? gh2 * a.W + gw2 : 0;
    }
    nbase = a.narrow + (long)b * a.CN * a.T * HW;
    st_t = t;
  };
  float* dma_w = lds;
  auto dma_begin = [&](float* buf) { dma_w = buf + wave * 64; };
  auto dma_wide = [&](int c) {  // rows strictly in order c = 0, 1, 2, ...
    if (c < nw) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (wln[j]) __builtin_amdgcn_global_load_lds((gptr_t)wptr[j], (lptr_t)(dma_w + j * 256), 4, 0, 0);
        wptr[j] += wstr[j];
      }
    }
    dma_w += DS;
  };
  auto dma_narrow = [&](int row, float* buf) {  // row = cn*KT + dt
    const int cn = row / KT, dt = row - cn * KT;
    const int tt = st_t + dt - pt;
    const bool valid = tt >= 0 && tt < a.T;
    const float* src = nbase + ((long)cn * a.T + (valid ? tt : 0)) * HW;
    float* dst = buf + 64 * DS + row * XPL + wave * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (nln[j]) __builtin_amdgcn_global_load_lds((gptr_t)((valid && nok[j]) ? src + noff[j] : g_wzero), (lptr_t)(dst + j * 256), 4, 0, 0);
  };

  int tile = blockIdx.x;
  __syncthreads();
  if (tile < a.ntiles) {
    setup(tile);
    dma_begin(lds);
    for (int c = 0; c < 64; ++c) dma_wide(c);
    for (int r = 0; r < NROW; ++r) dma_narrow(r, lds);
  }
  __syncthreads();

  // this wave's share of the K positions: [k_lo, k_hi) in steps of 2
  const int nk = a.QK >> 1;
  const int k_lo = (wave * nk) / 4, k_hi = ((wave + 1) * nk) / 4;
  int cur = 0;
  for (; tile < a.ntiles; tile += a.S) {
    const int next = tile + a.S;
    const bool have_next = next < a.ntiles;
    float* bufc = lds + cur * BUF;
    float* bufn = lds + (cur ^ 1) * BUF;
    int cnext = 64, rnext = NROW;
    if (have_next) {
      setup(next);
      dma_begin(bufn);
      cnext = 0;
      rnext = 0;
    }
    const float* al = bufc + l31 * DS + half;
    const float* bl = bufc + 64 * DS + half;
    for (int ks = k_lo; ks < k_hi; ++ks) {
      const int q0 = ks * 2;
      const float a0 = al[q0], a1 = al[32 * DS + q0];
      float bv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) bv[n] = bl[loff[n] + q0];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[n], acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[n], acc[1][n], 0, 0, 0);
      }
      // staging of the next tile, a few rows per k-step
#pragma unroll
      for (int r = 0; r < 6; ++r)
        if (cnext < 64) { dma_wide(cnext); ++cnext; }
      if (rnext < NROW) { dma_narrow(rnext, bufn); ++rnext; }
    }
    while (cnext < 64) { dma_wide(cnext); ++cnext; }
    while (rnext < NROW) { dma_narrow(rnext, bufn); ++rnext; }
    __syncthreads();
    cur ^= 1;
  }

  // partial fragments: part[(wg*4 + wave)][wb][m][n][e][lane]
  float* pp = a.part + ((((long)blockIdx.x * 4 + wave) * gridDim.y + wb) * 2 * NT) * 1024;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) pp[((m * NT + n) * 16 + e) * 64 + lane] = acc[m][n][e];
}


// ------------------------------------------------------------------------------------------
// Narrow backward-weight, second generation.  The first one stages the 64 wide channels of a tile in LDS by dword LDS-DMA
// (one instruction per channel and 64 positions) so that a lane can fetch "its channel at position k": 0.36 ms at stage 9,
// matrix pipe 24 % busy.  Here the wide operand never touches LDS: lane i IS channel i, it reads FOUR consecutive positions
// of its own channel row with one global_load_dwordx4 (half-wave h takes positions g+4h .. g+4h+3 of an 8-position group) and
// uses element j as the A operand of MFMA step j - the K index of a step is then the position pair {g+j, g+4+j}, which is as
// good as any other pairing as long as the B operand (read from the halo'd narrow tile in LDS) uses the same positions.
// 64 lanes touch 64 different rows, but every 128-byte line they open is consumed by the next three groups from L1/L2, and
// a load feeds 4 x 2 x NT MFMAs.  Only the narrow operand (<= 4 channels x KT planes, with halo and zero padding) is staged.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
struct Narrow2WArgs {
  const float* wide;
  const float* narrow;
  float* part;
  int B, CW, CN, T, H, W;
  int Th, nth, RS, XPL, S, ntiles;
};

template <int KT, int NT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_narrow2_kernel(const Narrow2WArgs a) {
  constexpr int TAPS = KT * 9;
  constexpr int D = 4;   // groups in flight per wave (two 16-byte loads each)
  extern __shared__ __attribute__((aligned(16))) float nl[];   // narrow tile: CN*KT planes of XPL = (Th+2)*RS floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wb = blockIdx.y;
  const int RS = a.RS, XPL = a.XPL;
  const int NROW = a.CN * KT;
  const long HW = (long)a.H * a.W;
  const int pt = (KT == 3 ? 1 : 0);
  const int W = a.W;

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  int loff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int j = n * 32 + l31;
    int o = 0;
    if (j < a.CN * TAPS) {
      const int cn = j / TAPS, tap = j - cn * TAPS;
      const int dt = tap / 9, r = tap - dt * 9, dh = r / 3, dw = r - dh * 3;
      o = (cn * KT + dt) * XPL + dh * RS + dw;
    }
    loff[n] = o;
  }
  const int c0 = wb * 64 + l31, c1 = c0 + 32;
  const bool ok0 = c0 < a.CW, ok1 = c1 < a.CW;
  const int G8 = (W + 7) / 8;                     // 8-position groups per image row

  for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S) {
    const int th = tile % a.nth;
    int r_ = tile / a.nth;
    const int t = r_ % a.T;
    const int b = r_ / a.T;
    const int h0 = th * a.Th;
    int rows = a.H - h0;
    if (rows > a.Th) rows = a.Th;
    __syncthreads();   // every wave is done reading the previous narrow tile
    // ---- narrow tile with halo and zero padding: plane (cn, dt), tile row hh <-> image row h0+hh-1, column ww <-> ww-1
    for (int row = 0; row < NROW; ++row) {
      const int cn = row / KT, dt = row - cn * KT;
      const int tt = t + dt - pt;
      const bool tok = tt >= 0 && tt < a.T;
      const float* src = a.narrow + (((long)b * a.CN + cn) * a.T + (tok ? tt : 0)) * HW;
      for (int idx = tid; idx < XPL; idx += 256) {
        const int hh = idx / RS, ww = idx - hh * RS;
        const int gh = h0 + hh - 1, gw = ww - 1;
        float v = 0.f;
        if (tok && gh >= 0 && gh < a.H && gw >= 0 && gw < W) v = src[(long)gh * W + gw];
        nl[row * XPL + idx] = v;
      }
    }
    __syncthreads();
    // ---- K loop: the tile's (row, 8-position group) items are dealt round-robin to the four waves (item = wave + 4 k), so a
    // tile of one or two rows still keeps every wave busy
    const int total = rows * G8;
    const int nit = total > wave ? (total - wave + 3) / 4 : 0;
    const float* w0 = a.wide + (((long)b * a.CW + (ok0 ? c0 : 0)) * a.T + t) * HW + (long)h0 * W;
    const float* w1 = a.wide + (((long)b * a.CW + (ok1 ? c1 : 0)) * a.T + t) * HW + (long)h0 * W;
    f32x4u A0[D], A1[D];
    auto issue = [&](int it, f32x4u& x0, f32x4u& x1) {
      const int item = wave + 4 * it;
      const int rr = item / G8, g = 8 * (item - rr * G8);
      int start = g + 4 * half;
      if (start > W - 4) start = W - 4;
      x0 = *reinterpret_cast<const f32x4u*>(w0 + (long)rr * W + start);
      x1 = *reinterpret_cast<const f32x4u*>(w1 + (long)rr * W + start);
    };
#pragma unroll
    for (int u = 0; u < D; ++u)
      if (u < nit) issue(u, A0[u], A1[u]);
    for (int it0 = 0; it0 < nit; it0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int it = it0 + u;
        if (it < nit) {
          const int item = wave + 4 * it;
          const int rr = item / G8, g = 8 * (item - rr * G8);
          const int gs = g + 4 * half;
          int start = gs;
          if (start > W - 4) start = W - 4;
          const f32x4u x0 = A0[u], x1 = A1[u];
          if (it + D < nit) issue(it + D, A0[u], A1[u]);
          const float* bl = nl + rr * RS + start;               // tap (dh, dw) of position (tile row rr, column c) sits at (rr + dh) * RS + c + dw
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int pos = start + j;
            const bool v = pos >= gs && pos < W;                // (a clamped tail group re-reads columns an earlier group owns)
            const float a0 = (v && ok0) ? x0[j] : 0.f;
            const float a1 = (v && ok1) ? x1[j] : 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const float bv = bl[loff[n] + j];
              acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][n], 0, 0, 0);
              acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][n], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // the four waves' partial sums are added up through LDS in wave order (fixed order: reproducible) and the workgroup
  // writes ONE fragment set: part[wg][wb][m][n][e][lane] (the reduce kernel sees gridDim.x partials per 64-channel block)
  __syncthreads();                       // the narrow tile is dead: its LDS becomes the exchange buffer
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) nl[((m * NT + n) * 16 + e) * 64 + lane] = acc[m][n][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[m][n][e] += nl[((m * NT + n) * 16 + e) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0) {
    float* pp = a.part + (((long)blockIdx.x * gridDim.y + wb) * 2 * NT) * 1024;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) pp[((m * NT + n) * 16 + e) * 64 + lane] = acc[m][n][e];
  }
}

// dW from the narrow kernel's fragments; mode 0: dW[o=cw][c=cn][tap]; mode 1: dW[o=cn][c=cw][ntaps-1-tap]
// one wave per output element: lanes stride over the partial slabs, fixed-order shuffle reduction (reproducible)
__global__ __launch_bounds__(256) void conv_wgrad_narrow_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                         int nparts, int nwb, int NT, int CW, int CN, int taps,
                                                                         int mode, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int total = CW * CN * taps;
  if (idx >= total) return;
  const int tap = idx % taps;
  const int cn = (idx / taps) % CN;
  const int cw = idx / (taps * CN);
  const int wb = cw / 64, c64 = cw % 64;
  const int m = c64 / 32, row = c64 % 32;
  const int hf = (row >> 2) & 1, e = (row & 3) + 4 * (row >> 3);
  const int j = cn * taps + tap;
  const int n = j / 32, col = j % 32;
  const long off = ((long)(wb * 2 * NT) + (m * NT + n)) * 1024 + e * 64 + hf * 32 + col;
  const long pstride = (long)nwb * 2 * NT * 1024;
  float s = 0.f;
  for (int p = lane; p < nparts; p += 64) s += part[(long)p * pstride + off];
  const float tot = hpvg_wave_sum(s);
  if (lane == 0) {
    float* dst = mode == 0 ? dw + ((long)cw * CN + cn) * taps + tap : dw + ((long)cn * CW + cw) * taps + (taps - 1 - tap);
    *dst = accumulate ? *dst + tot : tot;
  }
}

// dW[o][c][dt][tap9] = sum_s part[s][dt][z][tap9][o%64][c%64]; one thread per slab element, fixed order.
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S1, int S0, int KT, int nob,
                                         int ncb, int Cout, int Cin, int accumulate) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_s = (long)KT * nob * ncb * 9 * 4096;
  if (idx >= per_s) return;
  long r = idx;
  const int c64 = r % 64; r /= 64;
  const int o64 = r % 64; r /= 64;
  const int tap = r % 9; r /= 9;
  const int z = r % (nob * ncb); r /= (nob * ncb);
  const int dt = (int)r;
  const int o = (z / ncb) * 64 + o64, c = (z % ncb) * 64 + c64;
  if (o >= Cout || c >= Cin) return;
  const int S = (KT == 3 && dt != 1) ? S0 : S1;  // slots that wrote a slab for this time tap
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 4 <= S; s += 4) {
    s0 += part[(long)(s + 0) * per_s + idx];
    s1 += part[(long)(s + 1) * per_s + idx];
    s2 += part[(long)(s + 2) * per_s + idx];
    s3 += part[(long)(s + 3) * per_s + idx];
  }
  for (; s < S; ++s) s0 += part[(long)s * per_s + idx];
  const float tot = (s0 + s1) + (s2 + s3);
  float* dst = dw + (((long)o * Cin + c) * KT + dt) * 9 + tap;
  *dst = accumulate ? *dst + tot : tot;
}

// per-channel sum over (b, spatial): out[c] = sum_b sum_s x[b][c][s]   (bias gradient)
// two stages (grid (nsplit, C) partials in fp64, then one finishing thread per channel): reproducible, chip-filling
template <int V>
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x, int B, int C, long S, int nsplit,
                                                                   double* __restrict__ part, float* __restrict__ out,
                                                                   int accumulate) {
  typedef typename HpvgVec<V>::type Vec;
  __shared__ double sh[4];
  const int c = blockIdx.y, k = blockIdx.x;
  const long SV = S / V;
  const long chunk = (SV + nsplit - 1) / nsplit;
  const long lo = (long)k * chunk, hi = (lo + chunk < SV) ? lo + chunk : SV;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const Vec* p = reinterpret_cast<const Vec*>(x + ((long)b * C + c) * S);
    float loc = 0.f;
    int cnt = 0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      const Vec v = p[i];
#pragma unroll
      for (int e = 0; e < V; ++e) loc += hpvg_vget<V>(v, e);
      if (++cnt == 32 / V) { acc += loc; loc = 0.f; cnt = 0; }
    }
    acc += loc;
  }
  const double tot = hpvg_block_sum_d(acc, sh);
  if (threadIdx.x == 0) {
    if (out) out[c] = accumulate ? out[c] + (float)tot : (float)tot;  // nsplit == 1: one pass, no finishing launch
    else part[(long)c * nsplit + k] = tot;
  }
}
__global__ void channel_sum_finish_kernel(const double* __restrict__ part, int nsplit, int C, float* __restrict__ out,
                                          int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += part[(long)c * nsplit + k];
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

struct WPlan {
  int Th, Tw, RS, DS, XS, QK, nth, ntw, S, nob, ncb;
  size_t lds;
  int S0;  // slots of the outer time taps (KT == 3), <= S
  int g16; // conv_wgradw_kernel's 16-byte staging form (RS = the X row stride Tw + 8, QK = Th * Tw)
};

WPlan plan_wgrad_search(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  WPlan best{};
  double best_cost = 1e300;
  const int nob = hpvg_cdiv(Cout, 64), ncb = hpvg_cdiv(Cin, 64);
  for (int Tw = 1; Tw <= W; ++Tw) {
    const int ntw = hpvg_cdiv(W, Tw);
    if (Tw != hpvg_cdiv(W, ntw)) continue;
    const int RS = Tw + 2;
    for (int Th = 1; Th <= H; ++Th) {
      const int nth = hpvg_cdiv(H, Th);
      if (Th != hpvg_cdiv(H, nth)) continue;
      const int QK = (Th * RS + 3) & ~3;                 // K positions, padded to the 4-position loop step
      const int DS = (QK + 1) | 1;                       // dY row stride: >= QK, odd (bank spread)
      int XS = QK + 2 * RS + 4;                          // X row: reads reach QK-1 + 2*RS + 2
      if (XS < (Th + 2) * RS + 1) XS = (Th + 2) * RS + 1;
      XS |= 1;
      if (DS > 512 || XS > 512) break;
      const size_t lds = (size_t)2 * 64 * (DS + XS) * sizeof(float);  // two tile buffers
      if (lds > 156 * 1024) break;
      const long ntiles = (long)B * T * nth * ntw;
      // useful fraction of the K loop and per-tile fixed cost (barrier + pipeline segments)
      const double work = (double)ntiles * (QK * 0.5 * 9.0 + 40.0);
      if (work < best_cost) {
        best_cost = work;
        best = WPlan{Th, Tw, RS, DS, XS, QK, nth, ntw, 0, nob, ncb, lds};
      }
    }
  }
  if (best.Th) {
    const long ntiles = (long)B * T * best.nth * best.ntw;
    long cap = (long)HPVG_NUM_CU / ((long)KT * nob * ncb);  // one persistent workgroup per CU
    if (cap < 1) cap = 1;
    best.S = (int)(ntiles < cap ? ntiles : cap);
    best.S0 = best.S;
    static const bool balance = [] { const char* e = getenv("HPVG_WGRAD_BALANCE"); return !e || atoi(e) != 0; }();
    if (balance && KT == 3 && T >= 2 && best.S >= 2) {
      // split the 3*S workgroups of a channel-block pair so that tiles-with-work per workgroup are equal:
      // centre tap: ntiles / S1, outer taps: ntiles * (T-1)/T / S0
      const long Stot = 3L * best.S;
      long S1 = (Stot * T + (3L * T - 2) / 2) / (3L * T - 2);
      if (S1 > ntiles) S1 = ntiles;
      long S0 = (Stot - S1) / 2;
      if (S0 < 1) S0 = 1;
      if (S0 > S1) S0 = S1;
      best.S = (int)S1;
      best.S0 = (int)S0;
    }
  }
  return best;
}

WPlan plan_wgrad(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  struct Key { int B, Cin, Cout, T, H, W, KT; };
  struct Entry { Key k; WPlan p; bool used; };
  constexpr int NSLOT = 2048;
  static thread_local Entry cache[NSLOT];
  static thread_local int filled = 0;
  unsigned h = 2166136261u;
  for (int v : {B, Cin, Cout, T, H, W, KT}) h = (h ^ (unsigned)v) * 16777619u;
  for (int probe = 0; probe < NSLOT; ++probe) {
    Entry& e = cache[(h + probe) & (NSLOT - 1)];
    if (!e.used) {
      const WPlan p = plan_wgrad_search(B, Cin, Cout, T, H, W, KT);
      if (filled < NSLOT / 2) {
        e.k = Key{B, Cin, Cout, T, H, W, KT}; e.p = p; e.used = true;
        ++filled;
      }
      return p;
    }
    const Key& c = e.k;
    if (c.B == B && c.Cin == Cin && c.Cout == Cout && c.T == T && c.H == H && c.W == W && c.KT == KT) return e.p;
  }
  return plan_wgrad_search(B, Cin, Cout, T, H, W, KT);
}

// Tile plan of conv_wgradw_kernel (Winograd along W): the tile family of conv_wgrad_kernel with an EVEN band width and row
// stride, channel strides of 2 (mod 4) and the Winograd K loop's cost (12 MFMAs per 4 positions instead of 18).
// persistent slots of a conv_wgradw_kernel plan
void wgradw_slots(WPlan& best, int B, int T, int KT) {
  if (!best.Th) return;
  const long ntiles = (long)B * T * best.nth * best.ntw;
  long cap = (long)HPVG_NUM_CU / ((long)KT * best.nob * best.ncb);  // one persistent workgroup per CU
  if (cap < 1) cap = 1;
  best.S = (int)(ntiles < cap ? ntiles : cap);
  best.S0 = best.S;
  if (KT == 3 && T >= 2 && best.S >= 2) {   // tiles-with-work per workgroup equal over the three time taps (plan_wgrad_search)
    const long Stot = 3L * best.S;
    long S1 = (Stot * T + (3L * T - 2) / 2) / (3L * T - 2);
    if (S1 > ntiles) S1 = ntiles;
    long S0 = (Stot - S1) / 2;
    if (S0 < 1) S0 = 1;
    if (S0 > S1) S0 = S1;
    best.S = (int)S1;
    best.S0 = (int)S0;
  }
}
WPlan plan_wgradw_search(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  WPlan best{};
  double best_cost = 1e300;
  const int nob = hpvg_cdiv(Cout, 64), ncb = hpvg_cdiv(Cin, 64);
  int prev_tw = 0;
  for (int ntw = W; ntw >= 1; --ntw) {
    int Tw = hpvg_cdiv(W, ntw);
    Tw += Tw & 1;
    if (Tw == prev_tw || (long)(ntw - 1) * Tw >= W) continue;
    prev_tw = Tw;
    const int RS = Tw + 2;
    for (int Th = 1; Th <= H; ++Th) {
      const int nth = hpvg_cdiv(H, Th);
      if (Th != hpvg_cdiv(H, nth)) continue;
      const int QK = (Th * RS + 3) & ~3;                 // K positions, padded to the 4-position (2-pair) loop step
      const int DS = QK + 2;                             // dY row stride: >= QK, 2 (mod 4)
      int XS = QK + 2 * RS + 2;                          // X row: reads reach QK + 2*RS + 1
      if (XS < (Th + 2) * RS) XS = (Th + 2) * RS;
      while ((XS & 3) != 2) ++XS;
      if (DS > 512 || XS > 512) break;
      const size_t lds = (size_t)2 * 64 * (DS + XS) * sizeof(float);  // two tile buffers
      if (lds > 156 * 1024) break;
      const long ntiles = (long)B * T * nth * ntw;
      const double work = (double)ntiles * (QK * 0.25 * 12.0 + 40.0);
      if (work < best_cost) {
        best_cost = work;
        best = WPlan{Th, Tw, RS, DS, XS, QK, nth, ntw, 0, nob, ncb, lds};
      }
    }
  }
  wgradw_slots(best, B, T, KT);
  return best;
}
// the 16-byte staging form (conv_wgradw_kernel<.., G16>): W a multiple of 4, bands of a multiple of 4 columns, dY rows of Tw
// floats, X rows of Tw + 8 from column w0 - 4; at most two 64-lane pieces per channel row and operand
int g_wgradw_g16 = -1;
int g_wgradw_w8 = -1;   // eight-wave form of the 16-byte kernel (HPVG_WGRADW_W8, hpvg_conv_bwd_weight_wino_config mode 4 = off)
int g_wgradw_gen = 0;   // bumped when hpvg_conv_bwd_weight_wino_config changes what the planner may pick: drops the plan caches
WPlan plan_wgradw16_search(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  WPlan best{};
  if (g_wgradw_g16 < 0) {
    const char* e = getenv("HPVG_WGRADW_G16");
    g_wgradw_g16 = e ? atoi(e) : 1;
    const char* e8 = getenv("HPVG_WGRADW_W8");
    g_wgradw_w8 = e8 ? atoi(e8) : 1;
  }
  if (W % 4 != 0 || !g_wgradw_g16) return best;
  double best_cost = 1e300;
  const int nob = hpvg_cdiv(Cout, 64), ncb = hpvg_cdiv(Cin, 64);
  int prev_tw = 0;
  for (int ntw = W / 4; ntw >= 1; --ntw) {
    const int Tw = 4 * hpvg_cdiv(hpvg_cdiv(W, ntw), 4);
    if (Tw == prev_tw || (long)(ntw - 1) * Tw >= W) continue;
    prev_tw = Tw;
    const int RSx = Tw + 8;
    for (int Th = 1; Th <= H; ++Th) {
      const int nth = hpvg_cdiv(H, Th);
      if (Th != hpvg_cdiv(H, nth)) continue;
      const int QK = Th * Tw;
      const int DS = QK + 2, XS = (Th + 2) * RSx + 2;    // both 2 (mod 4); the X rows start at float 1
      if (Th * (Tw / 4) > 128 || (Th + 2) * (Tw / 4 + 2) > 128) break;
      const size_t lds = (size_t)2 * 64 * (DS + XS) * sizeof(float);
      if (lds > 156 * 1024) break;
      const long ntiles = (long)B * T * nth * ntw;
      double work = (double)ntiles * (QK * 0.25 * 12.0 + 40.0);
      // development: HPVG_WG16_FORCE="Th,Tw" restricts the search to that tile
      static const int force = [] { const char* e = getenv("HPVG_WG16_FORCE"); int a = 0, b = 0; return e && sscanf(e, "%d,%d", &a, &b) == 2 ? a * 1000 + b : 0; }();
      if (force && force != Th * 1000 + Tw) continue;
      if (work < best_cost) {
        best_cost = work;
        best = WPlan{Th, Tw, RSx, DS, XS, QK, nth, ntw, 0, nob, ncb, lds, 0, 1};
      }
    }
  }
  wgradw_slots(best, B, T, KT);
  return best;
}
WPlan plan_wgradw(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  struct Key { int B, Cin, Cout, T, H, W, KT; };
  struct Entry { Key k; WPlan p; };
  constexpr int NE = 256;
  static thread_local Entry cache[NE];
  static thread_local int filled = 0;
  static thread_local int gen = 0;
  if (gen != g_wgradw_gen) {
    gen = g_wgradw_gen;
    filled = 0;
  }
  for (int i = 0; i < filled; ++i) {
    const Key& c = cache[i].k;
    if (c.B == B && c.Cin == Cin && c.Cout == Cout && c.T == T && c.H == H && c.W == W && c.KT == KT) return cache[i].p;
  }
  WPlan p = plan_wgradw16_search(B, Cin, Cout, T, H, W, KT);
  if (!p.Th) p = plan_wgradw_search(B, Cin, Cout, T, H, W, KT);
  if (filled < NE) cache[filled++] = Entry{Key{B, Cin, Cout, T, H, W, KT}, p};
  return p;
}
inline size_t wgradw_slab_bytes(const WPlan& p, int KT) { return (size_t)p.S * KT * p.nob * p.ncb * 12 * 4096 * sizeof(float); }
// (+ the per-slot bias partials [S][nob][64] behind the slabs)
inline size_t wgradw_ws_bytes(const WPlan& p, int KT) { return 256 + wgradw_slab_bytes(p, KT) + (size_t)p.S * p.nob * 64 * sizeof(float); }
// HPVG_WGRAD_WINO (read once; hpvg_conv_bwd_weight_wino_config changes it at run time): 0 = never the Winograd weight
// gradient (the direct kernels: conv_wgrad_kernel / conv_wgrad3_kernel), anything else = EVERY wide layer (Cin > 4 and Cout > 4).
// There is no size rule at this level: the one-axis kernel with the slot-grouped reduce wins at every pyramid stage, 2-D and
// 3-D (profiles/r02_perf_wgrad_wino.txt); what IS chosen by size is which Winograd kernel runs (wgradw2_wanted below).
int g_wgradw_mode = -1;
inline bool wgradw_wanted(const WPlan& p, int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (g_wgradw_mode < 0) {
    const char* e = getenv("HPVG_WGRAD_WINO");
    g_wgradw_mode = e ? atoi(e) : 1;
  }
  return g_wgradw_mode != 0 && p.Th != 0 && Cin > 4 && Cout > 4;
}

// Tile plan of conv_wgradw2_kernel (Winograd over H and W): the 16-byte staging form with an EVEN tile height (rows of quads)
// and its K loop's cost (16 MFMAs per two quads = 8 positions).  Any W (W % 4 != 0: the kernel patches the one group per row
// that straddles the right border).
WPlan plan_wgradw2_search(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  WPlan best{};
  double best_cost = 1e300;
  const int nob = hpvg_cdiv(Cout, 64), ncb = hpvg_cdiv(Cin, 64);
  int prev_tw = 0;
  // development: HPVG_WG2_FORCE="Th,Tw" restricts the search to that tile
  static const int force = [] { const char* e = getenv("HPVG_WG2_FORCE"); int a = 0, b = 0; return e && sscanf(e, "%d,%d", &a, &b) == 2 ? a * 1000 + b : 0; }();
  for (int ntw = hpvg_cdiv(W, 4); ntw >= 1; --ntw) {
    const int Tw = 4 * hpvg_cdiv(hpvg_cdiv(W, ntw), 4);
    if (Tw == prev_tw || (long)(ntw - 1) * Tw >= W) continue;
    prev_tw = Tw;
    const int RSx = Tw + 8;
    // bands of 4 / 8 columns stage twice their width in halo and lose to wider ones although they pad least (measured at
    // 5 x 57 x 102: the 10 x 8 tile the area rule picked 0.128 ms, 6 x 16 0.115, 4 x 24 0.105): only where nothing wider fits
    if (Tw < 12 && W >= 24 && !force) continue;
    for (int Th = 2; Th <= H + 3; Th += 2) {
      const int nth = hpvg_cdiv(H, Th);
      const int thb = 2 * hpvg_cdiv(hpvg_cdiv(H, nth), 2);       // balanced, even
      if (Th != thb && !(Tw % 8 != 0 && Th == 4 * hpvg_cdiv(thb, 4))) continue;   // (or the next multiple of 4: see QK below)
      const int QK = Th * Tw;
      const int DS = QK + 2, XS = (Th + 2) * RSx + 2;    // both 2 (mod 4); the X rows start at float 1
      if (Th * (Tw / 4) > 128 || (Th + 2) * (Tw / 4 + 2) > 128) break;
      const size_t lds = (size_t)2 * 64 * (DS + XS) * sizeof(float);
      if (lds > 156 * 1024) break;
      if (QK % 16 != 0) continue;                        // an even number of K steps (the kernel's loop body holds two)
      const long ntiles = (long)B * T * nth * ntw;
      const double work = (double)ntiles * (QK * 0.125 * 16.0 + 40.0);
      if (force && force != Th * 1000 + Tw) continue;
      if (work < best_cost) {
        best_cost = work;
        best = WPlan{Th, Tw, RSx, DS, XS, QK, nth, ntw, 0, nob, ncb, lds, 0, 2};
      }
    }
  }
  wgradw_slots(best, B, T, KT);
  return best;
}
WPlan plan_wgradw2(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  struct Key { int B, Cin, Cout, T, H, W, KT; };
  struct Entry { Key k; WPlan p; };
  constexpr int NE = 256;
  static thread_local Entry cache[NE];
  static thread_local int filled = 0;
  for (int i = 0; i < filled; ++i) {
    const Key& c = cache[i].k;
    if (c.B == B && c.Cin == Cin && c.Cout == Cout && c.T == T && c.H == H && c.W == W && c.KT == KT) return cache[i].p;
  }
  const WPlan p = plan_wgradw2_search(B, Cin, Cout, T, H, W, KT);
  if (filled < NE) cache[filled++] = Entry{Key{B, Cin, Cout, T, H, W, KT}, p};
  return p;
}
inline size_t wgradw2_slab_bytes(const WPlan& p, int KT) { return (size_t)p.S * KT * p.nob * p.ncb * 16 * 4096 * sizeof(float); }
inline size_t wgradw2_ws_bytes(const WPlan& p, int KT) { return 256 + wgradw2_slab_bytes(p, KT) + (size_t)p.S * p.nob * 64 * sizeof(float); }
// Where the two-axis kernel is taken.  g_wgradw2: 0 = by size, 1 = never, 2 = wherever it can run (HPVG_WGRADW2 at start;
// hpvg_conv_bwd_weight_wino_config modes 5 / 6 set 2 / 1).  By size: a workgroup must walk enough tiles to amortise its larger
// slab (16 instead of 12 point tiles per block) and reduce.  Measured on MI355X against the one-axis kernel (tools/ab_wgrad2_sizes.sh,
// profiles/r03_ab_wgrad2_sizes.txt; 64 -> 64, tiles per persistent workgroup of the centre tap): 3x3x3: 1.4 tiles x0.99, 2.8
// x1.06, 4.1 x1.14, 13 x1.31, 53 x1.38, 117 x1.42; 3x3 (three times the workgroups per launch, a third of the work each):
// 2 x0.89, 4 x1.02, 5.3 x1.01, 6 x1.08, 8 x1.17.  HPVG_WGRADW2_MIN_TILES overrides both thresholds.
int g_wgradw2 = -1;
long g_wgradw2_min_tiles = -1;    // tiles per persistent workgroup from which the two-axis kernel is taken (default: 2 / 5 for 3-D / 2-D)
inline bool wgradw2_wanted(const WPlan& p2, int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (g_wgradw2 < 0) {
    const char* e = getenv("HPVG_WGRADW2");
    g_wgradw2 = e ? atoi(e) : 0;
    const char* m = getenv("HPVG_WGRADW2_MIN_TILES");
    if (m) g_wgradw2_min_tiles = atol(m);
  }
  if (g_wgradw2 == 1 || p2.Th == 0 || Cin <= 4 || Cout <= 4) return false;
  if (!wgradw_wanted(p2, B, Cin, Cout, T, H, W, KT)) return false;     // (the Winograd weight gradient switched off altogether)
  if (g_wgradw2 == 2) return true;
  const long ntiles = (long)B * T * p2.nth * p2.ntw;
  const long mt = g_wgradw2_min_tiles >= 0 ? g_wgradw2_min_tiles : (KT == 3 ? 2 : 5);
  return ntiles >= mt * (long)p2.S;
}

// tile plan of conv_wgrad3_kernel: the same tile family as conv_wgrad_kernel under its own LDS budget (two dY tiles, four
// 32-channel X planes, one row of zeros); S persistent workgroups per (64 output, 32 input channel) block pair
struct W3Plan { int Th, Tw, RS, DS, XS, QK, nth, ntw, S, nob, ncb; size_t lds; long ntiles; bool ok; };
W3Plan plan_wgrad3(int B, int Cin, int Cout, int T, int H, int W) {
  struct Key { int B, Cin, Cout, T, H, W; };
  struct Entry { Key k; W3Plan p; };
  constexpr int NE = 256;
  static thread_local Entry cache[NE];
  static thread_local int filled = 0;
  for (int i = 0; i < filled; ++i) {
    const Key& c = cache[i].k;
    if (c.B == B && c.Cin == Cin && c.Cout == Cout && c.T == T && c.H == H && c.W == W) return cache[i].p;
  }
  W3Plan best{};
  double best_cost = 1e300;
  for (int Tw = 1; Tw <= W; ++Tw) {
    const int ntw = hpvg_cdiv(W, Tw);
    if (Tw != hpvg_cdiv(W, ntw)) continue;
    const int RS = Tw + 2;
    for (int Th = 1; Th <= H; ++Th) {
      const int nth = hpvg_cdiv(H, Th);
      if (Th != hpvg_cdiv(H, nth)) continue;
      const int QK = (Th * RS + 3) & ~3;                 // K positions, padded to the 4-position loop step
      const int DS = (QK + 1) | 1;                       // dY row stride: >= QK, odd (bank spread)
      int XS = QK + 2 * RS + 4;                          // X row: reads reach QK-1 + 2*RS + 2
      if (XS < (Th + 2) * RS + 1) XS = (Th + 2) * RS + 1;
      XS |= 1;
      if (DS > 256 || XS > 256) break;                   // one 256-lane DMA round per row
      const size_t lds = ((size_t)2 * 64 * DS + (size_t)4 * 32 * XS + XS) * sizeof(float);
      if (lds > 158 * 1024) break;
      const long nsp = (long)B * nth * ntw;
      const double work = (double)nsp * T * (QK * 0.5 * 27.0 + 80.0);   // MFMAs of the K loop + per-tile fixed cost
      if (work < best_cost) {
        best_cost = work;
        best = W3Plan{Th, Tw, RS, DS, XS, QK, nth, ntw, 0, hpvg_cdiv(Cout, 64), hpvg_cdiv(Cin, 32), lds, nsp * T, true};
      }
    }
  }
  if (best.ok) {
    long cap = (long)HPVG_NUM_CU / ((long)best.nob * best.ncb);  // one persistent workgroup per CU
    if (cap < 1) cap = 1;
    best.S = (int)(best.ntiles < cap ? best.ntiles : cap);
  }
  if (filled < NE) cache[filled++] = Entry{Key{B, Cin, Cout, T, H, W}, best};
  return best;
}
inline size_t wgrad3_ws_bytes(const W3Plan& q) { return 256 + (size_t)q.S * q.nob * q.ncb * 28 * 2048 * sizeof(float); }
// conv_wgrad3_kernel pays off where a workgroup walks many tiles (its tiles carry three times the work, its slab is 1.5x
// the size): measured on MI355X (tools/perf_conv.py, 64 -> 64) it wins 3-4 % at 78 and 156 tiles per workgroup (stage 9,
// B = 2 / 4) and loses 2-5 % at 50 and below (stages <= 8).  HPVG_WGRAD3: 0 = never, 2 = always (tests), default = by size.
static const int g_wgrad3_mode = [] { const char* e = getenv("HPVG_WGRAD3"); return e ? atoi(e) : 1; }();
inline bool wgrad3_wanted(const W3Plan& q) {
  if (!q.ok || g_wgrad3_mode == 0) return false;
  return g_wgrad3_mode == 2 || q.ntiles >= 64L * q.S;
}

// narrow path selection: 0 = head (Cin <= 4), 1 = tail (Cout <= 4), -1 = full kernel
inline int narrow_mode(int Cin, int Cout) {
  if (Cin <= 4 && Cout > 4) return 0;
  if (Cout <= 4 && Cin > 4) return 1;
  return -1;
}
// tile plan of conv_wgrad_narrow2_kernel: Th image rows per tile such that the halo'd narrow tile (CN*KT planes) fits 60 KB
struct N2Plan { int Th, nth, RS, XPL, S; long ntiles; size_t lds; bool ok; };
inline N2Plan plan_narrow2w(int B, int CW, int CN, int T, int H, int W, int KT) {
  N2Plan q{};
  q.ok = false;
  if (W < 4) return q;
  const int RS = W + 2, NROW = CN * KT;
  int Th = (int)((60 * 1024 / sizeof(float)) / ((size_t)NROW * RS)) - 2;
  if (Th < 1) return q;
  if (Th > H) Th = H;
  // enough tiles to fill the chip twice over (small pyramid stages): ~1024 tiles per 64-channel block
  const long want_rows = ((long)H * B * T + 1023) / 1024;
  if (Th > want_rows) Th = (int)(want_rows < 1 ? 1 : want_rows);
  const int nth = hpvg_cdiv(H, Th);
  Th = hpvg_cdiv(H, nth);                      // balanced
  const int nwb = hpvg_cdiv(CW, 64);
  q.Th = Th; q.nth = nth; q.RS = RS; q.XPL = (Th + 2) * RS;
  q.ntiles = (long)B * T * nth;
  long S = 2L * HPVG_NUM_CU / nwb;             // two workgroups per CU
  if (S < 1) S = 1;
  if (q.ntiles < S) S = q.ntiles;
  q.S = (int)S;
  q.lds = (size_t)NROW * q.XPL * sizeof(float);
  const size_t xch = (size_t)2 * hpvg_cdiv(CN * KT * 9, 32) * 1024 * sizeof(float);   // the waves' end-of-kernel exchange buffer
  if (q.lds < xch) q.lds = xch;
  q.ok = true;
  return q;
}
static const bool g_narrow2w_off = [] { const char* e = getenv("HPVG_WGRAD_NARROW2"); return e && atoi(e) == 0; }();

inline size_t narrow_ws_bytes(const WPlan& p, int CW, int CN, int KT, long ntiles) {
  const int nwb = hpvg_cdiv(CW, 64);
  const int NT = hpvg_cdiv(CN * KT * 9, 32);
  long S = 2L * HPVG_NUM_CU / nwb;             // the larger of the two generations' grids
  if (S < 1) S = 1;
  return 256 + (size_t)S * 4 * nwb * 2 * NT * 1024 * sizeof(float);
}

}  // namespace

extern "C" {

size_t hpvg_conv_bwd_weight_ws_bytes(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  const WPlan p = plan_wgrad(B, Cin, Cout, T, H, W, KT);
  const int nm = narrow_mode(Cin, Cout);
  if (nm >= 0) return narrow_ws_bytes(p, nm == 0 ? Cout : Cin, nm == 0 ? Cin : Cout, KT, (long)B * T * p.nth * p.ntw);
  size_t need = 256 + (size_t)p.S * KT * p.nob * p.ncb * 9 * 4096 * sizeof(float);
  if (KT == 3) {
    const W3Plan q = plan_wgrad3(B, Cin, Cout, T, H, W);
    if (wgrad3_wanted(q) && wgrad3_ws_bytes(q) > need) need = wgrad3_ws_bytes(q);
  }
  const WPlan pw = plan_wgradw(B, Cin, Cout, T, H, W, KT);
  if (g_wgradw_mode != 0 && pw.Th != 0 && wgradw_ws_bytes(pw, KT) > need) need = wgradw_ws_bytes(pw, KT);  // (any run-time mode)
  const WPlan p2 = plan_wgradw2(B, Cin, Cout, T, H, W, KT);
  if (g_wgradw_mode != 0 && p2.Th != 0 && wgradw2_ws_bytes(p2, KT) > need) need = wgradw2_ws_bytes(p2, KT);
  return need;
}

// dw: natural layout [Cout][Cin][KT][3][3]; accumulate != 0 adds into dw instead of overwriting.
static int bwd_weight_impl(const float* dy, const float* x, const float* in_scale, const float* in_shift, int in_lrelu,
                           float* dw, int accumulate, float* db, int accumulate_db, void* ws, size_t ws_bytes, int B, int Cin,
                           int Cout, int T, int H, int W, int KT, void* stream);
int hpvg_conv_bwd_weight_f32(const float* dy, const float* x, const float* in_scale, const float* in_shift, int in_lrelu,
                             float* dw, int accumulate, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H,
                             int W, int KT, void* stream) {
  return bwd_weight_impl(dy, x, in_scale, in_shift, in_lrelu, dw, accumulate, nullptr, 0, ws, ws_bytes, B, Cin, Cout, T, H, W, KT,
                         stream);
}
// does hpvg_conv_bwd_weight_bias_f32 produce the bias gradient for this layer (the Winograd weight-gradient kernel runs it)?
int hpvg_conv_bwd_weight_fuses_bias(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1 || (KT != 1 && KT != 3) || narrow_mode(Cin, Cout) >= 0) return 0;
  if (wgradw2_wanted(plan_wgradw2(B, Cin, Cout, T, H, W, KT), B, Cin, Cout, T, H, W, KT)) return 1;
  return wgradw_wanted(plan_wgradw(B, Cin, Cout, T, H, W, KT), B, Cin, Cout, T, H, W, KT) ? 1 : 0;
}
// the weight gradient AND db[o] (+)= sum over batch and positions of dy - the conv's bias gradient - from the same launch:
// the centre-tap workgroups of conv_wgradw_kernel hold every dY pair in registers anyway.  HPVG_ERR_UNSUPPORTED where
// hpvg_conv_bwd_weight_fuses_bias() says 0 (the caller then uses hpvg_channel_sum_f32).
int hpvg_conv_bwd_weight_bias_f32(const float* dy, const float* x, float* dw, int accumulate, float* db, int accumulate_db, void* ws,
                                  size_t ws_bytes, int B, int Cin, int Cout, int T, int H, int W, int KT, void* stream) {
  if (!db) return HPVG_ERR_ARG;
  if (!hpvg_conv_bwd_weight_fuses_bias(B, Cin, Cout, T, H, W, KT)) return HPVG_ERR_UNSUPPORTED;
  return bwd_weight_impl(dy, x, nullptr, nullptr, 0, dw, accumulate, db, accumulate_db, ws, ws_bytes, B, Cin, Cout, T, H, W, KT, stream);
}
static int bwd_weight_impl(const float* dy, const float* x, const float* in_scale, const float* in_shift, int in_lrelu,
                           float* dw, int accumulate, float* db, int accumulate_db, void* ws, size_t ws_bytes, int B, int Cin,
                           int Cout, int T, int H, int W, int KT, void* stream) {
  if (!dy || !x || !dw || !ws) return HPVG_ERR_ARG;
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1) return HPVG_ERR_ARG;
  if (KT != 1 && KT != 3) return HPVG_ERR_UNSUPPORTED;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return HPVG_ERR_ARG;
  const WPlan p = plan_wgrad(B, Cin, Cout, T, H, W, KT);
  if (p.Th == 0) return HPVG_ERR_UNSUPPORTED;
  const int nm = narrow_mode(Cin, Cout);
  if (nm >= 0) {
    // ---- narrow layer (head / tail): dedicated kernel, the wide operand is streamed once
    if (in_scale) return HPVG_ERR_UNSUPPORTED;
    const int CW = nm == 0 ? Cout : Cin, CN = nm == 0 ? Cin : Cout;
    const long ntiles = (long)B * T * p.nth * p.ntw;
    if (ws_bytes < narrow_ws_bytes(p, CW, CN, KT, ntiles)) return HPVG_ERR_WORKSPACE;
    const int nwb = hpvg_cdiv(CW, 64);
    const int NT = hpvg_cdiv(CN * KT * 9, 32);
    hipStream_t s = (hipStream_t)stream;
    const N2Plan q = plan_narrow2w(B, CW, CN, T, H, W, KT);
    if (q.ok && !g_narrow2w_off) {
      // ---- second generation: the wide operand straight from global memory, 16 bytes per lane (conv_wgrad_narrow2_kernel)
      Narrow2WArgs n2;
      n2.wide = nm == 0 ? dy : x; n2.narrow = nm == 0 ? x : dy; n2.part = (float*)((char*)ws + 256);
      n2.B = B; n2.CW = CW; n2.CN = CN; n2.T = T; n2.H = H; n2.W = W;
      n2.Th = q.Th; n2.nth = q.nth; n2.RS = q.RS; n2.XPL = q.XPL; n2.S = q.S; n2.ntiles = (int)q.ntiles;
      const dim3 grid2((unsigned)q.S, nwb);
#define HPVG_NW2_LAUNCH(K, N)                                                                                         \
  {                                                                                                                   \
    static bool attr = false;                                                                                         \
    if (!attr) {                                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_narrow2_kernel<K, N>),                         \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                  \
        (void)hipGetLastError();                                                                                      \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    hipLaunchKernelGGL((conv_wgrad_narrow2_kernel<K, N>), grid2, dim3(256), q.lds, s, n2);                            \
  }
      if (KT == 3) {
        if (NT == 1) HPVG_NW2_LAUNCH(3, 1) else if (NT == 2) HPVG_NW2_LAUNCH(3, 2) else if (NT == 3) HPVG_NW2_LAUNCH(3, 3) else HPVG_NW2_LAUNCH(3, 4)
      } else {
        if (NT == 1) HPVG_NW2_LAUNCH(1, 1) else HPVG_NW2_LAUNCH(1, 2)
      }
#undef HPVG_NW2_LAUNCH
      int st2 = hpvg_launch_status();
      if (st2 != HPVG_OK) return st2;
      const int total2 = CW * CN * KT * 9;
      hipLaunchKernelGGL(conv_wgrad_narrow_reduce_kernel, dim3(hpvg_cdiv(total2, 4)), dim3(256), 0, s, (const float*)n2.part, dw,
                         q.S, nwb, NT, CW, CN, KT * 9, nm, accumulate);
      return hpvg_launch_status();
    }
    long S = HPVG_NUM_CU / nwb;
    if (S < 1) S = 1;
    if (ntiles < S) S = ntiles;
    NarrowArgs na;
    na.wide = nm == 0 ? dy : x; na.narrow = nm == 0 ? x : dy; na.part = (float*)((char*)ws + 256);
    na.B = B; na.CW = CW; na.CN = CN; na.T = T; na.H = H; na.W = W;
    na.Th = p.Th; na.Tw = p.Tw; na.RS = p.RS; na.DS = p.DS; na.XPL = p.XS; na.QK = p.QK; na.nth = p.nth; na.ntw = p.ntw;
    na.S = (int)S; na.ntiles = (int)ntiles;
    const size_t lds = (size_t)2 * (64 * p.DS + CN * KT * p.XS) * sizeof(float);
    const dim3 grid((unsigned)S, nwb);
#define HPVG_NW_LAUNCH(K, N)                                                                                          \
  {                                                                                                                   \
    static bool attr = false;                                                                                         \
    if (!attr) {                                                                                                      \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_narrow_kernel<K, N>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                  \
        (void)hipGetLastError();                                                                                      \
      attr = true;                                                                                                    \
    }                                                                                                                 \
    hipLaunchKernelGGL((conv_wgrad_narrow_kernel<K, N>), grid, dim3(256), lds, s, na);                                \
  }
    if (KT == 3) {
      if (NT == 1) HPVG_NW_LAUNCH(3, 1) else if (NT == 2) HPVG_NW_LAUNCH(3, 2) else if (NT == 3) HPVG_NW_LAUNCH(3, 3) else HPVG_NW_LAUNCH(3, 4)
    } else {
      if (NT == 1) HPVG_NW_LAUNCH(1, 1) else HPVG_NW_LAUNCH(1, 2)
    }
#undef HPVG_NW_LAUNCH
    int st = hpvg_launch_status();
    if (st != HPVG_OK) return st;
    const int total = CW * CN * KT * 9;
    hipLaunchKernelGGL(conv_wgrad_narrow_reduce_kernel, dim3(hpvg_cdiv(total, 4)), dim3(256), 0, s, (const float*)na.part, dw,
                       (int)S * 4, nwb, NT, CW, CN, KT * 9, nm, accumulate);
    return hpvg_launch_status();
  }
  if (in_scale) return HPVG_ERR_UNSUPPORTED;  // the fused-producer prologue needs the register-staged variant
  {
    const WPlan p2 = plan_wgradw2(B, Cin, Cout, T, H, W, KT);
    if (wgradw2_wanted(p2, B, Cin, Cout, T, H, W, KT)) {
      // ---- Winograd over H and W (conv_wgradw2_kernel): 16 point accumulators per time tap, transformed by the reduce kernel
      if (ws_bytes < wgradw2_ws_bytes(p2, KT)) return HPVG_ERR_WORKSPACE;
      WgradArgs a;
      a.dy = dy; a.x = x; a.in_scale = nullptr; a.in_shift = nullptr;
      a.part = (float*)((char*)ws + 256);
      a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T; a.H = H; a.W = W;
      a.Th = p2.Th; a.Tw = p2.Tw; a.RS = p2.RS; a.DS = p2.DS; a.XS = p2.XS; a.QK = p2.QK; a.nth = p2.nth; a.ntw = p2.ntw;
      a.S = p2.S; a.S0 = p2.S0; a.ncb = p2.ncb; a.nob = p2.nob; a.in_lrelu = 0;
      a.bpart = db ? (float*)((char*)ws + 256 + wgradw2_slab_bytes(p2, KT)) : nullptr;
      static const int order_env = [] { const char* e = getenv("HPVG_WG2_ORDER"); return e ? atoi(e) : 1; }();   // plane-major: same time, 1.9 instead of 3.4 GB of HBM traffic per stage-9 launch (profiles/r03_ab_wgrad2_order.txt)
      a.order = order_env;
      hipStream_t s = (hipStream_t)stream;
      const dim3 grid((KT == 3 ? 2 * p2.S0 + p2.S : p2.S) * p2.nob * p2.ncb);
      const int gjd = hpvg_cdiv(p2.Th * (p2.Tw / 4), 64), gjx = hpvg_cdiv((p2.Th + 2) * (p2.Tw / 4 + 2), 64);
#define HPVG_W2_LAUNCH(K, D, X, TWC, ST)                                                                               \
  {                                                                                                                    \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgradw2_kernel<K, D, X, TWC, ST>),                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                   \
        (void)hipGetLastError();                                                                                       \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_wgradw2_kernel<K, D, X, TWC, ST>), grid, dim3(256), p2.lds, s, a);                        \
  }
      // (the 16-column band - the tile of the large launches - has its row strides as immediates: one address register per
      // operand; W % 4 != 0 runs the instance that patches the straddling groups)
#define HPVG_W2_PICK(K)                                                                                                \
  if (W & 3) {                                                                                                         \
    if (gjd == 1 && gjx == 1 && p2.Tw == 16) HPVG_W2_LAUNCH(K, 1, 1, 16, true)                                         \
    else if (gjd == 1 && gjx == 1) HPVG_W2_LAUNCH(K, 1, 1, 0, true)                                                    \
    else if (gjd == 1) HPVG_W2_LAUNCH(K, 1, 2, 0, true)                                                                \
    else HPVG_W2_LAUNCH(K, 2, 2, 0, true)                                                                              \
  } else {                                                                                                             \
    if (gjd == 1 && gjx == 1 && p2.Tw == 16) HPVG_W2_LAUNCH(K, 1, 1, 16, false)                                        \
    else if (gjd == 1 && gjx == 1) HPVG_W2_LAUNCH(K, 1, 1, 0, false)                                                   \
    else if (gjd == 1) HPVG_W2_LAUNCH(K, 1, 2, 0, false)                                                               \
    else HPVG_W2_LAUNCH(K, 2, 2, 0, false)                                                                             \
  }
      if (KT == 3) {
        HPVG_W2_PICK(3)
      } else {
        HPVG_W2_PICK(1)
      }
#undef HPVG_W2_PICK
#undef HPVG_W2_LAUNCH
      int st2 = hpvg_launch_status();
      if (st2 != HPVG_OK) return st2;
      const int nbw2 = KT * p2.nob * p2.ncb * 64;
      hipLaunchKernelGGL(conv_wgradw2_reduce_kernel, dim3(nbw2 + (db ? hpvg_cdiv(Cout, 64) : 0)), dim3(64, 16), 0, s,
                         (const float*)a.part, dw, p2.S, p2.S0, KT, p2.nob, p2.ncb, Cout, Cin, accumulate, nbw2,
                         (const float*)a.bpart, db, accumulate_db);
      return hpvg_launch_status();
    }
  }
  {
    const WPlan pw = plan_wgradw(B, Cin, Cout, T, H, W, KT);
    if (wgradw_wanted(pw, B, Cin, Cout, T, H, W, KT)) {
      // ---- Winograd along W (conv_wgradw_kernel): 12 tap-point accumulators per time tap, transformed by the reduce kernel
      if (ws_bytes < wgradw_ws_bytes(pw, KT)) return HPVG_ERR_WORKSPACE;
      WgradArgs a;
      a.dy = dy; a.x = x; a.in_scale = nullptr; a.in_shift = nullptr;
      a.part = (float*)((char*)ws + 256);
      a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T; a.H = H; a.W = W;
      a.Th = pw.Th; a.Tw = pw.Tw; a.RS = pw.RS; a.DS = pw.DS; a.XS = pw.XS; a.QK = pw.QK; a.nth = pw.nth; a.ntw = pw.ntw;
      a.S = pw.S; a.S0 = pw.S0; a.ncb = pw.ncb; a.nob = pw.nob; a.in_lrelu = 0;
      a.bpart = db ? (float*)((char*)ws + 256 + wgradw_slab_bytes(pw, KT)) : nullptr;
      hipStream_t s = (hipStream_t)stream;
      const dim3 grid((KT == 3 ? 2 * pw.S0 + pw.S : pw.S) * pw.nob * pw.ncb);
      const int njd = pw.DS > 256 ? 2 : 1, njx = pw.XS > 256 ? 2 : 1;
      // whole channel rows per wave (WCH) where an instance exists for the row lengths: pieces of 64 lanes
      const int wjd = hpvg_cdiv(pw.DS, 64), wjx = hpvg_cdiv(pw.XS, 64);
      // 16-byte form: pieces of 64 groups
      const int gjd = hpvg_cdiv(pw.Th * (pw.Tw / 4), 64), gjx = hpvg_cdiv((pw.Th + 2) * (pw.Tw / 4 + 2), 64);
      static const int wch_mode = [] { const char* e = getenv("HPVG_WGRADW_WCH"); return e ? atoi(e) : 1; }();
#define HPVG_WW_LAUNCH(K, D, X, C, G, E)                                                                               \
  {                                                                                                                    \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgradw_kernel<K, D, X, C, G, E>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                   \
        (void)hipGetLastError();                                                                                       \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_wgradw_kernel<K, D, X, C, G, E>), grid, dim3(E ? 512 : 256), pw.lds, s, a);               \
  }
      // eight waves (two per SIMD, half of the Winograd points each) for the whole-row staging forms: HPVG_WGRADW_W8=0 keeps four
      const int w8_mode = g_wgradw_w8;
#define HPVG_WW_WCH(K)                                                                                                 \
  if (pw.g16 && gjd == 1 && gjx == 1 && w8_mode) HPVG_WW_LAUNCH(K, 1, 1, true, true, true)                             \
  else if (pw.g16 && gjd == 1 && w8_mode) HPVG_WW_LAUNCH(K, 1, 2, true, true, true)                                    \
  else if (pw.g16 && w8_mode) HPVG_WW_LAUNCH(K, 2, 2, true, true, true)                                                \
  else if (pw.g16 && gjd == 1 && gjx == 1) HPVG_WW_LAUNCH(K, 1, 1, true, true, false)                                  \
  else if (pw.g16 && gjd == 1) HPVG_WW_LAUNCH(K, 1, 2, true, true, false)                                              \
  else if (pw.g16) HPVG_WW_LAUNCH(K, 2, 2, true, true, false)                                                          \
  else if (wch_mode && wjd == 1 && wjx == 2) HPVG_WW_LAUNCH(K, 1, 2, true, false, false)                               \
  else if (wch_mode && wjd == 1 && wjx == 3) HPVG_WW_LAUNCH(K, 1, 3, true, false, false)                               \
  else if (wch_mode && wjd == 2 && wjx == 3) HPVG_WW_LAUNCH(K, 2, 3, true, false, false)                               \
  else if (wch_mode && wjd == 2 && wjx == 4) HPVG_WW_LAUNCH(K, 2, 4, true, false, false)                               \
  else if (njd == 1 && njx == 1) HPVG_WW_LAUNCH(K, 1, 1, false, false, false)                                          \
  else if (njd == 1) HPVG_WW_LAUNCH(K, 1, 2, false, false, false)                                                      \
  else HPVG_WW_LAUNCH(K, 2, 2, false, false, false)
      if (KT == 3) {
        HPVG_WW_WCH(3)
      } else {
        HPVG_WW_WCH(1)
      }
#undef HPVG_WW_WCH
#undef HPVG_WW_LAUNCH
      int stw = hpvg_launch_status();
      if (stw != HPVG_OK) return stw;
      const long totw = (long)KT * pw.nob * pw.ncb * 3 * 4096;
      const int nbw = hpvg_cdiv(totw, 128);
      hipLaunchKernelGGL(conv_wgradw_reduce_kernel, dim3(nbw + (db ? hpvg_cdiv(Cout, 128) : 0)), dim3(128, 8), 0, s, (const float*)a.part,
                         dw, pw.S, pw.S0, KT, pw.nob, pw.ncb, Cout, Cin, accumulate, nbw, (const float*)a.bpart, db, accumulate_db);
      return hpvg_launch_status();
    }
  }
  if (db) return HPVG_ERR_UNSUPPORTED;   // (only the Winograd kernel above produces the bias gradient)
  if (KT == 3) {
    const W3Plan q = plan_wgrad3(B, Cin, Cout, T, H, W);
    if (wgrad3_wanted(q)) {
      // ---- all 27 taps in one workgroup (conv_wgrad3_kernel)
      if (ws_bytes < wgrad3_ws_bytes(q)) return HPVG_ERR_WORKSPACE;
      Wgrad3Args w3;
      w3.dy = dy; w3.x = x; w3.part = (float*)((char*)ws + 256);
      w3.B = B; w3.Cin = Cin; w3.Cout = Cout; w3.T = T; w3.H = H; w3.W = W;
      w3.Th = q.Th; w3.Tw = q.Tw; w3.RS = q.RS; w3.DS = q.DS; w3.XS = q.XS; w3.QK = q.QK; w3.nth = q.nth; w3.ntw = q.ntw;
      w3.S = q.S; w3.ncb = q.ncb; w3.nob = q.nob; w3.ntiles = (int)q.ntiles;
      hipStream_t s3 = (hipStream_t)stream;
      const dim3 grid3((unsigned)(q.S * q.nob * q.ncb));
      {
        static bool attr = false;
        if (!attr) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024) != hipSuccess)
            (void)hipGetLastError();
          attr = true;
        }
        hipLaunchKernelGGL(conv_wgrad3_kernel, grid3, dim3(256), q.lds, s3, w3);
      }
      int st3 = hpvg_launch_status();
      if (st3 != HPVG_OK) return st3;
      const long total3 = (long)q.nob * q.ncb * 27 * 2048;
      hipLaunchKernelGGL(conv_wgrad3_reduce_kernel, dim3(hpvg_cdiv(total3, 256)), dim3(256, 4), 0, s3, (const float*)w3.part, dw, q.S,
                         q.nob, q.ncb, Cout, Cin, accumulate);
      return hpvg_launch_status();
    }
  }
  const size_t need = 256 + (size_t)p.S * KT * p.nob * p.ncb * 9 * 4096 * sizeof(float);
  if (ws_bytes < need) return HPVG_ERR_WORKSPACE;
  WgradArgs a;
  a.dy = dy; a.x = x; a.in_scale = in_scale; a.in_shift = in_shift;
  a.part = (float*)((char*)ws + 256);
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T; a.H = H; a.W = W;
  a.Th = p.Th; a.Tw = p.Tw; a.RS = p.RS; a.DS = p.DS; a.XS = p.XS; a.QK = p.QK; a.nth = p.nth; a.ntw = p.ntw;
  a.S = p.S; a.S0 = p.S0; a.ncb = p.ncb; a.nob = p.nob; a.in_lrelu = in_lrelu; a.bpart = nullptr;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((KT == 3 ? 2 * p.S0 + p.S : p.S) * p.nob * p.ncb);
  const int njd = p.DS > 256 ? 2 : 1, njx = p.XS > 256 ? 2 : 1;
#define HPVG_WG_LAUNCH(K, D, X)                                                                                        \
  {                                                                                                                    \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<K, D, X>),                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                   \
        (void)hipGetLastError();                                                                                       \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((conv_wgrad_kernel<K, D, X>), grid, dim3(256), p.lds, s, a);                                    \
  }
  if (KT == 3) {
    if (njd == 1 && njx == 1) HPVG_WG_LAUNCH(3, 1, 1)
    else if (njd == 1) HPVG_WG_LAUNCH(3, 1, 2)
    else HPVG_WG_LAUNCH(3, 2, 2)
  } else {
    if (njd == 1 && njx == 1) HPVG_WG_LAUNCH(1, 1, 1)
    else if (njd == 1) HPVG_WG_LAUNCH(1, 1, 2)
    else HPVG_WG_LAUNCH(1, 2, 2)
  }
#undef HPVG_WG_LAUNCH
  int st = hpvg_launch_status();
  if (st != HPVG_OK) return st;
  const long per_s = (long)KT * p.nob * p.ncb * 9 * 4096;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(hpvg_cdiv(per_s, 256)), dim3(256), 0, s, (const float*)a.part, dw, p.S, p.S0, KT,
                     p.nob, p.ncb, Cout, Cin, accumulate);
  return hpvg_launch_status();
}

// out[c] (+)= sum over batch and all spatial positions of x[b][c][...]  (conv bias gradient); ws: C*64 doubles
size_t hpvg_channel_sum_ws_bytes(int C) { return (size_t)C * 64 * sizeof(double); }
int hpvg_channel_sum_f32(const float* x, float* out, int accumulate, void* ws, size_t ws_bytes, int B, int C, long S,
                         void* stream) {
  if (!x || !out || !ws || B < 1 || C < 1 || S < 1) return HPVG_ERR_ARG;
  if (ws_bytes < hpvg_channel_sum_ws_bytes(C)) return HPVG_ERR_WORKSPACE;
  long want = (1024 + C - 1) / C;
  const long maxs = (S * B + 2047) / 2048;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  int ns = (int)want;
  // one pass (ns = 1, the partial kernel writes out[] itself) only when a single split is all the tensor is worth anyway:
  // forcing it on stage-4/5 tensors (C workgroups summing ~50 K elements each) lost 5 % of the iteration (A/B measured)
  const bool direct = ns == 1;
  float* dout = direct ? out : nullptr;
  hipStream_t s = (hipStream_t)stream;
  switch (hpvg_vec_width(x, S)) {
    case 4: hipLaunchKernelGGL(channel_sum_partial_kernel<4>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws, dout, accumulate); break;
    case 2: hipLaunchKernelGGL(channel_sum_partial_kernel<2>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws, dout, accumulate); break;
    default: hipLaunchKernelGGL(channel_sum_partial_kernel<1>, dim3(ns, C), dim3(256), 0, s, x, B, C, S, ns, (double*)ws, dout, accumulate);
  }
  if (direct) return hpvg_launch_status();
  hipLaunchKernelGGL(channel_sum_finish_kernel, dim3(hpvg_cdiv(C, 64)), dim3(64), 0, s, (const double*)ws, ns, C, out,
                     accumulate);
  return hpvg_launch_status();
}

// Run-time switch of the Winograd weight gradient (tests and A/B tools): 0 = never (the direct kernels), 1 = the default (every
// wide layer on a Winograd kernel; the two-axis one where wgradw2_wanted's size rule picks it), 2 = every wide layer, 3 = every
// wide layer with the 4-byte staging form only (2 and below: the 16-byte form where the width allows it, unless
// HPVG_WGRADW_G16=0), 4 = every wide layer, the 16-byte form on four waves instead of eight; 5 = every wide layer, the TWO-axis
// kernel (conv_wgradw2_kernel) on every wide layer; 6 = the one-axis kernel only; a negative mode only queries.
// Returns the mode in force.
int hpvg_conv_bwd_weight_wino_config(int mode) {
  (void)wgradw_wanted(WPlan{}, 1, 8, 8, 1, 1, 1, 1);   // settle the defaults
  (void)wgradw2_wanted(WPlan{}, 1, 8, 8, 1, 1, 1, 1);
  (void)plan_wgradw16_search(1, 8, 8, 1, 1, 1, 1);
  static const int env_g16 = g_wgradw_g16, env_w8 = g_wgradw_w8, env_w2 = g_wgradw2;
  static int four_byte_only = 0, four_waves = 0;
  if (mode >= 0) {
    four_byte_only = mode == 3;
    four_waves = mode == 4;
    g_wgradw_w8 = four_waves ? 0 : env_w8;
    g_wgradw_mode = mode > 2 ? 2 : mode;
    // 5: the two-axis kernel wherever it can run; 2 / 3 / 4 / 6: the one-axis kernel only (2: its default staging forms);
    // 0 / 1: as the process started (HPVG_WGRADW2)
    g_wgradw2 = mode == 5 ? 2 : (mode == 2 || mode == 3 || mode == 4 || mode == 6 ? 1 : env_w2);
    const int g16 = mode == 3 ? 0 : env_g16;
    if (g16 != g_wgradw_g16) {
      g_wgradw_g16 = g16;
      ++g_wgradw_gen;
    }
  }
  if (g_wgradw_mode == 2 && g_wgradw2 == 2) return 5;
  if (g_wgradw_mode == 2 && four_byte_only) return 3;
  if (g_wgradw_mode == 2 && four_waves) return 4;
  return g_wgradw_mode;
}

// host only: which kernel family hpvg_conv_bwd_weight_f32 runs this shape on: 0 = conv_wgrad_kernel (direct, a workgroup per
// time tap), 1 = conv_wgrad3_kernel (direct, all taps per workgroup), 2 = conv_wgradw_kernel (Winograd along W: 2/3 of the direct
// matrix-core work), 3 = conv_wgradw2_kernel (Winograd over H and W: 4/9), 4 = the narrow kernels (heads / tails)
int hpvg_conv_bwd_weight_kernel_kind(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1 || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  if (narrow_mode(Cin, Cout) >= 0) return 4;
  if (wgradw2_wanted(plan_wgradw2(B, Cin, Cout, T, H, W, KT), B, Cin, Cout, T, H, W, KT)) return 3;
  if (wgradw_wanted(plan_wgradw(B, Cin, Cout, T, H, W, KT), B, Cin, Cout, T, H, W, KT)) return 2;
  if (KT == 3 && wgrad3_wanted(plan_wgrad3(B, Cin, Cout, T, H, W))) return 1;
  return 0;
}

// host only: the tile plan of the Winograd weight-gradient kernel: out[0..9] as hpvg_conv_bwd_weight_plan
int hpvg_conv_bwd_weight_wino_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3) || B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1) return HPVG_ERR_ARG;
  const WPlan p = plan_wgradw(B, Cin, Cout, T, H, W, KT);
  if (p.Th == 0) return HPVG_ERR_UNSUPPORTED;
  out[0] = p.Th; out[1] = p.Tw; out[2] = p.nth; out[3] = p.ntw; out[4] = p.QK; out[5] = p.S; out[6] = p.DS; out[7] = p.XS;
  out[8] = (int)p.lds; out[9] = B * T * p.nth * p.ntw;
  return HPVG_OK;
}

// host only: the tile plan of the two-axis Winograd weight-gradient kernel (conv_wgradw2_kernel): out[0..9] as
// hpvg_conv_bwd_weight_plan, out[10] = 1 when this shape runs it by default (the size rule); HPVG_ERR_UNSUPPORTED when no tile fits
int hpvg_conv_bwd_weight_wino2_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3) || B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1) return HPVG_ERR_ARG;
  const WPlan p = plan_wgradw2(B, Cin, Cout, T, H, W, KT);
  if (p.Th == 0) return HPVG_ERR_UNSUPPORTED;
  out[0] = p.Th; out[1] = p.Tw; out[2] = p.nth; out[3] = p.ntw; out[4] = p.QK; out[5] = p.S; out[6] = p.DS; out[7] = p.XS;
  out[8] = (int)p.lds; out[9] = B * T * p.nth * p.ntw;
  out[10] = wgradw2_wanted(p, B, Cin, Cout, T, H, W, KT) ? 1 : 0;
  return HPVG_OK;
}

int hpvg_conv_bwd_weight_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  const WPlan p = plan_wgrad(B, Cin, Cout, T, H, W, KT);
  out[0] = p.Th; out[1] = p.Tw; out[2] = p.nth; out[3] = p.ntw; out[4] = p.QK; out[5] = p.S; out[6] = p.DS; out[7] = p.XS;
  out[8] = (int)p.lds; out[9] = B * T * p.nth * p.ntw;
  return HPVG_OK;
}

}  // extern "C"
