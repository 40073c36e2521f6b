// 3x3 / 3x3x3 "same" convolution as an implicit GEMM on the gfx950 fp32 matrix cores.
//
// Replaces (reference, /root/reference): nn.Conv3d / nn.Conv2d forward as used by
//   ConvBlock3D       modules/networks_3d.py:48-56     ConvBlock2D   modules/networks_2d.py:53-61
//   ConvBlock3DSN     modules/networks_3d.py:59-70     ConvBlock2DSN modules/networks_2d.py:64-75
//   tails             modules/networks_3d.py:175,341,362
// and, through a flipped/transposed weight pack, their backward-data pass (stride 1, pad 1:
// bwd-data is a forward conv with W'[c][o][tap] = W[o][c][ntaps-1-tap]).
//
// GEMM view (per batch sample b and output time-plane t):
//     Y[o][q] = sum_{tap} sum_{c} W[o][c][tap] * X[c][q + off(tap)]
//   M = output channels (32 per MFMA tile, MB tiles per wave)
//   N = output positions q of one (Th x Tw) spatial tile, flattened with the LDS row stride
//       RS = Tw + 2 so that every tap is a constant offset (the two halo columns per row yield
//       junk columns of the GEMM that are never stored)
//   K = (tap, input-channel) - two input channels per v_mfma_f32_32x32x2_f32.
//
// Data movement: the input tile (CC channels x KT time planes x (Th+2) x (Tw+2), zero padded) is
// staged through LDS one channel chunk at a time (register staged, so the producer's
// BatchNorm-apply + LeakyReLU can be fused into the load); the B operand is read with
// conflict-free ds_read_b32 (32 consecutive dwords per half wave).  The A operand (weights) is
// pre-packed in fragment order and read straight from L2 with one 16-byte load per lane per
// (tap, m-tile); all workgroups read the same 442 KB so it stays L2 resident.
// Accumulation is exact fp32 (v_mfma_f32_32x32x2_f32 == chain of fmaf).
#include "hpvg_common.h"
#include <stdlib.h>

namespace {

// source of zero padding for the LDS-DMA staging (out-of-image lanes read this word)
__device__ const float g_zero_word[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct ConvFwdArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  float* y;
  int B, Cin, Cout, T, H, W;
  int Th, Tw, RS, PL, nth, ntw, nblocks, nchunk, ntiles, mbtot, nj, S;
  int in_lrelu, out_lrelu;
  // split-K (latency-bound grids): blockIdx.z owns channel chunks [z*cps, (z+1)*cps) and writes raw partial sums
  float* part;       // nullptr = no split
  long part_stride;  // floats per split slab (= B*Cout*T*H*W)
  int cps;
};

template <int CP> struct AVecT;
template <> struct AVecT<4> { typedef f32x4 type; };
template <> struct AVecT<2> { typedef f32x2 type; };

constexpr int NJMAX = 4;  // (Th+2)*RS <= 1024

template <int CC, int KT, int MB, int NB>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvFwdArgs a) {
  constexpr int CP = CC / 2;
  constexpr int TAPS = KT * 9;
  typedef typename AVecT<CP>::type AVec;
  extern __shared__ __attribute__((aligned(16))) float xs[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int tile = hpvg_xcd_remap(blockIdx.x, a.ntiles);
  const int tw_i = tile % a.ntw;
  int r = tile / a.ntw;
  const int th_i = r % a.nth;
  r /= a.nth;
  const int t = r % a.T;
  const int b = r / a.T;
  const int h0 = th_i * a.Th, w0 = tw_i * a.Tw;
  const int mb0 = blockIdx.y * MB;
  const long HW = (long)a.H * a.W;
  const int RS = a.RS, PL = a.PL;

  // ---- per-thread staging slots: position p = j*256+tid inside one (Th+2) x RS input plane
  int gofs[NJMAX];
  unsigned okmask = 0, wmask = 0;
  const int plload = (a.Th + 2) * RS;
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) {
    const int p = j * 256 + tid;
    gofs[j] = 0;
    if (j < a.nj && p < plload) {
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh - 1, gw = w0 + ww - 1;
      wmask |= 1u << j;
      if (gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) {
        okmask |= 1u << j;
        gofs[j] = gh * a.W + gw;
      }
    }
  }

  f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][i][e] = 0.f;

  const bool prologue = a.in_scale != nullptr;
  const float* xl = xs + half * KT * PL + l31 + wave * 32;

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int ch_lo = a.part ? (int)blockIdx.z * a.cps : 0;
  const int ch_hi = a.part ? (ch_lo + a.cps < a.nchunk ? ch_lo + a.cps : a.nchunk) : a.nchunk;
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    if (ch > ch_lo) __syncthreads();
    if (!prologue) {
      // ---------------- stage CC x KT planes by LDS-DMA: no VGPR round trip, every load of the chunk in flight at
      // once; out-of-image lanes read a global zero word, lanes past the plane end are masked off
#pragma unroll 1
      for (int pl = 0; pl < CC * KT; ++pl) {
        const int c = pl / KT, dt = pl - c * KT;
        const int cg = ch * CC + c;
        const int tt = t + dt - (KT == 3 ? 1 : 0);
        const bool valid = cg < a.Cin && tt >= 0 && tt < a.T;
        const float* src = a.x + (((long)b * a.Cin + (valid ? cg : 0)) * a.T + (valid ? tt : 0)) * HW;
        float* dst = xs + pl * PL + wave * 64;
#pragma unroll
        for (int j = 0; j < NJMAX; ++j)
          if ((wmask >> j) & 1u)
            __builtin_amdgcn_global_load_lds((gptr_t)((valid && ((okmask >> j) & 1u)) ? src + gofs[j] : g_zero_word),
                                             (lptr_t)(dst + j * 256), 4, 0, 0);
      }
    } else {
    // ---------------- stage CC x KT planes into LDS through registers (fused affine + LeakyReLU of the producer)
#pragma unroll 2
    for (int c = 0; c < CC; ++c) {
      const int cg = ch * CC + c;
      const bool cok = cg < a.Cin;
      float sc = 1.f, sh = 0.f;
      if (cok) {
        sc = a.in_scale[cg];
        sh = a.in_shift[cg];
      }
      float v[KT][NJMAX];
#pragma unroll
      for (int dt = 0; dt < KT; ++dt) {
        const int tt = t + dt - (KT == 3 ? 1 : 0);
        const bool tok = cok && tt >= 0 && tt < a.T;
        const float* src = a.x + (((long)b * a.Cin + (cok ? cg : 0)) * a.T + (tok ? tt : 0)) * HW;
#pragma unroll
        for (int j = 0; j < NJMAX; ++j) {
          const bool ld = tok && ((okmask >> j) & 1u);
          v[dt][j] = ld ? src[gofs[j]] : 0.f;
        }
      }
#pragma unroll
      for (int dt = 0; dt < KT; ++dt) {
        const int tt = t + dt - (KT == 3 ? 1 : 0);
        const bool tok = cok && tt >= 0 && tt < a.T;
#pragma unroll
        for (int j = 0; j < NJMAX; ++j) {
          if ((wmask >> j) & 1u) {
            float val = v[dt][j];
            if (tok && ((okmask >> j) & 1u)) {
              val = val * sc + sh;
              if (a.in_lrelu) val = hpvg_lrelu(val);
            }
            xs[(c * KT + dt) * PL + j * 256 + tid] = val;
          }
        }
      }
    }
    }
    __syncthreads();

    // ---------------- MFMA over (tap, channel pair)
    const AVec* wpt = reinterpret_cast<const AVec*>(a.wp) + ((long)(ch * TAPS) * a.mbtot + mb0) * 64 + lane;
    AVec av[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) av[m] = wpt[m * 64];
    // B fragments of k-step s+1 are read from LDS before the MFMAs of k-step s (two register sets; CP is even, so the
    // parity is static); past the chunk's last k-step this is a harmless in-buffer read.
    float bv[2][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bv[0][i] = xl[i * 128];
#pragma unroll 1
    for (int dt = 0; dt < KT; ++dt) {
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          // prefetch the next tap's A fragments (the pack has tail padding)
          wpt += (long)a.mbtot * 64;
          AVec an[MB];
#pragma unroll
          for (int m = 0; m < MB; ++m) an[m] = wpt[m * 64];
          const float* xt = xl + dt * PL + dh * RS + dw;
          const float* xn = (dw < 2) ? xt + 1 : (dh < 2 ? xl + dt * PL + (dh + 1) * RS : xl + (dt + 1 < KT ? dt + 1 : 0) * PL);
#pragma unroll
          for (int cp = 0; cp < CP; ++cp) {
            const float* nx = (cp + 1 < CP) ? xt + (2 * (cp + 1) * KT) * PL : xn;
#pragma unroll
            for (int i = 0; i < NB; ++i) bv[(cp + 1) & 1][i] = nx[i * 128];
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ABOVE this k-step's MFMAs
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
              for (int m = 0; m < MB; ++m)
                acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][cp], bv[cp & 1][i], acc[m][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
          }
#pragma unroll
          for (int m = 0; m < MB; ++m) av[m] = an[m];
        }
      }
    }
  }

  // ---------------- epilogue: bias, optional LeakyReLU, masked store
  // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float bias_r[MB][16];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = (mb0 + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      bias_r[m][e] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int blk = wave + 4 * i;
    const int q = blk * 32 + l31;
    const int hh = q / RS, ww = q - hh * RS;
    const int gh = h0 + hh, gw = w0 + ww;
    const bool ok = blk < a.nblocks && ww < a.Tw && hh < a.Th && gh < a.H && gw < a.W;
    if (!ok) continue;
    const long sp = (long)t * HW + (long)gh * a.W + gw;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = (mb0 + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (co < a.Cout) {
          const long oi = ((long)b * a.Cout + co) * a.T * HW + sp;
          if (a.part) {
            a.part[(long)blockIdx.z * a.part_stride + oi] = acc[m][i][e];  // raw partial; bias/activation in the reduce
          } else {
            float val = acc[m][i][e] + bias_r[m][e];
            if (a.out_lrelu) val = hpvg_lrelu(val);
            a.y[oi] = val;
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Pipelined variant (no fused producer on the input): one persistent workgroup per CU (1 wave per SIMD), two LDS
// buffers.  While the MFMA loop of work item i = (tile, channel chunk) runs out of buffer i&1, item i+1 is staged
// into the other buffer by LDS-DMA (global_load_lds_dword, no VGPR round trip), one input plane per tap slotted
// between the MFMAs so that staging rides in the matrix pipe's shadow.  One barrier per item.
template <int CC, int KT, int MB, int NB>
__global__ __launch_bounds__(256, 1) void conv_mfma_pipe_kernel(const ConvFwdArgs a) {
  constexpr int CP = CC / 2;
  constexpr int TAPS = KT * 9;
  constexpr int NPL = CC * KT;  // input planes per chunk
  typedef typename AVecT<CP>::type AVec;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) float xs[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int mb0 = blockIdx.y * MB;
  const long HW = (long)a.H * a.W;
  const int RS = a.RS, PL = a.PL;
  const int BUF = NPL * PL;
  const int plload = (a.Th + 2) * RS;
  const int pt = (KT == 3 ? 1 : 0);

  // staging state of the tile being loaded
  int goff[NJMAX];
  bool gok[NJMAX], gln[NJMAX];
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) gln[j] = j < a.nj && j * 256 + tid < plload;
  const float* sbase = a.x;  // x + b*Cin*T*HW of the staged tile
  int st_t = 0;
  auto setup = [&](int tile, int& b, int& t, int& h0, int& w0) {
    const int tw_i = tile % a.ntw;
    int r = tile / a.ntw;
    const int th_i = r % a.nth;
    r /= a.nth;
    t = r % a.T;
    b = r / a.T;
    h0 = th_i * a.Th;
    w0 = tw_i * a.Tw;
#pragma unroll
    for (int j = 0; j < NJMAX; ++j) {
      const int p = j * 256 + tid;
      const int hh = p / RS, ww = p - hh * RS;
      const int gh = h0 + hh - 1, gw = w0 + ww - 1;
      gok[j] = gln[j] && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      goff[j] = gok[j] ? gh * a.W + gw : 0;
    }
    sbase = a.x + (long)b * a.Cin * a.T * HW;
    st_t = t;
  };
  // stage input plane pl (= c*KT + dt) of channel chunk ch of the staged tile into buf
  auto dma_plane = [&](int pl, int ch, float* buf) {
    const int c = pl / KT, dt = pl - c * KT;
    const int cg = ch * CC + c;
    const int tt = st_t + dt - pt;
    const bool valid = cg < a.Cin && tt >= 0 && tt < a.T;
    const float* src = sbase + ((long)(valid ? cg : 0) * a.T + (valid ? tt : 0)) * HW;
    float* dst = buf + pl * PL + wave * 64;
#pragma unroll
    for (int j = 0; j < NJMAX; ++j)
      if (gln[j])
        __builtin_amdgcn_global_load_lds((gptr_t)((valid && gok[j]) ? src + goff[j] : g_zero_word), (lptr_t)(dst + j * 256), 4, 0, 0);
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][i][e] = 0.f;

  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  int cb, ct, ch0, cw0;  // coordinates of the tile being computed (for its epilogue)
  setup(tile, cb, ct, ch0, cw0);
  for (int pl = 0; pl < NPL; ++pl) dma_plane(pl, 0, xs);
  __syncthreads();

  int cur = 0;
  for (; tile < a.ntiles; tile += a.S) {
    for (int ch = 0; ch < a.nchunk; ++ch) {
      // ---- what to stage while computing (tile, ch)
      const bool last_chunk = ch + 1 == a.nchunk;
      int nch = ch + 1;
      bool have_next = true;
      int nb_, nt_, nh0_, nw0_;
      if (last_chunk) {
        nch = 0;
        have_next = tile + a.S < a.ntiles;
        if (have_next) setup(tile + a.S, nb_, nt_, nh0_, nw0_);
      }
      float* bufc = xs + cur * BUF;
      float* bufn = xs + (cur ^ 1) * BUF;
      int plnext = have_next ? 0 : NPL;

      // ---- MFMA over (tap, channel pair) of the current chunk.  One wave per SIMD: nothing but this wave's own
      // instruction order hides latency, so the B fragments of k-step s+1 are read from LDS before the MFMAs of
      // k-step s (two register sets, static parity because CP is even) and the A fragments one tap ahead.
      const float* xl = bufc + half * KT * PL + l31 + wave * 32;
      const AVec* wpt = reinterpret_cast<const AVec*>(a.wp) + ((long)(ch * TAPS) * a.mbtot + mb0) * 64 + lane;
      AVec av[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) av[m] = wpt[m * 64];
      float bv[2][NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) bv[0][i] = xl[i * 128];
#pragma unroll 1
      for (int dt = 0; dt < KT; ++dt) {
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            wpt += (long)a.mbtot * 64;
            AVec an[MB];
            const float* xt = xl + dt * PL + dh * RS + dw;
            // first k-step of the NEXT tap (past the chunk's last tap this is a harmless in-buffer read)
            const float* xn = (dw < 2) ? xt + 1 : (dh < 2 ? xl + dt * PL + (dh + 1) * RS : xl + (dt + 1) * PL);
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
              const float* nx = (cp + 1 < CP) ? xt + (2 * (cp + 1) * KT) * PL : xn;
#pragma unroll
              for (int i = 0; i < NB; ++i) bv[(cp + 1) & 1][i] = nx[i * 128];
              __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ABOVE this k-step's MFMAs
#pragma unroll
              for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int m = 0; m < MB; ++m)
                  acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][cp], bv[cp & 1][i], acc[m][i], 0, 0, 0);
              if (cp == 0) {
                // next tap's A fragments and one plane of DMA staging, issued right AFTER this tap's first MFMA
                // group: with LDS-DMA in flight hipcc waits vmcnt(0) (not a counted wait) at the first use of an
                // ordinary load, so every VMEM op must be >= ~3/4 tap old when the next tap starts
#pragma unroll
                for (int m = 0; m < MB; ++m) an[m] = wpt[m * 64];
                if (plnext < NPL) { dma_plane(plnext, nch, bufn); ++plnext; }
              }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) av[m] = an[m];
          }
        }
      }
      while (plnext < NPL) { dma_plane(plnext, nch, bufn); ++plnext; }

      if (last_chunk) {
        // ---- epilogue of the finished tile: bias, optional LeakyReLU, masked store; then reset the accumulators
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int blk = wave + 4 * i;
          const int q = blk * 32 + l31;
          const int hh = q / RS, ww = q - hh * RS;
          const int gh = ch0 + hh, gw = cw0 + ww;
          const bool ok = blk < a.nblocks && ww < a.Tw && hh < a.Th && gh < a.H && gw < a.W;
          const long sp = (long)ct * HW + (long)gh * a.W + gw;
#pragma unroll
          for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int co = (mb0 + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
              if (ok && co < a.Cout) {
                float val = acc[m][i][e];
                if (a.bias) val += a.bias[co];
                if (a.out_lrelu) val = hpvg_lrelu(val);
                a.y[((long)cb * a.Cout + co) * a.T * HW + sp] = val;
              }
              acc[m][i][e] = 0.f;
            }
          }
        }
        cb = nb_; ct = nt_; ch0 = nh0_; cw0 = nw0_;
      }
      __syncthreads();  // staged buffer complete (the barrier's fence waits for pending LDS-DMA); current one free
      cur ^= 1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight pack: natural [Cout][Cin][taps] -> MFMA A-fragment order
//   wp[chunk][tap][mblock][lane][cp] = Wsrc[o = mblock*32 + (lane&31)][c = chunk*CC + 2cp + (lane>>5)][tap]
// transpose_flip=1 packs the backward-data weights (Wsrc[o'][c'][tap] = W[c'][o'][ntaps-1-tap]).
// A device scalar `inv_scale` (1/sigma of spectral norm) is folded in when given.
__global__ void conv_pack_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, float* __restrict__ wp,
                                 int Cin_k, int Cout_k, int taps, int CC, int nchunk, int mbtot, int transpose_flip,
                                 long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CP = CC / 2;
  long r = idx;
  const int cp = r % CP; r /= CP;
  const int lane = r % 64; r /= 64;
  const int mb = r % mbtot; r /= mbtot;
  const int tap = r % taps; r /= taps;
  const int ch = (int)r;
  float val = 0.f;
  if (ch < nchunk) {  // else: tail padding stays zero
    const int o = mb * 32 + (lane & 31);
    const int c = ch * CC + 2 * cp + (lane >> 5);
    if (o < Cout_k && c < Cin_k) {
      if (!transpose_flip) val = w[((long)o * Cin_k + c) * taps + tap];
      else val = w[((long)c * Cout_k + o) * taps + (taps - 1 - tap)];
      if (inv_scale) val *= inv_scale[0];
    }
  }
  wp[idx] = val;
}

// y = act(bias[c] + sum_z part[z]) : finishing pass of the split-K launch (fixed summation order: reproducible)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                  float* __restrict__ y, int nsplit, long slab, int C, long S,
                                                                  int lrelu) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < slab; i += (long)gridDim.x * 256) {
    float v = part[i];
    for (int z = 1; z < nsplit; ++z) v += part[(long)z * slab + i];
    if (bias) v += bias[(i / S) % C];
    if (lrelu) v = hpvg_lrelu(v);
    y[i] = v;
  }
}

inline int conv_cc(int Cin) { return Cin <= 4 ? 4 : 8; }

struct Plan {
  int Th, Tw, RS, PL, nth, ntw, nblocks, NB, MB, gridy, nj;
  size_t lds;      // bytes of ONE tile buffer (the pipelined kernel uses two)
  bool pipelined;
};

// Tile planner: minimise (waves of workgroups over the chip) x (per-workgroup MFMA rounds).
Plan plan_conv_search(int B, int Cin, int Cout, int T, int H, int W, int KT, bool pipelined) {
  const int CC = conv_cc(Cin);
  const int mbtot = hpvg_cdiv(Cout, 32);
  Plan best{};
  double best_cost = 1e300;
  // development knobs: restrict the search to one NB / MB
  static const int only_nb = [] { const char* e = getenv("HPVG_PLAN_NB"); return e ? atoi(e) : 0; }();
  static const int only_mb = [] { const char* e = getenv("HPVG_PLAN_MB"); return e ? atoi(e) : 0; }();
  for (int MB = (mbtot >= 2 ? 2 : 1); MB >= 1; --MB) {
    if (only_mb && MB != only_mb && mbtot >= 2) continue;
    const int gridy = hpvg_cdiv(mbtot, MB);
    for (int Tw = 1; Tw <= W; ++Tw) {
      const int ntw = hpvg_cdiv(W, Tw);
      if (Tw != hpvg_cdiv(W, ntw)) continue;  // only balanced splits of W
      const int RS = Tw + 2;
      for (int Th = 1; Th <= H; ++Th) {
        const int nth = hpvg_cdiv(H, Th);
        if (Th != hpvg_cdiv(H, nth)) continue;
        const int qmax = (Th - 1) * RS + Tw - 1;
        const int nblocks = qmax / 32 + 1;
        if (nblocks > 16) break;
        if ((Th + 2) * RS > NJMAX * 256) break;
        const int rounds = hpvg_cdiv(nblocks, 4);
        const int NB = rounds <= 1 ? 1 : (rounds == 2 ? 2 : 4);
        if (only_nb && NB != only_nb) continue;
        int PL = (Th + 2) * RS;
        const int need = 128 * NB + 2 * RS + 2;
        if (PL < need) PL = need;
        const size_t lds = (size_t)CC * KT * PL * sizeof(float);
        if (lds > (pipelined ? 78 : 160) * 1024) continue;
        const int per_cu = pipelined ? 1 : (lds <= 80 * 1024 ? 2 : 1);
        const long nwg = (long)B * T * nth * ntw * gridy;
        const long slots = (long)HPVG_NUM_CU * per_cu;
        double cost;
        if (pipelined) {
          // persistent workgroups walk ceil(ntiles/S) tiles each; staging is hidden behind the MFMA loop
          long S = slots / gridy;
          if (S < 1) S = 1;
          const long ntl = nwg / gridy;
          const double rounds = (double)((ntl + S - 1) / S);
          cost = rounds * ((double)NB * MB + 0.15) + 5e-4 * (double)nwg;
        } else {
          const double waves = nwg <= slots ? 1.0 : (double)nwg / (double)slots;
          // per-WG time ~ MFMA rounds (scaled by m-tiles) + fixed staging/sync overhead
          const double per = (double)NB * MB + 0.6 + 0.002 * (Th + 2) * RS;
          cost = waves * per + 5e-4 * (double)nwg;  // tie-break: fewer, fuller workgroups
        }
        if (cost < best_cost - 1e-9) {
          best_cost = cost;
          best = Plan{Th, Tw, RS, PL, nth, ntw, nblocks, NB, MB, gridy, hpvg_cdiv((Th + 2) * RS, 256), lds, pipelined};
        }
      }
    }
  }
  return best;
}

// The search costs ~10-20 us of host time; shapes repeat every iteration, so plans are memoised (host-side, tiny).
Plan plan_conv(int B, int Cin, int Cout, int T, int H, int W, int KT, bool pipelined) {
  struct Key { int B, Cin, Cout, T, H, W, KT, pipe; };
  struct Entry { Key k; Plan p; };
  static thread_local Entry cache[128];
  static thread_local int used = 0;
  const Key k{B, Cin, Cout, T, H, W, KT, pipelined ? 1 : 0};
  for (int i = 0; i < used; ++i) {
    const Key& c = cache[i].k;
    if (c.B == B && c.Cin == Cin && c.Cout == Cout && c.T == T && c.H == H && c.W == W && c.KT == KT && c.pipe == k.pipe)
      return cache[i].p;
  }
  const Plan p = plan_conv_search(B, Cin, Cout, T, H, W, KT, pipelined);
  if (used < 128) cache[used++] = Entry{k, p};
  return p;
}

// Split-K decision for latency-bound grids: when the classic grid has fewer workgroups than the chip has slots, the
// channel chunks of every tile are spread over blockIdx.z (each workgroup then runs a 1/nsplit-long serial chain) and
// a finishing kernel sums the slabs.  Returns the number of splits (1 = no split).
inline int conv_nsplit(const Plan& pc, int B, int T, int nchunk, int KT) {
  const long nwg = (long)B * T * pc.nth * pc.ntw * pc.gridy;
  // measured (bench.py per-stage it/s): pays for 3x3x3 convs on grids of ~100-500 workgroups; loses for 2-D convs
  // (9 taps per chunk: the partial-slab write + finishing kernel outweigh the shorter chain) and for tiny grids
  if (KT != 3 || nchunk < 2 || nwg >= 2L * HPVG_NUM_CU || nwg < 128) return 1;
  long want = (3L * HPVG_NUM_CU + nwg - 1) / nwg;  // aim at ~3 workgroups per CU
  if (want > 4) want = 4;
  if (want > nchunk) want = nchunk;
  if (want < 2) return 1;
  const int cps = hpvg_cdiv(nchunk, (int)want);
  return hpvg_cdiv(nchunk, cps);
}

template <int CC, int KT, int MB, int NB>
int launch_conv(const ConvFwdArgs& a, const Plan& p, hipStream_t s) {
  static bool attr_set = false, attr_set2 = false;
  if (p.pipelined) {
    auto kern = conv_mfma_pipe_kernel<CC, KT, MB, NB>;
    if (!attr_set2) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
          hipSuccess)
        (void)hipGetLastError();
      attr_set2 = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.S, p.gridy), dim3(256), 2 * p.lds, s, a);
    return hpvg_launch_status();
  }
  auto kern = conv_mfma_kernel<CC, KT, MB, NB>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntiles, p.gridy, a.part ? hpvg_cdiv(a.nchunk, a.cps) : 1), dim3(256), p.lds, s, a);
  return hpvg_launch_status();
}

template <int CC, int KT>
int dispatch_conv(const ConvFwdArgs& a, const Plan& p, hipStream_t s) {
  if (p.MB == 2) {
    switch (p.NB) {
      case 1: return launch_conv<CC, KT, 2, 1>(a, p, s);
      case 2: return launch_conv<CC, KT, 2, 2>(a, p, s);
      default: return launch_conv<CC, KT, 2, 4>(a, p, s);
    }
  }
  switch (p.NB) {
    case 1: return launch_conv<CC, KT, 1, 1>(a, p, s);
    case 2: return launch_conv<CC, KT, 1, 2>(a, p, s);
    default: return launch_conv<CC, KT, 1, 4>(a, p, s);
  }
}

}  // namespace

extern "C" {

// number of floats of the packed-weight buffer for a conv with Cin -> Cout (kernel view)
size_t hpvg_conv_wpack_floats(int Cin, int Cout, int KT) {
  const int CC = conv_cc(Cin);
  const int nchunk = hpvg_cdiv(Cin, CC);
  const int mbtot = hpvg_cdiv(Cout, 32);
  const size_t per_tap = (size_t)mbtot * 64 * (CC / 2);
  return ((size_t)nchunk * KT * 9 + 2) * per_tap;  // +2 taps of zero tail padding (A prefetch runs one tap / one m-tile ahead)
}

// w: natural layout of the LAYER weight [Cout_layer][Cin_layer][KT][3][3].
// transpose_flip = 0: pack for the forward conv (kernel Cin=Cin_layer, Cout=Cout_layer)
// transpose_flip = 1: pack for backward-data (kernel Cin=Cout_layer, Cout=Cin_layer)
int hpvg_conv_pack_weight_f32(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                              int transpose_flip, void* stream) {
  if (!w || !wp || (KT != 1 && KT != 3) || Cin_layer < 1 || Cout_layer < 1) return HPVG_ERR_ARG;
  const int Cin_k = transpose_flip ? Cout_layer : Cin_layer;
  const int Cout_k = transpose_flip ? Cin_layer : Cout_layer;
  const int CC = conv_cc(Cin_k);
  const int nchunk = hpvg_cdiv(Cin_k, CC);
  const int mbtot = hpvg_cdiv(Cout_k, 32);
  const long total = (long)hpvg_conv_wpack_floats(Cin_k, Cout_k, KT);
  hipLaunchKernelGGL(conv_pack_kernel, dim3(hpvg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, inv_scale, wp,
                     Cin_k, Cout_k, KT * 9, CC, nchunk, mbtot, transpose_flip, total);
  return hpvg_launch_status();
}

// y[b][o][t][h][w] = bias[o] + sum_{c,tap} Wp[o][c][tap] * f(x)[b][c][t+dt-pt][h+dh-1][w+dw-1]
// f = identity, or (in_scale[c]*x + in_shift[c]) followed by LeakyReLU(0.2) when in_lrelu (zero padding
// is applied AFTER f, as in the reference where f is the previous block's BatchNorm+LeakyReLU output).
int hpvg_conv_fwd_f32(const float* x, const float* wp, const float* bias, const float* in_scale, const float* in_shift,
                      int in_lrelu, float* y, int out_lrelu, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H,
                      int W, int KT, void* stream) {
  if (!x || !wp || !y) return HPVG_ERR_ARG;
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1) return HPVG_ERR_ARG;
  if (KT != 1 && KT != 3) return HPVG_ERR_UNSUPPORTED;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return HPVG_ERR_ARG;
  // development knob: HPVG_CONV_PIPE = 0 classic (no split) / 1 pipelined persistent / 3 classic + split-K; default auto
  static const int force = [] { const char* e = getenv("HPVG_CONV_PIPE"); return e ? atoi(e) : -1; }();
  // Variant choice (measured, tools/perf_conv.py).  Chip-filling grids: classic kernel, two co-resident workgroups per
  // CU hide each other's staging.  Latency-bound grids (< 512 workgroups): split the channel chunks over blockIdx.z
  // (split-K) when the caller provided workspace, else the software-pipelined persistent kernel.
  const Plan pc = plan_conv(B, Cin, Cout, T, H, W, KT, false);
  const int nchunk_c = hpvg_cdiv(Cin, conv_cc(Cin));
  int nsplit = conv_nsplit(pc, B, T, nchunk_c, KT);
  const size_t slab = (size_t)B * Cout * T * H * W;
  if (nsplit > 1 && (!ws || ws_bytes < (size_t)nsplit * slab * sizeof(float))) nsplit = 1;
  if (force == 0 || force == 1) nsplit = 1;
  bool pipelined = false;
  if (nsplit == 1 && in_scale == nullptr) pipelined = (long)B * T * pc.nth * pc.ntw * pc.gridy <= 2L * HPVG_NUM_CU;
  if (force == 0 || force == 3) pipelined = false;
  if (force == 1 && in_scale == nullptr) pipelined = true;
  const Plan p = plan_conv(B, Cin, Cout, T, H, W, KT, pipelined);
  if (p.Th == 0) return HPVG_ERR_UNSUPPORTED;
  ConvFwdArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.y = y;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T; a.H = H; a.W = W;
  a.Th = p.Th; a.Tw = p.Tw; a.RS = p.RS; a.PL = p.PL; a.nth = p.nth; a.ntw = p.ntw; a.nblocks = p.nblocks;
  const int CC = conv_cc(Cin);
  a.nchunk = hpvg_cdiv(Cin, CC);
  a.ntiles = B * T * p.nth * p.ntw;
  a.mbtot = hpvg_cdiv(Cout, 32);
  a.nj = p.nj;
  {
    long S = HPVG_NUM_CU / p.gridy;
    if (S < 1) S = 1;
    a.S = (int)(a.ntiles < S ? a.ntiles : S);
  }
  a.in_lrelu = in_lrelu; a.out_lrelu = out_lrelu;
  a.part = nsplit > 1 ? (float*)ws : nullptr;
  a.part_stride = (long)slab;
  a.cps = nsplit > 1 ? hpvg_cdiv(a.nchunk, nsplit) : a.nchunk;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (CC == 8) rc = KT == 3 ? dispatch_conv<8, 3>(a, p, s) : dispatch_conv<8, 1>(a, p, s);
  else rc = KT == 3 ? dispatch_conv<4, 3>(a, p, s) : dispatch_conv<4, 1>(a, p, s);
  if (rc != HPVG_OK || nsplit == 1) return rc;
  long nb = (long)((slab + 1023) / 1024);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float*)ws, bias, y,
                     hpvg_cdiv(a.nchunk, a.cps), (long)slab, Cout, (long)T * H * W, out_lrelu);
  return hpvg_launch_status();
}

// workspace the split-K path of hpvg_conv_fwd_f32 wants for this shape (0: the shape never splits)
size_t hpvg_conv_fwd_ws_bytes(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1 || (KT != 1 && KT != 3)) return 0;
  const Plan pc = plan_conv(B, Cin, Cout, T, H, W, KT, false);
  const int nsplit = conv_nsplit(pc, B, T, hpvg_cdiv(Cin, conv_cc(Cin)), KT);
  return nsplit > 1 ? (size_t)nsplit * B * Cout * T * H * W * sizeof(float) : 0;
}

// Debug/introspection: the tile plan the launcher will use (for tests and DESIGN.md tables).
// out[0..9] = Th, Tw, nth, ntw, nblocks, NB, MB, gridy, lds_bytes, ntiles
int hpvg_conv_fwd_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  const Plan p = plan_conv(B, Cin, Cout, T, H, W, KT, true);
  out[0] = p.Th; out[1] = p.Tw; out[2] = p.nth; out[3] = p.ntw; out[4] = p.nblocks; out[5] = p.NB; out[6] = p.MB;
  out[7] = p.gridy; out[8] = (int)p.lds; out[9] = B * T * p.nth * p.ntw;
  return HPVG_OK;
}

}  // extern "C"
