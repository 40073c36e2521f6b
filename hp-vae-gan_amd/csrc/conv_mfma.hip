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
// staged through LDS one channel chunk at a time (LDS-DMA; register staged when the producer's
// BatchNorm-apply + LeakyReLU is fused into the load); the B operand is read with
// conflict-free ds_read_b32 (32 consecutive dwords per half wave).  The A operand (weights) is
// pre-packed in fragment order and read straight from L2 with one 16-byte load per lane per
// (tap, m-tile); all workgroups read the same 442 KB so it stays L2 resident.
// Accumulation is exact fp32 (v_mfma_f32_32x32x2_f32 == chain of fmaf).
#include "hpvg_common.h"
#include "hpvg.h"
#include <stdlib.h>

namespace {

// source of zero padding for the LDS-DMA staging (out-of-image lanes read this word)
__device__ const float g_zero_word[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
// a whole H x W plane of zeros (conv_wino2r_kernel: a time plane outside the clip is staged like any other, from here)
constexpr long HPVG_ZERO_PLANE_FLOATS = 1L << 18;
__device__ float g_zero_plane[HPVG_ZERO_PLANE_FLOATS];

#ifdef HPVG_TRACE
// development build only (tools/trace_conv.py): per-workgroup start/end (100 MHz clock) and placement
__device__ unsigned long long g_trace[8192 * 4];
__device__ unsigned long long g_phase[8192 * 4];  // per workgroup (wave 0): cycles issuing DMA, waiting for it, in the MFMA loop, epilogue
#define HPVG_PH(i) { unsigned long long now_ = __builtin_readcyclecounter(); ph_a[i] += now_ - ph_t; ph_t = now_; }
#define HPVG_PH_INIT unsigned long long ph_t = __builtin_readcyclecounter(); unsigned long long ph_a[4] = {0, 0, 0, 0};
#define HPVG_PH_END if (threadIdx.x == 0) { for (int i_ = 0; i_ < 4; ++i_) g_phase[4 * blockIdx.x + i_] = ph_a[i_]; }
#define HPVG_TRACE_BEGIN unsigned long long tr_t0 = wall_clock64(); unsigned long long tr_c0 = __builtin_readcyclecounter();
#define HPVG_TRACE_END                                                              \
  if (threadIdx.x == 0 && blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) < 8192) { \
    unsigned long long* tr = g_trace + 4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)); \
    tr[0] = tr_t0;                                                                  \
    tr[1] = wall_clock64();                                                         \
    tr[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                              \
    tr[3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) | ((__builtin_readcyclecounter() - tr_c0) << 8); \
  }
#else
#define HPVG_TRACE_BEGIN
#define HPVG_TRACE_END
#define HPVG_PH(i)
#define HPVG_PH_INIT
#define HPVG_PH_END
#endif

struct ConvFwdArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  const float* mask;  // optional [B][Cout][T][H][W]: y *= (mask > 0 ? 1 : 0.2) - leaky_relu_backward fused into the backward-data conv
  // 1-bit form of such a mask, [B][T*H*W][ceil(Cout/32)] words (bit c%32 of word c/32 = activation > 0):
  const unsigned* mask_bits;  // read by the masked epilogue INSTEAD of `mask` when given (8 loads per tile and lane, not 128)
  unsigned* bits_out;         // written by a LeakyReLU epilogue for its consumer's backward-data conv
  float* y;
  int B, Cin, Cout, T, H, W;
  // tile = `L` consecutive positions of the row-flattened band (W split into ntw bands of Tw columns, LDS row stride
  // RS = Tw + 2), nrange tiles per band and plane; Th only for the 2-D tiles of the narrow kernel (L = Th * RS)
  int Th, Tw, RS, PL, L, qstride, nrange, ntw, nblocks, nchunk, mbtot, nj;  // tile r of a band starts at r * qstride
  int in_lrelu, out_lrelu;
  // schedule: tiles [0, ndp*S) are data-parallel rounds (round k: tile k*S + g), tiles [skbase, ntl) are cut into
  // (tile, channel-chunk) items dealt evenly over the S workgroups (stream-K)
  int gridy, ntl, ndp, skbase;
  float* skpart;  // [S][2][MB*NB*16][256] raw accumulator slabs of partially computed tiles
  int stagger;    // conv_wino_kernel: the workgroup in a CU's odd wave slot starts this many s_sleep(127) (~4 us each) late
};

template <int CP> struct AVecT;
template <> struct AVecT<4> { typedef f32x4 type; };
template <> struct AVecT<2> { typedef f32x2 type; };

constexpr int NJMAX = 4;  // (Th+2)*RS <= 1024

// ------------------------------------------------------------------------------------------
// Schedule (stream-K).  A workgroup's serial K loop over one tile (8 chunks x 864 MFMAs per wave) lasts ~200 us, so a
// grid of one workgroup per tile with 1-5 tiles per CU slot is badly quantised (2080 tiles on 512 slots: the fifth
// round is 6 % full; 560 tiles: the second round 9 %).  The grid is therefore exactly the number of co-resident
// workgroups S (2 per CU, 512).  Tiles [0, ndp*S) are handed out whole (round k: tile k*S + g); the remaining tiles
// are cut into (tile, channel chunk) items and the item list is split evenly over the S workgroups, so every
// workgroup runs the same number of chunk iterations +-1.  A tile whose chunks end up in more than one workgroup is
// written as raw accumulator slabs (register order, fully coalesced) and finished by conv_fixup_kernel in workgroup
// order (reproducible; no float atomics).  With S = number of tiles the same kernel is the plain one-workgroup-per-tile
// launch (no workspace needed).
// Measured on MI355X (tools/trace_conv.py, per-workgroup timeline): one workgroup per CU already saturates the matrix
// pipe (2 per CU: +3 %); the sustained shader clock under this load is 2.03-2.07 GHz with real data (2.29 GHz when the
// operands are zeros), i.e. the fp32 MFMA ceiling of the part is ~134 TFLOP/s, not the 157.3 of 2.4 GHz.
struct SkTile { int b, t, q0, w0, mb0; };
__device__ __forceinline__ SkTile sk_decode_tile(const ConvFwdArgs& a, int tile, int MB) {
  SkTile c;
  // tile order: output-channel group fastest, then TIME, then the spatial tile.  Consecutive ids (= one XCD, thanks to
  // hpvg_xcd_remap) are then the same spatial tile at neighbouring t: the input planes t-1, t, t+1 they share are fetched
  // from HBM once and hit in that XCD's L2 for the other two (spatial-major order re-read every plane 3x: PMC)
  const int yb = tile % a.gridy;
  int r = tile / a.gridy;
  c.t = r % a.T; r /= a.T;
  const int tw_i = r % a.ntw; r /= a.ntw;
  const int r_i = r % a.nrange;
  c.b = r / a.nrange;
  c.q0 = r_i * a.qstride;
  c.w0 = tw_i * a.Tw;
  c.mb0 = yb * MB;
  return c;
}

// Per-thread staging slots of a tile: position p = j*256+tid inside one (Th+2) x RS input plane (zero padded).
struct StageSlots {
  int gofs[NJMAX];        // element offset inside a (c,t) plane (valid lanes)
  unsigned bofs[NJMAX];   // the same in bytes (planes are < 4 GB)
  unsigned okmask, wmask; // bit j: slot j lies inside the image / inside the staged plane
  unsigned long long em[NJMAX];  // EXEC mask of the lanes whose slot j lies inside the image (wave-uniform)
};
__device__ __forceinline__ StageSlots conv_stage_slots(const ConvFwdArgs& a, int q0, int w0, int tid) {
  // slot p holds position P = q0 + p of the zero-padded, row-flattened band: row P / RS - 1, column w0 + P % RS - 1
  StageSlots sl;
  sl.okmask = 0;
  sl.wmask = 0;
  const int RS = a.RS, plload = a.L + 2 * RS + 2;
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) {
    const int p = j * 256 + tid;
    sl.gofs[j] = 0;
    sl.bofs[j] = 0;
    if (j < a.nj && p < plload) {
      const int P = q0 + p;
      const int hh = P / RS, ww = P - hh * RS;
      const int gh = hh - 1, gw = w0 + ww - 1;
      sl.wmask |= 1u << j;
      if (gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) {
        sl.okmask |= 1u << j;
        sl.gofs[j] = gh * a.W + gw;
        sl.bofs[j] = (unsigned)sl.gofs[j] * 4u;
      }
    }
    sl.em[j] = __ballot((sl.okmask >> j) & 1u);
  }
  return sl;
}

#ifndef HPVG_CONV_ASM_STAGE
#define HPVG_CONV_ASM_STAGE 1
#endif
// One LDS-DMA piece of the staging loop as straight-line code: scalar plane base + per-lane byte offset (no address
// VALU), the out-of-image lanes switched off through EXEC inside the statement (the builtin under `if (lane ok)` is a
// saveexec + taken branch + a 64-bit VALU add per piece).  Invisible to the compiler: the barrier that publishes the chunk is
// preceded by an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ void conv_dma_piece(const char* base_, unsigned voff, unsigned lds_addr, unsigned long long mask) {
  // the base IS wave-uniform, but under SGPR pressure hipcc may keep it in VGPRs and then hands the asm statement a VGPR
  // pair for its "s" operand (an assembler error at best): rebuild it from two readfirstlanes so that its definition is scalar
  const unsigned long long bb = (unsigned long long)base_;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb);            // (the builtin returns int:
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bb >> 32));    //  no sign extension, please)
  const unsigned long long base = (unsigned long long)lo | ((unsigned long long)hi << 32);
  unsigned long long keep_exec;
  unsigned keep_m0;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b32 %1, m0\n\t"
      "s_mov_b64 exec, %3\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %2, %5\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(keep_exec), "=&s"(keep_m0)
      : "v"(voff), "s"(mask), "s"(lds_addr), "s"(base)
      : "memory");
}

// The same for a 16-byte piece (global_load_lds_dwordx4: LDS destination = m0 + 16 * lane).  Neither the per-lane global
// address nor the LDS base needs more than 4-byte alignment (tools/glds16_probe.hip, run on MI355X: correct for every
// source and destination offset, no measurable cost).
__device__ __forceinline__ void conv_dma_piece16(const char* base_, unsigned voff, unsigned lds_addr, unsigned long long mask) {
  const unsigned long long bb = (unsigned long long)base_;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bb >> 32));
  const unsigned long long base = (unsigned long long)lo | ((unsigned long long)hi << 32);
  unsigned long long keep_exec;
  unsigned keep_m0;
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b32 %1, m0\n\t"
      "s_mov_b64 exec, %3\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(keep_exec), "=&s"(keep_m0)
      : "v"(voff), "s"(mask), "s"(lds_addr), "s"(base)
      : "memory");
}

// The 16-byte piece for call sites where EVERY lane of the wave is active (the K loops): EXEC goes back to all ones by
// constant and m0 is declared clobbered instead of saved and restored - measured with tools/mfma_fillers.hip, a vector-memory
// instruction between two fp32 MFMAs costs the matrix pipe ~40 cycles whatever its width or its active lanes, and the
// save / restore pairs another 16 (the restoring s_mov of m0 and of exec each delay the MFMA behind them); this form: 47.
__device__ __forceinline__ void conv_dma_piece16_all(const char* base_, unsigned voff, unsigned lds_addr, unsigned long long mask) {
  const unsigned long long bb = (unsigned long long)base_;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bb >> 32));
  const unsigned long long base = (unsigned long long)lo | ((unsigned long long)hi << 32);
  asm volatile(
      "s_mov_b64 exec, %1\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %3\n\t"
      "s_mov_b64 exec, -1"
      :
      : "v"(voff), "s"(mask), "s"(lds_addr), "s"(base)
      : "memory", "m0");
}

// Stage channel chunk `ch` (CC channels x KT time planes) of the tile of sample b / output plane t into xs.
template <int CC, int KT, bool PRO>
__device__ __forceinline__ void conv_stage_chunk(const ConvFwdArgs& a, float* xs, const StageSlots& sl, int ch, bool first_chunk,
                                                 int b, int t, int tid, int wave) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const long HW = (long)a.H * a.W;
  const int PL = a.PL;
  if constexpr (!PRO) {
    // LDS-DMA staging with as little VALU work as possible (a co-resident wave with MFMAs queued keeps the SIMD's
    // VALU port busy, which stretched a select-per-load version of this loop from 2.4 to 15-20 us per chunk): the
    // per-lane byte offsets are tile constants, the plane base is scalar, out-of-image lanes are switched off by
    // EXEC and their LDS words are zeroed once per tile (no load ever writes them).
    // plane pointers advance by scalar adds (recomputing ((b*Cin + c)*T + t)*HW per plane is ~15 scalar instructions of
    // 64-bit multiply in a loop whose pieces cost ~8)
    const long HWb = HW * 4;
    const char* cbase = reinterpret_cast<const char*>(a.x) + (((long)b * a.Cin + (long)ch * CC) * a.T + (t - (KT == 3 ? 1 : 0))) * HWb;
    int pl = 0;
#pragma unroll 1
    for (int c = 0; c < CC; ++c, cbase += (long)a.T * HWb)
#pragma unroll
    for (int dt = 0; dt < KT; ++dt, ++pl) {
      const int cg = ch * CC + c;
      const int tt = t + dt - (KT == 3 ? 1 : 0);
      const bool valid = cg < a.Cin && tt >= 0 && tt < a.T;
      float* dst = xs + pl * PL + wave * 64;
      if (valid) {
        const char* src = cbase + dt * HWb;
#if HPVG_CONV_ASM_STAGE
        const unsigned dst_lds = (unsigned)(size_t)(lptr_t)dst;
#pragma unroll
        for (int j = 0; j < NJMAX; ++j)
          if (sl.em[j] != 0ull) conv_dma_piece(src, sl.bofs[j], dst_lds + j * 1024, sl.em[j]);
#else
#pragma unroll
        for (int j = 0; j < NJMAX; ++j)
          if ((sl.okmask >> j) & 1u)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + sl.bofs[j]), (lptr_t)(dst + j * 256), 4, 0, 0);
#endif
        if (first_chunk) {
#pragma unroll
          for (int j = 0; j < NJMAX; ++j)
            if (((sl.wmask & ~sl.okmask) >> j) & 1u) xs[pl * PL + j * 256 + tid] = 0.f;
        }
      } else if (first_chunk || cg >= a.Cin) {
        // a plane outside the clip (t) stays zero for the whole tile; a channel past Cin only exists in the last chunk
#pragma unroll
        for (int j = 0; j < NJMAX; ++j)
          if ((sl.wmask >> j) & 1u) xs[pl * PL + j * 256 + tid] = 0.f;
      }
    }
  } else {
    // through registers: the producer's BatchNorm-apply (+ LeakyReLU) is fused into the load
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
          const bool ld = tok && ((sl.okmask >> j) & 1u);
          v[dt][j] = ld ? src[sl.gofs[j]] : 0.f;
        }
      }
#pragma unroll
      for (int dt = 0; dt < KT; ++dt) {
        const int tt = t + dt - (KT == 3 ? 1 : 0);
        const bool tok = cok && tt >= 0 && tt < a.T;
#pragma unroll
        for (int j = 0; j < NJMAX; ++j) {
          if ((sl.wmask >> j) & 1u) {
            float val = v[dt][j];
            if (tok && ((sl.okmask >> j) & 1u)) {
              val = val * sc + sh;
              if (a.in_lrelu) val = hpvg_lrelu(val);
            }
            xs[(c * KT + dt) * PL + j * 256 + tid] = val;
          }
        }
      }
    }
  }
}

// Co-resident workgroups per CU.  Default: two (256 VGPRs each).  Development build -DHPVG_CONV_WGS4: instances with at
// most 4 accumulator tiles per wave (64 accumulator registers) are compiled for FOUR per CU (<= 128 VGPRs, LDS <= 38 KB)
// - more workgroups staging while others own the matrix pipe (tools/ab_conv_wgs.py: the A/B).
#ifdef HPVG_CONV_WGS4
#define HPVG_CONV_WGS(MB, NB) ((MB) * (NB) <= 4 ? 4 : 2)
#else
#define HPVG_CONV_WGS(MB, NB) 2
#endif

// VAR selects what is compiled around the MFMA loop, so that the plain instance - every forward conv and most backward-data
// convs of the path - carries none of it (a run-time `if (a.mask)` in the epilogue alone cost the plain launches +1.6 % at
// stage 9 through register allocation): 0 = plain, 1 = out_mask epilogue (leaky_relu_backward of the layer below),
// 2 = the producer's BatchNorm-apply (+ LeakyReLU) fused into the staging (register path instead of LDS-DMA).
constexpr int VAR_PLAIN = 0, VAR_MASK = 1, VAR_PRO = 2, VAR_BITS = 3;   // 3 = plain + bits_out (the activated forward convs)

template <int CC, int KT, int MB, int NB, int VAR>
__global__ __launch_bounds__(256, HPVG_CONV_WGS(MB, NB)) void conv_mfma_kernel(const ConvFwdArgs a) {
  constexpr int CP = CC / 2;
  constexpr int TAPS = KT * 9;
  typedef typename AVecT<CP>::type AVec;
  extern __shared__ __attribute__((aligned(16))) float xs[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const long HW = (long)a.H * a.W;
  const int RS = a.RS, PL = a.PL;
  const float* xl = xs + half * KT * PL + l31 + wave * 32;

  HPVG_TRACE_BEGIN
  const int S = gridDim.x;
  const int g = hpvg_xcd_remap(blockIdx.x, S);
  const long Isk = (long)(a.ntl - a.skbase) * a.nchunk;
  int it = (int)((long)g * Isk / S);
  const int it_hi = (int)((long)(g + 1) * Isk / S);
  const int first_sk_tile = a.skbase + it / a.nchunk;

  HPVG_PH_INIT
  bool first_stage = true;
  for (int k = 0;; ++k) {
    int tile, ch_lo, ch_hi;
    if (k < a.ndp) {
      tile = k * S + g;
      ch_lo = 0;
      ch_hi = a.nchunk;
    } else {
      if (it >= it_hi) break;
      const int tr = it / a.nchunk;
      tile = a.skbase + tr;
      ch_lo = it - tr * a.nchunk;
      const int n = (a.nchunk - ch_lo < it_hi - it) ? a.nchunk - ch_lo : it_hi - it;
      ch_hi = ch_lo + n;
      it += n;
    }
    const SkTile tc = sk_decode_tile(a, tile, MB);
    const int b = tc.b, t = tc.t, q0 = tc.q0, w0 = tc.w0, mb0 = tc.mb0;

    const StageSlots sl = conv_stage_slots(a, q0, w0, tid);

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][i][e] = 0.f;

    for (int ch = ch_lo; ch < ch_hi; ++ch) {
      if (!first_stage) __syncthreads();  // every wave is done reading the previous chunk
      first_stage = false;
      HPVG_PH(3)
      // Retire every outstanding vector-memory write to a VGPR (the A prefetch issued past the previous chunk's last tap
      // is never consumed) BEFORE the staging loop: the loop reuses those registers for addresses, and a write-after-
      // write hazard the compiler cannot count across the loop's back edge costs an s_waitcnt vmcnt(0) per DMA, i.e.
      // serialises all 72 loads of the chunk (measured: 1.98 -> 2.22 ms at stage 9).
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      conv_stage_chunk<CC, KT, VAR == VAR_PRO>(a, xs, sl, ch, ch == ch_lo, b, t, tid, wave);
      HPVG_PH(0)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the asm LDS-DMA pieces are invisible to the compiler's counters
      __syncthreads();
      HPVG_PH(1)

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
#pragma unroll
            for (int m = 0; m < MB; ++m) an[m] = wpt[m * 64];
            const float* xt = xl + dt * PL + dh * RS + dw;
            const float* xn = (dw < 2) ? xt + 1 : (dh < 2 ? xl + dt * PL + (dh + 1) * RS : xl + (dt + 1 < KT ? dt + 1 : 0) * PL);
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) {
              const float* nx = (cp + 1 < CP) ? xt + (2 * (cp + 1) * KT) * PL : xn;
#pragma unroll
              for (int i = 0; i < NB; ++i) bv[(cp + 1) & 1][i] = nx[i * 128];
              __builtin_amdgcn_sched_barrier(0);
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
      HPVG_PH(2)
    }

    if (ch_lo != 0 || ch_hi != a.nchunk) {
      // ---- part of a tile: raw accumulators, register order (each store instruction writes 256 contiguous bytes)
      const int seg = (tile == first_sk_tile) ? 0 : 1;
      float* dst = a.skpart + ((long)(g * 2 + seg) * (MB * NB * 16)) * 256 + tid;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((m * NB + i) * 16 + e) * 256] = acc[m][i][e];
      continue;
    }
    // ---- whole tile: bias (added AFTER the accumulation, as every torch conv backend does: a BatchNorm fed by a
    // nearly constant channel amplifies the difference between (bias + sum) and (sum + bias) to percent level),
    // optional LeakyReLU, masked store
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
      const int Q = q0 + q;
      const int gh = Q / RS, ww = Q - gh * RS;
      const int gw = w0 + ww;
      const bool ok = q < a.L && ww < a.Tw && gh < a.H && gw < a.W;
      const long sp = (long)t * HW + (long)gh * a.W + gw;
      const long wi = ((long)b * a.mbtot + mb0) * a.T * HW + sp;   // mask word of this position, m-tile mb0; m-tile + m: + m * T*H*W
      if constexpr (VAR == VAR_BITS) {
        // (every lane takes part in the cross-half exchange below, so no early exit before it)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          unsigned word = 0;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float val = acc[m][i][e] + bias_r[m][e];      // sign(LeakyReLU(v)) = sign(v)
            word |= (val > 0.f ? 1u : 0u) << ((e & 3) + 8 * (e >> 2) + 4 * half);
          }
          word |= (unsigned)__shfl_xor((int)word, 32, 64);      // the two half-waves hold complementary channels of one position
          if (ok && half == 0 && (mb0 + m) * 32 < a.Cout) a.bits_out[wi + (long)m * a.T * HW] = word;
        }
      }
      if (!ok) continue;
      unsigned mword[MB];
      if constexpr (VAR == VAR_MASK) {
        if (a.mask_bits) {
#pragma unroll
          for (int m = 0; m < MB; ++m) mword[m] = (mb0 + m) * 32 < a.Cout ? a.mask_bits[wi + (long)m * a.T * HW] : 0u;
        }
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = (mb0 + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (co < a.Cout) {
            float val = acc[m][i][e] + bias_r[m][e];
            if (a.out_lrelu) val = hpvg_lrelu(val);
            const long oi = ((long)b * a.Cout + co) * a.T * HW + sp;
            if constexpr (VAR == VAR_MASK) {
              if (a.mask_bits) val *= ((mword[m] >> ((e & 3) + 8 * (e >> 2) + 4 * half)) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
              else val *= a.mask[oi] > 0.f ? 1.f : HPVG_LRELU_SLOPE;
            }
            a.y[oi] = val;
          }
        }
      }
    }
  }
  HPVG_PH_END
  HPVG_TRACE_END
}

// Finishes the tiles conv_mfma_kernel computed in parts: one workgroup per stream-K tile, partial slabs summed in
// workgroup order, then the same epilogue.  Tiles computed whole by one workgroup were already stored: nothing to do.
__global__ __launch_bounds__(256) void conv_fixup_kernel(const ConvFwdArgs a, int S, int MB, int NB) {
  const int r = blockIdx.x;                   // stream-K tile
  const int mi = blockIdx.y;                  // (m-tile, position block) of the accumulator slab
  const int m = mi / NB, i = mi - m * NB;
  const long Isk = (long)(a.ntl - a.skbase) * a.nchunk;
  const long i0 = (long)r * a.nchunk, i1 = i0 + a.nchunk - 1;
  const int g0 = (int)(((i0 + 1) * S + Isk - 1) / Isk) - 1;  // owner of the tile's first item
  const int g1 = (int)(((i1 + 1) * S + Isk - 1) / Isk) - 1;  // ... and of its last
  if (g0 == g1) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const SkTile tc = sk_decode_tile(a, a.skbase + r, MB);
  const long HW = (long)a.H * a.W;
  const int RS = a.RS;
  const int blk = wave + 4 * i;
  const int q = blk * 32 + l31;
  const int Q = tc.q0 + q;
  const int gh = Q / RS, ww = Q - gh * RS;
  const int gw = tc.w0 + ww;
  if (!(q < a.L && ww < a.Tw && gh < a.H && gw < a.W)) return;
  const long slab = (long)MB * NB * 16 * 256;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = 0.f;
  for (int g = g0; g <= g1; ++g) {
    const long st = (long)g * Isk / S, en = (long)(g + 1) * Isk / S;
    if (st == en) continue;  // workgroup without stream-K items
    const int seg = (st / a.nchunk == r) ? 0 : 1;
    const float* src = a.skpart + (long)(g * 2 + seg) * slab + (long)mi * 16 * 256 + tid;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += src[e * 256];
  }
  const long sp = (long)tc.t * HW + (long)gh * a.W + gw;
  const long wi = ((long)tc.b * a.mbtot + tc.mb0 + m) * a.T * HW + sp;
  const unsigned mword = a.mask_bits ? a.mask_bits[wi] : 0u;
  unsigned word = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = (tc.mb0 + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
    if (co >= a.Cout) continue;
    float val = v[e];
    if (a.bias) val += a.bias[co];
    word |= (val > 0.f ? 1u : 0u) << ((e & 3) + 8 * (e >> 2) + 4 * half);
    if (a.out_lrelu) val = hpvg_lrelu(val);
    const long oi = ((long)tc.b * a.Cout + co) * a.T * HW + sp;
    if (a.mask_bits) val *= ((mword >> ((e & 3) + 8 * (e >> 2) + 4 * half)) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
    else if (a.mask) val *= a.mask[oi] > 0.f ? 1.f : HPVG_LRELU_SLOPE;
    a.y[oi] = val;
  }
  if (a.bits_out) {
    // the partner half-wave holds the other 16 channels of this position; lanes that returned early above belong to
    // positions outside the tile in BOTH halves (the test does not depend on `half`), so the exchange is well defined
    word |= (unsigned)__shfl_xor((int)word, 32, 64);
    if (half == 0) a.bits_out[wi] = word;
  }
}

#include "conv_wino.inl"
#include "conv_wino2d.inl"
#include "conv_wino2r.inl"

// ------------------------------------------------------------------------------------------
// Narrow-output convolution (Cout <= 4): the tails 64->3 / 64->1 (networks_3d.py:175,341,362) and the backward-data
// pass of the 3->64 heads.  With output channels as the GEMM's M these waste >= 90 % of every 32-row tile (1.07 ms at
// stage 9).  Here the roles are swapped and the 9 in-plane taps move into N:
//     P[pos][(o,dh,dw)] = sum_{c,dt} x[c][t+dt-pt][pos] * w[o][c][dt][dh][dw]          (GEMM: M = positions,
//     y[o][h][w]        = bias[o] + sum_{dh,dw} P[(h+dh-1, w+dw-1)][(o,dh,dw)]           N = 9*Cout <= 36, K = Cin*KT)
// Every input element is an A operand exactly once per tile (read straight from global, 128-byte segments, no LDS
// staging and no tap replication); P of the tile + halo goes through LDS once and the shifted 9-term sum is the
// epilogue.  84 % of the MFMA work is useful for Cout = 3 (27 of 32 columns).
// Weights: wn[kstep][ntile][lane] = B fragment (k = (lane/32)*ksteps + kstep -> (c, dt); n = 32*ntile + lane%32 -> (o, dh, dw)).
template <int KT, int NT>
__global__ __launch_bounds__(256, 2) void conv_narrow_kernel(const ConvFwdArgs a, int ksteps, int pitch) {
  extern __shared__ __attribute__((aligned(16))) float P[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const long HW = (long)a.H * a.W;
  const int RS = a.RS;
  const int CO9 = a.Cout * 9;
  const int pt = (KT == 3 ? 1 : 0);

  const int tile = hpvg_xcd_remap(blockIdx.x, gridDim.x);
  const SkTile tc = sk_decode_tile(a, tile, 1);
  const int b = tc.b, t = tc.t, h0 = tc.q0 / RS, w0 = tc.w0;  // 2-D tiles here: L = Th * RS
  const int plload = (a.Th + 2) * RS;
  const int nblk = (plload + 31) / 32;
  const float* xb0 = a.x + (long)b * a.Cin * a.T * HW;

  // K order: half-wave h of MFMA step s holds k = h*Kh + s (Kh = ksteps), i.e. channel h*Ch + s/KT (Cin even: Ch = Cin/2)
  // and time tap s%KT: the (channel, plane) offset of a step is the same scalar for both halves and the lane part of
  // the address is a per-block constant -> scalar-base + 32-bit lane offset loads, no per-load vector address math.
  const int Kh = ksteps;
  for (int blk = wave; blk < nblk; blk += 4) {
    const int p = blk * 32 + l31;
    const int hh = p / RS, ww = p - hh * RS;
    const int gh = h0 + hh - 1, gw = w0 + ww - 1;
    const bool inimg = p < plload && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    const int kofs = half * Kh;                 // first k of this half-wave
    const int c_h = kofs / KT, dt_h = kofs - c_h * KT;  // (channel, time tap) of k = kofs
    // lane part of the address in bytes: (c_h, dt_h) plane + pixel (fits 32 bits: one sample is < 4 GB)
    const unsigned lofs = (unsigned)((((long)c_h * a.T + dt_h) * HW + (inimg ? gh * a.W + gw : 0)) * 4);
    const char* xbase = reinterpret_cast<const char*>(a.x + ((long)b * a.Cin * a.T + (t - pt)) * HW);
    const float* wq = a.wp + lane;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    // K loop in chunks of KC steps, two register sets: the loads of chunk j+1 (inputs straight from global, weight
    // fragments from L1/L2) are in flight while the MFMAs of chunk j issue
    constexpr int KC = 16;
    float ab[2][KC], bb[2][NT][KC];
    auto load_chunk = [&](float (&av)[KC], float (&bv)[NT][KC], int s0) {
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const int s = s0 + j;                    // uniform
        const int cs = s / KT, ds = s - cs * KT; // uniform: channel / time-tap advance relative to (c_h, dt_h)
        // k = kofs + s -> (c, dt) = (c_h, dt_h) advanced by s steps of the (c, dt) odometer
        const int dtt = dt_h + ds;
        const int c = c_h + cs + (dtt >= KT ? 1 : 0);
        const int dt = dtt >= KT ? dtt - KT : dtt;
        const int tt = t + dt - pt;
        const bool ok = inimg && s < ksteps && c < a.Cin && tt >= 0 && tt < a.T;
        // byte offset of (c, dt) relative to (c_h, dt_h): ((c - c_h)*T + (dt - dt_h)) * HW * 4
        const long rel = (((long)(c - c_h) * a.T + (dt - dt_h)) * HW) * 4;
        float v = ok ? *reinterpret_cast<const float*>(xbase + rel + lofs) : 0.f;
        if (a.in_scale != nullptr && ok) {  // fused BatchNorm-apply (+ LeakyReLU) of the producer; padding stays zero
          v = v * a.in_scale[c] + a.in_shift[c];
          if (a.in_lrelu) v = hpvg_lrelu(v);
        }
        av[j] = v;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt][j] = s < ksteps ? wq[(s * NT + nt) * 64] : 0.f;
      }
    };
    auto mma_chunk = [&](const float (&av)[KC], const float (&bv)[NT][KC]) {
#pragma unroll
      for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[nt][j], acc[nt], 0, 0, 0);
    };
    load_chunk(ab[0], bb[0], 0);
    for (int s0 = 0; s0 < ksteps; s0 += 2 * KC) {
      load_chunk(ab[1], bb[1], s0 + KC);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(ab[0], bb[0]);
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(ab[0], bb[0], s0 + 2 * KC);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(ab[1], bb[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // C/D layout: column (n) = lane&31, row (position in the block) = (e&3) + 8*(e>>2) + 4*half
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + l31;
      if (n < CO9) {
#pragma unroll
        for (int e = 0; e < 16; ++e) P[n * pitch + blk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] = acc[nt][e];
      }
    }
  }
  __syncthreads();

  // ---- epilogue: shifted 9-term sums, bias (after the accumulation), optional LeakyReLU, masked store
  const int npos = a.Th * a.Tw;
  for (int idx = tid; idx < npos; idx += 256) {
    const int hh = idx / a.Tw, ww = idx - hh * a.Tw;
    const int gh = h0 + hh, gw = w0 + ww;
    if (gh >= a.H || gw >= a.W) continue;
    const long sp = (long)t * HW + (long)gh * a.W + gw;
    for (int o = 0; o < a.Cout; ++o) {
      float v = 0.f;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) v += P[(o * 9 + dh * 3 + dw) * pitch + (hh + dh) * RS + ww + dw];
      if (a.bias) v += a.bias[o];
      if (a.out_lrelu) v = hpvg_lrelu(v);
      const long oi = ((long)b * a.Cout + o) * a.T * HW + sp;
      if (a.mask) v *= a.mask[oi] > 0.f ? 1.f : HPVG_LRELU_SLOPE;
      a.y[oi] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------
// Narrow-output convolution, second generation (Cout <= 4, Cin even, no prologue / mask): same GEMM as conv_narrow_kernel
// (M = positions, N = (o, dh, dw), K = (c, dt); P through LDS, shifted 9-term sums as the epilogue) with the input
// streamed by 16-BYTE loads.  conv_narrow_kernel issues one dword load per lane per MFMA (vector-memory issue bound:
// 0.43 ms at stage 9 against an MFMA floor of ~0.12 ms).  Here a lane loads FOUR consecutive positions of one
// (channel, plane) with one global_load_dwordx4 and feeds them to four MFMAs as the A operand of four different
// 32-position blocks: MFMA j of a 128-position group computes the P rows of positions {128 g + 4 i + j, i = 0..31} - the
// M index of an MFMA is just a label, the P store puts every row where it belongs.  The weight fragment of a k-step is
// loaded once per four MFMAs.  For that every group of 4 positions must be contiguous in memory and inside the image:
// a tile is (Th + 2) rows x RS columns (RS % 4 == 0) of a column WINDOW that lies inside [0, W) - no halo columns are
// fetched from outside the image (band 0 starts at column 0, the last band ends at W; an output at the image border skips
// its out-of-image taps in the epilogue); rows outside the image are clamped for the load and zeroed by select.
struct Narrow2Geom {
  int RS, Th, nth, nb, npos, G, pitch;
  int ws[16], ob[16], on[16];  // per band: window start column, first output column, number of output columns
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte aligned address
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
template <int JB> struct AVecU;
template <> struct AVecU<4> { typedef f32x4u type; };
template <> struct AVecU<2> { typedef f32x2u type; };

template <int KT, int NT, int JB>
__global__ __launch_bounds__(256, 2) void conv_narrow2_kernel(const ConvFwdArgs a, const Narrow2Geom g, int ksteps) {
  typedef typename AVecU<JB>::type AV;
  constexpr int GP = 32 * JB;                 // positions per group (one lane: JB consecutive positions)
  constexpr int KC = KT == 3 ? 6 : 8;         // k-steps per register set (a multiple of KT: the plane of step j is j % KT)
  extern __shared__ __attribute__((aligned(16))) float P[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const long HW = (long)a.H * a.W;
  const int RS = g.RS, pitch = g.pitch;
  const int CO9 = a.Cout * 9;
  const int pt = (KT == 3 ? 1 : 0);

  // tile order: band fastest, then TIME, then the row block: consecutive workgroups (one XCD) share input planes t-1, t, t+1
  int r = hpvg_xcd_remap(blockIdx.x, gridDim.x);
  const int band = r % g.nb; r /= g.nb;
  const int t = r % a.T; r /= a.T;
  const int th = r % g.nth;
  const int b = r / g.nth;
  const int h0 = th * g.Th, wsb = g.ws[band];
  const float* xs = a.x + ((long)b * a.Cin * a.T + (t - pt)) * HW;      // plane (c = 0, dt = 0) of this tile
  const int chalf = (ksteps / KT) * half;                                // first channel of this half-wave's K range

  for (int grp = wave; grp < g.G; grp += 4) {
    const int p = GP * grp + JB * l31;
    const int pc = p < g.npos ? p : g.npos - JB;
    const int row = pc / RS, col = pc - row * RS;
    const int gh = h0 - 1 + row;
    const bool valid = p < g.npos && gh >= 0 && gh < a.H;
    const int ghc = gh < 0 ? 0 : (gh >= a.H ? a.H - 1 : gh);
    const float* lp = xs + (long)chalf * a.T * HW + (long)ghc * a.W + wsb + col;
    const float* wq = a.wp + lane;
    f32x16 acc[JB][NT];
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][nt][e] = 0.f;
    bool tok[KT];
#pragma unroll
    for (int dt = 0; dt < KT; ++dt) tok[dt] = (t + dt - pt >= 0) && (t + dt - pt < a.T);
    AV av[2][KC];
    float bv[2][KC][NT];
    auto load_chunk = [&](AV (&A)[KC], float (&Bf)[KC][NT], int s0) {
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const int s = s0 + j;
        const int dt = j % KT;                                // s0 is a multiple of KT
        const int cs = s0 / KT + j / KT;                      // channel of this step inside the half-wave's range
        if (s < ksteps && tok[dt]) {                          // uniform
          A[j] = *reinterpret_cast<const AV*>(lp + ((long)cs * a.T + dt) * HW);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) Bf[j][nt] = wq[(s * NT + nt) * 64];
        }
      }
    };
    auto mma_chunk = [&](const AV (&A)[KC], const float (&Bf)[KC][NT], int s0) {
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const int s = s0 + j;
        if (s < ksteps && tok[j % KT]) {
#pragma unroll
          for (int jj = 0; jj < JB; ++jj) {
            const float v = valid ? A[j][jj] : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[jj][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, Bf[j][nt], acc[jj][nt], 0, 0, 0);
          }
        }
      }
    };
    load_chunk(av[0], bv[0], 0);
    for (int s0 = 0; s0 < ksteps; s0 += 2 * KC) {
      load_chunk(av[1], bv[1], s0 + KC);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(av[0], bv[0], s0);
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(av[0], bv[0], s0 + 2 * KC);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(av[1], bv[1], s0 + KC);
      __builtin_amdgcn_sched_barrier(0);
    }
    // C/D layout: column (n) = lane & 31, row i = (e & 3) + 8 * (e >> 2) + 4 * half; row i of MFMA jj = position GP*grp + JB*i + jj
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + l31;
      if (n < CO9) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) P[n * pitch + GP * grp + JB * ((e & 3) + 8 * (e >> 2) + 4 * half) + jj] = acc[jj][nt][e];
      }
    }
  }
  __syncthreads();

  // ---- epilogue: shifted 9-term sums (taps outside the image skipped: the window holds no zero-padding columns), bias
  // after the accumulation, optional LeakyReLU
  const int ob = g.ob[band], on = g.on[band];
  const int nout = g.Th * on;
  for (int idx = tid; idx < nout; idx += 256) {
    const int hh = idx / on, cc = idx - hh * on;
    const int gh = h0 + hh, gw = ob + cc;
    if (gh >= a.H) continue;
    const long sp = (long)t * HW + (long)gh * a.W + gw;
    for (int o = 0; o < a.Cout; ++o) {
      float v = 0.f;
#pragma unroll
      for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          const int gwt = gw + dw - 1;
          if (gwt >= 0 && gwt < a.W) v += P[(o * 9 + dh * 3 + dw) * pitch + (hh + dh) * RS + (gwt - wsb)];
        }
      if (a.bias) v += a.bias[o];
      if (a.out_lrelu) v = hpvg_lrelu(v);
      a.y[((long)b * a.Cout + o) * a.T * HW + sp] = v;
    }
  }
}

// tile plan of conv_narrow2_kernel; returns false when the shape does not fit (then conv_narrow_kernel runs)
bool plan_narrow2(int B, int Cin, int Cout, int T, int H, int W, int KT, Narrow2Geom* out, long* ntiles) {
  if ((Cin & 1) || W < 4) return false;
  const int NT = hpvg_cdiv(9 * Cout, 32);
  const int JB = NT == 1 ? 4 : 2;
  const int GP = 32 * JB;
  double best = 1e300;
  bool found = false;
  // bands of a window width RS: band 0 = window [0, RS), outputs columns 0 .. RS-2 (all W of them when RS == W); a middle
  // band outputs RS-2 columns from the window that starts one column before its first output; a band whose remaining
  // columns number <= RS-1 is the last: its window is pulled back to end at W (the right image border needs no neighbour)
  auto bands = [&](int RS, Narrow2Geom* q) -> int {
    int nb = 0, o = 0;
    while (o < W) {
      if (nb == 16) return 0;
      int n, ws;
      if (nb == 0) { n = (RS == W) ? W : RS - 1; ws = 0; }
      else if (W - o <= RS - 1) { n = W - o; ws = W - RS; }
      else { n = RS - 2; ws = o - 1; }
      if (n < 1) return 0;
      if (q) { q->ws[nb] = ws; q->ob[nb] = o; q->on[nb] = n; }
      o += n;
      ++nb;
    }
    return nb;
  };
  for (int RS = 4; RS <= 256 && RS <= W; RS += 4) {
    const int nb = bands(RS, nullptr);
    if (nb == 0) continue;
    for (int Th = 1; Th <= H; ++Th) {
      const int nth = hpvg_cdiv(H, Th);
      if (Th != hpvg_cdiv(H, nth)) continue;
      const int npos = (Th + 2) * RS;
      const int G = hpvg_cdiv(npos, GP);
      if (G > 16) break;
      const int pitch = (G * GP) | 1;
      if ((size_t)(9 * Cout) * pitch * sizeof(float) > 76 * 1024) break;
      const long nt = (long)B * T * nth * nb;
      const double rounds = hpvg_cdiv(G, 4);
      const double waves = nt <= 2L * HPVG_NUM_CU ? 1.0 : (double)nt / (2.0 * HPVG_NUM_CU);
      const double cost = waves * (rounds * GP + 30.0);     // MFMA work of the slowest wave + a fixed per-tile cost (epilogue, barrier)
      if (cost < best) {
        best = cost;
        found = true;
        Narrow2Geom q{};
        q.RS = RS; q.Th = Th; q.nth = nth; q.nb = nb; q.npos = npos; q.G = G; q.pitch = pitch;
        bands(RS, &q);
        *out = q;
        *ntiles = nt;
      }
    }
  }
  if (!found) return false;
  // every output column needs its in-image neighbours inside the band's window
  for (int k = 0; k < out->nb; ++k) {
    const int lo = out->ob[k] - 1 < 0 ? 0 : out->ob[k] - 1;
    const int hi = out->ob[k] + out->on[k] > W - 1 ? W - 1 : out->ob[k] + out->on[k];
    if (out->on[k] < 0 || lo < out->ws[k] || hi >= out->ws[k] + out->RS) return false;
  }
  return true;
}

template <int KT>
int launch_conv_narrow2(const ConvFwdArgs& a, const Narrow2Geom& g, long ntiles, hipStream_t s) {
  const int ksteps = hpvg_cdiv(a.Cin * KT, 2);
  const size_t lds = (size_t)(9 * a.Cout) * g.pitch * sizeof(float);
#define HPVG_NARROW2(NT, JB)                                                                                         \
  {                                                                                                                  \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_narrow2_kernel<KT, NT, JB>),                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                 \
        (void)hipGetLastError();                                                                                     \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((conv_narrow2_kernel<KT, NT, JB>), dim3((unsigned)ntiles), dim3(256), lds, s, a, g, ksteps);  \
  }
  if (9 * a.Cout <= 32) HPVG_NARROW2(1, 4) else HPVG_NARROW2(2, 2)
#undef HPVG_NARROW2
  return hpvg_launch_status();
}

// B fragments of the narrow kernel; transpose_flip as in conv_pack_kernel
__global__ void conv_pack_narrow_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, float* __restrict__ wn,
                                        int Cin_k, int Cout_k, int KT, int NT, int ksteps, int transpose_flip, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  long r = idx >> 6;
  const int nt = r % NT;
  const int s = (int)(r / NT);
  const int k = (lane >> 5) * ksteps + s;  // half-wave h holds k = h*ksteps + s (see conv_narrow_kernel)
  const int n = nt * 32 + (lane & 31);
  const int c = k / KT, dt = k - c * KT;
  const int o = n / 9, tap = dt * 9 + (n - o * 9);
  const int taps = KT * 9;
  float val = 0.f;
  if (o < Cout_k && c < Cin_k) {
    if (!transpose_flip) val = w[((long)o * Cin_k + c) * taps + tap];
    else val = w[((long)c * Cout_k + o) * taps + (taps - 1 - tap)];
    if (inv_scale) val *= inv_scale[0];
  }
  wn[idx] = val;
}

// ------------------------------------------------------------------------------------------
// weight pack: natural [Cout][Cin][taps] -> MFMA A-fragment order
//   wp[chunk][tap][mblock][lane][cp] = Wsrc[o = mblock*32 + (lane&31)][c = chunk*CC + 2cp + (lane>>5)][tap]
// transpose_flip=1 packs the backward-data weights (Wsrc[o'][c'][tap] = W[c'][o'][ntaps-1-tap]).
// A device scalar `inv_scale` (1/sigma of spectral norm) is folded in when given.
// Behind the direct pack (`total_direct` floats) follow the Winograd U fragments of the same weight when the layer is wide
// enough for conv_wino_kernel (conv_wino.inl); which of the two a launch reads is decided per shape in conv_fwd_impl.
__global__ void conv_pack_kernel(const float* __restrict__ w, const float* __restrict__ inv_scale, float* __restrict__ wp,
                                 int Cin_k, int Cout_k, int taps, int CC, int nchunk, int mbtot, int transpose_flip,
                                 long total_direct, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  if (idx >= total_direct) {
    const long iw = idx - total_direct;
    const long nw1 = ((long)((Cin_k + WINO_CC - 1) / WINO_CC) * (taps / 9) * 3 * 4 + 4) * mbtot * 64 * (WINO_CC / 2);   // wino_pack_floats
    float val = iw < nw1 ? wino_pack_value(w, iw, Cin_k, Cout_k, taps / 9, (Cin_k + WINO_CC - 1) / WINO_CC, mbtot, transpose_flip)
                         : wino2_pack_value(w, iw - nw1, Cin_k, Cout_k, (Cin_k + 3) / 4, 2 * ((mbtot + 1) / 2), transpose_flip);
    if (inv_scale) val *= inv_scale[0];
    wp[idx] = val;
    return;
  }
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

// the same for up to HPVG_PACK_BATCH_MAX weights of ONE layer shape in a single launch (blockIdx.y = item); flip per item
struct PackBatchArgs {
  const float* w[HPVG_PACK_BATCH_MAX];
  float* wp[HPVG_PACK_BATCH_MAX];
  int flip[HPVG_PACK_BATCH_MAX];
};
__global__ void conv_pack_batch_kernel(const PackBatchArgs a, int C, int taps, int CC, int nchunk, int mbtot, long total_direct,
                                       long total) {
  // square layers only (Cin == Cout == C): the forward and the flipped pack then share every size
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int it = blockIdx.y;
  const float* __restrict__ w = a.w[it];
  const int transpose_flip = a.flip[it];
  if (idx >= total_direct) {
    const long iw = idx - total_direct;
    const long nw1 = ((long)((C + WINO_CC - 1) / WINO_CC) * (taps / 9) * 3 * 4 + 4) * mbtot * 64 * (WINO_CC / 2);   // wino_pack_floats
    a.wp[it][idx] = iw < nw1 ? wino_pack_value(w, iw, C, C, taps / 9, (C + WINO_CC - 1) / WINO_CC, mbtot, transpose_flip)
                             : wino2_pack_value(w, iw - nw1, C, C, (C + 3) / 4, 2 * ((mbtot + 1) / 2), transpose_flip);
    return;
  }
  const int CP = CC / 2;
  long r = idx;
  const int cp = r % CP; r /= CP;
  const int lane = r % 64; r /= 64;
  const int mb = r % mbtot; r /= mbtot;
  const int tap = r % taps; r /= taps;
  const int ch = (int)r;
  float val = 0.f;
  if (ch < nchunk) {
    const int o = mb * 32 + (lane & 31);
    const int c = ch * CC + 2 * cp + (lane >> 5);
    if (o < C && c < C) {
      if (!transpose_flip) val = w[((long)o * C + c) * taps + tap];
      else val = w[((long)c * C + o) * taps + (taps - 1 - tap)];
    }
  }
  a.wp[it][idx] = val;
}

inline int conv_cc(int Cin) { return Cin <= 4 ? 4 : 8; }

// floats of the direct kernel's A-fragment pack (the Winograd fragments, when the layer has them, follow it)
inline size_t direct_pack_floats(int Cin, int Cout, int KT) {
  const int CC = conv_cc(Cin);
  const int nchunk = hpvg_cdiv(Cin, CC);
  const int mbtot = hpvg_cdiv(Cout, 32);
  const size_t per_tap = (size_t)mbtot * 64 * (CC / 2);
  return ((size_t)nchunk * KT * 9 + 2) * per_tap;  // +2 taps of zero tail padding (A prefetch runs one tap / one m-tile ahead)
}

struct Plan {
  int Th, Tw, RS, PL, L, qstride, nrange, ntw, nblocks, NB, MB, gridy, nj;  // Th != 0 only for the narrow kernel's 2-D tiles
  size_t lds;  // bytes of the LDS tile buffer
  int stg;     // conv_wino_kernel: 2 = rows-as-in-memory staging with 16-byte pieces (conv_wino.inl), else the halo'd bands
};

constexpr long CONV_SLOTS = 2L * HPVG_NUM_CU;  // co-resident workgroups: two per CU (256 VGPRs each, LDS <= 80 KB)
inline long conv_slots(int MB, int NB) { return (long)HPVG_CONV_WGS(MB, NB) * HPVG_NUM_CU; }

// Tile planner.  streamk = true: the launch is S = min(items, 512) persistent workgroups that share the (tile, chunk)
// items evenly (needs the partial-slab workspace).  streamk = false: one workgroup per tile, dealt out by the
// hardware dispatcher (no workspace; grids of 1-5 tiles per slot are quantised - the reason stream-K exists).
// Cost in us; constants chosen to minimise the REGRET of the plan picked over the measured landscapes of stages 3-9 at
// B = 2 and 4 (tools/ab_conv_plan.sh: every band count x NB forced; mean regret 0.3 %, worst 1.6 %): an item costs its
// MFMA rounds (6.5 us per 32-position x 32-channel x 8-channel-chunk round when two workgroups share the CU's matrix
// pipe, half that when alone) + ~6 us of staging / barrier work the co-resident workgroup does not hide (2 when alone)
// + the input planes it stages (2 us per 1000 positions); tiles cut across workgroups travel through HBM as raw accumulator slabs (16 KB per 32x32 block,
// written by the main kernel, read back by the fix-up: ~5.5 us per block at the chip's share of bandwidth).
Plan plan_conv_search(int B, int Cin, int Cout, int T, int H, int W, int KT, bool streamk, bool narrow) {
  const int CC = conv_cc(Cin);
  const int nchunk = hpvg_cdiv(Cin, CC);
  const int mbtot = hpvg_cdiv(Cout, 32);
  Plan best{};
  double best_cost = 1e300;
  // development knobs: restrict the search to one NB / MB
  static const int only_nb = [] { const char* e = getenv("HPVG_PLAN_NB"); return e ? atoi(e) : 0; }();
  static const int only_mb = [] { const char* e = getenv("HPVG_PLAN_MB"); return e ? atoi(e) : 0; }();
  static const int only_ntw = [] { const char* e = getenv("HPVG_PLAN_NTW"); return e ? atoi(e) : 0; }();
  if (narrow) {
    // 2-D tiles (Th x Tw) for conv_narrow_kernel: a GEMM over the tile + halo positions, one workgroup per tile: blocks
    // of 32 positions over 4 waves, K steps of 64 cycles each, ~3 us of epilogue; the P image (9*Cout rows) must leave
    // room for >= 4 workgroups per CU (they hide the load latency)
    for (int Tw = 1; Tw <= W; ++Tw) {
      const int ntw = hpvg_cdiv(W, Tw);
      if (Tw != hpvg_cdiv(W, ntw)) continue;  // only balanced splits of W
      const int RS = Tw + 2;
      for (int Th = 1; Th <= H; ++Th) {
        const int nth = hpvg_cdiv(H, Th);
        if (Th != hpvg_cdiv(H, nth)) continue;
        const int nblk_h = hpvg_cdiv((Th + 2) * RS, 32);
        if ((Th + 2) * RS > NJMAX * 256) break;
        if ((size_t)(9 * Cout) * ((size_t)nblk_h * 32 + 1) * sizeof(float) > 38 * 1024) continue;
        const long ntl = (long)B * T * nth * ntw;
        const double per_tile = hpvg_cdiv(nblk_h, 4) * (double)hpvg_cdiv(Cin * KT, 2) * hpvg_cdiv(9 * Cout, 32) * 0.033 + 3.0;
        const double waves = ntl <= CONV_SLOTS ? 1.0 : (double)ntl / (double)CONV_SLOTS;
        const double cost = waves * per_tile + 1e-3 * (double)ntl;
        if (cost < best_cost - 1e-9) {
          best_cost = cost;
          best = Plan{Th, Tw, RS, (Th + 2) * RS, Th * RS, Th * RS, nth, ntw, hpvg_cdiv(Th * RS, 32), 4, 1, 1, hpvg_cdiv((Th + 2) * RS, 256), 0};
        }
      }
    }
    return best;
  }
  // MFMA kernel: W is cut into ntw balanced bands of Tw columns; a band plane, flattened row by row with stride RS = Tw+2
  // (its two halo columns are the junk GEMM columns), is cut into nrange balanced ranges of L <= NB*128 positions.  Wide
  // bands waste few junk columns, and a range - unlike a whole number of rows - fills its 32-position blocks.
  for (int MB = (mbtot >= 2 ? 2 : 1); MB >= 1; --MB) {
    if (only_mb && MB != only_mb && mbtot >= 2) continue;
    const int gridy = hpvg_cdiv(mbtot, MB);
    for (int ntw = 1; ntw <= W; ++ntw) {
      const int Tw = hpvg_cdiv(W, ntw);
      if (ntw > 1 && hpvg_cdiv(W, ntw - 1) == Tw) continue;  // same band width as the previous ntw: more bands, no gain
      if (only_ntw && ntw != only_ntw) continue;
      const int RS = Tw + 2;
      const long flat = (long)(H - 1) * RS + Tw;  // flattened positions that hold outputs
      // candidate tiles of this band.  3x3x3: balanced ranges of at most NB*128 positions.  3x3 (a third of the MFMA work
      // per staged plane, so the cost constants fitted on 3x3x3 launches do not rank these well): whole rows only, the
      // tile family the 2-D path was tuned and measured with.
      const int ncand = KT == 3 ? 3 : H;
      for (int cand = 0; cand < ncand; ++cand) {
        int NB, L, nrange, qstride;
        if (KT == 3) {
          NB = 1 << cand;
          nrange = hpvg_cdiv(flat, NB * 128);
          L = hpvg_cdiv(flat, nrange);
          qstride = L;
        } else {
          const int Th = cand + 1;
          nrange = hpvg_cdiv(H, Th);
          if (Th != hpvg_cdiv(H, nrange)) continue;  // only balanced splits of H
          L = Th * RS;
          const int nblk = (L - 3) / 32 + 1;          // last output position of the tile: (Th-1)*RS + Tw - 1
          if (nblk > 16) break;
          const int rounds = hpvg_cdiv(nblk, 4);
          NB = rounds <= 1 ? 1 : (rounds == 2 ? 2 : 4);
          qstride = L;
          if (L > NB * 128) L = NB * 128;             // the (at most two) positions cut off are halo columns
        }
        if (only_nb && NB != only_nb) continue;
        const int Lmax = NB * 128;
        const int plload = L + 2 * RS + 2;
        if (plload > NJMAX * 256) continue;
        const int PL = Lmax + 2 * RS + 2;  // lanes past L still read (junk) operands inside the buffer
        const size_t lds = (size_t)CC * KT * PL * sizeof(float);
        if (lds > (HPVG_CONV_WGS(MB, NB) == 4 ? 38 : 80) * 1024) continue;
        const long ntl = (long)B * T * nrange * ntw * gridy;
        const double stage_us = 0.002 * plload;
        double cost;
        if (streamk) {
          // S co-resident workgroups share the (tile, chunk) items evenly.  Cost in us, constants fitted to measured
          // launches (tools/perf_conv.py, stages 5-9, within 5 %): an item costs its MFMA rounds (6.5 us per 32-position
          // x 32-channel x 8-channel-chunk round when two workgroups share the CU's matrix pipe, half that when alone)
          // + ~6 us of staging / barrier work the co-resident workgroup does not hide + the input planes it stages;
          // tiles cut across workgroups travel through HBM as raw accumulator slabs (16 KB per 32x32 block, written by
          // the main kernel, read back by the fix-up: ~5.5 us per block at the chip's share of bandwidth)
          const long items_tot = ntl * nchunk;
          const long slots = conv_slots(MB, NB);
          const long S = items_tot < slots ? items_tot : slots;
          const long ndp = ntl / S;
          const long rem = (ntl - ndp * S) * nchunk;
          const double items = (double)(ndp * nchunk) + (double)((rem + S - 1) / S);
          const bool paired = S > HPVG_NUM_CU;
          const double per_item = (double)NB * MB * (paired ? 6.5 : 3.0) + (paired ? 6.0 : 2.0) + stage_us;
          const double tiles = (double)(ndp + (rem ? 1 : 0));
          double parts = 0.0;
          if (rem) {
            parts = 1.0 + (double)rem / (double)S / (double)nchunk;
            if (parts > 2.0) parts = 2.0;
          }
          cost = items * per_item + tiles * 0.4 * NB * MB + parts * 5.5 * NB * MB + 1e-4 * (double)ntl;
        } else {
          // one workgroup per tile: whole tiles per CU, pairs run together, an odd one runs alone at the end
          const long per_cu = (ntl + HPVG_NUM_CU - 1) / HPVG_NUM_CU;
          const double tile_paired = nchunk * ((double)NB * MB * 6.5 + 6.0 + stage_us) + 0.4 * NB * MB;
          const double tile_alone = nchunk * ((double)NB * MB * 3.0 + 2.0 + stage_us) + 0.4 * NB * MB;
          cost = (double)(per_cu / 2) * tile_paired + (double)(per_cu & 1) * tile_alone + 1e-4 * (double)ntl;
        }
        if (cost < best_cost - 1e-9) {
          best_cost = cost;
          best = Plan{0, Tw, RS, PL, L, qstride, nrange, ntw, hpvg_cdiv(L, 32), NB, MB, gridy, hpvg_cdiv(plload, 256), lds};
        }
      }
    }
  }
  return best;
}

// Tile plan of conv_wino_kernel: bands of an EVEN width Tw (row stride RS = Tw + 2 even: an output pair never straddles a
// row), a band plane cut into balanced ranges of an even L <= NBP*128 positions (NBP*64 pairs).  Same cost form as above
// with the Winograd item: 36 instead of 27 MFMA k-steps per channel pair on half the positions per accumulator set.
Plan plan_wino_search(int B, int Cin, int Cout, int T, int H, int W, int KT, bool streamk) {
  const int nchunk = hpvg_cdiv(Cin, WINO_CC);
  const int mbtot = hpvg_cdiv(Cout, 32);
  const int gridy = hpvg_cdiv(mbtot, 2);
  static const int only_ntw = [] { const char* e = getenv("HPVG_PLAN_NTW"); return e ? atoi(e) : 0; }();
  Plan best{};
  double best_cost = 1e300;
  const int Lmax = WINO_NBP * 128;
  if (g_wino_stg != 1 && (W & 1) == 0 && W >= 2) {
    // staging form 2: one band of the full width, row stride W, no halo columns, a quarter of the staging instructions.
    // Measured (profiles/r02_perf_wino_stg2.txt): -4.2 % at stage 9, -3.7 % at stage 8, +-1-2 % at stages 5-7, where a
    // workgroup sees fewer than two whole tiles: taken from two tiles per co-resident workgroup up (or when forced).
    const long flat = (long)H * W;
    const int nrange = hpvg_cdiv(flat, Lmax);
    int L = hpvg_cdiv(flat, nrange);
    L += L & 1;
    const int PL = 2 * W + Lmax + 8;                          // reads reach sh + 2 * (Lmax/2 - 1) + 2 * W + 3 <= 2 W + Lmax + 5
    const size_t lds = (size_t)WINO_CC * KT * PL * sizeof(float);
    const int ngmax = (L + 2 * W + 2 + 3 + 3) >> 2;           // 16-byte groups of the staged span: one per lane
    const long ntl = (long)B * T * nrange * gridy;
    if (lds <= 80 * 1024 && ngmax <= 256 && (g_wino_stg == 2 || ntl >= 2 * CONV_SLOTS))
      return Plan{0, W, W, PL, L, L, nrange, 1, hpvg_cdiv(L, 32), WINO_NBP, 2, gridy, 1, lds, 2};
  }
  int prev_tw = 0;
  for (int ntw = 1; ntw <= W; ++ntw) {
    int Tw = hpvg_cdiv(W, ntw);
    Tw += Tw & 1;
    if (Tw == prev_tw || (long)(ntw - 1) * Tw >= W) continue;  // same band width as before / an empty last band
    prev_tw = Tw;
    if (only_ntw && ntw != only_ntw) continue;
    const int RS = Tw + 2;
    const long flat = (long)(H - 1) * RS + Tw;
    const int nrange = hpvg_cdiv(flat, Lmax);
    int L = hpvg_cdiv(flat, nrange);
    L += L & 1;
    const int plload = L + 2 * RS + 2;
    if (plload > NJMAX * 256) continue;
    const int PL = Lmax + 2 * RS + 2;   // even; lanes past L still read (junk) operands inside the buffer
    const size_t lds = (size_t)WINO_CC * KT * PL * sizeof(float);
    if (lds > 80 * 1024) continue;
    const long ntl = (long)B * T * nrange * ntw * gridy;
    // (staging weighs half of what it does for the direct kernel: measured landscape, profiles/r02_ab_wino_plan.txt - wide
    // bands with few junk columns win at stages 7-9 although they stage three rows per row of outputs)
    const double stage_us = 0.001 * plload;
    const double mfma_paired = 6.5 * (2.0 * WINO_NBP) * (2.0 / 3.0);  // 64 ch x NBP*64 pairs = 2*NBP direct 32x32 block pairs, at 36/54 of their k-steps
    double cost;
    if (streamk) {
      const long items_tot = ntl * nchunk;
      const long slots = CONV_SLOTS;
      const long S = items_tot < slots ? items_tot : slots;
      const long ndp = ntl / S;
      const long rem = (ntl - ndp * S) * nchunk;
      const double items = (double)(ndp * nchunk) + (double)((rem + S - 1) / S);
      const bool paired = S > HPVG_NUM_CU;
      const double per_item = (paired ? mfma_paired : mfma_paired * 3.0 / 6.5) + (paired ? 6.0 : 2.0) + stage_us;
      const double tiles = (double)(ndp + (rem ? 1 : 0));
      double parts = 0.0;
      if (rem) {
        parts = 1.0 + (double)rem / (double)S / (double)nchunk;
        if (parts > 2.0) parts = 2.0;
      }
      cost = items * per_item + tiles * 0.4 * 4 + parts * 5.5 * 4 + 1e-4 * (double)ntl;
    } else {
      const long per_cu = (ntl + HPVG_NUM_CU - 1) / HPVG_NUM_CU;
      const double tile_paired = nchunk * (mfma_paired + 6.0 + stage_us) + 1.6;
      const double tile_alone = nchunk * (mfma_paired * 3.0 / 6.5 + 2.0 + stage_us) + 1.6;
      cost = (double)(per_cu / 2) * tile_paired + (double)(per_cu & 1) * tile_alone + 1e-4 * (double)ntl;
    }
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = Plan{0, Tw, RS, PL, L, L, nrange, ntw, hpvg_cdiv(L, 32), WINO_NBP, 2, gridy, hpvg_cdiv(plload, 256), lds};
    }
  }
  return best;
}

// Which kernel a wide conv runs on (see wino_env_mode in conv_wino.inl).  Default threshold, measured per pyramid stage at
// B = 2 (tools/perf_wino.py, profiles/r02_perf_wino.txt): the 3x3x3 convs gain at every stage (x1.14-1.18 at stages 0-1,
// parity at 2, x1.08-1.32 from 3 up); the 3x3 convs are launch / fix-up bound below ~25 K output positions (x0.75-1.0) and
// gain from there (x1.07-1.25 at stages 7-9).
inline bool conv_use_wino(int B, int Cin, int Cout, int T, int H, int W, int KT, bool prologue) {
  if (prologue || !conv_is_wino(Cin, Cout)) return false;
  if (g_wino_mode < 0) {
    g_wino_mode = wino_env_mode();
    const char* e = getenv("HPVG_WINO_MIN");
    if (e) g_wino_min_pos = atol(e);
    const char* f = getenv("HPVG_WINO_STG");   // staging form: 1 = halo'd bands only, 2 = rows-as-in-memory wherever it fits
    if (f) g_wino_stg = atoi(f) == 1 ? 1 : (atoi(f) == 2 ? 2 : 0);
  }
  if (g_wino_mode == 0) return false;
  if (g_wino_mode == 2) return true;
  const long min_pos = g_wino_min_pos >= 0 ? g_wino_min_pos : (KT == 3 ? 0L : 24000L);
  return (long)B * T * H * W >= min_pos;
}

// The two-axis Winograd kernel (conv_wino2d.inl): where it can run and where it is taken.  g_wino2d: 0 = by size, 1 = never,
// 2 = wherever it can run (hpvg_conv_wino_config modes 5 / 6 set 2 / 1).
// HPVG_WINO2R=0 keeps the first-generation kernel (every wave all 16 points of one block: conv_wino2d_kernel); default: the
// points split over the waves by row (conv_wino2r_kernel: half the input-transform instructions and U loads per MFMA)
static const bool g_wino2r = [] { const char* e = getenv("HPVG_WINO2R"); return !e || atoi(e) != 0; }();
int g_wino2d = 0;
struct W2Geom { int Cq, R, ntq, tqw, gridy, nsc, ntl; bool ok; };
inline W2Geom wino2d_geom(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  W2Geom q{};
  if (!conv_is_wino2d(Cin, Cout, KT) || ((W & 1) && !g_wino2r) || W < 2 || (long)H * W < 4) return q;   // (odd W: conv_wino2r_kernel's ODD instance)
  if (g_wino2r && (long)H * W + 4 > HPVG_ZERO_PLANE_FLOATS) return q;                                     // (its zero plane)
  q.Cq = (W + 1) / 2;
  q.R = hpvg_cdiv(H, 2);
  const int nq = q.R * q.Cq;
  q.ntq = hpvg_cdiv(nq, 64);                            // tiles per plane: 64 consecutive quads each
  q.tqw = 64;
  // the staged span of every tile (first input of its first quad, rounded down to a multiple of 4, to the last input of its
  // last quad): one 16-byte group per lane
  for (int tp = 0; tp < q.ntq; ++tp) {
    const int Q0 = tp * 64, R0 = Q0 / q.Cq, c0 = Q0 - R0 * q.Cq;
    int Ql = Q0 + 63;
    if (Ql > nq - 1) Ql = nq - 1;
    const int R1 = Ql / q.Cq, c1 = Ql - R1 * q.Cq;
    const int lo = (2 * R0 - 1) * W + 2 * c0 - 1;
    const int lo4 = lo >= 0 ? (lo & ~3) : -((3 - lo) & ~3);
    const int hi = (2 * R1 + 2) * W + 2 * c1 + 3;
    if (hi - lo4 > 1024) return q;
  }
  q.gridy = hpvg_cdiv(hpvg_cdiv(Cout, 32), 2);
  q.nsc = hpvg_cdiv(Cin, 4);
  const long ntl = (long)B * T * q.ntq * q.gridy;
  if (ntl > 0x7fffffffL) return q;
  q.ntl = (int)ntl;
  q.ok = true;
  return q;
}
// Taken by size where it was measured to win.  Whole tiles, one workgroup per CU, no stream-K: a launch takes ceil(tiles / 256)
// rounds of ~72 us (64 -> 64), the one-axis kernel + fix-up ~100-120 us per 256 tiles' worth of work (more below one round).
// Round 3, conv_wino2r_kernel against the one-axis kernel (tools/perf_wino2.py, profiles/r03_perf_wino2r_sizes.txt; rounds =
// tiles / 256): 0.35 x0.94, 0.59 x1.12, 0.70 x1.48, 0.94 x1.43, 1.17 x0.94, 1.88 x1.36, 1.97 x1.07 (odd W), 3.23 x1.18-1.21,
// 3.94 x1.13 (odd W), 4.98 x1.37, 6.45 x1.31, 14.6 x1.33.  (Before the plane-end patch ran on twelve lanes at once the TAIL
// instance - H * W % 4 != 0 - lost below two rounds; round 2's conv_wino2d_kernel wanted three rounds filled to 88 %.)
// Rule: a single round from half a chip of tiles (the one-axis kernel pays its fix-up launch and stream-K seams there), else
// the rounds at least 60 % full on average.
inline bool conv_use_wino2d(const W2Geom& q, bool prologue) {
  if (!q.ok || prologue || g_wino2d == 1 || g_wino_mode == 0) return false;
  if (g_wino2d == 2) return true;
  const long rounds = (q.ntl + HPVG_NUM_CU - 1) / HPVG_NUM_CU;
  if (rounds == 1) return q.ntl * 2 >= HPVG_NUM_CU;
  return (long)q.ntl * 10 >= 6L * rounds * HPVG_NUM_CU;
}
template <typename K>
int launch_wino2d_kern(K kern, bool& attr_set, const Wino2Args& a, hipStream_t s) {
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  const int S = a.ntl < HPVG_NUM_CU ? a.ntl : HPVG_NUM_CU;
  hipLaunchKernelGGL(kern, dim3(S), dim3(256), (size_t)3 * 12 * W2_PL * sizeof(float), s, a);   // three input buffers
  return hpvg_launch_status();
}
template <int VAR, bool TAIL>
int launch_wino2d_inst(const Wino2Args& a, hipStream_t s) {
  static bool attr_r = false, attr_o = false, attr_d = false;
  if (a.W & 1) return launch_wino2d_kern(conv_wino2r_kernel<VAR, TAIL, true>, attr_o, a, s);    // (only the row-split kernel has an odd-W instance)
  if (g_wino2r) return launch_wino2d_kern(conv_wino2r_kernel<VAR, TAIL, false>, attr_r, a, s);
  return launch_wino2d_kern(conv_wino2d_kernel<VAR, TAIL>, attr_d, a, s);
}

template <int VAR>
int launch_wino2d_var(const Wino2Args& a, hipStream_t s) {
  return (((long)a.H * a.W) & 3) ? launch_wino2d_inst<VAR, true>(a, s) : launch_wino2d_inst<VAR, false>(a, s);
}

// The search costs ~10-20 us of host time; shapes repeat every iteration, so plans are memoised (host-side, tiny).
inline bool conv_is_narrow(int Cin, int Cout) { return Cout <= 4 && Cin > 4; }

Plan plan_conv(int B, int Cin, int Cout, int T, int H, int W, int KT, bool streamk, bool wino = false) {
  const bool narrow = conv_is_narrow(Cin, Cout);
  if (narrow) streamk = false;
  // open-addressing table (a ten-stage run touches ~150 distinct shapes; the search is up to ~100 us for wide 2-D images)
  struct Key { int B, Cin, Cout, T, H, W, KT, sk; };
  struct Entry { Key k; Plan p; bool used; };
  constexpr int NSLOT = 2048;
  static thread_local Entry cache[NSLOT];
  static thread_local int filled = 0;
  const Key k{B, Cin, Cout, T, H, W, KT, (streamk ? 1 : 0) | (wino ? 2 : 0) | (wino ? g_wino_stg << 2 : 0)};
  unsigned h = 2166136261u;
  for (int v : {B, Cin, Cout, T, H, W, KT, k.sk}) h = (h ^ (unsigned)v) * 16777619u;
  for (int probe = 0; probe < NSLOT; ++probe) {
    Entry& e = cache[(h + probe) & (NSLOT - 1)];
    if (!e.used) {
      const Plan p = wino ? plan_wino_search(B, Cin, Cout, T, H, W, KT, streamk) : plan_conv_search(B, Cin, Cout, T, H, W, KT, streamk, narrow);
      if (filled < NSLOT / 2) {  // keep the table sparse; past that, shapes are simply planned again
        e.k = k; e.p = p; e.used = true;
        ++filled;
      }
      return p;
    }
    const Key& c = e.k;
    if (c.B == B && c.Cin == Cin && c.Cout == Cout && c.T == T && c.H == H && c.W == W && c.KT == KT && c.sk == k.sk) return e.p;
  }
  return wino ? plan_wino_search(B, Cin, Cout, T, H, W, KT, streamk) : plan_conv_search(B, Cin, Cout, T, H, W, KT, streamk, narrow);
}

// Launch of S workgroups.  The dynamic LDS request is padded to 56 KB so that a third workgroup never fits on a CU:
// the persistent grid is sized for exactly two.
constexpr size_t CONV_MIN_LDS = 56 * 1024;
template <int CC, int KT, int MB, int NB, int VAR>
int launch_conv_var(const ConvFwdArgs& a, const Plan& p, int S, hipStream_t s) {
  static bool attr_set = false;
  auto kern = conv_mfma_kernel<CC, KT, MB, NB, VAR>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  // (four per CU: pad to 33 KB so that a fifth never fits)
  const size_t min_lds = HPVG_CONV_WGS(MB, NB) == 4 ? (size_t)33 * 1024 : CONV_MIN_LDS;
  hipLaunchKernelGGL(kern, dim3(S), dim3(256), p.lds < min_lds ? min_lds : p.lds, s, a);
  int rc = hpvg_launch_status();
  if (rc != HPVG_OK) return rc;
  const int nsk = a.ntl - a.skbase;
  if (nsk > 0) {
    hipLaunchKernelGGL(conv_fixup_kernel, dim3(nsk, MB * NB), dim3(256), 0, s, a, S, MB, NB);
    rc = hpvg_launch_status();
  }
  return rc;
}

template <int CC, int KT, int MB, int NB>
int launch_conv(const ConvFwdArgs& a, const Plan& p, int S, hipStream_t s) {
  if (a.in_scale) return launch_conv_var<CC, KT, MB, NB, VAR_PRO>(a, p, S, s);
  if (a.mask || a.mask_bits) return launch_conv_var<CC, KT, MB, NB, VAR_MASK>(a, p, S, s);
  if (a.bits_out) return launch_conv_var<CC, KT, MB, NB, VAR_BITS>(a, p, S, s);
  return launch_conv_var<CC, KT, MB, NB, VAR_PLAIN>(a, p, S, s);
}

template <int CC, int KT>
int dispatch_conv(const ConvFwdArgs& a, const Plan& p, int S, hipStream_t s) {
  if (p.MB == 2) {
    switch (p.NB) {
      case 1: return launch_conv<CC, KT, 2, 1>(a, p, S, s);
      case 2: return launch_conv<CC, KT, 2, 2>(a, p, S, s);
      default: return launch_conv<CC, KT, 2, 4>(a, p, S, s);
    }
  }
  switch (p.NB) {
    case 1: return launch_conv<CC, KT, 1, 1>(a, p, S, s);
    case 2: return launch_conv<CC, KT, 1, 2>(a, p, S, s);
    default: return launch_conv<CC, KT, 1, 4>(a, p, S, s);
  }
}

template <int KT, int VAR, int STG>
int launch_wino_var(const ConvFwdArgs& a, const Plan& p, int S, hipStream_t s) {
  static bool attr_set = false;
  auto kern = conv_wino_kernel<KT, WINO_NBP, VAR, STG>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(S), dim3(256), p.lds < CONV_MIN_LDS ? CONV_MIN_LDS : p.lds, s, a);
  int rc = hpvg_launch_status();
  if (rc != HPVG_OK) return rc;
  const int nsk = a.ntl - a.skbase;
  if (nsk > 0) {
    hipLaunchKernelGGL(conv_wino_fixup_kernel, dim3(nsk, WINO_NBP * 2), dim3(256), 0, s, a, S, WINO_NBP);
    rc = hpvg_launch_status();
  }
  return rc;
}

template <int KT, int STG>
int launch_wino_stg(const ConvFwdArgs& a, const Plan& p, int S, hipStream_t s) {
  if (a.mask || a.mask_bits) return launch_wino_var<KT, VAR_MASK, STG>(a, p, S, s);
  if (a.bits_out) return launch_wino_var<KT, VAR_BITS, STG>(a, p, S, s);
  return launch_wino_var<KT, VAR_PLAIN, STG>(a, p, S, s);
}
template <int KT>
int launch_wino(const ConvFwdArgs& a, const Plan& p, int S, hipStream_t s) {
  return p.stg == 2 ? launch_wino_stg<KT, 2>(a, p, S, s) : launch_wino_stg<KT, 1>(a, p, S, s);
}

template <int KT>
int launch_conv_narrow(const ConvFwdArgs& a, const Plan& p, hipStream_t s) {
  const int ksteps = hpvg_cdiv(a.Cin * KT, 2);
  const int pitch = (hpvg_cdiv((p.Th + 2) * p.RS, 32) * 32) | 1;  // odd: the 32 columns of a block land in 32 banks
  const size_t lds = (size_t)(9 * a.Cout) * pitch * sizeof(float);
#define HPVG_NARROW(NT)                                                                                              \
  {                                                                                                                  \
    static bool attr = false;                                                                                        \
    if (!attr) {                                                                                                     \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_narrow_kernel<KT, NT>),                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                 \
        (void)hipGetLastError();                                                                                     \
      attr = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((conv_narrow_kernel<KT, NT>), dim3(a.ntl), dim3(256), lds, s, a, ksteps, pitch);              \
  }
  if (9 * a.Cout <= 32) HPVG_NARROW(1) else HPVG_NARROW(2)
#undef HPVG_NARROW
  return hpvg_launch_status();
}

// stream-K grid for a plan: all co-resident slots, or one workgroup per item when there are fewer items than slots
inline int conv_sk_grid(int ntl, int nchunk, const Plan& p) {
  const long items = (long)ntl * nchunk;
  const long slots = conv_slots(p.MB, p.NB);
  return (int)(items < slots ? items : slots);
}
inline size_t conv_sk_ws_bytes(const Plan& p, int S) { return (size_t)S * 2 * p.MB * p.NB * 16 * 256 * sizeof(float); }

}  // namespace

extern "C" {

// number of floats of the packed-weight buffer for a conv with Cin -> Cout (kernel view)
static size_t wpack_floats_impl(int Cin, int Cout, int KT, bool with2d) {
  if (conv_is_narrow(Cin, Cout)) return (size_t)hpvg_cdiv(Cin * KT, 2) * hpvg_cdiv(9 * Cout, 32) * 64;  // wn[kstep][ntile][lane]
  return direct_pack_floats(Cin, Cout, KT) + (conv_is_wino(Cin, Cout) ? wino_pack_floats(Cin, Cout, KT) : 0) +
         (with2d && conv_is_wino2d(Cin, Cout, KT) ? wino2_pack_floats(Cin, Cout) : 0);
}
// does a launch of this geometry run the two-axis Winograd kernel (and therefore read the pack's third section)?
static bool wants_wino2d(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (B < 1 || T < 1 || H < 1 || W < 1 || !conv_use_wino(B, Cin, Cout, T, H, W, KT, false)) return false;
  return conv_use_wino2d(wino2d_geom(B, Cin, Cout, T, H, W, KT), false);
}
// A pack made WITHOUT a geometry serves every launch of the layer (all sections); the _for forms leave out the two-axis
// Winograd fragments (786 KB for 64 -> 64 x 27) unless a launch of the given geometry (kernel view: the conv's input
// B, T, H, W) reads them, and are valid for launches of that geometry only.
size_t hpvg_conv_wpack_floats(int Cin, int Cout, int KT) { return wpack_floats_impl(Cin, Cout, KT, true); }
size_t hpvg_conv_wpack_floats_for(int Cin, int Cout, int KT, int B, int T, int H, int W) {
  return wpack_floats_impl(Cin, Cout, KT, wants_wino2d(B, Cin, Cout, T, H, W, KT));
}
int hpvg_conv_wants_wino2d(int B, int Cin, int Cout, int T, int H, int W, int KT) { return wants_wino2d(B, Cin, Cout, T, H, W, KT) ? 1 : 0; }

// w: natural layout of the LAYER weight [Cout_layer][Cin_layer][KT][3][3].
// transpose_flip = 0: pack for the forward conv (kernel Cin=Cin_layer, Cout=Cout_layer)
// transpose_flip = 1: pack for backward-data (kernel Cin=Cout_layer, Cout=Cin_layer)
static int pack_weight_impl(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                            int transpose_flip, bool with2d, void* stream);
int hpvg_conv_pack_weight_f32(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                              int transpose_flip, void* stream) {
  return pack_weight_impl(w, inv_scale, wp, Cin_layer, Cout_layer, KT, transpose_flip, true, stream);
}
int hpvg_conv_pack_weight_for_f32(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                                  int transpose_flip, int B, int T, int H, int W, void* stream) {
  const int Cin_k = transpose_flip ? Cout_layer : Cin_layer, Cout_k = transpose_flip ? Cin_layer : Cout_layer;
  return pack_weight_impl(w, inv_scale, wp, Cin_layer, Cout_layer, KT, transpose_flip, wants_wino2d(B, Cin_k, Cout_k, T, H, W, KT), stream);
}
static int pack_weight_impl(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                            int transpose_flip, bool with2d, void* stream) {
  if (!w || !wp || (KT != 1 && KT != 3) || Cin_layer < 1 || Cout_layer < 1) return HPVG_ERR_ARG;
  const int Cin_k = transpose_flip ? Cout_layer : Cin_layer;
  const int Cout_k = transpose_flip ? Cin_layer : Cout_layer;
  if (conv_is_narrow(Cin_k, Cout_k)) {
    const long total = (long)wpack_floats_impl(Cin_k, Cout_k, KT, with2d);
    hipLaunchKernelGGL(conv_pack_narrow_kernel, dim3(hpvg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, inv_scale, wp,
                       Cin_k, Cout_k, KT, hpvg_cdiv(9 * Cout_k, 32), hpvg_cdiv(Cin_k * KT, 2), transpose_flip, total);
    return hpvg_launch_status();
  }
  const int CC = conv_cc(Cin_k);
  const int nchunk = hpvg_cdiv(Cin_k, CC);
  const int mbtot = hpvg_cdiv(Cout_k, 32);
  const long total = (long)wpack_floats_impl(Cin_k, Cout_k, KT, with2d);
  hipLaunchKernelGGL(conv_pack_kernel, dim3(hpvg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, inv_scale, wp,
                     Cin_k, Cout_k, KT * 9, CC, nchunk, mbtot, transpose_flip, (long)direct_pack_floats(Cin_k, Cout_k, KT), total);
  return hpvg_launch_status();
}

// n <= HPVG_PACK_BATCH_MAX weights of one SQUARE layer shape (C -> C, C > 4) packed in one launch; flip[i] selects the
// backward-data pack for item i.  Host arrays of device pointers (copied into the kernel arguments).
static int pack_batch_impl(int n, const float* const* w, float* const* wp, const int* flip, int C, int KT, bool with2d, void* stream);
int hpvg_conv_pack_weight_batch_f32(int n, const float* const* w, float* const* wp, const int* flip, int C, int KT, void* stream) {
  return pack_batch_impl(n, w, wp, flip, C, KT, true, stream);
}
int hpvg_conv_pack_weight_batch_for_f32(int n, const float* const* w, float* const* wp, const int* flip, int C, int KT, int B, int T,
                                        int H, int W, void* stream) {
  return pack_batch_impl(n, w, wp, flip, C, KT, wants_wino2d(B, C, C, T, H, W, KT), stream);
}
static int pack_batch_impl(int n, const float* const* w, float* const* wp, const int* flip, int C, int KT, bool with2d, void* stream) {
  if (n < 1 || n > HPVG_PACK_BATCH_MAX || !w || !wp || !flip || C <= 4 || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  PackBatchArgs a;
  for (int i = 0; i < n; ++i) {
    if (!w[i] || !wp[i]) return HPVG_ERR_ARG;
    a.w[i] = w[i]; a.wp[i] = wp[i]; a.flip[i] = flip[i];
  }
  const int CC = conv_cc(C);
  const int nchunk = hpvg_cdiv(C, CC);
  const int mbtot = hpvg_cdiv(C, 32);
  const long total = (long)wpack_floats_impl(C, C, KT, with2d);
  hipLaunchKernelGGL(conv_pack_batch_kernel, dim3(hpvg_cdiv(total, 256), n), dim3(256), 0, (hipStream_t)stream, a, C, KT * 9, CC,
                     nchunk, mbtot, (long)direct_pack_floats(C, C, KT), total);
  return hpvg_launch_status();
}

// y[b][o][t][h][w] = bias[o] + sum_{c,tap} Wp[o][c][tap] * f(x)[b][c][t+dt-pt][h+dh-1][w+dw-1]
// f = identity, or (in_scale[c]*x + in_shift[c]) followed by LeakyReLU(0.2) when in_lrelu (zero padding
// is applied AFTER f, as in the reference where f is the previous block's BatchNorm+LeakyReLU output).
// out_mask (optional, the output's shape): y *= (out_mask > 0 ? 1 : 0.2) after bias / LeakyReLU - the leaky_relu_backward of
// the layer below, fused into this conv when it runs as that layer's consumer's backward-data pass.
// ws / ws_bytes: hpvg_conv_fwd_ws_bytes() of scratch enables the stream-K schedule; with ws = NULL (or too small)
// the same kernel runs one workgroup per tile.
static int conv_fwd_impl(const float* x, const float* wp, const float* bias, const float* in_scale, const float* in_shift,
                         int in_lrelu, float* y, int out_lrelu, const float* out_mask, const unsigned* mask_bits, unsigned* bits_out,
                         void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H, int W, int KT, void* stream);

int hpvg_conv_fwd_f32(const float* x, const float* wp, const float* bias, const float* in_scale, const float* in_shift,
                      int in_lrelu, float* y, int out_lrelu, const float* out_mask, void* ws, size_t ws_bytes, int B, int Cin,
                      int Cout, int T, int H, int W, int KT, void* stream) {
  return conv_fwd_impl(x, wp, bias, in_scale, in_shift, in_lrelu, y, out_lrelu, out_mask, nullptr, nullptr, ws, ws_bytes, B, Cin,
                       Cout, T, H, W, KT, stream);
}

// number of 32-bit words of the 1-bit LeakyReLU mask of a [B][C][T][H][W] activation: [B][ceil(C/32)][T*H*W] (position fastest:
// the lanes of a wave hold neighbouring positions, so the word stores / loads of every conv kernel's epilogue are contiguous)
size_t hpvg_conv_mask_words(int B, int C, int T, int H, int W) { return (size_t)B * T * H * W * hpvg_cdiv(C, 32); }

// y = conv(x) (+bias) with the LeakyReLU sign mask in 1-bit form: `bits_out` (with out_lrelu; written for the consumer's
// backward-data conv) and / or `mask_bits` (read instead of a float out_mask).  Cout > 4 only (the MFMA kernel).
int hpvg_conv_fwd_bits_f32(const float* x, const float* wp, const float* bias, float* y, int out_lrelu, const unsigned* mask_bits,
                           unsigned* bits_out, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H, int W, int KT,
                           void* stream) {
  if (conv_is_narrow(Cin, Cout)) return HPVG_ERR_UNSUPPORTED;
  if (mask_bits && bits_out) return HPVG_ERR_UNSUPPORTED;
  return conv_fwd_impl(x, wp, bias, nullptr, nullptr, 0, y, out_lrelu, nullptr, mask_bits, bits_out, ws, ws_bytes, B, Cin, Cout, T, H,
                       W, KT, stream);
}

static int conv_fwd_impl(const float* x, const float* wp, const float* bias, const float* in_scale, const float* in_shift,
                         int in_lrelu, float* y, int out_lrelu, const float* out_mask, const unsigned* mask_bits, unsigned* bits_out,
                         void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H, int W, int KT, void* stream) {
  if (!x || !wp || !y) return HPVG_ERR_ARG;
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1) return HPVG_ERR_ARG;
  if (KT != 1 && KT != 3) return HPVG_ERR_UNSUPPORTED;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return HPVG_ERR_ARG;
  if (in_scale && out_mask && !conv_is_narrow(Cin, Cout)) return HPVG_ERR_UNSUPPORTED;  // no caller combines prologue and mask
  // development knob: HPVG_CONV_SK=0 forces the one-workgroup-per-tile schedule
  static const bool sk_off = [] { const char* e = getenv("HPVG_CONV_SK"); return e && atoi(e) == 0; }();
  const int CC = conv_cc(Cin);
  const int nchunk = hpvg_cdiv(Cin, CC);
  bool streamk = !sk_off && ws != nullptr && !conv_is_narrow(Cin, Cout);
  bool wino = conv_use_wino(B, Cin, Cout, T, H, W, KT, in_scale != nullptr);
  if (wino) {
    const W2Geom q2 = wino2d_geom(B, Cin, Cout, T, H, W, KT);
    if (conv_use_wino2d(q2, in_scale != nullptr) && !out_mask) {   // (fp32 out-masks: one-axis kernel)
      Wino2Args w2;
      w2.x = x; w2.bias = bias; w2.mask_bits = mask_bits; w2.bits_out = bits_out; w2.y = y;
      w2.wp = wp + direct_pack_floats(Cin, Cout, KT) + wino_pack_floats(Cin, Cout, KT);
      w2.B = B; w2.Cin = Cin; w2.Cout = Cout; w2.T = T; w2.H = H; w2.W = W;
      w2.Cq = q2.Cq; w2.R = q2.R; w2.ntq = q2.ntq; w2.tqw = q2.tqw; w2.mbtot = wino2_mb(Cout); w2.gridy = q2.gridy;
      w2.nsc = q2.nsc; w2.ntl = q2.ntl; w2.PL = W2_PL; w2.out_lrelu = out_lrelu; w2.mbreal = hpvg_cdiv(Cout, 32);
      // (the mask words are laid out over the layer's real m-tile count)
      Wino2Args a2 = w2;
      hipStream_t s2 = (hipStream_t)stream;
      if (mask_bits) return launch_wino2d_var<VAR_MASK>(a2, s2);
      if (bits_out) return launch_wino2d_var<VAR_BITS>(a2, s2);
      return launch_wino2d_var<VAR_PLAIN>(a2, s2);
    }
  }
  Plan p = plan_conv(B, Cin, Cout, T, H, W, KT, streamk, wino);
  if (wino && p.L == 0) {  // no Winograd tile fits this shape: the direct kernel takes it
    wino = false;
    p = plan_conv(B, Cin, Cout, T, H, W, KT, streamk, false);
  }
  if (p.L == 0) return HPVG_ERR_UNSUPPORTED;
  int ntl = B * T * p.nrange * p.ntw * p.gridy;
  if (streamk && ws_bytes < conv_sk_ws_bytes(p, conv_sk_grid(ntl, nchunk, p))) {
    streamk = false;
    p = plan_conv(B, Cin, Cout, T, H, W, KT, false, wino);
    if (p.L == 0) return HPVG_ERR_UNSUPPORTED;
    ntl = B * T * p.nrange * p.ntw * p.gridy;
  }
  ConvFwdArgs a;
  a.x = x; a.wp = wp; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.mask = out_mask; a.y = y;
  a.mask_bits = mask_bits; a.bits_out = bits_out;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T; a.H = H; a.W = W;
  a.Th = p.Th; a.Tw = p.Tw; a.RS = p.RS; a.PL = p.PL; a.L = p.L; a.qstride = p.qstride; a.nrange = p.nrange; a.ntw = p.ntw; a.nblocks = p.nblocks;
  a.nchunk = nchunk;
  a.mbtot = hpvg_cdiv(Cout, 32);
  a.nj = p.nj;
  a.in_lrelu = in_lrelu; a.out_lrelu = out_lrelu;
  a.gridy = p.gridy;
  a.ntl = ntl;
  if (conv_is_narrow(Cin, Cout)) {
    a.ndp = 1; a.skbase = ntl; a.skpart = nullptr;
    // development knob: HPVG_NARROW2=0 keeps the first-generation kernel
    static const bool n2_off = [] { const char* e = getenv("HPVG_NARROW2"); return e && atoi(e) == 0; }();
    Narrow2Geom g2;
    long nt2 = 0;
    if (!n2_off && !in_scale && !out_mask && plan_narrow2(B, Cin, Cout, T, H, W, KT, &g2, &nt2))
      return KT == 3 ? launch_conv_narrow2<3>(a, g2, nt2, (hipStream_t)stream) : launch_conv_narrow2<1>(a, g2, nt2, (hipStream_t)stream);
    return KT == 3 ? launch_conv_narrow<3>(a, p, (hipStream_t)stream) : launch_conv_narrow<1>(a, p, (hipStream_t)stream);
  }
  const int S = streamk ? conv_sk_grid(ntl, nchunk, p) : ntl;  // S = ntl: one data-parallel round, no stream-K part
  a.ndp = ntl / S;
  a.skbase = a.ndp * S;
  a.skpart = (float*)ws;
  hipStream_t s = (hipStream_t)stream;
  a.stagger = 0;
  if (wino) {
    static const int stagger_env = [] { const char* e = getenv("HPVG_WINO_STAGGER"); return e ? atoi(e) : 0; }();
    a.stagger = (S > HPVG_NUM_CU) ? stagger_env : 0;   // only when two workgroups share a CU
    a.wp = wp + direct_pack_floats(Cin, Cout, KT);   // the U fragments follow the direct pack
    return KT == 3 ? launch_wino<3>(a, p, S, s) : launch_wino<1>(a, p, S, s);
  }
  if (CC == 8) return KT == 3 ? dispatch_conv<8, 3>(a, p, S, s) : dispatch_conv<8, 1>(a, p, S, s);
  return KT == 3 ? dispatch_conv<4, 3>(a, p, S, s) : dispatch_conv<4, 1>(a, p, S, s);
}

// host only: which kernel family hpvg_conv_fwd_f32 / hpvg_conv_fwd_bits_f32 run this shape on (no prologue, no fp32 out-mask):
// 0 = direct implicit GEMM (conv_mfma_kernel), 1 = Winograd F(2,3) along W (conv_wino_kernel: 2/3 of the direct matrix-core
// work), 2 = Winograd F(2x2,3x3) (conv_wino2d_kernel: 4/9), 3 = the narrow-output kernel (Cout <= 4)
int hpvg_conv_fwd_kernel_kind(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1 || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  if (conv_is_narrow(Cin, Cout)) return 3;
  if (!conv_use_wino(B, Cin, Cout, T, H, W, KT, false)) return 0;
  if (conv_use_wino2d(wino2d_geom(B, Cin, Cout, T, H, W, KT), false)) return 2;
  return plan_conv(B, Cin, Cout, T, H, W, KT, true, true).L != 0 ? 1 : 0;
}

// scratch the stream-K schedule of hpvg_conv_fwd_f32 wants for this shape
size_t hpvg_conv_fwd_ws_bytes(int B, int Cin, int Cout, int T, int H, int W, int KT) {
  if (B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1 || (KT != 1 && KT != 3)) return 0;
  if (conv_is_narrow(Cin, Cout)) return 0;
  // the larger of the two kernels' needs: a launch with a BatchNorm prologue runs the direct kernel on a Winograd shape
  size_t need = 0;
  for (int wino = 0; wino < 2; ++wino) {
    if (wino && !conv_use_wino(B, Cin, Cout, T, H, W, KT, false)) break;
    const Plan p = plan_conv(B, Cin, Cout, T, H, W, KT, true, wino != 0);
    if (p.L == 0) continue;
    const size_t n = conv_sk_ws_bytes(p, conv_sk_grid(B * T * p.nrange * p.ntw * p.gridy, hpvg_cdiv(Cin, conv_cc(Cin)), p));
    if (n > need) need = n;
  }
  return need;
}

// Debug/introspection: the tile plan of the stream-K launch (for tests and DESIGN.md tables).
// out[0..9] = L (positions per tile), Tw (band width), nrange (tiles per band plane), ntw (bands), nblocks, NB, MB, gridy,
// lds_bytes, ntiles
int hpvg_conv_fwd_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  const Plan p = plan_conv(B, Cin, Cout, T, H, W, KT, true);
  out[0] = p.L; out[1] = p.Tw; out[2] = p.nrange; out[3] = p.ntw; out[4] = p.nblocks; out[5] = p.NB; out[6] = p.MB;
  out[7] = p.gridy; out[8] = (int)p.lds; out[9] = B * T * p.nrange * p.ntw;
  return HPVG_OK;
}

// Run-time switch of the Winograd path (tests and A/B tools; see wino_env_mode): mode 0 = direct kernel only, 1 = from
// `min_positions` output positions up, 2 = every eligible launch; a negative argument leaves that setting alone.  Returns
// the mode in force, or HPVG_ERR_UNSUPPORTED when the library was started with HPVG_WINO=0 (packs carry no U fragments).
int hpvg_conv_wino_config(int mode, long min_positions) {
  if (wino_env_mode() == 0) return HPVG_ERR_UNSUPPORTED;
  (void)conv_use_wino(1, 8, 64, 1, 1, 1, 3, false);   // settle the defaults
  if (mode >= 0) {
    g_wino_mode = mode > 2 ? 2 : mode;
    g_wino_stg = mode == 3 ? 2 : (mode == 4 ? 1 : 0);   // 3 / 4: every eligible launch AND one staging form forced (tests)
    // 5: every eligible launch, the two-axis kernel wherever it can run; 3 / 4 / 6: the one-axis kernel only
    g_wino2d = mode == 5 ? 2 : (mode == 3 || mode == 4 || mode == 6 ? 1 : 0);
  }
  if (min_positions >= 0) g_wino_min_pos = min_positions;
  if (g_wino_mode == 2 && g_wino2d == 2) return 5;
  if (g_wino_mode == 2 && g_wino_stg) return g_wino_stg == 2 ? 3 : 4;
  if (g_wino_mode == 2 && g_wino2d == 1) return 6;
  return g_wino_mode;
}

// Debug/introspection: the tile plan conv_wino_kernel would run this shape with (whether or not the launch picks it):
// out[0..9] = L, Tw, nrange, ntw, RS, NBP, m-tiles per workgroup, gridy, lds_bytes, ntiles; HPVG_ERR_UNSUPPORTED when the
// layer is not eligible (Cin < 8, Cout <= 32) or no tile fits.
int hpvg_conv_wino_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3) || B < 1 || Cin < 1 || Cout < 1 || T < 1 || H < 1 || W < 1) return HPVG_ERR_ARG;
  if (!conv_is_wino(Cin, Cout)) return HPVG_ERR_UNSUPPORTED;
  const Plan p = plan_conv(B, Cin, Cout, T, H, W, KT, true, true);
  if (p.L == 0) return HPVG_ERR_UNSUPPORTED;
  out[0] = p.L; out[1] = p.Tw; out[2] = p.nrange; out[3] = p.ntw; out[4] = p.RS; out[5] = p.NB; out[6] = p.MB;
  out[7] = p.gridy; out[8] = (int)p.lds; out[9] = B * T * p.nrange * p.ntw * p.gridy;
  return HPVG_OK;
}

// Debug/introspection: the tile plan of the narrow-output kernel (conv_narrow2_kernel) for tests: out[0..6] = RS, Th, nth,
// nb, npos, G, pitch, then nb triples (window start, first output column, output columns).  Returns HPVG_ERR_UNSUPPORTED
// when the shape runs on the first-generation kernel instead.
int hpvg_conv_narrow_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out) {
  if (!out || (KT != 1 && KT != 3)) return HPVG_ERR_ARG;
  Narrow2Geom g;
  long nt = 0;
  if (!conv_is_narrow(Cin, Cout) || !plan_narrow2(B, Cin, Cout, T, H, W, KT, &g, &nt)) return HPVG_ERR_UNSUPPORTED;
  out[0] = g.RS; out[1] = g.Th; out[2] = g.nth; out[3] = g.nb; out[4] = g.npos; out[5] = g.G; out[6] = g.pitch;
  for (int k = 0; k < g.nb; ++k) { out[7 + 3 * k] = g.ws[k]; out[8 + 3 * k] = g.ob[k]; out[9 + 3 * k] = g.on[k]; }
  return HPVG_OK;
}

#ifdef HPVG_TRACE
int hpvg_debug_phase_read(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 4 * n) == hipSuccess ? 0 : -4;
}
int hpvg_debug_trace_read(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 4 * n) == hipSuccess ? 0 : -4;
}
#endif

}  // extern "C"
