// Winograd F(2,3) along W for the wide 3x3 / 3x3x3 convolutions (Cin >= 8, Cout > 32) - included by conv_mfma.hip inside
// its anonymous namespace (shares ConvFwdArgs, the LDS staging and the stream-K tile decode with conv_mfma_kernel).
//
// Same reference call sites as conv_mfma_kernel (nn.Conv3d / nn.Conv2d of ConvBlock3D/2D(SN), modules/networks_3d.py:48-70,
// modules/networks_2d.py:53-75, and their backward-data passes); only the arithmetic is regrouped:
//   for an output PAIR (w, w+1) of one row and one (dt, dh) the three dw taps g0 g1 g2 over inputs d0..d3 (columns w-1..w+2)
//       y0 = d0 g0 + d1 g1 + d2 g2,  y1 = d1 g0 + d2 g1 + d3 g2                                   (6 multiplies)
//   become four products  m_j = V_j * U_j  with
//       V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)          U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//       y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3                                                     (4 multiplies)
// and each m_j is summed over (channel, dt, dh) BEFORE the output transform, so the channel contraction - the MFMA work -
// shrinks 1.5x (36 instead of 27 MFMA k-steps per channel pair) while the GEMM's N axis counts output pairs.
// fp32 throughout; U is exact up to one rounding of the half sums, V costs one rounding per operand, the output transform
// two per value: measured error against the direct kernel ~2e-6 of the output scale (tests/test_hip_ops.py), inside the
// path's 1e-3 budget.
//
// GEMM view per (sample, output plane t):  M = output channels (one 32-row tile per wave: waves 0/2 take m-tile 0, waves
// 1/3 m-tile 1), N = output pairs of the row-flattened band (LDS row stride RS = Tw + 2, EVEN, so a pair never straddles
// a row; NBP blocks of 32 pairs per wave), K = (dt, dh, point j, channel).  A lane reads the four inputs of its pair with
// two 8-byte aligned ds_read_b64 (consecutive lanes 8 bytes apart: conflict free) and forms V_j with one VALU op per MFMA.
// The A operand U is pre-packed in fragment order behind the direct pack (conv_pack_kernel), one 16-byte load per lane
// per (dt, dh, j).  Accumulators: 4 points x NBP blocks x 16 registers = 128 for NBP = 2.

typedef float f32x2a __attribute__((ext_vector_type(2), aligned(8)));   // 8-byte aligned LDS pair
typedef float f32x2u4 __attribute__((ext_vector_type(2), aligned(4)));  // 8-byte global store from a 4-byte aligned address

constexpr int WINO_CC = 8;
constexpr int WINO_NBP = 2;

// ---- Staging, second form (STG = 2): the rows as they lie in memory.
// With ONE band of the full (even) width the inputs of a tile - flattened positions [q0 - W - 1, q0 + L + W] of each
// (channel, time) plane - are one contiguous span of global memory, so the LDS image of a plane is a verbatim copy of it
// (row stride W, NO halo columns: no junk GEMM columns either) and is filled by 16-BYTE LDS-DMA pieces: lane g of the
// workgroup moves elements [base + 4g, base + 4g + 4), base = the span start rounded down to a multiple of 4 (groups then
// never straddle the start of the plane) - 24 pieces per chunk and wave instead of 72-96 dword pieces.  The image sits one
// float into its plane slot, which makes the first input of a tile's first pair 8-byte aligned (`sh` is 2 or 4).  What the
// halo columns did is done by two per-lane constants instead: a pair at the left image border must see d0 = 0 and a pair at
// the right border d3 = 0 (the image holds the neighbouring ROW's element there); they are folded into the input transform
// (V0 = f0 d0 - d2, V3 = d1 - f3 d3: one fma each instead of one subtraction).  Rows above / below the image are positions
// outside [0, H W): never loaded, zeroed once per tile; a group cut by the END of the plane (H W not a multiple of 4) is
// loaded by a dword piece of at most three lanes.
struct StageSlots2 {
  unsigned bofs;               // byte offset of this lane's group inside a plane (lanes of em4)
  unsigned long long em4;      // lanes of this wave whose group lies wholly inside the plane
  unsigned long long em1;      // wave 0: lanes of the dword piece for the group cut by the plane end
  unsigned tofs;               // its per-lane byte offset inside the plane
  int tl;                      // ... and where that group starts in the plane image (floats)
  int i0;                      // first element of this lane's group
  int sh;                      // image offset of the tile's first input (q0 - W - 1): 2 or 4
  bool span;                   // this lane's group belongs to the staged span
};
__device__ __forceinline__ StageSlots2 wino_stage_slots2(const ConvFwdArgs& a, int q0, int tid, int wave) {
  StageSlots2 s;
  const int HWp = a.H * a.W;
  const int s0 = q0 - a.W - 1;                                   // odd: q0 and W are even
  const int base = s0 >= 0 ? (s0 & ~3) : -((3 - s0) & ~3);       // floor to a multiple of 4
  s.sh = s0 - base + 1;
  const int ng = (q0 + a.L + a.W + 1 - base + 3) >> 2;           // valid lanes read up to element q0 + L + W
  s.i0 = base + 4 * tid;
  s.span = tid < ng;
  const bool full = s.span && s.i0 >= 0 && s.i0 + 4 <= HWp;
  s.bofs = full ? (unsigned)s.i0 * 4u : 0u;
  s.em4 = __ballot(full);
  const int gt = (HWp - base) >> 2;                              // the group that holds element H*W
  const int r = HWp - (base + 4 * gt);                           // its elements inside the plane
  const bool tail = r > 0 && gt < ng;
  s.em1 = (tail && wave == 0) ? ((1ull << r) - 1ull) : 0ull;
  s.tofs = (unsigned)(HWp - r + (tid & 63)) * 4u;
  s.tl = 4 * gt;
  return s;
}

template <int CC, int KT>
__device__ __forceinline__ void wino_stage_chunk2(const ConvFwdArgs& a, float* xs, const StageSlots2& sl, int ch, bool first_chunk,
                                                  int b, int t, int tid, int wave) {
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int HWp = a.H * a.W;
  const long HWb = (long)HWp * 4;
  const int PL = a.PL;
  const char* cbase = reinterpret_cast<const char*>(a.x) + (((long)b * a.Cin + (long)ch * CC) * a.T + (t - (KT == 3 ? 1 : 0))) * HWb;
  int pl = 0;
#pragma unroll 1
  for (int c = 0; c < CC; ++c, cbase += (long)a.T * HWb)
#pragma unroll
    for (int dt = 0; dt < KT; ++dt, ++pl) {
      const int cg = ch * CC + c;
      const int tt = t + dt - (KT == 3 ? 1 : 0);
      const bool valid = cg < a.Cin && tt >= 0 && tt < a.T;
      float* img = xs + pl * PL + 1;          // element base + k of the plane at img[k]
      if (valid) {
        const char* src = cbase + dt * HWb;
        if (sl.em4 != 0ull) conv_dma_piece16(src, sl.bofs, (unsigned)(size_t)(lptr_t)(img + wave * 256), sl.em4);
        if (sl.em1 != 0ull) conv_dma_piece(src, sl.tofs, (unsigned)(size_t)(lptr_t)(img + sl.tl), sl.em1);
        if (first_chunk && sl.span) {
          // positions outside the plane (rows -1 and H) are never written by a load: zero them once per tile
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (sl.i0 + e < 0 || sl.i0 + e >= HWp) img[4 * tid + e] = 0.f;
        }
      } else if (first_chunk || cg >= a.Cin) {
        // a plane outside the clip (t) stays zero for the whole tile; a channel past Cin only exists in the last chunk
        if (sl.span) {
#pragma unroll
          for (int e = 0; e < 4; ++e) img[4 * tid + e] = 0.f;
        }
      }
    }
}

template <int KT, int NBP, int VAR, int STG>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvFwdArgs a) {
  constexpr int CC = WINO_CC, CP = CC / 2;
  constexpr int TH = KT * 3;  // (dt, dh) steps per chunk, four tap-points each
  extern __shared__ __attribute__((aligned(16))) float xs[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int mw = wave & 1, nw = wave >> 1;
  const long HW = (long)a.H * a.W;
  const int RS = a.RS, PL = a.PL;
  const float* xl0 = xs + half * KT * PL + 2 * (l31 + 32 * nw);  // pair block i of this wave: + 128 * i floats
  const long astep = (long)a.mbtot * 64;                          // f32x4 elements between two tap-points

  HPVG_TRACE_BEGIN
  if (a.stagger > 0) {
    // Two workgroups share a CU's matrix pipe and start together; their stage / compute phases then stay aligned (both
    // stage, then both compute at half rate): a phase offset neither grows nor decays, so one is set here, once.
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: bit 0 of the wave slot tells the two apart
    if (hwid & 1u)
      for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
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
    const SkTile tc = sk_decode_tile(a, tile, 2);
    const int b = tc.b, t = tc.t, q0 = tc.q0, w0 = tc.w0;
    const int mt = tc.mb0 + mw;   // this wave's m-tile (32 output channels)

    StageSlots sl;
    StageSlots2 sl2;
    const float* xl = xl0;
    float f0[NBP], f3[NBP];     // STG 2: 0 where the pair's d0 / d3 lies outside the image row, else 1
    if constexpr (STG == 2) {
      sl2 = wino_stage_slots2(a, q0, tid, wave);
      xl = xl0 + sl2.sh;
#pragma unroll
      for (int i = 0; i < NBP; ++i) {
        const int Q = q0 + 2 * ((nw + 2 * i) * 32 + l31);
        const int gw = Q - (Q / RS) * RS;
        f0[i] = gw == 0 ? 0.f : 1.f;
        f3[i] = gw + 2 >= a.W ? 0.f : 1.f;
      }
    } else {
      sl = conv_stage_slots(a, q0, w0, tid);
    }

    f32x16 acc[4][NBP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NBP; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    for (int ch = ch_lo; ch < ch_hi; ++ch) {
      if (!first_stage) __syncthreads();  // every wave is done reading the previous chunk
      first_stage = false;
      HPVG_PH(3)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see conv_mfma_kernel (dead A prefetch vs the staging loop's registers)
#ifdef HPVG_ABL_STAGE   // development ablation (timing only, wrong results): stage the first chunk of a tile only
      if (ch == ch_lo)
#endif
      {
        if constexpr (STG == 2) wino_stage_chunk2<CC, KT>(a, xs, sl2, ch, ch == ch_lo, b, t, tid, wave);
        else conv_stage_chunk<CC, KT, false>(a, xs, sl, ch, ch == ch_lo, b, t, tid, wave);
      }
      HPVG_PH(0)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the asm LDS-DMA pieces are invisible to the compiler's counters
      __syncthreads();
      HPVG_PH(1)

      const f32x4* wpt = reinterpret_cast<const f32x4*>(a.wp) + ((long)(ch * TH * 4) * a.mbtot + mt) * 64 + lane;
      f32x4 av[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) av[j] = wpt[j * astep];
      f32x2a r0[2][NBP], r1[2][NBP];   // inputs d0 d1 | d2 d3 of the pair, two register sets (one channel pair ahead)
#pragma unroll
      for (int i = 0; i < NBP; ++i) {
        r0[0][i] = *reinterpret_cast<const f32x2a*>(xl + i * 128);
        r1[0][i] = *reinterpret_cast<const f32x2a*>(xl + i * 128 + 2);
      }
#pragma unroll 1
      for (int dt = 0; dt < KT; ++dt) {
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
          wpt += 4 * astep;
          f32x4 an[4];
#ifdef HPVG_ABL_ALOAD   // development ablation (timing only, wrong results): no A-fragment loads inside the loop
#pragma unroll
          for (int j = 0; j < 4; ++j) an[j] = av[j];
#else
#pragma unroll
          for (int j = 0; j < 4; ++j) an[j] = wpt[j * astep];
#endif
          const float* xt = xl + dt * PL + dh * RS;
          const float* xn = dh < 2 ? xl + dt * PL + (dh + 1) * RS : xl + (dt + 1 < KT ? dt + 1 : 0) * PL;
#pragma unroll
          for (int cp = 0; cp < CP; ++cp) {
            const float* nx = (cp + 1 < CP) ? xt + (2 * (cp + 1) * KT) * PL : xn;
#pragma unroll
            for (int i = 0; i < NBP; ++i) {
              r0[(cp + 1) & 1][i] = *reinterpret_cast<const f32x2a*>(nx + i * 128);
              r1[(cp + 1) & 1][i] = *reinterpret_cast<const f32x2a*>(nx + i * 128 + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < NBP; ++i) {
                const f32x2a d01 = r0[cp & 1][i], d23 = r1[cp & 1][i];
                float v;
                if constexpr (STG == 2)
                  v = j == 0 ? __builtin_fmaf(f0[i], d01[0], -d23[0])
                             : (j == 1 ? d01[1] + d23[0] : (j == 2 ? d23[0] - d01[1] : __builtin_fmaf(-f3[i], d23[1], d01[1])));
                else
                  v = j == 0 ? d01[0] - d23[0] : (j == 1 ? d01[1] + d23[0] : (j == 2 ? d23[0] - d01[1] : d01[1] - d23[1]));
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][cp], v, acc[j][i], 0, 0, 0);
              }
            __builtin_amdgcn_s_setprio(0);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) av[j] = an[j];
        }
      }
      HPVG_PH(2)
    }

    // ---- output transform in place: acc[0] <- y0 (even column of the pair), acc[1] <- y1 (odd column)
#pragma unroll
    for (int i = 0; i < NBP; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = acc[0][i][e], m1 = acc[1][i][e], m2 = acc[2][i][e], m3 = acc[3][i][e];
        acc[0][i][e] = (m0 + m1) + m2;
        acc[1][i][e] = (m1 - m2) - m3;
      }

    if (ch_lo != 0 || ch_hi != a.nchunk) {
      // ---- part of a tile: the (linear) output transform of the partial sums, register order
      const int seg = (tile == first_sk_tile) ? 0 : 1;
      float* dst = a.skpart + ((long)(g * 2 + seg) * (NBP * 2 * 16)) * 256 + tid;
#pragma unroll
      for (int i = 0; i < NBP; ++i)
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[((i * 2 + par) * 16 + e) * 256] = acc[par][i][e];
      continue;
    }
    // ---- whole tile: bias after the accumulation, optional LeakyReLU / mask, 8-byte stores
    float bias_r[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      bias_r[e] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    }
    const bool mt_ok = mt * 32 < a.Cout;
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
      const int q = 2 * ((nw + 2 * i) * 32 + l31);
      const int Q = q0 + q;
      const int gh = Q / RS, ww = Q - gh * RS;
      const int gw = w0 + ww;
      const bool ok0 = q < a.L && ww < a.Tw && gh < a.H && gw < a.W;
      const bool ok1 = ok0 && gw + 1 < a.W;
      const long sp = (long)t * HW + (long)gh * a.W + gw;
      const long wi = ((long)b * a.mbtot + mt) * a.T * HW + sp;   // mask word of the even position; the odd one: + 1
      if constexpr (VAR == VAR_BITS) {
        unsigned word0 = 0, word1 = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int sh = (e & 3) + 8 * (e >> 2) + 4 * half;
          word0 |= (acc[0][i][e] + bias_r[e] > 0.f ? 1u : 0u) << sh;
          word1 |= (acc[1][i][e] + bias_r[e] > 0.f ? 1u : 0u) << sh;
        }
        word0 |= (unsigned)__shfl_xor((int)word0, 32, 64);   // the two half-waves hold complementary channels of one position
        word1 |= (unsigned)__shfl_xor((int)word1, 32, 64);
        if (half == 0 && mt_ok) {
          if (ok0) a.bits_out[wi] = word0;
          if (ok1) a.bits_out[wi + 1] = word1;
        }
      }
      if (!ok0) continue;
      unsigned mword0 = 0, mword1 = 0;
      if constexpr (VAR == VAR_MASK) {
        if (a.mask_bits && mt_ok) {
          mword0 = a.mask_bits[wi];
          if (ok1) mword1 = a.mask_bits[wi + 1];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int sh = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int co = mt * 32 + sh;
        if (co < a.Cout) {
          float v0 = acc[0][i][e] + bias_r[e], v1 = acc[1][i][e] + bias_r[e];
          if (a.out_lrelu) {
            v0 = hpvg_lrelu(v0);
            v1 = hpvg_lrelu(v1);
          }
          const long oi = ((long)b * a.Cout + co) * a.T * HW + sp;
          if constexpr (VAR == VAR_MASK) {
            if (a.mask_bits) {
              v0 *= ((mword0 >> sh) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
              v1 *= ((mword1 >> sh) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
            } else {
              v0 *= a.mask[oi] > 0.f ? 1.f : HPVG_LRELU_SLOPE;
              if (ok1) v1 *= a.mask[oi + 1] > 0.f ? 1.f : HPVG_LRELU_SLOPE;
            }
          }
          if (ok1) {
            f32x2u4 o2;
            o2[0] = v0;
            o2[1] = v1;
            *reinterpret_cast<f32x2u4*>(a.y + oi) = o2;
          } else {
            a.y[oi] = v0;
          }
        }
      }
    }
  }
  HPVG_PH_END
  HPVG_TRACE_END
}

// Finishes the tiles conv_wino_kernel computed in parts (the slabs hold output-transformed partial sums): one workgroup
// per (stream-K tile, pair block i, column parity), slabs summed in workgroup order, then the same epilogue.
__global__ __launch_bounds__(256) void conv_wino_fixup_kernel(const ConvFwdArgs a, int S, int NBP) {
  const int r = blockIdx.x;
  const int mi = blockIdx.y;           // i * 2 + parity
  const int i = mi >> 1, par = mi & 1;
  const long Isk = (long)(a.ntl - a.skbase) * a.nchunk;
  const long i0 = (long)r * a.nchunk, i1 = i0 + a.nchunk - 1;
  const int g0 = (int)(((i0 + 1) * S + Isk - 1) / Isk) - 1;  // owner of the tile's first item
  const int g1 = (int)(((i1 + 1) * S + Isk - 1) / Isk) - 1;  // ... and of its last
  if (g0 == g1) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int mw = wave & 1, nw = wave >> 1;
  const SkTile tc = sk_decode_tile(a, a.skbase + r, 2);
  const int mt = tc.mb0 + mw;
  const long HW = (long)a.H * a.W;
  const int RS = a.RS;
  const int q = 2 * ((nw + 2 * i) * 32 + l31);
  const int Q = tc.q0 + q;
  const int gh = Q / RS, ww = Q - gh * RS;
  const int gw = tc.w0 + ww + par;
  // (the test does not depend on `half`: both half-waves of a position leave or stay together, see the shuffle below)
  if (!(q < a.L && ww < a.Tw && gh < a.H && gw < a.W) || mt * 32 >= a.Cout) return;
  const long slab = (long)NBP * 2 * 16 * 256;
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
  const long wi = ((long)tc.b * a.mbtot + mt) * a.T * HW + sp;
  const unsigned mword = a.mask_bits ? a.mask_bits[wi] : 0u;
  unsigned word = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int sh = (e & 3) + 8 * (e >> 2) + 4 * half;
    const int co = mt * 32 + sh;
    if (co >= a.Cout) continue;
    float val = v[e];
    if (a.bias) val += a.bias[co];
    word |= (val > 0.f ? 1u : 0u) << sh;
    if (a.out_lrelu) val = hpvg_lrelu(val);
    const long oi = ((long)tc.b * a.Cout + co) * a.T * HW + sp;
    if (a.mask_bits) val *= ((mword >> sh) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
    else if (a.mask) val *= a.mask[oi] > 0.f ? 1.f : HPVG_LRELU_SLOPE;
    a.y[oi] = val;
  }
  if (a.bits_out) {
    word |= (unsigned)__shfl_xor((int)word, 32, 64);
    if (half == 0) a.bits_out[wi] = word;
  }
}

// U fragments of one Winograd weight: wpw[chunk][th = dt*3+dh][j][mblock][lane][cp] (see conv_pack_kernel for the direct
// layout this follows); idx runs over the whole buffer including the zero tail (four tap-points: the A prefetch runs one
// (dt, dh) step ahead).
__device__ __forceinline__ float wino_pack_value(const float* __restrict__ w, long idx, int Cin_k, int Cout_k, int KT, int nchunk,
                                                 int mbtot, int transpose_flip) {
  constexpr int CP = WINO_CC / 2;
  long r = idx;
  const int cp = r % CP; r /= CP;
  const int lane = r % 64; r /= 64;
  const int mb = r % mbtot; r /= mbtot;
  const int j = r % 4; r /= 4;
  const int th = r % (KT * 3); r /= (KT * 3);
  const int ch = (int)r;
  if (ch >= nchunk) return 0.f;
  const int o = mb * 32 + (lane & 31);
  const int c = ch * WINO_CC + 2 * cp + (lane >> 5);
  if (o >= Cout_k || c >= Cin_k) return 0.f;
  const int taps = KT * 9;
  float gk[3];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw) {
    const int tap = th * 3 + dw;
    gk[dw] = !transpose_flip ? w[((long)o * Cin_k + c) * taps + tap] : w[((long)c * Cout_k + o) * taps + (taps - 1 - tap)];
  }
  return j == 0 ? gk[0] : (j == 1 ? 0.5f * ((gk[0] + gk[2]) + gk[1]) : (j == 2 ? 0.5f * ((gk[0] + gk[2]) - gk[1]) : gk[2]));
}

// HPVG_WINO (read once): 0 = no Winograd path at all (the weight pack carries no U fragments either), 2 = every eligible
// launch, 1 / unset = eligible launches of at least HPVG_WINO_MIN output positions (B*T*H*W); mode and threshold can be
// changed at run time through hpvg_conv_wino_config (tests, A/B tools) - except from / to 0 at start-up, which sizes the packs.
inline int wino_env_mode() {
  static const int m = [] { const char* e = getenv("HPVG_WINO"); return e ? atoi(e) : 1; }();
  return m;
}
int g_wino_mode = -1;          // -1: not configured yet, take the environment's
long g_wino_min_pos = -1;
int g_wino_stg = 0;            // staging form of conv_wino_kernel: 0 = by size, 1 = halo'd bands only, 2 = rows-as-in-memory wherever it fits
inline bool conv_is_wino(int Cin, int Cout) { return wino_env_mode() != 0 && Cin >= 8 && Cout > 32; }
// the two-axis kernel (conv_wino2d.inl) has its own fragment section behind this one for the 3x3x3 layers; HPVG_WINO2D=0
// (read once) leaves it out of the packs and of the dispatch
inline bool conv_is_wino2d(int Cin, int Cout, int KT) {
  static const bool off = [] { const char* e = getenv("HPVG_WINO2D"); return e && atoi(e) == 0; }();
  return !off && KT == 3 && conv_is_wino(Cin, Cout);
}
inline size_t wino_pack_floats(int Cin, int Cout, int KT) {
  const int nchunk = hpvg_cdiv(Cin, WINO_CC);
  const int mbtot = hpvg_cdiv(Cout, 32);
  return ((size_t)nchunk * KT * 3 * 4 + 4) * mbtot * 64 * (WINO_CC / 2);
}
