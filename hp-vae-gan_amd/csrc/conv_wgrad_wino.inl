// Weight gradient of the wide 3x3 / 3x3x3 convs with the TRANSPOSE of the forward Winograd F(2,3) along W (conv_wino.inl) -
// included by conv_wgrad.hip inside its anonymous namespace (shares WgradArgs, the zero word and the tile family with
// conv_wgrad_kernel, whose structure it keeps: one time tap and one 64 x 64 (o, c) block per persistent workgroup, one
// 32 x 32 sub-block per wave, two LDS tile buffers filled by LDS-DMA in the shadow of the MFMAs, one partial slab at the end).
//
// Same reference call sites as conv_wgrad_kernel (the weight half of aten::convolution_backward, train_video.py:182,200,
// train_image.py:193,215, and the gradient penalty's double backward, modules/utils.py:14-18).  For one pair of output
// columns (w, w+1) of a row, gradients y0 y1 and inputs d0..d3 (columns w-1..w+2) of row h+dh-1, the three dw taps
//       g0 += y0 d0 + y1 d1,  g1 += y0 d1 + y1 d2,  g2 += y0 d2 + y1 d3                              (6 multiplies)
// are regrouped into four products  m_j = Y_j * V_j  with
//       Y = (y0, y0 + y1, y0 - y1, -y1)              V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
//       g0 = m0 + (m1 + m2)/2,  g1 = (m1 - m2)/2,  g2 = (m1 + m2)/2 + m3                             (4 multiplies)
// and every m_j is summed over all pairs, planes and samples BEFORE the (linear) output transform, which the reduce kernel
// applies once per weight.  GEMM per (dt, dh, j): M = o, N = c, K = output PAIRS - 12 accumulator tiles per wave instead
// of 9 on half the k-steps: 2/3 of the matrix-core work.  fp32 throughout.
// Layout rules on top of conv_wgrad_kernel's: the row stride RS = Tw + 2 is EVEN (a pair never straddles a row) and the
// channel strides DS, XS are 2 (mod 4): a half-wave's 32 channel rows read 8-byte pairs from 32 different bank pairs.

typedef float wf32x2a __attribute__((ext_vector_type(2), aligned(8)));

// development ablation (timing only, results wrong): -DHPVG_ABLW_NODMA stages nothing from inside the K loop
#ifdef HPVG_ABLW_NODMA
#define WGW_ABL_DY {}
#define WGW_ABL_XX { if (cnext < NCH) ++cnext; }
#define WGW_ABL_PIECE(Q) { if (cnext < NCH && (Q) == NP - 1) ++cnext; }
#else
#define WGW_ABL_DY { if (cnext < NCH) dma_dy(cnext); }
#define WGW_ABL_XX { if (cnext < NCH) { dma_xx(cnext); ++cnext; } }
#define WGW_ABL_PIECE(Q) dma_piece(Q);
#endif
// WCH = false: all four waves stage every channel row (lane p of the workgroup moves element p: 64 x 2 pieces of 256 bytes per
// wave and tile, about half of their lanes idle).  WCH = true: wave w stages the WHOLE rows of channels w, w + 4, ... in
// NJD + NJX pieces of 64 lanes (16 x (NJD + NJX) pieces per wave and tile, one behind each MFMA from the second one of a K
// step on, two channels per step when they fit), so the pieces are fewer and the last of them is issued earlier in the K loop
// (eight waves, two pieces per row: behind MFMAs 1 and 4 of the step's six - 1-2 % over 1 and 2)
// one LDS-DMA piece (the builtin wants a literal size)
#define WGW_PIECE(SRC, DST)                                                                      \
  {                                                                                              \
    if constexpr (G16) __builtin_amdgcn_global_load_lds((gptr_t)(SRC), (lptr_t)(DST), 16, 0, 0); \
    else __builtin_amdgcn_global_load_lds((gptr_t)(SRC), (lptr_t)(DST), 4, 0, 0);                \
  }
#ifndef HPVG_WCH_SPREAD
#define HPVG_WCH_SPREAD 1
#endif
// G16 (with WCH; W, the band width and the band origins multiples of 4): the rows are staged in 16-byte pieces - lane =
// (tile row, group of four columns), a dY row of Tw floats (no junk columns: the K loop walks rows x Tw / 4 steps), an X row
// of Tw + 8 floats from column w0 - 4 on, so that every group lies wholly inside the image or wholly outside (zero source);
// the X rows sit one float into their channel row so that the operand pairs (first column w0 + ww - 1) stay 8-byte aligned.
// Two or three pieces per channel row instead of five.
// W8 (with WCH): EIGHT waves, two per SIMD - waves w and w + 4 own the same 32 x 32 (o, c) sub-block and two of the four
// Winograd points each (six accumulator tiles, six MFMAs per K step and wave, selected by coefficients: no divergent code),
// so that one wave's LDS reads, transforms and staging pieces run under the other's MFMAs (with one wave per SIMD nothing
// hides them); each wave stages the rows of eight channels.  Worth 2.5 % with the 16-byte form (stages 8, 9), a loss with
// the 4-byte WCH form (more pieces per wave pair): instantiated for G16 only.  (hipcc reads a patch row as ds_read2_b64;
// forcing two ds_read_b64 - the 64-bank path - is 1 % slower here as well.)
template <int KT, int NJD, int NJX, bool WCH, bool G16, bool W8>
__global__ __launch_bounds__(W8 ? 512 : 256, 1) void conv_wgradw_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NW = W8 ? 8 : 4;              // waves
  constexpr int NT = NW * 64;
  static_assert(!W8 || WCH, "W8 stages whole rows per wave");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int w4 = wave & 3, jh = wave >> 2;    // sub-block, half of the points (W8)
  const int oblk = w4 >> 1, cblk = w4 & 1;
  constexpr int NCH = WCH ? 64 / NW : 64;     // staging calls per tile and wave
  constexpr int LW = WCH ? 64 : 256;          // lanes that share a channel row
  constexpr int PW = G16 ? 256 : LW;          // floats of LDS per piece
  static_assert(!G16 || WCH, "G16 stages whole rows per wave");
  constexpr int NP = NJD + NJX;
  const int lid = WCH ? lane : tid;
  // workgroup ids as in conv_wgrad_kernel: time tap fastest, then the persistent slot; the outer taps get S0 <= S slots
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
  const int pt = (KT == 3 ? 1 : 0);
  const int ntiles = a.B * a.T * a.nth * a.ntw;

  constexpr int JN = W8 ? 2 : 4;              // points per wave
  constexpr int NACC = 3 * JN;
  f32x16 acc[NACC];   // [dh][j]
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  // Bias gradient on the side (a.bpart): db[o] = sum of dY[o] over batch and positions.  The centre-tap workgroups walk every
  // tile, and the waves of their first input-channel half hold y0 + y1 of every dY pair of their 32 output channels (the
  // Winograd Y_1): one more add per K step.  Per tile in fp32, tiles added with Kahan compensation (a lane sums thousands
  // of values), slots summed in double by the reduce kernel.
  const float bflag = (a.bpart != nullptr && dt == pt && cb == 0 && cblk == 0 && jh == 0) ? 1.f : 0.f;
  float bsum = 0.f, brun = 0.f, bcomp = 0.f;

  // zero both buffers once: rows of absent channels are never written afterwards
  for (int i = tid; i < 2 * BUF; i += NT) lds[i] = 0.f;

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* dptr[NJD];
  const char* xptr[NJX];
  unsigned dstr[NJD], xstr[NJX];
  bool dln[NJD], xln[NJX];
  const int gpr = a.Tw >> 2, gprx = gpr + 2;          // G16: 16-byte groups per dY / X row
#pragma unroll
  for (int j = 0; j < NJD; ++j) dln[j] = j * LW + lid < (G16 ? a.Th * gpr : DS);
#pragma unroll
  for (int j = 0; j < NJX; ++j) xln[j] = j * LW + lid < (G16 ? (a.Th + 2) * gprx : XS);
  const unsigned cbytes = (unsigned)(cstride * 4) * (WCH ? (unsigned)NW : 1u);   // to the next channel this wave stages
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
    const int ch0 = WCH ? wave : 0;
    const float* dyb = a.dy + (((long)b * a.Cout + ob * 64 + ch0) * a.T + t) * HW;
    const float* xb = a.x + (((long)b * a.Cin + cb * 64 + ch0) * a.T + (tok ? tt : 0)) * HW;
#pragma unroll
    for (int j = 0; j < NJD; ++j) {
      const int p = j * LW + lid;
      const int hh = G16 ? p / gpr : p / RS, ww = G16 ? 4 * (p - hh * gpr) : p - hh * RS;
      const int gh = h0 + hh, gw = w0 + ww;
      const bool ok = hh < a.Th && ww < a.Tw && gh < a.H && gw < a.W;
      dptr[j] = ok ? (const char*)(dyb + gh * a.W + gw) : (const char*)g_wzero;
      dstr[j] = ok ? cbytes : 0u;
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int p = j * LW + lid;
      const int hh = G16 ? p / gprx : p / RS, ww = G16 ? 4 * (p - hh * gprx) - 3 : p - hh * RS;
      const int gh = h0 + hh - 1, gw = w0 + ww - 1;
      const bool ok = tok && hh < a.Th + 2 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      xptr[j] = ok ? (const char*)(xb + gh * a.W + gw) : (const char*)g_wzero;
      xstr[j] = ok ? cbytes : 0u;
    }
  };
  float* dma_d = lds;
  float* dma_x = lds;
  auto dma_begin = [&](float* buf) {
    dma_d = buf + (WCH ? wave * DS : wave * 64);
    dma_x = buf + 64 * DS + (WCH ? wave * XS : wave * 64) + (G16 ? 1 : 0);
  };
  // (the two halves of a channel are issued behind DIFFERENT MFMAs of the K loop: an LDS-DMA instruction holds the issue port
  // for about one MFMA's duration, and with one wave per SIMD whatever does not fit an MFMA's shadow stalls the matrix pipe)
  auto dma_dy = [&](int c) {
    if ((WCH ? NW * c + wave : c) < no) {
#pragma unroll
      for (int j = 0; j < NJD; ++j) {
        if (dln[j]) WGW_PIECE(dptr[j], dma_d + j * PW);
        dptr[j] += dstr[j];
      }
    }
    dma_d += WCH ? NW * DS : DS;
  };
  auto dma_xx = [&](int c) {
    if ((WCH ? NW * c + wave : c) < nc) {
#pragma unroll
      for (int j = 0; j < NJX; ++j) {
        if (xln[j]) WGW_PIECE(xptr[j], dma_x + j * PW);
        xptr[j] += xstr[j];
      }
    }
    dma_x += WCH ? NW * XS : XS;
  };
  auto dma_channel = [&](int c) {
    dma_dy(c);
    dma_xx(c);
  };
  // WCH: piece q (dY pieces first) of the channel row(s) this wave stages next; the last piece of a row moves on
  int cnext = NCH;                 // next staging call (NCH = nothing left)
  auto dma_piece = [&](int q) __attribute__((always_inline)) {
    if (cnext < NCH) {
      const int ch = NW * cnext + wave;
#pragma unroll
      for (int j = 0; j < NJD; ++j)
        if (q == j) {
          if (ch < no && dln[j]) WGW_PIECE(dptr[j], dma_d + j * PW);
          dptr[j] += dstr[j];
        }
#pragma unroll
      for (int j = 0; j < NJX; ++j)
        if (q == NJD + j) {
          if (ch < nc && xln[j]) WGW_PIECE(xptr[j], dma_x + j * PW);
          xptr[j] += xstr[j];
        }
      if (q == NP - 1) {
        dma_d += NW * DS;
        dma_x += NW * XS;
        ++cnext;
      }
    }
  };

  int tile = slot;
  __syncthreads();  // zero fill done
  if (tile < ntiles) {
    setup(tile);
    dma_begin(lds);
    for (int c = 0; c < NCH; ++c) dma_channel(c);
  }
  __syncthreads();  // (waits for the DMA: pending LDS-DMA counts on vmcnt)

  const int nsteps = a.QK >> 2;  // K-loop iterations: 4 positions = 2 pairs = ONE MFMA k-step per tap-point
  int cur = 0;
  for (; tile < ntiles; tile += nslot) {
    const int next = tile + nslot;
    const bool have_next = next < ntiles;
    float* bufc = lds + cur * BUF;
    cnext = NCH;
    if (have_next) {
#ifdef HPVG_ABLW_NOSETUP   // (development ablation, timing only: every tile stages the first one again, the set-up hoisted)
      setup(slot);
#else
      setup(next);
#endif
      dma_begin(lds + (cur ^ 1) * BUF);
      cnext = 0;
    }
    const int tt_cur = tile % a.T + dt - pt;
#ifdef HPVG_ABLW_NOMMA
    if (false) {   // development ablation (timing only): everything but the K loop
#else
    if (active && tt_cur >= 0 && tt_cur < a.T) {
#endif
      // this lane's pair of step st: positions 4 st + 2 half, + 1
      const float* dl = bufc + (oblk * 32 + l31) * DS + 2 * half;
      const float* xl = bufc + 64 * DS + (cblk * 32 + l31) * XS + 2 * half + (G16 ? 1 : 0);
      // G16: X offset of the step being loaded (row r, step wc of the row: r * RS + 4 * wc + 3; steps are loaded in order)
      int xo = 3, wc = 0;
      // two register sets: the LDS reads of step st+1 are issued before the MFMAs of step st
      wf32x2a pa, pb0[3], pb1[3], qa, qb0[3], qb1[3];
#define WW_LOAD(A, B0, B1, ST)                                                           \
  {                                                                                      \
    const int q0_ = (ST) * 4;                                                            \
    const int x0_ = G16 ? xo : q0_;                                                      \
    A = *reinterpret_cast<const wf32x2a*>(dl + q0_);                                     \
    _Pragma("unroll") for (int dh = 0; dh < 3; ++dh) {                                   \
      B0[dh] = *reinterpret_cast<const wf32x2a*>(xl + x0_ + dh * RS);                    \
      B1[dh] = *reinterpret_cast<const wf32x2a*>(xl + x0_ + dh * RS + 2);                \
    }                                                                                    \
    if (G16) {                                                                           \
      xo += 4;                                                                           \
      if (++wc == gpr) { wc = 0; xo += 8; }                                              \
    }                                                                                    \
  }
// 12 MFMAs with three channels of DMA staging between them (the rate per position of conv_wgrad_kernel): a channel's dY
// piece behind the second MFMA of a group, its X piece behind the fourth
// staging behind MFMA M of a K step
#define WW_AFTER(M)                                                                                                \
  {                                                                                                                \
    if (!WCH) {                                                                                                    \
      if (((M) & 3) == 1) WGW_ABL_DY                                                                               \
      if (((M) & 3) == 3) WGW_ABL_XX                                                                               \
    } else {                                                                                                       \
      if (HPVG_WCH_SPREAD == 0 && (M) >= 1 && (M) <= NP) WGW_ABL_PIECE((M) - 1)                                    \
      if (HPVG_WCH_SPREAD == 0 && 2 * NP <= 11 && (M) > NP && (M) <= 2 * NP) WGW_ABL_PIECE((M) - 1 - NP)           \
      if (HPVG_WCH_SPREAD == 1 && !(W8 && NP == 2) && (M) >= 1 && (M) <= NP) WGW_ABL_PIECE((M) - 1)                \
      if (HPVG_WCH_SPREAD == 1 && W8 && NP == 2 && ((M) == 1 || (M) == 4)) WGW_ABL_PIECE((M) / 3)                  \
      if (HPVG_WCH_SPREAD == 2 && ((M) & 1) == 1 && ((M) >> 1) < NP) WGW_ABL_PIECE((M) >> 1)                       \
      if (HPVG_WCH_SPREAD == 3 && ((M) % 3) == 1 && ((M) / 3) < NP) WGW_ABL_PIECE((M) / 3)                         \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }
#define WW_MMA(A, B0, B1)                                                                                          \
  {                                                                                                                \
    const float y0_ = A[0], y1_ = A[1];                                                                            \
    const float ys_ = y0_ + y1_, yd_ = y0_ - y1_, yn_ = -y1_;                                                      \
    bsum = __builtin_fmaf(bflag, ys_, bsum);                                                                       \
    _Pragma("unroll") for (int dh = 0; dh < 3; ++dh) {                                                             \
      const float d0_ = B0[dh][0], d1_ = B0[dh][1], d2_ = B1[dh][0], d3_ = B1[dh][1];                              \
      acc[dh * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0_, d0_ - d2_, acc[dh * 4 + 0], 0, 0, 0);            \
      WW_AFTER(dh * 4 + 0)                                                                                         \
      acc[dh * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ys_, d1_ + d2_, acc[dh * 4 + 1], 0, 0, 0);            \
      WW_AFTER(dh * 4 + 1)                                                                                         \
      acc[dh * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(yd_, d2_ - d1_, acc[dh * 4 + 2], 0, 0, 0);            \
      WW_AFTER(dh * 4 + 2)                                                                                         \
      acc[dh * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(yn_, d1_ - d3_, acc[dh * 4 + 3], 0, 0, 0);            \
      WW_AFTER(dh * 4 + 3)                                                                                         \
    }                                                                                                              \
  }
// W8: this wave's two points by coefficients (jh = 0: (y0, d0 - d2), (y0 + y1, d1 + d2); jh = 1: (y0 - y1, d2 - d1),
// (-y1, d1 - d3); multiplications by 0 / 1 / -1 are exact)
#define WW_MMA8(A, B0, B1)                                                                                         \
  {                                                                                                                \
    const float y0_ = A[0], y1_ = A[1];                                                                            \
    const float a0_ = __builtin_fmaf(-sj, y1_, y0_);                                                               \
    const float a1_ = __builtin_fmaf(c0j, y0_, c1j * y1_);                                                         \
    bsum = __builtin_fmaf(bflag, a1_, bsum);                                                                       \
    _Pragma("unroll") for (int dh = 0; dh < 3; ++dh) {                                                             \
      const float d0_ = B0[dh][0], d1_ = B0[dh][1], d2_ = B1[dh][0], d3_ = B1[dh][1];                              \
      const float b0_ = __builtin_fmaf(c0j, d0_, __builtin_fmaf(-sj, d1_, -c1j * d2_));                            \
      const float b1_ = __builtin_fmaf(c0j, d2_, __builtin_fmaf(-sj, d3_, d1_));                                   \
      acc[dh * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b0_, acc[dh * 2 + 0], 0, 0, 0);                  \
      WW_AFTER(dh * 2 + 0)                                                                                         \
      acc[dh * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b1_, acc[dh * 2 + 1], 0, 0, 0);                  \
      WW_AFTER(dh * 2 + 1)                                                                                         \
    }                                                                                                              \
  }
#define WW_KLOOP(MMA)                                                                    \
  {                                                                                      \
    int st = 0;                                                                          \
    if (st < nsteps) WW_LOAD(pa, pb0, pb1, st);                                          \
    for (; st + 1 < nsteps; st += 2) {                                                   \
      WW_LOAD(qa, qb0, qb1, st + 1);                                                     \
      MMA(pa, pb0, pb1);                                                                 \
      if (st + 2 < nsteps) WW_LOAD(pa, pb0, pb1, st + 2);                                \
      MMA(qa, qb0, qb1);                                                                 \
    }                                                                                    \
    if (st < nsteps) MMA(pa, pb0, pb1);                                                  \
  }
      const float sj = (float)jh, c0j = 1.f - sj, c1j = 1.f - 2.f * sj;
      if constexpr (W8) WW_KLOOP(WW_MMA8) else WW_KLOOP(WW_MMA)
#undef WW_KLOOP
#undef WW_LOAD
#undef WW_MMA
#undef WW_MMA8
#undef WW_AFTER
    }
    {   // this tile's bias sum into the running one (Kahan)
      const float yk = bsum - bcomp, tk = brun + yk;
      bcomp = (tk - brun) - yk;
      brun = tk;
      bsum = 0.f;
    }
    while (cnext < NCH) { dma_channel(cnext); ++cnext; }  // whatever did not fit into the K loop (short loops, idle waves)
#ifndef HPVG_ABLW_NOBAR   // (development ablation, timing only: no barrier between tiles)
    __syncthreads();  // next buffer complete (the barrier's fence waits for the pending LDS-DMA), current one free
#endif
    cur ^= 1;
  }

  if (a.bpart != nullptr && dt == pt && cb == 0 && cblk == 0 && jh == 0) {
    // the two half-waves hold the even / odd pairs of the same channels
    const float tot = brun + __shfl_xor(brun, 32, 64);
    if (half == 0) a.bpart[((long)slot * a.nob + ob) * 64 + oblk * 32 + l31] = active ? tot : 0.f;
  }
  // ---- partial slab: part[s][dt][z][dh*4 + j][o64][c64]
  float* pp = a.part + ((((long)slot * KT + dt) * nz + z) * 12) * 4096;
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = oblk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      const int kk = W8 ? (k >> 1) * 4 + 2 * jh + (k & 1) : k;     // tile (dh, point)
      pp[(long)kk * 4096 + row * 64 + cblk * 32 + l31] = active ? acc[k][e] : 0.f;
    }
}

// dW[o][c][dt][dh][0..2] = G^T sum_s part[s][dt][z][dh*4 + j][o%64][c%64].  A block is 128 (dt, z, dh, o, c) elements x 8
// slot groups: group g sums the slots [g*S/8, (g+1)*S/8) of the four points in order (four independent chains), the group
// sums are added in group order through LDS (reproducible), then the output transform.
__global__ __launch_bounds__(1024) void conv_wgradw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S1,
                                                                  int S0, int KT, int nob, int ncb, int Cout, int Cin,
                                                                  int accumulate, int nbw, const float* __restrict__ bpart,
                                                                  float* __restrict__ db, int accumulate_db) {
  __shared__ float sm[8][4][128];
  const int g = threadIdx.y;
  if ((int)blockIdx.x >= nbw) {
    // the bias gradient: db[o] = sum over the centre tap's S1 slots of bpart[slot][o / 64][o % 64]: eight slot groups in slot
    // order (two chains each), the group sums in group order through LDS, in double
    double* smd = reinterpret_cast<double*>(&sm[0][0][0]);     // [8][128] doubles
    const int o = ((int)blockIdx.x - nbw) * 128 + threadIdx.x;
    double t0 = 0.0, t1 = 0.0;
    if (o < Cout) {
      const float* q = bpart + (long)(o / 64) * 64 + (o & 63);
      const long sstride = (long)nob * 64;
      int sl = (int)((long)g * S1 / 8);
      const int hi = (int)((long)(g + 1) * S1 / 8);
      for (; sl + 2 <= hi; sl += 2) {
        t0 += (double)q[(long)sl * sstride];
        t1 += (double)q[(long)(sl + 1) * sstride];
      }
      if (sl < hi) t0 += (double)q[(long)sl * sstride];
    }
    smd[g * 128 + threadIdx.x] = t0 + t1;
    __syncthreads();
    if (g == 0 && o < Cout) {
      double t = smd[threadIdx.x];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += smd[k * 128 + threadIdx.x];
      db[o] = accumulate_db ? db[o] + (float)t : (float)t;
    }
    return;
  }
  const long idx = (long)blockIdx.x * 128 + threadIdx.x;
  const long per_s = (long)KT * nob * ncb * 12 * 4096;
  const long total = (long)KT * nob * ncb * 3 * 4096;
  long r = idx < total ? idx : total - 1;
  const int c64 = r % 64; r /= 64;
  const int o64 = r % 64; r /= 64;
  const int dh = r % 3; r /= 3;
  const int z = r % (nob * ncb); r /= (nob * ncb);
  const int dt = (int)r;
  const int o = (z / ncb) * 64 + o64, c = (z % ncb) * 64 + c64;
  const bool live = idx < total && o < Cout && c < Cin;
  const int S = (KT == 3 && dt != 1) ? S0 : S1;  // slots that wrote a slab for this time tap
  const float* p0 = part + (((long)dt * nob * ncb + z) * 12 + dh * 4) * 4096 + o64 * 64 + c64;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  if (live) {
    const int hi = (int)((long)(g + 1) * S / 8);
    for (int sl = (int)((long)g * S / 8); sl < hi; ++sl) {
      const float* q = p0 + (long)sl * per_s;
      m0 += q[0];
      m1 += q[4096];
      m2 += q[2 * 4096];
      m3 += q[3 * 4096];
    }
  }
  sm[g][0][threadIdx.x] = m0;
  sm[g][1][threadIdx.x] = m1;
  sm[g][2][threadIdx.x] = m2;
  sm[g][3][threadIdx.x] = m3;
  __syncthreads();
  if (g == 0 && live) {
    float m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = sm[0][j][threadIdx.x];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += sm[k][j][threadIdx.x];
      m[j] = t;
    }
    const float hs = 0.5f * (m[1] + m[2]), hd = 0.5f * (m[1] - m[2]);
    float* dst = dw + ((((long)o * Cin + c) * KT + dt) * 3 + dh) * 3;
    const float g0 = m[0] + hs, g1 = hd, g2 = hs + m[3];
    dst[0] = accumulate ? dst[0] + g0 : g0;
    dst[1] = accumulate ? dst[1] + g1 : g1;
    dst[2] = accumulate ? dst[2] + g2 : g2;
  }
}
