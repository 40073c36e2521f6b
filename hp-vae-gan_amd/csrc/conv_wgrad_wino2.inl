// Weight gradient of the wide 3x3 / 3x3x3 convs with the TRANSPOSE of the forward Winograd F(2x2, 3x3) (conv_wino2d.inl) -
// included by conv_wgrad.hip inside its anonymous namespace (shares WgradArgs, the zero word and the 16-byte staging form
// with conv_wgradw_kernel, whose structure it keeps: one time tap and one 64 x 64 (o, c) block per persistent workgroup, one
// 32 x 32 sub-block per wave, two LDS tile buffers filled by LDS-DMA in the shadow of the MFMAs, one partial slab at the end).
//
// Same reference call sites as conv_wgrad_kernel (the weight half of aten::convolution_backward, train_video.py:182,200,
// train_image.py:193,215, and the gradient penalty's double backward, modules/utils.py:14-18).
//
// For a 2 x 2 block of output positions ("quad": rows 2R, 2R+1, columns 2C, 2C+1) with gradients y (2 x 2) and the 4 x 4 input
// patch d (rows 2R-1..2R+2, columns 2C-1..2C+2) of one time tap, the 36 multiplies of the nine in-plane taps
//       g[a][b] += sum_{r,s} y[r][s] d[r+a][s+b]
// are regrouped into 16:  M = (A y A^T) (.) (B^T d B),  g += G^T M G,  with the F(2,3) matrices of conv_wgrad_wino.inl along
// both axes (A = [[1,0],[1,1],[1,-1],[0,-1]], B^T d = (d0-d2, d1+d2, d2-d1, d1-d3), G^T m = (m0+(m1+m2)/2, (m1-m2)/2,
// (m1+m2)/2+m3)).  Every point of M is summed over ALL quads, planes and samples first - the K axis of the GEMM counts quads -
// and the (linear) output transform G^T . G runs once per weight in the reduce kernel.  GEMM per (dt, point): M = o, N = c,
// K = quads, two per v_mfma_f32_32x32x2_f32: 16 MFMAs per pair of quads instead of 36 (direct) or 24 (one-axis): 4/9 of the
// direct matrix-core work.  fp32 throughout.  The kernel works with A' = |A| (last row + instead of -: no negations in the K
// loop); the reduce kernel gives point (i, j) the sign s_i s_j, s = (1, 1, 1, -1).
//
// Sixteen points x 16 registers = 256 accumulator registers per wave: one wave per SIMD (256 AGPRs + 256 VGPRs), and the K step
// is laid out in 16 slots of one MFMA + a unit of other work that fits its 64-cycle shadow, pinned with sched_barrier (the
// scheme of conv_wino2d_kernel): slot 0 the LDS reads of the next step's operands, 3 the vertical pass of its dY quad, 4-7 the
// vertical pass of its input patch, 8-11 the horizontal pass, 12-15 the horizontal pass of the dY quad, 1 / 5 / 9 (/ 13) an
// LDS-DMA piece of the next tile.
// Widths: any.  The LDS tile is laid out by TILE columns (band origins are multiples of 4), so the 8-byte operand pairs are
// aligned whatever the row pitch W; the 16-byte global pieces need only 4-byte alignment (tools/glds16_probe.hip).  W % 4 != 0:
// the one 16-byte group per row that straddles the right image border (it holds the row's last W % 4 columns) is zero-sourced
// like every group outside, and the wave that stages the channel patches its 1-3 valid floats in with ordinary in-bounds
// loads + LDS writes after the tile's pieces have landed; an odd W's last quad column has its second column outside the image
// (zeros on both operands) (border-band tiles only: ~4 loads and 8 LDS writes per
// wave and tile, behind the same wait the barrier needs anyway) - nothing is ever read past a row or a tensor.
// Tiles: Th x Tw output positions, Th even, Tw and the band origins multiples of 4: the 16-byte staging form
// of conv_wgradw_kernel - a dY row is Tw floats, an X row Tw + 8 floats from column w0 - 4 one float into its channel row, so
// that the patch pairs (first column w0 + 2C - 1) are 8-byte aligned; groups outside the image are zero-sourced, rows past H
// too (odd H: the second row of the last quad row holds zeros on both sides).

// a.order: the walk over the tiles.  0: time-major (tile = ((b * nth + th) * ntw + tw) * T + t: the slots of an XCD hold one
// spatial tile at neighbouring t); 1: plane-major (tile = ((b * T + t) * nth + th) * ntw + tw: the slots of an XCD hold
// neighbouring tiles of one plane, which share halo columns / rows and the other halves of their 128-byte lines in that XCD's L2)
// STRAD: W % 4 != 0 - a separate instance, so that the usual one carries none of the patch code (present but never executed it
// cost the stage-9 launch 3 %: registers)
template <int KT, int NJD, int NJX, int TWC, bool STRAD>
__global__ __launch_bounds__(256, 1) void conv_wgradw2_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NW = 4, NT = 256, NCH = 16, LW = 64, PW = 256;
  constexpr int NP = NJD + NJX;
  constexpr int PS = NP == 4 ? 4 : 3;         // LDS-DMA piece slots per K step (slots 1, 5, 9 (, 13))
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int oblk = wave >> 1, cblk = wave & 1;
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
  const int RS = a.RS, DS = a.DS, XS = a.XS, Tw = a.Tw;
  const int BUF = 64 * (DS + XS);
  const long HW = (long)a.H * a.W;
  const long cstride = (long)a.T * HW;
  const bool active = (ob * 64 + oblk * 32 < a.Cout) && (cb * 64 + cblk * 32 < a.Cin);
  int no = a.Cout - ob * 64; if (no > 64) no = 64;
  int nc = a.Cin - cb * 64;  if (nc > 64) nc = 64;
  const int pt = (KT == 3 ? 1 : 0);
  const int ntiles = a.B * a.T * a.nth * a.ntw;

  f32x16 acc[16];   // [i][j]: point (i = vertical, j = horizontal)
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  // bias gradient on the side (see conv_wgradw_kernel): the centre-tap workgroups' first input-channel half holds the sum of
  // every dY quad of its 32 output channels (the point (1, 1) of A' y A'^T)
  const float bflag = (a.bpart != nullptr && dt == pt && cb == 0 && cblk == 0) ? 1.f : 0.f;
  float bsum = 0.f, brun = 0.f, bcomp = 0.f;

  for (int i = tid; i < 2 * BUF; i += NT) lds[i] = 0.f;

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const char* dptr[NJD];
  const char* xptr[NJX];
  unsigned dstr[NJD], xstr[NJX];
  bool dln[NJD], xln[NJX];
  const int gpr = Tw >> 2, gprx = gpr + 2;          // 16-byte groups per dY / X row
#pragma unroll
  for (int j = 0; j < NJD; ++j) dln[j] = j * LW + lane < a.Th * gpr;
#pragma unroll
  for (int j = 0; j < NJX; ++j) xln[j] = j * LW + lane < (a.Th + 2) * gprx;
  const unsigned cbytes = (unsigned)(cstride * 4) * (unsigned)NW;   // to the next channel this wave stages
  auto decode = [&](int tile, int& b, int& t, int& th_i, int& tw_i) __attribute__((always_inline)) {
    if (a.order == 0) {
      t = tile % a.T;
      int r = tile / a.T;
      tw_i = r % a.ntw;
      r /= a.ntw;
      th_i = r % a.nth;
      b = r / a.nth;
    } else {
      tw_i = tile % a.ntw;
      int r = tile / a.ntw;
      th_i = r % a.nth;
      r /= a.nth;
      t = r % a.T;
      b = r / a.T;
    }
  };
  // the staged tile's straddling groups (W = 2 mod 4; -1: none): float offset of the group inside a dY row / an X row, and what
  // the patch needs to find the rows again
  int pt_sd = -1, pt_sx = -1, pt_h0 = 0;
  bool pt_tok = false;
  bool st_tok = false;      // the staged tile's input plane lies inside the clip (= it has work for this time tap)
  const float* pt_dyb = nullptr;
  const float* pt_xb = nullptr;
  // this lane's group of piece j inside the tile: row (<< 16) | column offset + 8 - the same for every tile, and a division by a
  // run-time group count per piece: hoisted out of the tile loop (the set-up of a tile is not hidden behind anything: one wave
  // per SIMD)
  int dgrp[NJD], xgrp[NJX];
#pragma unroll
  for (int j = 0; j < NJD; ++j) {
    const int p = j * LW + lane;
    const int hh = p / gpr;
    dgrp[j] = (hh << 16) | (4 * (p - hh * gpr) + 8);
  }
#pragma unroll
  for (int j = 0; j < NJX; ++j) {
    const int p = j * LW + lane;
    const int hh = p / gprx;
    xgrp[j] = (hh << 16) | (4 * (p - hh * gprx) - 3 + 8);
  }
  auto setup = [&](int b, int t, int th_i, int tw_i) {
    const int tt = t + dt - pt;
    const bool tok = tt >= 0 && tt < a.T;
    st_tok = tok;
    const int h0 = th_i * a.Th, w0 = tw_i * Tw;
    const float* dyb = a.dy + (((long)b * a.Cout + ob * 64 + wave) * a.T + t) * HW;
    const float* xb = a.x + (((long)b * a.Cin + cb * 64 + wave) * a.T + (tok ? tt : 0)) * HW;
    if constexpr (STRAD) {
      const int gs = a.W - (a.W & 3);                 // first column of the group that holds the row's last W % 4 columns
      const int sd = gs - w0, sx = gs - (w0 - 4);
      pt_sd = (sd >= 0 && sd < Tw) ? sd : -1;
      pt_sx = (sx >= 0 && sx < Tw + 8) ? sx : -1;
      pt_h0 = h0; pt_tok = tok; pt_dyb = dyb; pt_xb = xb;
    }
#pragma unroll
    for (int j = 0; j < NJD; ++j) {
      const int hh = dgrp[j] >> 16, ww = (dgrp[j] & 0xffff) - 8;
      const int gh = h0 + hh, gw = w0 + ww;
      const bool ok = hh < a.Th && gh < a.H && gw + (STRAD ? 3 : 0) < a.W;   // (a group is loaded when it lies wholly inside the row)
      dptr[j] = ok ? (const char*)(dyb + gh * a.W + gw) : (const char*)g_wzero;
      dstr[j] = ok ? cbytes : 0u;
    }
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      const int hh = xgrp[j] >> 16, ww = (xgrp[j] & 0xffff) - 8;
      const int gh = h0 + hh - 1, gw = w0 + ww - 1;
      const bool ok = tok && hh < a.Th + 2 && gh >= 0 && gh < a.H && gw >= 0 && gw + (STRAD ? 3 : 0) < a.W;
      xptr[j] = ok ? (const char*)(xb + gh * a.W + gw) : (const char*)g_wzero;
      xstr[j] = ok ? cbytes : 0u;
    }
  };
  // W = 2 (mod 4): the two valid floats of the straddling group of every row of the tile staged into `buf`, for the 16 channels
  // this wave stages - after the tile's pieces were issued (the load's wait also covers them: vector-memory results return in
  // order, so the zero words the 16-byte pieces wrote there have landed before these LDS writes)
  auto patch_straddle = [&](float* buf) {
    if (pt_sd < 0 && pt_sx < 0) return;
    const int nrow = 2 * a.Th + 2;                 // per channel: Th dY rows, Th + 2 X rows
    for (int id0 = 0; id0 < NCH * nrow; id0 += 64) {
      const int id = id0 + lane;
      const int k = id / nrow, rr = id - k * nrow;
      const bool isd = rr < a.Th;
      const int hh = isd ? rr : rr - a.Th;
      const int ch = NW * k + wave;
      const int gh = isd ? pt_h0 + hh : pt_h0 + hh - 1;
      const bool live = id < NCH * nrow && gh >= 0 && gh < a.H &&
                        (isd ? (pt_sd >= 0 && ch < no) : (pt_sx >= 0 && ch < nc && pt_tok));
      if (live) {
        const int nv = a.W & 3;                        // 1, 2 or 3 valid floats
        const float* src = (isd ? pt_dyb : pt_xb) + (long)k * NW * cstride + (long)gh * a.W + (a.W - nv);
        float* dst = isd ? buf + ch * DS + hh * Tw + pt_sd : buf + 64 * DS + ch * XS + 1 + hh * RS + pt_sx;
        const float v0 = src[0];
        const float v1 = nv > 1 ? src[1] : 0.f;
        const float v2 = nv > 2 ? src[2] : 0.f;
        dst[0] = v0;
        if (nv > 1) dst[1] = v1;
        if (nv > 2) dst[2] = v2;
      }
    }
  };
  float* dma_d = lds;
  float* dma_x = lds;
  auto dma_begin = [&](float* buf) {
    dma_d = buf + wave * DS;
    dma_x = buf + 64 * DS + wave * XS + 1;
  };
#define WG2_PIECE(SRC, DST) __builtin_amdgcn_global_load_lds((gptr_t)(SRC), (lptr_t)(DST), 16, 0, 0);
  int cnext = NCH;                 // next channel row (of this wave's 16) to stage; NCH = nothing left
  // piece q (dY pieces first) of the channel row this wave stages next; the last piece of a row moves on
  auto dma_piece = [&](int q) __attribute__((always_inline)) {
    if (cnext < NCH) {
      const int ch = NW * cnext + wave;
#pragma unroll
      for (int j = 0; j < NJD; ++j)
        if (q == j) {
          if (ch < no && dln[j]) WG2_PIECE(dptr[j], dma_d + j * PW);
          dptr[j] += dstr[j];
        }
#pragma unroll
      for (int j = 0; j < NJX; ++j)
        if (q == NJD + j) {
          if (ch < nc && xln[j]) WG2_PIECE(xptr[j], dma_x + j * PW);
          xptr[j] += xstr[j];
        }
      if (q == NP - 1) {
        dma_d += NW * DS;
        dma_x += NW * XS;
        ++cnext;
      }
    }
  };
  auto dma_channel = [&]() {
#pragma unroll
    for (int q = 0; q < NP; ++q) dma_piece(q);
  };

  int tile = slot;
  __syncthreads();  // zero fill done
  // the tile's coordinates (sample, plane, band row, band column) move on by the decoded step `nslot` with carries instead of
  // three divisions per tile (order 1: the column runs fastest; order 0: the plane)
  int cb_ = 0, ct_ = 0, ch_ = 0, cw_ = 0, sb_ = 0, st_ = 0, sh_ = 0, sw_ = 0;
  decode(tile < ntiles ? tile : 0, cb_, ct_, ch_, cw_);
  decode(nslot, sb_, st_, sh_, sw_);
  auto advance = [&]() __attribute__((always_inline)) {
    if (a.order == 0) {
      ct_ += st_; int c = ct_ >= a.T ? 1 : 0; ct_ -= c ? a.T : 0;
      cw_ += sw_ + c; c = cw_ >= a.ntw ? 1 : 0; cw_ -= c ? a.ntw : 0;
      ch_ += sh_ + c; c = ch_ >= a.nth ? 1 : 0; ch_ -= c ? a.nth : 0;
      cb_ += sb_ + c;
    } else {
      cw_ += sw_; int c = cw_ >= a.ntw ? 1 : 0; cw_ -= c ? a.ntw : 0;
      ch_ += sh_ + c; c = ch_ >= a.nth ? 1 : 0; ch_ -= c ? a.nth : 0;
      ct_ += st_ + c; c = ct_ >= a.T ? 1 : 0; ct_ -= c ? a.T : 0;
      cb_ += sb_ + c;
    }
  };
  if (tile < ntiles) {
    setup(cb_, ct_, ch_, cw_);
    dma_begin(lds);
    cnext = 0;
    while (cnext < NCH) dma_channel();
    if constexpr (STRAD) patch_straddle(lds);
  }
  __syncthreads();  // (waits for the DMA: pending LDS-DMA counts on vmcnt)

  const int nsteps = a.QK >> 3;  // K-loop iterations: two quads = 8 positions = ONE MFMA k-step per point
  int cur = 0;
  bool cur_tok = st_tok;      // (set by the first tile's setup)
  for (; tile < ntiles; tile += nslot) {
    const int next = tile + nslot;
    const bool have_next = next < ntiles;
    float* bufc = lds + cur * BUF;
    cnext = NCH;
    if (have_next) {
#ifdef HPVG_ABLW2_NOSETUP   // (development ablation, timing only: every tile stages the first one again, the set-up hoisted)
      { int b0_, t0_, h0_, w0_; decode(slot, b0_, t0_, h0_, w0_); setup(b0_, t0_, h0_, w0_); }
#else
      advance();
      setup(cb_, ct_, ch_, cw_);
#endif
      dma_begin(lds + (cur ^ 1) * BUF);
      cnext = 0;
    }
#ifdef HPVG_ABLW2_NOMMA
    if (false) {   // development ablation (timing only): everything but the K loop
#else
    const bool work = cur_tok;
    cur_tok = st_tok;           // (setup(next) above decoded the next tile)
    if (active && work) {
#endif
      // this lane's quad of step st: quad 2 st + half of the tile's row-major quad index (Tw / 2 quads per quad row, an even
      // count: both halves of a step sit in the same quad row)
      // VALU instructions do NOT run in the shadow of an fp32 MFMA (tools/mfma_fillers.hip: v_mfma_f32_32x32x2_f32 and the
      // vector ALU share the fp32 datapath - every v_add / v_mov between two MFMAs adds its 4-5 cycles to the 64; LDS reads and
      // scalar instructions are free), so the K step is written for the fewest vector instructions: packed transforms
      // (v_pk_add_f32 on the 8-byte pairs as they come from LDS: 22 per step), operands consumed where they are produced (no
      // copies: two register sets A / B), one address register per operand (TWC: the row strides are immediates).
      const float* dl = bufc + (oblk * 32 + l31) * DS + 2 * half;
      const float* xl = bufc + 64 * DS + (cblk * 32 + l31) * XS + 2 * half + 4;
      int dyo = 0, xo = 0, wc = 0, ls = 0;   // offsets of the step being LOADED (steps are loaded in order, then wrap)
      // this lane's addresses of that step, made in the step's vector slot (WG2_ADDR: not a lone vector add beside the reads)
      // (32-bit LDS addresses: a generic pointer laundered through an asm statement loses its address space - flat loads)
      typedef __attribute__((address_space(3))) const wf32x2a* wg2_lp;
      const unsigned dl0 = (unsigned)(size_t)(lptr_t)const_cast<float*>(dl), xl0 = (unsigned)(size_t)(lptr_t)const_cast<float*>(xl);
      unsigned dlp = dl0, xlp = xl0;
#define WG2_ADDR() { dlp = dl0 + 4u * (unsigned)dyo; xlp = xl0 + 4u * (unsigned)xo; asm volatile("" : "+v"(dlp), "+v"(xlp)); }
      wf32x2a rx[4][2] = {};                  // raw input patch of the next step: rx[row][pair]
      // per register set: the dY quad rows ry[0..1] (also points (0, 0/3) and (3, 0/3)), their sum / difference rv1, rv2 (rows
      // 1, 2: points (i, 0/3)), y12[i] = points (i, 1), (i, 2); v03[i] = points (i, 0), (i, 3), v12[i] = (i, 1), (i, 2)
      wf32x2a ryA[2] = {}, rv1A, rv2A, y12A[4], v03A[4], v12A[4];
      wf32x2a ryB[2] = {}, rv1B, rv2B, y12B[4], v03B[4], v12B[4];
      const int TwR = TWC > 0 ? TWC : Tw, RSR = TWC > 0 ? TWC + 8 : RS;
#ifdef HPVG_ABLW2_NOLDS
#define WG2_LOAD(RY) { if (++ls == nsteps) ls = 0; }
#else
#define WG2_LOAD(RY)                                                                        \
  {                                                                                         \
    RY[0] = *(wg2_lp)(size_t)(dlp);                                                         \
    RY[1] = *(wg2_lp)(size_t)(dlp + 4u * (unsigned)TwR);                                    \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                      \
      rx[r_][0] = *(wg2_lp)(size_t)(xlp + 4u * (unsigned)(r_ * RSR));                       \
      rx[r_][1] = *(wg2_lp)(size_t)(xlp + 4u * (unsigned)(r_ * RSR + 2));                   \
    }                                                                                       \
    dyo += 4;                                                                               \
    xo += 4;                                                                                \
    const bool roww_ = ++wc == gpr;                                                         \
    wc = roww_ ? 0 : wc;                                                                    \
    dyo += roww_ ? TwR : 0;                                                                 \
    xo += roww_ ? 2 * RSR - TwR : 0;                                                        \
    const bool tilew_ = ++ls == nsteps;                                                     \
    ls = tilew_ ? 0 : ls;                                                                   \
    wc = tilew_ ? 0 : wc;                                                                   \
    dyo = tilew_ ? 0 : dyo;                                                                 \
    xo = tilew_ ? 0 : xo;                                                                   \
  }
#endif
// (lo, hi) -> (src.lo + oth.hi, src.lo - oth.hi): the two middle points of a row from its pairs
#define WG2_MID(S, O) (__builtin_shufflevector(S, S, 0, 0) + __builtin_shufflevector(O, O, 1, 1) * wf32x2a{1.f, -1.f})
#ifdef HPVG_ABLW2_NOXF
#define WG2_YXF(RY, RV1, RV2, Y12) { RV1 = RY[0]; RV2 = RY[1]; Y12[0] = RY[0]; Y12[1] = RY[1]; Y12[2] = RY[0]; Y12[3] = RY[1]; }
#define WG2_XV() {}
#define WG2_XH(I, V03, V12) { V03[I] = rx[I][0]; V12[I] = rx[I][1]; }
#else
// dY quad: vertical pass rv1 = row0 + row1, rv2 = row0 - row1 (rows 0 / 3 are the raw rows), then per row P = (p0, p1) the
// points (P.lo, P.lo + P.hi, P.lo - P.hi, P.hi)
#define WG2_YXF(RY, RV1, RV2, Y12)                                                          \
  {                                                                                         \
    RV1 = RY[0] + RY[1];                                                                    \
    RV2 = RY[0] - RY[1];                                                                    \
    Y12[0] = WG2_MID(RY[0], RY[0]);                                                         \
    Y12[1] = WG2_MID(RV1, RV1);                                                             \
    Y12[2] = WG2_MID(RV2, RV2);                                                             \
    Y12[3] = WG2_MID(RY[1], RY[1]);                                                         \
  }
// input patch, vertical pass on the column pairs: tn[i][pair] = (B^T d)[i][2 pair .. 2 pair + 1]
#define WG2_XV()                                                                            \
  {                                                                                         \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                      \
      tn[0][p_] = rx[0][p_] - rx[2][p_];                                                    \
      tn[1][p_] = rx[1][p_] + rx[2][p_];                                                    \
      tn[2][p_] = rx[2][p_] - rx[1][p_];                                                    \
      tn[3][p_] = rx[1][p_] - rx[3][p_];                                                    \
    }                                                                                       \
  }
// horizontal pass of row I: A = tn[I][0] = (t0, t1), B = tn[I][1] = (t2, t3): (V0, V3) = A - B, (V1, V2) = (t2 + t1, t2 - t1)
#define WG2_XH(I, V03, V12)                                                                 \
  {                                                                                         \
    V03[I] = tn[I][0] - tn[I][1];                                                           \
    V12[I] = WG2_MID(tn[I][1], tn[I][0]);                                                   \
  }
#endif
      wf32x2a tn[4][2];
#define WG2_TRANSFORM(RY, RV1, RV2, Y12, V03, V12)                                          \
  {                                                                                         \
    WG2_YXF(RY, RV1, RV2, Y12) WG2_XV()                                                     \
    WG2_XH(0, V03, V12) WG2_XH(1, V03, V12) WG2_XH(2, V03, V12) WG2_XH(3, V03, V12)         \
  }
#ifdef HPVG_ABLW2_NODMA
#define WG2_DMA(Q) { if (cnext < NCH && (Q) == NP - 1) ++cnext; }
#else
#define WG2_DMA(Q) dma_piece(Q);
#endif
// operands of point K = 4 i + j out of a register set
#define WG2_YOP(K, RY, RV1, RV2, Y12)                                                                              \
  (((K) & 3) == 0 ? ((K) >> 2 == 0 ? RY[0][0] : ((K) >> 2 == 1 ? RV1[0] : ((K) >> 2 == 2 ? RV2[0] : RY[1][0])))     \
   : ((K) & 3) == 3 ? ((K) >> 2 == 0 ? RY[0][1] : ((K) >> 2 == 1 ? RV1[1] : ((K) >> 2 == 2 ? RV2[1] : RY[1][1])))  \
   : Y12[(K) >> 2][((K) & 3) - 1])
#define WG2_VOP(K, V03, V12) (((K) & 3) == 0 ? V03[(K) >> 2][0] : (((K) & 3) == 3 ? V03[(K) >> 2][1] : V12[(K) >> 2][((K) & 3) - 1]))
// slot K of a step of parity SP: one MFMA on the current set (C) + a unit of the next step's preparation (into set N)
#define WG2_SLOT(SP, K, C, N)                                                                                      \
  {                                                                                                                \
    acc[K] = __builtin_amdgcn_mfma_f32_32x32x2f32(WG2_YOP(K, ry##C, rv1##C, rv2##C, y12##C), WG2_VOP(K, v03##C, v12##C), acc[K], 0, 0, 0); \
    if ((K) == 0) WG2_LOAD(ry##N)                                                                                  \
    WG2_XF_SLOTS(K, C, N)                                                                                          \
    if (((K) & 3) == 1 && ((K) >> 2) < PS) WG2_DMA(((SP) * PS + ((K) >> 2)) % NP)                                  \
    WG2_PIN                                                                                                        \
  }
// (the slots are pinned: left to itself hipcc gathers the vector work of a step behind its first MFMAs, where the LDS reads
// issued in slot 0 have not landed yet - stage 9 -2.4 %, stage 8 -5 %)
#define WG2_PIN __builtin_amdgcn_sched_barrier(0);
// the whole vector work of a step in ONE slot (after the LDS reads of slot 0 have had four MFMAs to land): beside fp32 MFMAs a
// group of vector instructions costs ~10 cycles + ~4.5 per instruction (tools/mfma_fillers.hip: 4 per gap +29 cycles, 8: +45,
// 16: +82; a lone one +14) - spread over seven slots the step paid that fixed part seven times (-1 %)
#define WG2_XF_SLOTS(K, C, N)                                                                                      \
    if ((K) == 4) { WG2_XV() WG2_XH(0, v03##N, v12##N) WG2_XH(1, v03##N, v12##N) WG2_XH(2, v03##N, v12##N) WG2_XH(3, v03##N, v12##N) \
                    WG2_YXF(ry##N, rv1##N, rv2##N, y12##N)                                                         \
                    bsum = __builtin_fmaf(bflag, y12##C[1][0], bsum);                                              \
                    WG2_ADDR() }
#define WG2_STEP(SP, C, N)                                                                                         \
  {                                                                                                                \
    WG2_SLOT(SP, 0, C, N) WG2_SLOT(SP, 1, C, N) WG2_SLOT(SP, 2, C, N) WG2_SLOT(SP, 3, C, N)                        \
    WG2_SLOT(SP, 4, C, N) WG2_SLOT(SP, 5, C, N) WG2_SLOT(SP, 6, C, N) WG2_SLOT(SP, 7, C, N)                        \
    WG2_SLOT(SP, 8, C, N) WG2_SLOT(SP, 9, C, N) WG2_SLOT(SP, 10, C, N) WG2_SLOT(SP, 11, C, N)                      \
    WG2_SLOT(SP, 12, C, N) WG2_SLOT(SP, 13, C, N) WG2_SLOT(SP, 14, C, N) WG2_SLOT(SP, 15, C, N)                    \
  }
      // first step's operands, then two steps per iteration: A -> B -> A.  ONE loop body holds every MFMA of the kernel
      // (nsteps is even: the planner's tiles have Th * Tw % 16 == 0): hipcc gives the accumulators of a peeled step other
      // registers and copies / spills all 256 around it.  The last step therefore prepares operands as well (WG2_LOAD wraps to
      // the tile's first step: in-bounds reads whose results are dropped).
      WG2_LOAD(ryA)
      WG2_TRANSFORM(ryA, rv1A, rv2A, y12A, v03A, v12A)
      WG2_ADDR()
      for (int st = 0; st < nsteps; st += 2) {
        WG2_STEP(0, A, B)
        WG2_STEP(1, B, A)
      }
#undef WG2_STEP
#undef WG2_ADDR
#undef WG2_XF_SLOTS
#undef WG2_PIN
#undef WG2_SLOT
#undef WG2_VOP
#undef WG2_YOP
#undef WG2_DMA
#undef WG2_TRANSFORM
#undef WG2_XH
#undef WG2_XV
#undef WG2_YXF
#undef WG2_MID
#undef WG2_LOAD
    }
    {   // this tile's bias sum into the running one (Kahan)
      const float yk = bsum - bcomp, tk = brun + yk;
      bcomp = (tk - brun) - yk;
      brun = tk;
      bsum = 0.f;
    }
    while (cnext < NCH) dma_channel();  // whatever did not fit into the K loop (short loops, idle waves, skipped tiles)
    if constexpr (STRAD) { if (have_next) patch_straddle(lds + (cur ^ 1) * BUF); }
    __syncthreads();  // next buffer complete (the barrier's fence waits for the pending LDS-DMA), current one free
    cur ^= 1;
  }
#undef WG2_PIECE

  if (a.bpart != nullptr && dt == pt && cb == 0 && cblk == 0) {
    // the two half-waves hold the even / odd quads of the same channels
    const float tot = brun + __shfl_xor(brun, 32, 64);
    if (half == 0) a.bpart[((long)slot * a.nob + ob) * 64 + oblk * 32 + l31] = active ? tot : 0.f;
  }
  // ---- partial slab: part[s][dt][z][i*4 + j][o64][c64]
  float* pp = a.part + ((((long)slot * KT + dt) * nz + z) * 16) * 4096;
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = oblk * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      pp[(long)k * 4096 + row * 64 + cblk * 32 + l31] = active ? acc[k][e] : 0.f;
    }
}

// dW[o][c][dt][a][b] = (G^T M G)[a][b],  M[i][j] = s_i s_j sum_s part[s][dt][z][i*4 + j][o%64][c%64].  A block is one (dt, z, o)
// row of 64 input channels x 16 threadIdx.y = (slot group g of 4, point row i): each thread sums the slots [g*S/4, (g+1)*S/4)
// of its four points (i, 0..3) in order (four independent chains), the group sums are added in group order through LDS
// (reproducible), then nine of the sixteen y rows apply the output transform for one (a, b) each.  Blocks past nbw: the bias
// gradient (as in conv_wgradw_reduce_kernel).
__global__ __launch_bounds__(1024) void conv_wgradw2_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S1,
                                                                   int S0, int KT, int nob, int ncb, int Cout, int Cin,
                                                                   int accumulate, int nbw, const float* __restrict__ bpart,
                                                                   float* __restrict__ db, int accumulate_db) {
  __shared__ float sm[4][16][64];
  const int y = threadIdx.y, tx = threadIdx.x;
  if ((int)blockIdx.x >= nbw) {
    double* smd = reinterpret_cast<double*>(&sm[0][0][0]);     // [16][64] doubles
    const int o = ((int)blockIdx.x - nbw) * 64 + tx;
    double t0 = 0.0;
    if (o < Cout) {
      const float* q = bpart + (long)(o / 64) * 64 + (o & 63);
      const long sstride = (long)nob * 64;
      const int hi = (int)((long)(y + 1) * S1 / 16);
      for (int sl = (int)((long)y * S1 / 16); sl < hi; ++sl) t0 += (double)q[(long)sl * sstride];
    }
    smd[y * 64 + tx] = t0;
    __syncthreads();
    if (y == 0 && o < Cout) {
      double t = smd[tx];
#pragma unroll
      for (int k = 1; k < 16; ++k) t += smd[k * 64 + tx];
      db[o] = accumulate_db ? db[o] + (float)t : (float)t;
    }
    return;
  }
  int r = blockIdx.x;
  const int o64 = r % 64; r /= 64;
  const int z = r % (nob * ncb); r /= (nob * ncb);
  const int dt = r;
  const int o = (z / ncb) * 64 + o64, c = (z % ncb) * 64 + tx;
  const bool live = o < Cout && c < Cin;
  const int S = (KT == 3 && dt != 1) ? S0 : S1;  // slots that wrote a slab for this time tap
  const long per_s = (long)KT * nob * ncb * 16 * 4096;
  const int g = y >> 2, i = y & 3;
  const float* p0 = part + (((long)dt * nob * ncb + z) * 16 + i * 4) * 4096 + o64 * 64 + tx;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
  if (live) {
    const int hi = (int)((long)(g + 1) * S / 4);
    for (int sl = (int)((long)g * S / 4); sl < hi; ++sl) {
      const float* q = p0 + (long)sl * per_s;
      m0 += q[0];
      m1 += q[4096];
      m2 += q[2 * 4096];
      m3 += q[3 * 4096];
    }
  }
  sm[g][i * 4 + 0][tx] = m0;
  sm[g][i * 4 + 1][tx] = m1;
  sm[g][i * 4 + 2][tx] = m2;
  sm[g][i * 4 + 3][tx] = m3;
  __syncthreads();
  if (y < 9 && live) {
    const int ta = y / 3, tb = y - 3 * ta;
    // G^T along one axis: tap 0 = m0 + (m1 + m2)/2, tap 1 = (m1 - m2)/2, tap 2 = (m1 + m2)/2 + m3, with m3 carrying the sign -1
    float col[4];   // col[i] = (row i of M) transformed along j for tap tb
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      float m[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = ((sm[0][ii * 4 + j][tx] + sm[1][ii * 4 + j][tx]) + sm[2][ii * 4 + j][tx]) + sm[3][ii * 4 + j][tx];
      const float hs = 0.5f * (m[1] + m[2]), hd = 0.5f * (m[1] - m[2]);
      col[ii] = tb == 0 ? m[0] + hs : (tb == 1 ? hd : hs - m[3]);
    }
    const float hs = 0.5f * (col[1] + col[2]), hd = 0.5f * (col[1] - col[2]);
    const float gv = ta == 0 ? col[0] + hs : (ta == 1 ? hd : hs - col[3]);
    float* dst = dw + ((((long)o * Cin + c) * KT + dt) * 3 + ta) * 3 + tb;
    *dst = accumulate ? *dst + gv : gv;
  }
}
