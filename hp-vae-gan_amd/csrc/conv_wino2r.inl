// Winograd F(2x2, 3x3) over (H, W), second generation: the 16 points are split over the four waves BY ROW - included by
// conv_mfma.hip inside its anonymous namespace, behind the same launches, weight pack, tile geometry, staging and schedule as
// conv_wino2d_kernel (conv_wino2d.inl: read that header first; reference call sites: nn.Conv3d of ConvBlock3D(SN),
// modules/networks_3d.py:48-70, and its backward-data pass).
//
// Why: the fp32 MFMA and the vector ALU share one fp32 datapath on this part (tools/mfma_fillers.hip, run on MI355X: every
// v_add / v_fma / v_mov between two v_mfma_f32_32x32x2_f32 adds its ~5 cycles to the MFMA's 64 - nothing runs in the MFMA's
// shadow but LDS reads, scalar and memory instructions), so a Winograd kernel's time is  MFMA cycles + vector-instruction
// cycles  and the input transform is paid in full.  conv_wino2d_kernel gives a wave one 32 x 32 block (m-tile x quad half) and
// all 16 points: the two waves that share a quad half compute the SAME V = B^T d B (32 vector instructions per step each), and
// each wave loads 4 U fragments per step.  Here wave i owns point ROW i of all four blocks (2 m-tiles x 2 quad halves x 4
// points (i, 0..3) = 16 accumulator tiles, the same 256 registers): per step it transforms row i only - two patch rows of each
// of its two quads, 8 fma for the vertical pass + 8 for the horizontal one, 16 instead of 32 - and loads 2 U fragments (the
// pack already stores the four points of a row together).  The U ring shrinks from 96 to 48 registers (no spills).
// Price: the output transform Y = A^T M A needs all four rows of a block: at the end of a tile every wave applies the
// horizontal half to its rows (16 -> 8 values per accumulator element), hands three blocks' worth to the other waves through
// the two input buffers that are idle at that point (96 KB, conflict-free 16-byte writes / reads, two barriers per tile:
// ~1.5 % of a tile's 1536 MFMAs), and finishes the block it owns (wave = quad half * 2 + m-tile, as in conv_wino2d_kernel)
// with the unchanged epilogue.
// ODD (a separate instance): odd W.  The LDS image of a plane is a verbatim copy of memory, so with an odd row pitch every other
// patch row starts at an odd float: the rows are read with four ds_read_b32 each instead of two ds_read_b64 (LDS reads are free
// beside the MFMAs), the last quad column's input columns W, W + 1 are zeroed by a third border factor (f2), its second output
// column is not stored, and the 1-bit mask words are accessed one by one; the plane-end patch (TAIL) copies H * W % 4 floats.

#define W2R_WAIT_A(N, S) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(au[S][0]), "+v"(au[S][1]) : : "memory")

#ifdef HPVG_ABL2_NOSTAGE
#define W2R_ABL_STAGE(P) {}
#else
#define W2R_ABL_STAGE(P) W2R_STAGE(P)
#endif
#ifdef HPVG_ABL2_NOA
#define W2R_ABL_LOADA(D, O, B) {}
#else
#define W2R_ABL_LOADA(D, O, B) w2_load_a(D, O, B)
#endif

template <int VAR, bool TAIL, bool ODD>
__global__ __launch_bounds__(256, 1) void conv_wino2r_kernel(const Wino2Args a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int mw = wave & 1, nw = wave >> 1;          // the block this wave owns in the epilogue
  // point row of this wave: vertical pass  t[c] = d[ra][c] + sv * d[rb][c]  (B^T rows: d0 - d2, d1 + d2, d2 - d1, d1 - d3)
  const int ri = wave;
  const int ra = ri == 0 ? 0 : (ri == 2 ? 2 : 1);
  const int rb = ri == 2 ? 1 : (ri == 3 ? 3 : 2);
  const float sv = ri == 1 ? 1.f : -1.f;
  constexpr int PL = W2_PL;
  constexpr int BUFF = 12 * PL;
  const int W = a.W, HWp = a.H * a.W;
  const long HWb = (long)HWp * 4;
  const int S = gridDim.x;
  const int g = hpvg_xcd_remap(blockIdx.x, S);
  if (g >= a.ntl) return;
  const int nmy = (a.ntl - 1 - g) / S + 1;          // this workgroup's tiles: g, g + S, ...
  const int nsc = a.nsc;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)xs;
  const char* zero_ptr = reinterpret_cast<const char*>(g_zero_word);
  const char* zero_plane = reinterpret_cast<const char*>(g_zero_plane);

  // U-fragment stream: [sc][step = dt*2 + cp][row i][m-tile] fragments of 64 lanes x 16 bytes (the four points of a row)
  const long afrag = (long)a.mbtot * 64 * 16;        // bytes between two rows' fragments
  const unsigned aoff = (unsigned)(lane * 16);
  f32x4 au[6][2];                                    // ring: slot = step of the sub-chunk; [m-tile]

  const int nq = a.R * a.Cq;                        // quads of a plane
  auto decode = [&](int tile, int& b, int& t, int& tp, int& yb) __attribute__((always_inline)) {
    yb = tile % a.gridy;
    int r = tile / a.gridy;
    t = r % a.T; r /= a.T;
    tp = r % a.ntq;
    b = r / a.ntq;
  };
  auto span_lo4 = [&](int tp) __attribute__((always_inline)) -> int {
    const int Q0 = tp * 64, R0 = Q0 / a.Cq, c0 = Q0 - R0 * a.Cq;
    const int lo = (2 * R0 - 1) * W + 2 * c0 - 1;               // first input element of the tile (odd)
    return lo >= 0 ? (lo & ~3) : -((3 - lo) & ~3);
  };
  struct StageT { int b, t; unsigned voff; bool ok, zl; unsigned long long okm; bool anyz; int tail; };
  auto stage_setup = [&](int tile) __attribute__((always_inline)) -> StageT {
    StageT q;
    int tp, yb;
    decode(tile, q.b, q.t, tp, yb);
    const int lo4 = span_lo4(tp);
    int Ql = tp * 64 + 63;
    if (Ql > nq - 1) Ql = nq - 1;
    const int R1 = Ql / a.Cq, c1 = Ql - R1 * a.Cq;
    const int hi = (2 * R1 + 2) * W + 2 * c1 + 3;                // one past the last input element
    const int ng = (hi - lo4 + 3) >> 2;
    const int i0 = lo4 + 4 * tid;
    q.ok = tid < ng && i0 >= 0 && i0 + 4 <= HWp;                 // a group is loaded when it lies wholly inside the plane
    q.voff = q.ok ? (unsigned)i0 * 4u : 0u;
    // lanes of the span that lie outside the plane (image rows above / below): never loaded (EXEC), their LDS words are zeroed
    // by the lane itself when the item is staged; lanes past the span are never read
    q.zl = tid < ng && !q.ok;
    q.okm = __builtin_amdgcn_ballot_w64(q.ok);
    q.anyz = __builtin_amdgcn_ballot_w64(q.zl) != 0ull;
    const int gt = (HWp - lo4) >> 2;
    q.tail = TAIL ? __builtin_amdgcn_readfirstlane(gt < ng ? gt : -1) : -1;   // uniform: keep it scalar
    return q;
  };
  // (the tile is decoded again inside the rare branch: keeping b / t of two tiles alive through the K loop for it cost the TAIL
  // instance its scalar registers - 24 v_readfirstlane and a scratch reload per item)
  // Twelve lanes of the wave that owns the tail lane take one plane each (one load latency per item instead of twelve in a row:
  // with the serial loop the plane's last tile took ~2.5x as long as the others - the whole launch when it is a single round,
  // 18 % at stage 7); same wave as the lane that zero-filled the group (stage_zero), so the LDS writes stay in order.
  auto stage_tail = [&](int tile_, int tail_, int sc, int bf) __attribute__((always_inline)) {
    if (tail_ < 0) return;
    if ((tid >> 6) == (tail_ >> 6)) {
      if (lane < 12) {
        int qb, qt, tp, yb;
        decode(tile_, qb, qt, tp, yb);
        const int pl = lane;
        const int cc = pl / 3, dt = pl - 3 * cc;
        const int tt = qt + dt - 1;
        if (sc * 4 + cc < a.Cin && tt >= 0 && tt < a.T) {
          const int nv = HWp & 3;                    // floats of the plane in its last group (2 for even W; 1 or 3 for odd H * W)
          const float* src = reinterpret_cast<const float*>(a.x) + ((((long)qb * a.Cin + (long)sc * 4 + cc) * a.T + tt) * (long)HWp) + (HWp - nv);
          float* dst = xs + bf * BUFF + pl * PL + 1 + 4 * tail_;
          const float v0 = src[0];
          const float v1 = nv > 1 ? src[1] : 0.f;
          const float v2 = nv > 2 ? src[2] : 0.f;
          dst[0] = v0;
          if (nv > 1) dst[1] = v1;
          if (nv > 2) dst[2] = v2;
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
  };

  // how this lane reads the patch rows ra / rb of its TWO quads (one per quad half) and their image-border factors
  struct ReadT { int ba[2], bb[2]; float f0[2], f3[2], f2[2]; };
  int c_yb = 0, c_b = 0, c_t = 0, c_h = 0, c_w = 0;
  bool c_vq = false;
  auto read_setup = [&](int tile) __attribute__((always_inline)) -> ReadT {
    ReadT q;
    int b, t, tp, yb;
    decode(tile, b, t, tp, yb);
    const int lo4 = span_lo4(tp);
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      int Q = tp * 64 + qh * 32 + l31;
      if (Q > nq - 1) Q = nq - 1;                                // lanes past the plane's last quad read (and discard) its patch
      const int Rq = Q / a.Cq, w = 2 * (Q - Rq * a.Cq);
      const int base = half * 3 * PL + 1 + ((2 * Rq - 1) * W + w - 1 - lo4);   // even; + (2 cp * 3 + dt) * PL + r * W + c
      q.ba[qh] = base + ra * W;
      q.bb[qh] = base + rb * W;
      q.f0[qh] = w == 0 ? 0.f : 1.f;
      q.f3[qh] = w + 2 >= W ? 0.f : 1.f;
      q.f2[qh] = w + 1 >= W ? 0.f : 1.f;     // (odd W only: the last quad column's second column is outside the image)
    }
    return q;
  };
  auto cmp_setup = [&](int tile) __attribute__((always_inline)) {   // the owned block's quad of this lane (epilogue)
    int tp;
    decode(tile, c_b, c_t, tp, c_yb);
    const int Q = tp * 64 + nw * 32 + l31;
    c_vq = Q < nq;
    const int Rq = Q / a.Cq;
    c_h = 2 * Rq;
    c_w = 2 * (Q - Rq * a.Cq);
  };
  const long THWb = (long)a.T * HWb;
  unsigned vzero = 0u;
  asm volatile("" : "+v"(vzero));                    // a zero offset that lives in a vector register (the zero-plane pieces)
  // the LDS words of this lane's group in the 12 plane slots of buffer bf := 0, for the lanes of the span outside the plane
  // (only tiles at the top / bottom of a plane have any: a wave-uniform test first)
  auto stage_zero = [&](const StageT& q, int bf) __attribute__((always_inline)) {
    if (q.anyz) {
      if (q.zl) {
        float* dst = xs + bf * BUFF + 1 + 4 * tid;
#pragma unroll
        for (int pl = 0; pl < 12; ++pl) {
          dst[pl * PL + 0] = 0.f; dst[pl * PL + 1] = 0.f; dst[pl * PL + 2] = 0.f; dst[pl * PL + 3] = 0.f;
        }
      }
    }
  };
  auto plane0 = [&](const StageT& q, int sc) __attribute__((always_inline)) -> const char* {
    return reinterpret_cast<const char*>(a.x) + (((long)q.b * a.Cin + (long)sc * 4) * a.T + (q.t - 1)) * HWb;
  };

  // raw rows of the next step's two patches: rwa[qh][pair], rwb[qh][pair]; XA / XB: this lane's LDS byte addresses of its rows.
  // Hand-written ds_read_b64 with the plane offset as a 16-bit immediate (hipcc merges the two pairs of a row into one
  // ds_read2_b64, whose 8-bit offsets do not reach the plane slots: a vector add per row and step - and vector instructions are
  // what this kernel pays for); LDS reads issue in the MFMAs' shadow.  W2R_WAIT_RAW before the first use (form (ii) of
  // cdna_hip_programming.md 5.7: the destinations pass through the wait statement).
#define W2R_DSREAD(DST, ADDR, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define W2R_DSREAD1(DST, ADDR, OFF) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define W2R_LOAD_RAW(XA, XB, STEP)                                                              \
  {                                                                                             \
    constexpr int so_ = (((STEP) & 1) * 6 + ((STEP) >> 1)) * (PL / 2) * 8;                      \
    if constexpr (!ODD) {                                                                       \
      W2R_DSREAD(rwa[0][0], (XA)[0], so_); W2R_DSREAD(rwa[0][1], (XA)[0], so_ + 8);             \
      W2R_DSREAD(rwb[0][0], (XB)[0], so_); W2R_DSREAD(rwb[0][1], (XB)[0], so_ + 8);             \
      W2R_DSREAD(rwa[1][0], (XA)[1], so_); W2R_DSREAD(rwa[1][1], (XA)[1], so_ + 8);             \
      W2R_DSREAD(rwb[1][0], (XB)[1], so_); W2R_DSREAD(rwb[1][1], (XB)[1], so_ + 8);             \
    } else {                                                                                    \
      _Pragma("unroll") for (int qh_ = 0; qh_ < 2; ++qh_) {                                     \
        W2R_DSREAD1(rfa[qh_][0], (XA)[qh_], so_); W2R_DSREAD1(rfa[qh_][1], (XA)[qh_], so_ + 4); \
        W2R_DSREAD1(rfa[qh_][2], (XA)[qh_], so_ + 8); W2R_DSREAD1(rfa[qh_][3], (XA)[qh_], so_ + 12); \
        W2R_DSREAD1(rfb[qh_][0], (XB)[qh_], so_); W2R_DSREAD1(rfb[qh_][1], (XB)[qh_], so_ + 4); \
        W2R_DSREAD1(rfb[qh_][2], (XB)[qh_], so_ + 8); W2R_DSREAD1(rfb[qh_][3], (XB)[qh_], so_ + 12); \
      }                                                                                         \
    }                                                                                           \
  }
#define W2R_WAIT_RAW()                                                                          \
  {                                                                                             \
    if constexpr (!ODD)                                                                         \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rwa[0][0]), "+v"(rwa[0][1]), "+v"(rwa[1][0]), "+v"(rwa[1][1]),  \
                   "+v"(rwb[0][0]), "+v"(rwb[0][1]), "+v"(rwb[1][0]), "+v"(rwb[1][1]));         \
    else                                                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rfa[0][0]), "+v"(rfa[0][1]), "+v"(rfa[0][2]), "+v"(rfa[0][3]),  \
                   "+v"(rfa[1][0]), "+v"(rfa[1][1]), "+v"(rfa[1][2]), "+v"(rfa[1][3]),          \
                   "+v"(rfb[0][0]), "+v"(rfb[0][1]), "+v"(rfb[0][2]), "+v"(rfb[0][3]),          \
                   "+v"(rfb[1][0]), "+v"(rfb[1][1]), "+v"(rfb[1][2]), "+v"(rfb[1][3]));         \
  }
#define W2R_RA(QH, C) (ODD ? rfa[QH][C] : rwa[QH][(C) >> 1][(C) & 1])
#define W2R_RB(QH, C) (ODD ? rfb[QH][C] : rwb[QH][(C) >> 1][(C) & 1])
// vertical pass of row ri, columns 2 P, 2 P + 1 of quad half QH
// (even W: on the 8-byte pairs as they come from LDS - one packed fma per pair: beside the MFMAs a packed instruction costs
// ~5.3 cycles against 2 x 4.5)
#define W2R_VERT(QH, P)                                                                         \
  {                                                                                             \
    if constexpr (!ODD) {                                                                       \
      asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(tp[QH][P]) : "v"(sv2), "v"(rwb[QH][P]), "v"(rwa[QH][P])); \
      tn[QH][2 * (P)] = tp[QH][P][0];                                                           \
      tn[QH][2 * (P) + 1] = tp[QH][P][1];                                                       \
    } else {                                                                                    \
      tn[QH][2 * (P)] = __builtin_fmaf(sv, W2R_RB(QH, 2 * (P)), W2R_RA(QH, 2 * (P)));           \
      tn[QH][2 * (P) + 1] = __builtin_fmaf(sv, W2R_RB(QH, 2 * (P) + 1), W2R_RA(QH, 2 * (P) + 1)); \
    }                                                                                           \
  }
// horizontal pass: points (ri, 0), (ri, 1) (P = 0) or (ri, 2), (ri, 3) (P = 1) with the image-border factors folded in
// (ODD: column 2 of the patch may lie outside the image as well: t2 is scaled by f2 first)
// (even W: the two middle points (t1 + t2, t2 - t1) as ONE packed add on the pairs (t0, t1), (t2, t3): both lanes take t2 from
// the second pair's low half, t1 from the first pair's high half, negated in the high lane)
#define W2R_HORZ(QH, P, OUT, F0, F3, F2)                                                        \
  {                                                                                             \
    if ((P) == 0) {                                                                             \
      if constexpr (ODD) tn[QH][2] *= F2[QH];                                                   \
      OUT[QH][0] = __builtin_fmaf(F0[QH], tn[QH][0], -tn[QH][2]);                               \
      if constexpr (ODD) OUT[QH][1] = tn[QH][1] + tn[QH][2];                                    \
    } else {                                                                                    \
      if constexpr (ODD) OUT[QH][2] = tn[QH][2] - tn[QH][1];                                    \
      else {                                                                                    \
        f32x2a m2_;                                                                             \
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(m2_) : "v"(tp[QH][1]), "v"(tp[QH][0])); \
        OUT[QH][1] = m2_[0];                                                                    \
        OUT[QH][2] = m2_[1];                                                                    \
      }                                                                                         \
      OUT[QH][3] = __builtin_fmaf(-(F3[QH]), tn[QH][3], tn[QH][1]);                             \
    }                                                                                           \
  }
  f32x2a rwa[2][2], rwb[2][2];
  const f32x2a sv2 = {sv, sv};
  f32x2a tp[2][2];                                  // (even W: the vertical pass as pairs (t0, t1), (t2, t3))
  float rfa[2][4], rfb[2][4];                       // (ODD: the rows as four dwords)
  float vA[2][4], vB[2][4], tn[2][4];

  // ---- prologue: first item of the first tile (nothing to overlap with)
  int tile = g;
  StageT st_cur = stage_setup(tile);
  ReadT rd_cur = read_setup(tile);
  {
    const char* ab = reinterpret_cast<const char*>(a.wp) + ((long)((tile % a.gridy) * 2)) * 1024 + (long)ri * afrag;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      w2_load_a(au[s][0], aoff, ab + ((long)(s * 4)) * afrag);
      w2_load_a(au[s][1], aoff, ab + ((long)(s * 4)) * afrag + 1024);
    }
    const char* p0 = plane0(st_cur, 0);
#pragma unroll 1
    for (int pl = 0; pl < 12; ++pl) {
      const int cc = pl / 3, dt = pl - 3 * cc;
      const int tt = st_cur.t + dt - 1;
      const bool pok = cc < a.Cin && tt >= 0 && tt < a.T;
      const unsigned dst = lds0 + (unsigned)((pl * PL + 1) * 4 + wave * 1024);
      if (pok) conv_dma_piece16(p0 + ((long)cc * THWb + (long)dt * HWb), st_cur.voff, dst, st_cur.okm);
      else conv_dma_piece16(zero_ptr, vzero, dst, ~0ull);
    }
    stage_zero(st_cur, 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (TAIL) stage_tail(tile, st_cur.tail, 0, 0);
  asm volatile("s_barrier" ::: "memory");
  {
    const unsigned xa0[2] = {lds0 + (unsigned)rd_cur.ba[0] * 4u, lds0 + (unsigned)rd_cur.ba[1] * 4u};
    const unsigned xb0[2] = {lds0 + (unsigned)rd_cur.bb[0] * 4u, lds0 + (unsigned)rd_cur.bb[1] * 4u};
    W2R_LOAD_RAW(xa0, xb0, 0)
    W2R_WAIT_RAW();
    W2R_VERT(0, 0) W2R_VERT(0, 1) W2R_VERT(1, 0) W2R_VERT(1, 1)
    W2R_HORZ(0, 0, vA, rd_cur.f0, rd_cur.f3, rd_cur.f2) W2R_HORZ(0, 1, vA, rd_cur.f0, rd_cur.f3, rd_cur.f2)
    W2R_HORZ(1, 0, vA, rd_cur.f0, rd_cur.f3, rd_cur.f2) W2R_HORZ(1, 1, vA, rd_cur.f0, rd_cur.f3, rd_cur.f2)
  }

  int bufsel = 0;
  const long scstride = 4 * THWb;          // bytes between the first planes of two sub-chunks
  const long ascstride = 24 * afrag;       // bytes between the U fragments of two sub-chunks
#pragma unroll 1
  for (int k = 0; k < nmy; ++k) {
    const bool last_tile = k + 1 == nmy;
    const int ntile = last_tile ? tile : tile + S;
    const StageT st_nxt = last_tile ? st_cur : stage_setup(ntile);
    const ReadT rd_nxt = last_tile ? rd_cur : read_setup(ntile);
    cmp_setup(tile);
    const char* abase_cur = reinterpret_cast<const char*>(a.wp) + ((long)((tile % a.gridy) * 2)) * 1024 + (long)ri * afrag;
    const char* abase_nxt = reinterpret_cast<const char*>(a.wp) + ((long)((ntile % a.gridy) * 2)) * 1024 + (long)ri * afrag;
    float f0[2] = {rd_cur.f0[0], rd_cur.f0[1]}, f3[2] = {rd_cur.f3[0], rd_cur.f3[1]}, f2[2] = {rd_cur.f2[0], rd_cur.f2[1]};
    int last_cb = 0;

    f32x16 acc[16];    // [(m-tile * 2 + quad half) * 4 + j]
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;

#pragma unroll 1
    for (int sc = 0; sc < nsc; ++sc) {
      const bool wrap = sc + 1 == nsc;
      const int nsc_i = wrap ? 0 : sc + 1;
      const int cb = bufsel, nb = bufsel == 2 ? 0 : bufsel + 1;
      bufsel = nb;
      last_cb = cb;
      const int nch0 = nsc_i * 4;
      const int n_t = wrap ? st_nxt.t : st_cur.t;
      const char* anext = (wrap ? abase_nxt : abase_cur) + (long)nsc_i * ascstride;
      const unsigned ldsn = lds0 + (unsigned)((nb * BUFF + 1) * 4 + wave * 1024);
      float nf0[2], nf3[2], nf2[2];
      nf2[0] = wrap ? rd_nxt.f2[0] : f2[0]; nf2[1] = wrap ? rd_nxt.f2[1] : f2[1];
      nf0[0] = wrap ? rd_nxt.f0[0] : f0[0]; nf0[1] = wrap ? rd_nxt.f0[1] : f0[1];
      nf3[0] = wrap ? rd_nxt.f3[0] : f3[0]; nf3[1] = wrap ? rd_nxt.f3[1] : f3[1];
      // this item's row addresses (buffer cb) and the next item's (buffer nb; the next tile's geometry at a tile wrap)
      const unsigned cbb = lds0 + (unsigned)(cb * BUFF) * 4u, nbb = lds0 + (unsigned)(nb * BUFF) * 4u;
      const unsigned xa[2] = {cbb + (unsigned)rd_cur.ba[0] * 4u, cbb + (unsigned)rd_cur.ba[1] * 4u};
      const unsigned xb[2] = {cbb + (unsigned)rd_cur.bb[0] * 4u, cbb + (unsigned)rd_cur.bb[1] * 4u};
      const unsigned xan[2] = {nbb + (unsigned)(wrap ? rd_nxt.ba[0] : rd_cur.ba[0]) * 4u, nbb + (unsigned)(wrap ? rd_nxt.ba[1] : rd_cur.ba[1]) * 4u};
      const unsigned xbn[2] = {nbb + (unsigned)(wrap ? rd_nxt.bb[0] : rd_cur.bb[0]) * 4u, nbb + (unsigned)(wrap ? rd_nxt.bb[1] : rd_cur.bb[1]) * 4u};
      // plane pl = cc * 3 + dt of the next item into buffer nb: through a scalar base + this lane's tile-constant byte offset
      // under the tile's EXEC mask (no vector instruction at all: the fp32 MFMA shares the vector ALU); a plane outside the clip /
      // past Cin reads g_zero_plane with the same offsets and mask - ONE straight-line piece per plane, the source a scalar
      // select (with a branch to a second piece that read the zero word the stage-9 launch ran 3 % longer)
      const char* pl_base = wrap ? plane0(st_nxt, 0) : plane0(st_cur, 0) + (long)nsc_i * scstride;
      const unsigned n_voff = wrap ? st_nxt.voff : st_cur.voff;
      const unsigned long long n_okm = wrap ? st_nxt.okm : st_cur.okm;
#define W2R_STAGE_ADDR(PLN) {}
#define W2R_STAGE(PLN)                                                                                            \
  {                                                                                                               \
    const int cc_ = (PLN) / 3, dt_ = (PLN) - 3 * cc_;                                                             \
    const bool pok_ = nch0 + cc_ < a.Cin && n_t + dt_ - 1 >= 0 && n_t + dt_ - 1 < a.T;                            \
    const char* src_ = pok_ ? pl_base + ((long)cc_ * THWb + (long)dt_ * HWb) : zero_plane;                        \
    conv_dma_piece16_all(src_, n_voff, ldsn + (unsigned)((PLN) * PL * 4), n_okm);                                 \
  }
      // ---- wait counts (vector-memory ops return in order).  Per item this wave issues, in program order: step s < 3: four
      // DMA pieces (slots 1, 5, 9, 13), then the two U loads of ring slot s for the NEXT item (slots 14, 15); steps 3-5: the
      // two U loads only.  The U loads of slot s are used six steps later; issued after them by then: 5 x 2 U loads and
      // every DMA piece of one item except the four of step s itself -> N = 10 + 8 = 18 for s < 3, 10 + 12 = 22 for s >= 3.
      // Extra younger ops the count does not know (the epilogue's loads and stores) only make a wait stricter.
      // Before step 5 the next buffer must be complete: younger than the last piece (step 2, slot 13) are the U loads of
      // steps 2, 3, 4 -> vmcnt(6), then the barrier.  The kernel ends with vmcnt(0).
      // ---- one step = 16 slots of one MFMA each (K = (m-tile * 2 + quad half) * 4 + j) plus: slot 0 the LDS reads of the next
      // step's rows, 4 its transform (vertical + horizontal pass: one group of vector instructions), 1 / 5 / 9 / 13 a DMA piece
      // (steps 0-2), 14-15 the U loads.  Steps alternate between the operand sets vA / vB (no copies); step 5 prepares step 0 of
      // the NEXT item (buffer nb, the next tile's geometry at a tile wrap).
#define W2R_SLOT(STEP, K, VC, VN)                                                                                  \
  {                                                                                                                \
    acc[K] = __builtin_amdgcn_mfma_f32_32x32x2f32(au[STEP][(K) >> 3][(K) & 3], VC[((K) >> 2) & 1][(K) & 3], acc[K], 0, 0, 0); \
    if ((K) == 0 && (STEP) < 5) W2R_LOAD_RAW(xa, xb, (STEP) + 1)                                                   \
    if ((K) == 0 && (STEP) == 5) W2R_LOAD_RAW(xan, xbn, 0)                                                         \
    W2R_XF_SLOTS(STEP, K, VN)                                                                                      \
    if (((K) & 3) == 0 && (STEP) < 3) W2R_STAGE_ADDR((STEP) * 4 + ((K) >> 2))                                      \
    if (((K) & 3) == 1 && (STEP) < 3) W2R_ABL_STAGE((STEP) * 4 + ((K) >> 2))                                       \
    if ((K) >= 14) W2R_ABL_LOADA(au[STEP][(K) & 1], aoff, anext + (long)((STEP) * 4) * afrag + ((K) & 1) * 1024);  \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }
// the transform of the next step's patches as ONE group in slot 4 (the LDS reads of slot 0 have had four MFMAs to land): beside
// fp32 MFMAs a group of vector instructions costs ~10 cycles + ~4.5 per instruction (tools/mfma_fillers.hip: 4 per gap +29
// cycles, 8: +45, 16: +82; a lone one +14) - spread over slots 4-11 the step paid the fixed part eight times: stage 9 -3.1 %
#define W2R_XF_SLOTS(STEP, K, VN)                                                                                  \
    if ((K) == 4) {                                                                                                \
      W2R_WAIT_RAW();                                                                                              \
      W2R_VERT(0, 0) W2R_VERT(0, 1) W2R_VERT(1, 0) W2R_VERT(1, 1)                                                  \
      if ((STEP) < 5) { W2R_HORZ(0, 0, VN, f0, f3, f2) W2R_HORZ(0, 1, VN, f0, f3, f2) W2R_HORZ(1, 0, VN, f0, f3, f2) W2R_HORZ(1, 1, VN, f0, f3, f2) } \
      else { W2R_HORZ(0, 0, VN, nf0, nf3, nf2) W2R_HORZ(0, 1, VN, nf0, nf3, nf2) W2R_HORZ(1, 0, VN, nf0, nf3, nf2) W2R_HORZ(1, 1, VN, nf0, nf3, nf2) } \
    }
#define W2R_STEP(STEP, NWAIT, VC, VN)                                                                              \
  {                                                                                                                \
    W2R_WAIT_A(NWAIT, STEP);                                                                                       \
    W2R_SLOT(STEP, 0, VC, VN) W2R_SLOT(STEP, 1, VC, VN) W2R_SLOT(STEP, 2, VC, VN) W2R_SLOT(STEP, 3, VC, VN)        \
    W2R_SLOT(STEP, 4, VC, VN) W2R_SLOT(STEP, 5, VC, VN) W2R_SLOT(STEP, 6, VC, VN) W2R_SLOT(STEP, 7, VC, VN)        \
    W2R_SLOT(STEP, 8, VC, VN) W2R_SLOT(STEP, 9, VC, VN) W2R_SLOT(STEP, 10, VC, VN) W2R_SLOT(STEP, 11, VC, VN)      \
    W2R_SLOT(STEP, 12, VC, VN) W2R_SLOT(STEP, 13, VC, VN) W2R_SLOT(STEP, 14, VC, VN) W2R_SLOT(STEP, 15, VC, VN)    \
  }
      stage_zero(wrap ? st_nxt : st_cur, nb);
      W2R_STEP(0, 18, vA, vB)
      W2R_STEP(1, 18, vB, vA)
      W2R_STEP(2, 18, vA, vB)
      W2R_STEP(3, 22, vB, vA)
      W2R_STEP(4, 22, vA, vB)
#ifndef HPVG_ABL2_NOBAR
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");     // (lgkmcnt: the zero words of stage_zero)
      if constexpr (TAIL) stage_tail(wrap ? ntile : tile, wrap ? st_nxt.tail : st_cur.tail, nsc_i, nb);
      asm volatile("s_barrier" ::: "memory");   // buffer nb is complete for every wave (and everybody is past the item before)
#endif
      W2R_STEP(5, 22, vB, vA)
#undef W2R_STEP
#undef W2R_XF_SLOTS
#undef W2R_SLOT
#undef W2R_STAGE
#undef W2R_STAGE_ADDR
    }

    // ---- tile complete.  Every wave holds ROW ri of the 4 x 4 points of all four blocks; the output transform needs the four
    // rows of a block in one place.  Horizontal half here: s[b] = sum_j M[ri][j] A[j][b] = (m0 + m1 + m2, m1 - m2 - m3); the
    // three blocks owned by other waves go through LDS: nobody reads buffer last_cb any more (every wave is past the barrier in
    // front of step 5, whose LDS reads are the next buffer's) nor the third buffer (the item before), and no DMA targets
    // either before the next item starts.  Region (sender v, destination block d != v) = 8 x 64 lanes x 16 bytes, six regions
    // per buffer; write k of a region holds (s0, s1) of the accumulator elements 2k and 2k + 1.
#ifdef HPVG_ABL2_NOEPI   // (development ablation, timing only: no exchange, no output transform, no stores)
    if (a.nsc == -12345) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) a.y[(q * 16 + e) * 256 + tid] = acc[q][e];
    }
    if (false)
#endif
    {
      const int third = bufsel == 2 ? 0 : bufsel + 1;       // (bufsel = the next item's buffer)
      // ---- what the epilogue of the owned block (m-tile mw, quad half nw) loads is issued first: its latency passes behind
      // the exchange (one wave per SIMD: nothing else would hide it)
      const int mt = c_yb * 2 + mw;
      const bool mt_ok = mt * 32 < a.Cout;
      // (opaque per tile: hipcc otherwise hoists the 16 channels' 64-bit output offsets out of the tile loop as invariants,
      // spills them around the K loop and reloads them from scratch here - ~30 dependent reloads per tile: the bit-mask variants
      // lost 4-7 % to it)
      int half_e = half;
      asm volatile("" : "+v"(half_e));
      const long HW = (long)HWp;
      const bool r0ok = c_vq && c_h < a.H, r1ok = c_vq && c_h + 1 < a.H;
      const long sp0 = (long)c_t * HW + (long)c_h * W + c_w;
      const long wi0 = ((long)c_b * a.mbreal + mt) * a.T * HW + sp0;      // mask word of (h, w); (h, w+1): + 1; row h+1: + W
      unsigned wrd[4] = {0u, 0u, 0u, 0u};
      unsigned mwd[4] = {0u, 0u, 0u, 0u};
      float bias_r[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half_e;
        bias_r[e] = a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.f;
      }
      if constexpr (VAR == VAR_MASK) {
        const long wmax = (long)a.B * a.T * HW * a.mbreal - 1;      // clamp: lanes without a valid position read a valid word
        const long w0 = mt_ok ? wi0 : 0;
        // (position-fastest words: a lane's two columns are one aligned 8-byte load - w0 is even - and the lanes' loads contiguous)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (!ODD) {
          const u32x2 m01 = *reinterpret_cast<const u32x2*>(a.mask_bits + (r0ok ? w0 : 0));
          const u32x2 m23 = *reinterpret_cast<const u32x2*>(a.mask_bits + (r1ok ? (w0 + W + 1 <= wmax ? w0 + W : 0) : 0));
          mwd[0] = m01[0]; mwd[1] = m01[1]; mwd[2] = m23[0]; mwd[3] = m23[1];
        } else {   // (odd row pitch: the pairs are not 8-byte aligned, the last quad column has one column only)
          const bool c1 = c_w + 1 < W;
          mwd[0] = a.mask_bits[r0ok ? w0 : 0];
          mwd[1] = a.mask_bits[r0ok && c1 ? w0 + 1 : 0];
          mwd[2] = a.mask_bits[r1ok ? w0 + W : 0];
          mwd[3] = a.mask_bits[r1ok && c1 ? w0 + W + 1 : 0];
        }
      }
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int bmt = blk & 1, bqh = blk >> 1;
        const int a0 = (bmt * 2 + bqh) * 4;
        if (blk != wave) {
          const int ridx = wave * 3 + (blk > wave ? blk - 1 : blk);
          float* rg = xs + (ridx < 6 ? last_cb : third) * BUFF + (ridx < 6 ? ridx : ridx - 6) * 2048 + lane * 4;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            f32x4 v;
            v[0] = (acc[a0 + 0][2 * kk] + acc[a0 + 1][2 * kk]) + acc[a0 + 2][2 * kk];
            v[1] = (acc[a0 + 1][2 * kk] - acc[a0 + 2][2 * kk]) - acc[a0 + 3][2 * kk];
            v[2] = (acc[a0 + 0][2 * kk + 1] + acc[a0 + 1][2 * kk + 1]) + acc[a0 + 2][2 * kk + 1];
            v[3] = (acc[a0 + 1][2 * kk + 1] - acc[a0 + 2][2 * kk + 1]) - acc[a0 + 3][2 * kk + 1];
            *reinterpret_cast<f32x4*>(rg + kk * 256) = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // vertical half in row order (fixed: reproducible), started from the bias: Y[0][b] = S0 + S1 + S2, Y[1][b] = S1 - S2 - S3.
      // The wave's own row comes straight from its accumulators (block index = wave = row index on that branch).
      float y0[16][2], y1[16][2];
#pragma unroll
      for (int e = 0; e < 16; ++e) { y0[e][0] = bias_r[e]; y0[e][1] = bias_r[e]; y1[e][0] = bias_r[e]; y1[e][1] = bias_r[e]; }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (v == wave) {
          const int a0 = ((v & 1) * 2 + (v >> 1)) * 4;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float s0 = (acc[a0 + 0][e] + acc[a0 + 1][e]) + acc[a0 + 2][e];
            const float s1 = (acc[a0 + 1][e] - acc[a0 + 2][e]) - acc[a0 + 3][e];
            if (v < 3) { y0[e][0] += s0; y0[e][1] += s1; }
            if (v == 1) { y1[e][0] += s0; y1[e][1] += s1; }
            if (v >= 2) { y1[e][0] -= s0; y1[e][1] -= s1; }
          }
        } else {
          const int ridx = v * 3 + (wave > v ? wave - 1 : wave);
          const float* rg = xs + (ridx < 6 ? last_cb : third) * BUFF + (ridx < 6 ? ridx : ridx - 6) * 2048 + lane * 4;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(rg + kk * 256);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int e = 2 * kk + h2;
              const float s0 = q4[2 * h2], s1 = q4[2 * h2 + 1];
              if (v < 3) { y0[e][0] += s0; y0[e][1] += s1; }
              if (v == 1) { y1[e][0] += s0; y1[e][1] += s1; }
              if (v >= 2) { y1[e][0] -= s0; y1[e][1] -= s1; }
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the regions may be overwritten (the next item's DMA)

      // ---- epilogue of the owned block: masks, 8-byte stores of the two output rows
      const bool lrelu = a.out_lrelu != 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int shb = (e & 3) + 8 * (e >> 2) + 4 * half_e;
        const int co = mt * 32 + shb;
        float yv[4];
        yv[0] = y0[e][0];
        yv[1] = y0[e][1];
        yv[2] = y1[e][0];
        yv[3] = y1[e][1];
        if constexpr (VAR == VAR_BITS) {
#pragma unroll
          for (int p = 0; p < 4; ++p) wrd[p] |= (yv[p] > 0.f ? 1u : 0u) << shb;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) yv[p] = (lrelu && yv[p] < 0.f) ? HPVG_LRELU_SLOPE * yv[p] : yv[p];
        const long oi = ((long)c_b * a.Cout + co) * a.T * HW + sp0;
        const bool cok = co < a.Cout;
        if constexpr (VAR == VAR_MASK) {
#pragma unroll
          for (int p = 0; p < 4; ++p) yv[p] *= ((mwd[p] >> shb) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
        }
#ifdef HPVG_ABL2_NOSTORE   // (development ablation, timing only: the epilogue without its output stores)
        if (a.nsc == -12345)
#endif
        if (!ODD || c_w + 1 < W) {
          if (cok && r0ok) {
            f32x2u4 o2;
            o2[0] = yv[0]; o2[1] = yv[1];
            *reinterpret_cast<f32x2u4*>(a.y + oi) = o2;
          }
          if (cok && r1ok) {
            f32x2u4 o2;
            o2[0] = yv[2]; o2[1] = yv[3];
            *reinterpret_cast<f32x2u4*>(a.y + oi + W) = o2;
          }
        } else {   // (odd W, last quad column: one output column)
          if (cok && r0ok) a.y[oi] = yv[0];
          if (cok && r1ok) a.y[oi + W] = yv[2];
        }
      }
      if constexpr (VAR == VAR_BITS) {
#pragma unroll
        for (int p = 0; p < 4; ++p) wrd[p] |= (unsigned)__shfl_xor((int)wrd[p], 32, 64);   // the other half-wave's channels
        if (half == 0 && mt_ok) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          if constexpr (!ODD) {
            if (r0ok) { u32x2 v2; v2[0] = wrd[0]; v2[1] = wrd[1]; *reinterpret_cast<u32x2*>(a.bits_out + wi0) = v2; }
            if (r1ok) { u32x2 v2; v2[0] = wrd[2]; v2[1] = wrd[3]; *reinterpret_cast<u32x2*>(a.bits_out + wi0 + W) = v2; }
          } else {
            const bool c1 = c_w + 1 < W;
            if (r0ok) { a.bits_out[wi0] = wrd[0]; if (c1) a.bits_out[wi0 + 1] = wrd[1]; }
            if (r1ok) { a.bits_out[wi0 + W] = wrd[2]; if (c1) a.bits_out[wi0 + W + 1] = wrd[3]; }
          }
        }
      }
    }
    st_cur = st_nxt;
    rd_cur = rd_nxt;
    tile = ntile;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef W2R_HORZ
#undef W2R_VERT
#undef W2R_LOAD_RAW
#undef W2R_DSREAD
#undef W2R_DSREAD1
#undef W2R_WAIT_RAW
#undef W2R_RA
#undef W2R_RB
#undef W2R_WAIT_A
