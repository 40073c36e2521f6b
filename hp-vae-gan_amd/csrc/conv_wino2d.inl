// Winograd F(2x2, 3x3) over (H, W) for the wide 3x3x3 convolutions at large sizes - included by conv_mfma.hip inside its
// anonymous namespace.  Same reference call sites as conv_mfma_kernel / conv_wino_kernel (nn.Conv3d of ConvBlock3D(SN),
// modules/networks_3d.py:48-70, and its backward-data pass).
//
// For a 2 x 2 block of outputs ("quad") of one plane and one time tap dt, the 3 x 3 in-plane taps g over the 4 x 4 input
// patch d (rows h-1..h+2, columns w-1..w+2) become 16 products  M = U (.) V,  U = G g G^T,  V = B^T d B,  Y = A^T M A  with
// the F(2,3) matrices of conv_wino.inl applied along both axes: 16 multiplies for 4 outputs instead of 36, i.e. 4/9 of the
// direct kernel's matrix-core work (2/3 of the one-axis kernel's).  M is summed over (channel, dt) before the output
// transform.  fp32 throughout.
//
// Sixteen points x 16 registers = 256 accumulator registers per wave for ONE 32-quad block, so this kernel runs ONE
// workgroup per CU (256 AGPRs + 256 VGPRs per lane) and hides its own memory latency instead of relying on a co-resident
// workgroup:
//  * tile = 64 output channels x 64 consecutive quads of the plane's row-major quad index (2 rows x 128 columns when a
//    row is a multiple of 64 quads; otherwise a tile may run over into the next quad row); waves 0/2 hold channel tile 0,
//    waves 1/3 tile 1; waves 0-1 the first 32 quads, waves 2-3 the second;
//  * the K loop runs over ITEMS = (tile, 4-channel sub-chunk): 6 steps (dt, channel pair) of 16 MFMAs.  The input planes
//    of item i+1 (12 planes, rows as they lie in memory: one 16-byte LDS-DMA piece of 256 lanes per plane, source address
//    per lane = inside the plane or a zero word - no masks, no zero-fill pass, a constant number of pieces) land in the
//    second LDS buffer while item i computes; the U fragments (one 16-byte load per lane = the four points of one row for
//    one channel pair) run in a ring of six steps, i.e. are fetched a whole item (~3 us) ahead;
//  * every vector-memory instruction of the loop is inline asm and the waits are counted by hand (both queues return in
//    order): the compiler would otherwise drain the LDS-DMA queue in front of every use of a loaded U fragment
//    (cdna_hip_programming.md, "Pipelining across barriers").  Count rules are next to WAIT_A below.
// Eligibility (conv_use_wino2d): 3x3x3, Cin >= 8, Cout > 32, even W (a 16-byte group never straddles the START of a plane;
// the one that straddles its end when H*W = 2 (mod 4) is patched in by stage_tail), a staged span of at most 256 groups (W <= 298; <= ~210 when tiles wrap over quad rows), enough
// tiles to fill the chip several times.

struct Wino2Args {
  const float* x;
  const float* wp;        // U fragments: [sub-chunk][dt][cp][row i][m-tile][lane][col j]
  const float* bias;
  const unsigned* mask_bits;
  unsigned* bits_out;
  float* y;
  int B, Cin, Cout, T, H, W;
  int Cq, R, ntq, tqw;    // quads per row, quad rows, tiles per plane (64 consecutive quads each), unused
  int mbtot, gridy, nsc, ntl, PL, out_lrelu;
  int mbreal;             // m-tiles of the layer (the 1-bit mask words' layout); mbtot is rounded up to an even count
};

constexpr int W2_PL = 1024 + 8;   // floats per plane slot: 256 lanes x 16 bytes + the one-float shift (+ pad)

// U = G g G^T of one (o, c, dt): the value at (row i, col j)
__device__ __forceinline__ float wino2_u(const float (&gk)[3][3], int i, int j) {
  float r[3];   // row transform: r[b] = sum_a G[i][a] g[a][b]
#pragma unroll
  for (int b = 0; b < 3; ++b)
    r[b] = i == 0 ? gk[0][b] : (i == 1 ? 0.5f * ((gk[0][b] + gk[2][b]) + gk[1][b]) : (i == 2 ? 0.5f * ((gk[0][b] + gk[2][b]) - gk[1][b]) : gk[2][b]));
  return j == 0 ? r[0] : (j == 1 ? 0.5f * ((r[0] + r[2]) + r[1]) : (j == 2 ? 0.5f * ((r[0] + r[2]) - r[1]) : r[2]));
}

// idx runs over the 2-D fragment section of a weight pack: [sc][dt][cp2][i][mblock][lane][j]
// (mbtot here = m-tiles rounded up to an even count: a workgroup always reads a pair)
__device__ __forceinline__ float wino2_pack_value(const float* __restrict__ w, long idx, int Cin_k, int Cout_k, int nsc, int mbtot,
                                                  int transpose_flip) {
  long r = idx;
  const int j = r % 4; r /= 4;
  const int lane = r % 64; r /= 64;
  const int mb = r % mbtot; r /= mbtot;
  const int i = r % 4; r /= 4;
  const int cp = r % 2; r /= 2;
  const int dt = r % 3; r /= 3;
  const int sc = (int)r;
  if (sc >= nsc) return 0.f;
  const int o = mb * 32 + (lane & 31);
  const int c = sc * 4 + 2 * cp + (lane >> 5);
  if (o >= Cout_k || c >= Cin_k) return 0.f;
  float gk[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int tap = dt * 9 + a * 3 + b;
      gk[a][b] = !transpose_flip ? w[((long)o * Cin_k + c) * 27 + tap] : w[((long)c * Cout_k + o) * 27 + (26 - tap)];
    }
  return wino2_u(gk, i, j);
}
inline int wino2_mb(int Cout) { return 2 * hpvg_cdiv(hpvg_cdiv(Cout, 32), 2); }
inline size_t wino2_pack_floats(int Cin, int Cout) {
  return (size_t)hpvg_cdiv(Cin, 4) * 3 * 2 * 4 * wino2_mb(Cout) * 64 * 4;
}

// one 16-byte LDS-DMA piece, all lanes active: LDS destination m0 + 16 * lane, per-lane 64-bit source address
__device__ __forceinline__ void w2_dma16(const char* src, unsigned lds_addr) {
  unsigned keep_m0;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep_m0)
      : "v"(src), "s"(lds_addr)
      : "memory");
}
// one U fragment (16 bytes per lane): scalar base + 32-bit lane offset; the result is NOT ready when the statement ends -
// every use goes through W2_WAIT_A first
__device__ __forceinline__ void w2_load_a(f32x4& dst, unsigned voff, const char* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
#define W2_WAIT_A(N, S)                                                                               \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(av[S][0]), "+v"(av[S][1]), "+v"(av[S][2]), "+v"(av[S][3]) : : "memory")

// development ablations (timing only, results wrong): -DHPVG_ABL2_NOSTAGE no DMA pieces inside the loop, -DHPVG_ABL2_NOA no U
// loads inside the loop, -DHPVG_ABL2_NOBAR no wait + barrier at the end of an item
#ifdef HPVG_ABL2_NOSTAGE
#define W2_ABL_STAGE(P) {}
#else
#define W2_ABL_STAGE(P) W2_STAGE(P)
#endif
#ifdef HPVG_ABL2_NOA
#define W2_ABL_LOADA(D, O, B) {}
#else
#define W2_ABL_LOADA(D, O, B) w2_load_a(D, O, B)
#endif
// TAIL: the plane size is 2 (mod 4) and the group cut by the plane end needs stage_tail - a separate instance, so that the
// usual one carries none of that code (present but never executed it cost the bit-mask variant 13 % through register
// allocation)
template <int VAR, bool TAIL>
__global__ __launch_bounds__(256, 1) void conv_wino2d_kernel(const Wino2Args a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int mw = wave & 1, nw = wave >> 1;
  constexpr int PL = W2_PL;
  constexpr int BUFF = 12 * PL;
  const int W = a.W, HWp = a.H * a.W;
  const long HWb = (long)HWp * 4;
  const int S = gridDim.x;
  const int g = hpvg_xcd_remap(blockIdx.x, S);
  if (g >= a.ntl) return;
  const int nmy = (a.ntl - 1 - g) / S + 1;          // this workgroup's tiles: g, g + S, ...
  const int nsc = a.nsc;
  const long nitems = (long)nmy * nsc;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)xs;
  const char* zero_ptr = reinterpret_cast<const char*>(g_zero_word);

  // U-fragment stream of a tile: [sc][step = dt*2 + cp][row i] fragments of mbtot*64 lanes x 16 bytes; this lane's bytes
  const long afrag = (long)a.mbtot * 64 * 16;        // bytes between two fragments
  const unsigned aoff = (unsigned)(lane * 16);
  f32x4 av[6][4];                                    // ring: slot = step of the sub-chunk

  // a tile = 64 consecutive quads of a plane's row-major quad index (it may run over the end of a quad row into the next:
  // every lane holds a quad even when a row is not a multiple of 64 quads wide); tile order: channel group, time, tile of
  // the plane, sample
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
  // staging state of a tile: sample / plane of the tile, this lane's 16-byte group of the staged span
  struct StageT { int b, t; unsigned voff; bool ok; int tail; };
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
    // H*W = 2 (mod 4): the group that holds the plane's last two elements also holds two of the next plane (or of nothing:
    // the end of the tensor), so it is zero-sourced like every group outside; the tile that stages it patches the two
    // elements in afterwards (stage_tail)
    const int gt = (HWp - lo4) >> 2;
    q.tail = TAIL ? __builtin_amdgcn_readfirstlane(gt < ng ? gt : -1) : -1;   // uniform: keep it scalar
    return q;
  };
  // the two elements stage_setup's `tail` group lost, for the 12 planes of sub-chunk sc of tile q in buffer bf: an ordinary
  // in-bounds 8-byte load and an LDS write by the lane that owns the group, after the buffer's DMA pieces have landed
  // (a later piece would overwrite them with zeros) and before the barrier that publishes it.  Rare (one tile per plane)
  // and compiler-visible: its waits are the compiler's
  auto stage_tail = [&](const StageT& q, int sc, int bf) __attribute__((always_inline)) {
    if (q.tail < 0) return;
    if (tid == q.tail) {
      const char* p0 = reinterpret_cast<const char*>(a.x) + (((long)q.b * a.Cin + (long)sc * 4) * a.T + (q.t - 1)) * HWb;
#pragma unroll 1
      for (int pl = 0; pl < 12; ++pl) {
        const int cc = pl / 3, dt = pl - 3 * cc;
        const int tt = q.t + dt - 1;
        if (sc * 4 + cc < a.Cin && tt >= 0 && tt < a.T) {
          const float* src = reinterpret_cast<const float*>(p0 + ((long)cc * a.T + dt) * HWb) + (HWp - 2);
          float* dst = xs + bf * BUFF + pl * PL + 1 + 4 * q.tail;
          dst[0] = src[0];
          dst[1] = src[1];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  };

  // compute state of a tile: where this lane's quad is (epilogue) and how it reads its patches (K loop)
  struct ReadT { int base; float f0, f3; };
  int c_yb = 0, c_b = 0, c_t = 0, c_h = 0, c_w = 0;
  bool c_vq = false;
  auto read_setup = [&](int tile) __attribute__((always_inline)) -> ReadT {
    ReadT q;
    int b, t, tp, yb;
    decode(tile, b, t, tp, yb);
    int Q = tp * 64 + nw * 32 + l31;
    if (Q > nq - 1) Q = nq - 1;                                  // lanes past the plane's last quad read (and discard) its patch
    const int Rq = Q / a.Cq, w = 2 * (Q - Rq * a.Cq);
    q.base = half * 3 * PL + 1 + ((2 * Rq - 1) * W + w - 1 - span_lo4(tp));   // even; + (2 cp * 3 + dt) * PL + r * W + c
    q.f0 = w == 0 ? 0.f : 1.f;
    q.f3 = w + 2 >= W ? 0.f : 1.f;
    return q;
  };
  auto cmp_setup = [&](int tile) __attribute__((always_inline)) {
    int tp;
    decode(tile, c_b, c_t, tp, c_yb);
    const int Q = tp * 64 + nw * 32 + l31;
    c_vq = Q < nq;
    const int Rq = Q / a.Cq;
    c_h = 2 * Rq;
    c_w = 2 * (Q - Rq * a.Cq);
  };
  // first plane (channel 4 sc, time t - 1) of a sub-chunk of a staged tile; the 12 planes follow at (cc * T + dt) * HWb
  const long THWb = (long)a.T * HWb;
  auto plane0 = [&](const StageT& q, int sc) __attribute__((always_inline)) -> const char* {
    return reinterpret_cast<const char*>(a.x) + (((long)q.b * a.Cin + (long)sc * 4) * a.T + (q.t - 1)) * HWb;
  };

  // ---- three LDS buffers: item i computes from buffer i % 3 while item i + 1 lands in the next one; the barrier that
  // publishes it sits BEFORE the last step of item i, whose slots read and transform the first patch of item i + 1, so an
  // item starts with its first MFMA (the third buffer keeps the DMA of item i + 2 off a buffer some wave may still read).
#define W2_LOAD_RAW(XL, STEP)                                                                   \
  {                                                                                             \
    const int so_ = (((STEP) & 1) * 6 + ((STEP) >> 1)) * (PL / 2);                              \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                          \
      raw[r_][0] = (XL)[so_ + r_ * (W / 2)];                                                    \
      raw[r_][1] = (XL)[so_ + r_ * (W / 2) + 1];                                                \
    }                                                                                           \
  }
// input transform V = B^T d B of the patch in `raw`: the row pass of column C, then the column pass of row I (with the image
// border factors folded in)
#define W2_ROWPASS(C)                                                                           \
  {                                                                                             \
    const float d0_ = raw[0][(C) >> 1][(C) & 1], d1_ = raw[1][(C) >> 1][(C) & 1];               \
    const float d2_ = raw[2][(C) >> 1][(C) & 1], d3_ = raw[3][(C) >> 1][(C) & 1];               \
    tn[0][C] = d0_ - d2_; tn[1][C] = d1_ + d2_; tn[2][C] = d2_ - d1_; tn[3][C] = d1_ - d3_;     \
  }
#define W2_COLPASS(I, OUT, F0, F3)                                                              \
  {                                                                                             \
    OUT[(I) * 4 + 0] = __builtin_fmaf(F0, tn[I][0], -tn[I][2]);                                 \
    OUT[(I) * 4 + 1] = tn[I][1] + tn[I][2];                                                     \
    OUT[(I) * 4 + 2] = tn[I][2] - tn[I][1];                                                     \
    OUT[(I) * 4 + 3] = __builtin_fmaf(-(F3), tn[I][3], tn[I][1]);                               \
  }
  f32x2a raw[4][2];
  float vc[16], vn[16], tn[4][4];

  // ---- prologue: first item of the first tile (nothing to overlap with)
  int tile = g;
  StageT st_cur = stage_setup(tile);
  ReadT rd_cur = read_setup(tile);
  {
    const char* ab = reinterpret_cast<const char*>(a.wp) + ((long)((tile % a.gridy) * 2 + mw)) * 1024;
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) w2_load_a(av[s][i], aoff, ab + ((long)(s * 4 + i)) * afrag);
    const char* p0 = plane0(st_cur, 0);
#pragma unroll 1
    for (int pl = 0; pl < 12; ++pl) {
      const int cc = pl / 3, dt = pl - 3 * cc;
      const int tt = st_cur.t + dt - 1;
      const bool pok = cc < a.Cin && tt >= 0 && tt < a.T;
      const char* src = (pok && st_cur.ok) ? p0 + ((long)cc * THWb + (long)dt * HWb) + st_cur.voff : zero_ptr;
      w2_dma16(src, lds0 + (unsigned)((pl * PL + 1) * 4 + wave * 1024));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (TAIL) stage_tail(st_cur, 0, 0);
  asm volatile("s_barrier" ::: "memory");
  {
    const f32x2a* xl0 = reinterpret_cast<const f32x2a*>(xs + rd_cur.base);
    W2_LOAD_RAW(xl0, 0)
    W2_ROWPASS(0) W2_ROWPASS(1) W2_ROWPASS(2) W2_ROWPASS(3)
    W2_COLPASS(0, vc, rd_cur.f0, rd_cur.f3) W2_COLPASS(1, vc, rd_cur.f0, rd_cur.f3)
    W2_COLPASS(2, vc, rd_cur.f0, rd_cur.f3) W2_COLPASS(3, vc, rd_cur.f0, rd_cur.f3)
  }

  int bufsel = 0;
  const long scstride = 4 * THWb;          // bytes between the first planes of two sub-chunks
  const long ascstride = 24 * afrag;       // bytes between the U fragments of two sub-chunks
#pragma unroll 1
  for (int k = 0; k < nmy; ++k) {
    const bool last_tile = k + 1 == nmy;
    const int ntile = last_tile ? tile : tile + S;
    // what the LAST sub-chunk of this tile stages and prefetches: sub-chunk 0 of the next tile (the workgroup's last tile
    // takes its own sub-chunk 0 again - never used, but the instruction stream, and with it every wait count, is the same
    // for all items)
    const StageT st_nxt = last_tile ? st_cur : stage_setup(ntile);
    const ReadT rd_nxt = last_tile ? rd_cur : read_setup(ntile);
    cmp_setup(tile);
    const char* abase_cur = reinterpret_cast<const char*>(a.wp) + ((long)((tile % a.gridy) * 2 + mw)) * 1024;
    const char* abase_nxt = reinterpret_cast<const char*>(a.wp) + ((long)((ntile % a.gridy) * 2 + mw)) * 1024;
    const char* lane0_cur = plane0(st_cur, 0) + st_cur.voff;
    const char* lane0_nxt = plane0(st_nxt, 0) + st_nxt.voff;
    const float f0 = rd_cur.f0, f3 = rd_cur.f3;

    f32x16 acc[16];
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
      // the item after this one: its planes (validity per channel / time tap, scalar; this lane's address in a valid plane)
      const int nch0 = nsc_i * 4;
      const char* lane_real = wrap ? lane0_nxt : lane0_cur + (long)nsc_i * scstride;
      const bool n_ok = wrap ? st_nxt.ok : st_cur.ok;
      const int n_t = wrap ? st_nxt.t : st_cur.t;
      const char* anext = (wrap ? abase_nxt : abase_cur) + (long)nsc_i * ascstride;
      const unsigned ldsn = lds0 + (unsigned)((nb * BUFF + 1) * 4 + wave * 1024);
      const float nf0 = wrap ? rd_nxt.f0 : f0, nf3 = wrap ? rd_nxt.f3 : f3;
      const f32x2a* xl = reinterpret_cast<const f32x2a*>(xs + cb * BUFF + rd_cur.base);   // 8-byte units (base, PL, W are even)
      const f32x2a* xln = reinterpret_cast<const f32x2a*>(xs + nb * BUFF + (wrap ? rd_nxt.base : rd_cur.base));
      // plane pl = cc * 3 + dt of the next item into buffer nb: the source address one slot ahead of the piece itself (an
      // LDS-DMA instruction occupies the issue port for ~60 cycles: its slot holds nothing else that can be moved)
      const char* stg_src = zero_ptr;
#define W2_STAGE_ADDR(PLN)                                                                                        \
  {                                                                                                               \
    const int cc_ = (PLN) / 3, dt_ = (PLN) - 3 * cc_;                                                             \
    const bool pok_ = nch0 + cc_ < a.Cin && n_t + dt_ - 1 >= 0 && n_t + dt_ - 1 < a.T;                            \
    stg_src = (pok_ && n_ok) ? lane_real + ((long)cc_ * THWb + (long)dt_ * HWb) : zero_ptr;                       \
  }
#define W2_STAGE(PLN) w2_dma16(stg_src, ldsn + (unsigned)((PLN) * PL * 4));
      // ---- wait counts (vector-memory ops return in order; a wait for "at most N outstanding" retires everything but the N
      // youngest).  Per item this wave issues, in program order: step s < 3: four DMA pieces (slots 1, 5, 9, 13), then the
      // four U loads of ring slot s for the NEXT item (slots 12-15; the piece of slot 13 follows the first of them: counted
      // as if it preceded all four - a wait that is one op too strict at worst); steps 3-5: the four U loads only.  The U loads
      // of slot s are used six steps later; issued after them by then: 5 x 4 U loads and every DMA piece of one item except the
      // four of step s itself -> N = 20 + 12 - 4 = 28 for s < 3, 20 + 12 = 32 for s >= 3.  Extra younger ops the count does not
      // know (the epilogue's stores, the compiler's own loads) only make a wait stricter, never too weak.
      // Before step 5 the next buffer must be complete: younger than the last piece (step 2, slot 13) are the U loads of slots
      // 14, 15 of step 2 and the eight of steps 3, 4 -> vmcnt(10), then the barrier.
      // The kernel ends with vmcnt(0): the last item's unused pieces must have landed before the LDS is given back.
      // ---- one step = 16 slots: ONE MFMA and a small unit of other work that fits its 64-cycle shadow (one wave per SIMD:
      // nothing else hides them): slot 0 the LDS reads of the next step's patch, 4-7 its row pass, 8-11 its column pass,
      // 12-15 the U loads of this ring slot for the next item, 1 / 5 / 9 / 13 a DMA piece (steps 0-2).  Step 5 does that for
      // step 0 of the NEXT item (other buffer, the next tile's border factors at a tile wrap).
#define W2_SLOT(STEP, K)                                                                                           \
  {                                                                                                                \
    acc[K] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[STEP][(K) >> 2][(K) & 3], vc[K], acc[K], 0, 0, 0);            \
    if ((K) == 0 && (STEP) < 5) W2_LOAD_RAW(xl, (STEP) + 1)                                                        \
    if ((K) == 0 && (STEP) == 5) W2_LOAD_RAW(xln, 0)                                                               \
    if ((K) >= 4 && (K) < 8) W2_ROWPASS((K) & 3)                                                                   \
    if ((K) >= 8 && (K) < 12 && (STEP) < 5) W2_COLPASS((K) & 3, vn, f0, f3)                                        \
    if ((K) >= 8 && (K) < 12 && (STEP) == 5) W2_COLPASS((K) & 3, vn, nf0, nf3)                                     \
    if (((K) & 3) == 0 && (STEP) < 3) W2_STAGE_ADDR((STEP) * 4 + ((K) >> 2))                                       \
    if (((K) & 3) == 1 && (STEP) < 3) W2_ABL_STAGE((STEP) * 4 + ((K) >> 2))                                        \
    if ((K) >= 12) W2_ABL_LOADA(av[STEP][(K) & 3], aoff, anext + (long)((STEP) * 4 + ((K) & 3)) * afrag);          \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }
#define W2_STEP(STEP, NWAIT)                                                                                       \
  {                                                                                                                \
    W2_WAIT_A(NWAIT, STEP);                                                                                        \
    W2_SLOT(STEP, 0) W2_SLOT(STEP, 1) W2_SLOT(STEP, 2) W2_SLOT(STEP, 3)                                            \
    W2_SLOT(STEP, 4) W2_SLOT(STEP, 5) W2_SLOT(STEP, 6) W2_SLOT(STEP, 7)                                            \
    W2_SLOT(STEP, 8) W2_SLOT(STEP, 9) W2_SLOT(STEP, 10) W2_SLOT(STEP, 11)                                          \
    W2_SLOT(STEP, 12) W2_SLOT(STEP, 13) W2_SLOT(STEP, 14) W2_SLOT(STEP, 15)                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) vc[q_] = vn[q_];                                             \
  }
      W2_STEP(0, 28)
      W2_STEP(1, 28)
      W2_STEP(2, 28)
      W2_STEP(3, 32)
      W2_STEP(4, 32)
#ifndef HPVG_ABL2_NOBAR
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      if constexpr (TAIL) stage_tail(wrap ? st_nxt : st_cur, nsc_i, nb);
      asm volatile("s_barrier" ::: "memory");   // buffer nb is complete for every wave (and everybody is past the item before)
#endif
      W2_STEP(5, 32)
#undef W2_STEP
#undef W2_SLOT
#undef W2_STAGE
#undef W2_STAGE_ADDR
    }

    {
      // ---- tile complete: output transform Y = A^T M A per channel row, epilogue, 8-byte stores of the two output rows
      const int mt = c_yb * 2 + mw;
      const bool mt_ok = mt * 32 < a.Cout;
      const long HW = (long)HWp;
      const bool r0ok = c_vq && c_h < a.H, r1ok = c_vq && c_h + 1 < a.H;
      const long sp0 = (long)c_t * HW + (long)c_h * W + c_w;
      const long wi0 = ((long)c_b * a.mbreal + mt) * a.T * HW + sp0;      // mask word of (h, w); (h, w+1): + 1; row h+1: + W
      unsigned wrd[4] = {0u, 0u, 0u, 0u};
      unsigned mwd[4] = {0u, 0u, 0u, 0u};
      // every load of the epilogue is issued up front (one latency per tile, not one per channel row: nothing else runs on
      // this CU while a wave waits)
      float bias_r[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        bias_r[e] = a.bias ? a.bias[co < a.Cout ? co : a.Cout - 1] : 0.f;
      }
      if constexpr (VAR == VAR_MASK) {
        {
          const long wmax = (long)a.B * a.T * HW * a.mbreal - 1;      // clamp: lanes without a valid position read a valid word
          const long w0 = mt_ok ? wi0 : 0;
          mwd[0] = a.mask_bits[r0ok ? w0 : 0];
          mwd[1] = a.mask_bits[r0ok ? w0 + 1 : 0];
          mwd[2] = a.mask_bits[r1ok ? (w0 + W < wmax ? w0 + W : wmax) : 0];
          mwd[3] = a.mask_bits[r1ok ? (w0 + W + 1 < wmax ? w0 + W + 1 : wmax) : 0];
        }
      }
      const bool lrelu = a.out_lrelu != 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int shb = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int co = mt * 32 + shb;
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = acc[0 + j][e], m1 = acc[4 + j][e], m2 = acc[8 + j][e], m3 = acc[12 + j][e];
          s0[j] = (m0 + m1) + m2;
          s1[j] = (m1 - m2) - m3;
        }
        const float bv = bias_r[e];
        float yv[4];
        yv[0] = ((s0[0] + s0[1]) + s0[2]) + bv;
        yv[1] = ((s0[1] - s0[2]) - s0[3]) + bv;
        yv[2] = ((s1[0] + s1[1]) + s1[2]) + bv;
        yv[3] = ((s1[1] - s1[2]) - s1[3]) + bv;
        if constexpr (VAR == VAR_BITS) {
#pragma unroll
          for (int p = 0; p < 4; ++p) wrd[p] |= (yv[p] > 0.f ? 1u : 0u) << shb;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) yv[p] = (lrelu && yv[p] < 0.f) ? HPVG_LRELU_SLOPE * yv[p] : yv[p];
        const long oi = ((long)c_b * a.Cout + co) * a.T * HW + sp0;
        const bool cok = co < a.Cout;
        if constexpr (VAR == VAR_MASK) {   // (1-bit masks only: launches with an fp32 out-mask stay on the one-axis kernel)
#pragma unroll
          for (int p = 0; p < 4; ++p) yv[p] *= ((mwd[p] >> shb) & 1u) ? 1.f : HPVG_LRELU_SLOPE;
        }
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
      }
      if constexpr (VAR == VAR_BITS) {
#pragma unroll
        for (int p = 0; p < 4; ++p) wrd[p] |= (unsigned)__shfl_xor((int)wrd[p], 32, 64);   // the other half-wave's channels
        if (half == 0 && mt_ok) {
          if (r0ok) { a.bits_out[wi0] = wrd[0]; a.bits_out[wi0 + 1] = wrd[1]; }
          if (r1ok) { a.bits_out[wi0 + W] = wrd[2]; a.bits_out[wi0 + W + 1] = wrd[3]; }
        }
      }
    }
    st_cur = st_nxt;
    rd_cur = rd_nxt;
    tile = ntile;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef W2_COLPASS
#undef W2_ROWPASS
#undef W2_LOAD_RAW
#undef W2_WAIT_A
