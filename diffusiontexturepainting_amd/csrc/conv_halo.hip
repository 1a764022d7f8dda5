// Halo-tiled 3x3 convolution for gfx950 (stride 1, pad 1): the LDS-staged im2col of the stamp path.
//
// gemm_kernel treats a 3x3 conv as a GEMM whose A operand is re-gathered from L2 for every tap: each input
// element crosses L2 -> LDS nine times.  Here a workgroup owns a TH x TW patch of output pixels of one image;
// for every 64-channel block it stages the (TH+2) x (TW+2) input patch (with its zero-padded halo) in LDS ONCE
// and all nine taps read their shifted rows from it, so the A-side LDS fill traffic drops ~6.4x and the only
// per-tap DMA is the 64-wide weight slice.  Weights are packed channel-block-major: k' = (cb*9 + tap)*64 + c.
//
// Same MFMA mapping, LDS swizzle, epilogue and split-K convention as gemm_kernel (see gemm_conv.hip); the
// pipeline is: weights in a 3-deep ring (counted vmcnt, raw s_barrier per tap), halo double-buffered and
// refilled piece by piece during taps 0..7 of the previous channel block.
//
// The k-loop is written for instruction count, not only for MFMA and DMA overlap: measured on the level-0..2 convs of the stamp,
// removing the MFMAs from the first version of this kernel changed its run time by < 3 % -- the waves were bound by VALU / SALU
// issue (~320 instructions per 8-MFMA k-block: per-read address arithmetic with a run-time tap, 64-bit source-pointer selects per
// DMA piece).  Now the nine taps unroll at compile time, every fragment address is a per-lane constant computed before the loop
// (+ one v_xad per activation read), and the DMA pieces are buffer loads whose per-lane 32-bit offsets are loop constants
// (out-of-image lanes are out of range of the descriptor: the DMA writes their zeros), the block advance riding in the scalar offset.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Swizzle key of halo pixel (hy, hx): 16-byte chunk c of its 128-byte LDS row lives at chunk c ^ key.  The rows a ds_read_b128
// lane group touches come from two image rows of the 18-wide halo of an 8x16 tile, shifted by the tap: with key = (hx >> 1) & 7
// all 16 lanes of every group hit distinct slots of the 256-byte bank row for every tap (exhaustive check), whereas the GEMM
// key (row >> 1) & 7 is 2-way conflicted there (+3..8 % on the level-0 convs).  The 8x8 tile keeps the row key: its
// conflict-free key (hy + 4 * (hx >> 1)) & 7 measured 15-20 % SLOWER (more address arithmetic per fragment than it saves).
template <int TW>
__device__ __forceinline__ int halo_key(int hy, int hx) {
  if constexpr (TW == 16) return (hx >> 1) & 7;
  else return ((hy * (TW + 2) + hx) >> 1) & 7;
}

// GN: the input is the RAW pre-GroupNorm tensor (GF_GNAPPLY).  Every workgroup finalises its image's group statistics from the
// partial sums of the statistics pass and builds the per-channel (scale, shift) table in LDS while its first DMA is in flight; each
// time a halo block has landed, every thread normalises (+ SiLU) exactly the 16-byte chunks it DMA'd itself -- in place, before the
// barrier that publishes the block -- so the nine taps read the normalised patch.  Pixels outside the image keep their zeros (the
// padding applies to the normalised tensor).  One launch and one read + write of the whole tensor fewer per GroupNorm -> conv pair.
// NI: images per workgroup tile.  NI = 3 (8 x 8 pixel tiles only): the workgroup owns the SAME TH x TW pixel tile of three consecutive
// images -- the three guidance branches of a batch-1 stamp at UNet levels 2-3, where an image is 16 x 16 or 8 x 8 -- so the weight
// slices, the bulk of the LDS fill at those levels, are staged once for 192 output rows instead of once per image.
template <int TH, int TW, int BN, bool GN = false, int NI = 1>
__global__ __launch_bounds__(256) void conv_halo_kernel(const GemmParams p) {
  static_assert(NI == 1 || (!GN && TH * TW == 64), "image groups: 8 x 8 tiles, no fused GroupNorm");
  constexpr int PT = TH * TW;                       // output pixels per image and workgroup
  constexpr int BM = NI * PT;                       // output rows per workgroup (64, 128 or 192)
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int WR = BN / 32;                       // weight DMA instructions per wave per tap
  constexpr int HP1 = (TH + 2) * (TW + 2);          // halo pixels per image
  constexpr int HP = NI * HP1;                      // halo pixels
  constexpr int HL = (HP + 31) / 32;                // halo DMA instructions per wave per channel block (8 pixels each)
  constexpr int TL = BM / 32;                       // DMA instructions per wave of a dense shortcut block (tile rows)
  constexpr int HROWS = HL * 32;                    // padded halo rows
  constexpr int HBYTES = HROWS * 128, WBYTES = BN * 128;
  constexpr int SLD = BN + 8;
  static_assert(2 * HBYTES + 3 * WBYTES >= BM * SLD * 2, "staging tile must fit");
  static_assert(TL <= HL, "a dense block has fewer rows than the halo");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo = smem;                 // [2][HROWS][128]
  char* const wring = smem + 2 * HBYTES;   // [3][BN][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.Hi, W = p.Wi;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, tiles_n = (p.N + BN - 1) / BN;
  const int nimg = p.M / (H * W) / NI;  // image groups
  const int nwg = nimg * tiles_y * tiles_x * tiles_n;
  int wg, zid = blockIdx.z;
  if (p.flags & GF_XCDSPLIT) dtp_xcd_split(blockIdx.x, nwg, p.splits, wg, zid);  // K-slice zid lives on XCD zid % 8 (common.h)
  else {
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = wg % tiles_n;  // neighbouring workgroups share the input patch
  int rest = wg / tiles_n;
  const int tx = rest % tiles_x; rest /= tiles_x;
  const int ty = rest % tiles_y;
  const int img = (rest / tiles_y) * NI;  // first image of this workgroup's group
  const int y0 = ty * TH, x0 = tx * TW, n0 = tile_n * BN;
  // k-block sequence: 9 per 64-channel block (channel-block-major), then Cin2/64 dense "shortcut" blocks that read the ResBlock
  // input A2 at the output pixel itself (the fused 1x1 conv).  Split-K cuts this sequence at multiples of 9 k-blocks
  // (dtp_halo_split, common.h), so a slice is [whole channel blocks][dense blocks] and the nine taps unroll at compile time.
  const int ncb = p.Cin >> 6, nmain = ncb * 9, ntail = p.A2 ? (p.Cin2 >> 6) : 0, ntot = nmain + ntail;
  const int ita = zid * p.kb_per_split;
  const int itb = min(ita + p.kb_per_split, ntot);
  const int mb0 = min(ita, nmain) / 9, mb1 = min(itb, nmain) / 9;              // channel blocks [mb0, mb1)
  const int tk0 = max(ita, nmain) - nmain, tk1 = max(itb, nmain) - nmain;      // dense blocks [tk0, tk1)

  // ---- DMA sources: buffer loads (one 32-bit byte offset per lane and piece, the per-block advance in the scalar offset; a lane
  // whose pixel lies outside the image carries offset -1: out of range of the 2 GiB descriptor, and the DMA writes zeros for it).
  // Instruction i of this wave covers LDS rows (i*4 + wave)*8 .. +7; lane -> (row, slot).  halo blocks: row = halo pixel; dense
  // blocks: row = tile row.
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)0x80000000u, 0x00020000);
  const auto rsT = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)0x80000000u, 0x00020000);
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)0x80000000u, 0x00020000);
  const int kc8 = lane & 7;
  int voffA[HL], voffT[HL];
  int hkc[HL];  // GN: channel offset (inside the 64-channel block) of the chunk this thread stages with piece i
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int hp = (i * 4 + wave) * 8 + (lane >> 3);
    const int il = hp / HP1, hq = hp - il * HP1;       // image inside the group, halo pixel inside the image
    const int hy = hq / (TW + 2), hx = hq - hy * (TW + 2);
    const int kc = ((kc8 ^ halo_key<TW>(hy, hx)) << 3);
    hkc[i] = kc;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = (hp < HP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
    voffA[i] = ok ? ((((img + il) * H + iy) * W + ix) * p.lda + kc) * 2 : -1;
    const int il2 = hp / PT, rq = hp - il2 * PT;       // hp reinterpreted as a tile row (dense shortcut blocks: row key)
    const int ty2 = y0 + rq / TW, tx2 = x0 + rq % TW;
    const bool ok2 = p.A2 && (hp < BM) && (ty2 < H) && (tx2 < W);
    voffT[i] = ok2 ? ((((img + il2) * H + ty2) * W + tx2) * p.lda2 + ((kc8 ^ ((hp >> 1) & 7)) << 3)) * 2 : -1;
  }
  const int lrow = wave * 8 + (lane >> 3);
  int voffW[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) voffW[i] = ((n0 + i * 32 + lrow) * p.ldw + ((kc8 ^ ((lrow >> 1) & 7)) << 3)) * 2;

  // piece i of A-block `idx` (kind 1: channel block idx of the input; kind 2: dense block idx of the shortcut input)
  auto a_piece = [&](int buf, int kind, int idx, int i) {
    lds_ptr_t dst = (lds_ptr_t)(halo + buf * HBYTES + (i * 4 + wave) * 8 * 128);
    const int va = voffA[i], vt = voffT[i];  // (by value: the host pass of hipcc rejects an array element as the builtin's argument)
    if (kind == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, va, idx * 128, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, dst, 16, vt, idx * 128, 0, 0);
  };
  auto w_piece_at = [&](uint32_t ring_bytes, int it2, int i) {
    const int vw = voffW[i];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(wring + ring_bytes + (i * 32 + wave * 8) * 128), 16, vw, it2 * 128, 0, 0);
  };
  auto w_piece = [&](int buf, int it2, int i) { w_piece_at(buf * WBYTES, it2, i); };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses, all lane constants: the k-loop adds nothing per read but the k-step XOR and the buffer base.
  // xa[j][tap]: LDS byte address (buffer 0, k-step 0) of this lane's 16-byte chunk of halo row (pixel + tap); the chunk index of
  // k-step ks is (ks*2 + fhalf) ^ key = ((fhalf ^ key) ^ ks*2): bits 5-6 of the address flip with ks.  xt[j]: same for a dense
  // block (row = tile row); xw[ks]: weight rows (the 32-row blocks of a wave share the key: 32 rows = 0 mod 16).
  const int wn0 = (wave & 1) * (BN / 2), wm0 = (wave >> 1) * (BM / 2);
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint32_t h_lds = lds_addr(halo), w_lds = lds_addr(wring);
  uint32_t xa[TM][9], xt[TM], xw[4];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int r = wm0 + j * 32 + frow;
    const int il = r / PT, rq = r - il * PT;
    const int py = rq / TW, px = rq % TW;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int row = il * HP1 + (py + ky) * (TW + 2) + px + kx;
      xa[j][tap] = h_lds + row * 128 + ((fhalf ^ halo_key<TW>(py + ky, px + kx)) << 4);
    }
    xt[j] = h_lds + r * 128 + ((fhalf ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xw[ks] = w_lds + (wn0 + frow) * 128 + (((ks * 2 + fhalf) ^ (((wn0 + frow) >> 1) & 7)) << 4);

  float* const gn_tab = (float*)(smem + 2 * HBYTES + 3 * WBYTES);  // GN: [Cin][2] = (scale, shift) of this image's channels
  // ---- prologue: first A-block, W(ita), W(ita + 1)
  if (itb > ita) {
    const int kind = mb0 < mb1 ? 1 : 2, idx = mb0 < mb1 ? mb0 : tk0;
#pragma unroll
    for (int i = 0; i < HL; ++i) a_piece(0, kind, idx, i);
#pragma unroll
    for (int i = 0; i < WR; ++i) w_piece(0, ita, i);
    if (itb > ita + 1) {
#pragma unroll
      for (int i = 0; i < WR; ++i) w_piece(1, ita + 1, i);
    }
  }
  if constexpr (GN) {
    float* const gst = gn_tab + 2 * p.Cin;  // [32][2] mean, rstd
    const int groups = p.Cin / p.gn_cpg;
    {
      const int g = tid >> 3, j = tid & 7;
      double s = 0.0, q = 0.0;  // (fp64 totals and variance: common.h sum_pairs_strided_d)
      if (g < groups)
        sum_pairs_strided_d(p.gn_part + ((size_t)img * p.gn_nchunk + j) * groups * 2 + g * 2, (size_t)8 * groups * 2, (p.gn_nchunk - j + 7) / 8, s, q);
    s = sum8_d(s); q = sum8_d(q);
      if (g < groups && j == 0) gn_mean_rstd(s, q, 1.0f / ((float)(H * W) * (float)p.gn_cpg), p.gn_eps, gst[2 * g], gst[2 * g + 1]);
    }
    __syncthreads();
    for (int c = tid; c < p.Cin; c += 256) {
      const int g = c / p.gn_cpg;
      const float a = p.gn_gamma[c] * gst[2 * g + 1];
      gn_tab[2 * c] = a;
      gn_tab[2 * c + 1] = p.gn_beta[c] - gst[2 * g] * a;
    }
    __syncthreads();
  }

  // One k-block: 4 k-steps of TM x TN MFMAs.  rd_a(ks, j) / rd_w(ks, i) return the fragments; piece(q) issues DMA instruction q of
  // this k-block (q < NP) -- between the MFMAs, where its issue cost hides under the matrix pipe.
  constexpr int NF = TM + TN, LA = (4 * NF <= 16) ? 4 : 2, NM = TM * TN;
#define DTP_KBLOCK(NP, RD_A, RD_W, PIECE)                                                                   \
  {                                                                                                         \
    constexpr int PPS = ((NP) + 3) / 4;                                                                     \
    static_assert(PPS <= 4, "DMA pieces per k-step");                                                       \
    f16x8 fr[LA][NF];                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < LA; ++ks) {                                                     \
      _Pragma("unroll") for (int j = 0; j < TM; ++j) fr[ks][j] = RD_A(ks, j);                               \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) fr[ks][TM + i] = RD_W(ks, i);                          \
    }                                                                                                       \
    DTP_MMA_STEP(0, RD_A, RD_W, PIECE) DTP_MMA_STEP(1, RD_A, RD_W, PIECE)                                   \
    DTP_MMA_STEP(2, RD_A, RD_W, PIECE) DTP_MMA_STEP(3, RD_A, RD_W, PIECE)                                   \
  }
#define DTP_PIECE_AT(q, PIECE)                                                                              \
  {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    PIECE(q);                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  }
#define DTP_MMA_STEP(ks, RD_A, RD_W, PIECE)                                                                 \
  {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    wait_lds_frags<((ks + LA < 4 ? ks + LA : 4) - ks - 1) * NF, NF>(fr[ks % LA]);                           \
    _Pragma("unroll") for (int i = 0; i < TN; ++i) _Pragma("unroll") for (int j = 0; j < TM; ++j) {         \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ks % LA][TM + i], fr[ks % LA][j], acc[i][j], 0, 0, 0); \
      if constexpr (PPS >= 1) { if (i * TM + j == 0) DTP_PIECE_AT(ks * PPS, PIECE) }                        \
      if constexpr (PPS >= 2) { if (i * TM + j == NM / PPS) DTP_PIECE_AT(ks * PPS + 1, PIECE) }             \
      if constexpr (PPS >= 3) { if (i * TM + j == 2 * NM / PPS) DTP_PIECE_AT(ks * PPS + 2, PIECE) }         \
      if constexpr (PPS >= 4) { if (i * TM + j == 3 * NM / PPS) DTP_PIECE_AT(ks * PPS + 3, PIECE) }         \
    }                                                                                                       \
    if constexpr (ks + LA < 4) {                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      _Pragma("unroll") for (int j = 0; j < TM; ++j) fr[(ks + LA) % LA][j] = RD_A(ks + LA, j);              \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) fr[(ks + LA) % LA][TM + i] = RD_W(ks + LA, i);         \
    }                                                                                                       \
  }

  // The halo of the next A-block is fetched while the current channel block computes: its HL pieces are spread over taps 0..7
  // (HPT per tap, ahead of the tap's weight pieces, so that the wait for W(first tap of the next block) covers them all).
  constexpr int HPT = (HL + 7) / 8;
  int abuf = 0;        // halo buffer of the current A-block
  uint32_t ha = 0;     // abuf * HBYTES
  int it = ita;

  // k-block `tap` of channel block blk.  next_kind: 0 = no A-block follows inside this slice, 1 = channel block, 2 = dense block.
  auto kb_main = [&](auto tapc, int blk, int next_kind, int next_idx) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr int RING = TAP % 3, RNEXT = (TAP + 2) % 3;
    constexpr int H0 = TAP < 8 ? (TAP * HPT < HL ? TAP * HPT : HL) : HL, H1 = TAP < 8 ? ((TAP + 1) * HPT < HL ? (TAP + 1) * HPT : HL) : HL;
    constexpr int NPH = H1 - H0;                                                // halo pieces issued in this tap
    constexpr int P0 = TAP == 0 ? 0 : ((TAP - 1) * HPT < HL ? (TAP - 1) * HPT : HL);
    constexpr int NPREV = TAP == 0 ? 0 : (TAP * HPT < HL ? TAP * HPT : HL) - P0;  // ... and in the previous one
    // W(it) -- and with it every older group: this block's halo -- has landed once only the groups issued during the previous
    // k-block are outstanding: that tap's halo pieces (if an A-block follows) and W(it + 1)
    if (it + 1 >= itb) wait_vmcnt<0>();
    else if (NPREV > 0 && next_kind) wait_vmcnt<WR + NPREV>();
    else wait_vmcnt<WR>();
    if constexpr (GN && TAP == 0) {  // a fresh halo block: this wave's pieces of it have landed (the wait above)
      char* const hb = halo + abuf * HBYTES;
      const bool silu = p.gn_silu != 0;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        if (voffA[i] != -1) {  // pixels outside the image (and rows beyond the patch) stay zero
          f16x8* const q = (f16x8*)(hb + ((i * 4 + wave) * 8) * 128 + lane * 16);
          const f16x8 v = *q;
          const float* tb = gn_tab + 2 * (blk * 64 + hkc[i]);
          const f32x4 t0 = *(const f32x4*)tb, t1 = *(const f32x4*)(tb + 4), t2 = *(const f32x4*)(tb + 8), t3 = *(const f32x4*)(tb + 12);
          const float ab[16] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3], t3[0], t3[1], t3[2], t3[3]};
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)v[e], ab[2 * e], ab[2 * e + 1]);
            if (silu) f = silu_f(f);
            o[e] = (f16)f;
          }
          *q = o;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rewritten chunks are in LDS before the barrier publishes the block
    }
    __builtin_amdgcn_s_barrier();
    const bool more_w = it + 2 < itb;
    auto rd_a = [&](int ks, int j) { return lds_read16((xa[j][TAP] ^ (uint32_t)(ks << 5)) + ha); };
    auto rd_w = [&](int ks, int i) { return lds_read16(xw[ks] + (uint32_t)(RING * WBYTES + i * 4096)); };
    auto piece = [&](int q) {
      if (q < NPH) { if (next_kind) a_piece(abuf ^ 1, next_kind, next_idx, H0 + q); }
      else if (q < NPH + WR) { if (more_w) w_piece(RNEXT, it + 2, q - NPH); }
    };
    DTP_KBLOCK(NPH + WR, rd_a, rd_w, piece)
    ++it;
  };
  for (int blk = mb0; blk < mb1; ++blk) {
    const int next_kind = blk + 1 < mb1 ? 1 : (tk0 < tk1 ? 2 : 0);
    const int next_idx = blk + 1 < mb1 ? blk + 1 : tk0;
    kb_main(std::integral_constant<int, 0>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 1>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 2>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 3>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 4>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 5>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 6>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 7>{}, blk, next_kind, next_idx);
    kb_main(std::integral_constant<int, 8>{}, blk, next_kind, next_idx);
    abuf ^= 1;
    ha = abuf * HBYTES;
  }
  // dense shortcut blocks: one k-block each; the weight ring position is a run-time value here (it restarts at 0: 9 = 0 mod 3)
  {
    uint32_t wr = 0, wn = 2 * WBYTES;  // ring slots of W(it), W(it + 2), in bytes
    for (int tk = tk0; tk < tk1; ++tk) {
      if (it + 1 >= itb) wait_vmcnt<0>();
      else wait_vmcnt<WR>();  // A(tk) was issued ahead of W(it + 1) during the previous k-block
      __builtin_amdgcn_s_barrier();
      const bool more_w = it + 2 < itb, more_a = tk + 1 < tk1;
      auto rd_a = [&](int ks, int j) { return lds_read16((xt[j] ^ (uint32_t)(ks << 5)) + ha); };
      auto rd_w = [&](int ks, int i) { return lds_read16(xw[ks] + wr + (uint32_t)(i * 4096)); };
      auto piece = [&](int q) {
        if (q < TL) { if (more_a) a_piece(abuf ^ 1, 2, tk + 1, q); }
        else if (q < TL + WR) {
          if (more_w) w_piece_at(wn, it + 2, q - TL);
        }
      };
      DTP_KBLOCK(TL + WR, rd_a, rd_w, piece)
      ++it;
      wr = (wr == 2 * WBYTES) ? 0 : wr + WBYTES;
      wn = (wn == 2 * WBYTES) ? 0 : wn + WBYTES;
      abuf ^= 1;
      ha = abuf * HBYTES;
    }
  }
#undef DTP_KBLOCK
#undef DTP_MMA_STEP
#undef DTP_PIECE_AT

  // ---- epilogue.  Tile row r <-> pixel (y0 + r / TW, x0 + r % TW) of image `img`.
  auto row_m = [&](int r, bool& ok) -> size_t {
    const int il = r / PT, rq = r - il * PT;
    const int y = y0 + rq / TW, x = x0 + rq % TW;
    ok = (y < H) && (x < W);
    return ((size_t)(img + il) * H + y) * W + x;
  };
  if (p.splits > 1) {
    float* part = p.part + (size_t)zid * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bool ok;
        const size_t m = row_m(wm0 + j * 32 + frow, ok);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn0 + i * 32 + 8 * q + 4 * fhalf;
          if (ok && n + 4 <= p.N) {
            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *(f32x4*)(part + m * p.N + n) = v;
          }
        }
      }
    return;
  }
  __syncthreads();
  f16* stg = (f16*)smem;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int ml = wm0 + j * 32 + frow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn0 + i * 32 + 8 * q + 4 * fhalf;
        f16x4 v = {(f16)acc[i][j][4 * q], (f16)acc[i][j][4 * q + 1], (f16)acc[i][j][4 * q + 2], (f16)acc[i][j][4 * q + 3]};
        *(f16x4*)(stg + ml * SLD + nl) = v;
      }
    }
  __syncthreads();
  const int fl = p.flags;
  constexpr int NC = BN / 8;
  const int nc = tid % NC, n = n0 + nc * 8;  // loop-invariant (256 % NC == 0): the bias vector is loaded once
  float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((fl & GF_BIAS) && n + 8 <= p.N) {
    const f32x4 t0 = *(const f32x4*)(p.bias + n), t1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = t0[e]; bv[4 + e] = t1[e]; }
  }
  // residual rows three iterations ahead (three-register ring, unconditional loads from clamped pixels): see gemm_conv.hip
  constexpr int EIT = BM * NC / 256;
  const bool pre_r = (fl & GF_RESID) && n + 8 <= p.N;
  auto load_r = [&](int it) {
    const int r = (tid + min(it, EIT - 1) * 256) / NC;
    const int il = r / PT, rq = r - il * PT;
    const int y = min(y0 + rq / TW, H - 1), x = min(x0 + rq % TW, W - 1);
    return *(const f16x8*)(p.R + (((size_t)(img + il) * H + y) * W + x) * p.ldr + n);
  };
  f16x8 r0 = {0, 0, 0, 0, 0, 0, 0, 0}, r1 = r0, r2 = r0;
  if (pre_r) { r0 = load_r(0); r1 = load_r(1); r2 = load_r(2); }
  int eit = 0;
  for (int idx = tid; idx < BM * NC; idx += 256, ++eit) {
    const f16x8 rcur = r0;
    r0 = r1; r1 = r2;
    if (pre_r) r2 = load_r(eit + 3);
    const int ml = idx / NC;
    bool ok;
    const size_t m = row_m(ml, ok);
    if (!ok || n + 8 > p.N) continue;  // N % 8 == 0 is required by the launcher
    const f16x8 v = *(const f16x8*)(stg + ml * SLD + nc * 8);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)v[e] + bv[e];
    if (fl & GF_RESID) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += (float)rcur[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)x[e];
    *(f16x8*)((f16*)p.C + m * p.ldc + n) = o;
  }
}

template <int TH, int TW, int BN, int NI = 1>
constexpr int halo_lds() {
  constexpr int HP = NI * (TH + 2) * (TW + 2), HL = (HP + 31) / 32;
  return 2 * HL * 32 * 128 + 3 * BN * 128;
}

template <int BN>
int launch_halo3(const GemmParams& pin, hipStream_t s) {  // 8 x 8 pixel tiles of three consecutive images per workgroup
  GemmParams p = pin;
  const int H = p.Hi, W = p.Wi;
  const int blocks = (p.M / (H * W) / 3) * ((H + 7) / 8) * ((W + 7) / 8) * ((p.N + BN - 1) / BN);
  constexpr int lds = halo_lds<8, 8, BN, 3>();
  static const bool xcd_off = [] { const char* e = getenv("DTP_NO_XCD_SPLIT"); return e && e[0] && e[0] != '0'; }();
  const bool xs = !xcd_off && dtp_xcd_split_ok(blocks, p.splits);
  if (xs) p.flags |= GF_XCDSPLIT;
  const dim3 grid = xs ? dim3(blocks * p.splits, 1, 1) : dim3(blocks, 1, p.splits);
  hipLaunchKernelGGL((conv_halo_kernel<8, 8, BN, false, 3>), grid, dim3(256), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

constexpr int GN_MAX_CIN = 1024;  // (scale, shift) table: 8 KB at most, so that the 8x16 x 64 variant still fits twice on a CU

template <int TH, int TW, int BN>
int launch_halo(const GemmParams& pin, hipStream_t s) {
  GemmParams p = pin;
  const int H = p.Hi, W = p.Wi;
  const int blocks = (p.M / (H * W)) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * ((p.N + BN - 1) / BN);
  constexpr int lds = halo_lds<TH, TW, BN>();
  static const bool xcd_off = [] { const char* e = getenv("DTP_NO_XCD_SPLIT"); return e && e[0] && e[0] != '0'; }();
  const bool xs = !xcd_off && dtp_xcd_split_ok(blocks, p.splits);
  if (xs) p.flags |= GF_XCDSPLIT;
  const dim3 grid = xs ? dim3(blocks * p.splits, 1, 1) : dim3(blocks, 1, p.splits);
#ifdef DTP_EXPERIMENTAL  // (GroupNorm on the staged patch measured +4.9 ms per stamp, DESIGN.md 3.6: an experiment, not in the product library)
  if (p.flags & GF_GNAPPLY)
    hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, true>), grid, dim3(256), lds + p.Cin * 8 + 256, s, p);
  else
#endif
    hipLaunchKernelGGL((conv_halo_kernel<TH, TW, BN, false>), grid, dim3(256), lds, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

}  // namespace

template <int TH, int TW, int BN>
static void set_halo_attr() {
  constexpr int lds = halo_lds<TH, TW, BN>();
  (void)hipFuncSetAttribute((const void*)conv_halo_kernel<TH, TW, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef DTP_EXPERIMENTAL
  (void)hipFuncSetAttribute((const void*)conv_halo_kernel<TH, TW, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + GN_MAX_CIN * 8 + 256);
#endif
}

void dtp_conv_halo_init() {
  (void)hipFuncSetAttribute((const void*)conv_halo_kernel<8, 8, 64, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds<8, 8, 64, 3>());
  (void)hipFuncSetAttribute((const void*)conv_halo_kernel<8, 8, 128, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds<8, 8, 128, 3>());
  set_halo_attr<8, 16, 64>();
  set_halo_attr<8, 16, 128>();
  set_halo_attr<8, 8, 64>();
  set_halo_attr<8, 8, 128>();
}

// variant: 0 = 8x16 x 64, 1 = 8x16 x 128, 2 = 8x8 x 64, 3 = 8x8 x 128.  p.W must be the channel-block-major packing
// ([cb][tap][64] then the fused-shortcut columns); p.kb_per_split / p.splits count 64-wide k-blocks like gemm_kernel.
bool dtp_conv_halo_supported(const GemmParams& p) {
#ifndef DTP_EXPERIMENTAL
  if (p.flags & GF_GNAPPLY) return false;
#endif
  if ((p.flags & GF_GNAPPLY) && (!p.gn_part || !p.gn_gamma || !p.gn_beta || p.Cin > GN_MAX_CIN || p.gn_cpg < 1 || (p.Cin % p.gn_cpg) || p.Cin / p.gn_cpg > 32 ||
                                 p.gn_nchunk < 1))
    return false;
  // 32-bit byte offsets into 2 GiB buffer descriptors
  if ((size_t)p.M * p.lda * 2 >= ((size_t)1 << 31) || (size_t)p.M * p.lda2 * 2 >= ((size_t)1 << 31) || (size_t)(p.N + 128) * p.ldw * 2 >= ((size_t)1 << 31)) return false;
  return (p.flags & GF_CONV3) && !(p.flags & (GF_UPS2 | GF_GEGLU | GF_OUT_F32 | GF_LNFOLD | GF_ROWSTATS | GF_BIAS_M | GF_GELU | GF_QUICKGELU | GF_SILU)) &&
         p.stride == 1 && p.pad == 1 && (!p.A2 || ((p.Cin2 & 63) == 0 && (p.lda2 & 7) == 0)) && (p.Cin & 63) == 0 &&
         (p.N & 7) == 0 && p.Ho == p.Hi && p.Wo == p.Wi &&
         (p.ldc & 7) == 0 && (!(p.flags & GF_RESID) || (p.ldr & 7) == 0) && p.M % (p.Hi * p.Wi) == 0;
}

// variants 4 / 5: three images per workgroup (8 x 8 x 64 / 128): image count a multiple of 3, no fused GroupNorm
bool dtp_conv_halo3_supported(const GemmParams& p) {
  return dtp_conv_halo_supported(p) && !(p.flags & GF_GNAPPLY) && (p.M / (p.Hi * p.Wi)) % 3 == 0;
}

int dtp_launch_conv_halo(const GemmParams& p, int variant, hipStream_t s) {
  if (!dtp_conv_halo_supported(p) || (variant >= 4 && !dtp_conv_halo3_supported(p))) { dtp_set_error("conv_halo: unsupported problem"); return DTP_ERR_ARG; }
  if (p.splits > 1 && (p.kb_per_split % 9)) { dtp_set_error("conv_halo: K slices are whole channel blocks (kb_per_split %d is not a multiple of 9)", p.kb_per_split); return DTP_ERR_ARG; }
  int rc;
  switch (variant) {
    case 4: rc = launch_halo3<64>(p, s); break;
    case 5: rc = launch_halo3<128>(p, s); break;
    case 0: rc = launch_halo<8, 16, 64>(p, s); break;
    case 1: rc = launch_halo<8, 16, 128>(p, s); break;
    case 2: rc = launch_halo<8, 8, 64>(p, s); break;
    case 3: rc = launch_halo<8, 8, 128>(p, s); break;
    default: dtp_set_error("conv_halo: bad variant %d", variant); return DTP_ERR_ARG;
  }
  if (rc != DTP_OK) { dtp_set_error("conv_halo launch failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
  if (p.splits > 1 && !(p.flags & GF_NOREDUCE)) return dtp_launch_splitk_reduce(p, s);
  return DTP_OK;
}
