// Weight-streaming 3x3 convolution for the small feature maps of the stamp path (UNet levels 2-3: 16 x 16 and 8 x 8 pixels per
// image, M = 768 / 192 output rows at batch 1, Cin / Cout = 1280 ... 2560).
//
// At these levels a conv is 30-60 MB of weights against 0.5-2 MB of activations: the tiled kernels (gemm_kernel, conv_halo_kernel)
// stage every weight byte through LDS behind a barrier per k-block and need 6-12 K-slices with fp32 slabs to fill the chip; they
// reach 1.1-1.5 TB/s of weight stream at M = 192 and 600-790 TFLOP/s at M = 768.  Here
//   * the weights are packed in MFMA fragment order (pack_conv_ws_kernel): fragment (n-tile of 32 output channels, 64-channel block,
//     channel quarter w, tap) is 1 KB = lane l -> 16 bytes, and a wave loads it with ONE fully coalesced buffer_load_dwordx4 straight
//     into the A-operand registers -- no LDS, no barrier, no address arithmetic on the weight side; every weight byte is fetched by
//     exactly one wave of one workgroup per pixel group, nine fragments ahead of its use (a register ring the compiler's vmcnt
//     bookkeeping tracks);
//   * a workgroup owns ALL output pixels of its pixel group (three 8 x 8 images = 192 rows, or one 16 x 16 image = 256 rows) for one
//     n-tile and a range of channel blocks; the input patch of a channel block (with its zero halo) is staged once by LDS-DMA
//     (three buffers, two channel blocks ahead) and the nine taps read shifted rows of it.  The patch row pitch is ODD (TW + 3 pixel slots) and the
//     16-byte chunk c of pixel slot (hy, hx) lives at chunk c ^ (hx & 7): every ds_read_b128 lane group is bank-conflict-free for
//     every tap (exhaustive check in tests/test_conv_ws_layout.py), and a fragment address is ONE per-lane register per tap column
//     plus a compile-time immediate -- 3 address registers for the whole kernel;
//   * the four waves split the CONTRACTION, not the tile: wave w multiplies channels [16 w, 16 w + 16) of every channel block, all
//     nine taps, into its own full set of accumulators; the four partial tiles are summed through LDS once, after the loop (fixed
//     order: deterministic).  So a K-slice of the launch is a whole workgroup-lifetime of 4 x more contraction per slab than a tiled
//     kernel's: 2 slabs fill the chip at M = 768 (3 images x 40 n-tiles x 2), 5 at M = 192 -- the consumer (reduce + GroupNorm)
//     reads 2-5 fp32 slabs instead of 6-12.
// MFMA operand order as everywhere in this code base: acc[n][m] = W-fragment (A operand) x pixel-fragment (B operand); lane =
// (pixel = lane & 31, half = lane >> 5), register r <-> channel 8 (r / 4) + 4 half + r % 4.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ f16x8 lds_read16_off(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// CO: the configuration for TWO co-resident workgroups per CU (<= 80 KB of LDS, <= 256 registers): the four partial tiles are combined
// one n-tile at a time (the combine area is then no larger than the three patch buffers) and the patch fragments use two register
// sets instead of three (no request across the channel-block boundary: the co-resident workgroup's waves cover that latency).
template <int TH, int TW, int NI, int NT, bool CO = false>
struct WsGeom {
  static constexpr int PW = TW + 3, PH = TH + 2;        // patch pitch (odd) and rows, in pixel slots
  static constexpr int HP1 = PH * PW, HP = NI * HP1;    // slots per image / per pixel group
  static constexpr int HL = (HP + 31) / 32;             // LDS-DMA pieces (8 slots each) per wave and channel block
  static constexpr int PBYTES = HL * 32 * 128;          // one patch buffer
  static constexpr int TM = NI * TH * TW / 32;          // 32-pixel MFMA tiles of the group
  static constexpr int RPT = 32 / TW, TPI = TH / RPT;   // image rows per tile, tiles per image
  static constexpr int CBLK = 1152;                     // pitch of a 1 KB (wave, tile, register group) block of the combine area
  static constexpr int CMB = 4 * (CO ? 1 : NT) * TM * 4 * CBLK;  // the four waves' partial tiles (fp32) of one combine round
  static constexpr int MAINB = (3 * PBYTES > CMB ? 3 * PBYTES : CMB);   // patches, later the combine area
  static constexpr int STATB = (NI == 1 && NT == 2) ? (4 * NT * 32 * 2 + NT * 32 * 2) * 8 : 0;  // GF_GNSTATS: per-wave and per-workgroup channel sums (fp64)
  static constexpr int LDS = MAINB + STATB;
  static constexpr int LT = (HL - 1) / (NT * TM);       // the tap in which the last patch piece of a channel block is issued
  static constexpr int WAITB = (17 - LT) * NT;          // vmcnt at the per-block barrier (see the main loop)
  static_assert(TW == 8 || TW == 16, "pixel tile width");
  static_assert(TH % RPT == 0 && (NI * TH * TW) % 32 == 0, "whole MFMA tiles");
  static_assert(HP * 128 < 65536, "fragment offsets are 16-bit immediates");
  static_assert(LT < 8 && WAITB < 64, "patch pieces fit in the taps before the barrier; vmcnt is a 6-bit counter");
  static_assert(LDS <= (CO ? 80 : 160) * 1024, "LDS");
};

// Weight packing.  Fragment index f = ((nt * ncb + cb) * 4 + w) * 9 + tap; element e of lane l of fragment f is
// W[n = nt * 32 + (l & 31)][ch = cb * 64 + 16 w + 8 (l >> 5) + e][tap] (zero for n >= Cout).
__global__ void pack_conv_ws_kernel(const float* __restrict__ w, f16* __restrict__ out, int Cout, int Cin, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
    long long f = i >> 9;
    const int tap = (int)(f % 9); f /= 9;
    const int wq = (int)(f & 3); f >>= 2;
    const int ncb = Cin >> 6;
    const int cb = (int)(f % ncb), nt = (int)(f / ncb);
    const int n = nt * 32 + (l & 31), ch = cb * 64 + 16 * wq + 8 * (l >> 5) + e;
    out[i] = n < Cout ? (f16)w[((size_t)n * Cin + ch) * 9 + tap] : (f16)0.f;
  }
}

// The fused 1x1 shortcut (GemmParams::A2, Cin2 channels): fragment index g = (nt * ntb + tb) * 4 + w behind the 3x3 fragments; element e
// of lane l is W1[n = nt * 32 + (l & 31)][ch = tb * 64 + 16 w + 8 (l >> 5) + e].
__global__ void pack_conv_ws_tail_kernel(const float* __restrict__ w, f16* __restrict__ out, int Cout, int Cin2, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
    long long g = i >> 9;
    const int wq = (int)(g & 3); g >>= 2;
    const int ntb = Cin2 >> 6;
    const int tb = (int)(g % ntb), nt = (int)(g / ntb);
    const int n = nt * 32 + (l & 31), ch = tb * 64 + 16 * wq + 8 * (l >> 5) + e;
    out[i] = n < Cout ? (f16)w[(size_t)n * Cin2 + ch] : (f16)0.f;
  }
}

// fragment of tile J for tap row KY: the lane's column term + a compile-time row offset (tile J = image J / TPI, rows RPT * (J % TPI) ..)
template <class G, int KY, int J>
__device__ __forceinline__ void rd_frags(f16x8 (&dst)[G::TM], uint32_t ct) {
  if constexpr (J < G::TM) {
    dst[J] = lds_read16_off<((J / G::TPI) * G::HP1 + (G::RPT * (J % G::TPI) + KY) * G::PW) * 128>(ct);
    rd_frags<G, KY, J + 1>(dst, ct);
  }
}

template <int TH, int TW, int NI, int NT, bool NTW, bool CO = false>
__global__ __launch_bounds__(256, CO ? 2 : 1) void convws_kernel(const GemmParams p) {
  using G = WsGeom<TH, TW, NI, NT, CO>;
  constexpr int TM = G::TM, HL = G::HL, PW = G::PW, PH = G::PH, HP1 = G::HP1, PB = G::PBYTES, NTM = NT * TM;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;

  // ---- which (pixel group, n-range, K-slice).  A pixel group = the TH x TW tile (ty, tx) of NI consecutive images; an n-range = NT
  // n-tiles of 32 output channels; a unit = (n-range, K-slice).  Block b runs on XCD b % 8.  With a multiple of 8 units (the deep
  // levels: 40 n-tiles, megabytes of weights per unit) the pixel groups of one unit sit next to each other on ONE XCD, whose L2
  // then serves all but the first reader of every weight fragment; otherwise (few units, many pixel groups: the wide levels, whose
  // whole weight set fits any L2) the units of one pixel group sit on one XCD and share its patches.
  // H x W: the map the 3x3 window slides over = the output map (stride 1); with GF_UPS2 it is the nearest-2x upsample of the stored
  // Hi x Wi input, which only the patch DMA's source addresses know about
  // (round 5: this decode was nine integer divisions, two of them 64-bit -- ~500 instructions in front of the first memory request of
  // a 20-50 us launch.  Map sizes and tile counts are powers of two everywhere in this model: shifts on that path, one division pair
  // for the block index, none for the K-slices of an unsplit launch; the general forms stay as the fallback.)
  const int H = p.Ho, W = p.Wo;
  const int tiles_x = W / TW, tiles_y = H / TH;  // (TW, TH: compile-time powers of two)
  const int hw = H * W;
  const bool pow2 = ((hw & (hw - 1)) | (tiles_x & (tiles_x - 1)) | (tiles_y & (tiles_y - 1))) == 0;
  const int images = pow2 ? (p.M >> __builtin_ctz(hw)) : p.M / hw;
  const int npg = (images / NI) * tiles_y * tiles_x;
  const int nts = (p.N + 31) >> 5, nrs = (nts + NT - 1) / NT, S = p.splits;
  const int units = nrs * S;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int pg, u;
  if ((units & 7) == 0) { const int q = idx / npg; pg = idx - q * npg; u = q * 8 + xcd; }
  else { const int pgq = (npg + 7) >> 3; const int q = idx / pgq; pg = (idx - q * pgq) * 8 + xcd; u = q; }
  if (u >= units || pg >= npg) return;
  const int ncb = p.Cin >> 6;
  int z = 0, nr = u, mb0 = 0, mb1 = ncb;  // channel blocks [mb0, mb1)
  if (S > 1) {
    nr = u / S; z = u - nr * S;
    mb0 = z * ncb / S; mb1 = (z + 1) * ncb / S;  // (S <= Cin / 64 <= 40: 32-bit)
  }
  int tx, ty, img0;
  if (pow2) {
    const int sx = __builtin_ctz(tiles_x), sy = __builtin_ctz(tiles_y);
    tx = pg & (tiles_x - 1); ty = (pg >> sx) & (tiles_y - 1); img0 = (pg >> (sx + sy)) * NI;
  } else {
    tx = pg % tiles_x; ty = (pg / tiles_x) % tiles_y; img0 = (pg / (tiles_x * tiles_y)) * NI;
  }
  const int y0 = ty * TH, x0 = tx * TW;

  // ---- weight ring first: fragment (n-tile, cb, tap) of this wave's channel quarter; the loads of the first channel block are in
  // flight while the patch addresses are computed.  An n-tile beyond N (odd n-tile count, NT = 2) loads zeros.
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wfr, 0, (int)0x80000000u, 0x00020000);
  constexpr int OOB = (int)0x80000000u;
  int wvoff[NT], wbase[NT];  // fragment index = ((wbase + cb * 4) * 9 + tap)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = nr * NT + i;
    wvoff[i] = nt < nts ? lane * 16 : OOB;
    wbase[i] = (nt < nts ? nt : 0) * ncb * 4 + wave;
  }
  auto wload = [&](int i, int cb, int tap) -> f16x8 {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, wvoff[i], ((wbase[i] + cb * 4) * 9 + tap) * 1024, NTW ? 2 : 0);
    return __builtin_bit_cast(f16x8, v);
  };
  f16x8 wf[NT][9];
  if (mb0 < mb1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i][t] = wload(i, mb0, t);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- patch DMA: piece i of this wave covers pixel slots (i*4 + wave)*8 .. +7, lane -> (slot, 16-byte position kc8); the position
  // holds source chunk kc8 ^ (hx & 7).  Slots outside the image (or the pad column / the buffer's tail) carry offset -1: out of
  // range of the descriptor, the DMA writes zeros.  Slot s = il * HP1 + hy * PW + hx; consecutive pieces are 32 slots apart:
  // (il, hy, hx) advance incrementally (no divisions).
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)0x80000000u, 0x00020000);
  int voffA[HL];
  {
    const int ups = (p.flags & GF_UPS2) ? 1 : 0;
    const int s0 = wave * 8 + (lane >> 3);  // < 32 <= HP1
    int il = 0, hy = s0 / PW, hx = s0 - hy * PW;
    constexpr int DY = 32 / PW, DX = 32 - DY * PW;
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = il < NI && hx < TW + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      voffA[i] = ok ? ((((img0 + il) * p.Hi + (iy >> ups)) * p.Wi + (ix >> ups)) * p.lda + (((lane & 7) ^ (hx & 7)) << 3)) * 2 : -1;
      hx += DX; hy += DY;
      if (hx >= PW) { hx -= PW; ++hy; }
      if (hy >= PH) { hy -= PH; ++il; }
    }
  }
  auto a_piece_at = [&](uint32_t bo, int cb, int i) {
    const int vo = voffA[i];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + bo + (i * 4 + wave) * 1024), 16, vo, cb * 128, 0, 0);
  };
  if (mb0 < mb1) {
#pragma unroll
    for (int i = 0; i < HL; ++i) a_piece_at(0u, mb0, i);
    if (mb0 + 1 < mb1) {
#pragma unroll
      for (int i = 0; i < HL; ++i) a_piece_at((uint32_t)PB, mb0 + 1, i);
    }
  }

  // ---- fragment addresses: colterm[kx] (per lane) + a compile-time row offset.  Lane's pixel inside tile j: row frow / TW of the
  // tile's RPT image rows, column frow % TW.
  const int pxl = frow % TW, pyl = frow / TW;
  uint32_t colterm[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
    colterm[kx] = lds_addr(smem) + (pyl * PW + pxl + kx) * 128 + ((((2 * wave + fhalf) ^ ((pxl + kx) & 7))) << 4);

  f32x16 acc[NT][TM];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main loop.  patch(cb) lives in buffer (cb - mb0) % 3; its pieces are issued during the first taps of cb - 2 (one after each
  // MFMA).  One barrier per channel block, at the start of tap 8 and after this wave's last reads of patch(cb) have returned: it
  // publishes patch(cb + 1) to every wave, and it is what makes the buffer of patch(cb) free for the pieces of patch(cb + 3),
  // which are issued after it (first taps of cb + 1).  The fragments of the next tap are always requested one tap ahead, across the
  // channel-block boundary too (three fragment sets: 9 taps = 0 mod 3), so no LDS latency is exposed anywhere.  Every fragment feeds NT
  // MFMAs: with one n-tile per wave the kernel is bound by the LDS read path (1 KB per MFMA; measured: removing the MFMAs changes
  // nothing), with two it is not.
  // vmcnt at the barrier: the ops younger than the last piece of patch(cb + 1) (issued in tap LT of cb - 1) are wload(cb, LT..8) and,
  // during cb, the pieces of patch(cb + 2) (if any) and wload(cb + 1, 0..7): (17 - LT) NT, + HL -- waiting for the smaller count is
  // exact or stricter.
  constexpr int NBF = CO ? 2 : 3;
  f16x8 bf[NBF][TM];
  auto rd = [&](auto tapc, uint32_t bo, f16x8 (&dst)[TM]) {
    constexpr int TAP = decltype(tapc)::value, KY = TAP / 3, KX = TAP % 3;
#ifndef DTP_WS_NO_LDSREAD
    rd_frags<G, KY, 0>(dst, colterm[KX] + bo);
#else
#pragma unroll
    for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(dst[j]));
#endif
  };
  if (mb0 < mb1 && !(p.sm_valid & 1)) {  // (sm_valid: diagnostic bits of $DTP_WS_DEBUG, 0 in production)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    rd(std::integral_constant<int, 0>{}, 0u, bf[0]);
    // MORE: channel block cb + 1 exists; MORE2: cb + 2 exists.  bo / bo1 / bo2: byte offsets of the buffers of patch(cb), (cb+1), (cb+2)
    auto do_cb = [&](auto morec, auto more2c, int cb, uint32_t bo, uint32_t bo1, uint32_t bo2) {
      constexpr bool MORE = decltype(morec)::value, MORE2 = decltype(more2c)::value;
      auto tap = [&](auto tapc) {
        constexpr int TAP = decltype(tapc)::value, CUR = TAP % NBF, NXT = (TAP + 1) % NBF;
        __builtin_amdgcn_sched_barrier(0);
#ifdef DTP_WS_NO_LDSREAD
#define DTP_WS_WAIT(n, f) asm volatile("" ::: "memory")
#else
#define DTP_WS_WAIT(n, f) wait_lds_frags<n, TM>(f)
#endif
        if constexpr (TAP < 8) {
          rd(std::integral_constant<int, TAP + 1>{}, bo, bf[NXT]);
          __builtin_amdgcn_sched_barrier(0);
          DTP_WS_WAIT(TM, bf[CUR]);
        } else {
          DTP_WS_WAIT(0, bf[CUR]);  // this wave's last reads of patch(cb) are complete
          if constexpr (MORE) {
            wait_vmcnt<G::WAITB>();
            __builtin_amdgcn_s_barrier();
            if constexpr (!CO) {
              rd(std::integral_constant<int, 0>{}, bo1, bf[NXT]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int i = 0; i < NT; ++i) {
#ifndef DTP_WS_NO_MFMA  // (diagnostic builds only: tools/ws_variants.sh)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][TAP], bf[CUR][j], acc[i][j], 0, 0, 0);
#else
            asm volatile("" : "+v"(acc[i][j]) : "v"(wf[i][TAP]), "v"(bf[CUR][j]));
#endif
            if constexpr (MORE2) {
              if (TAP * NTM + j * NT + i < HL) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef DTP_WS_NO_DMA
                a_piece_at(bo2, cb + 2, TAP * NTM + j * NT + i);
#endif
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        if constexpr (MORE) {
          __builtin_amdgcn_sched_barrier(0);
#ifndef DTP_WS_NO_WLOAD
#pragma unroll
          for (int i = 0; i < NT; ++i) wf[i][TAP] = wload(i, cb + 1, TAP);
#endif
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (CO && TAP == 8) rd(std::integral_constant<int, 0>{}, bo1, bf[0]);  // two sets: tap 8 has just released set 0
        }
      };
#undef DTP_WS_WAIT
      tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
      tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
      tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
    };
    typedef std::true_type Y;
    typedef std::false_type N_;
    uint32_t bo = 0, bo1 = PB, bo2 = 2 * PB;
    int cb = mb0;
    for (; cb + 2 < mb1; ++cb) {
      do_cb(Y{}, Y{}, cb, bo, bo1, bo2);
      const uint32_t t = bo; bo = bo1; bo1 = bo2; bo2 = t;
    }
    if (cb + 1 < mb1) {
      do_cb(Y{}, N_{}, cb, bo, bo1, bo2);
      ++cb;
      bo = bo1;
    }
    do_cb(N_{}, N_{}, cb, bo, bo1, bo2);
  }

  // ---- fused 1x1 shortcut: dense 64-channel blocks of A2 at the output pixels.  No patch, no LDS, no barrier: wave w takes the
  // WHOLE blocks tk0 + w, tk0 + w + 4, ... of the slice (all four k-steps) and loads its B operands straight from memory -- lane (pixel,
  // half) reads 16 bytes of its pixel's row, and the four k-step loads of one pixel tile hit the same 32 lines back to back, so every
  // line crosses L2 -> L1 once per workgroup (the first version split each block's k-steps over the waves like the 3x3 part: four
  // waves x 32-byte pieces of every line = 4 x the L2 traffic, 13 us for the eight blocks of an M = 192 slice).  Tiles are requested
  // D tiles ahead (a register ring), across block boundaries; blocks beyond the slice carry
  // out-of-range offsets: zeros, which add nothing.  Weight fragments of block tb: tbase + tb * 4 + k-step, behind the 3x3 fragments.
  if (p.A2 && !(p.sm_valid & 1)) {
    const int ntb = p.Cin2 >> 6;
    const int tk0 = (int)((long long)z * ntb / S), tk1 = (int)((long long)(z + 1) * ntb / S);
    if (tk0 < tk1) {
      const auto rsT = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, (int)0x80000000u, 0x00020000);
      int voffT[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int il = j / G::TPI, py = G::RPT * (j % G::TPI) + pyl;
        voffT[j] = ((((img0 + il) * H + y0 + py) * W + x0 + pxl) * p.lda2 + 8 * fhalf) * 2;
      }
      int tbase[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) tbase[i] = nts * ncb * 36 + (nr * NT + i < nts ? nr * NT + i : 0) * ntb * 4;
      constexpr int D = CO ? 2 : (TM % 4 == 0) ? 4 : 3;   // prefetch distance in pixel tiles; divides TM: ring slot = j % D
      static_assert(TM % D == 0, "tile ring");
      f16x8 ring[4 * D];  // 4 k-step fragments per ring slot: [slot * 4 + ks] (the patch fragments' registers are dead by now)
      constexpr int WTB = CO ? 1 : 2;  // weight fragment sets: the co-resident build reloads one set per block (register budget)
      f16x8 wt[WTB][NT][4];
      auto frag_load = [&](int tb, int j, int slot) {   // the four k-step fragments of pixel tile j of block tb
        const int vo = tb < tk1 ? voffT[j] : OOB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsT, vo, tb * 128 + ks * 32, 0);
          ring[slot * 4 + ks] = __builtin_bit_cast(f16x8, v);
        }
      };
      auto wt_load = [&](int tb, int par) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int vo = tb < tk1 ? wvoff[i] : OOB;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW, vo, (tbase[i] + tb * 4 + ks) * 1024, NTW ? 2 : 0);
            wt[par % WTB][i][ks] = __builtin_bit_cast(f16x8, v);
          }
        }
      };
      const int nround = (tk1 - tk0 + 3) >> 2;  // blocks per wave (uniform: the surplus blocks are zeros)
      int tb = tk0 + wave;
      wt_load(tb, 0);
#pragma unroll
      for (int j = 0; j < D; ++j) frag_load(tb, j, j);
      auto block = [&](auto parc, int tbc, bool more) {
        constexpr int PAR = decltype(parc)::value % WTB;
        if (WTB == 2 && more) wt_load(tbc + 4, PAR ^ 1);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < NT; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wt[PAR][i][ks], ring[(j % D) * 4 + ks], acc[i][j], 0, 0, 0);
          if (j + D < TM) frag_load(tbc, j + D, j % D);
          else if (more) frag_load(tbc + 4, j + D - TM, j % D);
        }
        if (WTB == 1 && more) wt_load(tbc + 4, 0);
      };
      for (int r = 0; r < nround; r += 2) {
        block(std::integral_constant<int, 0>{}, tb, r + 1 < nround);
        if (r + 1 < nround) block(std::integral_constant<int, 1>{}, tb + 4, r + 2 < nround);
        tb += 8;
      }
    }
  }

  // ---- the four partial tiles -> LDS, block (wave, tile, register group q) = 64 x 16 bytes at a 1152-byte pitch, slot (2 * pixel + half) ^ ((pixel >> 2) & 1):
  // thread t of pass (i, j) then sums the four waves' values of (pixel t / 8 of tile j, channels 4 (t % 8) .. of n-tile i) -- conflict-
  // free for the lane groups of ds_read_b128 (tests/test_conv_ws_layout.py) -- and eight neighbouring lanes store one pixel's 128
  // contiguous bytes.
  if (p.sm_valid & 2) return;
  __builtin_amdgcn_s_barrier();  // every wave has left the patches
  constexpr int CB = G::CBLK;
  constexpr int RT = CO ? 1 : NT;        // n-tiles per combine round
  const int epx = tid >> 3, c4 = tid & 7;            // pixel inside the tile, channel quad
  const int epxl = epx % TW, epyl = epx / TW;
  // slot of (pixel, half) inside a 1 KB block: (2 pixel + half) ^ ((pixel >> 2) & 1).  The plain 2 pixel + half is conflict-free for the
  // READS below (16-lane groups, 64 banks) but 2-way conflicted for the WRITES (ds_write_b128: groups of 8 consecutive lanes, 32 banks --
  // eight lanes 32 bytes apart cover 128 bytes twice): that was every conflict cycle SQ_LDS_BANK_CONFLICT counted in this kernel
  // (profiles/r05_pmc_unet_mfma.json: 468 k cycles per launch = 480 workgroups x 4 waves x 32 stores x 8 extra cycles; round-5 verdict,
  // item 4).  The XOR keeps both sides conflict-free (tests/test_conv_ws_layout.py).
  const char* const src0 = smem + (c4 >> 1) * CB + ((epx * 2 + (c4 & 1)) ^ ((epx >> 2) & 1)) * 16;
  // GF_GNSTATS (unsplit launches of the two-n-tile builds): the GroupNorm that consumes this conv's output gets its statistics from
  // here -- per (pixel tile, group) sums of the ROUNDED fp16 outputs -- instead of from a pass of its own over the tensor.
  constexpr bool CAN_STATS = G::STATB > 0;
  const bool stats = CAN_STATS && (p.flags & GF_GNSTATS) && p.splits == 1;
  // (round 6: fp64 from the lane butterflies on -- a chunk's partial is the sum of 128 x cpg squares; as a tree of fp32 sums its rounding
  // decided the variance of a group whose mean is hundreds of standard deviations away from zero; one rounding to fp32 at the very end)
  double* const sred = (double*)(smem + G::MAINB);   // [wave][NT * 32 channels][2], then [NT * 32][2] totals
#pragma unroll
  for (int i0 = 0; i0 < NT; i0 += RT) {
    if (i0 > 0) __syncthreads();  // the previous round's reads are done
    {
      char* const mine = smem + wave * RT * TM * 4 * CB + ((frow * 2 + fhalf) ^ ((frow >> 2) & 1)) * 16;
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[i0 + i][j][4 * q], acc[i0 + i][j][4 * q + 1], acc[i0 + i][j][4 * q + 2], acc[i0 + i][j][4 * q + 3]};
            *(f32x4*)(mine + ((i * TM + j) * 4 + q) * CB) = v;
          }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int n = (nr * NT + i0 + i) * 32 + 4 * c4;
      const bool ncol = n + 4 <= p.N;                  // N % 4 == 0 is required by the launcher
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.splits == 1 && (p.flags & GF_BIAS) && ncol) bv = *(const f32x4*)(p.bias + n);
      float ssum[4] = {0.f, 0.f, 0.f, 0.f}, qsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const char* src = src0 + (i * TM + j) * 4 * CB;
        const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + RT * TM * 4 * CB), v2 = *(const f32x4*)(src + 2 * RT * TM * 4 * CB),
                    v3 = *(const f32x4*)(src + 3 * RT * TM * 4 * CB);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((v0[e] + v1[e]) + v2[e]) + v3[e];
        const int il = j / G::TPI, py = G::RPT * (j % G::TPI) + epyl;
        const size_t m = ((size_t)(img0 + il) * H + y0 + py) * W + x0 + epxl;
        if (!ncol) continue;
        if (p.splits > 1) {
          *(f32x4*)(p.part + ((size_t)z * p.M + m) * p.N + n) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
          if (p.flags & GF_RESID) {
            const f16x4 r = *(const f16x4*)(p.R + m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
          }
          const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
          *(f16x4*)((f16*)p.C + m * p.ldc + n) = o;
          if constexpr (CAN_STATS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float f = (float)o[e]; ssum[e] += f; qsum[e] += f * f; }
          }
        }
      }
      if constexpr (CAN_STATS) {
        if (stats) {  // this wave's 8 pixels (lanes 8 apart share a channel quad): fixed-order butterfly, then one writer per quad
          double sd[4], qd[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sd[e] = (double)ssum[e]; qd[e] = (double)qsum[e];
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { sd[e] += shfl_xor_d(sd[e], o); qd[e] += shfl_xor_d(qd[e], o); }
          }
          if (lane < 8) {
            double* d = sred + ((wave * NT + i0 + i) * 32 + 4 * c4) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[2 * e] = sd[e]; d[2 * e + 1] = qd[e]; }
          }
        }
      }
    }
  }
  if constexpr (CAN_STATS) {
    if (stats) {
      __syncthreads();
      double* const tot = sred + 4 * NT * 32 * 2;
      if (tid < NT * 32 * 2) tot[tid] = ((sred[tid] + sred[NT * 64 + tid]) + sred[2 * NT * 64 + tid]) + sred[3 * NT * 64 + tid];
      __syncthreads();
      // partial sums [image][2 * pixel tiles][groups][2]: chunk 2 * tile + (n-range & 1).  A group overlaps one n-range (64 channels >=
      // channels per group) or two NEIGHBOURING ones, which differ in parity: every (chunk, group) slot has exactly one writer -- the
      // n-range that holds a group entirely also writes the zero of the other parity -- so the buffer needs no clearing.
      const int cpg = p.gn_cpg, groups = p.N / cpg;
      const int c_lo = nr * NT * 32, c_hi = min(p.N, c_lo + NT * 32);
      const int g_lo = c_lo / cpg, g_hi = (c_hi - 1) / cpg;
      const int g = g_lo + tid;
      if (g <= g_hi) {
        const int a = max(g * cpg, c_lo), b = min((g + 1) * cpg, c_hi);
        double sg = 0.0, qg = 0.0;
        for (int c = a; c < b; ++c) { sg += tot[(c - c_lo) * 2]; qg += tot[(c - c_lo) * 2 + 1]; }
        const int ntile = tiles_x * tiles_y, pt = ty * tiles_x + tx;
        float* const base = p.st_out + ((size_t)img0 * 2 * ntile + 2 * pt) * groups * 2;
        float* const mine = base + (size_t)(nr & 1) * groups * 2 + g * 2;
        mine[0] = (float)sg; mine[1] = (float)qg;
        if (g * cpg >= c_lo && (g + 1) * cpg <= c_hi) {
          float* const other = base + (size_t)((nr & 1) ^ 1) * groups * 2 + g * 2;
          other[0] = 0.f; other[1] = 0.f;
        }
      }
    }
  }
}

template <int TH, int TW, int NI, int NT, bool CO = false>
int launch_ws(const GemmParams& p, hipStream_t s) {
  using G = WsGeom<TH, TW, NI, NT, CO>;
  const int npg = (p.M / (p.Ho * p.Wo) / NI) * (p.Ho / TH) * (p.Wo / TW);
  const int nrs = (((p.N + 31) >> 5) + NT - 1) / NT;
  const int units = nrs * p.splits;
  const int blocks = (units & 7) == 0 ? units * npg : ((npg + 7) >> 3) * 8 * units;  // the kernel's two block -> (pixel group, unit) maps
  // a fragment that one workgroup reads once is streamed past the caches (nt); with several pixel groups per unit the L2 serves the
  // re-reads, so it keeps the default policy
  static const int dbg = [] { const char* e = getenv("DTP_WS_DEBUG"); return e ? atoi(e) : 0; }();
  GemmParams q = p;
  q.sm_valid = dbg & 3;
  const bool nt = (dbg & 4) ? false : (dbg & 8) ? true : npg == 1;
  if (nt) hipLaunchKernelGGL((convws_kernel<TH, TW, NI, NT, true, CO>), dim3(blocks), dim3(256), G::LDS, s, q);
  else hipLaunchKernelGGL((convws_kernel<TH, TW, NI, NT, false, CO>), dim3(blocks), dim3(256), G::LDS, s, q);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

template <int TH, int TW, int NI, int NT, bool CO = false>
void set_ws_attr() {
  (void)hipFuncSetAttribute((const void*)convws_kernel<TH, TW, NI, NT, true, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, WsGeom<TH, TW, NI, NT, CO>::LDS);
  (void)hipFuncSetAttribute((const void*)convws_kernel<TH, TW, NI, NT, false, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, WsGeom<TH, TW, NI, NT, CO>::LDS);
}

}  // namespace

void dtp_conv_ws_init() {
  set_ws_attr<8, 8, 3, 1>();
  set_ws_attr<16, 16, 1, 1>();
  set_ws_attr<8, 16, 1, 2>();
  set_ws_attr<8, 16, 1, 2, true>();
}

// elements of the packing: the 3x3 fragments, then (Cin2 > 0) the fragments of the fused 1x1 shortcut
size_t dtp_conv_ws_packed_elems(int Cout, int Cin, int Cin2) { return (size_t)((Cout + 31) / 32) * ((size_t)(Cin / 64) * 36 + (size_t)(Cin2 / 64) * 4) * 512; }

// w: [Cout][Cin][3][3]; w1 (may be null): [Cout][Cin2] -- the fused shortcut's 1x1 weights
int dtp_launch_pack_conv_ws(const float* w, const float* w1, f16* out, int Cout, int Cin, int Cin2, hipStream_t s) {
  if ((Cin & 63) || (Cin2 & 63) || (Cin2 > 0) != (w1 != nullptr)) { dtp_set_error("pack_conv_ws: Cin %d / Cin2 %d must be multiples of 64", Cin, Cin2); return DTP_ERR_ARG; }
  const long long total = (long long)dtp_conv_ws_packed_elems(Cout, Cin, 0);
  const int blocks = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_ws_kernel, dim3(blocks), dim3(256), 0, s, w, out, Cout, Cin, total);
  if (Cin2 > 0) {
    const long long t2 = (long long)dtp_conv_ws_packed_elems(Cout, 0, Cin2);
    const int b2 = (int)((t2 + 255) / 256 > 65535 ? 65535 : (t2 + 255) / 256);
    hipLaunchKernelGGL(pack_conv_ws_tail_kernel, dim3(b2), dim3(256), 0, s, w1, out + total, Cout, Cin2, t2);
  }
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

// variant 0: pixel group = three 8 x 8 images (the image IS the tile), one n-tile per workgroup; variant 1: one 16 x 16 image, one
// n-tile; variant 2: an 8 x 16 pixel tile of one image of any size (H % 8 == 0, W % 16 == 0), TWO n-tiles (64 output channels) per
// workgroup; variant 3: variant 2 built for two co-resident workgroups per CU (WsGeom CO).  Stride 1, pad 1 (optionally over the nearest-2x upsample of the input: GF_UPS2), Cin % 64 == 0 (and Cin2 % 64 == 0 with a fused shortcut).  nsplit K-slices (<= Cin / 64), each a
// range of whole channel blocks; nsplit > 1 leaves fp32 slabs.
bool dtp_conv_ws_supported(const GemmParams& p, int variant, int nsplit) {
  if (!p.Wfr || !(p.flags & GF_CONV3) || p.batch > 1) return false;
  if (p.A2 && ((p.Cin2 & 63) || (p.lda2 & 7) || (size_t)p.M * p.lda2 * 2 >= ((size_t)1 << 31))) return false;
  if (p.flags & (GF_GEGLU | GF_OUT_F32 | GF_LNFOLD | GF_ROWSTATS | GF_BIAS_M | GF_GELU | GF_QUICKGELU | GF_SILU | GF_GNAPPLY | GF_SOFTMAX16)) return false;
  if (p.stride != 1 || p.pad != 1 || (p.Cin & 63) || (p.N & 3) || (p.lda & 7) || (p.ldc & 3)) return false;
  if (p.flags & GF_UPS2) { if (p.Ho != 2 * p.Hi || p.Wo != 2 * p.Wi) return false; }   // the window slides over the nearest-2x upsample
  else if (p.Ho != p.Hi || p.Wo != p.Wi) return false;
  if ((p.flags & GF_RESID) && (p.ldr & 3)) return false;
  if (p.Ho < 1 || p.Wo < 1 || p.M % (p.Ho * p.Wo)) return false;
  const int images = p.M / (p.Ho * p.Wo);
  if (variant == 0) { if (p.Ho != 8 || p.Wo != 8 || images % 3) return false; }
  else if (variant == 1) { if (p.Ho != 16 || p.Wo != 16) return false; }
  else if (variant == 2 || variant == 3) { if ((p.Ho & 7) || (p.Wo & 15)) return false; }
  else return false;
  if (p.flags & GF_GNSTATS) {  // statistics for the consuming GroupNorm: unsplit launches of the two-n-tile builds, groups of <= 64 channels
    if (variant < 2 || nsplit != 1 || !p.st_out || p.gn_cpg < 4 || p.gn_cpg > 64 || (p.N % p.gn_cpg) || p.N / p.gn_cpg > 32) return false;
  }
  if (nsplit < 1 || nsplit > p.Cin / 64) return false;
  if ((size_t)images * p.Hi * p.Wi * p.lda * 2 >= ((size_t)1 << 31) || dtp_conv_ws_packed_elems(p.N, p.Cin, p.A2 ? p.Cin2 : 0) * 2 >= ((size_t)1 << 31)) return false;
  return true;
}

int dtp_launch_conv_ws(const GemmParams& pin, int variant, hipStream_t s) {
  if (!dtp_conv_ws_supported(pin, variant, pin.splits)) {
    dtp_set_error("conv_ws: unsupported problem (M %d N %d Cin %d %dx%d flags %#x, variant %d, %d slices)", pin.M, pin.N, pin.Cin, pin.Hi, pin.Wi, pin.flags, variant, pin.splits);
    return DTP_ERR_ARG;
  }
  const int rc = variant == 0 ? launch_ws<8, 8, 3, 1>(pin, s) : variant == 1 ? launch_ws<16, 16, 1, 1>(pin, s) : variant == 2 ? launch_ws<8, 16, 1, 2>(pin, s) : launch_ws<8, 16, 1, 2, true>(pin, s);
  if (rc != DTP_OK) { dtp_set_error("conv_ws launch failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
  if (pin.splits > 1 && !(pin.flags & GF_NOREDUCE)) return dtp_launch_splitk_reduce(pin, s);
  return DTP_OK;
}
