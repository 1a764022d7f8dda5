// Cross-attention against the 14 brush tokens as ONE launch per transformer block.
//
// unet.hip folds attn2 (to_q . K^T, softmax over the 14 tokens, . V, to_out) into two grouped GEMMs against per-sample matrices
// prepared once per stamp:   P = softmax_16(LN2(y2) W1^T + b1)   [rows x 8 heads x 16]      (K = C)
//                            y3 = P W2^T + b_o + y2              [rows x C]                 (K = 128)
// At batch 1 both launches are pure fixed cost (0.1-0.5 GFLOP each, ~5.5 us per graph node): this kernel runs them
// back to back inside one workgroup.  A workgroup owns 64 rows of one sample and ONE 128-column tile of y3: it computes the
// 64 x 128 probability tile itself (phase 1, the whole K = C contraction -- recomputed by the C/128 workgroups that share the rows:
// 3-10x a few hundred MFLOP), keeps it in LDS as the A operand of phase 2 and multiplies it with its W2 tile, whose DMA was issued
// before phase 1 started.  Same MFMA mapping, LDS image (128-byte rows, XOR-swizzled 16-byte chunks) and epilogue arithmetic as
// gemm_kernel (gemm_conv.hip): against the unfused pair on 4-wave tiles the result is bit-identical -- P is rounded to fp16 exactly
// where the first GEMM would have stored it.
// Round 5: a workgroup may own SEVERAL consecutive 128-column tiles (XattnParams::ct): the probability tile is computed once and multiplied
// with one W2 tile after the other (the next tile's DMA runs under the current tile's MFMAs, into the ring phase 1 no longer needs).
// UNet level 0 (C = 320: three column tiles, 64 row blocks x 3 samples) then is ONE 192-workgroup launch without any recomputation
// instead of the two grouped GEMMs it used to be (the fused form lost there: it recomputed the scores three times).
// Reference: diffusers CrossAttention (attn2 of BasicTransformerBlock) as called from trt_inference/models.py:1097-1139 (the UNet
// engine); the algebraic fusion itself is described in unet.hip transformer().
#include "common.h"

namespace {

template <int OFF>
__device__ __forceinline__ f16x8 lds_read16_off(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int XBM = 64, XBN = 128;              // rows per workgroup; columns of both phases' tiles
constexpr int XSTAGE = (XBM + XBN) * 128;       // one phase-1 k-block: 64 activation rows + 128 W1 rows
constexpr int XW2 = 2 * XBN * 128;              // the W2 tile: two k-blocks of 128 rows
constexpr int XP = 2 * XBM * 128;               // the probability tile: two k-blocks of 64 rows
constexpr int XSLD = XBN + 8;                   // staging row stride (f16)
constexpr int XNS = 4;                          // phase-1 ring depth: three k-blocks in flight (two stages left the DMA latency of
                                                // every one of the C/64 k-blocks exposed: ~1 us each in a 20 us launch)
constexpr int XLDS = XNS * XSTAGE + XW2 + XP + XBM * 8;

// MULTI = false: exactly one column tile per workgroup (the round-4 kernel: with the tile loop around phase 2 its launches were 4-6 %
// slower -- loop-carried epilogue operands, an extra wait and barrier); MULTI = true: XattnParams::ct tiles
template <bool MULTI>
__global__ __launch_bounds__(256) void xattn_kernel(const XattnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;                       // [XNS][XSTAGE]; reused as the fp16 staging tile of both epilogues
  char* const w2s = smem + XNS * XSTAGE;         // [2][128][128 B]
  char* const ps = w2s + XW2;                    // [2][64][128 B]
  float* const rowst = (float*)(ps + XP);        // [64][2] mean, rstd

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_m = blockIdx.x, smp = blockIdx.z;
  const int ntiles = (p.C + XBN - 1) / XBN;
  const int ct0 = MULTI ? blockIdx.y * p.ct : blockIdx.y, ct1 = MULTI ? min(ct0 + p.ct, ntiles) : ct0 + 1;  // this workgroup's column tiles
  const int m0 = tile_m * XBM;
  int tile_n = ct0, n0 = ct0 * XBN;
  const f16* X = p.X + (size_t)smp * p.S * p.ldx;
  const f16* W1 = p.W1 + (size_t)smp * p.w1_bs;
  const f16* W2 = p.W2 + (size_t)smp * p.w2_bs;
  const size_t row0 = (size_t)smp * p.S;         // first global row of this sample (statistics tables, output)

  // ---- the epilogue's per-thread operands first: bias chunk and the residual rows of this thread's four output chunks (clamped rows:
  // unconditional loads).  Requested before any DMA, consumed after phase 2 -- loaded there they cost one exposed L2 round trip at
  // the very end of every launch.
  constexpr int NC = XBN / 8;  // 16: a thread owns the same 8-column chunk in every epilogue iteration
  constexpr int EIT = XBM * NC / 256;  // 4
  const int nc = tid % NC;
  int n = n0 + nc * 8;
  bool col_ok = (n + 8 <= p.C);  // C % 8 == 0
  float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col_ok && p.b2) {
    const f32x4 t0 = *(const f32x4*)(p.b2 + n), t1 = *(const f32x4*)(p.b2 + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = t0[e]; bv[4 + e] = t1[e]; }
  }
  // (epilogue 1: this thread's chunk of the per-sample score bias and weight row sums)
  const f32x4 e1b0 = *(const f32x4*)(p.b1 + (size_t)smp * 128 + nc * 8), e1b1 = *(const f32x4*)(p.b1 + (size_t)smp * 128 + nc * 8 + 4);
  const f32x4 e1l0 = *(const f32x4*)(p.lns1 + (size_t)smp * 128 + nc * 8), e1l1 = *(const f32x4*)(p.lns1 + (size_t)smp * 128 + nc * 8 + 4);
  f16x8 rv[EIT];
#pragma unroll
  for (int it = 0; it < EIT; ++it) {
    const int mr = min(m0 + (tid + it * 256) / NC, p.S - 1);
    rv[it] = col_ok ? *(const f16x8*)(p.R + (row0 + mr) * p.ldr + n) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }

  // ---- DMA sources.  LDS row r = i*32 + wave*8 + (lane>>3); slot lane&7 holds source chunk slot ^ ((r>>1)&7)
  const int lrow = wave * 8 + (lane >> 3);
  const int kc = (((lane & 7) ^ ((lrow >> 1) & 7)) << 3);
  // Buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds): loop-constant 32-bit lane offsets, the k-block as a scalar offset -- no
  // 64-bit address arithmetic per piece (round 4: the global_load_lds form it replaces was most of a k-block's issue time).  Rows
  // beyond S carry an out-of-range offset: the DMA writes zeros.
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)0x80000000u, 0x00020000);
  const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)W1, 0, (int)0x80000000u, 0x00020000);
  const auto rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)W2, 0, (int)0x80000000u, 0x00020000);
  int a_off[2], w1_off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + i * 32 + lrow;
    a_off[i] = (m < p.S) ? (m * p.ldx + kc) * 2 : -1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) w1_off[i] = ((i * 32 + lrow) * p.C + kc) * 2;
  // one W2 tile (rows beyond C are zero padding of the per-sample matrices).  The first one now: it lands while phase 1 runs.
  char* const w2alt = ring + 32768;  // the second W2 buffer: ring memory behind the staging tile, free once phase 1 is over
  auto issue_w2 = [&](int ct, char* dst) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int vo = ((ct * XBN + i * 32 + lrow) * 128 + kc) * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (lds_ptr_t)(dst + kb * (XBN * 128) + (i * 32 + wave * 8) * 128), 16, vo, kb * 128, 0, 0);
      }
  };
  issue_w2(ct0, w2s);
  auto issue = [&](int stage, int kb) {
    char* As = ring + stage * XSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vo = a_off[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_t)(As + (i * 32 + wave * 8) * 128), 16, vo, kb * 128, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vo = w1_off[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (lds_ptr_t)(As + XBM * 128 + (i * 32 + wave * 8) * 128), 16, vo, kb * 128, 0, 0);
    }
  };
  const int nkb = p.C >> 6;
#pragma unroll
  for (int t = 0; t < XNS - 1; ++t)
    if (t < nkb) issue(t, t);

  // ---- LayerNorm-2 statistics of this tile's rows from the producer's partial sums (GF_LNFOLD of gemm_kernel)
  if (tid < XBM) {
    const int m = m0 + tid;
    float s1 = 0.f, s2 = 0.f;
    if (m < p.S) sum_pairs_strided(p.st_in + (row0 + m) * 2, (size_t)p.st_rows * 2, p.st_parts, s1, s2);
    const float mean = s1 / (float)p.C;
    rowst[2 * tid] = mean;
    rowst[2 * tid + 1] = rsqrtf(fmaxf(s2 / (float)p.C - mean * mean, 0.f) + p.ln_eps);
  }

  const int wn0 = (wave & 1) * 64, wm0 = (wave >> 1) * 32;
  const int frow = lane & 31, fhalf = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // one 64-wide k-block of a [64 rows | 128 rows] image pair: weight fragment = MFMA A operand, activation fragment = B operand.
  // All twelve fragment reads of the k-block are requested up front (untracked ds_read_b128, hand-counted lgkmcnt: with LDS-DMA in the
  // loop hipcc only ever waits lgkmcnt(0), i.e. one exposed LDS round trip per MFMA -- 8 per k-block with a single wave per SIMD and
  // nothing to hide them: 0.5 of the 0.6 us a k-block took).  Request order: (a, w0, w1) of k-step 0, 1, 2, 3.
  uint32_t xa[4], xw[4];  // per-lane byte addresses inside a stage: activation row wm0 + frow, weight row wn0 + frow (+ 32 rows = +4096)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = ks * 2 + fhalf;
    const int ar = wm0 + frow, wr = wn0 + frow;
    xa[ks] = lds_addr(ring) + ar * 128 + ((c ^ ((ar >> 1) & 7)) << 4);
    xw[ks] = lds_addr(ring) + wr * 128 + ((c ^ ((wr >> 1) & 7)) << 4);
  }
  auto kblock = [&](uint32_t ao, uint32_t wo) {  // byte offsets (from the ring's base) of the activation and the weight image
    f16x8 af[4], wf[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      af[ks] = lds_read16(xa[ks] + ao);
      wf[ks][0] = lds_read16(xw[ks] + wo);
      wf[ks][1] = lds_read16_off<4096>(xw[ks] + wo);
    }
#define DTP_XA_STEP(ks)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(af[ks]), "+v"(wf[ks][0]) : "n"(10 - 3 * (ks)));                          \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][0], af[ks], acc[0], 0, 0, 0);                                 \
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(wf[ks][1]) : "n"(9 - 3 * (ks)));                                         \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][1], af[ks], acc[1], 0, 0, 0);
    DTP_XA_STEP(0) DTP_XA_STEP(1) DTP_XA_STEP(2) DTP_XA_STEP(3)
#undef DTP_XA_STEP
  };

  // ---- phase 1: S = X W1^T over K = C; XNS-deep ring, counted vmcnt (6 DMA instructions per wave and k-block; the W2 tile's 8 are
  // older than every k-block)
  {
    int slot = 0, nslot = XNS - 1;
    for (int t = 0; t < nkb; ++t) {
      const int ahead = min(XNS - 2, nkb - 1 - t);  // younger k-blocks already issued
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // k-block t has landed for every wave; everyone has left k-block t-1.  A RAW barrier: __syncthreads() makes hipcc wait vmcnt(0)
      // while LDS-DMA is in flight, i.e. it drained the two younger k-blocks of the ring at every k-block (round 4: that was the whole
      // exposed latency of this loop, ~1 us per k-block)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t + XNS - 1 < nkb) issue(nslot, t + XNS - 1);
      kblock((uint32_t)(slot * XSTAGE), (uint32_t)(slot * XSTAGE + XBM * 128));
      slot = (slot + 1 == XNS) ? 0 : slot + 1;
      nslot = (nslot + 1 == XNS) ? 0 : nslot + 1;
    }
  }
  __syncthreads();  // the ring is free: it becomes the staging tile

  // D layout (32x32): lane holds column (lane&31) = row m, registers r -> channel (r&3) + 8*(r>>2) + 4*(lane>>5)
  f16* const stg = (f16*)ring;
  auto stage_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ml = wm0 + frow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn0 + i * 32 + 8 * q + 4 * fhalf;
        f16x4 v = {(f16)acc[i][4 * q], (f16)acc[i][4 * q + 1], (f16)acc[i][4 * q + 2], (f16)acc[i][4 * q + 3]};
        *(f16x4*)(stg + ml * XSLD + nl) = v;
      }
    }
  };
  stage_acc();
  __syncthreads();

  // ---- epilogue 1: LayerNorm fold + bias + softmax over each 16-column head group (sm_valid columns) -> P (fp16) into the
  // phase-2 A-operand image.  A thread owns the same 8-column chunk in every iteration; its pair lane (nc ^ 1) holds the other half.
  {
    float bv[8], lv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = e1b0[e]; bv[4 + e] = e1b1[e]; lv[e] = e1l0[e]; lv[4 + e] = e1l1[e]; }
    const int half = (nc & 1) * 8;
    for (int idx = tid; idx < XBM * NC; idx += 256) {
      const int ml = idx / NC;
      const f16x8 v = *(const f16x8*)(stg + ml * XSLD + nc * 8);
      const float mean = rowst[2 * ml], rstd = rowst[2 * ml + 1];
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = rstd * ((float)v[e] - mean * lv[e]);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += bv[e];
      float mx = -3.0e38f;
#pragma unroll
      for (int e = 0; e < 8; ++e) if (half + e < p.sm_valid) mx = fmaxf(mx, x[e]);
      mx = group_allmax<2>(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[e] = (half + e < p.sm_valid) ? __expf(x[e] - mx) : 0.f; sum += x[e]; }
      sum = group_allsum<2>(sum);
      const float inv = 1.0f / sum;
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(x[e] * inv);
      // column chunk nc of row ml = chunk (nc & 7) of k-block (nc >> 3)
      *(f16x8*)(ps + (nc >> 3) * (XBM * 128) + ml * 128 + (((nc & 7) ^ ((ml >> 1) & 7)) << 4)) = o;
    }
  }
  __syncthreads();  // P complete (the first W2 tile landed before the first phase-1 barrier)

  // ---- per column tile: phase 2 (y3 tile = P W2^T over K = 128) + epilogue 2
  for (int ct = ct0; ct < ct1; ++ct) {
    char* const w2cur = ((ct - ct0) & 1) ? w2alt : w2s;
    if (MULTI && ct > ct0) {
      // this tile's W2 has landed (for every wave), and everyone has left the previous tile's staging tile and W2 buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // the next tile: its W2 into the other buffer (last read one tile ago), its bias chunk and residual rows into registers -- all of it
    // lands under this tile's MFMAs and epilogue
    const bool more = MULTI && ct + 1 < ct1;
    float bvn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f16x8 rvn[EIT];
    const int nn = (ct + 1) * XBN + nc * 8;
    const bool col_okn = more && (nn + 8 <= p.C);
    if (more) {
      issue_w2(ct + 1, ((ct - ct0) & 1) ? w2s : w2alt);
      if (col_okn && p.b2) {
        const f32x4 t0 = *(const f32x4*)(p.b2 + nn), t1 = *(const f32x4*)(p.b2 + nn + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bvn[e] = t0[e]; bvn[4 + e] = t1[e]; }
      }
#pragma unroll
      for (int it = 0; it < EIT; ++it) {
        const int mr = min(m0 + (tid + it * 256) / NC, p.S - 1);
        rvn[it] = col_okn ? *(const f16x8*)(p.R + (row0 + mr) * p.ldr + nn) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    kblock((uint32_t)(ps - ring), (uint32_t)(w2cur - ring));
    kblock((uint32_t)(ps - ring) + XBM * 128, (uint32_t)(w2cur - ring) + XBN * 128);
    __syncthreads();  // everyone has read the staging tile of the previous epilogue
    stage_acc();
    __syncthreads();

    // ---- epilogue 2: + bias + residual, fp16 store, per-row (sum, sum of squares) of the stored values for the next LayerNorm fold
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int ml = (tid + it * 256) / NC, m = m0 + ml;
      const bool active = (m < p.S) && col_ok;
      float s1 = 0.f, s2 = 0.f;
      if (active) {
        const f16x8 v = *(const f16x8*)(stg + ml * XSLD + nc * 8);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += bv[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += (float)rv[it][e];
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = (f16)x[e];
          const float f = (float)o[e];
          s1 += f; s2 += f * f;
        }
        *(f16x8*)(p.Y + (row0 + m) * p.ldy + n) = o;
      }
      if (p.st_out) {  // NC consecutive lanes hold one row of this column tile: fixed-order shuffle reduce
        s1 = group_allsum<NC>(s1); s2 = group_allsum<NC>(s2);
        if (nc == 0 && m < p.S) {
          p.st_out[((size_t)ct * p.st_rows + row0 + m) * 2] = s1;
          p.st_out[((size_t)ct * p.st_rows + row0 + m) * 2 + 1] = s2;
        }
      }
    }
    if constexpr (!MULTI) break;
    // hand over to the next tile
    n = nn; col_ok = col_okn;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bvn[e];
#pragma unroll
    for (int it = 0; it < EIT; ++it) rv[it] = rvn[it];
  }
}

}  // namespace

void dtp_xattn_init() {
  (void)hipFuncSetAttribute((const void*)xattn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS);
  (void)hipFuncSetAttribute((const void*)xattn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS);
}

bool dtp_xattn_supported(const XattnParams& p) {
  return p.C >= 64 && (p.C & 63) == 0 && (p.ldx & 7) == 0 && (p.ldy & 7) == 0 && (p.ldr & 7) == 0 && p.S >= 1 && p.N >= 1 && p.sm_valid >= 1 &&
         p.sm_valid <= 16 && p.st_in && p.st_parts >= 1 &&
         (size_t)p.S * p.ldx * 2 < ((size_t)1 << 31) && (size_t)(p.C + 128) * 256 < ((size_t)1 << 31);  // 32-bit lane offsets of the DMA
}

// column tiles per workgroup: as many as still leave ~0.6 x 256 workgroups (fewer recomputations of the probability tile; below that the
// launch no longer fills the chip and a workgroup's serial chain of tiles decides)
int dtp_xattn_tiles_per_wg(int S, int C, int N) {
  const int rb = ((S + XBM - 1) / XBM) * N, nt = (C + XBN - 1) / XBN;
  int ct = 1;
  for (int c = 2; c <= nt; ++c)
    if (rb * ((nt + c - 1) / c) >= 160) ct = c;
  return ct;
}

int dtp_launch_xattn(const XattnParams& pin, hipStream_t s) {
  if (!dtp_xattn_supported(pin)) { dtp_set_error("xattn: unsupported problem (C=%d S=%d)", pin.C, pin.S); return DTP_ERR_ARG; }
  XattnParams p = pin;
  const int nt = (p.C + XBN - 1) / XBN;
  if (p.ct < 1) p.ct = dtp_xattn_tiles_per_wg(p.S, p.C, p.N);
  if (p.ct > nt) p.ct = nt;
  const dim3 grid((p.S + XBM - 1) / XBM, (nt + p.ct - 1) / p.ct, p.N);
  if (p.ct > 1) hipLaunchKernelGGL(xattn_kernel<true>, grid, dim3(256), XLDS, s, p);
  else hipLaunchKernelGGL(xattn_kernel<false>, grid, dim3(256), XLDS, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
