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
// Reference: diffusers CrossAttention (attn2 of BasicTransformerBlock) as called from trt_inference/models.py:1097-1139 (the UNet
// engine); the algebraic fusion itself is described in unet.hip transformer().
#include "common.h"

namespace {

__device__ __forceinline__ void glds16(const void* src, void* lds_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

constexpr int XBM = 64, XBN = 128;              // rows per workgroup; columns of both phases' tiles
constexpr int XSTAGE = (XBM + XBN) * 128;       // one phase-1 k-block: 64 activation rows + 128 W1 rows
constexpr int XW2 = 2 * XBN * 128;              // the W2 tile: two k-blocks of 128 rows
constexpr int XP = 2 * XBM * 128;               // the probability tile: two k-blocks of 64 rows
constexpr int XSLD = XBN + 8;                   // staging row stride (f16)
constexpr int XNS = 4;                          // phase-1 ring depth: three k-blocks in flight (two stages left the DMA latency of
                                                // every one of the C/64 k-blocks exposed: ~1 us each in a 20 us launch)
constexpr int XLDS = XNS * XSTAGE + XW2 + XP + XBM * 8;

__global__ __launch_bounds__(256) void xattn_kernel(const XattnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;                       // [XNS][XSTAGE]; reused as the fp16 staging tile of both epilogues
  char* const w2s = smem + XNS * XSTAGE;         // [2][128][128 B]
  char* const ps = w2s + XW2;                    // [2][64][128 B]
  float* const rowst = (float*)(ps + XP);        // [64][2] mean, rstd

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_m = blockIdx.x, tile_n = blockIdx.y, smp = blockIdx.z;
  const int m0 = tile_m * XBM, n0 = tile_n * XBN;
  const f16* X = p.X + (size_t)smp * p.S * p.ldx;
  const f16* W1 = p.W1 + (size_t)smp * p.w1_bs;
  const f16* W2 = p.W2 + (size_t)smp * p.w2_bs;
  const size_t row0 = (size_t)smp * p.S;         // first global row of this sample (statistics tables, output)

  // ---- DMA sources.  LDS row r = i*32 + wave*8 + (lane>>3); slot lane&7 holds source chunk slot ^ ((r>>1)&7)
  const int lrow = wave * 8 + (lane >> 3);
  const int kc = (((lane & 7) ^ ((lrow >> 1) & 7)) << 3);
  const f16* a_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + i * 32 + lrow;
    a_row[i] = (m < p.S) ? X + (size_t)m * p.ldx + kc : nullptr;
  }
  const f16* w1_row[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w1_row[i] = W1 + (size_t)(i * 32 + lrow) * p.C + kc;
  // the whole W2 tile now: it lands while phase 1 runs (rows beyond C are zero padding of the per-sample matrices)
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(W2 + (size_t)(n0 + i * 32 + lrow) * 128 + kb * 64 + kc, w2s + kb * (XBN * 128) + (i * 32 + wave * 8) * 128);
  auto issue = [&](int stage, int kb) {
    char* As = ring + stage * XSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_row[i] ? a_row[i] + (size_t)kb * 64 : p.zero, As + (i * 32 + wave * 8) * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w1_row[i] + (size_t)kb * 64, As + XBM * 128 + (i * 32 + wave * 8) * 128);
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
  // one 64-wide k-block of a [64 rows | 128 rows] image pair: weight fragment = MFMA A operand, activation fragment = B operand
  auto kblock = [&](const char* As, const char* Ws) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + fhalf;
      const int ar = wm0 + frow;
      const f16x8 af = *(const f16x8*)(As + ar * 128 + ((c ^ ((ar >> 1) & 7)) << 4));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int wr = wn0 + i * 32 + frow;
        const f16x8 wf = *(const f16x8*)(Ws + wr * 128 + ((c ^ ((wr >> 1) & 7)) << 4));
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, acc[i], 0, 0, 0);
      }
    }
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
      __syncthreads();                                 // k-block t has landed for every wave; everyone has left k-block t-1
      if (t + XNS - 1 < nkb) issue(nslot, t + XNS - 1);
      const char* As = ring + slot * XSTAGE;
      kblock(As, As + XBM * 128);
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
  constexpr int NC = XBN / 8;  // 16
  const int nc = tid % NC;
  {
    const float* b1 = p.b1 + (size_t)smp * 128 + nc * 8;
    const float* l1 = p.lns1 + (size_t)smp * 128 + nc * 8;
    const f32x4 t0 = *(const f32x4*)b1, t1 = *(const f32x4*)(b1 + 4), u0 = *(const f32x4*)l1, u1 = *(const f32x4*)(l1 + 4);
    float bv[8], lv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = t0[e]; bv[4 + e] = t1[e]; lv[e] = u0[e]; lv[4 + e] = u1[e]; }
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
      mx = fmaxf(mx, __shfl_xor(mx, 1));
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { x[e] = (half + e < p.sm_valid) ? __expf(x[e] - mx) : 0.f; sum += x[e]; }
      sum += __shfl_xor(sum, 1);
      const float inv = 1.0f / sum;
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)(x[e] * inv);
      // column chunk nc of row ml = chunk (nc & 7) of k-block (nc >> 3)
      *(f16x8*)(ps + (nc >> 3) * (XBM * 128) + ml * 128 + (((nc & 7) ^ ((ml >> 1) & 7)) << 4)) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  __syncthreads();  // P complete (the W2 tile landed before the first phase-1 barrier)

  // ---- phase 2: y3 tile = P W2^T over K = 128
  kblock(ps, w2s);
  kblock(ps + XBM * 128, w2s + XBN * 128);
  __syncthreads();  // everyone has read the staging tile of epilogue 1
  stage_acc();
  __syncthreads();

  // ---- epilogue 2: + bias + residual, fp16 store, per-row (sum, sum of squares) of the stored values for the next LayerNorm fold
  const int n = n0 + nc * 8;
  const bool col_ok = (n + 8 <= p.C);  // C % 8 == 0
  float bv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col_ok && p.b2) {
    const f32x4 t0 = *(const f32x4*)(p.b2 + n), t1 = *(const f32x4*)(p.b2 + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bv[e] = t0[e]; bv[4 + e] = t1[e]; }
  }
  constexpr int EIT = XBM * NC / 256;  // 4
  f16x8 rv[EIT];
#pragma unroll
  for (int it = 0; it < EIT; ++it) {  // all residual rows up front (clamped rows: unconditional loads)
    const int mr = min(m0 + (tid + it * 256) / NC, p.S - 1);
    rv[it] = col_ok ? *(const f16x8*)(p.R + (row0 + mr) * p.ldr + n) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
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
#pragma unroll
      for (int o = NC / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      if (nc == 0 && m < p.S) {
        p.st_out[((size_t)tile_n * p.st_rows + row0 + m) * 2] = s1;
        p.st_out[((size_t)tile_n * p.st_rows + row0 + m) * 2 + 1] = s2;
      }
    }
  }
}

}  // namespace

void dtp_xattn_init() { (void)hipFuncSetAttribute((const void*)xattn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XLDS); }

bool dtp_xattn_supported(const XattnParams& p) {
  return p.C >= 64 && (p.C & 63) == 0 && (p.ldx & 7) == 0 && (p.ldy & 7) == 0 && (p.ldr & 7) == 0 && p.S >= 1 && p.N >= 1 && p.sm_valid >= 1 &&
         p.sm_valid <= 16 && p.st_in && p.st_parts >= 1;
}

int dtp_launch_xattn(const XattnParams& p, hipStream_t s) {
  if (!dtp_xattn_supported(p)) { dtp_set_error("xattn: unsupported problem (C=%d S=%d)", p.C, p.S); return DTP_ERR_ARG; }
  hipLaunchKernelGGL(xattn_kernel, dim3((p.S + XBM - 1) / XBM, (p.C + XBN - 1) / XBN, p.N), dim3(256), XLDS, s, p);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
