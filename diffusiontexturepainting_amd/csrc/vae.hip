// AutoencoderKL encoder / decoder on the libdtp kernels and the `vae_encoder` / `vae` engine
// entry points.  Reference: trt_inference/models.py:1237-1244 (decode(x).sample), :1328-1335
// (encode(x).latent_dist.sample()), I/O contracts :1253-1284, :1343-1377; topology SURVEY.md A.2.
#include <math.h>

#include <stdlib.h>

#include "engine.h"

int launch_nhwc_f32_to_nchw(const float* x, float* y, int B, int C, int HW, int ldx, hipStream_t s);

static int load_res_v(Ctx* c, const std::string& p, ResW& w) {
  RC(load_norm(c, p + ".norm1", w.n1));
  RC(load_conv(c, p + ".conv1", w.c1));
  RC(load_norm(c, p + ".norm2", w.n2));
  w.has_sc = ctx_find(c, p + ".conv_shortcut.weight") != nullptr;
  if (w.has_sc) RC(load_conv_with_shortcut(c, p + ".conv2", p + ".conv_shortcut", w.c2));
  else RC(load_conv(c, p + ".conv2", w.c2));
  return DTP_OK;
}

static int load_attn_v(Ctx* c, const std::string& p, VaeAttnW& w) {
  RC(load_norm(c, p + ".group_norm", w.gn));
  RC(load_linear(c, {p + ".query", p + ".key"}, w.qk, true));
  RC(load_linear(c, {p + ".proj_attn"}, w.out, true));
  RC(load_plain_f16(c, p + ".value.weight", &w.wv));
  std::vector<float> bv;
  RC(ctx_fetch_host(c, p + ".value.bias", bv));
  return ctx_upload_f32(c, bv, &w.bv);
}

int load_vae_weights(Ctx* c) {
  struct WsScope {  // the VAE's 3x3 convs (64 ... 512 pixel maps, 128 ... 512 channels) also get the fragment-order packing (conv_ws.hip)
    Ctx* c;
    explicit WsScope(Ctx* cc) : c(cc) { const char* e = getenv("DTP_NO_WS"); const char* v = getenv("DTP_NO_WS_VAE"); c->pack_ws = !(e && e[0] && e[0] != '0') && !(v && v[0] && v[0] != '0'); }
    ~WsScope() { c->pack_ws = false; }
  } ws_scope(c);
  VaeW& v = c->vae;
  const std::string P = "vae.";
  RC(load_conv(c, P + "encoder.conv_in", v.enc_in, 8));
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j)
      RC(load_res_v(c, P + "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), v.enc_res[i][j]));
    if (i < 3) RC(load_conv(c, P + "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", v.enc_down[i]));
  }
  RC(load_res_v(c, P + "encoder.mid_block.resnets.0", v.enc_mid[0]));
  RC(load_attn_v(c, P + "encoder.mid_block.attentions.0", v.enc_attn));
  RC(load_res_v(c, P + "encoder.mid_block.resnets.1", v.enc_mid[1]));
  RC(load_norm(c, P + "encoder.conv_norm_out", v.enc_norm_out));
  RC(load_conv(c, P + "encoder.conv_out", v.enc_out));
  RC(load_conv(c, P + "decoder.conv_in", v.dec_in, 8));
  RC(load_res_v(c, P + "decoder.mid_block.resnets.0", v.dec_mid[0]));
  RC(load_attn_v(c, P + "decoder.mid_block.attentions.0", v.dec_attn));
  RC(load_res_v(c, P + "decoder.mid_block.resnets.1", v.dec_mid[1]));
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 3; ++j)
      RC(load_res_v(c, P + "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), v.dec_res[i][j]));
    if (i < 3) RC(load_conv(c, P + "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", v.dec_up[i]));
  }
  RC(load_norm(c, P + "decoder.conv_norm_out", v.dec_norm_out));
  RC(load_conv(c, P + "decoder.conv_out", v.dec_out));
  std::vector<float> t;
  RC(ctx_fetch_host(c, P + "quant_conv.weight", t)); RC(ctx_upload_f32(c, t, &v.quant_w));
  RC(ctx_fetch_host(c, P + "quant_conv.bias", t)); RC(ctx_upload_f32(c, t, &v.quant_b));
  RC(ctx_fetch_host(c, P + "post_quant_conv.weight", t)); RC(ctx_upload_f32(c, t, &v.pquant_w));
  RC(ctx_fetch_host(c, P + "post_quant_conv.bias", t)); RC(ctx_upload_f32(c, t, &v.pquant_b));
  return DTP_OK;
}

// Single-head attention over C = 512 (diffusers AttentionBlock): head dim 512 does not fit the
// flash kernel's register budget, and at 2 x 34 GFLOP per stamp it is 1.3 % of the VAE, so it runs
// as plain MFMA GEMMs: S = Q K^T, row softmax, O = P V with V^T produced by an operand-swapped GEMM.
static int vae_attention(Builder& b, const T& x, const VaeAttnW& w, T& out) {
  Ctx* c = b.c;
  const int C = x.C, S = x.H * x.W, B = x.B;
  T t, qk, o;
  RC(b.gn(x, w.gn, 1e-6f, false, t));
  RC(b.linear(t, w.qk, nullptr, 0, qk));
  o = b.alloc(x.B, x.H, x.W, C);
  T sc = b.alloc(1, 1, S, S), vt = b.alloc(1, 1, C, S);
  if (!o.p || !sc.p || !vt.p) return DTP_ERR_HIP;
  const float scale = 1.0f / sqrtf((float)C);
  for (int bi = 0; bi < B; ++bi) {
    auto push = [&](GemmParams g) {
      int tile = 0;
      g.zero = c->zero;
      g.nkb = (g.K + 63) / 64;
      dtp_gemm_pick(g, &tile, c->num_cu);
      g.splits = 1; g.kb_per_split = g.nkb;
      b.push(PK_GEMM0 + tile, 2.0 * g.M * (double)g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N),
             [=](hipStream_t s, int) { return dtp_launch_gemm(g, tile, s); });
    };
    GemmParams g = {};
    // V^T[c][s] = Wv[c][:] . t[s][:] + bv[c]
    g.A = w.wv; g.W = t.p + (size_t)bi * S * t.ld; g.C = vt.p; g.bias = w.bv;
    g.M = C; g.N = S; g.K = C; g.lda = C; g.ldw = t.ld; g.ldc = S; g.flags = GF_BIAS_M;
    push(g);
    // scores[q][k] = Q[q][:] . K[k][:]
    g = GemmParams();
    g.A = qk.p + (size_t)bi * S * qk.ld; g.W = qk.p + (size_t)bi * S * qk.ld + C; g.C = sc.p;
    g.M = S; g.N = S; g.K = C; g.lda = qk.ld; g.ldw = qk.ld; g.ldc = S;
    push(g);
    const T scc = sc;
    b.push(PK_SOFTMAX, 0.0, 4.0 * (double)S * S, [=](hipStream_t s, int) { return dtp_launch_softmax_rows(scc.p, S, scc.p, S, S, S, scale, s); });
    // O[q][c] = P[q][:] . V^T[c][:]
    g = GemmParams();
    g.A = sc.p; g.W = vt.p; g.C = o.p + (size_t)bi * S * o.ld;
    g.M = S; g.N = C; g.K = S; g.lda = S; g.ldw = S; g.ldc = o.ld;
    push(g);
  }
  b.release(t); b.release(qk); b.release(sc); b.release(vt);
  RC(b.linear(o, w.out, &x, 0, out));
  b.release(o);
  return DTP_OK;
}

// largest sub-batch whose [Bc][R][R][Cmax] f16 tensor stays below the 2 GiB the buffer descriptors of the GEMM / conv kernels span
static int vae_sub_batch(int B, int R, int Cmax) {
  const long long per = (long long)R * R * Cmax * 2;
  const int cap = (int)std::max(1LL, ((1LL << 31) - 1) / per);
  const int n = (B + cap - 1) / cap;  // number of sub-batches; spread evenly so they share their tuned tiles
  return (B + n - 1) / n;
}

int build_vae_enc_prog(Ctx* c, int B, VaeEncProg& pr) {
  const VaeW& v = c->vae;
  const int R = c->R;
  pr.B = B;
  void* p;
  RC(ctx_persistent(c, (size_t)B * R * R * 8 * 2, &p, true)); pr.in8 = (f16*)p;
  RC(ctx_persistent(c, (size_t)B * c->h * c->h * 8 * 4, &p, true)); pr.moments = (float*)p;
  Builder b{c, &pr.main};
  // the kernels address an operand by 32-bit byte offsets into a 2 GiB descriptor: a batch whose widest full-resolution tensor
  // (128 channels here) would reach that is run as sub-batches, one after the other, in the same program
  const int Bc = vae_sub_batch(B, R, 128);
  for (int b0 = 0; b0 < B; b0 += Bc) {
  const int Bn = std::min(Bc, B - b0);
  T x0;
  x0.p = pr.in8 + (size_t)b0 * R * R * 8; x0.B = Bn; x0.H = R; x0.W = R; x0.C = 8; x0.ld = 8;
  T x;
  RC(b.conv3(x0, v.enc_in, 1, 1, false, R, R, nullptr, -1, x));
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 2; ++j) {
      T y;
      RC(b.resnet(x, v.enc_res[i][j], 1e-6f, false, y));
      b.release(x);
      x = y;
    }
    if (i < 3) {  // F.pad(x, (0,1,0,1)) + conv stride 2 pad 0: top/left pad 0, the bottom/right tap reads the zero page
      T y;
      RC(b.conv3(x, v.enc_down[i], 2, 0, false, x.H / 2, x.W / 2, nullptr, -1, y));
      b.release(x);
      x = y;
    }
  }
  T y, z, w2, t;
  RC(b.resnet(x, v.enc_mid[0], 1e-6f, false, y)); b.release(x);
  RC(vae_attention(b, y, v.enc_attn, z)); b.release(y);
  RC(b.resnet(z, v.enc_mid[1], 1e-6f, false, w2)); b.release(z);
  RC(b.gn(w2, v.enc_norm_out, 1e-6f, true, t)); b.release(w2);
  T o;
  RC(b.conv3(t, v.enc_out, 1, 1, false, c->h, c->h, nullptr, -1, o, GF_OUT_F32, pr.moments + (size_t)b0 * c->h * c->h * 8, 8));
  b.release(t);
  }
  tune_cache_save(c);
  return ensure_ws(c);
}

int build_vae_dec_prog(Ctx* c, int B, VaeDecProg& pr) {
  const VaeW& v = c->vae;
  const int R = c->R, h = c->h;
  pr.B = B;
  void* p;
  RC(ctx_persistent(c, (size_t)B * h * h * 8 * 2, &p, true)); pr.in8 = (f16*)p;
  RC(ctx_persistent(c, (size_t)B * R * R * 4 * 4, &p, true)); pr.out32 = (float*)p;
  Builder b{c, &pr.main};
  const int Bc = vae_sub_batch(B, R, 256);  // (see build_vae_enc_prog; 256 channels at full resolution after the last upsampling)
  for (int b0 = 0; b0 < B; b0 += Bc) {
  const int Bn = std::min(Bc, B - b0);
  T x0;
  x0.p = pr.in8 + (size_t)b0 * h * h * 8; x0.B = Bn; x0.H = h; x0.W = h; x0.C = 8; x0.ld = 8;
  T x, y, z, w2;
  RC(b.conv3(x0, v.dec_in, 1, 1, false, h, h, nullptr, -1, x));
  RC(b.resnet(x, v.dec_mid[0], 1e-6f, false, y)); b.release(x);
  RC(vae_attention(b, y, v.dec_attn, z)); b.release(y);
  RC(b.resnet(z, v.dec_mid[1], 1e-6f, false, w2)); b.release(z);
  x = w2;
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 3; ++j) {
      T r;
      RC(b.resnet(x, v.dec_res[i][j], 1e-6f, false, r));
      b.release(x);
      x = r;
    }
    if (i < 3) {
      T r;
      RC(b.conv3(x, v.dec_up[i], 1, 1, true, x.H * 2, x.W * 2, nullptr, -1, r));
      b.release(x);
      x = r;
    }
  }
  T t, o;
  RC(b.gn(x, v.dec_norm_out, 1e-6f, true, t)); b.release(x);
  RC(b.conv3(t, v.dec_out, 1, 1, false, R, R, nullptr, -1, o, GF_OUT_F32, pr.out32 + (size_t)b0 * R * R * 4, 4));
  b.release(t);
  }
  tune_cache_save(c);
  return ensure_ws(c);
}

// moments [B][hw][8] (conv_out) -> quant_conv (8x8) -> mean/logvar -> latent NCHW f32 [B][4][hw] * scale
// (DiagonalGaussianDistribution.sample with the normal draw as an input; logvar clamped to [-30, 20])
__global__ void vae_sample_kernel(const float* __restrict__ mom, const float* __restrict__ qw, const float* __restrict__ qb,
                                  const float* __restrict__ eps, float* __restrict__ out, int B, int HW, float scale) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / HW), hw = (int)(i - (long long)b * HW);
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = mom[i * 8 + k];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      float mean = qb[ch], lv = qb[4 + ch];
#pragma unroll
      for (int k = 0; k < 8; ++k) { mean += qw[ch * 8 + k] * m[k]; lv += qw[(4 + ch) * 8 + k] * m[k]; }
      lv = fminf(fmaxf(lv, -30.f), 20.f);
      const size_t o = ((size_t)b * 4 + ch) * HW + hw;
      const float e = eps ? eps[o] : 0.f;
      out[o] = (mean + expf(0.5f * lv) * e) * scale;
    }
  }
}

int launch_vae_sample(Ctx* c, const float* mom, const float* eps, float* out, int B, float scale, hipStream_t s) {
  const int HW = c->h * c->h;
  const int blocks = std::min((B * HW + 255) / 256, 2048);
  hipLaunchKernelGGL(vae_sample_kernel, dim3(blocks), dim3(256), 0, s, mom, c->vae.quant_w, c->vae.quant_b, eps, out, B, HW, scale);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

// latent (NCHW or NHWC f32, * in_scale) -> post_quant_conv (4x4) -> NHWC f16 [B][hw][8]
__global__ void post_quant_kernel(const float* __restrict__ z, int nhwc, float in_scale, const float* __restrict__ pw,
                                  const float* __restrict__ pb, f16* __restrict__ out, int B, int HW) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / HW), hw = (int)(i - (long long)b * HW);
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (nhwc ? z[i * 4 + k] : z[((size_t)b * 4 + k) * HW + hw]) * in_scale;
    f16x8 o;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      float a = pb[ch];
#pragma unroll
      for (int k = 0; k < 4; ++k) a += pw[ch * 4 + k] * v[k];
      o[ch] = (f16)a;
    }
    o[4] = o[5] = o[6] = o[7] = (f16)0.f;
    *(f16x8*)(out + i * 8) = o;
  }
}

int launch_post_quant(Ctx* c, const float* z, int nhwc, float in_scale, f16* out, int B, hipStream_t s) {
  const int HW = c->h * c->h;
  const int blocks = std::min((B * HW + 255) / 256, 2048);
  hipLaunchKernelGGL(post_quant_kernel, dim3(blocks), dim3(256), 0, s, z, nhwc, in_scale, c->vae.pquant_w, c->vae.pquant_b, out, B, HW);
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}

int get_enc_prog(Ctx* c, int B, VaeEncProg** out) {
  auto it = c->enc_progs.find(B);
  if (it == c->enc_progs.end()) {
    VaeEncProg& p = c->enc_progs[B];
    RC(build_vae_enc_prog(c, B, p));
    *out = &p;
    return DTP_OK;
  }
  *out = &it->second;
  return DTP_OK;
}

int get_dec_prog(Ctx* c, int B, VaeDecProg** out) {
  auto it = c->dec_progs.find(B);
  if (it == c->dec_progs.end()) {
    VaeDecProg& p = c->dec_progs[B];
    RC(build_vae_dec_prog(c, B, p));
    *out = &p;
    return DTP_OK;
  }
  *out = &it->second;
  return DTP_OK;
}

extern "C" {

int dtp_vae_encode(dtp_ctx* ctx, const float* images, const float* eps, float* latent, int B, dtp_stream s_) {
  Ctx* c = (Ctx*)ctx;
  hipStream_t s = (hipStream_t)s_;
  if (!c || !c->finalized) { dtp_set_error("dtp_vae_encode: weights not finalized"); return DTP_ERR_STATE; }
  if (B < 1 || B > 2 * c->maxB) { dtp_set_error("dtp_vae_encode: batch %d outside 1..%d", B, 2 * c->maxB); return DTP_ERR_ARG; }
  HIP_CHECK(hipSetDevice(c->device));
  VaeEncProg* p;
  RC(get_enc_prog(c, B, &p));
  RC(dtp_launch_nchw_f32_to_nhwc_f16(images, p->in8, B, 3, c->R * c->R, 8, s));
  RC(p->main.run(s, 0));
  return launch_vae_sample(c, p->moments, eps, latent, B, 1.0f, s);
}

int dtp_vae_decode(dtp_ctx* ctx, const float* latent, float* images, int B, dtp_stream s_) {
  Ctx* c = (Ctx*)ctx;
  hipStream_t s = (hipStream_t)s_;
  if (!c || !c->finalized) { dtp_set_error("dtp_vae_decode: weights not finalized"); return DTP_ERR_STATE; }
  if (B < 1 || B > c->maxB) { dtp_set_error("dtp_vae_decode: batch %d outside 1..%d", B, c->maxB); return DTP_ERR_ARG; }
  HIP_CHECK(hipSetDevice(c->device));
  VaeDecProg* p;
  RC(get_dec_prog(c, B, &p));
  RC(launch_post_quant(c, latent, 0, 1.0f, p->in8, B, s));
  RC(p->main.run(s, 0));
  return launch_nhwc_f32_to_nchw(p->out32, images, B, 3, c->R * c->R, 4, s);
}

}  // extern "C"
