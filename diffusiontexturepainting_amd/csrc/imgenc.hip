// Brush (condition-patch) encoder on the libdtp kernels: crop/resize, 1+4+9 patch pyramid, CLIP
// ViT-B/32 tower, three 4-block patch-transformer stacks, projection -> [14,768] conditioning.
// Reference: trt_inference/trt_model.py:79-88 (set_brush), handler.py:36-45 (crop_resize_square),
// image_encoder.py:20-115 (ConditionPatchEncoder); the ViT tower is OpenAI CLIP ViT-B/32 with
// `visual.proj = None` (image_encoder.py:49-50), keyed here in HF CLIPVisionModel naming.
// Runs once per brush change; it reuses the GEMM / LayerNorm / attention kernels of the stamp path.
#include <math.h>

#include "engine.h"

namespace {

// torchvision CenterCrop(min side) + Resize(R): bilinear, align_corners=False, no antialias
__global__ void crop_resize_kernel(const float* __restrict__ img, int H, int W, float* __restrict__ out, int R) {
  const int m = min(H, W);
  // torchvision center_crop: int(round((H - m) / 2.0)) with Python's round-half-to-even
  const int top = (int)rintf((H - m) / 2.0f), left = (int)rintf((W - m) / 2.0f);
  const float scale = (float)m / (float)R;
  const int total = 3 * R * R;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int x = i % R, y = (i / R) % R, c = i / (R * R);
    const float* src = img + (size_t)c * H * W;
    float v;
    if (m == R) {
      v = src[(size_t)(top + y) * W + left + x];
    } else {
      float sy = fmaxf(scale * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(scale * (x + 0.5f) - 0.5f, 0.f);
      const int y0 = min((int)sy, m - 1), x0 = min((int)sx, m - 1);
      const int y1 = min(y0 + 1, m - 1), x1 = min(x0 + 1, m - 1);
      const float ly = sy - y0, lx = sx - x0;
      const float* r0 = src + (size_t)(top + y0) * W + left;
      const float* r1 = src + (size_t)(top + y1) * W + left;
      v = (1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1]);
    }
    out[i] = v;
  }
}

__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// F.interpolate(image, (224,224), mode="bicubic", align_corners=True) + CLIP mean/std (image_encoder.py:100-104)
__global__ void bicubic224_norm_kernel(const float* __restrict__ img, int R, float* __restrict__ out) {
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  const int total = 3 * 224 * 224;
  const float scale = (R > 1) ? (float)(R - 1) / 223.0f : 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int x = i % 224, y = (i / 224) % 224, c = i / (224 * 224);
    const float* src = img + (size_t)c * R * R;
    float v;
    if (R == 224) {
      v = src[y * 224 + x];
    } else {
      const float ry = scale * y, rx = scale * x;
      const int iy = (int)floorf(ry), ix = (int)floorf(rx);
      float wy[4], wx[4];
      cubic_coeffs(ry - iy, wy);
      cubic_coeffs(rx - ix, wx);
      v = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), R - 1);
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) row += wx[b] * src[(size_t)yy * R + min(max(ix - 1 + b, 0), R - 1)];
        v += wy[a] * row;
      }
    }
    out[i] = (v - mean[c]) / stdv[c];
  }
}

// 1+4+9 patch pyramid (image_encoder.py:34-40,106-113) written straight as the ViT patch-embedding GEMM
// operand: A[(img*49 + py*7 + px)][c*1024 + ky*32 + kx] (f16).  Tiles of 112 / 74 px are upsampled to 224
// with bilinear, align_corners=False (torchvision Resize on tensors).
__global__ void patchify_kernel(const float* __restrict__ img224, f16* __restrict__ A) {
  const int total = 14 * 3 * 224 * 224;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int x = i % 224, y = (i / 224) % 224, c = (i / (224 * 224)) % 3, n = i / (3 * 224 * 224);
    int side, t;  // tiles per side, tile index
    if (n == 0) { side = 1; t = 0; } else if (n < 5) { side = 2; t = n - 1; } else { side = 3; t = n - 5; }
    const int p = 224 / side;  // 224, 112, 74 (the last 2 rows/cols of the 3x3 split are dropped by unfold)
    const int ty = (t / side) * p, tx = (t % side) * p;
    const float* src = img224 + (size_t)c * 224 * 224;
    float v;
    if (side == 1) {
      v = src[y * 224 + x];
    } else {
      const float scale = (float)p / 224.0f;
      const float sy = fmaxf(scale * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(scale * (x + 0.5f) - 0.5f, 0.f);
      const int y0 = min((int)sy, p - 1), x0 = min((int)sx, p - 1);
      const int y1 = min(y0 + 1, p - 1), x1 = min(x0 + 1, p - 1);
      const float ly = sy - y0, lx = sx - x0;
      const float* r0 = src + (size_t)(ty + y0) * 224 + tx;
      const float* r1 = src + (size_t)(ty + y1) * 224 + tx;
      v = (1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1]);
    }
    A[(size_t)(n * 49 + (y >> 5) * 7 + (x >> 5)) * 3072 + c * 1024 + (y & 31) * 32 + (x & 31)] = (f16)v;
  }
}

// tokens[img][0] = class_embedding + pos[0]; tokens[img][1+i] = patch_embed[img*49+i] + pos[1+i]
__global__ void clip_tokens_kernel(const f16* __restrict__ pe, const float* __restrict__ cls_pos, const float* __restrict__ pos,
                                   f16* __restrict__ tok) {
  const int total = 14 * 50 * 768;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int d = i % 768, t = (i / 768) % 50, n = i / (768 * 50);
    tok[i] = (t == 0) ? (f16)cls_pos[d] : (f16)((float)pe[(size_t)(n * 49 + t - 1) * 768 + d] + pos[(t - 1) * 768 + d]);
  }
}

__global__ void add_table_kernel(f16* __restrict__ x, const float* __restrict__ table, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = (f16)((float)x[i] + table[i]);
}

__global__ void f16_to_f32_kernel(const f16* __restrict__ x, float* __restrict__ y, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = (float)x[i];
}

T rows_view(f16* p, int rows, int C, int ld) {
  T t;
  t.p = p; t.B = 1; t.H = 1; t.W = rows; t.C = C; t.ld = ld;
  return t;
}

}  // namespace

int load_imgenc_weights(Ctx* c) {
  ImgEncW& w = c->ienc;
  const std::string P = "clip.vision_model.";
  RC(load_linear(c, {P + "embeddings.patch_embedding"}, w.patch, false));  // [768][3*32*32]: conv k=c*1024+ky*32+kx
  std::vector<float> cls, pos;
  RC(ctx_fetch_host(c, P + "embeddings.class_embedding", cls));
  RC(ctx_fetch_host(c, P + "embeddings.position_embedding.weight", pos));
  std::vector<float> cls_pos(768);
  for (int d = 0; d < 768; ++d) cls_pos[d] = cls[d] + pos[d];
  RC(ctx_upload_f32(c, cls_pos, &w.cls_pos));
  RC(ctx_upload_f32(c, std::vector<float>(pos.begin() + 768, pos.end()), &w.pos));
  RC(load_norm(c, P + "pre_layrnorm", w.pre_ln));
  RC(load_norm(c, P + "post_layernorm", w.post_ln));
  for (int i = 0; i < 12; ++i) {
    const std::string L = P + "encoder.layers." + std::to_string(i);
    ClipLayerW& l = w.layers[i];
    RC(load_norm(c, L + ".layer_norm1", l.ln1));
    RC(load_norm(c, L + ".layer_norm2", l.ln2));
    RC(load_linear(c, {L + ".self_attn.q_proj", L + ".self_attn.k_proj", L + ".self_attn.v_proj"}, l.qkv, true));
    RC(load_linear(c, {L + ".self_attn.out_proj"}, l.out, true));
    RC(load_linear(c, {L + ".mlp.fc1"}, l.fc1, true));
    RC(load_linear(c, {L + ".mlp.fc2"}, l.fc2, true));
  }
  const char* scales = "lms";
  for (int s = 0; s < 3; ++s)
    for (int i = 0; i < 4; ++i) {
      const std::string B = std::string("penc.") + scales[s] + "_patch_encoder_layers." + std::to_string(i);
      PencBlockW& b = w.blocks[s][i];
      RC(load_norm(c, B + ".norm1", b.n1));
      RC(load_norm(c, B + ".norm3", b.n3));
      RC(load_linear(c, {B + ".attn1.to_q", B + ".attn1.to_k", B + ".attn1.to_v"}, b.qkv, true));
      RC(load_linear(c, {B + ".attn1.to_out.0"}, b.out, true));
      RC(load_linear(c, {B + ".ff.net.0.proj"}, b.ff1, true));
      RC(load_linear(c, {B + ".ff.net.2"}, b.ff2, true));
    }
  RC(load_norm(c, "penc.final_layer_norm", w.final_ln));
  RC(load_linear(c, {"penc.proj_out"}, w.proj_out, true));
  std::vector<float> unc;
  RC(ctx_fetch_host(c, "penc.uncond_vector", unc));
  RC(ctx_upload_f32(c, unc, &w.uncond));
  // pos_emb: positional_encoding_2d(768, s, s).view(1, n, 768) for n = 1, 4, 9 -- the raw CHW buffer
  // reinterpreted as [n][768] (image_encoder.py:20-31,54-56), reproduced bit-for-bit in layout
  std::vector<float> pe;
  for (int n : {1, 4, 9}) {
    const int s = (int)lround(sqrt((double)n)), ch = 768, d = ch / 2;
    std::vector<float> t((size_t)ch * s * s, 0.f);
    for (int k = 0; k < d / 2; ++k) {
      const float freq = 1.0f / powf(10000.0f, (float)(2 * k) / (float)d);
      for (int y = 0; y < s; ++y)
        for (int x = 0; x < s; ++x) {
          t[((size_t)(2 * k) * s + y) * s + x] = sinf(x * freq);
          t[((size_t)(2 * k + 1) * s + y) * s + x] = cosf(x * freq);
          t[((size_t)(d + 2 * k) * s + y) * s + x] = sinf(y * freq);
          t[((size_t)(d + 2 * k + 1) * s + y) * s + x] = cosf(y * freq);
        }
    }
    pe.insert(pe.end(), t.begin(), t.end());
  }
  RC(ctx_upload_f32(c, pe, &w.pos_emb));
  w.present = true;
  return DTP_OK;
}

static int build_ienc(Ctx* c, IencBufs& ib) {
  const ImgEncW& w = c->ienc;
  void* p;
  RC(ctx_persistent(c, 3 * 224 * 224 * 4, &p, true)); ib.img224 = (float*)p;
  RC(ctx_persistent(c, (size_t)768 * 3072 * 2, &p, true)); ib.patchA = (f16*)p;  // 686 rows used, padded to 768
  RC(ctx_persistent(c, 14 * 768 * 2, &p, true)); ib.out16 = (f16*)p;
  Builder b{c, &ib.prog};
  // ---- ViT-B/32 tower on the 14 patches
  T A = rows_view(ib.patchA, 14 * 49, 3072, 3072), pe;
  RC(b.linear(A, w.patch, nullptr, 0, pe));
  T tok = b.alloc(1, 1, 14 * 50, 768);
  if (!tok.p) return DTP_ERR_HIP;
  {
    const T pe2 = pe, tok2 = tok;
    const ImgEncW* ww = &c->ienc;
    b.push(PK_ELEM, 0, 0, [=](hipStream_t s, int) {
      hipLaunchKernelGGL(clip_tokens_kernel, dim3(1024), dim3(256), 0, s, pe2.p, ww->cls_pos, ww->pos, tok2.p);
      return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
    });
  }
  b.release(pe);
  T x;
  RC(b.ln(tok, w.pre_ln, x));
  b.release(tok);
  for (int i = 0; i < 12; ++i) {
    const ClipLayerW& l = w.layers[i];
    T h, qkv, a, x2, h2, f, x3;
    RC(b.ln(x, l.ln1, h));
    RC(b.linear(h, l.qkv, nullptr, 0, qkv));
    b.release(h);
    T q = qkv, k = qkv, v = qkv;
    q.C = k.C = v.C = 768;
    k.p += 768; v.p += 1536;
    RC(b.attention(q, k, v, 12, 50, 50, 14, a));
    b.release(qkv);
    RC(b.linear(a, l.out, &x, 0, x2));
    b.release(a); b.release(x);
    RC(b.ln(x2, l.ln2, h2));
    RC(b.linear(h2, l.fc1, nullptr, GF_QUICKGELU, f));
    b.release(h2);
    RC(b.linear(f, l.fc2, &x2, 0, x3));
    b.release(f); b.release(x2);
    x = x3;
  }
  // post-LN of the 14 class tokens (rows img*50), then + pos_emb
  T cls = rows_view(x.p, 14, 768, 50 * 768), feats;
  RC(b.ln(cls, w.post_ln, feats));
  b.release(x);
  {
    const T f2 = feats;
    const ImgEncW* ww = &c->ienc;
    b.push(PK_ELEM, 0, 0, [=](hipStream_t s, int) {
      hipLaunchKernelGGL(add_table_kernel, dim3(42), dim3(256), 0, s, f2.p, ww->pos_emb, 14 * 768);
      return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
    });
  }
  // ---- three independent 4-block stacks over tokens [0:1], [1:5], [5:14] (image_encoder.py:84-92)
  T catbuf = b.alloc(1, 1, 14, 768);
  if (!catbuf.p) return DTP_ERR_HIP;
  const int start[3] = {0, 1, 5}, count[3] = {1, 4, 9};
  for (int s = 0; s < 3; ++s) {
    T y = rows_view(feats.p + (size_t)start[s] * 768, count[s], 768, 768);
    bool own = false;
    for (int i = 0; i < 4; ++i) {
      const PencBlockW& k = w.blocks[s][i];
      T h, qkv, a, y2, h2, f, y3;
      RC(b.ln(y, k.n1, h));
      RC(b.linear(h, k.qkv, nullptr, 0, qkv));
      b.release(h);
      T q = qkv, kk = qkv, v = qkv;
      q.C = kk.C = v.C = 768;
      kk.p += 768; v.p += 1536;
      RC(b.attention(q, kk, v, 4, count[s], count[s], 1, a));
      b.release(qkv);
      RC(b.linear(a, k.out, &y, 0, y2));
      b.release(a);
      if (own) b.release(y);
      RC(b.ln(y2, k.n3, h2));
      RC(b.linear(h2, k.ff1, nullptr, GF_GELU, f));
      b.release(h2);
      if (i < 3) {
        RC(b.linear(f, k.ff2, &y2, 0, y3));
      } else {  // last block writes straight into its slice of the concatenated buffer
        GemmParams g = {};
        g.A = f.p; g.W = k.ff2.w; g.C = catbuf.p + (size_t)start[s] * 768; g.bias = k.ff2.b; g.R = y2.p; g.zero = c->zero;
        g.M = count[s]; g.N = 768; g.K = 3072; g.lda = f.ld; g.ldw = k.ff2.ldw; g.ldc = 768; g.ldr = y2.ld; g.nkb = 48;
        g.flags = GF_BIAS | GF_RESID; g.splits = 1; g.kb_per_split = 48;
        b.push(PK_GEMM0 + 3, 2.0 * g.M * 768.0 * 3072.0, 0, [=](hipStream_t st, int) { return dtp_launch_gemm(g, 3, st); });
        y3 = y2;
      }
      b.release(f);
      if (i < 3) b.release(y2);
      else b.release(y2);
      y = y3;
      own = true;
    }
  }
  b.release(feats);
  T fin, out;
  RC(b.ln(catbuf, w.final_ln, fin));
  b.release(catbuf);
  RC(b.linear(fin, w.proj_out, nullptr, 0, out));
  b.release(fin);
  {
    const T o2 = out;
    f16* dst = ib.out16;
    b.push(PK_ELEM, 0, 0, [=](hipStream_t s, int) {
      return hipMemcpyAsync(dst, o2.p, 14 * 768 * 2, hipMemcpyDeviceToDevice, s) == hipSuccess ? DTP_OK : DTP_ERR_HIP;
    });
  }
  b.release(out);
  RC(ensure_ws(c));
  ib.built = true;
  return DTP_OK;
}

extern "C" int dtp_set_brush(dtp_ctx* ctx, const float* image, int H, int W, float* image_out, dtp_stream s) {
  return dtp_set_brush_slot(ctx, 0, image, H, W, image_out, s);
}

extern "C" int dtp_set_brush_slot(dtp_ctx* ctx, int slot, const float* image, int H, int W, float* image_out, dtp_stream s_) {
  Ctx* c = (Ctx*)ctx;
  hipStream_t s = (hipStream_t)s_;
  if (slot < 0 || slot >= DTP_MAX_SLOTS) { dtp_set_error("dtp_set_brush: slot %d outside 0..%d", slot, DTP_MAX_SLOTS - 1); return DTP_ERR_ARG; }
  if (!c || !c->finalized) { dtp_set_error("dtp_set_brush: weights not finalized"); return DTP_ERR_STATE; }
  if (!c->ienc.present) { dtp_set_error("dtp_set_brush: image-encoder weights (clip.*, penc.*) were not loaded"); return DTP_ERR_STATE; }
  if (!image || H < 1 || W < 1) { dtp_set_error("dtp_set_brush: bad image"); return DTP_ERR_ARG; }
  HIP_CHECK(hipSetDevice(c->device));
  IencBufs& ib = c->ienc_bufs;
  if (!ib.built) RC(build_ienc(c, ib));
  const int R = c->R;
  float* const brush = c->brush32 + (size_t)slot * 3 * R * R;
  float* const cond = c->cond32 + (size_t)slot * 2 * 14 * 768;
  hipLaunchKernelGGL(crop_resize_kernel, dim3(1024), dim3(256), 0, s, image, H, W, brush, R);
  if (image_out) HIP_CHECK(hipMemcpyAsync(image_out, brush, (size_t)3 * R * R * 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(bicubic224_norm_kernel, dim3(588), dim3(256), 0, s, brush, R, ib.img224);
  hipLaunchKernelGGL(patchify_kernel, dim3(2048), dim3(256), 0, s, ib.img224, ib.patchA);
  RC(ib.prog.run(s, 0));
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(42), dim3(256), 0, s, ib.out16, cond, 14 * 768);
  HIP_CHECK(hipMemcpyAsync(cond + 14 * 768, c->ienc.uncond, 14 * 768 * 4, hipMemcpyDeviceToDevice, s));
  c->slot_set[slot] = true;
  c->slot_version[slot] = ++c->cond_version;
  return hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP;
}
