// The stamp: everything TRTConditionalInpainter.generate_raw / InpaintPipeline.infer do around the
// three networks, as fused HIP kernels + hipGraph replay.
// Reference: trt_inference/trt_model.py:90-121, handler.py:25-33,55-56, model_base.py:51-58,
// inpaint_pipeline.py:39-153, stable_diffusion_pipeline.py:340-355,407-484, utilities.py:370-529.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "engine.h"
#include <dlfcn.h>

// roctx ranges around the stages of a stamp (host side, like the reference's NVTX ranges: stable_diffusion_pipeline.py:358-366), so that
// a rocprofv3 --marker-trace carries the stage names.  Resolved at run time from the ROCm marker library: no link dependency, a no-op
// when the library is absent.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
      if (!h) continue;
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
const Roctx& roctx() { static Roctx r; return r; }
struct RoctxRange {
  explicit RoctxRange(const char* name) { if (roctx().push) roctx().push(name); }
  ~RoctxRange() { if (roctx().pop) roctx().pop(); }
};
}  // namespace

int get_unet_prog(Ctx* c, int N, int dupB, UNetProg** out);
int get_enc_prog(Ctx* c, int B, VaeEncProg** out);
int get_dec_prog(Ctx* c, int B, VaeDecProg** out);
int launch_vae_sample(Ctx* c, const float* mom, const float* eps, float* out, int B, float scale, hipStream_t s);
int launch_post_quant(Ctx* c, const float* z, int nhwc, float in_scale, f16* out, int B, hipStream_t s);
int load_imgenc_weights(Ctx* c);
void dtp_gemm_init();
void dtp_conv_halo_init();
void dtp_gemm_wide_init();
void dtp_gemm_fp8_init();

#define VAE_SCALE 0.18215f

// ---------------------------------------------------------------- DDIM tables (host, fp32 like torch)
// utilities.py:383-388 (betas, cumprod), :432-439 (timesteps), :416 (gather), :397 (final alpha).
extern "C" int dtp_ddim_tables(int steps, int64_t* timesteps, float* alphas, float* final_alpha) {
  // the largest timestep is (steps-1)*(1000/steps) + 1: steps = 1000 would index alphas_cumprod[1000] (the reference raises
  // IndexError there, utilities.py:416)
  if (steps < 1 || steps > 999) { dtp_set_error("ddim: steps %d outside 1..999", steps); return DTP_ERR_ARG; }
  const int T = 1000;
  static float full[1000];
  static bool have = false;
  if (!have) {
    const float start = (float)sqrt(0.00085), end = (float)sqrt(0.012);
    const float step = (end - start) / (float)(T - 1);
    double acc = 1.0;  // torch's CPU cumprod accumulates float in double and rounds every output
    for (int i = 0; i < T; ++i) {
      const float l = (i < T / 2) ? start + step * (float)i : end - step * (float)(T - 1 - i);
      const float beta = l * l;
      acc *= (double)(1.0f - beta);
      full[i] = (float)acc;
    }
    have = true;
  }
  const int ratio = T / steps;
  for (int i = 0; i < steps; ++i) {
    const int64_t t = (int64_t)(steps - 1 - i) * ratio + 1;
    if (timesteps) timesteps[i] = t;
    if (alphas) alphas[i] = full[t];
  }
  if (final_alpha) *final_alpha = full[0];
  return DTP_OK;
}

// ---------------------------------------------------------------- kernels
namespace {

// separable flat dilation (kornia.morphology.dilation with ones(pad,pad), geodesic border):
// out[i] = max over [i - pad/2, i + pad - pad/2 - 1] clipped to the image.
__global__ void dilate_row_kernel(const float* __restrict__ canvas, float* __restrict__ tmp, int B, int R, int lo, int hi) {
  const long long total = (long long)B * R * R;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % R);
    const long long by = i / R;
    const int y = (int)(by % R), b = (int)(by / R);
    const float* a = canvas + ((size_t)b * 4 + 3) * R * R + (size_t)y * R;
    float m = -1e4f;
    for (int xx = max(0, x - lo); xx <= min(R - 1, x + hi); ++xx) m = fmaxf(m, a[xx]);
    tmp[i] = m;
  }
}
__global__ void dilate_col_kernel(const float* __restrict__ tmp, float* __restrict__ out, int B, int R, int lo, int hi) {
  const long long total = (long long)B * R * R;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % R);
    const long long by = i / R;
    const int y = (int)(by % R), b = (int)(by / R);
    const float* a = tmp + (size_t)b * R * R + x;
    float m = -1e4f;
    for (int yy = max(0, y - lo); yy <= min(R - 1, y + hi); ++yy) m = fmaxf(m, a[(size_t)yy * R]);
    out[i] = m;
  }
}

// trt_model.py:103-109 + handler.py:25-33: canvas -> VAE-encoder inputs (NHWC f16, 8 channels,
// batch [masked x B | context x B]) and the two latent-resolution masks (nearest, 1 = paint).
__global__ void prep_kernel(const float* __restrict__ canvas, const float* __restrict__ brush_slots, const int* __restrict__ slot_map,
                            const float* __restrict__ dil, f16* __restrict__ enc_in, float* __restrict__ masks, int B, int R) {
  const int HW = R * R, h = R / 8;
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / HW), pix = (int)(i - (long long)b * HW);
    const float* cb = canvas + (size_t)b * 4 * HW;
    const float* brush = brush_slots + (size_t)slot_map[b] * 3 * HW;  // this stamp's brush (hint image source)
    const float a = cb[3 * HW + pix];
    const float hint = 1.0f - dil[i];
    f16x8 m8, c8;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float img = cb[ch * HW + pix] * 2.0f - 1.0f;
      const float masked = img * a;
      const float src = brush[ch * HW + pix] * 2.0f - 1.0f;
      m8[ch] = (f16)masked;
      c8[ch] = (f16)(masked + src * hint);
    }
#pragma unroll
    for (int ch = 3; ch < 8; ++ch) m8[ch] = c8[ch] = (f16)0.f;
    *(f16x8*)(enc_in + i * 8) = m8;
    *(f16x8*)(enc_in + ((size_t)B * HW + i) * 8) = c8;
    const int y = pix / R, x = pix - y * R;
    if ((y & 7) == 0 && (x & 7) == 0) {  // F.interpolate(size=(h,w)) default 'nearest': src = dst * 8
      const size_t o = (size_t)b * h * h + (size_t)(y >> 3) * h + (x >> 3);
      masks[o] = 1.0f - a;
      masks[(size_t)B * h * h + o] = 1.0f - fminf(fmaxf(a + hint, 0.f), 1.f);
    }
  }
}

// UNet input assembly (inpaint_pipeline.py:116,136; sdp:423-427): branch-major [uncond|cond|tg] x B.
__global__ void assemble_kernel(const float* __restrict__ lat_nchw, const float* __restrict__ masks,
                                const float* __restrict__ ml, f16* __restrict__ in16, float* __restrict__ x32, int B, int HWl,
                                int NB) {
  const long long total = (long long)B * HWl;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / HWl), p = (int)(i - (long long)b * HWl);
    float x[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (lat_nchw) {
        x[ch] = lat_nchw[((size_t)b * 4 + ch) * HWl + p];  // * init_noise_sigma (= 1.0)
        x32[i * 4 + ch] = x[ch];
      } else {
        x[ch] = x32[i * 4 + ch];  // mid-loop switch to the 2-branch program: keep the running latent
      }
    }
    for (int br = 0; br < NB; ++br) {
      const int src = (br < 2) ? b : B + b;  // branches 0,1: masked image; branch 2: context image
      f16x8 lo, hi;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) lo[ch] = (f16)x[ch];
      lo[4] = (f16)masks[(size_t)src * HWl + p];
      lo[5] = (f16)ml[((size_t)src * 4 + 0) * HWl + p];
      lo[6] = (f16)ml[((size_t)src * 4 + 1) * HWl + p];
      lo[7] = (f16)ml[((size_t)src * 4 + 2) * HWl + p];
      hi[0] = (f16)ml[((size_t)src * 4 + 3) * HWl + p];
#pragma unroll
      for (int ch = 1; ch < 8; ++ch) hi[ch] = (f16)0.f;
      f16* o = in16 + ((size_t)(br * B + b) * HWl + p) * 16;
      *(f16x8*)o = lo;
      *(f16x8*)(o + 8) = hi;
    }
  }
}

// guidance combine + DDIM eta=0 step + refresh of the latent channels of the UNet input
// (sdp:419-420,449-455; utilities.py:463-503).  params: [0]=cfg [1]=tg [2]=tg_steps, then 4 per step.
__global__ void step_kernel(const float* __restrict__ eps_out, float* __restrict__ x32, f16* __restrict__ in16,
                            const float* __restrict__ params, int step_index, int B, int HWl, int NB) {
  const float cfg = params[0];
  const float tg = ((float)step_index > params[2] - 1.0f) ? 0.f : params[1];
  const float* k = params + 4 + 4 * step_index;
  const float sqrt_beta_t = k[0], sqrt_alpha_t = k[1], sqrt_alpha_prev = k[2], sqrt_beta_prev = k[3];
  const long long total = (long long)B * HWl * 4;
  const size_t bs = (size_t)B * HWl * 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const float u = eps_out[i], c = eps_out[bs + i];
    float e = u + cfg * (c - u);
    if (NB == 3) e += tg * (eps_out[2 * bs + i] - c);
    const float x = x32[i];
    const float x0 = (x - sqrt_beta_t * e) / sqrt_alpha_t;
    const float xn = sqrt_alpha_prev * x0 + sqrt_beta_prev * e;
    x32[i] = xn;
    const long long pix = i >> 2;
    const int ch = (int)(i & 3);
    for (int br = 0; br < NB; ++br) in16[((size_t)br * B * HWl + pix) * 16 + ch] = (f16)xn;
  }
}

// inpaint_pipeline.py:148 clamp, optional alpha composite (model_base.py:56-58) and the truncating
// u8 conversion (handler.py:55-56).
__global__ void finish_kernel(const float* __restrict__ dec, const float* __restrict__ canvas, void* __restrict__ out, int B,
                              int HW, int composite, int u8) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / HW), pix = (int)(i - (long long)b * HW);
    const float a = composite ? canvas[((size_t)b * 4 + 3) * HW + pix] : 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float v = fminf(fmaxf(dec[i * 4 + ch] / 2.0f + 0.5f, 0.f), 1.f);
      if (composite) v = canvas[((size_t)b * 4 + ch) * HW + pix] * a + v * (1.0f - a);
      if (u8) ((unsigned char*)out)[i * 3 + ch] = (unsigned char)(v * 255.0f);
      else ((float*)out)[((size_t)b * 3 + ch) * HW + pix] = v;
    }
  }
}

// ctx16[n][14][768]: branch 0 <- uncond, branches 1.. <- cond  (inpaint_pipeline.py:140), each from the stamp's own slot
__global__ void build_ctx_kernel(const float* __restrict__ cond_slots, const int* __restrict__ slot_map, f16* __restrict__ ctx16, int B,
                                 int NB) {
  const int per = 14 * 768;
  const long long total = (long long)NB * B * per;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / per), j = (int)(i - (long long)n * per);
    const int br = n / B, b = n - br * B;
    ctx16[i] = (f16)cond_slots[(size_t)slot_map[b] * 2 * per + (br == 0 ? per : 0) + j];
  }
}

struct SlotArgs { int s[64]; };
// the per-stamp slot ids travel as a kernel argument (no host staging buffer to keep alive, no sync)
__global__ void set_slots_kernel(int* __restrict__ dst, SlotArgs a, int B) {
  if ((int)threadIdx.x < B) dst[threadIdx.x] = a.s[threadIdx.x];
}

// cfg / tg / tg_steps travel as kernel ARGUMENTS into the device parameter block the captured step kernels read: no host
// staging buffer, so dtp_stamp never has to wait for the stream
__global__ void set_header_kernel(float* __restrict__ params, float cfg, float tg, float tg_steps) {
  if (threadIdx.x == 0) { params[0] = cfg; params[1] = tg; params[2] = tg_steps; params[3] = 0.f; }
}

// post-loop finiteness guard (the reference asserts `not isnan` after every step, stable_diffusion_pipeline.py:415, at the
// price of a host sync per step; here ONE pass over the final latents and the decoded image, debug option "check_finite")
__global__ void finite_check_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b, long long nb,
                                    int* __restrict__ flag) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (long long)gridDim.x * 256) {
    const float v = i < na ? a[i] : b[i - na];
    bad |= !(fabsf(v) <= 3.0e38f);  // false for NaN and +-inf
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

inline int nblk(long long total) { return (int)std::min<long long>((total + 255) / 256, 4096); }

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? DTP_OK : DTP_ERR_HIP)

int stamp_init(Ctx* c) {
  void* p;
  const size_t hw = (size_t)c->h * c->h, RR = (size_t)c->R * c->R;
  RC(ctx_persistent(c, c->maxB * hw * 4 * 4, &p, true)); c->x32 = (float*)p;
  RC(ctx_persistent(c, c->maxB * 4 * RR * 4, &p, true)); c->canvas32 = (float*)p;
  RC(ctx_persistent(c, 2 * c->maxB * RR * 4, &p, true)); c->alpha_tmp = (float*)p;
  RC(ctx_persistent(c, (4 + 4 * 1000) * 4, &p, true)); c->stamp_params = (float*)p;
  RC(ctx_persistent(c, (size_t)DTP_MAX_SLOTS * 2 * 14 * 768 * 4, &p, true)); c->cond32 = (float*)p;
  RC(ctx_persistent(c, (size_t)DTP_MAX_SLOTS * 3 * RR * 4, &p, true)); c->brush32 = (float*)p;
  RC(ctx_persistent(c, 64 * sizeof(int), &p, true)); c->slot_map = (int*)p;
  RC(ctx_persistent(c, 256, &p, true)); c->finite_flag = (int*)p;
  return DTP_OK;
}

static int get_bufs(Ctx* c, int B, StampBufs** out) {
  auto it = c->stamp_bufs.find(B);
  if (it == c->stamp_bufs.end()) {
    StampBufs sb;
    void* p;
    const size_t hw = (size_t)c->h * c->h;
    RC(ctx_persistent(c, 2 * B * hw * 4, &p, true)); sb.masks = (float*)p;
    RC(ctx_persistent(c, 2 * B * 4 * hw * 4, &p, true)); sb.ml = (float*)p;
    RC(ctx_persistent(c, B * 4 * hw * 4, &p, true)); sb.lat = (float*)p;
    RC(ctx_persistent(c, 2 * B * 4 * hw * 4, &p, true)); sb.eps = (float*)p;
    it = c->stamp_bufs.emplace(B, sb).first;
  }
  *out = &it->second;
  return DTP_OK;
}

// run `body` on stream s, replaying a captured hipGraph when possible
template <class F>
static int run_stage(Ctx* c, long long key, hipStream_t s, F body) {
  if (!c->use_graph || c->profile || s == nullptr) return body(s);
  auto it = c->graphs.find(key);
  if (it == c->graphs.end()) {
    StampGraph g;
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = body(s);
    hipError_t e = hipStreamEndCapture(s, &g.graph);
    if (rc != DTP_OK) { if (g.graph) (void)hipGraphDestroy(g.graph); return rc; }
    if (e != hipSuccess) { dtp_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return DTP_ERR_HIP; }
    size_t n = 0;
    (void)hipGraphGetNodes(g.graph, nullptr, &n);
    g.nodes = (int)n;
    HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    it = c->graphs.emplace(key, g).first;
  }
  c->last_nodes += it->second.nodes;
  HIP_CHECK(hipGraphLaunch(it->second.exec, s));
  return DTP_OK;
}

extern "C" {

// kernel-level entry point: the separable flat dilation of add_extra_context (handler.py:28-29) on the alpha plane of a canvas
int dtp_op_dilate(const float* canvas, float* tmp, float* out, int B, int R, int pad, dtp_stream s_) {
  if (!canvas || !tmp || !out || B < 1 || R < 1 || pad < 1) { dtp_set_error("dtp_op_dilate: bad argument"); return DTP_ERR_ARG; }
  hipStream_t s = (hipStream_t)s_;
  const int lo = pad / 2, hi = pad - pad / 2 - 1;
  hipLaunchKernelGGL(dilate_row_kernel, dim3(nblk((long long)B * R * R)), dim3(256), 0, s, canvas, tmp, B, R, lo, hi);
  hipLaunchKernelGGL(dilate_col_kernel, dim3(nblk((long long)B * R * R)), dim3(256), 0, s, tmp, out, B, R, lo, hi);
  return LAUNCH_OK();
}

int dtp_finalize_weights(dtp_ctx* ctx) {
  Ctx* c = (Ctx*)ctx;
  if (!c) { dtp_set_error("dtp_finalize_weights: null handle"); return DTP_ERR_ARG; }
  if (c->finalized) { dtp_set_error("dtp_finalize_weights: already finalized"); return DTP_ERR_STATE; }
  HIP_CHECK(hipSetDevice(c->device));
  dtp_gemm_init();
  dtp_conv_halo_init();
  dtp_gemm_wide_init();
  dtp_gemm_fp8_init();
  dtp_xattn_init();
  dtp_lnlin_init();
  dtp_xchain_init();
  dtp_ffchain_init();
  dtp_conv_ws_init();
  dtp_gemm_ws_init();
  RC(load_unet_weights(c));
  RC(load_vae_weights(c));
  bool has_clip = false;
  for (auto& kv : c->staged)
    if (kv.first.rfind("clip.", 0) == 0) { has_clip = true; break; }
  if (has_clip) RC(load_imgenc_weights(c));
  HIP_CHECK(hipDeviceSynchronize());
  for (auto& s : c->staged) (void)hipFree(s.second.d);
  c->staged.clear();
  RC(stamp_init(c));
  c->finalized = true;
  return DTP_OK;
}

int dtp_set_conditioning_slot(dtp_ctx* ctx, int slot, const float* cond, const float* uncond, const float* brush, dtp_stream s_) {
  Ctx* c = (Ctx*)ctx;
  hipStream_t s = (hipStream_t)s_;
  if (!c || !c->finalized || !cond || !uncond || !brush) { dtp_set_error("dtp_set_conditioning: bad state/argument"); return DTP_ERR_STATE; }
  if (slot < 0 || slot >= DTP_MAX_SLOTS) { dtp_set_error("dtp_set_conditioning: slot %d outside 0..%d", slot, DTP_MAX_SLOTS - 1); return DTP_ERR_ARG; }
  HIP_CHECK(hipSetDevice(c->device));
  float* dst = c->cond32 + (size_t)slot * 2 * 14 * 768;
  HIP_CHECK(hipMemcpyAsync(dst, cond, 14 * 768 * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(dst + 14 * 768, uncond, 14 * 768 * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(c->brush32 + (size_t)slot * 3 * c->R * c->R, brush, (size_t)3 * c->R * c->R * 4, hipMemcpyDeviceToDevice, s));
  c->slot_set[slot] = true;
  c->slot_version[slot] = ++c->cond_version;
  return DTP_OK;
}

int dtp_set_conditioning(dtp_ctx* ctx, const float* cond, const float* uncond, const float* brush, dtp_stream s) {
  return dtp_set_conditioning_slot(ctx, 0, cond, uncond, brush, s);
}

int dtp_get_conditioning_slot(dtp_ctx* ctx, int slot, float* cond, float* uncond, dtp_stream s_) {
  Ctx* c = (Ctx*)ctx;
  hipStream_t s = (hipStream_t)s_;
  if (!c || slot < 0 || slot >= DTP_MAX_SLOTS || !c->slot_set[slot]) { dtp_set_error("dtp_get_conditioning: no brush set in slot %d", slot); return DTP_ERR_STATE; }
  const float* src = c->cond32 + (size_t)slot * 2 * 14 * 768;
  HIP_CHECK(hipMemcpyAsync(cond, src, 14 * 768 * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(uncond, src + 14 * 768, 14 * 768 * 4, hipMemcpyDeviceToDevice, s));
  return DTP_OK;
}

int dtp_get_conditioning(dtp_ctx* ctx, float* cond, float* uncond, dtp_stream s) { return dtp_get_conditioning_slot(ctx, 0, cond, uncond, s); }

int dtp_stamp(dtp_ctx* ctx, const float* canvas, const dtp_settings* st, const float* latents, const float* vae_eps,
              void* out, int B, dtp_stream s) {
  return dtp_stamp_slots(ctx, canvas, st, latents, vae_eps, out, B, nullptr, s);
}

int dtp_stamp_slots(dtp_ctx* ctx, const float* canvas, const dtp_settings* st, const float* latents, const float* vae_eps,
                    void* out, int B, const int* slots, dtp_stream s_) {
  Ctx* c = (Ctx*)ctx;
  hipStream_t s = (hipStream_t)s_;
  if (!c || !c->finalized) { dtp_set_error("dtp_stamp: weights not finalized"); return DTP_ERR_STATE; }
  if (!canvas || !st || !latents || !out || B < 1 || B > c->maxB) { dtp_set_error("dtp_stamp: bad argument (B=%d, max %d)", B, c->maxB); return DTP_ERR_ARG; }
  SlotArgs sa = {};
  for (int b = 0; b < B; ++b) {
    const int sl = slots ? slots[b] : 0;
    if (sl < 0 || sl >= DTP_MAX_SLOTS) { dtp_set_error("dtp_stamp: slot %d of stamp %d outside 0..%d", sl, b, DTP_MAX_SLOTS - 1); return DTP_ERR_ARG; }
    if (!c->slot_set[sl]) { dtp_set_error("dtp_stamp: no brush set in slot %d (call dtp_set_brush / dtp_set_conditioning)", sl); return DTP_ERR_STATE; }
    sa.s[b] = sl;
  }
  if (st->steps < 2 || st->steps > 999) { dtp_set_error("dtp_stamp: steps=%d outside 2..999", st->steps); return DTP_ERR_ARG; }
  if (st->context_pad < 1) { dtp_set_error("dtp_stamp: context_pad must be >= 1"); return DTP_ERR_ARG; }
  HIP_CHECK(hipSetDevice(c->device));
  const int R = c->R, h = c->h, HW = R * R, HWl = h * h;
  const int steps = st->steps, E = steps - 1;
  // the third (texture-guided) branch contributes nothing once its coefficient is 0: skip it (bit-identical)
  const int tg_evals = (st->tg_weight == 0.0f) ? 0 : std::max(0, std::min(E, st->tg_steps));

  UNetProg *u3 = nullptr, *u2 = nullptr;
  VaeEncProg* enc;
  VaeDecProg* dec;
  StampBufs* sb;
  // branches 0 (uncond) and 1 (cond) see identical samples: the programs evaluate the UNet prefix once for both (unet.hip, struct Dup)
  if (tg_evals > 0) RC(get_unet_prog(c, 3 * B, B, &u3));
  if (tg_evals < E) RC(get_unet_prog(c, 2 * B, B, &u2));
  RC(get_enc_prog(c, 2 * B, &enc));
  RC(get_dec_prog(c, B, &dec));
  RC(get_bufs(c, B, &sb));

  // ---- schedule tables (update_infer_settings, inpaint_pipeline.py:39-50): rebuilt when the step count changes
  if (c->sched_steps != steps) {  // rare (a settings change): the only host-blocking part of dtp_stamp
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int64_t> ts(steps);
    std::vector<float> al(steps);
    float fin;
    RC(dtp_ddim_tables(steps, ts.data(), al.data(), &fin));
    std::vector<float> tsf;
    for (int i = 1; i < steps; ++i) tsf.push_back((float)ts[i]);  // timesteps[1:] (sdp:348-355)
    RC(ensure_temb(c, tsf));
    std::vector<float> k(4 * E);
    for (int i = 0; i < E; ++i) {
      const int idx = 1 + i;
      const float a_t = al[idx], a_prev = (idx + 1 < steps) ? al[idx + 1] : fin;
      k[4 * i + 0] = sqrtf(1.0f - a_t);
      k[4 * i + 1] = sqrtf(a_t);
      k[4 * i + 2] = sqrtf(a_prev);
      k[4 * i + 3] = sqrtf(1.0f - a_prev);
    }
    HIP_CHECK(hipMemcpy(c->stamp_params + 4, k.data(), k.size() * 4, hipMemcpyHostToDevice));
    c->sched_steps = steps;
  }
  hipLaunchKernelGGL(set_header_kernel, dim3(1), dim3(64), 0, s, c->stamp_params, st->cfg_weight, st->tg_weight, (float)st->tg_steps);
  hipLaunchKernelGGL(set_slots_kernel, dim3(1), dim3(64), 0, s, c->slot_map, sa, B);

  // ---- cross-attention K/V for the current brush
  for (UNetProg* up : {u3, u2}) {
    if (!up) continue;
    const int NB = up->N / B;
    // the cached per-stamp matrices are valid for exactly this (B, NB) split, these slots and these slot versions
    bool valid = up->kv_ver != 0 && up->kv_B == B && up->kv_NB == NB && (int)up->kv_slots.size() == B;
    for (int b = 0; valid && b < B; ++b) valid = up->kv_slots[b] == sa.s[b] && up->kv_slot_ver[b] == c->slot_version[sa.s[b]];
    if (!valid) {
      hipLaunchKernelGGL(build_ctx_kernel, dim3(nblk((long long)up->N * 14 * 768)), dim3(256), 0, s, c->cond32, c->slot_map, up->ctx16, B, NB);
      RC(up->kv.run(s, 0));
      up->kv_ver = c->cond_version; up->kv_B = B; up->kv_NB = NB;
      up->kv_slots.assign(sa.s, sa.s + B);
      up->kv_slot_ver.resize(B);
      for (int b = 0; b < B; ++b) up->kv_slot_ver[b] = c->slot_version[sa.s[b]];
    }
  }

  c->last_nodes = 0;
  c->last_evals = E;
  RoctxRange whole("dtp_stamp");
  HIP_CHECK(hipEventRecord(c->ev[0], s));
  // ---- stage 0: pre-processing + both VAE encodes (one batch-2B pass)
  {
  RoctxRange r0("dtp_stamp: pre-processing + vae_encoder x2");
  HIP_CHECK(hipMemcpyAsync(sb->lat, latents, (size_t)B * 4 * HWl * 4, hipMemcpyDeviceToDevice, s));
  if (vae_eps) HIP_CHECK(hipMemcpyAsync(sb->eps, vae_eps, (size_t)2 * B * 4 * HWl * 4, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipMemcpyAsync(c->canvas32, canvas, (size_t)B * 4 * HW * 4, hipMemcpyDeviceToDevice, s));
  const int lo = st->context_pad / 2, hi = st->context_pad - st->context_pad / 2 - 1;
  hipLaunchKernelGGL(dilate_row_kernel, dim3(nblk((long long)B * HW)), dim3(256), 0, s, c->canvas32, c->alpha_tmp, B, R, lo, hi);
  hipLaunchKernelGGL(dilate_col_kernel, dim3(nblk((long long)B * HW)), dim3(256), 0, s, c->alpha_tmp,
                     c->alpha_tmp + (size_t)c->maxB * HW, B, R, lo, hi);
  const int first_nb = tg_evals > 0 ? 3 : 2;
  UNetProg* first = tg_evals > 0 ? u3 : u2;
  RC(run_stage(c, ((long long)B << 32) | (vae_eps ? 2 : 0) | (first_nb == 3 ? 1 : 0) | (1LL << 60), s, [&](hipStream_t q) -> int {
    hipLaunchKernelGGL(prep_kernel, dim3(nblk((long long)B * HW)), dim3(256), 0, q, c->canvas32, c->brush32, c->slot_map,
                       c->alpha_tmp + (size_t)c->maxB * HW, enc->in8, sb->masks, B, R);
    RC(enc->main.run(q, 0));
    RC(launch_vae_sample(c, enc->moments, vae_eps ? sb->eps : nullptr, sb->ml, 2 * B, VAE_SCALE, q));
    hipLaunchKernelGGL(assemble_kernel, dim3(nblk((long long)B * HWl)), dim3(256), 0, q, sb->lat, sb->masks, sb->ml, first->in16,
                       c->x32, B, HWl, first_nb);
    return LAUNCH_OK();
  }));
  }
  HIP_CHECK(hipEventRecord(c->ev[1], s));
  // fp8 (configs[4]): the first stamp of a program measures its activation ranges once, before the loop is captured
  if ((c->fp8_linear || c->fp8_attention)) {
    if (u3 && !u3->fp8_calibrated) RC(fp8_calibrate(c, u3, s, 0));
    if (u2 && !u2->fp8_calibrated) {
      if (tg_evals > 0)  // (u2 is not the first program:) its input is normally assembled where the loop switches programs: do it now, from the initial latents
        hipLaunchKernelGGL(assemble_kernel, dim3(nblk((long long)B * HWl)), dim3(256), 0, s, (const float*)nullptr, sb->masks, sb->ml, u2->in16,
                           c->x32, B, HWl, 2);
      RC(fp8_calibrate(c, u2, s, 0));
    }
  }
  // ---- stage 1: the denoise loop
  {
  RoctxRange r1("dtp_stamp: denoise loop (unet)");
  RC(run_stage(c, ((long long)B << 32) | ((long long)steps << 12) | tg_evals | (2LL << 60), s, [&](hipStream_t q) -> int {
    for (int i = 0; i < E; ++i) {
      UNetProg* up = (i < tg_evals) ? u3 : u2;
      const int NB = (i < tg_evals) ? 3 : 2;
      if (i == tg_evals && i > 0) {
        // switching to the 2-branch program: its input needs mask/masked-latent channels + current x
        hipLaunchKernelGGL(assemble_kernel, dim3(nblk((long long)B * HWl)), dim3(256), 0, q, (const float*)nullptr, sb->masks,
                           sb->ml, up->in16, c->x32, B, HWl, 2);
      }
      RC(up->main.run(q, i));
      hipLaunchKernelGGL(step_kernel, dim3(nblk((long long)B * HWl * 4)), dim3(256), 0, q, up->out32, c->x32, up->in16,
                         c->stamp_params, i, B, HWl, NB);
    }
    return LAUNCH_OK();
  }));
  }
  HIP_CHECK(hipEventRecord(c->ev[2], s));
  // ---- stage 2: latents / 0.18215 -> VAE decode -> clamp (+ composite, u8)
  {
  RoctxRange r2("dtp_stamp: vae decode + post-processing");
  RC(run_stage(c, ((long long)B << 32) | (3LL << 60), s, [&](hipStream_t q) -> int {
    RC(launch_post_quant(c, c->x32, 1, 1.0f / VAE_SCALE, dec->in8, B, q));
    return dec->main.run(q, 0);
  }));
  hipLaunchKernelGGL(finish_kernel, dim3(nblk((long long)B * HW)), dim3(256), 0, s, dec->out32, c->canvas32, out, B, HW,
                     st->composite, st->output_u8);
  }
  c->finite_pending = c->check_finite;
  if (c->check_finite) {
    HIP_CHECK(hipMemsetAsync(c->finite_flag, 0, sizeof(int), s));
    hipLaunchKernelGGL(finite_check_kernel, dim3(nblk((long long)B * HW * 4)), dim3(256), 0, s, c->x32, (long long)B * HWl * 4, dec->out32,
                       (long long)B * HW * 4, c->finite_flag);
  }
  HIP_CHECK(hipEventRecord(c->ev[3], s));
  return LAUNCH_OK();
}

int dtp_last_stamp_finite(dtp_ctx* ctx, int* finite) {
  Ctx* c = (Ctx*)ctx;
  if (!c || !finite) return DTP_ERR_ARG;
  if (!c->finite_pending) { dtp_set_error("dtp_last_stamp_finite: the last stamp ran without the \"check_finite\" option"); return DTP_ERR_STATE; }
  HIP_CHECK(hipEventSynchronize(c->ev[3]));
  int flag = 0;
  HIP_CHECK(hipMemcpy(&flag, c->finite_flag, sizeof(int), hipMemcpyDeviceToHost));
  *finite = flag ? 0 : 1;
  return DTP_OK;
}

int dtp_last_stamp_times(dtp_ctx* ctx, float ms[3]) {
  Ctx* c = (Ctx*)ctx;
  if (!c || !ms) return DTP_ERR_ARG;
  HIP_CHECK(hipEventSynchronize(c->ev[3]));
  for (int i = 0; i < 3; ++i) HIP_CHECK(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
  return DTP_OK;
}

int dtp_profile(dtp_ctx* ctx, int enable) {
  Ctx* c = (Ctx*)ctx;
  if (!c) return DTP_ERR_ARG;
  HIP_CHECK(hipDeviceSynchronize());
  for (ProfRec& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  c->prof.clear();
  c->profile = enable != 0;
  return DTP_OK;
}

int dtp_profile_rows(dtp_ctx* ctx, dtp_prof_row* rows, int max_rows, int* n_rows) {
  Ctx* c = (Ctx*)ctx;
  if (!c || !rows || !n_rows) return DTP_ERR_ARG;
  HIP_CHECK(hipDeviceSynchronize());
  dtp_prof_row acc[PK_COUNT] = {};  // PK_COUNT kinds, see include/dtp.h
  for (int k = 0; k < PK_COUNT; ++k) acc[k].kind = k;
  for (const ProfRec& r : c->prof) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
    dtp_prof_row& a = acc[r.kind];
    a.launches += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
  }
  int n = 0;
  for (int k = 0; k < PK_COUNT && n < max_rows; ++k)
    if (acc[k].launches) rows[n++] = acc[k];
  *n_rows = n;
  return DTP_OK;
}

int dtp_profile_dump(dtp_ctx* ctx, const char* path) {
  Ctx* c = (Ctx*)ctx;
  if (!c || !path) return DTP_ERR_ARG;
  HIP_CHECK(hipDeviceSynchronize());
  FILE* f = fopen(path, "w");
  if (!f) { dtp_set_error("dtp_profile_dump: cannot open %s", path); return DTP_ERR_ARG; }
  fprintf(f, "kind,us,tflops,algo_GBps,label\n");
  for (const ProfRec& r : c->prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    fprintf(f, "%d,%.2f,%.1f,%.1f,%s\n", r.kind, ms * 1e3, r.flops / (ms * 1e-3) / 1e12, r.bytes / (ms * 1e-3) / 1e9, r.label ? r.label : "");
  }
  fclose(f);
  return DTP_OK;
}

int dtp_set_option(dtp_ctx* ctx, const char* name, int value) {
  Ctx* c = (Ctx*)ctx;
  if (!c || !name) return DTP_ERR_ARG;
  if (!strcmp(name, "use_graph")) { c->use_graph = value != 0; return DTP_OK; }
  if (!strcmp(name, "autotune")) { c->autotune = value != 0; return DTP_OK; }
  if (!strcmp(name, "check_finite")) { c->check_finite = value != 0; return DTP_OK; }
  if (!strcmp(name, "fuse_gn_conv")) {
#ifndef DTP_EXPERIMENTAL
    if (value) { dtp_set_error("dtp_set_option: fuse_gn_conv is an experiment (slower: DESIGN.md 3.6) -- build with DTP_EXPERIMENTAL=1"); return DTP_ERR_ARG; }
#endif
    if (!c->unet_progs.empty() || !c->enc_progs.empty() || !c->dec_progs.empty()) {
      dtp_set_error("dtp_set_option: fuse_gn_conv must be chosen before the first launch program is built");
      return DTP_ERR_STATE;
    }
    c->fuse_gn_conv = value != 0;
    return DTP_OK;
  }
  if (!strcmp(name, "dedupe_prefix")) {  // programs are keyed by it: switching only affects which (cached) program a stamp uses
    c->dedupe_prefix = value != 0;
    for (auto& g : c->graphs) {  // captured stages hold the old program's launches
      if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
      if (g.second.graph) (void)hipGraphDestroy(g.second.graph);
    }
    c->graphs.clear();
    return DTP_OK;
  }
  if (!strcmp(name, "fp8_linear")) {
    if (!c->unet_progs.empty() && c->fp8_linear != (value != 0)) {
      dtp_set_error("dtp_set_option: fp8_linear must be chosen before the first UNet program is built");
      return DTP_ERR_STATE;
    }
    c->fp8_linear = value != 0;
    return DTP_OK;
  }
  if (!strcmp(name, "fp8_attention")) {
    if (!c->unet_progs.empty() && c->fp8_attention != (value != 0)) {
      dtp_set_error("dtp_set_option: fp8_attention must be chosen before the first UNet program is built");
      return DTP_ERR_STATE;
    }
    c->fp8_attention = value != 0;
    return DTP_OK;
  }
  dtp_set_error("dtp_set_option: unknown option '%s'", name);
  return DTP_ERR_ARG;
}

int dtp_last_stamp_info(dtp_ctx* ctx, int* unet_evals, int* graph_nodes) {
  Ctx* c = (Ctx*)ctx;
  if (!c) return DTP_ERR_ARG;
  if (unet_evals) *unet_evals = c->last_evals;
  if (graph_nodes) *graph_nodes = c->last_nodes;
  return DTP_OK;
}

}  // extern "C"
