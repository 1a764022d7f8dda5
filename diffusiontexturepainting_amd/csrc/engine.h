// Host-side engine of libdtp: weight staging/packing, static activation planning and the
// launch programs of the three networks.  One Ctx = one GPU; not thread-safe (include/dtp.h).
#pragma once
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dtp.h"
#include "common.h"

#define RC(x)            \
  do {                   \
    int rc_ = (x);       \
    if (rc_) return rc_; \
  } while (0)

struct Staged {
  float* d = nullptr;  // device fp32
  std::vector<int64_t> shape;
  size_t n = 0;
};

struct ConvW {  // 3x3 (taps = 9) or 1x1 (taps = 1) convolution / any Linear (taps = 1)
  f16* w = nullptr;
  f16* wcb = nullptr;  // 3x3 only: channel-block-major packing for conv_halo_kernel (Cin % 64 == 0, no fused shortcut)
  f16* wfr = nullptr;  // UNet only: MFMA-fragment-order packing -- of a 3x3 conv for convws_kernel (conv_ws.hip; Cin % 64 == 0, + the fused shortcut),
                       // of a Linear for gemmws_kernel (gemm_ws.hip; K % 64 == 0)
  float* b = nullptr;  // fp32 bias (may be null)
  float* lns = nullptr;  // LayerNorm folded in: row sums of the packed weights (GF_LNFOLD)
  int cout = 0, cin = 0 /* padded */, cin_true = 0, taps = 1, K = 0, ldw = 0;
  int cin2 = 0;  // >0: a 1x1 shortcut conv over cin2 channels is fused as a 10th tap (K = 9*cin + cin2)
  unsigned char* w8 = nullptr;  // Linear only, option fp8_linear: per-tensor e4m3 copy of `w` [rows][ldw8] (K padded to 128)
  int ldw8 = 0;
  float w8_scale = 1.f;
};
struct NormW {
  float *g = nullptr, *b = nullptr;
  int c = 0;
};

// NHWC fp16 view
struct T {
  f16* p = nullptr;
  int B = 0, H = 0, W = 0, C = 0, ld = 0;
  long long rows() const { return (long long)B * H * W; }
};

struct Ctx;

// A launch program: a flat list of closures bound to statically planned buffers.
using Op = std::function<int(hipStream_t, int /*step*/)>;
// profiling classes (dtp_profile_rows): 0-11 = gemm_kernel<BM,BN,NS> variants (id = shape + 4*(NS-2)), then the rest
enum { PK_GEMM0 = 0, PK_ATTN = 12, PK_GN = 13, PK_LN = 14, PK_ELEM = 15, PK_SOFTMAX = 16, PK_HALO0 = 17, PK_BIG0 = 21, PK_WIDE0 = 25, PK_FP8 = 27, PK_KH2 = 28, PK_LW = 36, PK_XATTN = 44, PK_HALO3 = 45, PK_LNLIN = 47, PK_WS0 = 48, PK_GEMMWS = 52, PK_COUNT = 53 };
struct ProfRec {
  int kind;
  double flops, bytes;
  hipEvent_t e0, e1;
  const char* label;  // owned by the closure's std::string (lives as long as the program)
};
// the split-K GEMM pushed last (if nothing was pushed after it): a GroupNorm consuming its output folds the reduce in
struct LastGemm {
  bool valid = false;
  GemmParams p;
  int tile = 0, bias_step_off = -1;
  size_t op_index = 0;
  int kind = 0;
  double flops = 0, bytes = 0;
  std::string label;
};
struct Prog {
  std::vector<Op> ops;
  LastGemm last_gemm;
  int run(hipStream_t s, int step) const {
    for (const Op& o : ops) RC(o(s, step));
    return DTP_OK;
  }
};

struct ResW {
  NormW n1, n2;
  ConvW c1, c2, sc;
  bool has_sc = false;
  int temb_off = -1;  // offset of this block's (conv1.bias + time_emb_proj(...)) slice in the step-bias table
};
struct XfW {
  NormW gn, ln1, ln2, ln3;
  ConvW proj_in, qkv, out1, q2, kv2, out2, ff1;
  ConvW ff2_proj;  // ff.net.2 and proj_out merged (load_linear_pair): [f | y3] -> block output in one GEMM
  int kv_index = -1;  // which cross-attention K/V buffer
  f16* q2T = nullptr; // LayerNorm-folded to_q of the cross-attention, transposed and packed: [up(C,128)][C] (rows = input channel)
};
struct VaeAttnW {
  NormW gn;
  ConvW qk, out;
  f16* wv = nullptr;  // [512][512] plain fp16 (used as the activation-side operand: V^T = Wv x^T)
  float* bv = nullptr;
};

struct UNetW {
  ConvW conv_in, conv_out, t1, t2, tproj;  // tproj: all 22 time_emb_proj stacked
  NormW norm_out;
  ResW down_res[4][2], mid_res[2], up_res[4][3];
  XfW down_xf[3][2], mid_xf, up_xf[4][3];
  ConvW down_conv[3], up_conv[3];
  int temb_total = 0;
};
struct VaeW {
  ConvW enc_in, enc_out, dec_in, dec_out;
  ResW enc_res[4][2], enc_mid[2], dec_mid[2], dec_res[4][3];
  ConvW enc_down[3], dec_up[3];
  VaeAttnW enc_attn, dec_attn;
  NormW enc_norm_out, dec_norm_out;
  float *quant_w = nullptr, *quant_b = nullptr, *pquant_w = nullptr, *pquant_b = nullptr;  // fp32 8x8 / 4x4
};
struct ClipLayerW {
  NormW ln1, ln2;
  ConvW qkv, out, fc1, fc2;
};
struct PencBlockW {
  NormW n1, n3;
  ConvW qkv, out, ff1, ff2;
};
struct ImgEncW {
  ConvW patch;  // 32x32 s32 conv as a [768][3072] GEMM
  float *cls_pos = nullptr;   // class_embedding + pos[0]   fp32 [768]
  float *pos = nullptr;       // pos[1..49] fp32 [49][768]
  NormW pre_ln, post_ln, final_ln;
  ClipLayerW layers[12];
  PencBlockW blocks[3][4];
  ConvW proj_out;
  float* uncond = nullptr;    // [14][768]
  float* pos_emb = nullptr;   // [14][768] (image_encoder.py:54-56)
  bool present = false;
};

struct Pool {
  struct Block { char* p; size_t bytes; bool free; };
  std::vector<Block> blocks;
  size_t total = 0;
};

// fp8 calibration (configs[4]): every fp8 problem whose activation operand is not LayerNorm'd gets its scale from the absolute maximum
// its operand reached in one evaluation of the program (Ctx::calibrating: the ops also launch dtp_launch_amax_f16 on their inputs)
struct Fp8Cal {
  int kind = 0;            // 0: Linear / 1x1 conv (slot0 = A [| A2]), 1: self-attention (slot0 .. slot0 + 2 = Q, K, V)
  int slot0 = 0;
  float* s0 = nullptr;     // Linear: a_scale; attention: q_scale
  float* s1 = nullptr;     // attention: v_scale
  float softmax_scale = 1.f;
};
constexpr int DTP_FP8_SLOTS = 2048;
constexpr float DTP_FP8_LN_A_SCALE = 0.125f;  // LayerNorm'd operands: |x| <= sqrt(K - 1) < 36 -> x * 8 < 448 never clips, three more octaves above the subnormals
constexpr float DTP_FP8_MARGIN = 2.0f;        // head-room over the calibration evaluation's absolute maximum

struct UNetProg {
  int N = 0;          // UNet batch (3B or 2B)
  size_t cal_begin = 0, cal_end = 0;  // this program's records in Ctx::fp8_cals
  bool fp8_calibrated = false;
  Prog kv, main;
  f16* in16 = nullptr;     // [N][h][w][16] input (latent 0-3, mask 4, masked latents 5-8, zero 9-15)
  f16* ctx16 = nullptr;    // [N][14][768]
  float* out32 = nullptr;  // [N][h][w][4]
  std::vector<f16*> kvbuf; // 16 x [N*14][2C]
  // Cross-attention against the 14 context tokens, fused algebraically: per sample n and block i
  //   xW1[i][n] [8*16][C]   = (scale * K_n,h,j restricted to head h) . Wq'      scores = LN2(x) . xW1^T   (+ xb1, LN fold via xl1)
  //   xW2[i][n] [C][8*16]   = Wo . (V_n,h,j restricted to head h)^T              out    = softmax_j(scores) . xW2^T + bo + x
  // both recomputed once per stamp by the `kv` program; the per-evaluation work is two small grouped GEMMs per block.
  std::vector<f16*> xW1, xW2;
  std::vector<float*> xb1, xl1;
  f16 *kexp = nullptr, *vexp = nullptr;  // scratch [N*128][1280]
  // validity of ctx16 / the per-stamp cross-attention matrices: the layout [uncond x B | cond x (NB-1)B] depends on the
  // (B, NB) split, not only on N = NB*B (B=2,NB=3 and B=3,NB=2 share a program)
  unsigned long long kv_ver = 0;
  int kv_B = 0, kv_NB = 0;
  std::vector<int> kv_slots;                       // conditioning slot of every stamp the K/V were built for
  std::vector<unsigned long long> kv_slot_ver;     // ... and the version of that slot at the time
};
struct VaeEncProg {
  int B = 0;
  Prog main;
  f16* in8 = nullptr;       // [B][R][R][8]
  float* moments = nullptr; // [B][h][w][8] (conv_out output, before quant_conv)
};
struct VaeDecProg {
  int B = 0;
  Prog main;
  f16* in8 = nullptr;     // [B][h][w][8] (after post_quant_conv)
  float* out32 = nullptr; // [B][R][R][4] (3 used)
};

struct StampBufs {  // per-batch persistent staging of dtp_stamp
  float *masks = nullptr, *ml = nullptr, *lat = nullptr, *eps = nullptr;
};
struct IencBufs {  // brush-encoder program + buffers (built on the first dtp_set_brush)
  float* img224 = nullptr;
  f16* patchA = nullptr;
  Prog prog;
  f16* out16 = nullptr;  // [14][768] final embeddings (f16)
  bool built = false;
};

struct StampGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int nodes = 0;
};

struct Ctx {
  int device = 0, R = 0, h = 0, maxB = 1, num_cu = 256;
  bool finalized = false;
  std::unordered_map<std::string, Staged> staged;

  // weight arena (zero-initialised chunks, bump allocated)
  std::vector<void*> chunks;
  char* cur = nullptr;
  size_t cur_left = 0, arena_total = 0;
  Pool pool;
  std::vector<void*> persistent;  // dedicated buffers (freed at destroy)
  f16* zero = nullptr;
  float* ws = nullptr;            // shared split-K / GroupNorm workspace
  size_t ws_bytes = 0, ws_need = 0;

  UNetW unet;
  VaeW vae;
  ImgEncW ienc;

  // programs keyed by batch
  std::map<int, UNetProg> unet_progs;  // key = N * 128 + dupB (dupB = samples filled by duplication, 0 = none)
  std::map<int, VaeEncProg> enc_progs;
  std::map<int, VaeDecProg> dec_progs;

  // step-bias table for the current step set: fp32 [nsteps][temb_total]
  float* temb_table = nullptr;
  int temb_rows = 0;         // capacity
  f16* temb_sin = nullptr;   // [rows][320]
  f16 *temb_h1 = nullptr, *temb_h2 = nullptr;  // [rows][1280]
  int sched_steps = -1;      // step count the table/coefficients were built for

  // conditioning
  // conditioning SLOTS: one brush per slot (a client of the multi-client server); slot 0 is what the single-brush entry
  // points use.  cond32 [DTP_MAX_SLOTS][2][14][768] (cond, uncond), brush32 [DTP_MAX_SLOTS][3][R][R]
  float* cond32 = nullptr;
  float* brush32 = nullptr;
  bool slot_set[DTP_MAX_SLOTS] = {};
  unsigned long long slot_version[DTP_MAX_SLOTS] = {};  // bumped by every (re)definition of the slot
  unsigned long long cond_version = 0;                  // bumped by every change of any slot
  int* slot_map = nullptr;   // device int[maxB]: conditioning slot of stamp b of the stamp batch being processed

  // stamp state
  float* x32 = nullptr;       // [maxB][h][w][4] current latent (fp32, NHWC)
  float* canvas32 = nullptr;  // [maxB][4][R][R] copy of the canvas (for compositing inside the graph-free tail)
  float* alpha_tmp = nullptr; // dilation scratch [2][maxB][R][R]
  float* stamp_params = nullptr;  // device: per-step DDIM coefficients + weights
  std::map<int, StampBufs> stamp_bufs;  // keyed by stamp batch B; the buffers live in `persistent` and die with the context
  IencBufs ienc_bufs;
  int* finite_flag = nullptr;     // device: set to 1 by the post-loop finiteness check ("check_finite" option)
  bool check_finite = false;
  bool dedupe_prefix = true;      // uncond and cond branches share the UNet prefix up to the first cross-attention ($DTP_NO_DEDUPE=1: off, A/B)
  bool fuse_gn_conv = false;      // GroupNorm + SiLU applied on the halo conv's staged input patch (GF_GNAPPLY).  OFF: measured +4.9 ms per
                                  // stamp (the normalisation costs every workgroup ~40 % -- DESIGN.md 3.6); option "fuse_gn_conv" / $DTP_GN_CONV=1
  bool fuse_xattn = true;         // the two grouped GEMMs of a cross-attention as one launch (xattn.hip; $DTP_NO_XATTN=1: off, A/B)
  bool fold_gn_linear = true;     // transformer GroupNorm folded into per-sample proj_in weights at HW >= 1024 ($DTP_NO_FOLD_GN=1: off, A/B)
  bool fuse_reduce_gn = true;     // fold a split-K conv's reduce into the GroupNorm that consumes it ($DTP_NO_FUSE_REDUCE_GN=1: off, A/B)
  bool pack_ws = false;           // load_conv also builds the fragment-order packing (set while the UNet's weights load; $DTP_NO_WS=1: never)
  bool fp8_linear = false;        // UNet transformer Linears / 1x1 convs on the fp8 MX MFMA (configs[4]); fixed once a UNet program exists
  bool fp8_attention = false;     // UNet self-attention on the fp8 MX MFMA (BASELINE configs[4]); fixed once a UNet program exists
  std::deque<float> fp8_scales;   // host copies of the calibrated scales (stable addresses: the ops read them at enqueue time)
  std::deque<Fp8Cal> fp8_cals;
  unsigned int* fp8_amax = nullptr;  // device: DTP_FP8_SLOTS float bit patterns
  int fp8_nslots = 0;
  bool calibrating = false;
  bool finite_pending = false;    // the last stamp ran the check; dtp_last_stamp_finite reads the flag
  std::map<long long, StampGraph> graphs;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int last_evals = 0, last_nodes = 0;
  bool use_graph = true;
  bool exec_imgenc_ready = false;
  bool profile = false;
  std::vector<ProfRec> prof;
  bool autotune = true;            // time every (tile, split-K) candidate of each distinct GEMM shape at build time
  std::map<std::string, std::pair<int, int>> tuned;  // shape key -> (tile, splits)
  std::map<std::string, std::pair<float, float>> tune_ms;  // $DTP_TUNE_REPORT: (cold, hot) ms of the chosen configuration
  double rep_cold_ms = 0, rep_hot_ms = 0;            // ... summed over every GEMM pushed into a program
  hipEvent_t tune_ev[2] = {nullptr, nullptr};
  void* tune_thrash = nullptr;     // 512 MiB scratch written before every timed tuning launch (cold weights)
  std::string tune_cache_path;     // $DTP_TUNE_CACHE: persisted (shape -> tile, splits) table
  size_t tune_saved = 0;
};

// ---- engine.hip
int ctx_arena_alloc(Ctx* c, size_t bytes, void** out);
int ctx_pool_get(Ctx* c, size_t bytes, void** out);
void ctx_pool_put(Ctx* c, void* p);
int ctx_persistent(Ctx* c, size_t bytes, void** out, bool zero);
const Staged* ctx_find(Ctx* c, const std::string& name);
int ctx_fetch_host(Ctx* c, const std::string& name, std::vector<float>& out);
int ctx_upload_f32(Ctx* c, const std::vector<float>& v, float** out);
int load_norm(Ctx* c, const std::string& name, NormW& n);
// conv weight [Cout][Cin][k][k] -> packed; cin_pad = padded input channels (>= Cin, multiple of 8)
int load_conv(Ctx* c, const std::string& name, ConvW& w, int cin_pad = 0, bool bias = true);
// ResBlock tail: conv2 (3x3) and the 1x1 shortcut conv packed as ONE contraction [W2 | Wsc], bias = b2 + bsc
int load_conv_with_shortcut(Ctx* c, const std::string& conv, const std::string& shortcut, ConvW& w);
// two Linears in a row with only a residual between them, out = Wb (Wa f + ba + r) + bb, merged into ONE contraction over
// [f | r]: W = [Wb Wa | Wb] (product in fp32 at load time), bias = Wb ba + bb.  K = Ka + Kb.
int load_linear_pair(Ctx* c, const std::string& first, const std::string& second, ConvW& w);
// stacked linear: rows of several [n_i][K] matrices one after another; geglu packs the [a|gate] tile order
int load_linear(Ctx* c, const std::vector<std::string>& names, ConvW& w, bool bias, bool geglu = false,
                const std::string& fold_ln = std::string());  // fold_ln: name of the LayerNorm feeding this Linear
int load_plain_f16(Ctx* c, const std::string& name, f16** out);  // unpadded fp16 copy of a matrix

// per-row (sum, sumsq) partials handed from a producer GEMM (GF_ROWSTATS) to the LayerNorm-folded consumer
struct RowStats {
  float* buf = nullptr;  // [parts][M][2]
  int parts = 0, M = 0;
  // the producer covers only rows [row_off, row_off + its M) of a rows_total-row table (de-duplicated prefix, unet.hip)
  int rows_total = 0, row_off = 0;
};

// ---- builder helpers (engine.hip): every function appends ops to `prog` and returns planned buffers
int push_gemm(Ctx* c, Prog* prog, GemmParams p, int bias_step_off, double k_alg, RowStats* emit = nullptr);

struct Builder {
  Ctx* c;
  Prog* prog;
  bool fp8 = false;  // dense Linears pushed through linear() / the transformer tail run on gemm_fp8_kernel when they can
  // append an op; when profiling is on, every launch is bracketed by HIP events on its own stream
  void push(int kind, double flops, double bytes, Op fn, const std::string& label = std::string());
  T alloc(int B, int H, int W, int C);
  void release(const T& t);
  int gn(const T& x, const NormW& n, float eps, bool silu, T& y);
  bool claim_reduce(const T& x, GemmParams& gp, int& bias_step_off, bool allow_concat = false);
  bool claim_stats(const T& x, float** partials, int* nchunk);
  bool gn_linear_supported(const T& x, const ConvW& w) const;
  int gn_linear(const T& x, const NormW& n, float eps, const ConvW& w, T& y, RowStats* emit);
  // GroupNorm + SiLU + 3x3 conv (stride 1, pad 1); the apply pass rides on the conv's staged input where the halo kernel can take it
  int gn_conv3(const T& x, const NormW& n, float eps, const ConvW& w, const T* resid, int bias_step_off, T& y, const T* tail, const T* dst);
  struct { bool active = false; const float *part = nullptr, *gamma = nullptr, *beta = nullptr; float eps = 0.f; int nchunk = 0, cpg = 0; } gn_fused;
  int ln(const T& x, const NormW& n, T& y);
  // conv3x3; bias_step_off >= 0 selects the per-step bias slice from the temb table instead of w.b
  int conv3(const T& x, const ConvW& w, int stride, int pad, bool ups, int Ho, int Wo, const T* resid, int bias_step_off,
            T& y, int extra_flags = 0, void* out_override = nullptr, int ldc_override = 0, const T* tail = nullptr,
            const T* dst = nullptr);  // dst: write into this (possibly strided) view instead of a fresh buffer
  int linear(const T& x, const ConvW& w, const T* resid, int flags, T& y, RowStats* emit = nullptr, const RowStats* use = nullptr,
             const T* dst = nullptr);
  int alloc_stats(long long rows, int C, RowStats& st);  // room for one partial per 64-column tile
  void release_stats(RowStats& st);
  int attention(const T& q, const T& k, const T& v, int heads, int Sq, int Skv, int Bn, T& o);
  int concat(const T& a, const T& b, T& y);
  int resnet(const T& x, const ResW& w, float eps, bool temb, T& y, const T* dst = nullptr);
};

int build_unet_prog(Ctx* c, int N, int dupB, UNetProg& up);
int build_vae_enc_prog(Ctx* c, int B, VaeEncProg& p);
int build_vae_dec_prog(Ctx* c, int B, VaeDecProg& p);
int load_unet_weights(Ctx* c);
int load_vae_weights(Ctx* c);
int ensure_ws(Ctx* c);
int ensure_w8(Ctx* c, ConvW& w);
// fp8 calibration: new scale / amax-slot pair for a Linear (returns the scale's address, sets *slot1), and the pass itself
float* fp8_new_linear_scale(Ctx* c, int* slot1);
int fp8_calibrate(Ctx* c, UNetProg* up, hipStream_t s, int step);  // build the e4m3 copy of a Linear's packed weights (once)
void tune_cache_load(Ctx* c);
void tune_cache_save(Ctx* c);
int ensure_temb(Ctx* c, const std::vector<float>& timesteps);  // fills temb_table rows 0..n-1

// ---- stamp.hip
int stamp_init(Ctx* c);
