// Engine core: context lifecycle, weight staging + packing, static buffer planning, op builder.
#include "engine.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>

#include <algorithm>

static size_t up_to(size_t x, size_t m) { return (x + m - 1) / m * m; }
static Op make_gemm_op(Ctx* c, GemmParams p, int tile, int bias_step_off);
void prog_push(Ctx* c, Prog* prog, int kind, double flops, double bytes, Op fn, const std::string& label);

// ---------------------------------------------------------------- memory
int ctx_arena_alloc(Ctx* c, size_t bytes, void** out) {
  bytes = up_to(bytes, 256);
  if (bytes > c->cur_left) {
    const size_t chunk = std::max(bytes, (size_t)512 << 20);
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, chunk));
    HIP_CHECK(hipMemset(p, 0, chunk));
    c->chunks.push_back(p);
    c->cur = (char*)p;
    c->cur_left = chunk;
    c->arena_total += chunk;
  }
  *out = c->cur;
  c->cur += bytes;
  c->cur_left -= bytes;
  return DTP_OK;
}

int ctx_pool_get(Ctx* c, size_t bytes, void** out) {
  bytes = up_to(bytes, 4096);
  int best = -1;
  for (size_t i = 0; i < c->pool.blocks.size(); ++i) {
    const Pool::Block& b = c->pool.blocks[i];
    if (b.free && b.bytes >= bytes && b.bytes <= bytes + bytes / 2 + (1 << 20) &&
        (best < 0 || b.bytes < c->pool.blocks[best].bytes))
      best = (int)i;
  }
  if (best < 0) {
    // +256 KiB slack: operand tiles may over-read up to 127 rows past the last valid row
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes + (256 << 10)));
    HIP_CHECK(hipMemset(p, 0, bytes + (256 << 10)));
    c->pool.blocks.push_back({(char*)p, bytes, false});
    c->pool.total += bytes;
    *out = p;
    return DTP_OK;
  }
  c->pool.blocks[best].free = false;
  *out = c->pool.blocks[best].p;
  return DTP_OK;
}

void ctx_pool_put(Ctx* c, void* p) {
  for (auto& b : c->pool.blocks)
    if (b.p == (char*)p) { b.free = true; return; }
}

int ctx_persistent(Ctx* c, size_t bytes, void** out, bool zero) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes + (256 << 10)));
  if (zero) HIP_CHECK(hipMemset(p, 0, bytes + (256 << 10)));
  c->persistent.push_back(p);
  *out = p;
  return DTP_OK;
}

const Staged* ctx_find(Ctx* c, const std::string& name) {
  auto it = c->staged.find(name);
  return it == c->staged.end() ? nullptr : &it->second;
}

int ctx_fetch_host(Ctx* c, const std::string& name, std::vector<float>& out) {
  const Staged* s = ctx_find(c, name);
  if (!s) { dtp_set_error("missing tensor '%s'", name.c_str()); return DTP_ERR_MISSING; }
  out.resize(s->n);
  HIP_CHECK(hipMemcpy(out.data(), s->d, s->n * sizeof(float), hipMemcpyDeviceToHost));
  return DTP_OK;
}

int ctx_upload_f32(Ctx* c, const std::vector<float>& v, float** out) {
  void* p;
  RC(ctx_arena_alloc(c, v.size() * sizeof(float), &p));
  HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = (float*)p;
  return DTP_OK;
}

// ---------------------------------------------------------------- weight loaders
int load_norm(Ctx* c, const std::string& name, NormW& n) {
  const Staged *g = ctx_find(c, name + ".weight"), *b = ctx_find(c, name + ".bias");
  if (!g || !b) { dtp_set_error("missing norm '%s'", name.c_str()); return DTP_ERR_MISSING; }
  void *pg, *pb;
  RC(ctx_arena_alloc(c, g->n * 4, &pg));
  RC(ctx_arena_alloc(c, b->n * 4, &pb));
  HIP_CHECK(hipMemcpy(pg, g->d, g->n * 4, hipMemcpyDeviceToDevice));
  HIP_CHECK(hipMemcpy(pb, b->d, b->n * 4, hipMemcpyDeviceToDevice));
  n.g = (float*)pg; n.b = (float*)pb; n.c = (int)g->n;
  return DTP_OK;
}

int load_conv(Ctx* c, const std::string& name, ConvW& w, int cin_pad, bool bias) {
  const Staged* s = ctx_find(c, name + ".weight");
  if (!s) { dtp_set_error("missing weight '%s.weight'", name.c_str()); return DTP_ERR_MISSING; }
  const int cout = (int)s->shape[0], cin = (int)s->shape[1];
  const int taps = s->shape.size() == 4 ? (int)(s->shape[2] * s->shape[3]) : 1;
  if (cin_pad <= 0) cin_pad = (int)up_to(cin, 8);
  w.cout = cout; w.cin = cin_pad; w.cin_true = cin; w.taps = taps; w.K = taps * cin_pad; w.ldw = (int)up_to(w.K, 64);
  void* p;
  RC(ctx_arena_alloc(c, up_to(cout, 128) * (size_t)w.ldw * 2, &p));
  w.w = (f16*)p;
  RC(dtp_launch_pack_conv_weight(s->d, w.w, cout, cin, cin_pad, taps, w.ldw, 0));
  if (taps == 9 && (cin & 63) == 0) {  // second packing for the halo-tiled kernel
    void* p2;
    RC(ctx_arena_alloc(c, up_to(cout, 128) * (size_t)w.ldw * 2, &p2));
    w.wcb = (f16*)p2;
    RC(dtp_launch_pack_conv_weight_cb(s->d, w.wcb, cout, cin, w.ldw, 0));
  }
  if (c->pack_ws && taps == 9 && (cin & 63) == 0 && cout >= 32) {  // third packing: fragment order for the weight-streaming kernel
    void* p3;
    RC(ctx_arena_alloc(c, dtp_conv_ws_packed_elems(cout, cin, 0) * 2, &p3));
    w.wfr = (f16*)p3;
    RC(dtp_launch_pack_conv_ws(s->d, nullptr, w.wfr, cout, cin, 0, 0));
  }
  w.b = nullptr;
  if (bias) {
    const Staged* b = ctx_find(c, name + ".bias");
    if (!b) { dtp_set_error("missing bias '%s.bias'", name.c_str()); return DTP_ERR_MISSING; }
    void* pb;
    RC(ctx_arena_alloc(c, b->n * 4, &pb));
    HIP_CHECK(hipMemcpy(pb, b->d, b->n * 4, hipMemcpyDeviceToDevice));
    w.b = (float*)pb;
  }
  return DTP_OK;
}

int load_conv_with_shortcut(Ctx* c, const std::string& conv, const std::string& shortcut, ConvW& w) {
  const Staged *s = ctx_find(c, conv + ".weight"), *t = ctx_find(c, shortcut + ".weight");
  const Staged *sb = ctx_find(c, conv + ".bias"), *tb = ctx_find(c, shortcut + ".bias");
  if (!s || !t || !sb || !tb) { dtp_set_error("missing tensors for '%s' + '%s'", conv.c_str(), shortcut.c_str()); return DTP_ERR_MISSING; }
  const int cout = (int)s->shape[0], cin = (int)s->shape[1], cin2 = (int)t->shape[1];
  if ((int)t->shape[0] != cout || ((9 * cin) & 63) || (cin2 & 63)) { dtp_set_error("shortcut fusion: unsupported shape at '%s'", conv.c_str()); return DTP_ERR_ARG; }
  w.cout = cout; w.cin = cin; w.cin_true = cin; w.taps = 9; w.cin2 = cin2;
  w.K = 9 * cin + cin2; w.ldw = (int)up_to(w.K, 64);
  void* p;
  RC(ctx_arena_alloc(c, up_to(cout, 128) * (size_t)w.ldw * 2, &p));
  w.w = (f16*)p;
  RC(dtp_launch_pack_conv_weight(s->d, w.w, cout, cin, cin, 9, w.ldw, 0));
  RC(dtp_launch_pack_conv_weight(t->d, w.w + 9 * cin, cout, cin2, cin2, 1, w.ldw, 0));
  void* p2;  // halo-kernel packing: channel-block-major 3x3 part, same shortcut columns
  RC(ctx_arena_alloc(c, up_to(cout, 128) * (size_t)w.ldw * 2, &p2));
  w.wcb = (f16*)p2;
  RC(dtp_launch_pack_conv_weight_cb(s->d, w.wcb, cout, cin, w.ldw, 0));
  RC(dtp_launch_pack_conv_weight(t->d, w.wcb + 9 * cin, cout, cin2, cin2, 1, w.ldw, 0));
  if (c->pack_ws && (cin & 63) == 0 && cout >= 32) {  // fragment order for the weight-streaming kernel: 3x3 part, then the shortcut
    void* p3;
    RC(ctx_arena_alloc(c, dtp_conv_ws_packed_elems(cout, cin, cin2) * 2, &p3));
    w.wfr = (f16*)p3;
    RC(dtp_launch_pack_conv_ws(s->d, t->d, w.wfr, cout, cin, cin2, 0));
  }
  std::vector<float> b1, b2;
  RC(ctx_fetch_host(c, conv + ".bias", b1));
  RC(ctx_fetch_host(c, shortcut + ".bias", b2));
  for (size_t i = 0; i < b1.size(); ++i) b1[i] += b2[i];
  b1.resize(up_to(b1.size(), 128), 0.f);
  return ctx_upload_f32(c, b1, &w.b);
}

// fragment-order copy of a packed Linear for gemmws_kernel (gemm_ws.hip) -- while the UNet's weights load (Ctx::pack_ws)
static int pack_linear_ws(Ctx* c, ConvW& w) {
  if (!c->pack_ws || w.taps != 1 || (w.K & 63) || w.K < 128 || w.cout < 32 || (w.cout & 3)) return DTP_OK;
  // opt-in ($DTP_GEMMWS=1): measured in round 4, the kernel matches the tiled ones within +-10 % on the single-round launches and loses
  // on the multi-round ones -- without the packing no problem carries Wfr and the tuner never sees tile 55
  static const bool on = [] { const char* e = getenv("DTP_GEMMWS"); return e && e[0] && e[0] != '0'; }();
  if (!on) return DTP_OK;
  void* p;
  RC(ctx_arena_alloc(c, dtp_gemm_ws_packed_elems(w.cout, w.K) * 2, &p));
  w.wfr = (f16*)p;
  return dtp_launch_pack_linear_ws(w.w, w.ldw, w.wfr, w.cout, w.K, 0);
}

int load_linear_pair(Ctx* c, const std::string& first, const std::string& second, ConvW& w) {
  const Staged *wa = ctx_find(c, first + ".weight"), *wb = ctx_find(c, second + ".weight");
  const Staged *ba = ctx_find(c, first + ".bias"), *bb = ctx_find(c, second + ".bias");
  if (!wa || !wb || !ba || !bb) { dtp_set_error("linear pair: missing '%s' / '%s'", first.c_str(), second.c_str()); return DTP_ERR_MISSING; }
  const int Na = (int)wa->shape[0], Ka = (int)(wa->n / Na), Nb = (int)wb->shape[0], Kb = (int)(wb->n / Nb);  // wb may be a 1x1 conv
  if (Kb != Na || (Ka & 63) || (Kb & 63)) { dtp_set_error("linear pair: shapes [%d,%d] then [%d,%d] do not chain", Na, Ka, Nb, Kb); return DTP_ERR_ARG; }
  const int K = Ka + Kb;
  float *prod = nullptr, *cat = nullptr, *bias = nullptr;
  HIP_CHECK(hipMalloc(&prod, (size_t)Nb * Ka * 4));
  HIP_CHECK(hipMalloc(&cat, (size_t)Nb * K * 4));
  HIP_CHECK(hipMalloc(&bias, (size_t)Nb * 4));
  RC(dtp_launch_matmul_f32(wb->d, wa->d, prod, Nb, Ka, Kb, 0));  // Wb . Wa  [Nb][Ka]
  HIP_CHECK(hipMemcpy2DAsync(cat, (size_t)K * 4, prod, (size_t)Ka * 4, (size_t)Ka * 4, Nb, hipMemcpyDeviceToDevice, 0));
  HIP_CHECK(hipMemcpy2DAsync(cat + Ka, (size_t)K * 4, wb->d, (size_t)Kb * 4, (size_t)Kb * 4, Nb, hipMemcpyDeviceToDevice, 0));
  RC(dtp_launch_rowdot(wb->d, ba->d, bias, Nb, Kb, 0));  // Wb . ba
  w = ConvW();
  w.cout = Nb; w.cin = K; w.cin_true = K; w.taps = 1; w.K = K; w.ldw = (int)up_to(K, 64);
  void* p;
  RC(ctx_arena_alloc(c, up_to(Nb, 128) * (size_t)w.ldw * 2, &p));
  w.w = (f16*)p;
  RC(dtp_launch_pack_linear_weight(cat, w.w, Nb, K, w.ldw, nullptr, 0));
  RC(pack_linear_ws(c, w));
  std::vector<float> hb(Nb), hb2;
  HIP_CHECK(hipMemcpy(hb.data(), bias, (size_t)Nb * 4, hipMemcpyDeviceToHost));
  RC(ctx_fetch_host(c, second + ".bias", hb2));
  for (int i = 0; i < Nb; ++i) hb[i] += hb2[i];
  hb.resize(up_to(hb.size(), 128), 0.f);
  RC(ctx_upload_f32(c, hb, &w.b));
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipFree(prod)); HIP_CHECK(hipFree(cat)); HIP_CHECK(hipFree(bias));
  return DTP_OK;
}

int load_linear(Ctx* c, const std::vector<std::string>& names, ConvW& w, bool bias, bool geglu, const std::string& fold_ln) {
  int N = 0, K = -1;
  for (const auto& nm : names) {
    const Staged* s = ctx_find(c, nm + ".weight");
    if (!s) { dtp_set_error("missing weight '%s.weight'", nm.c_str()); return DTP_ERR_MISSING; }
    const int k = (int)(s->n / s->shape[0]);
    if (K >= 0 && k != K) { dtp_set_error("stacked linear '%s': K mismatch", nm.c_str()); return DTP_ERR_ARG; }
    K = k;
    N += (int)s->shape[0];
  }
  w.cout = N; w.cin = K; w.cin_true = K; w.taps = 1; w.K = K; w.ldw = (int)up_to(K, 64);
  void* p;
  RC(ctx_arena_alloc(c, up_to(N, 128) * (size_t)w.ldw * 2, &p));
  w.w = (f16*)p;
  std::vector<int> map;
  int* dmap = nullptr;
  if (geglu) {
    if (names.size() != 1 || N % 256) { dtp_set_error("geglu pack: bad shape"); return DTP_ERR_ARG; }
    map.resize(N);
    for (int f = 0; f < N / 2; ++f) {
      map[f] = (f / 64) * 128 + (f % 64);
      map[N / 2 + f] = (f / 64) * 128 + 64 + (f % 64);
    }
    HIP_CHECK(hipMalloc(&dmap, N * sizeof(int)));
    HIP_CHECK(hipMemcpy(dmap, map.data(), N * sizeof(int), hipMemcpyHostToDevice));
  }
  // LayerNorm fold: LN(x) W^T + b = rstd * (x W'^T - mean * rowsum(W')) + (b + W beta), W' = W diag(gamma)
  std::vector<float> wbeta;  // (W beta)[n] over the stacked rows
  if (!fold_ln.empty()) {
    const Staged *g = ctx_find(c, fold_ln + ".weight"), *be = ctx_find(c, fold_ln + ".bias");
    if (!g || !be || (int)g->n != K) { dtp_set_error("LN fold: missing or mismatched norm '%s'", fold_ln.c_str()); return DTP_ERR_MISSING; }
    float* tmp = nullptr;
    HIP_CHECK(hipMalloc(&tmp, (size_t)N * 4));
    int r0 = 0;
    for (const auto& nm : names) {
      const Staged* s = ctx_find(c, nm + ".weight");
      const int n = (int)s->shape[0];
      RC(dtp_launch_rowdot(s->d, be->d, tmp + r0, n, K, 0));
      RC(dtp_launch_scale_cols(s->d, g->d, n, K, 0));
      r0 += n;
    }
    wbeta.resize(N);
    HIP_CHECK(hipMemcpy(wbeta.data(), tmp, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipFree(tmp));
  }
  int row = 0;
  for (const auto& nm : names) {
    const Staged* s = ctx_find(c, nm + ".weight");
    const int n = (int)s->shape[0];
    RC(dtp_launch_pack_linear_weight(s->d, w.w + (size_t)row * w.ldw, n, K, w.ldw, dmap, 0));
    row += n;
  }
  if (!fold_ln.empty()) {
    void* pl;
    const int rows = (int)up_to(N, 128);
    RC(ctx_arena_alloc(c, (size_t)rows * 4, &pl));
    w.lns = (float*)pl;
    RC(dtp_launch_rowsum_f16(w.w, w.ldw, K, w.lns, rows, 0));
  }
  if (dmap) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipFree(dmap)); }
  if (!geglu) RC(pack_linear_ws(c, w));
  w.b = nullptr;
  if (bias || !fold_ln.empty()) {
    std::vector<float> all;
    for (const auto& nm : names) {
      std::vector<float> b;
      if (bias && ctx_find(c, nm + ".bias")) RC(ctx_fetch_host(c, nm + ".bias", b));
      else b.assign((size_t)ctx_find(c, nm + ".weight")->shape[0], 0.f);
      all.insert(all.end(), b.begin(), b.end());
    }
    for (size_t i = 0; i < wbeta.size(); ++i) all[i] += wbeta[i];
    if (geglu) {
      std::vector<float> perm(all.size());
      for (size_t i = 0; i < all.size(); ++i) perm[map[i]] = all[i];
      all.swap(perm);
    }
    all.resize(up_to(all.size(), 128), 0.f);
    RC(ctx_upload_f32(c, all, &w.b));
  }
  return DTP_OK;
}

int load_plain_f16(Ctx* c, const std::string& name, f16** out) {
  const Staged* s = ctx_find(c, name);
  if (!s) { dtp_set_error("missing tensor '%s'", name.c_str()); return DTP_ERR_MISSING; }
  void* p;
  RC(ctx_arena_alloc(c, s->n * 2, &p));
  RC(dtp_launch_f32_to_f16(s->d, (f16*)p, (long long)s->n, 0));
  *out = (f16*)p;
  return DTP_OK;
}

int ensure_w8(Ctx* c, ConvW& w) {
  if (w.w8 || w.taps != 1 || !w.w) return DTP_OK;
  const size_t rows = up_to(w.cout, 128);
  w.ldw8 = (int)up_to(w.K, 128);
  void* p;
  RC(ctx_arena_alloc(c, rows * (size_t)w.ldw8, &p));  // arena chunks are zero-initialised: padded rows / columns stay 0
  w.w8 = (unsigned char*)p;
  return dtp_quantize_weights_fp8(w.w, w.ldw, w.K, (int)rows, w.w8, w.ldw8, &w.w8_scale, 0);
}

int ensure_ws(Ctx* c) {
  if (c->ws_need <= c->ws_bytes) return DTP_OK;
  HIP_CHECK(hipDeviceSynchronize());
  if (c->ws) HIP_CHECK(hipFree(c->ws));
  c->ws = nullptr;
  HIP_CHECK(hipMalloc(&c->ws, c->ws_need));
  c->ws_bytes = c->ws_need;
  // captured graphs hold the old workspace pointer: drop them, they are re-captured on next use
  for (auto& g : c->graphs) {
    if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    if (g.second.graph) (void)hipGraphDestroy(g.second.graph);
  }
  c->graphs.clear();
  return DTP_OK;
}

// ---------------------------------------------------------------- builder
void prog_push(Ctx* c, Prog* prog, int kind, double flops, double bytes, Op fn, const std::string& label) {
  prog->last_gemm.valid = false;
  prog->ops.push_back([=](hipStream_t s, int step) -> int {
    if (!c->profile) return fn(s, step);
    ProfRec r;
    r.kind = kind; r.flops = flops; r.bytes = bytes; r.label = label.c_str();
    HIP_CHECK(hipEventCreate(&r.e0));
    HIP_CHECK(hipEventCreate(&r.e1));
    HIP_CHECK(hipEventRecord(r.e0, s));
    const int rc = fn(s, step);
    HIP_CHECK(hipEventRecord(r.e1, s));
    c->prof.push_back(r);
    return rc;
  });
}
void Builder::push(int kind, double flops, double bytes, Op fn, const std::string& label) { prog_push(c, prog, kind, flops, bytes, fn, label); }

T Builder::alloc(int B, int H, int W, int C) {
  T t;
  t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
  void* p = nullptr;
  if (ctx_pool_get(c, (size_t)B * H * W * C * sizeof(f16), &p) != DTP_OK) p = nullptr;
  t.p = (f16*)p;
  return t;
}
void Builder::release(const T& t) { ctx_pool_put(c, t.p); }

// Column ranges of an lnlin_kernel launch when no tuner picks them: the LARGEST count the kernel accepts within one partial per 64
// output columns -- independent of the row count, so that the row-statistics partials (one per range) are summed in the same grouping
// whatever the batch (the de-duplicated UNet prefix evaluates two of three samples and must stay bit-identical with the tuner off).
static int lnlin_default_ranges(const GemmParams& p) {
  for (int r = (p.N + 63) / 64; r >= 1; --r)
    if (dtp_lnlin_supported(p, r)) return r;
  return 0;
}

// Which conv outputs can carry their consumer GroupNorm's statistics (GF_GNSTATS of the two-n-tile convws builds)?  ONE predicate for
// Builder::claim_stats (which re-pushes the conv with the flag) and for tune_gemm (which credits such a candidate with the statistics
// pass it saves) -- round-5 advisor: the two had repeated parts of each other's conditions (chunk cap, channels-per-group range) and
// could drift apart.  `p` is the UNSPLIT problem.  What the shape cannot tell is the consumer: an output view inside a concatenation
// buffer (ldc > N) is claimed when the next GroupNorm runs over that view alone (down path: the skip slot IS the layer output) and not
// when it runs over the whole concatenation (up path); the tuner's credit stays a shape-level estimate.
bool dtp_conv_output_can_carry_gn_stats(const GemmParams& p) {
  const int keep = GF_BIAS | GF_RESID | GF_CONV3 | GF_UPS2 | GF_MFAST;
  if ((p.flags & ~keep) || !(p.flags & GF_CONV3) || p.batch > 1) return false;
  if ((p.N % 32) || p.N / 32 < 4 || p.N / 32 > 64) return false;
  // every block of the consumer re-reads its image's whole partials table (chunks x 32 x 8 bytes): beyond ~1k chunks (the VAE's 512^2 and
  // 256^2 maps: 1 MB per image) that is more traffic than the statistics pass it replaces (round-4 advisor) -- those keep the pass
  if ((p.Ho & 7) || (p.Wo & 15) || 2 * (p.Ho / 8) * (p.Wo / 16) > 1024) return false;
  return true;
}

// A pool block that overlaps none of `ranges` (operands some launch still reads while the block is being written).  The pool hands out
// whatever fits -- also a block an operand was released from a moment ago, because the Builder releases tensors as soon as their last
// consumer is PUSHED, not run; a launch that both reads such an operand and writes the new block would race with itself (round-4 / 5
// advisor: GroupNorm output vs the claimed reduce's residual, statistics partials vs the conv's operands).  Overlapping blocks are set
// aside for the duration of the search and returned.
struct MemRange { const void* p; size_t bytes; };
static void* pool_get_clear_of(Ctx* c, size_t need, const std::vector<MemRange>& ranges, int attempts = 6) {
  std::vector<void*> aside;
  void* got = nullptr;
  for (int attempt = 0; attempt < attempts; ++attempt) {
    void* pp = nullptr;
    if (ctx_pool_get(c, need, &pp) != DTP_OK) break;
    bool hit = false;
    for (const MemRange& r : ranges)
      if (r.p && r.bytes && (const char*)pp < (const char*)r.p + r.bytes && (const char*)r.p < (const char*)pp + need) { hit = true; break; }
    if (!hit) { got = pp; break; }
    aside.push_back(pp);
  }
  for (void* q : aside) ctx_pool_put(c, q);
  return got;
}
// the operands a (re-pushed) GEMM / conv launch reads
static std::vector<MemRange> gemm_operand_ranges(const GemmParams& gp, long long images) {
  const size_t in_bytes = (gp.flags & GF_CONV3) ? (size_t)images * gp.Hi * gp.Wi * gp.lda * sizeof(f16) : (size_t)gp.M * gp.lda * sizeof(f16);
  return {{gp.A, in_bytes}, {gp.A2, gp.A2 ? (size_t)gp.M * gp.lda2 * sizeof(f16) : 0}, {gp.R, gp.R ? (size_t)gp.M * gp.ldr * sizeof(f16) : 0}};
}

// The producer of x was a split-K conv whose reduce has not run yet: take the reduce over (the GroupNorm-side kernel sums the slabs
// and writes x itself).  Re-pushes the conv with GF_NOREDUCE and returns its parameters.
bool Builder::claim_reduce(const T& x, GemmParams& gp, int& bias_step_off, bool allow_concat) {
  const LastGemm lg = prog->last_gemm;
  const int keep = GF_BIAS | GF_RESID | GF_CONV3 | GF_UPS2 | GF_MFAST;
  // round 5 (allow_concat): x may be a zero-copy concatenation [producer's N channels | skip] -- the split producer wrote (will write) the
  // FIRST lg.p.N channels of x's rows; the single-launch reduce + GroupNorm sums those from the slabs and reads the rest from x itself
  const bool whole = lg.p.N == x.C;
  const bool front = allow_concat && lg.p.N < x.C && (lg.p.N & 7) == 0 && x.H * x.W <= 256 && dtp_reduce_groupnorm_supported(x.H * x.W, x.C, 32);
  if (!(c->fuse_reduce_gn && lg.valid && lg.p.splits > 1 && (f16*)lg.p.C == x.p && lg.p.ldc == x.ld && lg.p.M == (int)x.rows() && (whole || front) &&
        !(lg.p.flags & ~keep) && lg.p.batch <= 1 && (x.C & 7) == 0))
    return false;
  {  // what the GroupNorm-side reduce kernels accept (norm.hip): checked here, where the separate reduce is still the fallback
    const int cpg = x.C / 32;
    if ((x.C % 32) || cpg < 4 || (cpg < 8 && cpg != 4) || (lg.p.N & 3) || x.C / 8 > 1024 || (x.ld & 7) || ((lg.p.flags & GF_RESID) && (lg.p.ldr & 7))) return false;
  }
  gp = lg.p;
  gp.flags |= GF_NOREDUCE;
  prog->ops[lg.op_index] = Op();  // rebuilt below through the profiling wrapper
  prog->ops.pop_back();
  prog_push(c, prog, lg.kind, lg.flops, lg.bytes, make_gemm_op(c, gp, lg.tile, lg.bias_step_off), lg.label + " (reduce in gn)");
  bias_step_off = lg.bias_step_off;
  return true;
}

// The producer of x was an UNSPLIT two-n-tile convws launch (tile 53 / 54): let its epilogue emit the GroupNorm partial sums of x (GF_GNSTATS,
// conv_ws.hip) -- the consumer then needs no statistics pass over x.  Re-pushes the conv with the flag and a planned partials buffer
// [B][nchunk][32][2]; the caller returns the buffer to the pool once its consumer is pushed.
bool Builder::claim_stats(const T& x, float** partials, int* nchunk) {
  static const bool off = [] { const char* e = getenv("DTP_NO_GN_EPILOGUE"); return e && e[0] && e[0] != '0'; }();
  const LastGemm lg = prog->last_gemm;
  if (off || !lg.valid || lg.tile < DTP_TILE_WS0 + 2 || !dtp_is_ws_tile(lg.tile) || lg.p.splits != 1 || (f16*)lg.p.C != x.p || lg.p.ldc != x.ld ||
      lg.p.M != (int)x.rows() || lg.p.N != x.C || !dtp_conv_output_can_carry_gn_stats(lg.p))
    return false;
  GemmParams gp = lg.p;
  const int chunks = 2 * (gp.Ho / 8) * (gp.Wo / 16);
  // The conv's input (and residual / shortcut operand) may already be back in the pool -- gn_conv3 releases it before its consumer is
  // built -- and the pool would happily hand that very block out for the partials, which the re-pushed conv WRITES while other
  // workgroups still read the operand (round-4 advisor: a latent aliasing race): pool_get_clear_of.
  const size_t need = (size_t)x.B * chunks * 32 * 2 * sizeof(float);
  void* pp = pool_get_clear_of(c, need, gemm_operand_ranges(gp, x.B), 4);
  if (!pp) return false;
  gp.flags |= GF_GNSTATS;
  gp.st_out = (float*)pp;
  gp.gn_cpg = x.C / 32;
  if (!dtp_conv_ws_supported(gp, lg.tile - DTP_TILE_WS0, 1)) { ctx_pool_put(c, pp); return false; }
  prog->ops.pop_back();
  prog_push(c, prog, lg.kind, lg.flops, lg.bytes, make_gemm_op(c, gp, lg.tile, lg.bias_step_off), lg.label + " (+gn stats)");
  *partials = (float*)pp;
  *nchunk = chunks;
  return true;
}

int Builder::gn(const T& x, const NormW& n, float eps, bool silu, T& y) {
  Ctx* cc = c;
  GemmParams gp;
  int bso = -1;
  static const bool concat_off = [] { const char* e = getenv("DTP_NO_REDUCE_IN_CONCAT_GN"); return e && e[0] && e[0] != '0'; }();  // A/B
  const bool claimed = claim_reduce(x, gp, bso, !concat_off);
  // The claimed reduce adds the producer's residual INSIDE the GroupNorm launch -- and that residual (a transformer block's input) is
  // normally back in the pool by now, so the pool may hand its block out for y: one workgroup would write y where another still reads
  // the residual.  (With y laid out exactly like the residual every thread reads the element it later overwrites, which is why the
  // whole-tensor case never showed it; over a concatenation the pitches differ and the stamp stopped being reproducible.)  Blocks
  // that overlap the residual are set aside while y is taken.
  {
    std::vector<MemRange> busy;
    if (claimed && (gp.flags & GF_RESID)) busy.push_back({gp.R, (size_t)gp.M * gp.ldr * sizeof(f16)});
    y = T();
    y.B = x.B; y.H = x.H; y.W = x.W; y.C = x.C; y.ld = x.C;
    y.p = (f16*)pool_get_clear_of(cc, (size_t)x.B * x.H * x.W * x.C * sizeof(f16), busy);
    if (!y.p) { dtp_set_error("gn: no output block clear of the claimed reduce's residual"); return DTP_ERR_HIP; }
  }
  cc->ws_need = std::max(cc->ws_need, dtp_groupnorm_ws_bytes(x.B, x.H * x.W, x.C, 32));
  const T xx = x, yy = y;
  const NormW nn = n;
  if (claimed) {
    const bool has_bias = (gp.flags & GF_BIAS) != 0;
    // small maps: one launch does it all; large maps: the reduce rides in the statistics pass, whose partial sums live behind the slabs
    const size_t slab_bytes = (dtp_gemm_workspace_bytes(gp) + 255) & ~(size_t)255;
    cc->ws_need = std::max(cc->ws_need, slab_bytes + dtp_groupnorm_ws_bytes(x.B, x.H * x.W, x.C, 32));
    push(PK_GN, 0.0, 4.0 * (double)xx.rows() * xx.C, [=](hipStream_t s, int step) {
      const float* bias = !has_bias ? nullptr : (bso >= 0 ? cc->temb_table + (size_t)step * cc->unet.temb_total + bso : gp.bias);
      return dtp_launch_reduce_groupnorm(cc->ws, gp.splits, (long long)gp.M * gp.N, gp.N, bias, (gp.flags & GF_RESID) ? gp.R : nullptr, gp.ldr,
                                         xx.p, xx.ld, yy.p, yy.ld, nn.g, nn.b, xx.B, xx.H * xx.W, xx.C, 32, eps, silu ? 1 : 0,
                                         (float*)((char*)cc->ws + slab_bytes), s, gp.N);
    }, std::string(gp.N < x.C ? "reduce(front " + std::to_string(gp.N) + ")+gn B=" : "reduce+gn B=") + std::to_string(x.B) + " HW=" + std::to_string(x.H * x.W) + " C=" + std::to_string(x.C) + " splits=" + std::to_string(gp.splits));
    return DTP_OK;
  }
  float* partials = nullptr;
  int nchunk = 0;
  if (claim_stats(x, &partials, &nchunk)) {  // statistics from the producing conv's epilogue: the apply pass alone
    push(PK_GN, 0.0, 4.0 * (double)xx.rows() * xx.C, [=](hipStream_t s, int) {
      return dtp_launch_groupnorm_apply(xx.p, xx.ld, yy.p, yy.ld, nn.g, nn.b, partials, nchunk, xx.B, xx.H * xx.W, xx.C, 32, eps, silu ? 1 : 0, s);
    }, "gn-apply B=" + std::to_string(x.B) + " HW=" + std::to_string(x.H * x.W) + " C=" + std::to_string(x.C));
    ctx_pool_put(cc, partials);
    return DTP_OK;
  }
  push(PK_GN, 0.0, 4.0 * (double)xx.rows() * xx.C, [=](hipStream_t s, int) {
    return dtp_launch_groupnorm(xx.p, xx.ld, yy.p, yy.ld, nn.g, nn.b, cc->ws, xx.B, xx.H * xx.W, xx.C, 32, eps, silu ? 1 : 0, s);
  }, "gn B=" + std::to_string(x.B) + " HW=" + std::to_string(x.H * x.W) + " C=" + std::to_string(x.C));
  return DTP_OK;
}

// GroupNorm (no activation) + the Linear / 1x1 conv that consumes it, with the normalisation folded into per-sample weights
// (norm.hip gn_fold_weights_kernel): statistics pass (+ the producer's split-K reduce) -> fold -> ONE grouped GEMM on the raw tensor.
// The apply pass and the normalised tensor do not exist.  Same three launches as stats + apply + GEMM, but the middle one touches
// N * C * C weights instead of reading and writing the whole activation tensor.
bool Builder::gn_linear_supported(const T& x, const ConvW& w) const {
  const int HW = x.H * x.W, C = x.C;
  return c->fold_gn_linear && HW >= 1024 && w.taps == 1 && w.K == C && w.ldw == C && (C % 64) == 0 && C <= 2048 && (C % 32) == 0 && !w.lns &&
         !(fp8 && w.w8) && C / 32 >= 8;
}

int Builder::gn_linear(const T& x, const NormW& n, float eps, const ConvW& w, T& y, RowStats* emit) {
  Ctx* cc = c;
  const int HW = x.H * x.W, C = x.C, N = x.B, Cp = (w.cout + 127) / 128 * 128;
  GemmParams gp;
  int bso = -1;
  const bool claimed = claim_reduce(x, gp, bso);
  float* ep_part = nullptr;  // statistics emitted by the producing conv's epilogue (claim_stats): no statistics launch at all
  int ep_chunks = 0;
  const bool from_epilogue = !claimed && claim_stats(x, &ep_part, &ep_chunks);
  const size_t slab_bytes = claimed ? ((dtp_gemm_workspace_bytes(gp) + 255) & ~(size_t)255) : 0;
  cc->ws_need = std::max(cc->ws_need, slab_bytes + dtp_groupnorm_ws_bytes(N, HW, C, 32));
  // Round 6: where the activation-stationary Linear takes the problem (K = C in {320, 640}: UNet levels 0-1), the GroupNorm is applied to
  // its RESIDENT activation fragments (lnlin_kernel GNA) -- [statistics ->] ONE launch on the raw tensor with the shared weights; the fold
  // launch (190 per batch-1 stamp), the per-sample weight copies and the grouped problem disappear.  $DTP_NO_GNA_LNLIN=1: the fold (A/B).
  static const bool gna_off = [] { const char* e = getenv("DTP_NO_GNA_LNLIN"); return e && e[0] && e[0] != '0'; }();
  if (!gna_off && (C == 320 || C == 640) && (HW & 127) == 0 && !fp8) {
    GemmParams g = {};
    g.A = x.p; g.lda = x.ld; g.W = w.w; g.ldw = w.ldw; g.nkb = w.ldw / 64;
    g.M = HW; g.N = w.cout; g.K = w.K;
    g.bias = w.b; g.flags = (w.b ? GF_BIAS : 0) | GF_GNAPPLY;
    g.batch = N; g.a_bs = (long long)HW * x.ld; g.c_bs = (long long)HW * w.cout; g.w_bs = 0; g.bias_bs = 0;
    g.gn_gamma = n.g; g.gn_beta = n.b; g.gn_eps = eps; g.gn_cpg = C / 32;
    g.gn_nchunk = from_epilogue ? ep_chunks : dtp_groupnorm_stat_chunks(HW);
    g.gn_part = (const float*)x.p;  // (a non-null placeholder for the support check: the partials block is planned below)
    g.ldc = w.cout;
    if (lnlin_default_ranges(g) > 0) {
      // the partial sums outlive the statistics launch inside the shared workspace only until the next split launch: own planned buffer
      float* part = ep_part;
      if (!from_epilogue) {
        std::vector<MemRange> busy;
        if (claimed && (gp.flags & GF_RESID)) busy.push_back({gp.R, (size_t)gp.M * gp.ldr * sizeof(f16)});
        part = (float*)pool_get_clear_of(cc, dtp_groupnorm_ws_bytes(N, HW, C, 32), busy);
        if (!part) { dtp_set_error("gn_linear: no partials block"); return DTP_ERR_HIP; }
        const T xx = x;
        const bool has_bias = claimed && (gp.flags & GF_BIAS) != 0;
        push(PK_GN, 0.0, 2.0 * (double)xx.rows() * C, [=](hipStream_t s, int step) {
          if (claimed) {
            GnReduceSrc rd;
            rd.part = cc->ws; rd.splits = gp.splits; rd.slab = (long long)gp.M * gp.N; rd.ldp = gp.N;
            rd.bias = !has_bias ? nullptr : (bso >= 0 ? cc->temb_table + (size_t)step * cc->unet.temb_total + bso : gp.bias);
            rd.R = (gp.flags & GF_RESID) ? gp.R : nullptr; rd.ldr = gp.ldr;
            return dtp_launch_groupnorm_stats(xx.p, xx.ld, part, N, HW, C, 32, &rd, s);
          }
          return dtp_launch_groupnorm_stats(xx.p, xx.ld, part, N, HW, C, 32, nullptr, s);
        }, std::string(claimed ? "reduce+gn-stats" : "gn-stats") + " B=" + std::to_string(N) + " HW=" + std::to_string(HW) + " C=" + std::to_string(C) + " (apply in proj_in)");
      }
      g.gn_part = part;
      y = alloc(x.B, x.H, x.W, w.cout);
      if (!y.p) return DTP_ERR_HIP;
      g.C = y.p; g.ldc = y.ld; g.c_bs = (long long)HW * y.ld;
      if (emit && emit->buf) {
        g.flags |= GF_ROWSTATS; g.st_out = emit->buf; g.st_rows = N * HW;
        if (emit->rows_total > 0) { g.st_out = emit->buf + (size_t)emit->row_off * 2; g.st_rows = emit->rows_total; }
      }
      RC(push_gemm(cc, prog, g, -1, (double)w.K, (emit && emit->buf) ? emit : nullptr));
      ctx_pool_put(cc, part);
      return DTP_OK;
    }
  }
  void *pw = nullptr, *pb = nullptr;
  RC(ctx_pool_get(cc, (size_t)N * Cp * w.ldw * sizeof(f16), &pw));
  RC(ctx_pool_get(cc, (size_t)N * Cp * sizeof(float), &pb));
  f16* Wf = (f16*)pw;
  float* bf = (float*)pb;
  const T xx = x;
  const NormW nn = n;
  const ConvW ww = w;
  const bool has_bias = claimed && (gp.flags & GF_BIAS) != 0;
  if (!from_epilogue)
  push(PK_GN, 0.0, 2.0 * (double)xx.rows() * C, [=](hipStream_t s, int step) {
    float* part_ws = (float*)((char*)cc->ws + slab_bytes);
    if (claimed) {
      GnReduceSrc rd;
      rd.part = cc->ws; rd.splits = gp.splits; rd.slab = (long long)gp.M * gp.N; rd.ldp = gp.N;
      rd.bias = !has_bias ? nullptr : (bso >= 0 ? cc->temb_table + (size_t)step * cc->unet.temb_total + bso : gp.bias);
      rd.R = (gp.flags & GF_RESID) ? gp.R : nullptr; rd.ldr = gp.ldr;
      return dtp_launch_groupnorm_stats(xx.p, xx.ld, part_ws, N, HW, C, 32, &rd, s);
    }
    return dtp_launch_groupnorm_stats(xx.p, xx.ld, part_ws, N, HW, C, 32, nullptr, s);
  }, std::string(claimed ? "reduce+gn-stats" : "gn-stats") + " B=" + std::to_string(N) + " HW=" + std::to_string(HW) + " C=" + std::to_string(C));
  push(PK_GN, 0.0, 2.0 * (double)N * w.cout * C * 2, [=](hipStream_t s, int) {
    const float* part = from_epilogue ? ep_part : (const float*)((char*)cc->ws + slab_bytes);
    return dtp_launch_gn_fold_weights(ww.w, ww.ldw, ww.b, nn.g, nn.b, part, N, HW, C, ww.cout, 32, eps, Wf, (long long)Cp * ww.ldw, bf, Cp, s,
                                      from_epilogue ? ep_chunks : 0);
  }, std::string(from_epilogue ? "gn-fold (stats from conv) B=" : "gn-fold B=") + std::to_string(N) + " C=" + std::to_string(C) + " N=" + std::to_string(w.cout));
  if (from_epilogue) ctx_pool_put(cc, ep_part);
  y = alloc(x.B, x.H, x.W, w.cout);
  if (!y.p) return DTP_ERR_HIP;
  GemmParams g = {};
  g.A = x.p; g.lda = x.ld; g.W = Wf; g.ldw = w.ldw; g.nkb = w.ldw / 64;
  g.M = HW; g.N = w.cout; g.K = w.K;
  g.C = y.p; g.ldc = y.ld;
  g.bias = bf; g.flags = GF_BIAS;
  g.batch = N; g.a_bs = (long long)HW * x.ld; g.w_bs = (long long)Cp * w.ldw; g.c_bs = (long long)HW * y.ld; g.bias_bs = Cp;
  if (emit && emit->buf) {
    g.flags |= GF_ROWSTATS; g.st_out = emit->buf; g.st_rows = N * HW;
    if (emit->rows_total > 0) { g.st_out = emit->buf + (size_t)emit->row_off * 2; g.st_rows = emit->rows_total; }
  }
  RC(push_gemm(cc, prog, g, -1, (double)w.K, (emit && emit->buf) ? emit : nullptr));
  ctx_pool_put(cc, pw);
  ctx_pool_put(cc, pb);
  return DTP_OK;
}

int Builder::ln(const T& x, const NormW& n, T& y) {
  y = alloc(x.B, x.H, x.W, x.C);
  if (!y.p) return DTP_ERR_HIP;
  const T xx = x, yy = y;
  const NormW nn = n;
  push(PK_LN, 0.0, 4.0 * (double)xx.rows() * xx.C, [=](hipStream_t s, int) {
    return dtp_launch_layernorm(xx.p, xx.ld, yy.p, yy.ld, nn.g, nn.b, (int)xx.rows(), xx.C, 1e-5f, s);
  });
  return DTP_OK;
}

void prog_push(Ctx* c, Prog* prog, int kind, double flops, double bytes, Op fn, const std::string& label);
static void tune_read_file(Ctx* c, const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return;
  char key[256];
  int tile, splits;
  while (fscanf(f, "%255s %d %d", key, &tile, &splits) == 3)
    if (tile >= 0 && tile < DTP_TILE_IDS && splits >= 1 && splits <= 64) c->tuned[key] = std::make_pair(tile, splits);  // shape-level checks: tune_entry_valid()
  fclose(f);
}

// $DTP_TUNE_SEED: a read-only table shipped with the package (the choices measured on the build's own MI355X), read first;
// $DTP_TUNE_CACHE: the per-user table this process may extend.  Entries are only trusted after tune_entry_valid().
void tune_cache_load(Ctx* c) {
  const char* seed = getenv("DTP_TUNE_SEED");
  if (seed && *seed) tune_read_file(c, seed);
  const char* e = getenv("DTP_TUNE_CACHE");
  if (!e || !*e) return;
  c->tune_cache_path = e;
  tune_read_file(c, e);
  c->tune_saved = c->tuned.size();
}

// Is (tile, splits) a configuration the launcher accepts for THIS problem?  A persisted table can be stale (older build,
// different packing) or hand-edited: a halo tile without the channel-block-major packing, a GEGLU problem on a tile that is
// not 128 wide, or a split LayerNorm-fold would otherwise reach the kernels.
static bool tune_entry_valid(const GemmParams& p, int tile, int sp) {
  if (sp < 1 || (sp > p.nkb && tile != DTP_TILE_LNLIN)) return false;
  if (tile == DTP_TILE_GEMMWS) return dtp_gemm_ws_supported(p, sp);
  if (tile != DTP_TILE_LNLIN && !dtp_is_ws_tile(tile)) {
    int kbps, n;
    dtp_split_k(p.nkb, tile, sp, &kbps, &n);
    if (n != sp) return false;  // not a factor this tile can realise
  }
  if (tile == DTP_TILE_LNLIN) return dtp_lnlin_supported(p, sp);
  if (dtp_is_ws_tile(tile)) return dtp_conv_ws_supported(p, tile - DTP_TILE_WS0, sp);
  const bool halo = dtp_is_halo_tile(tile);
  if (halo) return p.Wcb && (tile >= 48 ? dtp_conv_halo3_supported(p) : dtp_conv_halo_supported(p)) && p.batch <= 1;
  if (p.flags & GF_GNAPPLY) return false;  // only the halo kernel normalises its staged input
  int bm = 0, bn = 0, ns = 0;
  if (!dtp_gemm_tile_dims(tile, &bm, &bn, &ns)) return false;
  if (tile >= 24 && tile < 32) { GemmParams q = p; q.splits = 1; return sp == 1 && tile <= 28 && dtp_gemm_fp8_supported(q) && !((p.flags & GF_GEGLU) && (bn % 128)); }
  if (tile >= 20 && tile < 32) { GemmParams q = p; q.splits = 1; return sp == 1 && dtp_gemm_wide_supported(q, tile - 20); }
  if ((p.flags & GF_GEGLU) && (bn != 128 || sp != 1)) return false;
  if (sp > 1 && ((p.flags & (GF_LNFOLD | GF_SOFTMAX16)) || p.batch > 1)) return false;
  if (sp > 1 && (size_t)sp * p.M * p.N * sizeof(float) > ((size_t)512 << 20)) return false;  // the fp32 slabs of a split
  return true;
}

void tune_cache_save(Ctx* c) {
  if (c->rep_cold_ms > 0) fprintf(stderr, "[tune] sum over pushed GEMMs: cold %.2f ms, hot %.2f ms\n", c->rep_cold_ms, c->rep_hot_ms);
  if (c->tune_thrash) { (void)hipDeviceSynchronize(); (void)hipFree(c->tune_thrash); c->tune_thrash = nullptr; }
  if (c->tune_cache_path.empty() || c->tuned.size() == c->tune_saved) return;
  // several ranks may share the path: write a private file and rename it into place (atomic)
  const std::string tmp = c->tune_cache_path + ".tmp." + std::to_string((long long)getpid());
  FILE* f = fopen(tmp.c_str(), "w");
  if (!f) return;
  for (auto& kv : c->tuned) fprintf(f, "%s %d %d\n", kv.first.c_str(), kv.second.first, kv.second.second);
  fclose(f);
  (void)rename(tmp.c_str(), c->tune_cache_path.c_str());
  c->tune_saved = c->tuned.size();
}

// Build-time autotuning: the stamp path has ~100 distinct contraction shapes, most of them far from
// "large square GEMM" (M from 192 to 524288, N from 3 to 10240).  Each distinct shape is timed once
// with every tile variant x split-K factor on the real buffers and the fastest pair is kept.
static int tune_gemm(Ctx* c, GemmParams& p, int* tile_out) {
  char key[200];
  // "k8|": bump when tile ids or pipelines change, so that a persisted table written by an older build is ignored
  int kl = snprintf(key, sizeof(key), "k8|%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", p.M, p.N, p.K, p.flags & ~GF_MFAST, p.Hi, p.Wi, p.Cin,
                    p.stride, p.lda, p.ldc, p.ldw, p.st_parts, p.Cin2, p.lda2);
  if (p.batch > 1) kl += snprintf(key + kl, sizeof(key) - kl, ",b%d", p.batch);
  if (p.W8) kl += snprintf(key + kl, sizeof(key) - kl, ",f8");
  // problems the weight-streaming conv can take were tuned without it by older tables: their key carries a marker
  bool ws_ok[DTP_WS_VARIANTS];
  bool ws_any = false;
  for (int v = 0; v < DTP_WS_VARIANTS; ++v) { ws_ok[v] = p.Wfr && (p.flags & GF_CONV3) && dtp_conv_ws_supported(p, v, 1); ws_any = ws_any || ws_ok[v]; }
  if (ws_any) kl += snprintf(key + kl, sizeof(key) - kl, ",ws2");
  // ... and so do the plain problems the activation-stationary Linear takes since round 4 (attention output projection, grouped proj_in)
  if (!(p.flags & GF_LNFOLD)) {
    bool ll = false;
    for (int r = 1; r <= 40 && !ll; ++r) ll = dtp_lnlin_supported(p, r);
    if (ll) snprintf(key + kl, sizeof(key) - kl, ",ll");
  }
  // ... and the dense problems the weight-streaming GEMM takes (gemm_ws.hip)
  const bool gw_ok = dtp_gemm_ws_supported(p, 1);
  if (gw_ok) { kl = (int)strlen(key); snprintf(key + kl, sizeof(key) - kl, ",gw"); }
  auto it = c->tuned.find(key);
  if (it != c->tuned.end() && !tune_entry_valid(p, it->second.first, it->second.second)) {
    fprintf(stderr, "[dtp] tune table entry '%s' -> (%d, %d) does not fit the problem; re-tuning\n", key, it->second.first, it->second.second);
    c->tuned.erase(it);
    it = c->tuned.end();
  }
  if (it == c->tuned.end()) {
    if (!c->tune_ev[0]) { HIP_CHECK(hipEventCreate(&c->tune_ev[0])); HIP_CHECK(hipEventCreate(&c->tune_ev[1])); }
    constexpr size_t THRASH_BYTES = (size_t)512 << 20;
    if (!c->tune_thrash) HIP_CHECK(hipMalloc(&c->tune_thrash, THRASH_BYTES));
    const size_t a_bytes = (p.flags & GF_CONV3) ? (size_t)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.lda * 2 : (size_t)p.M * p.lda * 2;
    float best = 1e30f;
    int bt = *tile_out, bs = p.splits;
    const bool geglu = (p.flags & GF_GEGLU) != 0;
    static const int cand_splits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    // One candidate = (tile id, split-K factor); tile ids 12..15 are the halo-tiled conv kernels (they read the Wcb packing).
    // Timed the way the stamp sees it: weights COLD (1.7 GB of them stream through the 256 MiB Infinity Cache every UNet
    // evaluation), activations warm (just written by the previous kernel); minimum over `reps` runs (a single cold run is noisy:
    // DVFS, thrash write-back still draining).  Returns < 0 when the candidate does not apply.
    auto time_cfg = [&](int tile, int sp, int reps, float* out_ms) -> int {
      *out_ms = -1.f;
      GemmParams q = p;
      const bool halo = dtp_is_halo_tile(tile);
      if (halo) q.W = p.Wcb;
      dtp_split_k(p.nkb, tile, sp, &q.kb_per_split, &q.splits);
      if (tile == DTP_TILE_LNLIN) q.col_ranges = sp;
      else if (q.splits != sp && sp > 1) return DTP_OK;
      const size_t need = dtp_gemm_workspace_bytes(q);
      if (need > ((size_t)512 << 20)) return DTP_OK;
      if (need > c->ws_bytes) { c->ws_need = std::max(c->ws_need, need); RC(ensure_ws(c)); }
      q.part = c->ws;
      q.zero = c->zero;
      float ms = 1e30f;
      for (int rep = 0; rep < reps; ++rep) {
        HIP_CHECK(hipMemsetAsync(c->tune_thrash, rep, THRASH_BYTES, 0));
        RC(dtp_launch_touch(q.A, a_bytes, (float*)c->tune_thrash, 0));
        if (q.R) RC(dtp_launch_touch(q.R, (size_t)q.M * q.ldr * 2, (float*)c->tune_thrash, 0));
        HIP_CHECK(hipEventRecord(c->tune_ev[0], 0));
        if (tile == DTP_TILE_LNLIN) RC(dtp_launch_lnlin(q, sp, 0));
        else if (tile == DTP_TILE_GEMMWS) RC(dtp_launch_gemm_ws(q, 0));
        else if (dtp_is_ws_tile(tile)) RC(dtp_launch_conv_ws(q, tile - DTP_TILE_WS0, 0));
        else if (halo) RC(dtp_launch_conv_halo(q, dtp_halo_variant(tile), 0)); else RC(dtp_launch_gemm(q, tile, 0));
        HIP_CHECK(hipEventRecord(c->tune_ev[1], 0));
        HIP_CHECK(hipEventSynchronize(c->tune_ev[1]));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, c->tune_ev[0], c->tune_ev[1]));
        ms = std::min(ms, t);
      }
      *out_ms = ms;
      return DTP_OK;
    };
    struct Cand { float ms; int tile, sp; };
    std::vector<Cand> cands;
    for (int tile = 0; tile < 48 && !(p.flags & GF_GNAPPLY); ++tile) {  // 4 tile shapes x 3 pipeline depths, the 256-row / 256-column tiles, the 8-wave wide tiles, fp8, the 8-wave twins of the small shapes, their loader-wave variants
      int bm = 0, bn = 0, ns = 0;
      if (!dtp_gemm_tile_dims(tile, &bm, &bn, &ns)) continue;
      // fp8 tiles need the e4m3 weight copy.  An fp8 problem keeps the choice of an fp16 tile while it is small (the register-
      // staged activation operand costs latency-bound launches more than the MX MFMA returns: 256^2 / 8 steps 32.6 -> 27.5 ms);
      // from M = 6144 on (every level-0..2 Linear of a batch-8 stamp) it runs on the fp8 tiles only -- there the cold single-launch
      // timing of the tuner under-rates them (batch 8: 588 ms with fp8 tiles throughout, 606 ms with the tuner's mix, 605 ms in fp16)
      const bool f8t = tile >= 24 && tile < 32;
      if (f8t ? !p.W8 : (p.W8 && p.M >= 6144)) continue;
      if (tile >= 24 && tile < 32) {
        if (geglu && (bn % 128)) continue;
        if (tile == 28 && (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) < 96) continue;
        float ms;
        RC(time_cfg(tile, 1, 5, &ms));
        if (ms >= 0.f) cands.push_back({ms, tile, 1});
        continue;
      }
      if (tile >= 20 && tile < 32) {  // gemm_wide_kernel: unsplit big-M problems only (at least half a wave of 256 CUs worth of tiles)
        GemmParams q = p;
        q.splits = 1;
        if (!dtp_gemm_wide_supported(q, tile - 20)) continue;
        if ((long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) < 96) continue;
        if (tile == 21 && (p.N % 320) > 0 && (p.N % 320) <= 192) continue;  // a mostly empty last column tile: 256 x 256 covers it better
        float ms;
        RC(time_cfg(tile, 1, 5, &ms));
        if (ms >= 0.f) cands.push_back({ms, tile, 1});
        continue;
      }
      if (geglu && bn != 128) continue;
      if (p.nkb < 3 && ns > 2) continue;
      if ((bm == 256 && p.M < 192) || (bn == 256 && p.N < 192)) continue;
      for (int sp : cand_splits) {
        if (sp > 1 && (geglu || (p.flags & GF_LNFOLD) || p.batch > 1 || p.nkb / sp < 2)) break;
        float ms;
        RC(time_cfg(tile, sp, 5, &ms));
        if (ms >= 0.f) cands.push_back({ms, tile, sp});
      }
    }
    {  // the activation-stationary kernel of the short LayerNorm-folded contractions: column ranges per 128-row block
      static const bool no_lnlin = [] { const char* e = getenv("DTP_NO_LNLIN"); return e && e[0] && e[0] != '0'; }();
      static const int ranges[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 40};
      for (int sp : ranges) {
        if (no_lnlin || !dtp_lnlin_supported(p, sp)) continue;
        float ms;
        RC(time_cfg(DTP_TILE_LNLIN, sp, 5, &ms));
        if (ms >= 0.f) cands.push_back({ms, DTP_TILE_LNLIN, sp});
      }
    }
    if (p.Wcb && dtp_conv_halo_supported(p)) {
      for (int v = 0; v < 4; ++v) {
        if ((v >= 2) != (p.Hi * p.Wi <= 256)) continue;  // 8x8 pixel tiles for small feature maps, 8x16 otherwise
        for (int sp : cand_splits) {
          if (sp > 1 && p.nkb / sp < 9) break;
          float ms;
          RC(time_cfg(12 + v, sp, 5, &ms));
          if (ms >= 0.f) cands.push_back({ms, 12 + v, sp});
        }
      }
      // three images per workgroup: the small maps of a batch-1 stamp, where the weight slices are most of the LDS fill
      static const bool no_halo3 = [] { const char* e = getenv("DTP_NO_HALO3"); return e && e[0] && e[0] != '0'; }();
      for (int tile = 48; tile < 50 && !no_halo3 && p.Hi * p.Wi <= 256 && dtp_conv_halo3_supported(p); ++tile)
        for (int sp : cand_splits) {
          if (sp > 1 && p.nkb / sp < 9) break;
          float ms;
          RC(time_cfg(tile, sp, 5, &ms));
          if (ms >= 0.f) cands.push_back({ms, tile, sp});
        }
    }
    for (int v = 0; v < DTP_WS_VARIANTS; ++v) {  // the weight-streaming conv: K-slices = ranges of whole channel blocks
      static const bool no_ws = [] { const char* e = getenv("DTP_NO_WS"); return e && e[0] && e[0] != '0'; }();
      static const int slices[] = {1, 2, 3, 4, 5, 6, 8, 10};
      for (int sp : slices) {
        if (no_ws || !ws_ok[v] || !dtp_conv_ws_supported(p, v, sp)) continue;
        if (v >= 2 && sp > 4) continue;
        float ms;
        RC(time_cfg(DTP_TILE_WS0 + v, sp, 5, &ms));
        if (ms >= 0.f) cands.push_back({ms, DTP_TILE_WS0 + v, sp});
      }
    }
    if (gw_ok) {  // the weight-streaming GEMM: K-slices = ranges of whole k-blocks
      static const int slices[] = {1, 2, 3, 4, 5, 6, 8};
      for (int sp : slices) {
        if (!dtp_gemm_ws_supported(p, sp) || (sp > 1 && p.nkb / sp < 4)) continue;
        float ms;
        RC(time_cfg(DTP_TILE_GEMMWS, sp, 5, &ms));
        if (ms >= 0.f) cands.push_back({ms, DTP_TILE_GEMMWS, sp});
      }
    }
    // second round: the three fastest candidates are usually within the measurement noise of each other -- time them again,
    // longer, and keep the minimum over both rounds
    std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.ms < y.ms; });
    for (size_t i = 0; i < cands.size() && i < 3; ++i) {
      float ms;
      RC(time_cfg(cands[i].tile, cands[i].sp, 8, &ms));
      if (ms >= 0.f) cands[i].ms = std::min(cands[i].ms, ms);
    }
    for (size_t i = 0; i < cands.size() && i < 3; ++i)
      if (cands[i].ms < best) { best = cands[i].ms; bt = cands[i].tile; bs = cands[i].sp; }
    // A conv is ranked by its own launch, but an UNSPLIT two-n-tile convws launch (tile 53 / 54) also delivers the GroupNorm statistics of
    // its output (Builder::claim_stats), i.e. it saves its consumer a statistics pass: one dispatch floor plus one read of the tensor.
    // Round 5 found the level-0 long-shortcut convs on the halo kernel by 1-2 us -- and 76 statistics launches per stamp behind them
    // (switched by hand in the shipped table: -0.4 % at batch 1; at batch 8 the halo kernel's lead is larger than the pass).  The
    // statistics-capable candidate is credited with that pass when its output is one claim_stats would take.
    // (the tune key does not know the consumer: an output of such a shape is followed by a GroupNorm everywhere in these networks --
    // conv_out has N = 4 / 3, which the predicate excludes -- but in the up path that GroupNorm runs over a concatenation and cannot claim)
    if (ws_any && !dtp_is_ws_tile(bt) && dtp_conv_output_can_carry_gn_stats(p)) {
      const float stats_ms = 0.005f + (float)((double)p.M * p.N * 2.0 / 4.0e12 * 1e3);  // dispatch floor + the tensor once at ~4 TB/s
      for (const Cand& cd : cands)
        if (cd.sp == 1 && (cd.tile == DTP_TILE_WS0 + 2 || cd.tile == DTP_TILE_WS0 + 3) && cd.ms - stats_ms < best) {
          best = cd.ms - stats_ms; bt = cd.tile; bs = 1;
        }
    }
    it = c->tuned.emplace(key, std::make_pair(bt, bs)).first;
    if (getenv("DTP_TUNE_REPORT")) {  // how much of the chosen configuration's time is the cold operands?
      GemmParams q = p;
      if (dtp_is_halo_tile(bt)) q.W = p.Wcb;
      dtp_split_k(p.nkb, bt, bs, &q.kb_per_split, &q.splits);
      if (bt == DTP_TILE_LNLIN) q.col_ranges = bs;
      q.part = c->ws;
      q.zero = c->zero;
      float hot = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        HIP_CHECK(hipEventRecord(c->tune_ev[0], 0));
        if (bt == DTP_TILE_LNLIN) RC(dtp_launch_lnlin(q, bs, 0));
        else if (bt == DTP_TILE_GEMMWS) RC(dtp_launch_gemm_ws(q, 0));
        else if (dtp_is_ws_tile(bt)) RC(dtp_launch_conv_ws(q, bt - DTP_TILE_WS0, 0));
        else if (dtp_is_halo_tile(bt)) RC(dtp_launch_conv_halo(q, dtp_halo_variant(bt), 0)); else RC(dtp_launch_gemm(q, bt, 0));
        HIP_CHECK(hipEventRecord(c->tune_ev[1], 0));
        HIP_CHECK(hipEventSynchronize(c->tune_ev[1]));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, c->tune_ev[0], c->tune_ev[1]));
        if (rep) hot = std::min(hot, t);
      }
      c->tune_ms[key] = std::make_pair(best, hot);
    }
  }
  {
    auto m = c->tune_ms.find(key);
    if (m != c->tune_ms.end()) {
      c->rep_cold_ms += m->second.first;
      c->rep_hot_ms += m->second.second;
      fprintf(stderr, "[tune] %s tile=%d sp=%d cold %.1f us hot %.1f us\n", key, it->second.first, it->second.second, m->second.first * 1e3,
              m->second.second * 1e3);
    }
  }
  *tile_out = it->second.first;
  dtp_split_k(p.nkb, it->second.first, it->second.second, &p.kb_per_split, &p.splits);
  if (it->second.first == DTP_TILE_LNLIN) p.col_ranges = it->second.second;
  return DTP_OK;
}

float* fp8_new_linear_scale(Ctx* c, int* slot1) {
  if (c->fp8_nslots + 1 > DTP_FP8_SLOTS) {  // (round-4 advisor: this fallback to the default scale used to be silent)
    static bool warned = false;
    if (!warned) { fprintf(stderr, "[dtp] fp8: the %d calibration slots are used up -- further fp8 operands keep the default activation scale\n", DTP_FP8_SLOTS); warned = true; }
    *slot1 = 0;
    return nullptr;
  }
  c->fp8_scales.push_back(1.0f);
  Fp8Cal r;
  r.kind = 0; r.slot0 = c->fp8_nslots; r.s0 = &c->fp8_scales.back();
  c->fp8_cals.push_back(r);
  *slot1 = ++c->fp8_nslots;
  return r.s0;
}

// One evaluation of the program with every fp8 op also measuring the absolute maximum of its operands, then power-of-two scales:
// amax * margin / scale <= 448 (the largest e4m3 value), as large a mantissa use as that allows.  Synchronises the stream once.
int fp8_calibrate(Ctx* c, UNetProg* up, hipStream_t s, int step) {
  if (up->fp8_calibrated || up->cal_begin == up->cal_end) { up->fp8_calibrated = true; return DTP_OK; }
  if (!c->fp8_amax) { void* p; RC(ctx_persistent(c, DTP_FP8_SLOTS * sizeof(unsigned int), &p, true)); c->fp8_amax = (unsigned int*)p; }
  HIP_CHECK(hipMemsetAsync(c->fp8_amax, 0, DTP_FP8_SLOTS * sizeof(unsigned int), s));
  c->calibrating = true;
  const int rc = up->main.run(s, step);
  c->calibrating = false;
  RC(rc);
  std::vector<float> amax(DTP_FP8_SLOTS);
  HIP_CHECK(hipMemcpyAsync(amax.data(), c->fp8_amax, DTP_FP8_SLOTS * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  auto pow2_for = [](float a) { return a > 0.f ? exp2f(ceilf(log2f(a * DTP_FP8_MARGIN / 448.0f))) : 1.0f; };
  for (size_t i = up->cal_begin; i < up->cal_end; ++i) {
    Fp8Cal& r = c->fp8_cals[i];
    if (r.kind == 0) {
      *r.s0 = pow2_for(amax[r.slot0]);
    } else {
      // Q' = Q * (softmax_scale * log2 e * q_scale), K' = K / q_scale: balance the two absolute maxima (their product is fixed)
      const float aq = amax[r.slot0] * r.softmax_scale * 1.4426950408889634f, ak = amax[r.slot0 + 1], av = amax[r.slot0 + 2];
      *r.s0 = (aq > 0.f && ak > 0.f) ? exp2f(roundf(0.5f * log2f(ak / aq))) : 1.0f;
      *r.s1 = pow2_for(av);
    }
  }
  up->fp8_calibrated = true;
  return DTP_OK;
}

static Op make_gemm_op(Ctx* c, GemmParams p, int tile, int bias_step_off) {
  return [=](hipStream_t s, int step) -> int {
    GemmParams q = p;
    q.part = c->ws;
    if (p.a_scale_host) {  // fp8 with a calibrated activation scale
      if (c->calibrating && p.amax_slot1 > 0) {
        const int ka = p.A2 ? p.K - p.Cin2 : p.K;
        RC(dtp_launch_amax_f16(p.A, p.M, ka, p.lda, c->fp8_amax + p.amax_slot1 - 1, s));
        if (p.A2) RC(dtp_launch_amax_f16(p.A2, p.M, p.Cin2, p.lda2, c->fp8_amax + p.amax_slot1 - 1, s));
      }
      q.a_scale = *p.a_scale_host;
    }
    if (bias_step_off >= 0) q.bias = c->temb_table + (size_t)step * c->unet.temb_total + bias_step_off;
    if (tile == DTP_TILE_LNLIN) return dtp_launch_lnlin(q, q.col_ranges, s);
    if (tile == DTP_TILE_GEMMWS) return dtp_launch_gemm_ws(q, s);
    if (dtp_is_ws_tile(tile)) return dtp_launch_conv_ws(q, tile - DTP_TILE_WS0, s);
    if (dtp_is_halo_tile(tile)) { q.W = q.Wcb; return dtp_launch_conv_halo(q, dtp_halo_variant(tile), s); }
    return dtp_launch_gemm(q, tile, s);
  };
}

int push_gemm(Ctx* c, Prog* prog, GemmParams p, int bias_step_off, double k_alg, RowStats* emit) {
  int tile = 0;
  dtp_gemm_pick(p, &tile, c->num_cu);
  if (p.W8) {  // fp8: unsplit, one of the four fp8 tiles
    p.splits = 1; p.kb_per_split = p.nkb;
    tile = 24 + ((p.flags & GF_GEGLU) ? (p.M >= 512 ? 0 : 3) : (p.M >= 512 ? 0 : 2));
  }
  if ((p.flags & GF_GNAPPLY) && (p.flags & GF_CONV3)) { tile = (p.Hi * p.Wi <= 256) ? 14 : 12; p.splits = 1; p.kb_per_split = p.nkb; }  // halo kernel only
  if ((p.flags & GF_GNAPPLY) && !(p.flags & GF_CONV3)) {  // dense: GroupNorm on the resident fragments of lnlin_kernel, nothing else applies it
    tile = DTP_TILE_LNLIN; p.splits = 1; p.kb_per_split = p.nkb;
    const int pick = lnlin_default_ranges(p);
    if (!pick) { dtp_set_error("push_gemm: no lnlin configuration for the GroupNorm-on-load Linear (M %d N %d K %d)", p.M, p.N, p.K); return DTP_ERR_ARG; }
    p.col_ranges = pick;
  }
  if (c->autotune) RC(tune_gemm(c, p, &tile));
  if (emit) {  // the consumer must know how many partials this launch configuration writes per row
    int bm = 0, bn = 128, ns = 0;
    (void)dtp_gemm_tile_dims(tile, &bm, &bn, &ns);
    if (tile == DTP_TILE_GEMMWS) bn = 64;
    emit->parts = tile == DTP_TILE_LNLIN ? p.col_ranges : p.splits > 1 ? 1 : (p.N + bn - 1) / bn;
    emit->M = p.M * (p.batch > 1 ? p.batch : 1);
  }
  c->ws_need = std::max(c->ws_need, dtp_gemm_workspace_bytes(p));
  p.zero = c->zero;
  // algorithmic work: 2*M*N*K on the UNPADDED contraction; bytes = A once + W once + C once (fp16)
  const double n_out = (p.flags & GF_GEGLU) ? p.N / 2.0 : (double)p.N;
  const double a_elems = (p.flags & GF_CONV3) ? (double)p.M * (k_alg / 9.0) * ((p.flags & GF_UPS2) ? 0.25 : (double)(p.stride * p.stride))
                                               : (double)p.M * k_alg;
  const double nb = p.batch > 1 ? (double)p.batch : 1.0;
  const double bytes = 2.0 * nb * (a_elems + (double)p.N * k_alg + (double)p.M * n_out);
  char lab[160];
  const bool f8tile = tile >= 24 && tile < 32;  // an fp8 problem may have kept an fp16 tile (tune_gemm)
  snprintf(lab, sizeof(lab), "%s M=%d N=%d K=%d tile=%d splits=%d%s%s%s%s", (p.flags & GF_CONV3) ? "conv3" : "gemm", p.M, p.N, p.K, tile,
           tile == DTP_TILE_LNLIN ? p.col_ranges : p.splits, (p.flags & GF_UPS2) ? " ups" : "", (p.flags & GF_GEGLU) ? (f8tile ? " geglu fp8" : " geglu") : (f8tile ? " fp8" : ""), p.stride == 2 ? " s2" : "",
           p.batch > 1 ? (" x" + std::to_string(p.batch)).c_str() : "");
  const int kind = tile == DTP_TILE_GEMMWS ? PK_GEMMWS : dtp_is_ws_tile(tile) ? PK_WS0 + tile - DTP_TILE_WS0 : tile == DTP_TILE_LNLIN ? PK_LNLIN : tile >= 48 ? PK_HALO3 + tile - 48 : tile >= 40 ? PK_LW + tile - 40 : tile >= 32 ? PK_KH2 + tile - 32 : tile >= 24 ? PK_FP8 : tile >= 20 ? PK_WIDE0 + tile - 20 : tile >= 16 ? PK_BIG0 + tile - 16 : tile >= 12 ? PK_HALO0 + tile - 12 : PK_GEMM0 + tile;
  const double flops = 2.0 * nb * p.M * (double)p.N * k_alg;
  prog_push(c, prog, kind, flops, bytes, make_gemm_op(c, p, tile, bias_step_off), lab);
  if (p.splits > 1 || (dtp_is_ws_tile(tile) && tile >= DTP_TILE_WS0 + 2)) {  // a GroupNorm pushed next may take over the reduce (Builder::gn) -- or, behind an unsplit
    LastGemm& lg = prog->last_gemm;                // two-n-tile convws launch, get its statistics from the conv's epilogue (claim_stats)
    lg.valid = true; lg.p = p; lg.tile = tile; lg.bias_step_off = bias_step_off; lg.op_index = prog->ops.size() - 1;
    lg.kind = kind; lg.flops = flops; lg.bytes = bytes; lg.label = lab;
  }
  return DTP_OK;
}

int Builder::conv3(const T& x, const ConvW& w, int stride, int pad, bool ups, int Ho, int Wo, const T* resid,
                   int bias_step_off, T& y, int extra_flags, void* out_override, int ldc_override, const T* tail, const T* dst) {
  if ((w.cin2 > 0) != (tail != nullptr) || (tail && tail->C != w.cin2)) { dtp_set_error("conv3: shortcut tail mismatch"); return DTP_ERR_ARG; }
  if (x.C != w.cin || w.taps != 9) { dtp_set_error("conv3: channel mismatch %d vs %d", x.C, w.cin); return DTP_ERR_ARG; }
  GemmParams p = {};
  p.A = x.p; p.W = w.w;
  p.M = x.B * Ho * Wo; p.N = w.cout; p.K = w.K;
  p.lda = x.ld; p.ldw = w.ldw;
  p.nkb = w.ldw / 64;
  p.Hi = x.H; p.Wi = x.W; p.Ho = Ho; p.Wo = Wo; p.Cin = w.cin; p.stride = stride; p.pad = pad;
  p.flags = GF_CONV3 | (ups ? GF_UPS2 : 0) | extra_flags;
  if (tail) { p.A2 = tail->p; p.lda2 = tail->ld; p.Cin2 = w.cin2; }
  p.Wcb = w.wcb;
  p.Wfr = w.wfr;
  if (extra_flags & GF_GNAPPLY) {
    if (!gn_fused.active) { dtp_set_error("conv3: GF_GNAPPLY without GroupNorm parameters"); return DTP_ERR_ARG; }
    p.gn_part = gn_fused.part; p.gn_gamma = gn_fused.gamma; p.gn_beta = gn_fused.beta; p.gn_eps = gn_fused.eps;
    p.gn_nchunk = gn_fused.nchunk; p.gn_cpg = gn_fused.cpg; p.gn_silu = 1;
  }
  if (out_override) {
    y = T();
    y.p = (f16*)out_override; y.B = x.B; y.H = Ho; y.W = Wo; y.C = w.cout; y.ld = ldc_override;
  } else if (dst) {
    if (dst->C != w.cout || dst->rows() != (long long)x.B * Ho * Wo) { dtp_set_error("conv3: destination view mismatch"); return DTP_ERR_ARG; }
    y = *dst;
  } else {
    y = alloc(x.B, Ho, Wo, w.cout);
    if (!y.p) return DTP_ERR_HIP;
  }
  p.C = y.p; p.ldc = y.ld;
  if (w.b || bias_step_off >= 0) { p.flags |= GF_BIAS; p.bias = w.b; }
  if (resid) { p.flags |= GF_RESID; p.R = resid->p; p.ldr = resid->ld; }
  return push_gemm(c, prog, p, bias_step_off, 9.0 * w.cin_true + w.cin2);
}

// GroupNorm (+ SiLU) followed by a 3x3 conv.  Where the conv can run on the halo kernel and the GroupNorm is the two-launch kind
// (maps of >= 1024 pixels), the apply pass is folded into the conv (GF_GNAPPLY, conv_halo.hip): statistics pass (+ the producer's
// split-K reduce) -> conv on the RAW tensor.  Otherwise: gn() + conv3().
int Builder::gn_conv3(const T& x, const NormW& n, float eps, const ConvW& w, const T* resid, int bias_step_off, T& y, const T* tail, const T* dst) {
  Ctx* cc = c;
  const int HW = x.H * x.W, C = x.C;
  // (bigger problems -- batched stamps, the VAE at 512^2 -- are not launch-bound: there the apply pass costs less than what the
  // normalisation adds to every workgroup of the conv, and their tuned tiles are not the halo kernel's)
  const bool fuse = cc->fuse_gn_conv && HW >= 1024 && x.rows() <= 16384 && w.wcb && w.taps == 9 && (C & 63) == 0 && C <= 1024 && x.C == w.cin && (C % 32) == 0 && C / 32 >= 8 &&
                    (w.cout & 7) == 0 && (x.ld & 7) == 0;
  if (!fuse) {
    T t;
    RC(gn(x, n, eps, true, t));
    RC(conv3(t, w, 1, 1, false, x.H, x.W, resid, bias_step_off, y, 0, nullptr, 0, tail, dst));
    release(t);
    return DTP_OK;
  }
  GemmParams gp;
  int bso = -1;
  const bool claimed = claim_reduce(x, gp, bso);
  const size_t slab_bytes = claimed ? ((dtp_gemm_workspace_bytes(gp) + 255) & ~(size_t)255) : 0;
  // the partial sums outlive the statistics launch (the conv's workgroups read them while other workgroups may already write
  // split-K slabs into the shared workspace): they get their own planned buffer
  // (the claimed reduce's residual may be back in the pool already and the reduce + statistics launch that writes the partials still reads
  // it: round-5 advisor, the same aliasing as in gn() / claim_stats)
  std::vector<MemRange> busy;
  if (claimed && (gp.flags & GF_RESID)) busy.push_back({gp.R, (size_t)gp.M * gp.ldr * sizeof(f16)});
  void* pp = pool_get_clear_of(cc, dtp_groupnorm_ws_bytes(x.B, HW, C, 32), busy);
  if (!pp) { dtp_set_error("gn_conv3: no partials block clear of the claimed reduce's residual"); return DTP_ERR_HIP; }
  float* partials = (float*)pp;
  cc->ws_need = std::max(cc->ws_need, slab_bytes);
  const T xx = x;
  const bool has_bias = claimed && (gp.flags & GF_BIAS) != 0;
  push(PK_GN, 0.0, 2.0 * (double)xx.rows() * C, [=](hipStream_t s, int step) {
    if (claimed) {
      GnReduceSrc rd;
      rd.part = cc->ws; rd.splits = gp.splits; rd.slab = (long long)gp.M * gp.N; rd.ldp = gp.N;
      rd.bias = !has_bias ? nullptr : (bso >= 0 ? cc->temb_table + (size_t)step * cc->unet.temb_total + bso : gp.bias);
      rd.R = (gp.flags & GF_RESID) ? gp.R : nullptr; rd.ldr = gp.ldr;
      return dtp_launch_groupnorm_stats(xx.p, xx.ld, partials, xx.B, HW, C, 32, &rd, s);
    }
    return dtp_launch_groupnorm_stats(xx.p, xx.ld, partials, xx.B, HW, C, 32, nullptr, s);
  }, std::string(claimed ? "reduce+gn-stats" : "gn-stats") + " B=" + std::to_string(x.B) + " HW=" + std::to_string(HW) + " C=" + std::to_string(C) + " (apply in conv)");
  gn_fused.part = partials; gn_fused.gamma = n.g; gn_fused.beta = n.b; gn_fused.eps = eps; gn_fused.nchunk = dtp_groupnorm_stat_chunks(HW); gn_fused.cpg = C / 32;
  gn_fused.active = true;
  const int rc = conv3(x, w, 1, 1, false, x.H, x.W, resid, bias_step_off, y, GF_GNAPPLY, nullptr, 0, tail, dst);
  gn_fused.active = false;
  ctx_pool_put(cc, pp);
  return rc;
}

int Builder::alloc_stats(long long rows, int C, RowStats& st) {
  void* p = nullptr;
  RC(ctx_pool_get(c, (size_t)((C + 63) / 64) * rows * 2 * sizeof(float), &p));
  st.buf = (float*)p;
  st.parts = 0;
  st.M = (int)rows;
  return DTP_OK;
}
void Builder::release_stats(RowStats& st) { ctx_pool_put(c, st.buf); st.buf = nullptr; }

int Builder::linear(const T& x, const ConvW& w, const T* resid, int flags, T& y, RowStats* emit, const RowStats* use, const T* dst) {
  if (x.C != w.K || w.taps != 1) { dtp_set_error("linear: K mismatch %d vs %d", x.C, w.K); return DTP_ERR_ARG; }
  GemmParams p = {};
  p.A = x.p; p.W = w.w; p.Wfr = w.wfr;
  p.M = (int)x.rows(); p.N = w.cout; p.K = w.K;
  p.lda = x.ld; p.ldw = w.ldw; p.nkb = w.ldw / 64;
  p.flags = flags;
  if (dst) {
    if (dst->C != w.cout || (flags & GF_GEGLU) || dst->rows() != x.rows()) { dtp_set_error("linear: destination view mismatch"); return DTP_ERR_ARG; }
    y = *dst;
  } else {
    y = alloc(x.B, x.H, x.W, (flags & GF_GEGLU) ? w.cout / 2 : w.cout);
    if (!y.p) return DTP_ERR_HIP;
  }
  p.C = y.p; p.ldc = y.ld;
  if (w.b) { p.flags |= GF_BIAS; p.bias = w.b; }
  if (w.lns) {  // x is the raw pre-LayerNorm tensor
    p.flags |= GF_LNFOLD; p.lns = w.lns; p.ln_eps = 1e-5f;
    if (use && use->buf && use->parts > 0 && use->M == p.M) { p.st_in = use->buf; p.st_parts = use->parts; }
  }
  if (resid) { p.flags |= GF_RESID; p.R = resid->p; p.ldr = resid->ld; }
  if (emit && emit->buf) {
    p.flags |= GF_ROWSTATS; p.st_out = emit->buf;
    if (emit->rows_total > 0) { p.st_out = emit->buf + (size_t)emit->row_off * 2; p.st_rows = emit->rows_total; }
  }
  if (fp8 && w.w8) {
    GemmParams q = p;
    q.W8 = w.w8; q.ldw8 = w.ldw8; q.w_scale = w.w8_scale; q.splits = 1;
    // activation scale: a LayerNorm'd operand is bounded (fixed scale); anything else is calibrated (fp8_calibrate)
    q.a_scale = (p.flags & GF_LNFOLD) ? DTP_FP8_LN_A_SCALE : DTP_FP8_LN_A_SCALE * 8.0f;
    if (dtp_gemm_fp8_supported(q)) {
      if (!(p.flags & GF_LNFOLD) && (q.K & 7) == 0) q.a_scale_host = fp8_new_linear_scale(c, &q.amax_slot1);
      p = q;  // otherwise the fp16 kernel takes it
    }
  }
  return push_gemm(c, prog, p, -1, (double)w.K, (emit && emit->buf) ? emit : nullptr);
}

int Builder::attention(const T& q, const T& k, const T& v, int heads, int Sq, int Skv, int Bn, T& o) {
  o = alloc(Bn, 1, Sq, q.C);
  if (!o.p) return DTP_ERR_HIP;
  AttnParams a;
  a.Q = q.p; a.K = k.p; a.V = v.p; a.O = o.p;
  a.ldq = q.ld; a.ldk = k.ld; a.ldv = v.ld; a.ldo = o.ld;
  a.B = Bn; a.H = heads; a.Sq = Sq; a.Skv = Skv; a.D = q.C / heads;
  a.qbs = (long long)Sq * q.ld; a.kbs = (long long)Skv * k.ld; a.vbs = (long long)Skv * v.ld; a.obs = (long long)Sq * o.ld;
  a.scale = 1.0f / sqrtf((float)a.D);
  const bool fp8 = c->fp8_attention && (a.D % 64) != 0 && a.D <= 184 && Skv >= 64;  // the brush encoder's tiny attentions stay f16
  const float *qs = nullptr, *vs = nullptr;
  int slot0 = -1;
  if (fp8 && c->fp8_nslots + 3 <= DTP_FP8_SLOTS && (q.C & 7) == 0) {  // calibrated Q / K and V scales (fp8_calibrate)
    Ctx* cc0 = c;
    cc0->fp8_scales.push_back(1.0f); float* s0 = &cc0->fp8_scales.back();
    cc0->fp8_scales.push_back(1.0f); float* s1 = &cc0->fp8_scales.back();
    Fp8Cal r;
    r.kind = 1; r.slot0 = cc0->fp8_nslots; r.s0 = s0; r.s1 = s1; r.softmax_scale = a.scale;
    cc0->fp8_cals.push_back(r);
    slot0 = cc0->fp8_nslots;
    cc0->fp8_nslots += 3;
    qs = s0; vs = s1;
  }
  Ctx* cc = c;
  const T qq = q, kk = k, vv = v;
  push(PK_ATTN, 4.0 * Bn * heads * (double)Sq * Skv * a.D, 2.0 * Bn * q.C * (2.0 * Sq + 2.0 * Skv),
       [=](hipStream_t s, int) {
         if (!fp8) return dtp_launch_attention(a, s);
         if (cc->calibrating && slot0 >= 0) {
           RC(dtp_launch_amax_f16(qq.p, (long long)Bn * Sq, qq.C, qq.ld, cc->fp8_amax + slot0, s));
           RC(dtp_launch_amax_f16(kk.p, (long long)Bn * Skv, kk.C, kk.ld, cc->fp8_amax + slot0 + 1, s));
           RC(dtp_launch_amax_f16(vv.p, (long long)Bn * Skv, vv.C, vv.ld, cc->fp8_amax + slot0 + 2, s));
         }
         return dtp_launch_attention_fp8(a, qs ? *qs : 1.0f, vs ? *vs : 1.0f, s);
       },
       "attn B=" + std::to_string(Bn) + " Sq=" + std::to_string(Sq) + " Skv=" + std::to_string(Skv) + " D=" + std::to_string(a.D));
  return DTP_OK;
}

int Builder::concat(const T& a, const T& b, T& y) {
  y = alloc(a.B, a.H, a.W, a.C + b.C);
  if (!y.p) return DTP_ERR_HIP;
  const T aa = a, bb = b, yy = y;
  push(PK_ELEM, 0.0, 4.0 * (double)aa.rows() * (aa.C + bb.C), [=](hipStream_t s, int) {
    return dtp_launch_concat_channels(aa.p, aa.ld, aa.C, bb.p, bb.ld, bb.C, yy.p, yy.ld, aa.rows(), s);
  });
  return DTP_OK;
}

int Builder::resnet(const T& x, const ResW& w, float eps, bool temb, T& y, const T* dst) {
  T h;
  RC(gn_conv3(x, w.n1, eps, w.c1, nullptr, temb ? w.temb_off : -1, h, nullptr, nullptr));
  if (w.has_sc) {  // conv2 and the 1x1 shortcut are one contraction: [im2col(GN(h)) | x] . [W2 | Wsc]^T
    RC(gn_conv3(h, w.n2, eps, w.c2, nullptr, -1, y, &x, dst));
  } else {
    RC(gn_conv3(h, w.n2, eps, w.c2, &x, -1, y, nullptr, dst));
  }
  release(h);
  return DTP_OK;
}

// ---------------------------------------------------------------- C ABI: lifecycle + weights
extern "C" {

int dtp_create(int device, int resolution, int max_batch, dtp_ctx** out) {
  if (!out || resolution < 64 || resolution % 64 || max_batch < 1 || max_batch > 64) {
    dtp_set_error("dtp_create: resolution must be a positive multiple of 64, 1 <= max_batch <= 64");
    return DTP_ERR_ARG;
  }
  HIP_CHECK(hipSetDevice(device));
  Ctx* c = new Ctx();
  c->device = device; c->R = resolution; c->h = resolution / 8; c->maxB = max_batch;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount;
  void* z;
  HIP_CHECK(hipMalloc(&z, 4096));
  HIP_CHECK(hipMemset(z, 0, 4096));
  c->zero = (f16*)z;
  for (int i = 0; i < 4; ++i) HIP_CHECK(hipEventCreate(&c->ev[i]));
  tune_cache_load(c);
  if (const char* e = getenv("DTP_NO_FUSE_REDUCE_GN")) c->fuse_reduce_gn = !(e[0] && e[0] != '0');
  if (const char* e = getenv("DTP_NO_DEDUPE")) c->dedupe_prefix = !(e[0] && e[0] != '0');
#ifdef DTP_EXPERIMENTAL
  if (const char* e = getenv("DTP_GN_CONV")) c->fuse_gn_conv = (e[0] && e[0] != '0');
#endif
  if (const char* e = getenv("DTP_NO_XATTN")) c->fuse_xattn = !(e[0] && e[0] != '0');
  if (const char* e = getenv("DTP_NO_FOLD_GN")) c->fold_gn_linear = !(e[0] && e[0] != '0');
  *out = (dtp_ctx*)c;
  return DTP_OK;
}

void dtp_destroy(dtp_ctx* ctx) {
  Ctx* c = (Ctx*)ctx;
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& g : c->graphs) {
    if (g.second.exec) (void)hipGraphExecDestroy(g.second.exec);
    if (g.second.graph) (void)hipGraphDestroy(g.second.graph);
  }
  for (auto& s : c->staged) (void)hipFree(s.second.d);
  for (void* p : c->chunks) (void)hipFree(p);
  for (auto& b : c->pool.blocks) (void)hipFree(b.p);
  for (void* p : c->persistent) (void)hipFree(p);
  if (c->ws) (void)hipFree(c->ws);
  if (c->zero) (void)hipFree(c->zero);
  if (c->tune_thrash) (void)hipFree(c->tune_thrash);
  for (int i = 0; i < 4; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  delete c;
}

int dtp_load_tensor(dtp_ctx* ctx, const char* name, const float* data, int is_device, const int64_t* shape, int ndim) {
  Ctx* c = (Ctx*)ctx;
  if (!c || !name || !data || ndim < 1 || ndim > 4) { dtp_set_error("dtp_load_tensor: bad argument"); return DTP_ERR_ARG; }
  if (c->finalized) { dtp_set_error("dtp_load_tensor: weights already finalized"); return DTP_ERR_STATE; }
  HIP_CHECK(hipSetDevice(c->device));
  Staged s;
  s.n = 1;
  for (int i = 0; i < ndim; ++i) { s.shape.push_back(shape[i]); s.n *= (size_t)shape[i]; }
  HIP_CHECK(hipMalloc(&s.d, std::max<size_t>(s.n * 4, 16)));
  HIP_CHECK(hipMemcpy(s.d, data, s.n * 4, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  auto it = c->staged.find(name);
  if (it != c->staged.end()) { (void)hipFree(it->second.d); c->staged.erase(it); }
  c->staged.emplace(name, std::move(s));
  return DTP_OK;
}

}  // extern "C"
