// dex_text.hip — C ABI of the text path (include/dex_amd.h, dex_text_*): TextEncoder.forward (GeDEX-TTS/model/text_encoder.py:129-146;
// ConvReluNorm :34-67, DurationPredictor :70-93), RetNetModel in its parallel form (retnet.py:56-178; RetNetDecoderLayer
// retention.py:446-501, MultiScaleRetention :182-294 with use_softmax, GLU :357-390, RetNetRelPos :67-166 with use_decay off),
// DEX's AdaptiveLayerNorm after each residual sum (DEX-TTS/model/retention.py:489-509, base.py:161-194) and the duration /
// alignment lines of the TTS forward (tts.py:37-50, utils.py:26-39).
//
// Channels-last activations [B*T][ld].  Every Linear / Conv1d is an implicit GEMM on the exact-fp32 MFMA kernel (x * mask while
// gathering, bias / ReLU / residual / * mask in its epilogue); q|k|v|g and gate|fc1 are one GEMM each; the retention core is the
// fp32 softmax-attention kernel on head-padded operands with the utterance length as key bound (masked keys get exp(-1e4 - max)
// = 0 in the reference: the same thing); norms, rotation, gates and the path are the small kernels of text_elem.hip.
// ~150 launches over a few hundred token rows, once per utterance.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/dex_amd.h"
#include "kernels.h"

using namespace dex;

namespace {
struct TRaw { float* p = nullptr; std::vector<int64_t> shape; long numel = 0; bool loaded = false; };
struct TConv { const float* w = nullptr; const float* b = nullptr; int cin, cout, k; };      // packed [k*cin][cout], bias or null
struct TLayer { const float *rln, *fln, *wqkvg, *wout, *wgf, *wfc2; const float *a1sw, *a1sb, *a1bw, *a1bb, *a2sw, *a2sb, *a2bw, *a2bb; };
constexpr int HP = 128;               // padded head width of the attention operands
}  // namespace

struct DexText {
    DexTextConfig cfg{};
    std::string err;
    std::vector<std::string> keys;
    std::map<std::string, TRaw> raw;
    std::vector<void*> owned;
    bool finalized = false;
    int E = 0, kd = 0;
    TConv pre[3], pre_proj, dp1, dp2;
    std::vector<TLayer> layers;
    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
    const float* R(const std::string& k) const { return raw.at(k).p; }
};

#define TCHK(v, call)                                                                                  \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                 \
        return (v)->fail(DEX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

namespace {
void tkey(DexText* v, const std::string& k, std::vector<int64_t> shape) {
    v->keys.push_back(k);
    TRaw r; r.shape = std::move(shape); r.numel = 1;
    for (auto d : r.shape) r.numel *= d;
    v->raw[k] = r;
}
}  // namespace

extern "C" {

int dex_text_create(const DexTextConfig* cfg, DexText** out) {
    if (!cfg || !out) return DEX_ERR_ARG;
    DexText* v = new DexText();
    v->cfg = *cfg;
    *out = v;
    const DexTextConfig& c = v->cfg;
    if (c.variant != DEX_VARIANT_GEDEX && c.variant != DEX_VARIANT_DEX) return v->fail(DEX_ERR_ARG, "variant must be DEX_VARIANT_GEDEX or DEX_VARIANT_DEX");
    if (c.use_softmax != 1 || c.use_decay != 0) return v->fail(DEX_ERR_ARG, "only use_softmax = 1 / use_decay = 0 is built (what every shipped config sets)");
    const int E = c.n_channels + (c.n_spks > 1 ? c.spk_emb_dim : 0);
    v->E = E;
    if (c.n_vocab < 1 || c.n_feats < 1 || c.n_feats > 256 || c.n_layers < 1 || c.n_layers > 32) return v->fail(DEX_ERR_ARG, "n_vocab / n_feats / n_layers out of range");
    if (c.n_channels % 64 || c.n_channels < 64 || E % 64 || E > 256) return v->fail(DEX_ERR_ARG, "n_channels and the RetNet width must be multiples of 64, at most 256 (got %d, %d)", c.n_channels, E);
    if (c.filter_channels % 64 || c.filter_channels < 64 || c.filter_channels_dp % 64 || c.filter_channels_dp > 256) return v->fail(DEX_ERR_ARG, "filter_channels must be a multiple of 64, filter_channels_dp a multiple of 64 up to 256");
    if (c.n_heads < 1 || E % c.n_heads || (E / c.n_heads) % 2 || E / c.n_heads > HP) return v->fail(DEX_ERR_ARG, "head width %d / %d must be even and at most %d", E, c.n_heads, HP);
    if (c.kernel_size != 3 && c.kernel_size != 5 && c.kernel_size != 1) return v->fail(DEX_ERR_ARG, "kernel_size must be 1, 3 or 5");
    if (c.variant == DEX_VARIANT_DEX && c.n_spks > 1) return v->fail(DEX_ERR_ARG, "the DEX text encoder takes the style vector, not a speaker embedding");
    v->kd = E / c.n_heads;
    const int nc = c.n_channels, F = c.filter_channels, D = c.filter_channels_dp;
    tkey(v, "emb.weight", {c.n_vocab, nc});
    for (int i = 0; i < 3; ++i) {
        const std::string s = std::to_string(i);
        tkey(v, "prenet.conv_layers." + s + ".weight", {nc, nc, 5}); tkey(v, "prenet.conv_layers." + s + ".bias", {nc});
        tkey(v, "prenet.norm_layers." + s + ".gamma", {nc}); tkey(v, "prenet.norm_layers." + s + ".beta", {nc});
    }
    tkey(v, "prenet.proj.weight", {nc, nc, 1}); tkey(v, "prenet.proj.bias", {nc});
    for (int i = 0; i < c.n_layers; ++i) {
        const std::string p = "encoder.layers." + std::to_string(i);
        for (const char* n : {"q_proj", "k_proj", "v_proj", "g_proj", "out_proj"}) tkey(v, p + ".retention." + n + ".weight", {E, E});
        tkey(v, p + ".retention_layer_norm.weight", {E});
        tkey(v, p + ".ffn.fc1.weight", {F, E}); tkey(v, p + ".ffn.fc2.weight", {E, F}); tkey(v, p + ".ffn.gate.weight", {F, E});
        tkey(v, p + ".final_layer_norm.weight", {E});
        if (c.variant == DEX_VARIANT_DEX)
            for (const char* a : {"adaln_1", "adaln_2"})
                for (const char* w : {"W_scale", "W_bias"}) { tkey(v, p + "." + a + "." + w + ".weight", {E, E}); tkey(v, p + "." + a + "." + w + ".bias", {E}); }
    }
    tkey(v, "encoder.layer_norm.weight", {E});
    tkey(v, "encoder.retnet_rel_pos.angle", {v->kd});
    tkey(v, "proj_m.weight", {c.n_feats, E, 1}); tkey(v, "proj_m.bias", {c.n_feats});
    tkey(v, "proj_w.conv_1.weight", {D, E, c.kernel_size}); tkey(v, "proj_w.conv_1.bias", {D});
    tkey(v, "proj_w.norm_1.gamma", {D}); tkey(v, "proj_w.norm_1.beta", {D});
    tkey(v, "proj_w.conv_2.weight", {D, D, c.kernel_size}); tkey(v, "proj_w.conv_2.bias", {D});
    tkey(v, "proj_w.norm_2.gamma", {D}); tkey(v, "proj_w.norm_2.beta", {D});
    tkey(v, "proj_w.proj.weight", {1, D, 1}); tkey(v, "proj_w.proj.bias", {1});
    return DEX_OK;
}

void dex_text_destroy(DexText* v) {
    if (!v) return;
    for (auto& kv : v->raw) if (kv.second.p) hipFree(kv.second.p);
    for (void* p : v->owned) hipFree(p);
    delete v;
}
const char* dex_text_last_error(const DexText* v) { return v ? v->err.c_str() : "null text context"; }
int dex_text_num_weights(const DexText* v) { return v ? (int)v->keys.size() : 0; }
int dex_text_weight_info(const DexText* v, int i, const char** key, int64_t shape[4], int* ndim) {
    if (!v || i < 0 || i >= (int)v->keys.size()) return DEX_ERR_ARG;
    const TRaw& r = v->raw.at(v->keys[i]);
    if (key) *key = v->keys[i].c_str();
    if (ndim) *ndim = (int)r.shape.size();
    if (shape) for (size_t k = 0; k < r.shape.size(); ++k) shape[k] = r.shape[k];
    return DEX_OK;
}
int dex_text_load_weight_async(DexText* v, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream) {
    if (!v || !key || !w_dev) return DEX_ERR_ARG;
    auto it = v->raw.find(key);
    if (it == v->raw.end()) return v->fail(DEX_ERR_ARG, "unknown text-encoder weight key '%s'", key);
    TRaw& r = it->second;
    if ((int)r.shape.size() != ndim) return v->fail(DEX_ERR_ARG, "weight '%s': expected %d dims, got %d", key, (int)r.shape.size(), ndim);
    for (int k = 0; k < ndim; ++k)
        if (r.shape[k] != shape[k]) return v->fail(DEX_ERR_ARG, "weight '%s': dim %d is %lld, expected %lld", key, k, (long long)shape[k], (long long)r.shape[k]);
    if (!r.p) TCHK(v, hipMalloc((void**)&r.p, r.numel * sizeof(float)));
    TCHK(v, hipMemcpyAsync(r.p, w_dev, r.numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    r.loaded = true;
    v->finalized = false;
    return DEX_OK;
}

int dex_text_finalize(DexText* v, dex_stream_t stream) {
    if (!v) return DEX_ERR_ARG;
    for (const auto& k : v->keys)
        if (!v->raw.at(k).loaded) return v->fail(DEX_ERR_STATE, "text-encoder weight '%s' was never loaded", k.c_str());
    for (void* p : v->owned) hipFree(p);
    v->owned.clear();
    hipStream_t st = (hipStream_t)stream;
    const DexTextConfig& c = v->cfg;
    const int E = v->E, F = c.filter_channels;
    int rc = DEX_OK;
    auto alloc = [&](long n) -> float* {
        float* p = nullptr;
        if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { rc = v->fail(DEX_ERR_HIP, "hipMalloc of %ld floats failed", n); return nullptr; }
        v->owned.push_back(p);
        return p;
    };
    // Conv1d / Linear weight [cout][cin][k] -> GEMM operand [(tap*cin + ci)][cout]
    auto conv = [&](const std::string& wkey, const char* bkey, int cin, int cout, int k) {
        TConv o{}; o.cin = cin; o.cout = cout; o.k = k;
        float* t = alloc((long)k * cin * cout);
        if (t) launch_permute4(v->R(wkey), t, cout, cin, k, 1, 2, 1, 0, 3, st);
        o.w = t; o.b = bkey ? v->R(bkey) : nullptr;
        return o;
    };
    // several Linear weights [n_i][K] side by side as one operand [K][sum n_i]
    auto concat_t = [&](std::vector<std::string> ks, int K, int n_each) -> const float* {
        const int N = n_each * (int)ks.size();
        float* d = alloc((long)K * N);
        for (size_t j = 0; j < ks.size(); ++j) {
            float* tmp = alloc((long)K * n_each);
            if (d && tmp) {
                launch_permute4(v->R(ks[j]), tmp, n_each, K, 1, 1, 1, 0, 2, 3, st);
                hipMemcpy2DAsync(d + j * n_each, (size_t)N * 4, tmp, (size_t)n_each * 4, (size_t)n_each * 4, K, hipMemcpyDeviceToDevice, st);
            }
        }
        return d;
    };
    for (int i = 0; i < 3; ++i) {
        const std::string s = std::to_string(i), b = "prenet.conv_layers." + s + ".bias";
        v->pre[i] = conv("prenet.conv_layers." + s + ".weight", b.c_str(), c.n_channels, c.n_channels, 5);
    }
    v->pre_proj = conv("prenet.proj.weight", "prenet.proj.bias", c.n_channels, c.n_channels, 1);
    v->layers.clear();
    for (int i = 0; i < c.n_layers; ++i) {
        const std::string p = "encoder.layers." + std::to_string(i);
        TLayer L{};
        L.rln = v->R(p + ".retention_layer_norm.weight"); L.fln = v->R(p + ".final_layer_norm.weight");
        L.wqkvg = concat_t({p + ".retention.q_proj.weight", p + ".retention.k_proj.weight", p + ".retention.v_proj.weight", p + ".retention.g_proj.weight"}, E, E);
        L.wout = concat_t({p + ".retention.out_proj.weight"}, E, E);
        L.wgf = concat_t({p + ".ffn.gate.weight", p + ".ffn.fc1.weight"}, E, F);
        L.wfc2 = concat_t({p + ".ffn.fc2.weight"}, F, E);
        if (c.variant == DEX_VARIANT_DEX) {
            L.a1sw = v->R(p + ".adaln_1.W_scale.weight"); L.a1sb = v->R(p + ".adaln_1.W_scale.bias");
            L.a1bw = v->R(p + ".adaln_1.W_bias.weight"); L.a1bb = v->R(p + ".adaln_1.W_bias.bias");
            L.a2sw = v->R(p + ".adaln_2.W_scale.weight"); L.a2sb = v->R(p + ".adaln_2.W_scale.bias");
            L.a2bw = v->R(p + ".adaln_2.W_bias.weight"); L.a2bb = v->R(p + ".adaln_2.W_bias.bias");
        }
        v->layers.push_back(L);
    }
    v->dp1 = conv("proj_w.conv_1.weight", "proj_w.conv_1.bias", E, c.filter_channels_dp, c.kernel_size);
    v->dp2 = conv("proj_w.conv_2.weight", "proj_w.conv_2.bias", c.filter_channels_dp, c.filter_channels_dp, c.kernel_size);
    if (rc != DEX_OK) return rc;
    TCHK(v, hipStreamSynchronize(st));
    TCHK(v, hipGetLastError());
    v->finalized = true;
    return DEX_OK;
}

}  // extern "C"

namespace {
struct TPlan { float *mask, *h0, *h1, *a, *qkvg, *Q, *K, *V, *O, *gf, *ff, *ada, *d0, *d1, *mu, *logw, *cum; size_t bytes; };
void text_plan(const DexText* v, int B, int T, void* ws, TPlan& P) {
    const DexTextConfig& c = v->cfg;
    const size_t rows = (size_t)B * T;
    const int E = v->E;
    char* base = (char*)ws; size_t off = 0;
    auto take = [&](size_t n) { off = (off + 255) & ~size_t(255); float* p = ws ? (float*)(base + off) : nullptr; off += n * sizeof(float); return p; };
    P.mask = take(rows);
    P.h0 = take(rows * E); P.h1 = take(rows * E); P.a = take(rows * E);
    P.qkvg = take(rows * 4 * E);
    P.Q = take(rows * c.n_heads * HP); P.K = take(rows * c.n_heads * HP); P.V = take(rows * c.n_heads * HP); P.O = take(rows * c.n_heads * HP);
    P.gf = take(rows * 2 * c.filter_channels); P.ff = take(rows * c.filter_channels);
    P.ada = take((size_t)4 * B * E);
    P.d0 = take(rows * c.filter_channels_dp); P.d1 = take(rows * c.filter_channels_dp);
    P.mu = take(rows * c.n_feats); P.logw = take(rows); P.cum = take(rows);
    P.bytes = (off + 255) & ~size_t(255);
}
// Conv1d(k, padding k/2) / Linear (k = 1) on [B][T][lda] -> [B][T][ldc]; act 0 / 2 (ReLU); res added before the mask
void conv1d(const float* X, int lda, int T, int B, const TConv& c, const float* inmask, int act, const float* res, int ldres, const float* outmask,
            float* out, int ldc, hipStream_t st) {
    IGemmP g{};
    g.A = X; g.lda = lda; g.a_bstride = (long)T * lda;
    g.Hi = 1; g.Wi = T; g.Cin = c.cin;
    g.KH = 1; g.KW = c.k; g.sh = 1; g.sw = 1; g.off_w = -(c.k - 1) / 2; g.step_h = 1; g.step_w = 1;
    g.Ho = 1; g.Wo = T;
    g.W = c.w; g.N = c.cout; g.K = c.k * c.cin; g.ksplit = 1; g.groups = 1; g.bias = c.b;
    g.C = out; g.ldc = ldc; g.c_bstride = (long)T * ldc;
    g.OHf = 1; g.OWf = T; g.osh = 1; g.osw = 1;
    g.inmask = inmask; g.inmask_ws = 1; g.outmask = outmask; g.outmask_ws = 1; g.mask_bstride = T; g.gate_nstride = 1;
    g.act = act;
    g.res = res; g.ldres = ldres; g.res_bstride = (long)T * ldres;
    g.B = B;
    launch_igemm(g, PREC_FP32, st);
}
void linear(const float* X, int lda, int T, int B, const float* W, int K, int N, const float* res, int ldres, float* out, int ldc, hipStream_t st) {
    TConv c{W, nullptr, K, N, 1};
    conv1d(X, lda, T, B, c, nullptr, 0, res, ldres, nullptr, out, ldc, st);
}
void row_norm(const float* X, int ldx, float* Y, int ldy, long rows, int C, int mode, const float* g, const float* b, float eps, int relu,
              const float* mask, int T, hipStream_t st) {
    RowNormP p{X, ldx, Y, ldy, rows, C, mode, g, b, eps, relu, mask, T};
    launch_row_norm(p, st);
}
}  // namespace

extern "C" {

size_t dex_text_workspace_bytes(const DexText* v, int B, int T) {
    if (!v || B < 1 || T < 1) return 0;
    TPlan P; text_plan(v, B, T, nullptr, P);
    return P.bytes;
}

int dex_text_encode(DexText* v, const DexTextArgs* a, dex_stream_t stream) {
    if (!v || !a) return DEX_ERR_ARG;
    if (!v->finalized) return v->fail(DEX_ERR_STATE, "dex_text_finalize has not been called");
    const DexTextConfig& c = v->cfg;
    if (a->B < 1 || a->T < 1) return v->fail(DEX_ERR_ARG, "B >= 1 and T >= 1");
    if (!a->tokens_dev || !a->lengths_dev || !a->mu_out_dev || !a->logw_out_dev || !a->w_ceil_out_dev || !a->y_lengths_out_dev || !a->workspace_dev)
        return v->fail(DEX_ERR_ARG, "null pointer in DexTextArgs");
    if (c.n_spks > 1 && !a->spk_dev) return v->fail(DEX_ERR_ARG, "n_spks > 1 needs spk_dev (the speaker embedding rows)");
    if (c.variant == DEX_VARIANT_DEX && !a->sty_dev) return v->fail(DEX_ERR_ARG, "the DEX text encoder needs sty_dev (the pooled style vector)");
    if (!(a->length_scale > 0.f)) return v->fail(DEX_ERR_ARG, "length_scale must be positive");
    if (((uintptr_t)a->workspace_dev & 255) != 0) return v->fail(DEX_ERR_ARG, "workspace must be 256-byte aligned");
    TPlan P; text_plan(v, a->B, a->T, nullptr, P);
    if (P.bytes > a->workspace_bytes) return v->fail(DEX_ERR_WORKSPACE, "text workspace too small: need %zu bytes, got %zu", P.bytes, a->workspace_bytes);
    text_plan(v, a->B, a->T, a->workspace_dev, P);
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, T = a->T, E = v->E, nc = c.n_channels, F = c.filter_channels, D = c.filter_channels_dp, H = c.n_heads, kd = v->kd;
    const long rows = (long)B * T;
    launch_len_mask(a->lengths_dev, P.mask, B, T, st);

    // ---- emb * sqrt(n_channels), ConvReluNorm prenet (text_encoder.py:130,135; :59-66): activations keep ld = E so that the
    // speaker columns can be appended in place
    launch_embed(a->tokens_dev, v->R("emb.weight"), P.h0, rows, nc, E, sqrtf((float)nc), c.n_vocab, st);
    const float* cur = P.h0;
    for (int i = 0; i < 3; ++i) {
        const std::string s = std::to_string(i);
        conv1d(cur, i == 0 ? E : nc, T, B, v->pre[i], P.mask, 0, nullptr, 0, nullptr, P.a, nc, st);
        float* dst = (i & 1) ? P.gf : P.ff;                        // (scratch: the FFN buffers are idle here)
        row_norm(P.a, nc, dst, nc, rows, nc, 0, v->R("prenet.norm_layers." + s + ".gamma"), v->R("prenet.norm_layers." + s + ".beta"), 1e-4f, 1, nullptr, T, st);
        cur = dst;
    }
    conv1d(cur, nc, T, B, v->pre_proj, nullptr, 0, P.h0, E, P.mask, P.h1, E, st);        // (x_org + proj(x)) * mask
    if (c.n_spks > 1) launch_bcast_cols(P.h1, E, nc, a->spk_dev, B, T, c.spk_emb_dim, st);
    float* h = P.h1; float* hn = P.h0;

    // ---- RetNet layers
    const float* angle = v->R("encoder.retnet_rel_pos.angle");
    for (int li = 0; li < c.n_layers; ++li) {
        const TLayer& L = v->layers[li];
        row_norm(h, E, P.a, E, rows, E, 1, L.rln, nullptr, 1e-6f, 0, nullptr, T, st);
        linear(P.a, E, T, B, L.wqkvg, E, 4 * E, nullptr, 0, P.qkvg, 4 * E, st);
        RetRotP rr{P.qkvg, 4 * E, rows, T, H, kd, E, angle, 1.f / sqrtf((float)kd), P.Q, P.K, P.V, H * HP};
        launch_ret_rotate(rr, st);
        AttnP at{};
        at.Q = P.Q; at.ldq = H * HP; at.qb = (long)T * H * HP; at.K = P.K; at.ldk = H * HP; at.kb = at.qb; at.V = P.V; at.ldv = H * HP; at.vb = at.qb;
        at.O = P.O; at.ldo = H * HP; at.ob = at.qb; at.Nq = T; at.Nk = T; at.kv_len = a->lengths_dev; at.kv_len_add = 0; at.heads = H; at.scale = 1.f; at.B = B;
        launch_attention(at, PREC_FP32, st);
        RetGateP rg{P.O, H * HP, P.qkvg, 4 * E, P.a, E, rows, H, kd, E, 1e-6f};
        launch_ret_gate(rg, st);
        linear(P.a, E, T, B, L.wout, E, E, h, E, hn, E, st);                                  // residual + out_proj(...)
        std::swap(h, hn);
        if (c.variant == DEX_VARIANT_DEX) {
            SmallLinP s1{a->sty_dev, nc, B, nc, L.a1sw, L.a1sb, E, P.ada, E, 0, 0};
            SmallLinP s2{a->sty_dev, nc, B, nc, L.a1bw, L.a1bb, E, P.ada + (long)B * E, E, 0, 0};
            launch_small_linear(s1, st); launch_small_linear(s2, st);
            row_norm(h, E, hn, E, rows, E, 2, P.ada, P.ada + (long)B * E, 1e-5f, 0, nullptr, T, st);
            std::swap(h, hn);
        }
        row_norm(h, E, P.a, E, rows, E, 1, L.fln, nullptr, 1e-6f, 0, nullptr, T, st);
        linear(P.a, E, T, B, L.wgf, E, 2 * F, nullptr, 0, P.gf, 2 * F, st);
        launch_glu(P.gf, P.ff, rows, F, st);
        linear(P.ff, F, T, B, L.wfc2, F, E, h, E, hn, E, st);
        std::swap(h, hn);
        if (c.variant == DEX_VARIANT_DEX) {
            SmallLinP s1{a->sty_dev, nc, B, nc, L.a2sw, L.a2sb, E, P.ada + 2L * B * E, E, 0, 0};
            SmallLinP s2{a->sty_dev, nc, B, nc, L.a2bw, L.a2bb, E, P.ada + 3L * B * E, E, 0, 0};
            launch_small_linear(s1, st); launch_small_linear(s2, st);
            row_norm(h, E, hn, E, rows, E, 2, P.ada + 2L * B * E, P.ada + 3L * B * E, 1e-5f, 0, nullptr, T, st);
            std::swap(h, hn);
        }
    }
    row_norm(h, E, P.a, E, rows, E, 1, v->R("encoder.layer_norm.weight"), nullptr, 1e-6f, 0, P.mask, T, st);      // final RMSNorm, * x_mask

    // ---- proj_m and the duration predictor (text_encoder.py:141-146, :81-93)
    SmallLinP pm{P.a, E, (int)rows, E, v->R("proj_m.weight"), v->R("proj_m.bias"), c.n_feats, P.mu, c.n_feats, 0, 0};
    launch_small_linear(pm, st);
    launch_cl_to_cf_mask(P.mu, c.n_feats, P.mask, a->mu_out_dev, B, T, c.n_feats, st);
    conv1d(P.a, E, T, B, v->dp1, P.mask, 2, nullptr, 0, nullptr, P.d0, D, st);
    row_norm(P.d0, D, P.d1, D, rows, D, 0, v->R("proj_w.norm_1.gamma"), v->R("proj_w.norm_1.beta"), 1e-4f, 0, nullptr, T, st);
    conv1d(P.d1, D, T, B, v->dp2, P.mask, 2, nullptr, 0, nullptr, P.d0, D, st);
    row_norm(P.d0, D, P.d1, D, rows, D, 0, v->R("proj_w.norm_2.gamma"), v->R("proj_w.norm_2.beta"), 1e-4f, 0, P.mask, T, st);   // ... * mask before proj
    SmallLinP pw{P.d1, D, (int)rows, D, v->R("proj_w.proj.weight"), v->R("proj_w.proj.bias"), 1, P.logw, 1, 0, 0};
    launch_small_linear(pw, st);
    launch_cl_to_cf_mask(P.logw, 1, P.mask, a->logw_out_dev, B, T, 1, st);                         // [B][1][T] == [B][T], * mask
    launch_durations(a->logw_out_dev, P.mask, a->length_scale, a->w_ceil_out_dev, P.cum, a->y_lengths_out_dev, B, T, st);
    TCHK(v, hipGetLastError());
    return DEX_OK;
}

int dex_text_align(DexText* v, const DexAlignArgs* a, dex_stream_t stream) {
    if (!v || !a) return DEX_ERR_ARG;
    if (a->B < 1 || a->T < 1 || a->Ty < 1) return v->fail(DEX_ERR_ARG, "B, T, Ty must be positive");
    if (!a->mu_x_dev || !a->w_ceil_dev || !a->x_lengths_dev || !a->y_lengths_dev || !a->mu_y_out_dev || !a->workspace_dev)
        return v->fail(DEX_ERR_ARG, "null pointer in DexAlignArgs");
    if (a->workspace_bytes < (size_t)a->B * a->T * sizeof(float)) return v->fail(DEX_ERR_WORKSPACE, "align workspace too small: need %zu bytes", (size_t)a->B * a->T * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* cum = (float*)a->workspace_dev;
    launch_cumsum_rows(a->w_ceil_dev, cum, a->B, a->T, st);
    AlignP p{a->mu_x_dev, cum, a->x_lengths_dev, a->y_lengths_dev, a->B, a->T, a->Ty, v->cfg.n_feats, a->mu_y_out_dev, a->y_mask_out_dev, a->attn_out_dev};
    launch_align(p, st);
    TCHK(v, hipGetLastError());
    return DEX_OK;
}

}  // extern "C"
