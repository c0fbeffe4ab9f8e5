// dex_style.hip — C ABI of the DEX style encoders (include/dex_amd.h, dex_style_*): DEX-TTS/model/ref_encoder.py
// (Projection :8-34, LF0Encoder :36-55, TIV/TVEncoderBlock :57-81, TIVEncoder :83-108, TVEncoder :110-140, VQEmbeddingEMA
// :199-237) and the pre-decoder lines of DeXTTS.forward (model/tts.py:55-66).
//
// Every Conv1d(k=3) is an implicit GEMM over channels-last activations [B][T][C] on the exact-fp32 MFMA kernel (x * mask on
// load, ReLU / residual / * mask in the epilogue; BatchNorm arrives folded into weight + bias); LayerNorms, InstanceNorm1D,
// the masked means, the VQ lookup and the GRU recurrence are the small kernels of style_elem.hip.  One call = ~70 launches
// over a few hundred frames: once per reference utterance, ahead of 50-100 sampler steps.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/dex_amd.h"
#include "kernels.h"

using namespace dex;

namespace {
struct SRaw { float* p = nullptr; std::vector<int64_t> shape; long numel = 0; bool loaded = false; };
struct SConv { const float* w = nullptr; const float* b = nullptr; int cin, cout, k; };      // packed [k*cin][cout], bias or null
struct SProj { SConv c1, c2, pr; const float *g1, *b1, *g2, *b2; };
constexpr int MEL_LD = 96, LF0_LD = 32;
}  // namespace

struct DexStyle {
    DexStyleConfig cfg{};
    std::string err;
    std::vector<std::string> keys;
    std::map<std::string, SRaw> raw;
    std::vector<void*> owned;
    bool finalized = false;
    SConv tiv_in; std::vector<SConv> tiv_b0, tiv_b1;
    SConv tv_in, tv_out, tv_p1; std::vector<SConv> tv_b0, tv_b1; SProj tv_p0;
    const float* embT = nullptr; const float* e2 = nullptr;       // codebook as a GEMM operand [D][M], |e|^2 [M]
    SConv lf_in, lf_out; SProj lf_proj;
    std::vector<const float*> gru_wih, gru_bih, gru_whh, gru_bhh;   // per layer: [in][2*3H], [2*3H], [2][3H][H], [2][3H]
    SConv sty;
    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
    const float* R(const std::string& k) const { return raw.at(k).p; }
};

#define SCHK(v, call)                                                                                  \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                 \
        return (v)->fail(DEX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

namespace {
void skey(DexStyle* v, const std::string& k, std::vector<int64_t> shape) {
    v->keys.push_back(k);
    SRaw r; r.shape = std::move(shape); r.numel = 1;
    for (auto d : r.shape) r.numel *= d;
    v->raw[k] = r;
}
void proj_keys(DexStyle* v, const std::string& p, int cin, int ch) {
    skey(v, p + ".conv_1.weight", {ch, cin, 3}); skey(v, p + ".conv_1.bias", {ch});
    skey(v, p + ".norm_1.gamma", {ch}); skey(v, p + ".norm_1.beta", {ch});
    skey(v, p + ".conv_2.weight", {ch, ch, 3}); skey(v, p + ".conv_2.bias", {ch});
    skey(v, p + ".norm_2.gamma", {ch}); skey(v, p + ".norm_2.beta", {ch});
    skey(v, p + ".proj.weight", {ch, ch, 1}); skey(v, p + ".proj.bias", {ch});
}
}  // namespace

extern "C" {

int dex_style_create(const DexStyleConfig* cfg, DexStyle** out) {
    if (!cfg || !out) return DEX_ERR_ARG;
    DexStyle* v = new DexStyle();
    v->cfg = *cfg;
    *out = v;
    const DexStyleConfig& c = v->cfg;
    if (c.n_mels < 1 || c.n_mels > MEL_LD) return v->fail(DEX_ERR_ARG, "n_mels must be in [1, %d]", MEL_LD);
    for (int ch : {c.tiv_ch, c.tv_ch, c.tv_cout, c.tv_cout_g, c.lf0_ch, c.lf0_cout, c.lf0_cout_g, c.sty_out})
        if (ch < 32 || ch % 32 || ch > 256) return v->fail(DEX_ERR_ARG, "channel counts must be multiples of 32 in [32, 256] (got %d)", ch);
    if (c.lf0_ch != 192) return v->fail(DEX_ERR_ARG, "the GRU kernel is built for hidden size 96 (lf0_encoder.c_h = 192)");
    if (c.tv_n_emb % 64 || c.tv_n_emb < 64) return v->fail(DEX_ERR_ARG, "n_emb must be a multiple of 64");
    if (c.lf0_cout != c.tv_cout || c.lf0_cout_g != c.tv_cout_g) return v->fail(DEX_ERR_ARG, "lf0 and tv encoder output widths must match (they are added, tts.py:62-65)");
    if (c.tiv_layers < 1 || c.tiv_layers > 8 || c.tv_layers < 1 || c.lf0_layers < 1 || c.lf0_layers > 4) return v->fail(DEX_ERR_ARG, "layer counts out of range");
    skey(v, "tiv_encoder.in_conv.conv.weight", {c.tiv_ch, c.n_mels, 3}); skey(v, "tiv_encoder.in_conv.conv.bias", {c.tiv_ch});
    for (int i = 0; i < c.tiv_layers; ++i) {
        const std::string p = "tiv_encoder.conv_blocks." + std::to_string(i) + ".conv_block";
        skey(v, p + ".0.conv.weight", {c.tiv_ch, c.tiv_ch, 3}); skey(v, p + ".0.conv.bias", {c.tiv_ch});
        skey(v, p + ".1.conv.weight", {c.tiv_ch, c.tiv_ch, 3});
    }
    skey(v, "tv_encoder.in_conv.conv.weight", {c.tv_ch, c.n_mels, 3});
    skey(v, "tv_encoder.in_conv.ln.weight", {c.tv_ch}); skey(v, "tv_encoder.in_conv.ln.bias", {c.tv_ch});
    for (int i = 0; i < c.tv_layers; ++i) {
        const std::string p = "tv_encoder.conv_blocks." + std::to_string(i) + ".conv_block";
        skey(v, p + ".0.conv.weight", {c.tv_ch, c.tv_ch, 3});
        skey(v, p + ".0.ln.weight", {c.tv_ch}); skey(v, p + ".0.ln.bias", {c.tv_ch});
        skey(v, p + ".1.conv.weight", {c.tv_ch, c.tv_ch, 3});
    }
    skey(v, "tv_encoder.out_conv.conv.weight", {c.tv_cout, c.tv_ch, 3});
    skey(v, "tv_encoder.vq.embedding", {c.tv_n_emb, c.tv_cout});
    proj_keys(v, "tv_encoder.proj_0", c.tv_cout, c.tv_cout_g);
    skey(v, "tv_encoder.proj_1.conv.weight", {c.tv_cout_g, c.tv_cout_g, 3}); skey(v, "tv_encoder.proj_1.conv.bias", {c.tv_cout_g});
    skey(v, "lf0_encoder.in_conv.conv.weight", {c.lf0_ch, 1, 3});
    skey(v, "lf0_encoder.in_conv.ln.weight", {c.lf0_ch}); skey(v, "lf0_encoder.in_conv.ln.bias", {c.lf0_ch});
    const int H = c.lf0_ch / 2;
    for (int l = 0; l < c.lf0_layers; ++l)
        for (const char* sfx : {"", "_reverse"}) {
            const std::string t = "_l" + std::to_string(l) + sfx;
            skey(v, "lf0_encoder.rnn_layer.weight_ih" + t, {3 * H, c.lf0_ch}); skey(v, "lf0_encoder.rnn_layer.weight_hh" + t, {3 * H, H});
            skey(v, "lf0_encoder.rnn_layer.bias_ih" + t, {3 * H}); skey(v, "lf0_encoder.rnn_layer.bias_hh" + t, {3 * H});
        }
    skey(v, "lf0_encoder.out_conv.conv.weight", {c.lf0_cout, c.lf0_ch, 3});
    skey(v, "lf0_encoder.out_conv.ln.weight", {c.lf0_cout}); skey(v, "lf0_encoder.out_conv.ln.bias", {c.lf0_cout});
    proj_keys(v, "lf0_encoder.proj", c.lf0_cout, c.lf0_cout_g);
    skey(v, "conv_sty.weight", {c.sty_out, c.tv_cout_g, 1}); skey(v, "conv_sty.bias", {c.sty_out});
    return DEX_OK;
}

void dex_style_destroy(DexStyle* v) {
    if (!v) return;
    for (auto& kv : v->raw) if (kv.second.p) hipFree(kv.second.p);
    for (void* p : v->owned) hipFree(p);
    delete v;
}
const char* dex_style_last_error(const DexStyle* v) { return v ? v->err.c_str() : "null style context"; }
int dex_style_num_weights(const DexStyle* v) { return v ? (int)v->keys.size() : 0; }
int dex_style_weight_info(const DexStyle* v, int i, const char** key, int64_t shape[4], int* ndim) {
    if (!v || i < 0 || i >= (int)v->keys.size()) return DEX_ERR_ARG;
    const SRaw& r = v->raw.at(v->keys[i]);
    if (key) *key = v->keys[i].c_str();
    if (ndim) *ndim = (int)r.shape.size();
    if (shape) for (size_t k = 0; k < r.shape.size(); ++k) shape[k] = r.shape[k];
    return DEX_OK;
}
int dex_style_load_weight_async(DexStyle* v, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream) {
    if (!v || !key || !w_dev) return DEX_ERR_ARG;
    auto it = v->raw.find(key);
    if (it == v->raw.end()) return v->fail(DEX_ERR_ARG, "unknown style weight key '%s'", key);
    SRaw& r = it->second;
    if ((int)r.shape.size() != ndim) return v->fail(DEX_ERR_ARG, "weight '%s': expected %d dims, got %d", key, (int)r.shape.size(), ndim);
    for (int k = 0; k < ndim; ++k)
        if (r.shape[k] != shape[k]) return v->fail(DEX_ERR_ARG, "weight '%s': dim %d is %lld, expected %lld", key, k, (long long)shape[k], (long long)r.shape[k]);
    if (!r.p) SCHK(v, hipMalloc((void**)&r.p, r.numel * sizeof(float)));
    SCHK(v, hipMemcpyAsync(r.p, w_dev, r.numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    r.loaded = true;
    v->finalized = false;
    return DEX_OK;
}

int dex_style_finalize(DexStyle* v, dex_stream_t stream) {
    if (!v) return DEX_ERR_ARG;
    for (const auto& k : v->keys)
        if (!v->raw.at(k).loaded) return v->fail(DEX_ERR_STATE, "style weight '%s' was never loaded", k.c_str());
    for (void* p : v->owned) hipFree(p);
    v->owned.clear();
    hipStream_t st = (hipStream_t)stream;
    const DexStyleConfig& c = v->cfg;
    int rc = DEX_OK;
    auto alloc = [&](long n) -> float* {
        float* p = nullptr;
        if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { rc = v->fail(DEX_ERR_HIP, "hipMalloc of %ld floats failed", n); return nullptr; }
        v->owned.push_back(p);
        return p;
    };
    // Conv1d [cout][cin][k] -> [(tap*cin_pad + ci)][cout], zero rows for padded input channels
    auto conv = [&](const std::string& wkey, const char* bkey_or_null, int cin, int cout, int k, int cin_pad) {
        SConv o{}; o.cin = cin_pad; o.cout = cout; o.k = k;
        const float* src = v->R(wkey);
        float* t = alloc((long)k * cin * cout);
        if (t) launch_permute4(src, t, cout, cin, k, 1, 2, 1, 0, 3, st);
        if (cin_pad == cin) o.w = t;
        else {
            float* d = alloc((long)k * cin_pad * cout);
            if (t && d) {
                hipMemsetAsync(d, 0, (size_t)k * cin_pad * cout * sizeof(float), st);
                hipMemcpy2DAsync(d, (size_t)cin_pad * cout * 4, t, (size_t)cin * cout * 4, (size_t)cin * cout * 4, k, hipMemcpyDeviceToDevice, st);
            }
            o.w = d;
        }
        o.b = bkey_or_null ? v->R(bkey_or_null) : nullptr;
        return o;
    };
    auto proj = [&](const std::string& p, int cin, int ch) {
        SProj o{};
        const std::string b1 = p + ".conv_1.bias", b2 = p + ".conv_2.bias", b3 = p + ".proj.bias";
        o.c1 = conv(p + ".conv_1.weight", b1.c_str(), cin, ch, 3, cin);
        o.c2 = conv(p + ".conv_2.weight", b2.c_str(), ch, ch, 3, ch);
        o.pr = conv(p + ".proj.weight", b3.c_str(), ch, ch, 1, ch);
        o.g1 = v->R(p + ".norm_1.gamma"); o.b1 = v->R(p + ".norm_1.beta"); o.g2 = v->R(p + ".norm_2.gamma"); o.b2 = v->R(p + ".norm_2.beta");
        return o;
    };
    v->tiv_in = conv("tiv_encoder.in_conv.conv.weight", "tiv_encoder.in_conv.conv.bias", c.n_mels, c.tiv_ch, 3, MEL_LD);
    v->tiv_b0.clear(); v->tiv_b1.clear(); v->tv_b0.clear(); v->tv_b1.clear();
    for (int i = 0; i < c.tiv_layers; ++i) {
        const std::string p = "tiv_encoder.conv_blocks." + std::to_string(i) + ".conv_block";
        const std::string bk = p + ".0.conv.bias";
        v->tiv_b0.push_back(conv(p + ".0.conv.weight", bk.c_str(), c.tiv_ch, c.tiv_ch, 3, c.tiv_ch));
        v->tiv_b1.push_back(conv(p + ".1.conv.weight", nullptr, c.tiv_ch, c.tiv_ch, 3, c.tiv_ch));
    }
    v->tv_in = conv("tv_encoder.in_conv.conv.weight", nullptr, c.n_mels, c.tv_ch, 3, MEL_LD);
    for (int i = 0; i < c.tv_layers; ++i) {
        const std::string p = "tv_encoder.conv_blocks." + std::to_string(i) + ".conv_block";
        v->tv_b0.push_back(conv(p + ".0.conv.weight", nullptr, c.tv_ch, c.tv_ch, 3, c.tv_ch));
        v->tv_b1.push_back(conv(p + ".1.conv.weight", nullptr, c.tv_ch, c.tv_ch, 3, c.tv_ch));
    }
    v->tv_out = conv("tv_encoder.out_conv.conv.weight", nullptr, c.tv_ch, c.tv_cout, 3, c.tv_ch);
    v->tv_p0 = proj("tv_encoder.proj_0", c.tv_cout, c.tv_cout_g);
    v->tv_p1 = conv("tv_encoder.proj_1.conv.weight", "tv_encoder.proj_1.conv.bias", c.tv_cout_g, c.tv_cout_g, 3, c.tv_cout_g);
    {   // codebook [M][D] -> GEMM operand [D][M]; |e|^2
        float* et = alloc((long)c.tv_n_emb * c.tv_cout);
        float* e2 = alloc(c.tv_n_emb);
        if (et) launch_permute4(v->R("tv_encoder.vq.embedding"), et, c.tv_n_emb, c.tv_cout, 1, 1, 1, 0, 2, 3, st);
        if (e2) launch_row_sumsq(v->R("tv_encoder.vq.embedding"), e2, c.tv_n_emb, c.tv_cout, st);
        v->embT = et; v->e2 = e2;
    }
    v->lf_in = conv("lf0_encoder.in_conv.conv.weight", nullptr, 1, c.lf0_ch, 3, LF0_LD);
    v->lf_out = conv("lf0_encoder.out_conv.conv.weight", nullptr, c.lf0_ch, c.lf0_cout, 3, c.lf0_ch);
    v->lf_proj = proj("lf0_encoder.proj", c.lf0_cout, c.lf0_cout_g);
    const int H = c.lf0_ch / 2;
    v->gru_wih.clear(); v->gru_bih.clear(); v->gru_whh.clear(); v->gru_bhh.clear();
    for (int l = 0; l < c.lf0_layers; ++l) {
        // input projection of both directions as ONE GEMM: [in][fwd 3H | rev 3H]; recurrent weights [2][3H][H]
        float* wih = alloc((long)c.lf0_ch * 6 * H); float* bih = alloc(6 * H); float* whh = alloc(2L * 3 * H * H); float* bhh = alloc(6 * H);
        for (int d = 0; d < 2; ++d) {
            const std::string t = "_l" + std::to_string(l) + (d ? "_reverse" : "");
            if (wih) {      // [3H][in] -> columns d*3H.. of [in][6H]
                float* tmp = alloc((long)c.lf0_ch * 3 * H);
                if (tmp) {
                    launch_permute4(v->R("lf0_encoder.rnn_layer.weight_ih" + t), tmp, 3 * H, c.lf0_ch, 1, 1, 1, 0, 2, 3, st);
                    hipMemcpy2DAsync(wih + d * 3 * H, (size_t)6 * H * 4, tmp, (size_t)3 * H * 4, (size_t)3 * H * 4, c.lf0_ch, hipMemcpyDeviceToDevice, st);
                }
            }
            if (bih) hipMemcpyAsync(bih + d * 3 * H, v->R("lf0_encoder.rnn_layer.bias_ih" + t), 3 * H * 4, hipMemcpyDeviceToDevice, st);
            if (whh) hipMemcpyAsync(whh + (long)d * 3 * H * H, v->R("lf0_encoder.rnn_layer.weight_hh" + t), (size_t)3 * H * H * 4, hipMemcpyDeviceToDevice, st);
            if (bhh) hipMemcpyAsync(bhh + d * 3 * H, v->R("lf0_encoder.rnn_layer.bias_hh" + t), 3 * H * 4, hipMemcpyDeviceToDevice, st);
        }
        v->gru_wih.push_back(wih); v->gru_bih.push_back(bih); v->gru_whh.push_back(whh); v->gru_bhh.push_back(bhh);
    }
    v->sty = conv("conv_sty.weight", "conv_sty.bias", c.tv_cout_g, c.sty_out, 1, c.tv_cout_g);
    if (rc != DEX_OK) return rc;
    SCHK(v, hipStreamSynchronize(st));
    SCHK(v, hipGetLastError());
    v->finalized = true;
    return DEX_OK;
}

}  // extern "C"

namespace {
struct SPlan { float *mr, *ms, *ml, *mel, *lf, *x, *a, *y, *z, *dots, *gi, *mean_a, *mean_b, *mean_c; size_t bytes; };
void style_plan(const DexStyle* v, int B, int Tr, int Ts, int Tl, void* ws, SPlan& P) {
    const DexStyleConfig& c = v->cfg;
    const int Tm = std::max(Tr, std::max(Ts, Tl));
    const int Cm = 256;
    char* base = (char*)ws; size_t off = 0;
    auto take = [&](size_t n) { off = (off + 255) & ~size_t(255); float* p = ws ? (float*)(base + off) : nullptr; off += n * sizeof(float); return p; };
    P.mr = take((size_t)B * Tr); P.ms = take((size_t)B * Ts); P.ml = take((size_t)B * Tl);
    P.mel = take((size_t)B * Tm * MEL_LD); P.lf = take((size_t)B * Tl * LF0_LD);
    P.x = take((size_t)B * Tm * Cm); P.a = take((size_t)B * Tm * Cm); P.y = take((size_t)B * Tm * Cm); P.z = take((size_t)B * Tm * Cm);
    P.dots = take((size_t)B * Ts * c.tv_n_emb);
    P.gi = take((size_t)B * Tl * 3 * c.lf0_ch);
    P.mean_a = take((size_t)B * Cm); P.mean_b = take((size_t)B * Cm); P.mean_c = take((size_t)B * Cm);
    P.bytes = (off + 255) & ~size_t(255);
}
// Conv1d(k, padding k/2) on [B][T][cin] -> [B][T][cout]; act 0 / 2 (ReLU); res added before the mask
void conv1d(const float* X, int T, int B, const SConv& c, const float* inmask, int act, const float* res, const float* outmask, float* out, hipStream_t st) {
    IGemmP g{};
    g.A = X; g.lda = c.cin; g.a_bstride = (long)T * c.cin;
    g.Hi = 1; g.Wi = T; g.Cin = c.cin;
    g.KH = 1; g.KW = c.k; g.sh = 1; g.sw = 1; g.off_w = -(c.k - 1) / 2; g.step_h = 1; g.step_w = 1;
    g.Ho = 1; g.Wo = T;
    g.W = c.w; g.N = c.cout; g.K = c.k * c.cin; g.ksplit = 1; g.groups = 1; g.bias = c.b;
    g.C = out; g.ldc = c.cout; g.c_bstride = (long)T * c.cout;
    g.OHf = 1; g.OWf = T; g.osh = 1; g.osw = 1;
    g.inmask = inmask; g.inmask_ws = 1; g.outmask = outmask; g.outmask_ws = 1; g.mask_bstride = T; g.gate_nstride = 1;
    g.act = act;
    g.res = res; g.ldres = c.cout; g.res_bstride = (long)T * c.cout;
    g.B = B;
    launch_igemm(g, PREC_FP32, st);
}
void layer_norm(const float* X, float* Y, long rows, int C, const float* g, const float* b, float eps, const float* mask, int T, hipStream_t st) {
    LnClP l{X, Y, rows, C, g, b, eps, mask, T};
    launch_ln_cl(l, st);
}
// Projection.forward (ref_encoder.py:24-34): in X (any), out -> `out`; a, y scratch
void projection(const SProj& p, const float* X, int T, int B, const float* mask, float* a, float* y, float* out, hipStream_t st) {
    const long rows = (long)B * T;
    conv1d(X, T, B, p.c1, mask, 2, nullptr, nullptr, a, st);
    layer_norm(a, y, rows, p.c1.cout, p.g1, p.b1, 1e-4f, nullptr, T, st);
    conv1d(y, T, B, p.c2, mask, 2, nullptr, nullptr, a, st);
    layer_norm(a, y, rows, p.c2.cout, p.g2, p.b2, 1e-4f, nullptr, T, st);
    conv1d(y, T, B, p.pr, mask, 0, nullptr, mask, out, st);
}
}  // namespace

extern "C" {

size_t dex_style_workspace_bytes(const DexStyle* v, int B, int Tr, int Ts, int Tl) {
    if (!v || B < 1 || Tr < 2 || Ts < 1 || Tl < 1) return 0;
    SPlan P; style_plan(v, B, Tr, Ts, Tl, nullptr, P);
    return P.bytes;
}

int dex_style_encode(DexStyle* v, const DexStyleArgs* a, dex_stream_t stream) {
    if (!v || !a) return DEX_ERR_ARG;
    if (!v->finalized) return v->fail(DEX_ERR_STATE, "dex_style_finalize has not been called");
    const DexStyleConfig& c = v->cfg;
    if (a->B < 1 || a->Tr < 2 || a->Ts < 1 || a->Tl < 1) return v->fail(DEX_ERR_ARG, "B >= 1, Tr >= 2 (InstanceNorm1D uses the unbiased variance), Ts >= 1, Tl >= 1");
    if (!a->ref_mel_dev || !a->ref_lengths_dev || !a->sty_mel_dev || !a->sty_lengths_dev || !a->lf0_dev || !a->lf0_lengths_dev ||
        !a->ref_skips_out_dev || !a->sty_dec_out_dev || !a->sty_enc_out_dev || !a->workspace_dev)
        return v->fail(DEX_ERR_ARG, "null pointer in DexStyleArgs");
    for (int i = 0; i < c.tiv_layers; ++i) if (!a->ref_skips_out_dev[i]) return v->fail(DEX_ERR_ARG, "ref_skips_out_dev[%d] is null", i);
    if (((uintptr_t)a->workspace_dev & 255) != 0) return v->fail(DEX_ERR_ARG, "workspace must be 256-byte aligned");
    SPlan P; style_plan(v, a->B, a->Tr, a->Ts, a->Tl, nullptr, P);
    if (P.bytes > a->workspace_bytes) return v->fail(DEX_ERR_WORKSPACE, "style workspace too small: need %zu bytes, got %zu", P.bytes, a->workspace_bytes);
    style_plan(v, a->B, a->Tr, a->Ts, a->Tl, a->workspace_dev, P);
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, Tr = a->Tr, Ts = a->Ts, Tl = a->Tl;
    launch_len_mask(a->ref_lengths_dev, P.mr, B, Tr, st);
    launch_len_mask(a->sty_lengths_dev, P.ms, B, Ts, st);
    launch_len_mask(a->lf0_lengths_dev, P.ml, B, Tl, st);

    // ---- LF0Encoder (ref_encoder.py:45-55)
    launch_lf0_to_cl(a->lf0_dev, P.ml, P.lf, B, Tl, LF0_LD, st);
    conv1d(P.lf, Tl, B, v->lf_in, nullptr, 2, nullptr, nullptr, P.a, st);                       // conv -> relu
    layer_norm(P.a, P.x, (long)B * Tl, c.lf0_ch, v->R("lf0_encoder.in_conv.ln.weight"), v->R("lf0_encoder.in_conv.ln.bias"), 1e-5f, P.ml, Tl, st);
    const int H = c.lf0_ch / 2;
    float* cur = P.x; float* nxt = P.y;
    for (int l = 0; l < c.lf0_layers; ++l) {
        IGemmP g{};         // gi = x W_ih^T + b_ih for both directions: [B*Tl][6H]
        g.A = cur; g.lda = c.lf0_ch; g.a_bstride = (long)Tl * c.lf0_ch; g.Hi = 1; g.Wi = Tl; g.Cin = c.lf0_ch;
        g.KH = 1; g.KW = 1; g.sh = 1; g.sw = 1; g.step_h = 1; g.step_w = 1; g.Ho = 1; g.Wo = Tl;
        g.W = v->gru_wih[l]; g.N = 6 * H; g.K = c.lf0_ch; g.ksplit = 1; g.groups = 1; g.bias = v->gru_bih[l];
        g.C = P.gi; g.ldc = 6 * H; g.c_bstride = (long)Tl * 6 * H; g.OHf = 1; g.OWf = Tl; g.osh = 1; g.osw = 1;
        g.inmask_ws = 1; g.outmask_ws = 1; g.gate_nstride = 1; g.B = B;
        launch_igemm(g, PREC_FP32, st);
        GruP gp{P.gi, v->gru_whh[l], v->gru_bhh[l], nxt, B, Tl, H};
        launch_gru_layer(gp, st);
        std::swap(cur, nxt);
    }
    conv1d(cur, Tl, B, v->lf_out, P.ml, 2, nullptr, nullptr, P.a, st);
    layer_norm(P.a, P.z, (long)B * Tl, c.lf0_cout, v->R("lf0_encoder.out_conv.ln.weight"), v->R("lf0_encoder.out_conv.ln.bias"), 1e-5f, P.ml, Tl, st);   // lf0_enc
    launch_masked_mean_cl(P.z, P.ml, P.mean_a, B, Tl, c.lf0_cout, st);                          // mean lf0_enc
    projection(v->lf_proj, P.z, Tl, B, P.ml, P.a, P.y, P.x, st);                                // lf0_dec -> P.x
    launch_masked_mean_cl(P.x, P.ml, P.mean_b, B, Tl, c.lf0_cout_g, st);                        // mean lf0_dec

    // ---- TVEncoder (ref_encoder.py:122-140)
    launch_mel_to_cl(a->sty_mel_dev, P.mel, B, c.n_mels, Ts, MEL_LD, st);
    conv1d(P.mel, Ts, B, v->tv_in, P.ms, 2, nullptr, nullptr, P.a, st);
    layer_norm(P.a, P.x, (long)B * Ts, c.tv_ch, v->R("tv_encoder.in_conv.ln.weight"), v->R("tv_encoder.in_conv.ln.bias"), 1e-5f, P.ms, Ts, st);
    cur = P.x; nxt = P.z;
    for (int i = 0; i < c.tv_layers; ++i) {
        const std::string p = "tv_encoder.conv_blocks." + std::to_string(i) + ".conv_block.0.ln";
        conv1d(cur, Ts, B, v->tv_b0[i], nullptr, 2, nullptr, nullptr, P.a, st);                 // cur is already masked
        layer_norm(P.a, P.y, (long)B * Ts, c.tv_ch, v->R(p + ".weight"), v->R(p + ".bias"), 1e-5f, nullptr, Ts, st);
        conv1d(P.y, Ts, B, v->tv_b1[i], nullptr, 0, cur, P.ms, nxt, st);                        // (x + conv_block(x)) * mask
        std::swap(cur, nxt);
    }
    conv1d(cur, Ts, B, v->tv_out, nullptr, 0, nullptr, P.ms, P.a, st);                          // z_beforeVQ -> P.a
    launch_masked_mean_cl(P.a, P.ms, P.mean_c, B, Ts, c.tv_cout, st);                           // mean sty_enc
    {   // VQ lookup: dots = z e^T, nearest code
        IGemmP g{};
        g.A = P.a; g.lda = c.tv_cout; g.a_bstride = (long)Ts * c.tv_cout; g.Hi = 1; g.Wi = Ts; g.Cin = c.tv_cout;
        g.KH = 1; g.KW = 1; g.sh = 1; g.sw = 1; g.step_h = 1; g.step_w = 1; g.Ho = 1; g.Wo = Ts;
        g.W = v->embT; g.N = c.tv_n_emb; g.K = c.tv_cout; g.ksplit = 1; g.groups = 1;
        g.C = P.dots; g.ldc = c.tv_n_emb; g.c_bstride = (long)Ts * c.tv_n_emb; g.OHf = 1; g.OWf = Ts; g.osh = 1; g.osw = 1;
        g.inmask_ws = 1; g.outmask_ws = 1; g.gate_nstride = 1; g.B = B;
        launch_igemm(g, PREC_FP32, st);
        VqP q{P.a, P.dots, v->R("tv_encoder.vq.embedding"), v->e2, P.ms, P.z, a->vq_idx_out_dev, (long)B * Ts, c.tv_n_emb, c.tv_cout};
        launch_vq_lookup(q, st);
    }
    // sty_enc = mean(z_beforeVQ) + mean(lf0_enc)   (tts.py:62)
    launch_add_bcast_cl(P.mean_c, P.mean_a, B, 1, c.tv_cout, st);
    SCHK(v, hipMemcpyAsync(a->sty_enc_out_dev, P.mean_c, (size_t)B * c.tv_cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    projection(v->tv_p0, P.z, Ts, B, P.ms, P.a, P.y, P.x, st);                                  // proj_0 -> P.x
    conv1d(P.x, Ts, B, v->tv_p1, P.ms, 2, nullptr, P.ms, P.z, st);                              // proj_1 (BN folded) -> relu -> * mask
    launch_add_bcast_cl(P.z, P.mean_b, B, Ts, c.tv_cout_g, st);                                 // + mean lf0_dec (tts.py:65)
    conv1d(P.z, Ts, B, v->sty, nullptr, 0, nullptr, nullptr, P.a, st);                          // conv_sty
    launch_cl_to_cf(P.a, a->sty_dec_out_dev, B, Ts, c.sty_out, st);

    // ---- TIVEncoder (ref_encoder.py:96-108); its out_conv result is not used downstream (tts.py:67: only the skips are)
    launch_mel_to_cl(a->ref_mel_dev, P.mel, B, c.n_mels, Tr, MEL_LD, st);
    conv1d(P.mel, Tr, B, v->tiv_in, P.mr, 2, nullptr, P.mr, P.x, st);
    cur = P.x; nxt = P.z;
    for (int i = 0; i < c.tiv_layers; ++i) {
        conv1d(cur, Tr, B, v->tiv_b0[i], nullptr, 2, nullptr, nullptr, P.a, st);                // cur = x * mask already
        conv1d(P.a, Tr, B, v->tiv_b1[i], nullptr, 0, cur, P.mr, P.y, st);                       // skip_i = (x + conv_block(x)) * mask
        launch_cl_to_cf(P.y, a->ref_skips_out_dev[i], B, Tr, c.tiv_ch, st);
        launch_inorm_cl(P.y, nxt, P.mr, B, Tr, c.tiv_ch, 1e-5f, st);                            // InstanceNorm1D, then the next block's * mask
        std::swap(cur, nxt);
    }
    SCHK(v, hipGetLastError());
    return DEX_OK;
}

}  // extern "C"
