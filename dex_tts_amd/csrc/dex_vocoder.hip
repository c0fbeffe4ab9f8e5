// dex_vocoder.hip — C ABI of the HiFi-GAN generator (include/dex_amd.h, dex_voc_*): GeDEX-TTS/hifigan/models.py:112-173.
//
//   x = conv_pre(mel)                                            Conv1d(80 -> C0, 7, pad 3)
//   per stage i:  x = ups[i](leaky_relu(x, 0.1))                 ConvTranspose1d(C -> C/2, k_i, stride u_i, pad (k_i-u_i)/2)
//                 x = (rb_0(x) + rb_1(x) + rb_2(x)) / 3          ResBlock(k_j, dilations d_j): 3 x [lrelu, conv(k,d), lrelu, conv(k,1), + x]
//   wav = tanh(conv_post(leaky_relu(x, 0.01)))                   Conv1d(C_last -> 1, 7, pad 3)
//
// Every Conv1d is an implicit GEMM over channels-last activations [B][L][C] (H = 1, KW = k, step_w = dilation) on the
// exact-fp32 MFMA kernel (igemm.hip), with the leaky_relu applied while the A tile is gathered and the residual added in
// the epilogue.  A ConvTranspose1d is one GEMM Y[l][j*Cout + co] = lrelu(x)[l][:] . w[:, co, j] followed by an overlap-add
// (k/u terms per output).  conv_post (one output channel) and the tanh are one element-wise kernel.
//
// BigVGAN (DEX-TTS/bigvgan/models.py:138-211; DexVocoderConfig::activation != 0) is the same network with AMPBlock1: the leaky_relu
// in front of every ResBlock conv and of conv_post becomes the anti-aliased Snake / SnakeBeta activation (alias_free_torch/act.py;
// one aa_snake launch into a scratch tensor), the transposed convs take x as it is.
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
struct VRaw { float* p = nullptr; std::vector<int64_t> shape; long numel = 0; bool loaded = false; };
// (wlp: the same matrix as bf16 [0] / fp16 [1], [N][K] with K contiguous - the reduced-precision GEMM's weight operand)
struct VConv { const float* w = nullptr; const float* b = nullptr; int cin, cout, k, dil; const void* wlp[2] = {nullptr, nullptr}; };     // packed [k*cin][cout]
struct VUp { const float* w = nullptr; const float* b = nullptr; int cin, cout, k, u, pad; const void* wlp[2] = {nullptr, nullptr}; };    // packed [cin][k*cout]
constexpr int MEL_LD = 96;          // num_mels padded to a multiple of 32 (K tiles of the implicit GEMM do not straddle taps)
}  // namespace

struct DexVoc {
    DexVocoderConfig cfg{};
    std::string err;
    std::vector<std::string> keys;
    std::map<std::string, VRaw> raw;
    std::vector<void*> owned;
    bool finalized = false;
    int precision = DEX_PREC_FP32;      // DEX_PREC_BF16 / DEX_PREC_FP16: the convolutions' operands (fp32 accumulation, fp32 activations in HBM)
    VConv pre;
    std::vector<VUp> ups;
    std::vector<VConv> rb;              // [stage][j][c1_0, c2_0, c1_1, c2_1, c1_2, c2_2] flattened
    const float *post_w = nullptr, *post_b = nullptr;
    // BigVGAN: per activation layer (a, 1/b) coefficient pairs: [stage][block j][layer l] then the post activation; the shared filter
    std::vector<const float*> act_a, act_ib;
    const float* filt = nullptr;
    bool big() const { return cfg.activation != 0; }
    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};

#define VCHK(v, call)                                                                                  \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                 \
        return (v)->fail(DEX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

namespace {
void vkey(DexVoc* v, const std::string& k, std::vector<int64_t> shape) {
    v->keys.push_back(k);
    VRaw r; r.shape = std::move(shape); r.numel = 1;
    for (auto d : r.shape) r.numel *= d;
    v->raw[k] = r;
}
int stage_ch(const DexVocoderConfig& c, int i) { return c.upsample_initial_channel >> (i + 1); }
long total_up(const DexVocoderConfig& c) { long u = 1; for (int i = 0; i < c.n_upsamples; ++i) u *= c.upsample_rates[i]; return u; }
}  // namespace

extern "C" {

int dex_voc_create(const DexVocoderConfig* cfg, DexVoc** out) {
    if (!cfg || !out) return DEX_ERR_ARG;
    DexVoc* v = new DexVoc();
    v->cfg = *cfg;
    *out = v;
    const DexVocoderConfig& c = v->cfg;
    if (c.num_mels < 1 || c.num_mels > MEL_LD) return v->fail(DEX_ERR_ARG, "num_mels must be in [1, %d]", MEL_LD);
    if (c.n_upsamples < 1 || c.n_upsamples > 6) return v->fail(DEX_ERR_ARG, "n_upsamples must be in [1, 6]");
    if (c.n_resblock_kernels != 3) return v->fail(DEX_ERR_ARG, "three ResBlock kernel sizes are expected (hifigan/config.json)");
    if (c.upsample_initial_channel % 64) return v->fail(DEX_ERR_ARG, "upsample_initial_channel must be a multiple of 64");
    for (int i = 0; i < c.n_upsamples; ++i) {
        const int k = c.upsample_kernel_sizes[i], u = c.upsample_rates[i], co = stage_ch(c, i);
        if (u < 1 || k < u || (k - u) % 2) return v->fail(DEX_ERR_ARG, "upsample %d: kernel %d / rate %d unsupported (needs k >= u, k - u even)", i, k, u);
        if (co % 32 || co < 32) return v->fail(DEX_ERR_ARG, "stage %d has %d channels: the implicit GEMM needs multiples of 32", i, co);
        if ((k * co) % 64) return v->fail(DEX_ERR_ARG, "stage %d: k * channels must be a multiple of 64", i);
    }
    if (stage_ch(c, c.n_upsamples - 1) > 64) return v->fail(DEX_ERR_ARG, "conv_post kernel handles <= 64 input channels");
    for (int j = 0; j < 3; ++j) if (c.resblock_kernel_sizes[j] % 2 == 0) return v->fail(DEX_ERR_ARG, "ResBlock kernel sizes must be odd");
    if (c.activation < 0 || c.activation > 2) return v->fail(DEX_ERR_ARG, "activation must be 0 (HiFi-GAN), 1 (BigVGAN snake) or 2 (BigVGAN snakebeta)");
    const bool big = c.activation != 0, beta = c.activation == 2;
    const std::string upsfx = big ? ".0" : "";                      // BigVGAN nests each transposed conv in a ModuleList
    const int c0 = c.upsample_initial_channel;
    vkey(v, "conv_pre.weight", {c0, c.num_mels, 7}); vkey(v, "conv_pre.bias", {c0});
    for (int i = 0; i < c.n_upsamples; ++i) {
        const int ci = c0 >> i, co = c0 >> (i + 1);
        vkey(v, "ups." + std::to_string(i) + upsfx + ".weight", {ci, co, c.upsample_kernel_sizes[i]});      // ConvTranspose1d: [in, out, k]
        vkey(v, "ups." + std::to_string(i) + upsfx + ".bias", {co});
    }
    for (int i = 0; i < c.n_upsamples; ++i)
        for (int j = 0; j < 3; ++j) {
            const int ch = stage_ch(c, i), k = c.resblock_kernel_sizes[j];
            const std::string p = "resblocks." + std::to_string(i * 3 + j);
            for (const char* cs : {".convs1.", ".convs2."})
                for (int m = 0; m < 3; ++m) {
                    vkey(v, p + cs + std::to_string(m) + ".weight", {ch, ch, k});
                    vkey(v, p + cs + std::to_string(m) + ".bias", {ch});
                }
            if (big)
                for (int l = 0; l < 6; ++l) {
                    vkey(v, p + ".activations." + std::to_string(l) + ".act.alpha", {ch});
                    if (beta) vkey(v, p + ".activations." + std::to_string(l) + ".act.beta", {ch});
                }
        }
    if (big) {
        const int cl = stage_ch(c, c.n_upsamples - 1);
        vkey(v, "activation_post.act.alpha", {cl});
        if (beta) vkey(v, "activation_post.act.beta", {cl});
        vkey(v, "activation_post.upsample.filter", {1, 1, 12}); vkey(v, "activation_post.downsample.lowpass.filter", {1, 1, 12});
    }
    vkey(v, "conv_post.weight", {1, stage_ch(c, c.n_upsamples - 1), 7}); vkey(v, "conv_post.bias", {1});
    return DEX_OK;
}

void dex_voc_destroy(DexVoc* v) {
    if (!v) return;
    for (auto& kv : v->raw) if (kv.second.p) hipFree(kv.second.p);
    for (void* p : v->owned) hipFree(p);
    delete v;
}
const char* dex_voc_last_error(const DexVoc* v) { return v ? v->err.c_str() : "null vocoder context"; }
int dex_voc_num_weights(const DexVoc* v) { return v ? (int)v->keys.size() : 0; }
int dex_voc_weight_info(const DexVoc* v, int i, const char** key, int64_t shape[4], int* ndim) {
    if (!v || i < 0 || i >= (int)v->keys.size()) return DEX_ERR_ARG;
    const VRaw& r = v->raw.at(v->keys[i]);
    if (key) *key = v->keys[i].c_str();
    if (ndim) *ndim = (int)r.shape.size();
    if (shape) for (size_t k = 0; k < r.shape.size(); ++k) shape[k] = r.shape[k];
    return DEX_OK;
}
int dex_voc_load_weight_async(DexVoc* v, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream) {
    if (!v || !key || !w_dev) return DEX_ERR_ARG;
    auto it = v->raw.find(key);
    if (it == v->raw.end()) return v->fail(DEX_ERR_ARG, "unknown vocoder weight key '%s'", key);
    VRaw& r = it->second;
    if ((int)r.shape.size() != ndim) return v->fail(DEX_ERR_ARG, "weight '%s': expected %d dims, got %d", key, (int)r.shape.size(), ndim);
    for (int k = 0; k < ndim; ++k)
        if (r.shape[k] != shape[k]) return v->fail(DEX_ERR_ARG, "weight '%s': dim %d is %lld, expected %lld", key, k, (long long)shape[k], (long long)r.shape[k]);
    if (!r.p) VCHK(v, hipMalloc((void**)&r.p, r.numel * sizeof(float)));
    VCHK(v, hipMemcpyAsync(r.p, w_dev, r.numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    r.loaded = true;
    v->finalized = false;
    return DEX_OK;
}

int dex_voc_finalize(DexVoc* v, dex_stream_t stream) {
    if (!v) return DEX_ERR_ARG;
    for (const auto& k : v->keys)
        if (!v->raw.at(k).loaded) return v->fail(DEX_ERR_STATE, "vocoder weight '%s' was never loaded", k.c_str());
    for (void* p : v->owned) hipFree(p);
    v->owned.clear();
    hipStream_t st = (hipStream_t)stream;
    const DexVocoderConfig& c = v->cfg;
    int rc = DEX_OK;
    auto alloc = [&](long n) -> float* {
        float* p = nullptr;
        if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { rc = v->fail(DEX_ERR_HIP, "hipMalloc of %ld floats failed", n); return nullptr; }
        v->owned.push_back(p);
        return p;
    };
    // reduced-precision copies of a packed fp32 [K][N] matrix: bf16 and fp16, [N][K]
    auto lp_copies = [&](const float* w, int K, int N, const void* (&out)[2]) {
        for (int t = 0; t < 2; ++t) {
            void* d = alloc(((long)K * N + 1) / 2);
            if (d) launch_pack_lp_nk(w, d, K, N, t ? DEX_PREC_FP16 : DEX_PREC_BF16, st);
            out[t] = d;
        }
    };
    // Conv1d [Cout][Cin][k] -> [(tap*Cin_pad + ci)][Cout]
    auto conv = [&](const std::string& name, int cin, int cout, int k, int dil, int cin_pad) {
        VConv o{}; o.cin = cin_pad; o.cout = cout; o.k = k; o.dil = dil;
        const float* src = v->raw.at(name + ".weight").p;
        if (cin_pad == cin) {
            float* d = alloc((long)k * cin * cout);
            if (d) launch_permute4(src, d, cout, cin, k, 1, 2, 1, 0, 3, st);      // [o][i][k][1] -> [k][i][o][1]
            o.w = d;
        } else {        // zero rows for the padded input channels
            float* t = alloc((long)k * cin * cout);
            float* d = alloc((long)k * cin_pad * cout);
            if (t && d) {
                launch_permute4(src, t, cout, cin, k, 1, 2, 1, 0, 3, st);
                hipMemsetAsync(d, 0, (size_t)k * cin_pad * cout * sizeof(float), st);
                hipMemcpy2DAsync(d, (size_t)cin_pad * cout * 4, t, (size_t)cin * cout * 4, (size_t)cin * cout * 4, k, hipMemcpyDeviceToDevice, st);
            }
            o.w = d;
        }
        o.b = v->raw.at(name + ".bias").p;
        if (o.w) lp_copies(o.w, k * cin_pad, cout, o.wlp);
        return o;
    };
    v->pre = conv("conv_pre", c.num_mels, c.upsample_initial_channel, 7, 1, MEL_LD);
    v->ups.clear(); v->rb.clear();
    for (int i = 0; i < c.n_upsamples; ++i) {
        const int ci = c.upsample_initial_channel >> i, co = stage_ch(c, i), k = c.upsample_kernel_sizes[i], u = c.upsample_rates[i];
        VUp up{}; up.cin = ci; up.cout = co; up.k = k; up.u = u; up.pad = (k - u) / 2;
        float* d = alloc((long)ci * k * co);
        const std::string upn = "ups." + std::to_string(i) + (v->big() ? ".0" : "");
        if (d) launch_permute4(v->raw.at(upn + ".weight").p, d, ci, co, k, 1, 0, 2, 1, 3, st);   // [ci][co][k] -> [ci][k][co]
        up.w = d; up.b = v->raw.at(upn + ".bias").p;
        if (d) lp_copies(d, ci, k * co, up.wlp);
        v->ups.push_back(up);
        for (int j = 0; j < 3; ++j) {
            const std::string p = "resblocks." + std::to_string(i * 3 + j);
            for (int m = 0; m < 3; ++m) {
                v->rb.push_back(conv(p + ".convs1." + std::to_string(m), co, co, c.resblock_kernel_sizes[j], c.resblock_dilation_sizes[j][m], co));
                v->rb.push_back(conv(p + ".convs2." + std::to_string(m), co, co, c.resblock_kernel_sizes[j], 1, co));
            }
        }
    }
    v->act_a.clear(); v->act_ib.clear(); v->filt = nullptr;
    if (v->big()) {
        const bool beta = c.activation == 2;
        auto coeffs = [&](const std::string& p, int ch) {
            float* a = alloc(ch); float* ib = alloc(ch);
            if (a && ib) launch_snake_coeffs(v->raw.at(p + ".alpha").p, v->raw.at(p + (beta ? ".beta" : ".alpha")).p, a, ib, ch, c.snake_logscale, st);
            v->act_a.push_back(a); v->act_ib.push_back(ib);
        };
        for (int i = 0; i < c.n_upsamples; ++i)
            for (int j = 0; j < 3; ++j)
                for (int l = 0; l < 6; ++l)
                    coeffs("resblocks." + std::to_string(i * 3 + j) + ".activations." + std::to_string(l) + ".act", stage_ch(c, i));
        coeffs("activation_post.act", stage_ch(c, c.n_upsamples - 1));
        v->filt = v->raw.at("activation_post.upsample.filter").p;      // (the host checks that every resampling filter of the checkpoint equals it)
    }
    {   // conv_post [1][C][7] -> [tap][c]
        const int cl = stage_ch(c, c.n_upsamples - 1);
        float* d = alloc(7L * cl);
        if (d) launch_permute4(v->raw.at("conv_post.weight").p, d, 1, cl, 7, 1, 0, 2, 1, 3, st);
        v->post_w = d; v->post_b = v->raw.at("conv_post.bias").p;
    }
    if (rc != DEX_OK) return rc;
    VCHK(v, hipStreamSynchronize(st));
    VCHK(v, hipGetLastError());
    v->finalized = true;
    return DEX_OK;
}

int dex_voc_samples(const DexVoc* v, int T) { return v ? (int)(T * total_up(v->cfg)) : 0; }

int dex_voc_set_precision(DexVoc* v, int precision) {
    if (!v) return DEX_ERR_ARG;
    if (precision != DEX_PREC_FP32 && precision != DEX_PREC_BF16 && precision != DEX_PREC_FP16) return v->fail(DEX_ERR_ARG, "unknown precision %d", precision);
    v->precision = precision;
    return DEX_OK;
}

}  // extern "C"

namespace {
struct VPlan { float *mel, *x, *y, *a, *q, *s, *p[3]; size_t bytes; };
void voc_plan(const DexVoc* v, int B, int T, void* ws, VPlan& P) {
    const DexVocoderConfig& c = v->cfg;
    // largest activation [B][L][C] and ConvTranspose GEMM output [B][L_in][k*Cout] over the stages
    size_t act = (size_t)B * T * c.upsample_initial_channel, ymax = 0;
    long L = T;
    for (int i = 0; i < c.n_upsamples; ++i) {
        ymax = std::max(ymax, (size_t)B * L * c.upsample_kernel_sizes[i] * stage_ch(c, i));
        L *= c.upsample_rates[i];
        act = std::max(act, (size_t)B * L * stage_ch(c, i));
    }
    char* base = (char*)ws; size_t off = 0;
    auto take = [&](size_t n) { off = (off + 255) & ~size_t(255); float* p = ws ? (float*)(base + off) : nullptr; off += n * sizeof(float); return p; };
    P.mel = take((size_t)B * T * MEL_LD);
    P.x = take(act); P.a = take(act); P.q = take(act);
    P.s = v->big() ? take(act) : nullptr;                 // output of the anti-aliased activation in front of a conv
    for (int j = 0; j < 3; ++j) P.p[j] = take(act);
    P.y = take(ymax);
    P.bytes = (off + 255) & ~size_t(255);
}
thread_local int g_voc_lp = -1;        // -1: fp32 operands; 0 / 1: bf16 / fp16 weight copies (set by dex_vocode for the duration of its enqueue)
IGemmP conv1d(const float* X, int L, int B, const VConv& c, float slope, float* out, const float* res) {
    IGemmP g{};
    g.A = X; g.lda = c.cin; g.a_bstride = (long)L * c.cin; g.a_coff = 0;
    g.Hi = 1; g.Wi = L; g.Cin = c.cin;
    g.KH = 1; g.KW = c.k; g.sh = 1; g.sw = 1; g.off_h = 0; g.off_w = -c.dil * (c.k - 1) / 2; g.step_h = 1; g.step_w = c.dil;
    g.Ho = 1; g.Wo = L;
    g.W = c.w; g.Wbf = g_voc_lp >= 0 ? c.wlp[g_voc_lp] : nullptr; g.N = c.cout; g.K = c.k * c.cin; g.ksplit = 1; g.groups = 1;
    g.bias = c.b;
    g.C = out; g.ldc = c.cout; g.c_bstride = (long)L * c.cout; g.c_coff = 0;
    g.OHf = 1; g.OWf = L; g.osh = 1; g.osw = 1;
    g.inmask_ws = 1; g.outmask_ws = 1; g.gate_nstride = 1;
    g.act_in_slope = slope;
    g.res = res; g.ldres = c.cout; g.res_bstride = (long)L * c.cout;
    g.B = B;
    return g;
}
}  // namespace

extern "C" {

size_t dex_voc_workspace_bytes(const DexVoc* v, int B, int T) {
    if (!v || B < 1 || T < 1) return 0;
    VPlan P; voc_plan(v, B, T, nullptr, P);
    return P.bytes;
}

int dex_vocode(DexVoc* v, const float* mel_dev, int B, int T, float* wav_dev, void* ws, size_t ws_bytes, dex_stream_t stream) {
    if (!v || !mel_dev || !wav_dev || !ws) return DEX_ERR_ARG;
    if (!v->finalized) return v->fail(DEX_ERR_STATE, "dex_voc_finalize has not been called");
    if (B < 1 || T < 1) return v->fail(DEX_ERR_ARG, "B and T must be >= 1");
    if (((uintptr_t)ws & 255) != 0) return v->fail(DEX_ERR_ARG, "workspace must be 256-byte aligned");
    VPlan P; voc_plan(v, B, T, nullptr, P);
    if (P.bytes > ws_bytes) return v->fail(DEX_ERR_WORKSPACE, "vocoder workspace too small: need %zu bytes, got %zu", P.bytes, ws_bytes);
    voc_plan(v, B, T, ws, P);
    hipStream_t st = (hipStream_t)stream;
    const DexVocoderConfig& c = v->cfg;
    const int prec = v->precision;
    g_voc_lp = prec == DEX_PREC_BF16 ? 0 : prec == DEX_PREC_FP16 ? 1 : -1;
    launch_mel_to_cl(mel_dev, P.mel, B, c.num_mels, T, MEL_LD, st);
    launch_igemm(conv1d(P.mel, T, B, v->pre, 0.f, P.x, nullptr), prec, st);                  // conv_pre
    long L = T;
    for (int i = 0; i < c.n_upsamples; ++i) {
        const VUp& up = v->ups[i];
        {   // ConvTranspose1d(leaky_relu(x, 0.1)): GEMM + overlap-add
            IGemmP g{};
            g.A = P.x; g.lda = up.cin; g.a_bstride = L * up.cin; g.Hi = 1; g.Wi = (int)L; g.Cin = up.cin;
            g.KH = 1; g.KW = 1; g.sh = 1; g.sw = 1; g.step_h = 1; g.step_w = 1; g.Ho = 1; g.Wo = (int)L;
            g.W = up.w; g.Wbf = g_voc_lp >= 0 ? up.wlp[g_voc_lp] : nullptr; g.N = up.k * up.cout; g.K = up.cin; g.ksplit = 1; g.groups = 1;
            g.C = P.y; g.ldc = g.N; g.c_bstride = L * g.N; g.OHf = 1; g.OWf = (int)L; g.osh = 1; g.osw = 1;
            g.inmask_ws = 1; g.outmask_ws = 1; g.gate_nstride = 1; g.act_in_slope = v->big() ? 0.f : 0.1f; g.B = B;     // BigVGAN: no activation here
            launch_igemm(g, prec, st);
            ConvTFoldP f{P.y, up.b, P.a, (int)L, up.cout, up.k, up.u, up.pad, B};
            launch_convt_fold(f, st);
        }
        L *= up.u;
        // three ResBlocks on the stage input P.a; block j's result ends in P.p[j]
        for (int j = 0; j < 3; ++j) {
            const VConv* cv = &v->rb[(size_t)(i * 3 + j) * 6];
            const float* cur = P.a;
            for (int m = 0; m < 3; ++m) {
                float* dst = (m == 1) ? P.q : P.p[j];                   // x -> p[j] -> q -> p[j]
                if (v->big()) {      // AMPBlock1 (bigvgan/models.py:76-85): xt = c1(a_{2m}(x)); x = c2(a_{2m+1}(xt)) + x
                    const size_t ai = ((size_t)(i * 3 + j) * 6) + 2 * m;
                    AaSnakeP s1{cur, P.s, (int)L, up.cout, B, v->act_a[ai], v->act_ib[ai], v->filt};
                    launch_aa_snake(s1, st);
                    launch_igemm(conv1d(P.s, (int)L, B, cv[2 * m], 0.f, P.x, nullptr), prec, st);
                    AaSnakeP s2{P.x, P.s, (int)L, up.cout, B, v->act_a[ai + 1], v->act_ib[ai + 1], v->filt};
                    launch_aa_snake(s2, st);
                    launch_igemm(conv1d(P.s, (int)L, B, cv[2 * m + 1], 0.f, dst, cur), prec, st);
                } else {
                    launch_igemm(conv1d(cur, (int)L, B, cv[2 * m], 0.1f, P.x, nullptr), prec, st);          // xt = c1(lrelu(x))
                    launch_igemm(conv1d(P.x, (int)L, B, cv[2 * m + 1], 0.1f, dst, cur), prec, st);          // x = c2(lrelu(xt)) + x
                }
                cur = dst;
            }
        }
        launch_avg3(P.p[0], P.p[1], P.p[2], P.x, (long)B * L * up.cout, st);
    }
    const float* xin = P.x;
    if (v->big()) {          // activation_post (models.py:205) replaces the leaky_relu in front of conv_post
        AaSnakeP sp{P.x, P.s, (int)L, stage_ch(c, c.n_upsamples - 1), B, v->act_a.back(), v->act_ib.back(), v->filt};
        launch_aa_snake(sp, st);
        xin = P.s;
    }
    ConvPostP cp{xin, v->post_w, v->post_b, wav_dev, (int)L, stage_ch(c, c.n_upsamples - 1), B, v->big() ? 1.f : 0.01f};
    launch_conv_post_tanh(cp, st);
    VCHK(v, hipGetLastError());
    return DEX_OK;
}

}  // extern "C"
