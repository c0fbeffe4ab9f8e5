// dex_api.hip — C ABI of libdexamd.so (include/dex_amd.h): context, weight packing, workspace plan,
// and the host-side enqueue of one EDM Euler step / the whole sampler.  No torch types; raw device
// pointers + hipStream_t only.  Reference call chain replaced: GeDEX-TTS/model/diffusion.py:220-229 ->
// model/edm.py:109-216 -> :88-98 -> diffusion.py:168-207 (+ model/dit.py:485-525, DEX ref_encoder.py).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cmath>
#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <mutex>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string_view>
#include <string>
#include <vector>

#include "../../include/dex_amd.h"
#include "kernels.h"

using namespace dex;

// ---- knob snapshot (kernels.h: knob()).  Every DEX_* variable the library consults while it ENQUEUES a call is listed here; a call
// reads them all once (one getenv each), keeps the values for its whole enqueue and hashes them into its graph-cache key.
namespace dex {
namespace {
const char* const KNOB_NAMES[] = {
    "DEX_CONV_STREAM", "DEX_CONV_PP", "DEX_CONV_SKIP_DEAD", "DEX_H_BF16", "DEX_ATTN_SEPARATE", "DEX_ATTN_Q64", "DEX_ATTN_Q64_TAIL", "DEX_ATTN_Q64_HALF", "DEX_DIT_XCDS", "DEX_ATTN_GENERIC", "DEX_RES2", "DEX_LP_INTER", "DEX_DIT_CHAIN",
    "DEX_DIT_CLUSTER", "DEX_DIT_CLUSTER_LOCAL", "DEX_ATTN_X_LP", "DEX_RES_X_LP", "DEX_CAT_LP", "DEX_XCD_MAP", "DEX_DEBUG_DROP_HANDOFF", "DEX_PATCH_FUSED",
    "DEX_CONV_DOWN", "DEX_CONVT_UP", "DEX_DEBUG_PLAN",
    // launcher heuristics (workgroup caps, shape thresholds, form switches) - process-static getenv reads until round 4
    "DEX_ATTN_SHARED_W8", "DEX_CONVT_MT", "DEX_CONVT_WGS", "DEX_CONV_DOWN_WGS", "DEX_CONV_REGW", "DEX_CONV_REGW_RES", "DEX_CONV_SMALL_MAX", "DEX_CONV_TH8",
    "DEX_CONV_W8", "DEX_DWCONV_CAP", "DEX_FINAL_CAP", "DEX_FIRST_CAP", "DEX_FIRST_MFMA", "DEX_GEMM_NWALK", "DEX_NWALK_SPLIT", "DEX_NWALK_BM", "DEX_POS_COL", "DEX_POS_COL_MIN",
    "DEX_POS_CT", "DEX_REGW_MIN_TILES", "DEX_REGW_WGS", "DEX_ROWCHAIN64", "DEX_ROWCHAIN64A", "DEX_TIV_CAP", "DEX_TV_CHAIN", "DEX_TIV_FOLD", "DEX_LINATTN_NSUB", "DEX_TV_FOLD", "DEX_OUT2_MIN", "DEX_NWALK_DMA"};
constexpr int N_KNOBS = (int)(sizeof(KNOB_NAMES) / sizeof(KNOB_NAMES[0]));
// value of one variable: a decimal integer; unset, empty or without digits ("true", "on") = KNOB_UNSET, so a typo leaves the default
// form on instead of silently switching it off
int knob_parse(const char* e) {
    if (!e) return KNOB_UNSET;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e) {
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "libdexamd: ignoring a DEX_* knob whose value '%s' is not an integer\n", e); }
        return KNOB_UNSET;
    }
    return (int)v;
}
struct KnobSnapshot {
    int v[N_KNOBS];
    KnobSnapshot() { for (int i = 0; i < N_KNOBS; ++i) v[i] = knob_parse(getenv(KNOB_NAMES[i])); }
};
thread_local const KnobSnapshot* t_knobs = nullptr;
struct WsplitScope {          // the split-weight mode of the call being built on this thread (lp_dispatch.hip's predicates read it)
    bool prev;
    explicit WsplitScope(bool on) : prev(g_lp_wsplit) { g_lp_wsplit = on; }
    ~WsplitScope() { g_lp_wsplit = prev; }
};
struct KnobScope {            // installs a snapshot for the calls below it on this thread
    const KnobSnapshot* prev;
    explicit KnobScope(const KnobSnapshot* k) : prev(t_knobs) { t_knobs = k; }
    ~KnobScope() { t_knobs = prev; }
};
}  // namespace
// name -> registry index through one hash lookup (the launchers consult several knobs per launch, hundreds of launches per
// eager step, and a lookup was a linear strcmp scan over the registry)
static int knob_index(const char* name) {
    static const std::unordered_map<std::string_view, int> idx = [] {
        std::unordered_map<std::string_view, int> m;
        for (int i = 0; i < N_KNOBS; ++i) m.emplace(KNOB_NAMES[i], i);
        return m;
    }();
    const auto it = idx.find(std::string_view(name));
    return it == idx.end() ? -1 : it->second;
}
int knob(const char* name) {
    if (t_knobs) {
        const int i = knob_index(name);
        if (i >= 0) return t_knobs->v[i];
    }
    return knob_parse(getenv(name));           // outside a call (tools that launch kernels directly), or a knob nobody registered
}
}  // namespace dex
static_assert(PREC_FP32 == DEX_PREC_FP32 && PREC_BF16 == DEX_PREC_BF16 && PREC_FP16 == DEX_PREC_FP16 && PREC_FP16X2 == DEX_PREC_FP16X2, "kernels.h mirrors DexPrecision");

namespace {
void xcd_map_probe();       // (defined with the cluster launch plan below)

struct RawW { float* p = nullptr; std::vector<int64_t> shape; long numel = 0; bool loaded = false; };
struct TD { float* p; int ld; int coff; int C; int lp = 0; };   // channels-last activation view; lp: 16-bit elements (1 bf16, 2 fp16)

struct ResW { const float *w1, *b1, *g1, *be1, *w2, *b2, *g2, *be2, *wr, *br, *mlp_w, *mlp_b; int cin, cout; };
constexpr int ATT_KSPLIT_MAX = 4;     // key-split attention partials kept per token by the 32-query forms (dit_rowchain.hip merges them)
// The 64-query form (attention_q64.hip) may split further where there are few tokens (long-form: one utterance, thousands of tokens, 40
// workgroup-sized query groups for 256 CUs): up to 8 partials there, 4 at batch size (the partial buffer is tokens x hidden x cap fp32).
static int att_split_cap(long tokens) { return tokens <= 20480 ? 8 : ATT_KSPLIT_MAX; }
// ... and takes the attention of a launch when it can fill the chip: >= 1024 tokens per element and >= 128 work units at its best split
static bool attention_q64_regime(int N, int B) {
    if (N < 1024) return dit_sep64_small_n(N, B) && dit_rowchain64_form(N, B, 0);
    const int ks = attention_q64_ksplit(N, B, att_split_cap((long)B * N));
    const int nt32 = (N + 31) / 32, ng = (nt32 + 7) / 8;
    return 2L * B * ng * ks >= 128;
}
struct LinW { const float *wqkv, *wqkv_raw, *wout_raw, *bias_eff, *g; const void *wq_lp[3], *wkv_lp[3], *wq_frag[3]; int C; };   // [0] bf16, [1] fp16, [2] fp16 hi + lo
struct DitBlockW { const float *wqkv, *bqkv, *wproj, *bproj, *wfc1, *bfc1, *wfc2, *bfc2, *ada_w, *ada_b; };

struct Prof { std::string name; hipEvent_t a, b; double flops, bytes; };
struct ProfAgg { std::string name; int calls; double ms, flops, bytes; };

struct Arena {                                             // bump allocator over the caller's workspace
    char* base = nullptr; size_t off = 0, cap = 0; bool dry = true;
    void* take(size_t bytes) {
        off = (off + 255) & ~size_t(255);
        void* p = dry ? nullptr : (void*)(base + off);
        if (!dry && trace) fprintf(stderr, "plan[%d] off %zu bytes %zu\n", trace++, off, bytes);     // DEX_DEBUG_PLAN=1: allocation order = make_plan's source order
        off += bytes;
        return p;
    }
    int trace = 0;
    float* f(size_t n) { return (float*)take(n * sizeof(float)); }
};

}  // namespace

struct DexCtx {
    DexConfig cfg{};
    std::string err;
    std::vector<std::string> keys;
    std::map<std::string, RawW> raw;
    std::vector<void*> owned;                              // hipMalloc'ed packed weights
    // low-precision twins of the fp32 [K][N] packs, one set per operand type: [0] bf16, [1] fp16 (both are packed at
    // finalize: the precision mode may change afterwards)
    // [2] = the split-weight mode (DEX_PREC_FP16X2): fp16 twins whose lo pack (fp16 of what the hi rounding lost, same layout) follows
    // the hi pack; lo_off_ maps a twin pointer to the distance in elements from a hi element to its lo element
    std::map<const float*, const void*> lp_of_[3];         // -> [N][K] twin
    std::map<const float*, const void*> frag_of_[3];       // -> MFMA-fragment-order twin (DiT row chain)
    std::map<const void*, long> lo_off_;
    long lo_off(const void* twin) const { auto it = lo_off_.find(twin); return it == lo_off_.end() ? 0 : it->second; }
    int lpi() const { return precision == DEX_PREC_FP16X2 ? 2 : precision == DEX_PREC_FP16 ? 1 : 0; }
    int lp_kind() const { return precision == DEX_PREC_BF16 ? 1 : 2; }      // the element-wise kernels' runtime code (fp16 for both fp16 modes)
    bool lp() const { return precision != DEX_PREC_FP32; }
    const std::map<const float*, const void*>& lp_of() const { return lp_of_[lpi()]; }
    const std::map<const float*, const void*>& frag_of() const { return frag_of_[lpi()]; }
    bool finalized = false;
    bool tuned = true;                                     // geometry the reduced-precision kernels are built for (dex_ctx_create)
    int precision = DEX_PREC_FP32;
    // packed weights
    std::vector<std::vector<ResW>> down_res, up_res;       // [stage][2]
    std::vector<LinW> down_lin, up_lin;
    std::vector<const float*> down_ds_w, down_ds_b;        // Downsample
    std::vector<const float*> up_us_w, up_us_b;            // Upsample: 4 parity matrices back to back
    const float *fc_w3 = nullptr, *fc_w1 = nullptr;        // first conv packs
    const float *fin_w = nullptr, *fin_b = nullptr, *fin_g = nullptr, *fin_be = nullptr, *fconv_w = nullptr, *fconv_b = nullptr;
    const float *pe_dw = nullptr, *pe_db = nullptr, *pe_pw = nullptr, *pe_pb = nullptr, *pos_w = nullptr, *pos_b = nullptr, *freq_pos = nullptr;
    const void* pos_wfrag[3] = {nullptr, nullptr, nullptr}; // pos-conv weights in MFMA fragment order (pos_conv.hip), bf16 / fp16 / fp16 hi + lo
    std::vector<DitBlockW> blocks;
    const float *fl_w = nullptr, *fl_b = nullptr, *fl_ada_w = nullptr, *fl_ada_b = nullptr;
    const float *tv_wq_raw = nullptr, *tv_wk = nullptr, *tv_wv = nullptr, *tv_wl = nullptr;
    // mel front-end constants
    float *mel_basis = nullptr, *mel_filt = nullptr; void* mel_ws = nullptr; size_t mel_ws_bytes = 0;
    const int* last_xerr = nullptr;     // time-out word of the last call's cluster row chain (inside that call's workspace)
    // asynchronous status (dex_call_status_begin / _poll): a pinned host word the stream copies the call's hand-off word into + the event behind it
    int* st_host = nullptr; hipEvent_t st_ev = nullptr; bool st_pending = false;
    // taps of the last call
    struct Tap { std::string name; const float* p; std::vector<int64_t> shape; };
    std::vector<Tap> taps;
    // profiling
    bool prof_on = false;
    std::vector<Prof> prof;
    std::vector<ProfAgg> prof_agg;
    // hipGraph cache: one captured graph per (shape, pointer set) holds a WHOLE sampler call (conditioning tables, every
    // network evaluation, the final copy) and is replayed with one hipGraphLaunch
    struct GraphEntry { std::vector<uint64_t> key; hipGraphExec_t exec; uint64_t stamp; const int* xerr; };   // xerr: the hand-off word the captured launches write (null: no in-launch hand-offs)
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    void drop_graphs() { for (auto& g : graphs) hipGraphExecDestroy(g.exec); graphs.clear(); }

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};

#define HIPCHK(ctx, call)                                                                              \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                 \
        return (ctx)->fail(DEX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// ================================================================================================
// parameter inventory (mirrors dex_tts_amd/config.py:param_shapes; validated against the reference
// state-dict manifests in tests/golden/manifest_*.json by tests/test_cabi.py)
namespace {

int stage_dim(const DexConfig& c, int i) { return c.dim * c.dim_mults[i]; }
int in_planes(const DexConfig& c) { return 2 + (c.n_spks > 1 ? 1 : 0); }
int mid_dim(const DexConfig& c) { return stage_dim(c, c.n_stages - 1); }
int mid_h(const DexConfig& c) { return c.n_feats >> (c.n_stages - 1); }
int grid_h(const DexConfig& c) { return mid_h(c) / c.dit_stride; }
int token_rows(const DexConfig& c) { return (mid_h(c) + 2 * (c.dit_patch / 2) - c.dit_patch) / c.dit_stride + 1; }
int token_cols(const DexConfig& c, int wmid) {
    const int p = c.dit_patch, s = c.dit_stride;
    const int wp = (wmid % p == 0) ? wmid : wmid + (p - wmid % p);
    return (wp + 2 * (p / 2) - p) / s + 1;
}
int mlp_hidden(const DexConfig& c) { return (int)(c.dit_hidden * c.dit_mlp_ratio); }

void add_key(DexCtx* x, const std::string& k, std::vector<int64_t> shape) {
    x->keys.push_back(k);
    RawW r; r.shape = std::move(shape); r.numel = 1;
    for (auto d : r.shape) r.numel *= d;
    x->raw[k] = r;
}
void add_resnet(DexCtx* x, const std::string& p, int cin, int cout, int tdim) {
    add_key(x, p + ".mlp.1.weight", {cout, tdim}); add_key(x, p + ".mlp.1.bias", {cout});
    for (int blk = 1; blk <= 2; ++blk) {
        const std::string q = p + ".block" + std::to_string(blk) + ".block";
        add_key(x, q + ".0.weight", {cout, blk == 1 ? cin : cout, 3, 3}); add_key(x, q + ".0.bias", {cout});
        add_key(x, q + ".1.weight", {cout}); add_key(x, q + ".1.bias", {cout});
    }
    if (cin != cout) { add_key(x, p + ".res_conv.weight", {cout, cin, 1, 1}); add_key(x, p + ".res_conv.bias", {cout}); }
}
void add_linattn(DexCtx* x, const std::string& p, int c) {
    add_key(x, p + ".fn.g", {1});
    add_key(x, p + ".fn.fn.to_qkv.weight", {384, c, 1, 1});
    add_key(x, p + ".fn.fn.to_out.weight", {c, 128, 1, 1});
    add_key(x, p + ".fn.fn.to_out.bias", {c});
}
void build_inventory(DexCtx* x) {
    const DexConfig& c = x->cfg;
    const int d = c.dim;
    add_key(x, "mlp.0.weight", {4 * d, d}); add_key(x, "mlp.0.bias", {4 * d});
    add_key(x, "mlp.2.weight", {d, 4 * d}); add_key(x, "mlp.2.bias", {d});
    if (c.variant == DEX_VARIANT_DEX)
        for (const char* n : {"mlp_adap", "mlp_adap_sty"}) {
            const std::string s(n);
            add_key(x, s + ".0.weight", {d, d}); add_key(x, s + ".0.bias", {d});
            add_key(x, s + ".2.weight", {2 * d, d}); add_key(x, s + ".2.bias", {2 * d});
        }
    if (c.n_spks > 1) {
        const int e = c.spk_emb_dim;
        add_key(x, "spk_mlp.0.weight", {4 * e, e}); add_key(x, "spk_mlp.0.bias", {4 * e});
        add_key(x, "spk_mlp.2.weight", {c.n_feats, 4 * e}); add_key(x, "spk_mlp.2.bias", {c.n_feats});
    }
    for (int i = 0; i < c.n_stages; ++i) {
        const int ci = i == 0 ? in_planes(c) : stage_dim(c, i - 1), co = stage_dim(c, i);
        const std::string p = "downs." + std::to_string(i);
        add_resnet(x, p + ".0", ci, co, d); add_resnet(x, p + ".1", co, co, d); add_linattn(x, p + ".2", co);
        if (i < c.n_stages - 1) { add_key(x, p + ".3.conv.weight", {co, co, 3, 3}); add_key(x, p + ".3.conv.bias", {co}); }
    }
    const int hid = c.dit_hidden, mid = mid_dim(c), mh = mlp_hidden(c);
    add_key(x, "vit.freq_new_pos_embed", {1, hid, grid_h(c), 1});
    add_key(x, "vit.x_embedder.proj.0.weight", {mid, 1, c.dit_patch, c.dit_patch}); add_key(x, "vit.x_embedder.proj.0.bias", {mid});
    add_key(x, "vit.x_embedder.proj.2.weight", {hid, mid, 1, 1}); add_key(x, "vit.x_embedder.proj.2.bias", {hid});
    add_key(x, "vit.t_embedder.mlp.0.weight", {hid, 256}); add_key(x, "vit.t_embedder.mlp.0.bias", {hid});
    add_key(x, "vit.t_embedder.mlp.2.weight", {hid, hid}); add_key(x, "vit.t_embedder.mlp.2.bias", {hid});
    add_key(x, "vit.pos_conv.0.weight", {hid, hid / c.dit_conv_pos_groups, c.dit_conv_pos, c.dit_conv_pos});
    add_key(x, "vit.pos_conv.0.bias", {hid});
    for (int k = 0; k < c.dit_depth; ++k) {
        const std::string p = "vit.blocks." + std::to_string(k);
        add_key(x, p + ".attn.qkv.weight", {3 * hid, hid}); add_key(x, p + ".attn.qkv.bias", {3 * hid});
        add_key(x, p + ".attn.proj.weight", {hid, hid}); add_key(x, p + ".attn.proj.bias", {hid});
        add_key(x, p + ".mlp.fc1.weight", {mh, hid}); add_key(x, p + ".mlp.fc1.bias", {mh});
        add_key(x, p + ".mlp.fc2.weight", {hid, mh}); add_key(x, p + ".mlp.fc2.bias", {hid});
        add_key(x, p + ".adaLN_modulation.1.weight", {6 * hid, hid}); add_key(x, p + ".adaLN_modulation.1.bias", {6 * hid});
    }
    const int so = c.dit_stride * c.dit_stride * mid;
    add_key(x, "vit.final_layer.linear.weight", {so, hid}); add_key(x, "vit.final_layer.linear.bias", {so});
    add_key(x, "vit.final_layer.adaLN_modulation.1.weight", {2 * hid, hid}); add_key(x, "vit.final_layer.adaLN_modulation.1.bias", {2 * hid});
    if (c.variant == DEX_VARIANT_DEX) {
        for (const char* w : {"w_q", "w_k", "w_v", "linear"}) add_key(x, std::string("tv_adaptor.") + w + ".weight", {mid, mid});
        for (const char* s : {"mean_sap", "std_sap"}) {
            add_key(x, std::string("tiv_adaptor.") + s + ".W.weight", {1, mid});
            add_key(x, std::string("tiv_adaptor.") + s + ".W.bias", {1});
        }
    }
    for (int j = 0; j < c.n_stages - 1; ++j) {
        const int i = c.n_stages - 1 - j;                   // consumes the skip of down stage i
        const int ci = stage_dim(c, i - 1), co = stage_dim(c, i);
        const std::string p = "ups." + std::to_string(j);
        add_resnet(x, p + ".0", co * 2, ci, d); add_resnet(x, p + ".1", ci, ci, d); add_linattn(x, p + ".2", ci);
        add_key(x, p + ".3.conv.weight", {ci, ci, 4, 4}); add_key(x, p + ".3.conv.bias", {ci});
    }
    add_key(x, "final_block.block.0.weight", {d, d, 3, 3}); add_key(x, "final_block.block.0.bias", {d});
    add_key(x, "final_block.block.1.weight", {d}); add_key(x, "final_block.block.1.bias", {d});
    add_key(x, "final_conv.weight", {1, d, 1, 1}); add_key(x, "final_conv.bias", {1});
}

// ConvTranspose2d(4,2,1) -> four 2x2-tap parity sub-convolutions.
// dst[par = ph*2+pw][(th*2+tw)*Cin + ci][co] = src[ci][co][kh][kw],  kh = (ph ? 0 : 1) + 2*th, kw likewise.
__global__ void pack_convt_kernel(const float* src, float* dst, int Cin, int Cout) {
    const long total = 4L * 4 * Cin * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const int ci = (int)((i / Cout) % Cin);
        const int tap = (int)((i / ((long)Cout * Cin)) % 4);
        const int par = (int)(i / ((long)Cout * Cin * 4));
        const int ph = par >> 1, pw = par & 1, th = tap >> 1, tw = tap & 1;
        const int kh = (ph ? 0 : 1) + 2 * th, kw = (pw ? 0 : 1) + 2 * tw;
        dst[i] = src[(((long)ci * Cout + co) * 4 + kh) * 4 + kw];
    }
}

}  // namespace

// ================================================================================================
extern "C" {

const char* dex_version(void) { return "dexamd 0.1 (gfx950)"; }

int dex_ctx_create(const DexConfig* cfg, DexCtx** out) {
    if (!cfg || !out) return DEX_ERR_ARG;
    DexCtx* x = new DexCtx();
    x->cfg = *cfg;
    const DexConfig& c = x->cfg;
    *out = x;
    if (c.n_feats != 80) return x->fail(DEX_ERR_ARG, "n_feats must be 80 (diffusion.py:226 hard-codes it)");
    if (c.n_stages < 2 || c.n_stages > 4) return x->fail(DEX_ERR_ARG, "n_stages must be in [2,4]");
    if (c.dim != 64 && c.dim != 128) return x->fail(DEX_ERR_ARG, "dim must be 64 (GeDEX, DEX-VCTK/ESD) or 128 (DEX-LibriTTS)");
    if (c.dit_heads < 1 || c.dit_hidden % c.dit_heads || !attention_head_dim_supported(c.dit_hidden / c.dit_heads))
        return x->fail(DEX_ERR_ARG, "DiT head_dim must be 64, 128, 192 or 256 (hidden %d heads %d)", c.dit_hidden, c.dit_heads);
    if (c.dit_hidden % 64 || c.dit_hidden > 512) return x->fail(DEX_ERR_ARG, "dit_hidden must be a multiple of 64, <= 512");
    if (mlp_hidden(c) % 64) return x->fail(DEX_ERR_ARG, "mlp hidden must be a multiple of 64");
    if (c.dit_conv_pos_groups < 1 || c.dit_hidden % c.dit_conv_pos_groups || (c.dit_hidden / c.dit_conv_pos_groups) % 16 ||
        c.dit_hidden / c.dit_conv_pos_groups > 64)
        return x->fail(DEX_ERR_ARG, "pos-conv groups must be 16, 32, 48 or 64 channels wide");
    if (c.dit_conv_pos % 2) return x->fail(DEX_ERR_ARG, "conv_pos must be even");
    if (c.variant == DEX_VARIANT_DEX && !attention_head_dim_supported(mid_dim(c)))
        return x->fail(DEX_ERR_ARG, "DEX adaptors need mid_dim 64, 128, 192 or 256 (the cross-attention's head_dim)");
    // The reduced-precision (bf16 / fp16) kernels are specialised for the geometry of every shipped config but DEX-LibriTTS
    // (dim 64, DiT hidden 256 = 2 x 128, 32-channel pos-conv groups); other geometries run the exact-fp32 path only.
    x->tuned = c.dim == 64 && c.dit_hidden == 256 && c.dit_heads == 2 && c.dit_hidden / c.dit_conv_pos_groups == 32 &&
               (c.variant != DEX_VARIANT_DEX || mid_dim(c) == 128);
    build_inventory(x);
    xcd_map_probe();          // per device, outside any stream capture of the caller (it allocates and synchronises)
    return DEX_OK;
}

void dex_ctx_destroy(DexCtx* x) {
    if (!x) return;
    for (auto& kv : x->raw) if (kv.second.p) hipFree(kv.second.p);
    for (void* p : x->owned) hipFree(p);
    if (x->mel_basis) hipFree(x->mel_basis);
    if (x->mel_filt) hipFree(x->mel_filt);
    if (x->mel_ws) hipFree(x->mel_ws);
    x->drop_graphs();
    for (auto& pr : x->prof) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); }
    if (x->st_ev) hipEventDestroy(x->st_ev);
    if (x->st_host) hipHostFree(x->st_host);
    delete x;
}

const char* dex_last_error(const DexCtx* x) { return x ? x->err.c_str() : "null context"; }
int dex_ctx_num_weights(const DexCtx* x) { return x ? (int)x->keys.size() : 0; }

int dex_ctx_weight_info(const DexCtx* x, int i, const char** key, int64_t shape[4], int* ndim) {
    if (!x || i < 0 || i >= (int)x->keys.size()) return DEX_ERR_ARG;
    const RawW& r = x->raw.at(x->keys[i]);
    if (key) *key = x->keys[i].c_str();
    if (ndim) *ndim = (int)r.shape.size();
    if (shape) for (size_t k = 0; k < r.shape.size(); ++k) shape[k] = r.shape[k];
    return DEX_OK;
}

static int load_weight_impl(DexCtx* x, const char* key, const float* w_dev, const int64_t* shape, int ndim, bool async, hipStream_t st) {
    if (!x || !key || !w_dev) return DEX_ERR_ARG;
    auto it = x->raw.find(key);
    if (it == x->raw.end()) return x->fail(DEX_ERR_ARG, "unknown weight key '%s'", key);
    RawW& r = it->second;
    if ((int)r.shape.size() != ndim) return x->fail(DEX_ERR_ARG, "weight '%s': expected %d dims, got %d", key, (int)r.shape.size(), ndim);
    for (int k = 0; k < ndim; ++k)
        if (r.shape[k] != shape[k]) return x->fail(DEX_ERR_ARG, "weight '%s': dim %d is %lld, expected %lld", key, k, (long long)shape[k], (long long)r.shape[k]);
    if (!r.p) HIPCHK(x, hipMalloc((void**)&r.p, r.numel * sizeof(float)));
    if (async) {
        HIPCHK(x, hipMemcpyAsync(r.p, w_dev, r.numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        HIPCHK(x, hipMemcpy(r.p, w_dev, r.numel * sizeof(float), hipMemcpyDeviceToDevice));
        HIPCHK(x, hipStreamSynchronize(nullptr));       // a device-to-device hipMemcpy may return before it has run
    }
    r.loaded = true;
    x->finalized = false;
    return DEX_OK;
}
int dex_ctx_load_weight(DexCtx* x, const char* key, const float* w_dev, const int64_t* shape, int ndim) {
    return load_weight_impl(x, key, w_dev, shape, ndim, false, nullptr);
}
int dex_ctx_load_weight_async(DexCtx* x, const char* key, const float* w_dev, const int64_t* shape, int ndim, dex_stream_t stream) {
    return load_weight_impl(x, key, w_dev, shape, ndim, true, (hipStream_t)stream);
}

int dex_ctx_set_precision(DexCtx* x, int precision) {
    if (!x || (precision != DEX_PREC_FP32 && precision != DEX_PREC_BF16 && precision != DEX_PREC_FP16 && precision != DEX_PREC_FP16X2)) return DEX_ERR_ARG;
    // Geometries the tuned reduced-precision kernels do not cover (DEX-LibriTTS: dim 128 -> 128 / 256-channel stages, DiT hidden 384 =
    // 2 x 192, 48-channel pos-conv groups, 256-channel TVAdaptor; DexCtx::tuned == false) run the same modes PER OPERATION: every
    // convolution / linear layer with a packed 16-bit weight twin goes through the generic reduced-precision implicit GEMM (or the
    // tuned kernel where its shape predicate holds: the 128 -> 128 convolutions, the 128-channel linear attention), and whatever has
    // no 16-bit form for the shape - softmax attention at head_dim 192 / 256, the padded 48-channel pos-conv groups, GroupNorm as
    // its own pass - stays on the exact-fp32 kernels.  Slower than a tuned geometry, same operand rounding.
    x->precision = precision;
    return DEX_OK;
}

}  // extern "C"

// ================================================================================================
// weight packing
namespace {

struct Packer {
    DexCtx* x; hipStream_t st; int rc = DEX_OK;
    float* alloc(long n) {
        float* p = nullptr;
        if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { rc = x->fail(DEX_ERR_HIP, "hipMalloc of %ld floats failed", n); return nullptr; }
        x->owned.push_back(p);
        return p;
    }
    // twin set t: 0 bf16, 1 fp16, 2 fp16 hi + lo.  pack3 runs `pack(src, dst, precision)` for the set: once, or (t == 2) on the weight
    // and on what its fp16 rounding lost (src - float(fp16(src)), through a scratch fp32 copy - packs are permutations, so the lo pack
    // has the hi pack's layout), the lo pack `n_dst` elements behind the hi pack.
    float* tmp = nullptr; long tmp_n = 0;
    float* scratch(long n) {
        if (n > tmp_n) { tmp = alloc(n); tmp_n = tmp ? n : 0; }     // (stream-ordered reuse: every pack of this Packer runs on `st`)
        return tmp;
    }
    static constexpr int NSETS = 3;
    static long set_elems(int t, long n) { return t == 2 ? 2 * n : n; }
    template <class F> void pack3(int t, const float* src, long n_src, unsigned short* dst, long n_dst, F pack) {
        if (t < 2) { pack(src, dst, t ? PREC_FP16 : PREC_BF16); return; }
        pack(src, dst, PREC_FP16);
        float* lo = scratch(n_src);
        if (!lo) return;
        launch_f32_residual_lp(src, lo, n_src, PREC_FP16, st);
        pack(lo, dst + n_dst, PREC_FP16);
    }
    // 16-bit [N][K] twins of `count` consecutive fp32 [K][N] matrices starting at p
    void twin(const float* p, int count, int K, int N) {
        if (!p) return;
        for (int t = 0; t < NSETS; ++t) {
            const long n = (long)count * K * N;
            unsigned short* d = (unsigned short*)alloc((set_elems(t, n) + 1) / 2);
            if (!d) return;
            for (int c = 0; c < count; ++c) {
                unsigned short* dc = d + (long)c * K * N;
                pack3(t, p + (long)c * K * N, (long)K * N, dc, n, [&](const float* s, unsigned short* o, int prec) { launch_pack_lp_nk(s, o, K, N, prec, st); });
                x->lp_of_[t][p + (long)c * K * N] = dc;
                if (t == 2) x->lo_off_[dc] = n;
            }
        }
    }
    void frag(const float* p, int K, int N) {
        if (!p) return;
        for (int t = 0; t < NSETS; ++t) {
            const long n = (long)K * N;
            unsigned short* d = (unsigned short*)alloc((set_elems(t, n) + 1) / 2);
            if (!d) return;
            pack3(t, p, n, d, n, [&](const float* s, unsigned short* o, int prec) { launch_pack_lp_frag(s, o, K, N, prec, st); });
            x->frag_of_[t][p] = d;
            if (t == 2) x->lo_off_[d] = n;
        }
    }
    const RawW& R(const std::string& k) { return x->raw.at(k); }
    const float* raw(const std::string& k) { return R(k).p; }
    // [d0,d1,d2,d3] -> permuted contiguous copy
    const float* perm(const std::string& k, int d0, int d1, int d2, int d3, int p0, int p1, int p2, int p3) {
        float* dst = alloc((long)d0 * d1 * d2 * d3);
        if (dst) launch_permute4(raw(k), dst, d0, d1, d2, d3, p0, p1, p2, p3, st);
        return dst;
    }
    // conv / linear weight [out, in, kh, kw] -> [(kh*KW+kw)*in + ci][out]
    const float* kn(const std::string& k) {
        const auto& s = R(k).shape;
        const int o = (int)s[0], i = (int)s[1], kh = s.size() > 2 ? (int)s[2] : 1, kw = s.size() > 3 ? (int)s[3] : 1;
        const float* f = perm(k, o, i, kh, kw, 2, 3, 1, 0);
        twin(f, 1, i * kh * kw, o);
        if (kh == 3 && kw == 3 && ((i == 128 && o == 128) || (i == 64 && o == 128))) frag(f, 9 * i, o);      // weights-in-registers strip convolutions (conv3x3_regw.hip)
        if (kh == 1 && kw == 1 && i == 64 && o == 128) frag(f, i, o);                                          // ... and the 1x1 shortcut fused into the 64 -> 128 one
        return f;
    }
    ResW resnet(const std::string& p, int cin, int cout, bool first) {
        ResW r{};
        r.cin = cin; r.cout = cout;
        const std::string b1 = p + ".block1.block", b2 = p + ".block2.block";
        if (first) {   // [C, planes, 3, 3] -> [planes*9][C]
            r.w1 = perm(b1 + ".0.weight", cout, cin, 3, 3, 1, 2, 3, 0);
            r.wr = perm(p + ".res_conv.weight", cout, cin, 1, 1, 1, 2, 3, 0);
        } else {
            r.w1 = kn(b1 + ".0.weight");
            r.wr = (cin != cout) ? kn(p + ".res_conv.weight") : nullptr;
        }
        r.br = (cin != cout) ? raw(p + ".res_conv.bias") : nullptr;
        r.b1 = raw(b1 + ".0.bias"); r.g1 = raw(b1 + ".1.weight"); r.be1 = raw(b1 + ".1.bias");
        r.w2 = kn(b2 + ".0.weight");
        r.b2 = raw(b2 + ".0.bias"); r.g2 = raw(b2 + ".1.weight"); r.be2 = raw(b2 + ".1.bias");
        r.mlp_w = raw(p + ".mlp.1.weight"); r.mlp_b = raw(p + ".mlp.1.bias");
        return r;
    }
    LinW linattn(const std::string& p, int c) {
        LinW l{};
        l.C = c;
        l.wqkv = kn(p + ".fn.fn.to_qkv.weight");
        l.wqkv_raw = raw(p + ".fn.fn.to_qkv.weight");                 // [384][C]: rows q | k | v
        l.wout_raw = raw(p + ".fn.fn.to_out.weight");
        {   // bf16 copy of the q | k | v rows in their native [N][K] layout (MFMA operands of the fused kernels)
            for (int t = 0; t < NSETS; ++t) {
                unsigned short* qb = (unsigned short*)alloc((set_elems(t, 384L * c) + 1) / 2);
                if (qb) pack3(t, l.wqkv_raw, 384L * c, qb, 384L * c, [&](const float* s, unsigned short* o, int prec) { launch_f32_to_lp(s, o, 384L * c, prec, st); });
                l.wq_lp[t] = qb; l.wkv_lp[t] = qb ? qb + 128L * c : nullptr;        // (t == 2: the lo rows 384 c elements behind)
                // the q rows again in MFMA fragment order (A operand of the tail's first GEMM: 1 KB contiguous per wave load)
                unsigned short* qf = (unsigned short*)alloc((set_elems(t, 128L * c) + 1) / 2);
                if (qf) pack3(t, l.wqkv_raw, 128L * c, qf, 128L * c, [&](const float* s, unsigned short* o, int prec) { launch_pack_lp_frag_nk(s, o, c, 128, prec, st); });
                l.wq_frag[t] = qf;
            }
        }
        l.g = raw(p + ".fn.g");
        float* be = alloc(c);
        if (be) launch_scale_copy(raw(p + ".fn.fn.to_out.bias"), be, c, l.g, st);     // Rezero gate folded into the bias
        l.bias_eff = be;
        return l;
    }
};

// mel front-end constants (audio/stft.py:26-47,145-147), built on the host in double, rounded like the reference
void build_mel_constants(std::vector<float>& basis, std::vector<float>& filt) {
    const int n_fft = 1024, nb = 513, NP = 1152, IM = 576;
    basis.assign((size_t)n_fft * NP, 0.f);
    for (int n = 0; n < n_fft; ++n) {
        const float win = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / n_fft));         // periodic Hann
        for (int k = 0; k < nb; ++k) {
            const double ang = 2.0 * M_PI * (double)(((long)k * n) % n_fft) / n_fft;
            basis[(size_t)n * NP + k] = (float)cos(ang) * win;
            basis[(size_t)n * NP + IM + k] = (float)(-sin(ang)) * win;
        }
    }
    // librosa.filters.mel(22050, 1024, 80, 0, 8000), Slaney scale + slaney area normalisation
    const int n_mels = 80; const double sr = 22050, fmin = 0, fmax = 8000;
    auto hz2mel = [](double f) { const double fsp = 200.0 / 3; return f >= 1000.0 ? 1000.0 / fsp + log(f / 1000.0) / (log(6.4) / 27.0) : f / fsp; };
    auto mel2hz = [](double m) { const double fsp = 200.0 / 3, mlm = 1000.0 / fsp; return m >= mlm ? 1000.0 * exp(log(6.4) / 27.0 * (m - mlm)) : fsp * m; };
    std::vector<double> mf(n_mels + 2);
    const double m0 = hz2mel(fmin), m1 = hz2mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i) mf[i] = mel2hz(m0 + (m1 - m0) * i / (n_mels + 1));
    filt.assign((size_t)n_mels * nb, 0.f);
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mf[i + 2] - mf[i]);
        for (int k = 0; k < nb; ++k) {
            const double fr = (sr / 2.0) * k / (nb - 1);
            const double lower = (fr - mf[i]) / (mf[i + 1] - mf[i]), upper = (mf[i + 2] - fr) / (mf[i + 2] - mf[i + 1]);
            const double w = fmax > 0 ? std::max(0.0, std::min(lower, upper)) : 0.0;
            filt[(size_t)i * nb + k] = (float)(w * enorm);
        }
    }
}

}  // namespace

extern "C" int dex_ctx_finalize(DexCtx* x, dex_stream_t stream) {
    if (!x) return DEX_ERR_ARG;
    for (const auto& k : x->keys)
        if (!x->raw.at(k).loaded) return x->fail(DEX_ERR_STATE, "weight '%s' was never loaded", k.c_str());
    for (void* p : x->owned) hipFree(p);
    x->owned.clear();
    for (int t = 0; t < 3; ++t) { x->lp_of_[t].clear(); x->frag_of_[t].clear(); }
    x->lo_off_.clear();
    x->drop_graphs();
    const DexConfig& c = x->cfg;
    hipStream_t st = (hipStream_t)stream;
    Packer P{x, st};
    x->down_res.assign(c.n_stages, {}); x->down_lin.clear(); x->down_ds_w.clear(); x->down_ds_b.clear();
    for (int i = 0; i < c.n_stages; ++i) {
        const int ci = i == 0 ? in_planes(c) : stage_dim(c, i - 1), co = stage_dim(c, i);
        const std::string p = "downs." + std::to_string(i);
        x->down_res[i].push_back(P.resnet(p + ".0", ci, co, i == 0));
        x->down_res[i].push_back(P.resnet(p + ".1", co, co, false));
        x->down_lin.push_back(P.linattn(p + ".2", co));
        if (i < c.n_stages - 1) {
            x->down_ds_w.push_back(P.kn(p + ".3.conv.weight")); x->down_ds_b.push_back(P.raw(p + ".3.conv.bias"));
            if (conv_down_supported(co, 2, 2, co, co, 0)) P.frag(x->down_ds_w.back(), 9 * co, co);      // conv_down.hip
        }
    }
    x->up_res.assign(c.n_stages - 1, {}); x->up_lin.clear(); x->up_us_w.clear(); x->up_us_b.clear();
    for (int j = 0; j < c.n_stages - 1; ++j) {
        const int i = c.n_stages - 1 - j, ci = stage_dim(c, i - 1), co = stage_dim(c, i);
        const std::string p = "ups." + std::to_string(j);
        x->up_res[j].push_back(P.resnet(p + ".0", co * 2, ci, false));
        x->up_res[j].push_back(P.resnet(p + ".1", ci, ci, false));
        x->up_lin.push_back(P.linattn(p + ".2", ci));
        float* wt = P.alloc(16L * ci * ci);
        if (wt) {
            hipLaunchKernelGGL(pack_convt_kernel, dim3(256), dim3(256), 0, st, P.raw(p + ".3.conv.weight"), wt, ci, ci);
            P.twin(wt, 4, 4 * ci, ci);
            if (convt_up_supported(ci, 1, 1, ci, ci)) for (int par = 0; par < 4; ++par) P.frag(wt + (long)par * 4 * ci * ci, 4 * ci, ci);   // convt_up.hip
        }
        x->up_us_w.push_back(wt); x->up_us_b.push_back(P.raw(p + ".3.conv.bias"));
    }
    x->fin_w = P.kn("final_block.block.0.weight"); x->fin_b = P.raw("final_block.block.0.bias");
    x->fin_g = P.raw("final_block.block.1.weight"); x->fin_be = P.raw("final_block.block.1.bias");
    x->fconv_w = P.raw("final_conv.weight"); x->fconv_b = P.raw("final_conv.bias");
    const int hid = c.dit_hidden, mid = mid_dim(c), G = c.dit_conv_pos_groups, kp = c.dit_conv_pos;
    x->pe_dw = P.perm("vit.x_embedder.proj.0.weight", mid, 1, c.dit_patch, c.dit_patch, 2, 3, 1, 0);
    x->pe_db = P.raw("vit.x_embedder.proj.0.bias");
    x->pe_pw = P.kn("vit.x_embedder.proj.2.weight"); x->pe_pb = P.raw("vit.x_embedder.proj.2.bias");
    x->pos_w = P.perm("vit.pos_conv.0.weight", G, hid / G, hid / G, kp * kp, 0, 3, 2, 1);   // [G][tap][ci][n]
    if ((hid / G) % 32) {       // e.g. 48-channel groups (hidden 384): [G][tap][ci][n] -> zero-padded [G][tap][cgp][cgp], cgp = 64
        const int cg = hid / G, cgp = 64;
        float* wp = P.alloc((long)G * kp * kp * cgp * cgp);
        if (wp) {
            hipMemsetAsync(wp, 0, (size_t)G * kp * kp * cgp * cgp * sizeof(float), st);
            for (long gt = 0; gt < (long)G * kp * kp; ++gt)     // one 2-D copy per (group, tap): cg rows of cg floats into a cgp x cgp tile
                hipMemcpy2DAsync(wp + gt * cgp * cgp, (size_t)cgp * 4, x->pos_w + gt * cg * cg, (size_t)cg * 4, (size_t)cg * 4, cg, hipMemcpyDeviceToDevice, st);
        }
        x->pos_w = wp;
    } else
    P.twin(x->pos_w, G, kp * kp * (hid / G), hid / G);
    x->pos_wfrag[0] = x->pos_wfrag[1] = x->pos_wfrag[2] = nullptr;
    if (pos_conv_direct_supported(hid, G, kp, token_rows(c))) {
        const long kn_ = (long)kp * kp * (hid / G) * (hid / G);
        for (int t = 0; t < Packer::NSETS; ++t) {
            unsigned short* wf = (unsigned short*)P.alloc((Packer::set_elems(t, G * kn_) + 1) / 2);
            if (wf) for (int g = 0; g < G; ++g)
                P.pack3(t, x->pos_w + g * kn_, kn_, wf + g * kn_, G * kn_, [&](const float* s, unsigned short* o, int prec) { launch_pack_lp_frag(s, o, kp * kp * (hid / G), hid / G, prec, st); });
            x->pos_wfrag[t] = wf;
        }
    }
    x->pos_b = P.raw("vit.pos_conv.0.bias");
    x->freq_pos = P.perm("vit.freq_new_pos_embed", 1, hid, grid_h(c), 1, 0, 2, 3, 1);       // [Hf][hid]
    x->blocks.clear();
    for (int k = 0; k < c.dit_depth; ++k) {
        const std::string p = "vit.blocks." + std::to_string(k);
        DitBlockW b{};
        b.wqkv = P.kn(p + ".attn.qkv.weight"); b.bqkv = P.raw(p + ".attn.qkv.bias");
        b.wproj = P.kn(p + ".attn.proj.weight"); b.bproj = P.raw(p + ".attn.proj.bias");
        b.wfc1 = P.kn(p + ".mlp.fc1.weight"); b.bfc1 = P.raw(p + ".mlp.fc1.bias");
        b.wfc2 = P.kn(p + ".mlp.fc2.weight"); b.bfc2 = P.raw(p + ".mlp.fc2.bias");
        b.ada_w = P.raw(p + ".adaLN_modulation.1.weight"); b.ada_b = P.raw(p + ".adaLN_modulation.1.bias");
        if (dit_rowchain_supported(hid, mlp_hidden(c))) {
            P.frag(b.wqkv, hid, 3 * hid); P.frag(b.wproj, hid, hid); P.frag(b.wfc1, hid, mlp_hidden(c)); P.frag(b.wfc2, mlp_hidden(c), hid);
        }
        x->blocks.push_back(b);
    }
    x->fl_w = P.kn("vit.final_layer.linear.weight"); x->fl_b = P.raw("vit.final_layer.linear.bias");
    if (hid == 256) P.frag(x->fl_w, hid, c.dit_stride * c.dit_stride * mid_dim(c));      // the column walker's LDS-DMA ring takes fragment-ordered tiles (igemm_lp_nwalk_kernel)
    x->fl_ada_w = P.raw("vit.final_layer.adaLN_modulation.1.weight"); x->fl_ada_b = P.raw("vit.final_layer.adaLN_modulation.1.bias");
    if (c.variant == DEX_VARIANT_DEX) {
        x->tv_wq_raw = P.raw("tv_adaptor.w_q.weight");
        x->tv_wk = P.kn("tv_adaptor.w_k.weight"); x->tv_wv = P.kn("tv_adaptor.w_v.weight"); x->tv_wl = P.kn("tv_adaptor.linear.weight");
    }
    if (!x->mel_basis) {
        std::vector<float> basis, filt;
        build_mel_constants(basis, filt);
        HIPCHK(x, hipMalloc((void**)&x->mel_basis, basis.size() * sizeof(float)));
        HIPCHK(x, hipMalloc((void**)&x->mel_filt, filt.size() * sizeof(float)));
        HIPCHK(x, hipMemcpy(x->mel_basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(x, hipMemcpy(x->mel_filt, filt.data(), filt.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (P.rc != DEX_OK) return P.rc;
    HIPCHK(x, hipStreamSynchronize(st));
    HIPCHK(x, hipGetLastError());
    x->finalized = true;
    return DEX_OK;
}

// ================================================================================================
// workspace plan + step enqueue
namespace {

constexpr int SCAL_STRIDE = 8;
constexpr int LA_CHUNK = 512;   // == LA_POS * LA_SUBT in linattn.hip
constexpr int POS_SPLIT = 8;

struct Dims { int B, T, Tr, Ts, n_steps; };

struct StageBuf {
    int H, W, C, mask_ws; long npix;
    float *h1, *a1, *h2, *rbuf, *r0out, *r1out;        // resblock scratch
    float *qkv, *pm, *ps, *pc, *weff, *ctxn; void* mbf; int nchunks, nblk_fused;
    float* attn_out; int attn_ld, attn_coff;            // where the stage's attention output lives
    float* ds_out;                                      // Downsample output (down stages except the last)
};

struct Plan {
    Dims d;
    float *sig2, *scal, *t_unet, *t_dit, *tmp_u, *temb, *c_tmp, *c_emb, *fin_mod;
    std::vector<float*> tadd_down, tadd_up, ada;
    float *adap_tmp, *t_adap, *t_sty, *tv_k0, *tv_v0, *sap_m, *sap_s, *ref_mean, *ref_std;
    float *spk_tmp, *spk_plane;
    int* step; int* step_tab; gnfix_t* stats; long stats_bytes; int n_gn;   // stats: two arenas (Euler-step parity)
    float* xbuf;
    float *xprime, *dbuf, *hsig, *htab;      // Heun: x', slope d_cur, per-evaluation sigma / h tables
    float *that, *hstep, *ncoef;             // churn tables per STEP: t_hat, h = t_next - t_hat, noise coefficient (edm.py:194-196)
    std::vector<StageBuf> down, up;
    std::vector<float*> cat;
    void* cat16;                                             // 16-bit twin of cat[0] (n_stages == 2, batch regime: Runner::cat_lp)
    float *up_out, *hF;
    int Hm, Wm, Hf, Wt, N;
    float *pe0, *emb, *emb_pad, *pos_part, *tok, *xn, *qkv, *ao, *att_ml, *hmlp, *dbg_tok;
    float* xslab; unsigned* xflag; size_t xflag_bytes;      // cluster form of the DiT row chain (small grids): exchange slabs + flag words
    int xlocal;                                             // ... with the members of a cluster on one XCD (xcd_map_ok())
    void *qh, *kh, *vt, *qh2, *kh2, *vt2; int Npad; size_t vt_bytes;   // bf16 attention operands of the row-chain path (two sets:
                                                                        // a fused block reads one while its workgroups write the other)
    float *tv_keys, *tv_K, *tv_V, *tv_q, *tv_ao, *tv_out, *tiv_out, *tv_weff, *tv_beff; gnfix_t *tv_stats, *tiv_stats; void* tv_wbf;
    float* tiv_aff;                            // TIV adaptor folded into the patch embedding's load: [B][2][mid] coefficients (launch_tiv_coef)
    void *tv_kp, *tv_vtp; int tv_nkpad;        // the one-launch TV adaptor's 16-bit key / value operands (TvKvPrepP)
    float *tv_G, *tv_Vp, *tv_g0, *tv_v0p, *tv_xmean;   // its folded form (TvFold2P): G = K W_q, V' = V W_l^T ([B][Ts + 1][mid], row 0 unused), the time token's rows per step, the IN2d means
    size_t bytes;
};

// Zero-fill as a KERNEL node.  hipMemsetAsync inside a captured call becomes a memset node, and replays of such a graph were
// measured to go wrong after unrelated eager work on the legacy null stream (rocm 7.2: the first Euler step then sees late
// zeroes - flags, statistics, key padding); a kernel node is ordered like every other kernel of the chain.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* p16, size_t n16, unsigned char* tail, int ntail) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p16[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
static void zero_fill(void* p, size_t bytes, hipStream_t st) {
    if (!p || !bytes) return;
    const size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256; if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(p), n16,
                       reinterpret_cast<unsigned char*>(p) + n16 * 16, (int)(bytes - n16 * 16));
}

// Does workgroup b of a launch run on XCD b % 8?  The XCD-local cluster form of the DiT row chain (dit_rowchain.hip) places the
// four members of a cluster by that rule so that their hand-offs stay inside one L2.  Probed once per process with launches of
// the same shape (256 workgroups x 512 threads, 64 KB of LDS) that record HW_REG_XCC_ID; run from the public entry points
// BEFORE any stream capture (it allocates).  The kernel re-checks every hand-off (flag words carry the writer's XCC id).
// State is PER DEVICE ORDINAL (a process driving several GPUs or partitions must not reuse the first device's answer) and carries a
// generation that is part of every graph-cache key: when a hand-off reports an XCC mismatch the form is switched off for the device
// and the generation moves, so EVERY context stops hitting graphs captured with the XCD-local form.
constexpr int MAX_DEVICES = 64;
static std::atomic<int> g_xcd_map_dev[MAX_DEVICES];      // 0 not probed, 1 no, 2 yes  (zero-initialised)
static std::atomic<int> g_xcd_gen{0};
static std::mutex g_xcd_probe_mutex;         // (hosts with one context per thread: the probe runs once per device)
static int cur_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return (d >= 0 && d < MAX_DEVICES) ? d : 0; }
struct XcdMapRef {           // drop-in for the old process-wide atomic: load() = -1 not probed / 0 no / 1 yes of the CURRENT device
    int load(std::memory_order = std::memory_order_seq_cst) const { return g_xcd_map_dev[cur_device()].load(std::memory_order_acquire) - 1; }
    void store(int v, std::memory_order = std::memory_order_seq_cst) const {
        const int old = g_xcd_map_dev[cur_device()].exchange(v + 1, std::memory_order_acq_rel);
        if (old != v + 1) g_xcd_gen.fetch_add(1, std::memory_order_acq_rel);
    }
};
static const XcdMapRef g_xcd_map;
__global__ __launch_bounds__(512) void xcc_probe_kernel(unsigned* out) {
    extern __shared__ unsigned char probe_lds[];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        probe_lds[0] = (unsigned char)x;
        out[blockIdx.x] = x & 15u;
    }
}
void xcd_map_probe() {
    if (g_xcd_map.load(std::memory_order_acquire) >= 0) return;
    std::lock_guard<std::mutex> lk(g_xcd_probe_mutex);
    if (g_xcd_map.load(std::memory_order_relaxed) >= 0) return;
    if (knob_off("DEX_DIT_CLUSTER_LOCAL")) { g_xcd_map.store(0, std::memory_order_release); return; }
    unsigned* d = nullptr;
    if (hipMalloc(&d, 256 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); g_xcd_map.store(0, std::memory_order_release); return; }
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&xcc_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess;
    unsigned h[256];
    for (int rep = 0; rep < 3 && ok; ++rep) {
        const int nb = rep == 0 ? 256 : rep == 1 ? 96 : 192;
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(nb), dim3(512), 65536, 0, d);
        ok = hipMemcpy(h, d, nb * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
        for (int b = 8; b < nb && ok; ++b) ok = h[b] == h[b & 7];
    }
    hipFree(d);
    g_xcd_map.store(ok ? 1 : 0, std::memory_order_release);
}
extern "C" int dex_debug_xcd_local() { xcd_map_probe(); return g_xcd_map.load(); }

void make_plan(const DexCtx* x, const Dims& d, void* ws, Plan& P) {
    const DexConfig& c = x->cfg;
    Arena A; A.base = (char*)ws; A.dry = (ws == nullptr);
    A.trace = knob_set("DEX_DEBUG_PLAN") ? 1 : 0;
    P.d = d;
    const int n = d.n_steps, dim = c.dim, hid = c.dit_hidden, mid = mid_dim(c), B = d.B;
    P.sig2 = A.f(64);
    P.scal = A.f((size_t)n * SCAL_STRIDE); P.t_unet = A.f((size_t)n * dim); P.t_dit = A.f((size_t)n * 256);
    P.tmp_u = A.f((size_t)n * 4 * dim); P.temb = A.f((size_t)n * dim);
    P.c_tmp = A.f((size_t)n * hid); P.c_emb = A.f((size_t)n * hid); P.fin_mod = A.f((size_t)n * 2 * hid);
    P.tadd_down.clear(); P.tadd_up.clear(); P.ada.clear();
    for (int i = 0; i < c.n_stages; ++i) for (int r = 0; r < 2; ++r) P.tadd_down.push_back(A.f((size_t)n * stage_dim(c, i)));
    for (int j = 0; j < c.n_stages - 1; ++j) for (int r = 0; r < 2; ++r) P.tadd_up.push_back(A.f((size_t)n * stage_dim(c, c.n_stages - 2 - j)));
    for (int k = 0; k < c.dit_depth; ++k) P.ada.push_back(A.f((size_t)n * 6 * hid));
    P.adap_tmp = P.t_adap = P.t_sty = P.tv_k0 = P.tv_v0 = P.sap_m = P.sap_s = P.ref_mean = P.ref_std = nullptr;
    if (c.variant == DEX_VARIANT_DEX) {
        P.adap_tmp = A.f((size_t)n * dim); P.t_adap = A.f((size_t)n * 2 * dim); P.t_sty = A.f((size_t)n * 2 * dim);
        P.tv_k0 = A.f((size_t)n * mid); P.tv_v0 = A.f((size_t)n * mid);
        P.sap_m = A.f((size_t)n * B * mid); P.sap_s = A.f((size_t)n * B * mid);
        P.ref_mean = A.f((size_t)B * 8 * mid); P.ref_std = A.f((size_t)B * 8 * mid);
    }
    P.spk_tmp = P.spk_plane = nullptr;
    if (c.n_spks > 1) { P.spk_tmp = A.f((size_t)B * 4 * c.spk_emb_dim); P.spk_plane = A.f((size_t)B * c.n_feats); }
    P.step = (int*)A.take(256);
    P.step_tab = (int*)A.take((size_t)(n + 1) * sizeof(int));
    P.n_gn = 4 * c.n_stages + 4 * (c.n_stages - 1) + 1;
    P.stats_bytes = (long)P.n_gn * B * 8 * GN_SLOTS * 2 * sizeof(gnfix_t);
    P.stats = (gnfix_t*)A.take(2 * P.stats_bytes);
    P.xbuf = A.f((size_t)B * 80 * d.T);
    P.xprime = A.f((size_t)B * 80 * d.T); P.dbuf = A.f((size_t)B * 80 * d.T);
    P.hsig = A.f((size_t)n + 2); P.htab = A.f((size_t)n + 2);
    P.that = A.f((size_t)n + 2); P.hstep = A.f((size_t)n + 2); P.ncoef = A.f((size_t)n + 2);
    P.cat.assign(c.n_stages - 1, nullptr);
    P.cat16 = nullptr;
    if (c.n_stages == 2) P.cat16 = A.take((size_t)B * (c.n_feats >> 1) * (d.T >> 1) * 2 * stage_dim(c, 1) * 2);
    for (int j = 0; j < c.n_stages - 1; ++j) {
        const int i = c.n_stages - 1 - j;
        P.cat[j] = A.f((size_t)B * (c.n_feats >> i) * (d.T >> i) * 2 * stage_dim(c, i));
    }
    auto stage = [&](int H, int W, int C, int mws) {
        StageBuf s{};
        s.H = H; s.W = W; s.C = C; s.mask_ws = mws; s.npix = (long)H * W;
        const size_t e = (size_t)B * s.npix;
        s.h1 = A.f(e * C); s.a1 = A.f(e * C); s.h2 = A.f(e * C); s.rbuf = A.f(e * C); s.r0out = A.f(e * C); s.r1out = A.f(e * C);
        s.qkv = A.f(e * 384);
        s.nchunks = (int)((s.npix + LA_CHUNK - 1) / LA_CHUNK);
        s.nblk_fused = (int)((s.npix + 127) / 128);             // fused bf16 path: one partial per 128-pixel workgroup
        s.pm = A.f((size_t)B * 4 * s.nblk_fused * 32); s.ps = A.f((size_t)B * 4 * s.nblk_fused * 32);
        s.pc = A.f((size_t)B * 4 * s.nblk_fused * 1024); s.weff = A.f((size_t)B * 128 * C);
        s.ctxn = A.f((size_t)B * 4096); s.mbf = A.take((size_t)B * C * 128 * 2);
        s.ds_out = nullptr;
        return s;
    };
    P.down.clear(); P.up.clear();
    for (int i = 0; i < c.n_stages; ++i) {
        StageBuf s = stage(c.n_feats >> i, d.T >> i, stage_dim(c, i), 1 << i);
        if (i >= 1) { const int j = c.n_stages - 1 - i; s.attn_out = P.cat[j]; s.attn_ld = 2 * s.C; s.attn_coff = s.C; }
        else { s.attn_out = A.f((size_t)B * s.npix * s.C); s.attn_ld = s.C; s.attn_coff = 0; }
        if (i < c.n_stages - 1) s.ds_out = A.f((size_t)B * (s.npix / 4) * s.C);
        P.down.push_back(s);
    }
    for (int j = 0; j < c.n_stages - 1; ++j) {
        const int i = c.n_stages - 1 - j;
        StageBuf s = stage(c.n_feats >> i, d.T >> i, stage_dim(c, i - 1), 1 << i);
        s.attn_out = A.f((size_t)B * s.npix * s.C); s.attn_ld = s.C; s.attn_coff = 0;
        P.up.push_back(s);
    }
    P.up_out = A.f((size_t)B * 80 * d.T * dim);
    P.hF = A.f((size_t)B * 80 * d.T * dim);
    P.Hm = mid_h(c); P.Wm = d.T >> (c.n_stages - 1);
    P.Hf = token_rows(c); P.Wt = token_cols(c, P.Wm); P.N = P.Hf * P.Wt;
    const size_t tok = (size_t)B * P.N;
    const int cg_ = hid / c.dit_conv_pos_groups, cgp_ = (cg_ % 32) ? 64 : cg_;      // pos-conv group width, padded to the GEMM's granule
    P.pe0 = A.f(tok * mid); P.emb = A.f(tok * hid); P.pos_part = A.f(tok * (size_t)c.dit_conv_pos_groups * cgp_ * POS_SPLIT); P.tok = A.f(tok * hid);
    P.emb_pad = cgp_ != cg_ ? A.f(tok * (size_t)c.dit_conv_pos_groups * cgp_) : nullptr;
    P.xn = A.f(tok * hid); P.qkv = A.f(tok * 3 * hid); P.ao = A.f(tok * hid * att_split_cap((long)tok)); P.att_ml = A.f(tok * c.dit_heads * 2 * att_split_cap((long)tok));
    P.Npad = (P.N + 31) / 32 * 32; P.vt_bytes = (size_t)B * hid * P.Npad * 2;
    P.qh = A.take(P.vt_bytes); P.kh = A.take(P.vt_bytes); P.vt = A.take(P.vt_bytes);    // all three padded to Npad rows
    P.qh2 = A.take(P.vt_bytes); P.kh2 = A.take(P.vt_bytes); P.vt2 = A.take(P.vt_bytes);
    P.hmlp = A.f(tok * mlp_hidden(c));
    P.xslab = nullptr; P.xflag = nullptr; P.xflag_bytes = 0; P.xlocal = 0;
    if (dit_rowchain_supported(hid, mlp_hidden(c)) && dit_rowchain_cluster_form(P.N, B)) {
        const size_t tiles = ((size_t)B * ((P.N + 31) / 32) + 7) / 8 * 8;                      // (whole rounds of 8 clusters: the XCD-local grid)
        P.xslab = A.f(tiles * DIT_CLUSTER_SLAB_FLOATS);
        P.xflag_bytes = tiles * DIT_CLUSTER_FLAG_WORDS * sizeof(unsigned) + sizeof(int);      // + the time-out word
        P.xflag = (unsigned*)A.take(P.xflag_bytes);
        P.xlocal = (dit_rowchain_cluster_local_fits(P.N, B) && g_xcd_map.load(std::memory_order_relaxed) == 1) ? 1 : 0;
    }
    P.dbg_tok = A.f(tok * hid * (c.dit_depth + 1));
    P.tv_keys = P.tv_K = P.tv_V = P.tv_q = P.tv_ao = P.tv_out = P.tiv_out = P.tv_weff = P.tv_beff = nullptr; P.tv_wbf = nullptr;
    P.tv_stats = P.tiv_stats = nullptr;
    P.tv_kp = P.tv_vtp = nullptr; P.tv_nkpad = 0;
    P.tv_G = P.tv_Vp = P.tv_g0 = P.tv_v0p = P.tv_xmean = nullptr;
    P.tiv_aff = nullptr;
    if (c.variant == DEX_VARIANT_DEX) {
        const size_t pm = (size_t)B * P.Hm * P.Wm;
        P.tv_keys = A.f((size_t)B * d.Ts * mid); P.tv_K = A.f((size_t)B * (d.Ts + 1) * mid); P.tv_V = A.f((size_t)B * (d.Ts + 1) * mid);
        P.tv_q = A.f(pm * mid); P.tv_ao = A.f(pm * mid); P.tv_out = A.f(pm * mid); P.tiv_out = A.f(pm * mid);
        P.tv_weff = A.f((size_t)B * mid * mid); P.tv_beff = A.f((size_t)B * mid);
        P.tv_wbf = A.take((size_t)B * mid * mid * 2 * 2);          // (x2: the lo halves of the split-weight mode behind the hi ones)
        P.tv_nkpad = (d.Ts + 1 + 63) / 64 * 64;
        P.tiv_aff = A.f((size_t)B * 2 * mid);
        P.tv_kp = A.take((size_t)B * P.tv_nkpad * mid * 2); P.tv_vtp = A.take((size_t)B * P.tv_nkpad * mid * 2);
        P.tv_stats = (gnfix_t*)A.take((size_t)B * mid * IN_SLOTS * 2 * 2 * sizeof(gnfix_t));   // IN2d partials of the TV input and the TIV input
        P.tiv_stats = A.dry ? nullptr : P.tv_stats + (size_t)B * mid * IN_SLOTS * 2;
        P.tv_G = A.f((size_t)B * (d.Ts + 1) * mid); P.tv_Vp = A.f((size_t)B * (d.Ts + 1) * mid);
        P.tv_g0 = A.f((size_t)n * mid); P.tv_v0p = A.f((size_t)n * mid); P.tv_xmean = A.f((size_t)B * mid);
    }
    P.bytes = (A.off + 255) & ~size_t(255);
}

struct Runner {
    DexCtx* x; const Plan& P; hipStream_t st;
    const float* mask; const float* mu; const float* xcur;
    const DexSampleArgs* args;
    bool debug;
    int gn_idx = 0;
    int sp = 0;                     // index of the current network evaluation (a by-value launch argument of every kernel)
    gnfix_t* stats_base = nullptr;  // statistics arena of this step
    gnfix_t* stats_other = nullptr;   // arena to clear for the next step (eager mode), or null
    int fin_mode = 0;               // FinalP::mode of this network evaluation (Heun predictor / corrector)
    const float* fin_htab = nullptr;

    template <typename F> void run(const char* name, double flops, double bytes, F&& f) {
        if (x->prof_on) {
            Prof pr; pr.name = name; pr.flops = flops; pr.bytes = bytes;
            hipEventCreate(&pr.a); hipEventCreate(&pr.b);
            g_last_symbol = nullptr;
            hipEventRecord(pr.a, st); f(); hipEventRecord(pr.b, st);
            if (g_last_symbol) pr.name = g_last_symbol;        // the instantiation the launcher picked (rocprofv3's name for it)
            x->prof.push_back(pr);
        } else f();
    }
    gnfix_t* next_stats() { return stats_base + (size_t)(gn_idx++) * P.d.B * 8 * GN_SLOTS * 2; }
    void tap(const char* name, const float* p, long rows, int C, int ld) {
        if (!debug) return;
        DexCtx::Tap t; t.name = name; t.p = p; t.shape = {rows, C, ld};
        for (auto& o : x->taps) if (o.name == t.name) { o = t; return; }
        x->taps.push_back(t);
    }

    IGemmP base_gemm(const float* A, int lda, int acoff, int H, int W, int Cin, const float* Wt, int N, const float* bias,
                     float* C, int ldc, int ccoff) {
        IGemmP g{};
        g.A = A; g.lda = lda; g.a_bstride = (long)H * W * lda; g.a_coff = acoff;
        g.Hi = H; g.Wi = W; g.Cin = Cin;
        g.KH = 1; g.KW = 1; g.sh = 1; g.sw = 1; g.off_h = 0; g.off_w = 0; g.step_h = 1; g.step_w = 1;
        g.Ho = H; g.Wo = W;
        g.W = Wt; g.w_bstride = 0; g.w_gstride = 0;
        { auto it = x->lp_of().find(Wt); g.Wbf = (it != x->lp_of().end()) ? it->second : nullptr; }
        g.N = N; g.K = Cin; g.ksplit = 1; g.groups = 1;
        g.bias = bias; g.bias_bstride = 0;
        g.C = C; g.ldc = ldc; g.c_bstride = (long)H * W * ldc; g.c_sstride = 0; g.c_coff = ccoff;
        g.OHf = H; g.OWf = W; g.osh = 1; g.osw = 1; g.oh0 = 0; g.ow0 = 0;
        g.inmask = nullptr; g.inmask_ws = 1; g.outmask = nullptr; g.outmask_ws = 1; g.mask_bstride = P.d.T;
        g.act = 0; g.gate = nullptr; g.gate_nstride = 1; g.gate_step_stride = 0;
        g.res = nullptr; g.ldres = 0; g.res_bstride = 0; g.res_coff = 0;
        g.step = sp; g.unpatch_s = 0; g.unpatch_C = 0; g.B = P.d.B;
        return g;
    }
    void gemm(const char* name, const IGemmP& g_in) {
        IGemmP g = g_in;
        if (!g.w_lo_off) g.w_lo_off = g.Wbf ? x->lo_off(g.Wbf) : 0;      // split-weight mode: where the lo pack of this twin sits (0: a plain 16-bit operand; set by the caller for operands built at run time)
        const double M = (double)g.Ho * g.Wo * g.B;
        const double fl = 2.0 * M * g.N * g.groups * g.K;
        const double by = 4.0 * (M * g.Cin * g.groups + M * g.N * g.groups * g.ksplit + (double)g.K * g.N * g.groups);
        run(name, fl, by, [&] { launch_igemm(g, x->precision, st); });
    }
    // Block conv (diffusion.py:44).  pro != null fuses the producer's GN-apply + Mish + time bias into the load
    // (bf16 patch kernel only).
    struct Pro { const gnfix_t* stats; const float *gamma, *beta, *tadd; const float* res = nullptr; float* xout = nullptr; bool x_bf16 = false;
                 bool res2 = false; FirstConvP res2f{};   // res2: `res` is not stored - recomputed from the first conv's inputs (res2f)
                 mutable bool xout_lp_ok = false;    // in: xout's one reader (the attention's context pass) can take it in the mode's 16-bit type
                 mutable bool xout_lp = false; };   // out: the conv that wrote xout did store it that way
    // raw conv outputs read only by a fused GroupNorm prologue (h1, h2 of a ResnetBlock, the final block's conv) are kept
    // in HBM as bf16 in bf16 mode: half the bytes of the largest tensors of the step.  DEX_H_BF16=0 keeps them fp32.
    static bool ln_fusable(int K) { return K == 64 || K == 128 || K == 256 || K == 512; }
    bool h_bf16() const { return x->lp() && !knob_off("DEX_H_BF16"); }
    bool fast_conv(int cin, int cout) const { return x->lp() && conv3x3_bf16_supported(cin, cout); }
    void conv3x3(const char* name, const TD& X, int H, int W, int mask_ws, bool inmask, const float* Wt, const float* bias, int Cout, float* out,
                 gnfix_t* gn = nullptr, const Pro* pro = nullptr, const ResW* shortcut = nullptr, float* shortcut_out = nullptr,
                 bool xb = false, bool yb = false) {
        auto it = x->lp_of().find(Wt);
        if (fast_conv(X.C, Cout) && it != x->lp_of().end()) {
            Conv3P c{};
            c.x_bf16 = (xb || X.lp) ? 1 : 0; c.y_bf16 = yb ? 1 : 0;
            c.X = X.p; c.ldx = X.ld; c.x_coff = X.coff; c.H = H; c.W = W; c.Cin = X.C; c.Cout = Cout;
            c.Wbf = it->second; c.w_lo_off = x->lo_off(c.Wbf); c.bias = bias; c.Y = out; c.mask = mask; c.mask_ws = mask_ws; c.mask_bstride = P.d.T;
            if (pro) { c.pro_stats = pro->stats; c.pro_gamma = pro->gamma; c.pro_beta = pro->beta; c.pro_tadd = pro->tadd; c.pro_res = pro->res; c.pro_xout = pro->xout;
                if (pro->res2) {
                    const FirstConvP& f = pro->res2f;
                    c.pro_res = nullptr; c.res2_w = f.W1; c.res2_b = f.b1; c.res2_mu = f.mu; c.res2_x = f.x; c.res2_spk = f.spk;
                    c.res2_scal = f.scal; c.res2_scal_stride = f.scal_stride; c.res2_planes = f.planes;
                } }
            c.step = sp; c.gn_stats = gn; c.B = P.d.B;
            { auto itf = x->frag_of().find(Wt); c.Wfrag = itf != x->frag_of().end() ? itf->second : nullptr; }   // conv3x3_regw.hip
            const bool want_xout_lp = pro && pro->xout && pro->xout_lp_ok;
            if (shortcut) {       // the block's 1x1 res_conv rides on the centre tap of this conv
                c.res_w = x->lp_of().at(shortcut->wr); c.res_lo_off = x->lo_off(c.res_w); c.res_b = shortcut->br; c.res_y = shortcut_out;
                { auto itf = x->frag_of().find(shortcut->wr); c.res_wfrag = itf != x->frag_of().end() ? itf->second : nullptr; }
            }
            if (want_xout_lp && (c.pro_res || c.res2_w) && conv3x3_strip_form(c)) { c.xout_lp = 1; pro->xout_lp = true; }
            const double M = (double)H * W * P.d.B;
            // algorithmic bytes: input + output at their stored width, + the residual read and the x write-out of the PRO2 form,
            // + the shortcut output of the RES form, + the weights once
            const double bytes = M * (((xb || X.lp) ? 2.0 : 4.0) * X.C + (yb ? 2.0 : 4.0) * Cout + (pro && (pro->res || pro->res2) ? (pro->res2 ? 4.0 * X.C : 8.0 * X.C) - (c.xout_lp ? 2.0 * X.C : 0.0) : 0.0) + (shortcut ? 4.0 * Cout : 0.0))
                                 + 2.0 * 9 * X.C * Cout;
            run(name, 2.0 * M * Cout * (9 * X.C + (shortcut ? X.C : 0)), bytes, [&] { launch_conv3x3_lp(c, x->precision, st); });
            return;
        }
        IGemmP g = base_gemm(X.p, X.ld, X.coff, H, W, X.C, Wt, Cout, bias, out, Cout, 0);
        g.KH = 3; g.KW = 3; g.off_h = -1; g.off_w = -1; g.K = 9 * X.C;
        if (inmask) { g.inmask = mask; g.inmask_ws = mask_ws; }
        g.gn_stats = gn; g.gn_groups = 8; g.gn_cpg = Cout / 8;
        gemm(name, g);
    }
    void gn_stats(const float* h, int C, long npix, gnfix_t* stats) {
        GnStatsP s{h, C, npix * C, (int)npix, C, 8, stats, P.d.B};
        run("gn_stats", 3.0 * npix * C * P.d.B, 4.0 * npix * C * P.d.B, [&] { launch_gn_stats(s, st); });
    }
    void gn_apply(const float* h, int C, long npix, int W, int mask_ws, const gnfix_t* stats, const float* gamma, const float* beta,
                  const float* tadd, const float* res, int ldres, long resb, bool res_under_mask, float* out) {
        GnApplyP a{};
        a.X = h; a.ldx = C; a.xb = npix * C; a.Y = out; a.ldy = C; a.yb = npix * C; a.y_coff = 0;
        a.npix = (int)npix; a.W = W; a.C = C; a.groups = 8; a.stats = stats; a.gamma = gamma; a.beta = beta;
        a.mask = mask; a.mask_ws = mask_ws; a.mask_bstride = P.d.T;
        a.tadd = tadd; a.tadd_step_stride = C; a.step = sp;
        a.res = res; a.ldres = ldres; a.resb = resb; a.res_under_mask = res_under_mask ? 1 : 0; a.B = P.d.B;
        run("gn_apply_mish", 12.0 * npix * C * P.d.B, (res ? 12.0 : 8.0) * npix * C * P.d.B, [&] { launch_gn_apply(a, st); });
    }

    // ResnetBlock (diffusion.py:66-71).  X: unmasked input view; out = block2(...) + res_conv(x*mask) (unmasked).
    // tail != null: the final GN-apply + Mish + shortcut is NOT launched; its operands are recorded in *tail and the
    // consumer (linattn_kvctx) applies it while loading and writes `out`.
    // head != null: X (= the previous ResnetBlock's output) has NOT been materialised; *head describes its tail
    // (raw conv output + GN + res_conv shortcut) and this block's first conv applies it while staging and writes X.
    // ctail != null: same deferral for THIS block's output towards the next block's first conv.
    void resblock(const ResW& w, const StageBuf& s, const TD& X, const float* tadd, float* out, bool first_layer, LinKvCtxP* tail = nullptr,
                  const Pro* head = nullptr, Pro* ctail = nullptr) {
        const long npix = s.npix;
        const float* resptr; int ldres; long resb; bool under = false;
        gnfix_t* st1 = nullptr;
        // h1 / h2 of this block as bf16: conv2 must be the GroupNorm-prologue conv that can read bf16, conv1 a kernel that can
        // write it; h2 additionally needs a fused consumer (the next block's first conv or the attention's context pass)
        const bool conv2_fast = fast_conv(w.cout, w.cout) && conv3x3_bf16_xb_supported(w.cout, w.cout) && x->lp_of().count(w.w2);
        const bool conv1_fast = first_layer || (head ? true : (fast_conv(X.C, w.cout) && x->lp_of().count(w.w1)));
        const bool h1b = h_bf16() && conv2_fast && conv1_fast;
        const bool h2b = h_bf16() && conv2_fast && (ctail || tail);
        if (first_layer) {
            FirstConvP f{};
            f.h1_bf16 = h1b ? x->lp_kind() : 0;
            f.mu = mu; f.x = xcur; f.spk = P.spk_plane; f.mask = mask; f.B = P.d.B; f.H = s.H; f.T = s.W; f.planes = w.cin; f.C = w.cout;
            f.W3 = w.w1; f.b3 = w.b1; f.W1 = w.wr; f.b1 = w.br; f.scal = P.scal; f.scal_stride = SCAL_STRIDE; f.step = sp;
            f.h1 = s.h1; f.res = s.rbuf;
            // the 1x1 shortcut of this block is two or three FMAs per value from the input planes: when the consumer is the next
            // block's fused-tail convolution it recomputes it, and the fp32 [B,H,T,64] tensor is neither written nor read
            // (DEX_RES2=0 stores it as before)
            { if (ctail && h2b && w.cout == 64 && conv3x3_res2_form(s.H, s.W, P.d.B) && !knob_off("DEX_RES2")) { f.res = nullptr; ctail->res2 = true; ctail->res2f = f; } }
            st1 = next_stats(); f.gn_stats = st1;
            run("first_conv", 2.0 * npix * P.d.B * w.cout * (w.cin * 10), npix * P.d.B * ((h1b ? 2.0 : 4.0) * w.cout + (f.res ? 4.0 * w.cout : 0.0) + 4.0 * w.cin), [&] { launch_first_conv(f, st); });
            resptr = s.rbuf; ldres = w.cout; resb = npix * w.cout;
        } else {
            st1 = next_stats();
            bool fused_res = false;
            if (head) {
                TD H2{s.h2, X.C, 0, X.C};                  // previous block's raw conv2 output
                // the previous block's output x (head->xout == X) is written by this conv for ONE later reader when this block's own
                // tail rides in the attention's context pass with the identity shortcut: it may then leave in the mode's 16-bit type
                // (the residual term of x' = Mish(GN(h2)) + x carries that rounding: not bit-neutral, DEX_RES_X_LP=0 keeps fp32)
                head->xout_lp_ok = tail != nullptr && !w.wr && lp_inter_cur && head->xout == X.p && X.coff == 0 && X.ld == X.C && !knob_off("DEX_RES_X_LP");
                conv3x3("conv3x3", H2, s.H, s.W, s.mask_ws, true, w.w1, w.b1, w.cout, s.h1, st1, head, nullptr, nullptr, head->x_bf16, h1b);
            } else {
                fused_res = w.wr && fast_conv(X.C, w.cout) && conv3x3_bf16_res_supported(X.C, w.cout) && x->lp_of().count(w.wr);
                conv3x3("conv3x3", X, s.H, s.W, s.mask_ws, true, w.w1, w.b1, w.cout, s.h1, st1, nullptr, fused_res ? &w : nullptr, s.rbuf, false, h1b);
            }
            if (fused_res) {
                resptr = s.rbuf; ldres = w.cout; resb = npix * w.cout;
            } else if (w.wr) {
                IGemmP g = base_gemm(X.p, X.ld, X.coff, s.H, s.W, X.C, w.wr, w.cout, w.br, s.rbuf, w.cout, 0);
                g.inmask = mask; g.inmask_ws = s.mask_ws;
                gemm("conv1x1_res", g);
                resptr = s.rbuf; ldres = w.cout; resb = npix * w.cout;
            } else {
                resptr = X.p + X.coff; ldres = X.ld; resb = npix * X.ld; under = true;
            }
        }
        if (!st1) { st1 = next_stats(); gn_stats(s.h1, w.cout, npix, st1); }
        gnfix_t* st2 = next_stats();
        if (fast_conv(w.cout, w.cout)) {
            // block1's GN-apply + Mish + time bias + mask is applied while block2's conv stages its input patch
            Pro pro{st1, w.g1, w.be1, tadd};
            TD H1{s.h1, w.cout, 0, w.cout};
            conv3x3("conv3x3", H1, s.H, s.W, s.mask_ws, true, w.w2, w.b2, w.cout, s.h2, st2, &pro, nullptr, nullptr, h1b, h2b);
        } else {
            gn_apply(s.h1, w.cout, npix, s.W, s.mask_ws, st1, w.g1, w.be1, tadd, nullptr, 0, 0, false, s.a1);
            TD A1{s.a1, w.cout, 0, w.cout};
            conv3x3("conv3x3", A1, s.H, s.W, s.mask_ws, false, w.w2, w.b2, w.cout, s.h2, st2);
        }
        if (ctail) {       // res_conv shortcut only (resptr is a dense [npix][cout] buffer, added outside the mask)
            ctail->stats = st2; ctail->gamma = w.g2; ctail->beta = w.be2; ctail->tadd = nullptr; ctail->res = resptr; ctail->xout = out;
            ctail->x_bf16 = h2b;
            return;
        }
        if (tail) {
            tail->H2 = s.h2; tail->gn_stats = st2; tail->gamma = w.g2; tail->beta = w.be2;
            tail->res = resptr; tail->ldres = ldres; tail->resb = resb; tail->res_under_mask = under ? 1 : 0;
            tail->res_lp = (head && head->xout_lp && resptr == X.p) ? 1 : 0;
            tail->mask = mask; tail->mask_ws = s.mask_ws; tail->mask_bstride = P.d.T; tail->W = s.W; tail->Xout = out;
            tail->h2_bf16 = h2b ? 1 : 0;
            return;
        }
        gn_apply(s.h2, w.cout, npix, s.W, s.mask_ws, st2, w.g2, w.be2, nullptr, resptr, ldres, resb, under, out);
    }

    // Residual(Rezero(LinearAttention)) (diffusion.py:74-102)
    bool lp_inter_cur = false;             // this step stores single-consumer activations in the mode's 16-bit type (step())
    // The up path's concatenation buffer [skip | DiT output] has one reader that matters for bytes, the 2C -> C conv with its fused
    // shortcut, which rounds x * mask to the operand type: at batch size (n_stages == 2) the attention tail writes a 16-bit copy of the
    // skip half next to its fp32 output (the DiT front-end reads that one) and the unpatchify GEMM writes its half in 16 bit only -
    // same bits, 0.25 GB less traffic per Euler step at GeDEX B = 32.  DEX_CAT_LP=0 keeps the fp32 buffer.
    bool cat_lp = false;
    // FinalLayer + unpatchify GEMM of the DiT (dit.py:97-110, 457-461)
    IGemmP final_gemm(int mask_ws, float* out, int ldo, int ocoff) {
        const DexConfig& c = x->cfg;
        const int hid = c.dit_hidden, mid = mid_dim(c);
        const bool fuse_lnf = x->lp() && ln_fusable(hid);
        const int s2c = c.dit_stride * c.dit_stride * mid;
        IGemmP fl = base_gemm(fuse_lnf ? P.tok : P.xn, hid, 0, P.Hf, P.Wt, hid, x->fl_w, s2c, x->fl_b, out, ldo, ocoff);
        if (fuse_lnf) { fl.ln_shift = P.fin_mod; fl.ln_scale = P.fin_mod + hid; fl.ln_step_stride = 2L * hid; }
        { auto it = x->frag_of().find(x->fl_w); fl.Wfrag = (it != x->frag_of().end()) ? it->second : nullptr; }
        fl.unpatch_s = c.dit_stride; fl.unpatch_C = mid; fl.OHf = P.Hm; fl.OWf = P.Wm;
        fl.c_bstride = (long)P.Hm * P.Wm * ldo;
        fl.outmask = mask; fl.outmask_ws = mask_ws;
        return fl;
    }
    bool linattn_fused(int C) const { return x->lp() && (C == 64 || C == 128) && linattn_fused_supported(C); }
    void linattn(const LinW& w, const StageBuf& s, const TD& X, float* out, int ldo, int ocoff, const LinKvCtxP* tail = nullptr, bool out_lp = false,
                 void* out2_lp = nullptr, int ldo2 = 0, int ocoff2 = 0) {
        const long npix = s.npix; const int B = P.d.B;
        if (linattn_fused(X.C)) {
            // fused: y = x + W2 (Wq x) + g*b  with  W2 = g Wout blockdiag(ctx^T)   (linattn_fused.hip)
            int nsub = 1;
            // measured at 80x512, B=1: 320 / 160 / 80 workgroups -> context 19.9 / 13.7 / 19.4 us, merge 8.7 / 6.1 / 4.7 us
            while (nsub < 4 && (npix + 128 * nsub - 1) / (128 * nsub) * B > 192) nsub *= 2;
            {
                // Batch regime (round 5): the grid is several rounds of the chip's workgroup slots - two per CU at C = 64, ONE at C = 128
                // (the context pass holds 256 + 73 registers there) - and powers of two land between rounds: DEX B = 32 ran 1 280 workgroups
                // on 512 slots (2.5 rounds) at full resolution and 320 on 256 (1.25) at half.  Choose the sub-tile count that minimises
                // rounds x sub-tiles; among equals the largest (fewer partials for the merge, fewer weight stagings).  DEX_LINATTN_NSUB forces one.
                const int ncu = device_cus();
                const long slots = (long)(X.C == 64 ? 2 : 1) * ncu;
                if ((npix + 127) / 128 * B >= 2 * slots) {
                    long best = -1;
                    for (int n = 1; n <= 8; ++n) {
                        const long wgs = (npix + 128L * n - 1) / (128L * n) * B;
                        const long cost = (wgs + slots - 1) / slots * n;
                        if (best < 0 || cost <= best) { best = cost; nsub = n; }
                    }
                }
                const int forced = knob_or("DEX_LINATTN_NSUB", 0);
                if (forced > 0) nsub = forced;
            }
            const int nblk = (int)((npix + 128L * nsub - 1) / (128L * nsub));
            LinKvCtxP k{};
            if (tail) k = *tail;
            k.X = X.p; k.ldx = X.ld; k.x_coff = X.coff; k.xb = npix * X.ld; k.npix = (int)npix; k.C = X.C; k.Wkv = w.wkv_lp[x->lpi()];
            k.wkv_lo_off = x->lpi() == 2 ? 384L * X.C : 0;          // (the q | k | v rows were converted as one array: Packer::linattn)
            k.nsub = nsub; k.nblk = nblk; k.part_m = s.pm; k.part_s = s.ps; k.part_c = s.pc; k.B = B;
            // x = the ResnetBlock output the context pass materialises has ONE reader, the tail kernel below, which rounds it to the
            // operand type for its q GEMM anyway and adds it back as the residual term: at batch size it is stored in that type
            // (half of the write + read; the residual term then carries the operand rounding - NOT bit-identical to the fp32
            // store, measured in DESIGN.md; DEX_ATTN_X_LP=0 keeps it fp32)
            const bool xlp = k.H2 && k.Xout && lp_inter_cur && linattn_out2_lp_out_supported((int)npix, B) && !knob_off("DEX_ATTN_X_LP");
            k.xout_lp = xlp ? 1 : 0;
            run("linattn_kvctx", 2.0 * npix * B * (256.0 * X.C + 128 * 32), (tail ? (tail->h2_bf16 ? 10.0 : 12.0) - (xlp ? 2.0 : 0.0) : 4.0) * npix * X.C * B, [&] { launch_linattn_kvctx(k, x->precision, st); });
            LinMergeP mg{s.pm, s.ps, s.pc, nblk, w.wout_raw, w.g, X.C, s.mbf, B};
            run("linattn_merge", 2.0 * 128 * 32 * X.C * B, 4.0 * nblk * 4 * 1088 * B, [&] { launch_linattn_merge(mg, x->precision, st); });
            LinOut2P o{X.p, X.ld, X.coff, npix * X.ld, (int)npix, X.C, w.wq_frag[x->lpi()], s.mbf, w.bias_eff, out, ldo, ocoff, npix * ldo, B, out_lp ? 1 : 0,
                       out2_lp, ldo2, ocoff2, npix * ldo2, xlp ? 1 : 0};
            o.wq_lo_off = x->lpi() == 2 ? 128L * X.C : 0;
            run("linattn_out", 4.0 * npix * B * 128.0 * X.C, ((out_lp ? 6.0 : 8.0) - (xlp ? 2.0 : 0.0) + (out2_lp ? 2.0 : 0.0)) * npix * X.C * B, [&] { launch_linattn_out2(o, x->precision, st); });
            return;
        }
        IGemmP g = base_gemm(X.p, X.ld, X.coff, s.H, s.W, X.C, w.wqkv, 384, nullptr, s.qkv, 384, 0);
        gemm("linattn_qkv", g);
        LinAttnCtxP cp{s.qkv, 384, npix * 384, (int)npix, 4, LA_CHUNK, s.nchunks, s.pm, s.ps, s.pc, B};
        run("linattn_ctx", 2.0 * npix * 128 * 32 * B, 4.0 * npix * 384 * B, [&] { launch_linattn_ctx(cp, st); });
        LinAttnCombineP cb{s.pm, s.ps, s.pc, s.nchunks, 4, w.wout_raw, w.g, w.C, s.weff, B};
        run("linattn_combine", 2.0 * 128 * 32 * w.C * B, 4.0 * s.nchunks * 4 * 1100 * B, [&] { launch_linattn_combine(cb, st); });
        IGemmP o = base_gemm(s.qkv, 384, 0, s.H, s.W, 128, s.weff, w.C, w.bias_eff, out, ldo, ocoff);
        o.w_bstride = 128L * w.C;
        o.res = X.p; o.ldres = X.ld; o.res_coff = X.coff; o.res_bstride = npix * X.ld;
        gemm("linattn_out", o);
    }

    // DiTMask.forward (dit.py:485-525): X = bottleneck input view; writes [B,Hm,Wm,mid] into (out, ldo, ocoff)
    void dit(const TD& X, bool mask_input, int mask_ws, float* out, int ldo, int ocoff, const float* in_aff = nullptr) {
        const DexConfig& c = x->cfg;
        const int B = P.d.B, hid = c.dit_hidden, mid = mid_dim(c), N = P.N, mh = mlp_hidden(c);
        DwConvP dw{};
        dw.X = X.p + X.coff; dw.ldx = X.ld; dw.xb = (long)P.Hm * P.Wm * X.ld; dw.Hi = P.Hm; dw.Wi = P.Wm; dw.C = mid;
        dw.k = c.dit_patch; dw.s = c.dit_stride; dw.pad = c.dit_patch / 2; dw.Wd = x->pe_dw; dw.bd = x->pe_db;
        dw.mask = mask_input ? mask : nullptr; dw.mask_ws = mask_ws; dw.mask_bstride = P.d.T;
        dw.Y = P.pe0; dw.Hf = P.Hf; dw.Wt = P.Wt; dw.B = B; dw.aff = in_aff;
        auto pw_lp = x->lp_of().find(x->pe_pw);
        if (patch_fused()) {
            // small grids: depthwise conv + SiLU + pointwise GEMM in ONE launch (bit-identical to the two-kernel form below)
            run("patch_embed", 2.0 * B * N * mid * (c.dit_patch * c.dit_patch + hid), 4.0 * B * (P.Hm * P.Wm * mid + N * hid),
                [&] { launch_patch_embed_fused(dw, pw_lp->second, x->pe_pb, P.emb, hid, x->precision, st); });
        } else {
            run("patch_dwconv_silu", 2.0 * B * N * mid * c.dit_patch * c.dit_patch, 4.0 * B * (P.Hm * P.Wm + N) * mid, [&] { launch_dwconv_silu(dw, st); });
            IGemmP pe = base_gemm(P.pe0, mid, 0, P.Hf, P.Wt, mid, x->pe_pw, hid, x->pe_pb, P.emb, hid, 0);
            gemm("patch_pointwise", pe);
        }
        // grouped 16x16 pos-conv, split-K partials (bias added in the tail)
        const int G = c.dit_conv_pos_groups, kp = c.dit_conv_pos, cg = hid / G;
        int nsplit = POS_SPLIT, pcg = 0, pcgp = 0;
        if (x->lp() && x->pos_wfrag[x->lpi()]) {
            PosConvP pcd{P.emb, x->pos_wfrag[x->lpi()], P.pos_part, P.Hf, P.Wt, hid, G, B};
            run("pos_conv", 2.0 * B * N * (double)kp * kp * cg * hid, 8.0 * B * N * hid + 2.0 * kp * kp * cg * hid, [&] { launch_pos_conv_direct(pcd, x->precision, st); });
            nsplit = 1;
        } else {
            const int cgp = (cg % 32) ? 64 : cg;           // groups padded to the implicit GEMM's 32-channel granule (zero weights / inputs)
            const float* src = P.emb;
            if (cgp != cg) {
                run("pos_group_pad", 0, 8.0 * B * N * hid, [&] { launch_group_pad(P.emb, P.emb_pad, (long)B * N, G, cg, cgp, st); });
                src = P.emb_pad;
            }
            IGemmP pc = base_gemm(src, G * cgp, 0, P.Hf, P.Wt, cgp, x->pos_w, cgp, nullptr, P.pos_part, G * cgp, 0);
            pc.KH = kp; pc.KW = kp; pc.off_h = -(kp / 2); pc.off_w = -(kp / 2); pc.K = kp * kp * cgp;
            pc.groups = G; pc.w_gstride = (long)kp * kp * cgp * cgp; pc.ksplit = POS_SPLIT; pc.c_sstride = (long)B * N * G * cgp;
            if (cgp != cg) pc.Wbf = nullptr;
            gemm("pos_conv", pc);
            pcg = cgp != cg ? cg : 0; pcgp = cgp;
        }
        PosFinishP pf{P.pos_part, nsplit, (long)B * N * (pcg ? G * pcgp : hid), x->pos_b, P.emb, x->freq_pos, P.tok, P.Hf, P.Wt, hid, B, pcg, pcgp};
        run("pos_finish", 20.0 * B * N * hid, 4.0 * B * N * hid * (nsplit + 2), [&] { launch_pos_finish(pf, st); });
        if (debug) hipMemcpyAsync(P.dbg_tok, P.tok, (size_t)B * N * hid * 4, hipMemcpyDeviceToDevice, st);
        tap("tok_in", P.dbg_tok, (long)B * N, hid, hid);
        const float scale = 1.0f / sqrtf((float)(hid / c.dit_heads));
        const bool chain = x->lp() && dit_rowchain_supported(hid, mh) && c.dit_heads == 2 && x->frag_of().count(x->blocks[0].wproj) &&
                           !knob_off("DEX_DIT_CHAIN");          // 0: one GEMM / attention launch per operation (A/B runs)
        for (int k = 0; k < c.dit_depth; ++k) {
            const DitBlockW& w = x->blocks[k];
            const float* ada = P.ada[k];
            const bool fuse_ln = x->lp() && ln_fusable(hid);      // LayerNorm+modulate inside the GEMM's A staging (single-shot K: 64 / 128 / 256 / 512)
            DitChainP ch{};
            if (chain) {
                ch.ksplit = 0; ch.heads = c.dit_heads; ch.rows_per_batch = N; ch.X = P.tok; ch.ada = ada; ch.step = sp; ch.M = B * N; ch.B = B;
                // operand set k & 1 is read by block k (separate attention kernel or in-kernel attention), the other is written
                ch.Qh = (k & 1) ? P.qh : P.qh2; ch.Kh = (k & 1) ? P.kh : P.kh2; ch.Vt = (k & 1) ? P.vt : P.vt2;
                ch.Qin = (k & 1) ? P.qh2 : P.qh; ch.Kin = (k & 1) ? P.kh2 : P.kh; ch.Vin = (k & 1) ? P.vt2 : P.vt;
                ch.Npad = P.Npad;
                ch.qscale = scale * 1.4426950408889634f;      // log2(e) folded in: the attention kernels use exp2
                if (P.xflag) {                                // cluster form: slabs, flags, an epoch unique within the call (never 0)
                    ch.xslab = P.xslab; ch.xflag = P.xflag; ch.epoch = (unsigned)(sp * (c.dit_depth + 1) + k + 1);
                    ch.xerr = reinterpret_cast<int*>(P.xflag + (P.xflag_bytes - sizeof(int)) / sizeof(unsigned));
                    x->last_xerr = ch.xerr;
                    ch.xdrop = knob_or("DEX_DEBUG_DROP_HANDOFF", 0);
                    ch.xlocal = (P.xlocal && ch.epoch < (1u << 24)) ? 1 : 0;
                    ch.xcds = dit_rowchain_cluster_xcds(N, B);
                }
            }
            if (chain && k == 0) {                                   // first block: LN + modulate + qkv only
                ch.qkv_only = 1; ch.Wq = x->frag_of().at(w.wqkv); ch.bq = w.bqkv;
                ch.Qh = P.qh; ch.Kh = P.kh; ch.Vt = P.vt;        // block 0 reads set 0
                ch.next_shift = ada; ch.next_scale = ada + hid; ch.next_step_stride = 6L * hid;
                run("dit_qkv", 2.0 * B * N * 3.0 * hid * hid, 4.0 * B * N * hid + 2.0 * B * N * 3 * hid, [&] { launch_dit_rowchain(ch, x->precision, st); });
                ch.qkv_only = 0; ch.Wq = nullptr; ch.bq = nullptr; ch.next_shift = ch.next_scale = nullptr;
                ch.Qh = P.qh2; ch.Kh = P.kh2; ch.Vt = P.vt2;     // ... and writes set 1
            }
            if (!chain) {
                IGemmP q = base_gemm(fuse_ln ? P.tok : P.xn, hid, 0, 1, N, hid, w.wqkv, 3 * hid, w.bqkv, P.qkv, 3 * hid, 0);
                if (fuse_ln) { q.ln_shift = ada + 0 * hid; q.ln_scale = ada + 1 * hid; q.ln_step_stride = 6L * hid; }
                else {
                    LnModP l1{P.tok, P.xn, N, hid, ada + 0 * hid, ada + 1 * hid, 6L * hid, sp, B};
                    run("ln_modulate", 8.0 * B * N * hid, 8.0 * B * N * hid, [&] { launch_ln_mod(l1, st); });
                }
                gemm("dit_qkv", q);
            }
            if (chain) {
                // few workgroups (small batch): split the keys over up to 4 workgroups per query tile so a wave sees
                // one or two key tiles (one global round trip); the row-chain kernel merges the partials on load
                // The attention core runs inside the row-chain launch (dit_rowchain_kernel<true>): every workgroup computes
                // the attention of its own 32 queries for both heads, so no attention launch and no partials in HBM.
                // DEX_ATTN_SEPARATE=1 restores the separate kernels (attention_direct.hip) for A/B measurements.
                // Large grids (batched synthesis: >= 1024 (query tile, head) items) run the attention as its OWN launch on the
                // shared-ring kernel (attention_direct.hip): K / V^T tiles are read from L2 once per 128 queries instead of
                // once per 32, 0.25 vs 0.20 of the MFMA peak.  DEX_ATTN_SEPARATE=0 / 1 forces either form (A/B, profiling).
                const int sep_env = knob("DEX_ATTN_SEPARATE");          // (bench.py flips it for one profiling pass)
                const bool batch_regime = attention_direct_batch_regime(N, B);
                // (measured end to end: DEX B=32 N=1300 +0.6 %, GeDEX B=32 N=650 -0.9 % — short key loops gain nothing from the
                // rings and pay for the extra launch and the fp32 O round trip, so the automatic switch wants N >= 1024 too)
                // Round 4: the 64-queries-per-wave form (attention_q64.hip: 4 waves x 64 queries, persistent units, generated instruction
                // streams) takes the separate launch wherever it can fill the chip - batched DEX (N = 1300: 0.31 of the MFMA peak against
                // 0.28 for the shared-ring kernel) and long-form synthesis (N = 5010, one utterance: 0.33 against 0.19 fused into the block).
                // DEX_ATTN_Q64=0 / 1 forces either (A/B, tests).
                const int q64_env = knob("DEX_ATTN_Q64");
                const bool q64_ok = q64_env != KNOB_UNSET ? q64_env != 0 : attention_q64_regime(N, B);
                const bool separate = sep_env != KNOB_UNSET ? sep_env != 0 : ((batch_regime && N >= 1024) || q64_ok);
                const bool q64 = separate && q64_ok;
                int ks = 1, tail_row0 = 0, tail_ks = 0;
                bool o_lp = false;
                if (separate) {
                    const long blocks = (long)((N + 31) / 32) * 2 * B;
                    const int ntiles = (N + 31) / 32;
                    ks = q64 ? attention_q64_ksplit(N, B, att_split_cap((long)B * N))
                       : batch_regime ? attention_direct_ksplit(N, B)
                                      : (int)std::max<long>(1, std::min<long>(std::min<long>(768 / blocks, (ntiles + 7) / 8), ATT_KSPLIT_MAX));
                    AttnDirectP ad{ch.Qin, ch.Kin, ch.Vin, N, P.Npad, B, P.ao, (long)B * N * hid, ks > 1 ? P.att_ml : nullptr, ks, nullptr};
                    // O has one reader, the row chain's projection GEMM, which rounds it to the operand type: with one key split on the
                    // batch forms (shared-ring / 64-query attention -> 64-row chain) it is stored in that type - same bits, half the bytes
                    // (DEX_LP_INTER=0 keeps fp32)
                    o_lp = (batch_regime || q64) && ks == 1 && lp_inter_cur && dit_rowchain64_form(N, B, 0);
                    // (round 6 also built 16-bit PARTIAL slots for the long form's six key splits; a net loss - the 32-row chain's merge is
                    // bound by its load count, not its bytes: profiles/round6_attention_16bit_partials_negative.txt - and removed again)
                    ad.o_lp = o_lp ? 1 : 0;
                    // tail split of the 64-query form (attention_q64.hip): whole units for the first query groups, a key split for the
                    // last ones, so that a chip-filling round of long units is followed by a round of short ones; the 64-row chain merges
                    // the tail rows' partials (O slots 1.., fp32) and reads the other rows as before.  OPT-IN (DEX_ATTN_Q64_TAIL=1): measured at
                    // DEX B = 32, N = 1300 the attention launch gains 4.5 us (70.0 -> 65.5: fp32 partials + (m, l) instead of 16-bit rows eat most
                    // of the 13 us the unit plan predicts and tools/attnq64 measures on fp32 outputs) and the 64-row chain loses 6.2 us merging
                    // the tail rows (79.2 -> 85.4): -0.3 % end to end (profiles/round4_attention_tail_split_ab.txt).
                    if (q64 && ks == 1 && dit_rowchain64_form(N, B, 0) && knob_or("DEX_ATTN_Q64_TAIL", 0) != 0) {
                        int pks = 1, tg = 0, tk = 1;
                        attention_q64_plan(N, B, att_split_cap((long)B * N), &pks, &tg, &tk);
                        if (pks == 1 && tk > 1) { ad.tail_g = tg; ad.tail_ks = tk; ad.ml = P.att_ml; tail_row0 = tg * 256; tail_ks = tk; }
                    }
                    run("dit_attention", 4.0 * B * (double)N * N * hid, 2.0 * 3 * B * N * hid + (o_lp ? 2.0 : 4.0) * B * N * hid * ks,
                        [&] { if (q64) launch_attention_q64(ad, x->precision, st); else launch_attention_direct(ad, x->precision, st); });
                }
                ch.attn_inline = separate ? 0 : 1; ch.o_lp = o_lp ? 1 : 0; ch.tail_row0 = tail_row0; ch.tail_ks = tail_ks;
                const bool last = k + 1 == c.dit_depth;
                ch.O = P.ao; ch.ksplit = ks; ch.o_sstride = (long)B * N * hid; ch.ml = P.att_ml;
                ch.Wp = x->frag_of().at(w.wproj); ch.W1 = x->frag_of().at(w.wfc1); ch.W2 = x->frag_of().at(w.wfc2);
                ch.bp = w.bproj; ch.b1 = w.bfc1; ch.b2 = w.bfc2;
                if (!last) {
                    const DitBlockW& wn = x->blocks[k + 1];
                    ch.Wq = x->frag_of().at(wn.wqkv); ch.bq = wn.bqkv;
                    ch.next_shift = P.ada[k + 1]; ch.next_scale = P.ada[k + 1] + hid; ch.next_step_stride = 6L * hid;
                }
                const double M = (double)B * N;
                const double wel = (double)hid * hid + 2.0 * hid * mh + (last ? 0.0 : 3.0 * hid * hid);
                run(separate ? "dit_rowchain" : "dit_block", 2.0 * M * wel + (separate ? 0.0 : 4.0 * B * (double)N * N * hid),
                    4.0 * M * hid * (last ? 3 : 6) + 2.0 * wel, [&] { launch_dit_rowchain(ch, x->precision, st); });
                if (debug) {
                    float* dst = P.dbg_tok + (size_t)(k + 1) * B * N * hid;
                    hipMemcpyAsync(dst, P.tok, (size_t)B * N * hid * 4, hipMemcpyDeviceToDevice, st);
                    char nm[32]; snprintf(nm, sizeof nm, "tok_blk%d", k);
                    tap(nm, dst, (long)B * N, hid, hid);
                }
                continue;
            }
            AttnP a{};
            a.Q = P.qkv; a.ldq = 3 * hid; a.qb = (long)N * 3 * hid;
            a.K = P.qkv + hid; a.ldk = 3 * hid; a.kb = a.qb; a.V = P.qkv + 2 * hid; a.ldv = 3 * hid; a.vb = a.qb;
            a.O = P.ao; a.ldo = hid; a.ob = (long)N * hid; a.Nq = N; a.Nk = N; a.kv_len = nullptr; a.kv_len_add = 0;
            a.heads = c.dit_heads; a.scale = scale; a.B = B; a.head_dim = hid / c.dit_heads;
            a.force_generic = (knob_or("DEX_ATTN_GENERIC", 0) && x->precision == DEX_PREC_FP32) ? 1 : 0;   // tests
            run("dit_attention", 4.0 * B * (double)N * N * hid, 4.0 * 4 * B * N * hid, [&] { launch_attention(a, x->precision, st); });
            IGemmP pr = base_gemm(P.ao, hid, 0, 1, N, hid, w.wproj, hid, w.bproj, P.tok, hid, 0);
            pr.gate = ada + 2 * hid; pr.gate_nstride = 1; pr.gate_step_stride = 6L * hid;
            pr.res = P.tok; pr.ldres = hid; pr.res_bstride = (long)N * hid;
            gemm("dit_proj", pr);
            IGemmP f1 = base_gemm(fuse_ln ? P.tok : P.xn, hid, 0, 1, N, hid, w.wfc1, mh, w.bfc1, P.hmlp, mh, 0);
            if (fuse_ln) { f1.ln_shift = ada + 3 * hid; f1.ln_scale = ada + 4 * hid; f1.ln_step_stride = 6L * hid; }
            else {
                LnModP l2{P.tok, P.xn, N, hid, ada + 3 * hid, ada + 4 * hid, 6L * hid, sp, B};
                run("ln_modulate", 8.0 * B * N * hid, 8.0 * B * N * hid, [&] { launch_ln_mod(l2, st); });
            }
            f1.act = 1;
            gemm("dit_fc1_gelu", f1);
            IGemmP f2 = base_gemm(P.hmlp, mh, 0, 1, N, mh, w.wfc2, hid, w.bfc2, P.tok, hid, 0);
            f2.gate = ada + 5 * hid; f2.gate_nstride = 1; f2.gate_step_stride = 6L * hid;
            f2.res = P.tok; f2.ldres = hid; f2.res_bstride = (long)N * hid;
            gemm("dit_fc2", f2);
            if (debug) {
                float* dst = P.dbg_tok + (size_t)(k + 1) * B * N * hid;
                hipMemcpyAsync(dst, P.tok, (size_t)B * N * hid * 4, hipMemcpyDeviceToDevice, st);
                char nm[32]; snprintf(nm, sizeof nm, "tok_blk%d", k);
                tap(nm, dst, (long)B * N, hid, hid);
            }
        }
        const bool fuse_lnf = x->lp() && ln_fusable(hid);
        if (!fuse_lnf) {
            LnModP lf{P.tok, P.xn, N, hid, P.fin_mod, P.fin_mod + hid, 2L * hid, sp, B};
            run("ln_modulate", 8.0 * B * N * hid, 8.0 * B * N * hid, [&] { launch_ln_mod(lf, st); });
        }
        IGemmP fl = final_gemm(mask_ws, out, ldo, ocoff);
        if (cat_lp) { fl.C = reinterpret_cast<float*>(P.cat16); fl.c_lp = x->lp_kind(); }     // (same ld / offset, in 16-bit elements)
        gemm("dit_final_unpatchify", fl);
    }

    // PatchEmbed2D runs as one launch (patch_embed.hip; small grids)
    bool patch_fused() const {
        const DexConfig& c = x->cfg;
        return x->lp() && !debug && !knob_off("DEX_PATCH_FUSED") && x->lp_of().count(x->pe_pw) &&
               patch_embed_fused_supported(c.dit_patch, mid_dim(c), c.dit_hidden, (long)P.d.B * P.N);
    }
    // The TIV adaptor's y = IN2d(x) * s + m (ref_encoder.py:271) has ONE consumer, the patch embedding's depthwise convolution: outside
    // debug calls (the "tiv" tap) it is applied there on load - in either form of the patch embedding - from per-channel coefficients (launch_tiv_coef: the same fmaf, the same
    // bits) and the adaptor's output never goes to HBM and back (168 MB per step at B = 32).  DEX_TIV_FOLD=0: the separate launch.
    bool tiv_fold() const { return !debug && P.tiv_aff && knob_or("DEX_TIV_FOLD", 1) != 0; }
    // the TV adaptor runs as one launch (attention_bf16.hip tv_chain_kernel)
    bool tv_chain_on() const {
        return x->lp() && x->lp_of().count(x->tv_wl) && P.tv_kp && tv_chain_form(P.Hm * P.Wm, mid_dim(x->cfg), P.d.B);
    }
    // ... in its folded form (w_q and `linear` inside the style operands: kernels.h TvFold2P).  bf16 / fp16 modes: the split-weight mode would need K' and
    // V' as hi + lo pairs, i.e. twice the MFMAs of the key tiles that bound the kernel - it keeps the projections (DEX_TV_FOLD=0: every mode does)
    bool tv_fold_on() const { return tv_chain_on() && x->lpi() != 2 && P.tv_G && knob_or("DEX_TV_FOLD", 1) != 0; }
    // TVAdaptor + TIVAdaptor (ref_encoder.py:154-179,264-273); X is the (unmasked) bottleneck view.
    void dex_adaptors(const TD& X, int mask_ws) {
        const DexConfig& c = x->cfg;
        const int B = P.d.B, mid = mid_dim(c);
        const long npix = (long)P.Hm * P.Wm;
        // (both statistics arrays are zero here: cleared once in prepare(), then by this step's / the previous step's
        // time-token kernel, which runs after their last reader)
        InStatsP is{X.p + X.coff, X.ld, npix * X.ld, (int)npix, mid, P.tv_stats, B, mask, mask_ws, (long)P.d.T, P.Wm};
        run("in2d_stats", 3.0 * npix * mid * B, 4.0 * npix * mid * B, [&] { launch_in_stats(is, st); });
        const long stats_floats = (long)B * mid * IN_SLOTS * 2 * (long)(sizeof(gnfix_t) / sizeof(float));      // of ONE of the two statistics arrays
        if (tv_fold_on()) {
            // folded form: one element-wise launch turns the statistics into this step's K' (and the time token's rows), then the chain
            TvFold2P f2{P.tv_stats, (int)npix, 1e-5f, P.tv_G, (long)(P.d.Ts + 1) * mid, P.tv_g0, P.tv_v0p, sp, P.d.Ts + 1, P.tv_nkpad, mid,
                        1.0f / sqrtf((float)mid), P.tv_kp, P.tv_vtp, P.tv_xmean, reinterpret_cast<float*>(P.tiv_stats), stats_floats, x->lp_kind(), B};
            run("tv_fold_keys", 2.0 * B * (P.d.Ts + 1) * mid, 6.0 * B * (P.d.Ts + 1) * mid, [&] { launch_tv_fold2(f2, st); });
            TvChainP tc{X.p, X.ld, X.coff, npix * X.ld, (int)npix, P.Wm, mask, mask_ws, (long)P.d.T,
                        nullptr, 0L, nullptr, nullptr, 0L,
                        P.tv_kp, P.tv_vtp, P.tv_nkpad, P.d.Ts + 1, args->sty_lengths_dev, 1, 1.0f,
                        P.tv_out, P.tiv_stats, B, P.tv_xmean, reinterpret_cast<float*>(P.tv_stats), stats_floats};
            run("tv_chain", 4.0 * B * (double)npix * (P.d.Ts + 1) * mid, 8.0 * B * npix * mid, [&] { launch_tv_chain(tc, x->precision, st); });
            tap("tv", P.tv_out, B * npix, mid, mid);
            tiv(npix);
            return;
        }
        const bool qbf = x->lp();      // reduced-precision modes: the folded per-utterance weight is written as the MFMA GEMM operand
        InFoldP fo{P.tv_stats, (int)npix, 1e-5f, x->tv_wq_raw, mid, P.tv_weff, P.tv_beff, B, qbf ? P.tv_wbf : nullptr, x->lp_kind()};
        fo.split = x->lpi() == 2 ? 1 : 0;                            // split-weight mode: the folded weight gets its lo half too
        run("tv_fold_in2d", 2.0 * mid * mid * B, 8.0 * mid * mid * B, [&] { launch_in_fold(fo, st); });
        IGemmP q = base_gemm(X.p, X.ld, X.coff, P.Hm, P.Wm, mid, P.tv_weff, mid, P.tv_beff, P.tv_q, mid, 0);
        q.w_bstride = (long)mid * mid; q.bias_bstride = mid; q.inmask = mask; q.inmask_ws = mask_ws;
        if (qbf) { q.Wbf = P.tv_wbf; q.w_lo_off = fo.split ? (long)B * mid * mid : 0; }
        const bool chain = tv_chain_on();
        if (!chain) gemm("tv_q", q);
        TvRow0P r0{P.tv_k0, P.tv_v0, sp, P.tv_K, P.tv_V, (long)(P.d.Ts + 1) * mid, mid, B, reinterpret_cast<float*>(P.tv_stats),
                   (long)B * mid * IN_SLOTS * 2 * 2 * (long)(sizeof(gnfix_t) / sizeof(float))};
        if (chain) { r0.Kp = P.tv_kp; r0.VTp = P.tv_vtp; r0.NkPad = P.tv_nkpad; r0.lp_kind = x->lp_kind(); }
        run("tv_time_token", 0, 8.0 * mid * B, [&] { launch_tv_row0(r0, st); });
        // Batch regime, reduced-precision modes: q projection, attention, output projection, residual, mask and the TIV statistics as
        // ONE launch per 128 pixels (attention_bf16.hip tv_chain_kernel; DEX_TV_CHAIN=0: the three launches below).  q and the
        // attention output never reach HBM; the style keys / values become 16-bit operands once per call (prepare()), the time token's row per step.
        if (chain) {
            const void* wl = x->lp_of().at(x->tv_wl);
            TvChainP tc{X.p, X.ld, X.coff, npix * X.ld, (int)npix, P.Wm, mask, mask_ws, (long)P.d.T,
                        P.tv_wbf, fo.split ? (long)B * mid * mid : 0L, P.tv_beff, wl, x->lo_off(wl),
                        P.tv_kp, P.tv_vtp, P.tv_nkpad, P.d.Ts + 1, args->sty_lengths_dev, 1, 1.0f / sqrtf((float)mid),
                        P.tv_out, P.tiv_stats, B};
            run("tv_chain", 4.0 * B * (double)npix * mid * mid + 4.0 * B * (double)npix * (P.d.Ts + 1) * mid, 8.0 * B * npix * mid,
                [&] { launch_tv_chain(tc, x->precision, st); });
            tap("tv", P.tv_out, B * npix, mid, mid);
            tiv(npix);
            return;
        }
        AttnP a{};
        a.Q = P.tv_q; a.ldq = mid; a.qb = npix * mid; a.K = P.tv_K; a.ldk = mid; a.kb = (long)(P.d.Ts + 1) * mid;
        a.V = P.tv_V; a.ldv = mid; a.vb = a.kb; a.O = P.tv_ao; a.ldo = mid; a.ob = npix * mid;
        a.Nq = (int)npix; a.Nk = P.d.Ts + 1; a.kv_len = args->sty_lengths_dev; a.kv_len_add = 1; a.heads = 1;
        a.scale = 1.0f / sqrtf((float)mid); a.B = B; a.head_dim = mid;
        // the attention output has one reader, the output projection below, which rounds it to the operand type: at batch size it is
        // stored in that type (same bits, half the bytes of one 84 MB tensor written and read per Euler step at B = 32)
        const bool ao_lp = lp_inter_cur && mid == 128 && x->lp_of().count(x->tv_wl) && attention_lp_shared_form((int)npix, 1, B, 1);
        a.o_lp = ao_lp ? 1 : 0;
        run("tv_attention", 4.0 * B * (double)npix * (P.d.Ts + 1) * mid, (ao_lp ? 2.0 : 4.0) * B * npix * mid + 4.0 * B * (npix + 2 * (P.d.Ts + 1)) * mid, [&] { launch_attention(a, x->precision, st); });
        IGemmP o = base_gemm(P.tv_ao, mid, 0, P.Hm, P.Wm, mid, x->tv_wl, mid, nullptr, P.tv_out, mid, 0);
        if (ao_lp) o.a_lp = x->lp_kind();
        o.res = X.p; o.ldres = X.ld; o.res_coff = X.coff; o.res_bstride = npix * X.ld;
        o.outmask = mask; o.outmask_ws = mask_ws;
        // the TIV adaptor's InstanceNorm statistics of this output ride in the epilogue (per-channel = groups of one)
        o.gn_stats = P.tiv_stats; o.gn_groups = mid; o.gn_cpg = 1; o.stats_final = 1;
        gemm("tv_out", o);
        tap("tv", P.tv_out, B * npix, mid, mid);
        tiv(npix);
    }
    void tiv(long npix) {
        const int B = P.d.B, mid = mid_dim(x->cfg);
        TivApplyP ta{P.tv_out, mid, npix * mid, P.tiv_out, mid, npix * mid, (int)npix, mid, P.tiv_stats, 1e-5f, P.sap_s, P.sap_m, sp, B};
        if (tiv_fold()) { run("tiv_coefficients", 0, 16.0 * mid * B, [&] { launch_tiv_coef(ta, P.tiv_aff, st); }); return; }
        run("tiv_adain", 2.0 * npix * mid * B, 8.0 * npix * mid * B, [&] { launch_tiv_apply(ta, st); });
        tap("tiv", P.tiv_out, B * npix, mid, mid);
    }

    // One EDMPrecond + Euler update (edm.py:88-98,199-208).
    void step(float* denoised, float* xnext) {
        const DexConfig& c = x->cfg;
        const int B = P.d.B, ns = c.n_stages;
        gn_idx = 0;
        if (debug) {                          // (G1 checkpoints: this step's rows of the conditioning tables - the time MLP and the DiT's TimestepEmbedder)
            tap("mlp", P.temb + (long)sp * c.dim, 1, c.dim, c.dim);
            tap("vit.t_embedder", P.c_emb + (long)sp * c.dit_hidden, 1, c.dit_hidden, c.dit_hidden);
        }
        if (!stats_other) zero_fill(stats_base, P.stats_bytes, st);   // single call (dex_denoise_once): clear in place
        TD cur{nullptr, 0, 0, 0};
        // Activations whose EVERY consumer rounds them to the MFMA operand type while staging (x * mask with a 0 / 1 mask) are
        // stored in that type: bit-identical results, half the bytes.  That is the down path's attention output into the
        // Downsample conv, the Downsample output into the next ResnetBlock's convs, the last up stage's attention output into
        // the Upsample, and the Upsample output into the final block's conv.  Batch regime only (the kernels that read / write
        // 16-bit tensors are the throughput forms); never with debug taps (they read fp32); DEX_LP_INTER=0 turns it off.
        const bool lp_inter = x->lp() && !debug && ns >= 2 && !knob_off("DEX_LP_INTER") && fast_conv(c.dim, c.dim);
        lp_inter_cur = lp_inter;
        cat_lp = false;
        if (lp_inter && ns == 2 && P.cat16) {
            const StageBuf& sm_ = P.down[ns - 1];
            const ResW& uw = x->up_res[0][0];
            const bool tail_ok = linattn_fused(sm_.C) && linattn_out2_lp_out_supported((int)sm_.npix, B);
            const bool conv_ok = uw.wr && fast_conv(2 * sm_.C, uw.cout) && conv3x3_bf16_res_supported(2 * sm_.C, uw.cout) && x->lp_of().count(uw.wr) &&
                                 x->lp_of().count(uw.w1) && conv3x3_cat_lp_in_supported(P.up[0].H, P.up[0].W, B, 2 * sm_.C, uw.cout);
            IGemmP fl = final_gemm(sm_.mask_ws, P.cat[0], 2 * sm_.C, 0);
            fl.c_lp = x->lp_kind();
            cat_lp = !knob_off("DEX_CAT_LP") && tail_ok && conv_ok && fl.Wbf && igemm_nwalk_form(fl);
        }
        const int lpk = x->lp_kind();
        for (int i = 0; i < ns; ++i) {
            const StageBuf& s = P.down[i];
            // block 0's tail (GN-apply + Mish + res_conv shortcut) rides in block 1's first conv when that conv has the
            // fused form (bf16 mode, 64/128 channels, block 0 has a res_conv)
            Pro t0{};
            const bool defer0 = x->lp() && conv3x3_bf16_tail_supported(s.C) && fast_conv(s.C, s.C) &&
                                (i == 0 || x->down_res[i][0].wr != nullptr);
            resblock(x->down_res[i][0], s, cur, P.tadd_down[2 * i], s.r0out, i == 0, nullptr, nullptr, defer0 ? &t0 : nullptr);
            TD r0{s.r0out, s.C, 0, s.C};
            LinKvCtxP tail{};
            const bool defer = linattn_fused(s.C);
            resblock(x->down_res[i][1], s, r0, P.tadd_down[2 * i + 1], s.r1out, false, defer ? &tail : nullptr, defer0 ? &t0 : nullptr);
            TD r1{s.r1out, s.C, 0, s.C};
            if (debug && !x->lp()) {          // module-level checkpoints (the reference's module paths): the exact-fp32 mode materialises them
                char n0[24], n1[24]; snprintf(n0, sizeof n0, "downs.%d.0", i); snprintf(n1, sizeof n1, "downs.%d.1", i);
                tap(n0, s.r0out, B * s.npix, s.C, s.C); tap(n1, s.r1out, B * s.npix, s.C, s.C);
            }
            // the Downsample conv is this output's only reader at level 0 (the reference's hiddens.append of this level is never
            // popped; deeper levels live in the up path's concatenation buffer, which is read as fp32)
            const bool t1_lp = lp_inter && i == 0 && i < ns - 1 && linattn_fused(s.C) && linattn_out2_lp_out_supported((int)s.npix, B) && x->lp_of().count(x->down_ds_w[i]);
            const bool skip16 = cat_lp && i == ns - 1;          // the mid stage's output is also the skip half of the up path's concatenation buffer
            linattn(x->down_lin[i], s, r1, s.attn_out, s.attn_ld, s.attn_coff, defer ? &tail : nullptr, t1_lp,
                    skip16 ? P.cat16 : nullptr, s.attn_ld, s.attn_coff);
            char nm[16]; snprintf(nm, sizeof nm, "down%d", i);
            tap(nm, s.attn_out + s.attn_coff, B * s.npix, s.C, s.attn_ld);
            if (i < ns - 1) {
                TD a{s.attn_out, s.attn_ld, s.attn_coff, s.C};
                IGemmP g = base_gemm(a.p, a.ld, a.coff, s.H, s.W, s.C, x->down_ds_w[i], s.C, x->down_ds_b[i], s.ds_out, s.C, 0);
                g.KH = 3; g.KW = 3; g.sh = 2; g.sw = 2; g.off_h = -1; g.off_w = -1; g.K = 9 * s.C;
                g.Ho = s.H / 2; g.Wo = s.W / 2; g.OHf = g.Ho; g.OWf = g.Wo; g.c_bstride = (long)g.Ho * g.Wo * s.C;
                g.inmask = mask; g.inmask_ws = s.mask_ws;
                // its output feeds the next stage's first ResnetBlock (3x3 conv + fused 1x1 shortcut) and nothing else
                const ResW& nw = x->down_res[i + 1][0];
                const bool t2_lp = lp_inter && x->lp_of().count(x->down_ds_w[i]) && nw.wr && fast_conv(s.C, nw.cout) && conv3x3_bf16_res_supported(s.C, nw.cout) &&
                                   x->lp_of().count(nw.wr) && x->lp_of().count(nw.w1) && conv3x3_plain_lp_in_supported(g.Ho, g.Wo, B, s.C, nw.cout);
                g.a_lp = t1_lp ? lpk : 0; g.c_lp = t2_lp ? lpk : 0;
                const bool strip_off = knob_off("DEX_CONV_DOWN");
                if (x->lp() && !strip_off && x->frag_of().count(x->down_ds_w[i]) && conv_down_supported(s.C, s.H, s.W, a.ld, s.C, a.coff)) {
                    ConvDownP d{};
                    d.X = a.p; d.a_lp = g.a_lp; d.ldx = a.ld; d.xb = (long)s.H * s.W * a.ld; d.x_coff = a.coff; d.H = s.H; d.W = s.W;
                    d.Wfrag = x->frag_of().at(x->down_ds_w[i]); d.bias = x->down_ds_b[i];
                    d.Y = s.ds_out; d.c_lp = g.c_lp; d.ldy = s.C; d.y_coff = 0;
                    d.inmask = mask; d.inmask_ws = s.mask_ws; d.mask_bstride = P.d.T; d.B = B;
                    const double M = 0.25 * s.H * s.W * B;
                    run("downsample", 2.0 * M * s.C * 9 * s.C, 4.0 * M * s.C * (g.a_lp ? 2.0 : 4.0) + M * s.C * (g.c_lp ? 2.0 : 4.0), [&] { launch_conv_down(d, x->precision, st); });
                } else
                gemm("downsample", g);
                cur = TD{s.ds_out, s.C, 0, s.C, t2_lp ? lpk : 0};
                if (debug && !x->lp()) { char nd[24]; snprintf(nd, sizeof nd, "downs.%d.3", i); tap(nd, s.ds_out, (long)B * g.Ho * g.Wo, s.C, s.C); }
            }
        }
        const StageBuf& sm = P.down[ns - 1];
        TD mid_in{sm.attn_out, sm.attn_ld, sm.attn_coff, sm.C};
        float* dit_dst = P.cat[0];
        const int dit_ld = 2 * sm.C;
        if (c.variant == DEX_VARIANT_DEX) {
            dex_adaptors(mid_in, sm.mask_ws);
            TD t{tiv_fold() ? P.tv_out : P.tiv_out, sm.C, 0, sm.C};
            dit(t, false, sm.mask_ws, dit_dst, dit_ld, 0, tiv_fold() ? P.tiv_aff : nullptr);
        } else {
            dit(mid_in, true, sm.mask_ws, dit_dst, dit_ld, 0);
        }
        tap("dit_out", dit_dst, B * sm.npix, sm.C, dit_ld);
        bool up_out_lp = false;
        for (int j = 0; j < ns - 1; ++j) {
            const StageBuf& s = P.up[j];
            const int i = ns - 1 - j;
            TD X{P.cat[j], 2 * stage_dim(c, i), 0, 2 * stage_dim(c, i)};
            if (cat_lp && j == 0) X = TD{reinterpret_cast<float*>(P.cat16), 2 * stage_dim(c, i), 0, 2 * stage_dim(c, i), lpk};
            Pro t0{};
            const bool defer0 = x->lp() && conv3x3_bf16_tail_supported(s.C) && fast_conv(s.C, s.C) &&
                                x->up_res[j][0].wr != nullptr;
            resblock(x->up_res[j][0], s, X, P.tadd_up[2 * j], s.r0out, false, nullptr, nullptr, defer0 ? &t0 : nullptr);
            TD r0{s.r0out, s.C, 0, s.C};
            LinKvCtxP tail{};
            const bool defer = linattn_fused(s.C);
            resblock(x->up_res[j][1], s, r0, P.tadd_up[2 * j + 1], s.r1out, false, defer ? &tail : nullptr, defer0 ? &t0 : nullptr);
            TD r1{s.r1out, s.C, 0, s.C};
            if (debug && !x->lp()) {
                char n0[24], n1[24]; snprintf(n0, sizeof n0, "ups.%d.0", j); snprintf(n1, sizeof n1, "ups.%d.1", j);
                tap(n0, s.r0out, B * s.npix, s.C, s.C); tap(n1, s.r1out, B * s.npix, s.C, s.C);
            }
            const bool t5_lp = lp_inter && linattn_fused(s.C) && linattn_out2_lp_out_supported((int)s.npix, B) && x->lp_of().count(x->up_us_w[j]);
            linattn(x->up_lin[j], s, r1, s.attn_out, s.C, 0, defer ? &tail : nullptr, t5_lp);
            char nm[16]; snprintf(nm, sizeof nm, "up%d", j);
            tap(nm, s.attn_out, B * s.npix, s.C, s.C);
            // Upsample = ConvTranspose2d(4,2,1) on x*mask: four parity sub-convolutions with 2x2 taps
            float* dst; int ldd;
            if (j < ns - 2) { dst = P.cat[j + 1]; ldd = 2 * stage_dim(c, i - 1); } else { dst = P.up_out; ldd = s.C; }
            {
                IGemmP g = base_gemm(s.attn_out, s.C, 0, s.H, s.W, s.C, x->up_us_w[j], s.C, x->up_us_b[j], dst, ldd, 0);
                g.KH = 2; g.KW = 2; g.step_h = -1; g.step_w = -1; g.K = 4 * s.C; g.parity = 1;
                g.OHf = 2 * s.H; g.OWf = 2 * s.W; g.osh = 2; g.osw = 2;
                g.c_bstride = 4L * s.H * s.W * ldd;
                g.inmask = mask; g.inmask_ws = s.mask_ws;
                g.a_lp = t5_lp ? lpk : 0;
                up_out_lp = lp_inter && j == ns - 2 && x->lp_of().count(x->up_us_w[j]) && x->lp_of().count(x->fin_w) && conv3x3_res2_form(80, P.d.T, B);
                g.c_lp = up_out_lp ? lpk : 0;
                const bool strip_off = knob_off("DEX_CONVT_UP");
                if (x->lp() && x->frag_of().count(x->up_us_w[j]) && !strip_off && convt_up_supported(s.C, s.H, s.W, s.C, ldd)) {
                    ConvTUpP u{};
                    u.X = s.attn_out; u.a_lp = g.a_lp; u.ldx = s.C; u.xb = (long)s.H * s.W * s.C; u.x_coff = 0; u.H = s.H; u.W = s.W;
                    for (int par = 0; par < 4; ++par) u.Wfrag[par] = x->frag_of().at(x->up_us_w[j] + (long)par * 4 * s.C * s.C);
                    u.bias = x->up_us_b[j];
                    u.Y = dst; u.c_lp = g.c_lp; u.ldy = ldd; u.y_coff = 0;
                    u.inmask = mask; u.inmask_ws = s.mask_ws; u.mask_bstride = P.d.T; u.B = B;
                    const double M = 4.0 * s.H * s.W * B;
                    run("upsample_convT", 2.0 * M * s.C * 4 * s.C, M * s.C * (g.c_lp ? 2.0 : 4.0) + 0.25 * M * s.C * (g.a_lp ? 2.0 : 4.0),
                        [&] { launch_convt_up(u, x->precision, st); });
                } else gemm("upsample_convT", g);
            }
        }
        tap("up_out", P.up_out, (long)B * 80 * P.d.T, c.dim, c.dim);
        TD U{P.up_out, c.dim, 0, c.dim, up_out_lp ? lpk : 0};
        gnfix_t* stf = next_stats();
        const bool hfb = h_bf16() && fast_conv(c.dim, c.dim) && x->lp_of().count(x->fin_w);
        conv3x3("conv3x3", U, 80, P.d.T, 1, true, x->fin_w, x->fin_b, c.dim, P.hF, stf, nullptr, nullptr, nullptr, false, hfb);
        FinalP f{};
        f.x_bf16 = hfb ? x->lp_kind() : 0;
        f.X = P.hF; f.xb = 80L * P.d.T * c.dim; f.npix = 80 * P.d.T; f.W = P.d.T; f.C = c.dim; f.groups = 8; f.stats = stf;
        f.gamma = x->fin_g; f.beta = x->fin_be; f.mask = mask; f.mask_bstride = P.d.T; f.wfc = x->fconv_w; f.bfc = x->fconv_b;
        f.xcur = xcur; f.denoised = denoised; f.xnext = xnext; f.scal = P.scal; f.scal_stride = SCAL_STRIDE; f.step = sp; f.B = B;
        f.zero_ptr = reinterpret_cast<float*>(stats_other); f.zero_n = P.stats_bytes / (long)sizeof(float);
        f.poison = P.xflag ? reinterpret_cast<const int*>(P.xflag + (P.xflag_bytes - sizeof(int)) / sizeof(unsigned)) : nullptr;
        f.mode = fin_mode; f.htab = fin_htab; f.dbuf = P.dbuf; f.xhat = P.xbuf;
        run("final_conv_euler", 14.0 * 80 * P.d.T * c.dim * B, 80.0 * P.d.T * ((hfb ? 2.0 : 4.0) * c.dim + 12.0) * B, [&] { launch_final(f, st); });
    }

    // conditioning tables for every Euler step (depend only on sigma_i)
    void prepare(const float* sigmas_dev, int n) {
        const DexConfig& c = x->cfg;
        const int dim = c.dim, hid = c.dit_hidden, B = P.d.B, mid = mid_dim(c);
        auto R = [&](const std::string& k) { return x->raw.at(k).p; };
        auto lin = [&](const float* X, int ldx, int rows, int K, const std::string& w, bool bias, int N, float* Y, int ai, int ao) {
            SmallLinP s{X, ldx, rows, K, R(w + ".weight"), bias ? R(w + ".bias") : nullptr, N, Y, N, ai, ao};
            run("cond_mlp", 2.0 * rows * K * N, 4.0 * K * N, [&] { launch_small_linear(s, st); });
        };
        if (P.tv_stats) zero_fill(P.tv_stats, (size_t)B * mid_dim(c) * IN_SLOTS * 2 * 2 * sizeof(gnfix_t), st);
        if (P.xflag) zero_fill(P.xflag, P.xflag_bytes, st);     // hand-off flags of the cluster row chain: zero before every call (epochs count within it)
        zero_fill(P.vt, P.vt_bytes, st);      // key padding of the transposed V operand (attention_direct.hip)
        zero_fill(P.vt2, P.vt_bytes, st);
        CondPrepP cp{sigmas_dev, n, c.pe_scale, dim, P.scal, SCAL_STRIDE, P.t_unet, P.t_dit};
        run("cond_prep", 0, 0, [&] { launch_cond_prep(cp, st); });
        lin(P.t_unet, dim, n, dim, "mlp.0", true, 4 * dim, P.tmp_u, 0, 1);
        lin(P.tmp_u, 4 * dim, n, 4 * dim, "mlp.2", true, dim, P.temb, 0, 0);
        for (int i = 0; i < c.n_stages; ++i)
            for (int r = 0; r < 2; ++r)
                lin(P.temb, dim, n, dim, "downs." + std::to_string(i) + "." + std::to_string(r) + ".mlp.1", true, stage_dim(c, i), P.tadd_down[2 * i + r], 1, 0);
        for (int j = 0; j < c.n_stages - 1; ++j)
            for (int r = 0; r < 2; ++r)
                lin(P.temb, dim, n, dim, "ups." + std::to_string(j) + "." + std::to_string(r) + ".mlp.1", true, stage_dim(c, c.n_stages - 2 - j), P.tadd_up[2 * j + r], 1, 0);
        lin(P.t_dit, 256, n, 256, "vit.t_embedder.mlp.0", true, hid, P.c_tmp, 0, 2);
        lin(P.c_tmp, hid, n, hid, "vit.t_embedder.mlp.2", true, hid, P.c_emb, 0, 0);
        for (int k = 0; k < c.dit_depth; ++k)
            lin(P.c_emb, hid, n, hid, "vit.blocks." + std::to_string(k) + ".adaLN_modulation.1", true, 6 * hid, P.ada[k], 2, 0);
        lin(P.c_emb, hid, n, hid, "vit.final_layer.adaLN_modulation.1", true, 2 * hid, P.fin_mod, 2, 0);
        if (c.n_spks > 1) {
            lin(args->spk_dev, c.spk_emb_dim, B, c.spk_emb_dim, "spk_mlp.0", true, 4 * c.spk_emb_dim, P.spk_tmp, 0, 1);
            lin(P.spk_tmp, 4 * c.spk_emb_dim, B, 4 * c.spk_emb_dim, "spk_mlp.2", true, c.n_feats, P.spk_plane, 0, 0);
        }
        if (c.variant == DEX_VARIANT_DEX) {
            lin(P.t_unet, dim, n, dim, "mlp_adap.0", true, dim, P.adap_tmp, 0, 1);
            lin(P.adap_tmp, dim, n, dim, "mlp_adap.2", true, 2 * dim, P.t_adap, 0, 0);
            lin(P.t_unet, dim, n, dim, "mlp_adap_sty.0", true, dim, P.adap_tmp, 0, 1);
            lin(P.adap_tmp, dim, n, dim, "mlp_adap_sty.2", true, 2 * dim, P.t_sty, 0, 0);
            lin(P.t_sty, mid, n, mid, "tv_adaptor.w_k", false, mid, P.tv_k0, 0, 0);
            lin(P.t_sty, mid, n, mid, "tv_adaptor.w_v", false, mid, P.tv_v0, 0, 0);
            const int L = args->n_ref;
            for (int j = 0; j < L; ++j)
                run("ref_stats", 0, 4.0 * B * mid * args->Tr, [&] {
                    launch_row_stats(args->ref_skips_dev[j], B, mid, args->Tr, 1e-5f, P.ref_mean + (long)j * mid, P.ref_std + (long)j * mid, (long)L * mid, st);
                });
            SapP sm{P.t_adap, 2 * dim, 0, n, P.ref_mean, L, mid, R("tiv_adaptor.mean_sap.W.weight"), R("tiv_adaptor.mean_sap.W.bias"), P.sap_m, B};
            SapP ss{P.t_adap, 2 * dim, 0, n, P.ref_std, L, mid, R("tiv_adaptor.std_sap.W.weight"), R("tiv_adaptor.std_sap.W.bias"), P.sap_s, B};
            run("sap_pool", 0, 0, [&] { launch_sap(sm, st); launch_sap(ss, st); });
            // style keys/values for the Ts style tokens are step-invariant (rows 1..Ts of K/V)
            run("sty_transpose", 0, 8.0 * B * mid * args->Ts, [&] { launch_transpose_cl(args->sty_dev, P.tv_keys, B, mid, args->Ts, 0, (long)args->Ts * mid, st); });
            for (int which = 0; which < 2; ++which) {
                float* dst = which ? P.tv_V : P.tv_K;
                IGemmP g = base_gemm(P.tv_keys, mid, 0, 1, args->Ts, mid, which ? x->tv_wv : x->tv_wk, mid, nullptr, dst + mid, mid, 0);
                g.c_bstride = (long)(args->Ts + 1) * mid;
                gemm("tv_kv", g);
            }
            if (tv_chain_on()) {       // ... and their 16-bit operand forms (row 0, the time token, is rewritten every step: launch_tv_row0)
                TvKvPrepP kp{P.tv_K, P.tv_V, (long)(args->Ts + 1) * mid, args->Ts + 1, P.tv_nkpad, P.tv_kp, P.tv_vtp, B};
                run("tv_kv_operands", 0, 12.0 * B * (args->Ts + 1) * mid, [&] { launch_tv_kv_prep(kp, x->precision, st); });
            }
            if (tv_fold_on()) {        // folded form: G = K W_q and V' = V W_l^T in fp32 (style rows once per call, the time token's rows of every step), V'^T as the 16-bit operand
                const long kvb = (long)(args->Ts + 1) * mid;
                for (int which = 0; which < 2; ++which) {
                    IGemmP g = base_gemm((which ? P.tv_V : P.tv_K) + mid, mid, 0, 1, args->Ts, mid, which ? x->tv_wl : x->tv_wq_raw, mid, nullptr,
                                         (which ? P.tv_Vp : P.tv_G) + mid, mid, 0);       // w_q raw [n][k] IS the packed [K = n][N = k] matrix of K W_q
                    g.a_bstride = kvb; g.c_bstride = kvb; g.Wbf = nullptr;
                    gemm("tv_fold_kv", g);
                    IGemmP t = base_gemm(which ? P.tv_v0 : P.tv_k0, mid, 0, 1, n, mid, which ? x->tv_wl : x->tv_wq_raw, mid, nullptr, which ? P.tv_v0p : P.tv_g0, mid, 0);
                    t.B = 1; t.Wbf = nullptr;
                    gemm("tv_fold_time_rows", t);
                }
                TvKvPrepP kp{P.tv_G, P.tv_Vp, kvb, args->Ts + 1, P.tv_nkpad, P.tv_kp, P.tv_vtp, B};      // (K' is written by every step's launch_tv_fold2)
                run("tv_fold_operands", 0, 6.0 * B * (args->Ts + 1) * mid, [&] { launch_tv_vfrag_prep(kp, x->precision, st); });
            }
        }
    }
};

int validate(DexCtx* x, const DexSampleArgs* a, bool need_z) {
    if (!x || !a) return DEX_ERR_ARG;
    if (!x->finalized) return x->fail(DEX_ERR_STATE, "dex_ctx_finalize has not been called");
    if (a->B < 1 || a->T < 4 || (a->T % 4) != 0) return x->fail(DEX_ERR_ARG, "T (%d) must be a positive multiple of 4 (fix_len_compatibility), B >= 1", a->T);
    if ((a->T >> (x->cfg.n_stages - 1)) * (1 << (x->cfg.n_stages - 1)) != a->T) return x->fail(DEX_ERR_ARG, "T must be divisible by 2^(n_stages-1)");
    if (need_z && a->n_steps < 2) return x->fail(DEX_ERR_ARG, "n_steps must be >= 2 (edm.py:157 divides by num_steps - 1)");
    if (!a->mu_dev || !a->mask_dev || !a->sigmas_dev || !a->out_dev || !a->workspace_dev) return x->fail(DEX_ERR_ARG, "null device pointer");
    if (need_z && !a->z_dev) return x->fail(DEX_ERR_ARG, "z_dev is null");
    if (need_z && a->S_churn < 0.f) return x->fail(DEX_ERR_ARG, "S_churn must be >= 0");
    if (need_z && a->S_churn > 0.f && !a->noise_dev) return x->fail(DEX_ERR_ARG, "S_churn > 0 needs noise_dev ([n_steps][B,80,T] draws of randn_like, edm.py:196)");
    if (x->cfg.n_spks > 1 && !a->spk_dev) return x->fail(DEX_ERR_ARG, "spk_dev required when n_spks > 1");
    if (x->cfg.variant == DEX_VARIANT_DEX) {
        if (!a->ref_skips_dev || !a->sty_dev || !a->sty_lengths_dev || a->Tr < 2 || a->Ts < 1 || a->n_ref < 1 || a->n_ref > 7)
            return x->fail(DEX_ERR_ARG, "DEX needs ref_skips_dev (1..7 tensors, Tr >= 2), sty_dev, sty_lengths_dev, Ts >= 1");
        for (int j = 0; j < a->n_ref; ++j) if (!a->ref_skips_dev[j]) return x->fail(DEX_ERR_ARG, "ref_skips_dev[%d] is null", j);
    }
    if (((uintptr_t)a->workspace_dev & 255) != 0) return x->fail(DEX_ERR_ARG, "workspace must be 256-byte aligned");
    return DEX_OK;
}

__global__ void set_sigma_pair(const float* src, float* dst) { dst[0] = src[0]; dst[1] = 0.f; }

// ablation_sampler(solver='heun', alpha=1) — edm.py:199-214.  2n-1 network evaluations: evaluation 2i is step i's
// predictor at t_i, evaluation 2i+1 its corrector at t' = t_i + h (none on the last step).  The conditioning tables are
// built per EVALUATION; the last kernel of each evaluation does the predictor / corrector update.  Eager launches only.
int enqueue_heun(DexCtx* x, const DexSampleArgs* a, hipStream_t st) {
    const int n = a->n_steps, E = 2 * n - 1;
    Plan P; Dims d{a->B, a->T, a->Tr, a->Ts, E};
    make_plan(x, d, a->workspace_dev, P);
    Runner R{x, P, st, a->mask_dev, a->mu_dev, P.xbuf, a, false};
    launch_churn_tables(a->sigmas_dev, n, a->S_churn, a->S_min, a->S_max, a->S_noise, P.that, P.hstep, P.ncoef, st);
    launch_heun_expand(P.that, P.hstep, n, P.hsig, P.htab, st);
    R.prepare(P.hsig, E);
    const bool churn = a->S_churn > 0.f;
    const long nx = (long)a->B * 80 * a->T;
    // x_0 = z * t_0 (edm.py:188-189; t_0, not t_hat_0)
    R.run("init_scale", 0, 8.0 * nx, [&] { launch_scale_copy(a->z_dev, P.xbuf, nx, a->sigmas_dev, st); });
    gnfix_t* arena[2] = {P.stats, P.stats + P.stats_bytes / (long)sizeof(gnfix_t)};
    zero_fill(P.stats, 2 * P.stats_bytes, st);
    R.fin_htab = P.htab;
    for (int e = 0; e < E; ++e) {
        const bool corrector = (e & 1) != 0, last = (e == E - 1);
        R.sp = e; R.stats_base = arena[e & 1]; R.stats_other = arena[(e + 1) & 1];
        if (churn && !corrector)        // x_hat = x_cur + sqrt(t_hat^2 - t_cur^2) S_noise randn_like(x_cur), in place (edm.py:196)
            R.run("churn_noise", 2.0 * nx, 12.0 * nx, [&] { launch_add_noise(P.xbuf, a->noise_dev + (long)(e / 2) * nx, P.ncoef + e / 2, nx, st); });
        R.xcur = corrector ? P.xprime : P.xbuf;
        R.fin_mode = corrector ? 2 : (last ? 0 : 1);
        R.step(nullptr, (corrector || last) ? P.xbuf : P.xprime);
    }
    R.run("copy_out", 0, 8.0 * nx, [&] { launch_scale_copy(P.xbuf, a->out_dev, nx, nullptr, st); });      // (a kernel node like every other link of a captured call: see zero_fill)
    return DEX_OK;
}

// ablation_sampler(solver='euler') — edm.py:186-208.  The step index is a launch argument and the GroupNorm statistics
// alternate between two arenas; each step's last kernel clears the arena of the next step.
int enqueue_euler(DexCtx* x, const DexSampleArgs* a, hipStream_t st) {
    Plan P; Dims d{a->B, a->T, a->Tr, a->Ts, a->n_steps};
    make_plan(x, d, a->workspace_dev, P);
    Runner R{x, P, st, a->mask_dev, a->mu_dev, P.xbuf, a, false};
    R.sp = 0;
    const bool churn = a->S_churn > 0.f;
    const long nx = (long)a->B * 80 * a->T;
    if (churn) {        // the network sees t_hat_i, the update uses h_i = t_{i+1} - t_hat_i (edm.py:194-199)
        launch_churn_tables(a->sigmas_dev, a->n_steps, a->S_churn, a->S_min, a->S_max, a->S_noise, P.that, P.hstep, P.ncoef, st);
        R.prepare(P.that, a->n_steps);
        R.fin_htab = P.hstep;
    } else {
        R.prepare(a->sigmas_dev, a->n_steps);
    }
    // x_0 = z * t_0 (edm.py:188-189)
    R.run("init_scale", 0, 8.0 * nx, [&] { launch_scale_copy(a->z_dev, P.xbuf, nx, a->sigmas_dev, st); });
    gnfix_t* arena[2] = {P.stats, P.stats + P.stats_bytes / (long)sizeof(gnfix_t)};
    zero_fill(P.stats, 2 * P.stats_bytes, st);
    for (int i = 0; i < a->n_steps; ++i) {
        R.sp = i; R.stats_base = arena[i & 1]; R.stats_other = arena[(i + 1) & 1];
        if (churn)          // x_hat = x_cur + sqrt(t_hat^2 - t_cur^2) S_noise randn_like(x_cur), in place (edm.py:196)
            R.run("churn_noise", 2.0 * nx, 12.0 * nx, [&] { launch_add_noise(P.xbuf, a->noise_dev + (long)i * nx, P.ncoef + i, nx, st); });
        R.step(nullptr, P.xbuf);
    }
    R.run("copy_out", 0, 8.0 * nx, [&] { launch_scale_copy(P.xbuf, a->out_dev, nx, nullptr, st); });      // (a kernel node like every other link of a captured call: see zero_fill)
    return DEX_OK;
}

}  // namespace

extern "C" {

size_t dex_workspace_bytes(const DexCtx* x, int B, int T, int Tr, int Ts, int n_steps) {
    if (!x || B < 1 || T < 4) return 0;
    const WsplitScope wsplit_scope(x->precision == DEX_PREC_FP16X2);      // (the plan asks the same shape predicates as the call will)
    Plan P; Dims d{B, T, Tr, Ts, n_steps < 1 ? 1 : n_steps};
    make_plan(x, d, nullptr, P);
    return P.bytes;
}

int dex_num_evals(int n_steps, int solver) { return solver == DEX_SOLVER_HEUN ? 2 * n_steps - 1 : n_steps; }

int dex_edm_sigmas(int n, float* out) {
    if (n < 2 || !out) return DEX_ERR_ARG;
    const double a = pow(80.0, 1.0 / 7.0), bq = pow(0.002, 1.0 / 7.0);
    for (int i = 0; i < n; ++i) {
        const float frac = (float)i / (float)(n - 1);
        const float base = (float)a + frac * (float)(bq - a);
        out[i] = powf(base, 7.0f);
    }
    out[n] = 0.f;
    return DEX_OK;
}

int dex_denoise_once(DexCtx* x, const DexDenoiseArgs* da, dex_stream_t stream) {
    if (!da) return DEX_ERR_ARG;
    const DexSampleArgs* a = &da->s;
    int rc = validate(x, a, false);
    if (rc) return rc;
    if (!da->x_dev) return x->fail(DEX_ERR_ARG, "x_dev is null");
    const KnobSnapshot knobs;
    const KnobScope knob_scope(&knobs);
    const WsplitScope wsplit_scope(x->precision == DEX_PREC_FP16X2);
    xcd_map_probe();
    x->last_xerr = nullptr;
    hipStream_t st = (hipStream_t)stream;
    Plan P; Dims d{a->B, a->T, a->Tr, a->Ts, 1};
    make_plan(x, d, nullptr, P);
    if (P.bytes > a->workspace_bytes) return x->fail(DEX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", P.bytes, a->workspace_bytes);
    make_plan(x, d, a->workspace_dev, P);
    x->taps.clear();
    if (x->prof_on) { for (auto& pr : x->prof) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); } x->prof.clear(); x->prof_agg.clear(); }     // (rows describe the last call, as in dex_sample)
    Runner R{x, P, st, a->mask_dev, a->mu_dev, da->x_dev, a, true};
    R.sp = 0; R.stats_base = P.stats; R.stats_other = nullptr;
    hipLaunchKernelGGL(set_sigma_pair, dim3(1), dim3(1), 0, st, a->sigmas_dev, P.sig2);
    R.prepare(P.sig2, 1);
    R.step(a->out_dev, nullptr);
    HIPCHK(x, hipGetLastError());
    return DEX_OK;
}

int dex_sample(DexCtx* x, const DexSampleArgs* a, dex_stream_t stream) {
    int rc = validate(x, a, true);
    if (rc) return rc;
    if (a->solver != DEX_SOLVER_EULER && a->solver != DEX_SOLVER_HEUN) return x->fail(DEX_ERR_ARG, "solver must be DEX_SOLVER_EULER or DEX_SOLVER_HEUN (edm.py:107)");
    const KnobSnapshot knobs;            // one reading of every knob for this call (graph-cache key below)
    const KnobScope knob_scope(&knobs);
    const WsplitScope wsplit_scope(x->precision == DEX_PREC_FP16X2);
    xcd_map_probe();                    // (once per device, before any capture; normally already done by dex_ctx_create)
    x->last_xerr = nullptr;             // set again by this call if it uses in-launch hand-offs (dex_call_status)
    hipStream_t st = (hipStream_t)stream;
    const bool heun = a->solver == DEX_SOLVER_HEUN;
    {
        Plan P; Dims d{a->B, a->T, a->Tr, a->Ts, dex_num_evals(a->n_steps, a->solver)};
        make_plan(x, d, nullptr, P);
        if (P.bytes > a->workspace_bytes)
            return x->fail(DEX_ERR_WORKSPACE, "workspace too small: need %zu bytes (dex_workspace_bytes with dex_num_evals(n_steps, solver)), got %zu", P.bytes, a->workspace_bytes);
    }
    x->taps.clear();
    if (x->prof_on) { for (auto& pr : x->prof) { hipEventDestroy(pr.a); hipEventDestroy(pr.b); } x->prof.clear(); x->prof_agg.clear(); }
    auto enqueue = [&]() { return heun ? enqueue_heun(x, a, st) : enqueue_euler(x, a, st); };
    const bool use_graph = a->use_graph && !x->prof_on;
    if (!use_graph) {
        rc = enqueue();
        if (rc) return rc;
        HIPCHK(x, hipGetLastError());
        return DEX_OK;
    }
    // One graph = the whole call.  Every device pointer the captured kernels dereference is part of the key, so a replay
    // is only ever issued against the buffers it was captured with (the host mirror keeps persistent staging buffers, which
    // makes every call of one shape a cache hit).
    if (st == nullptr) return x->fail(DEX_ERR_ARG, "use_graph needs a non-default stream (the legacy null stream cannot be captured)");
    std::vector<uint64_t> key = {(uint64_t)a->B, (uint64_t)a->T, (uint64_t)a->Tr, (uint64_t)a->Ts, (uint64_t)a->n_steps, (uint64_t)a->solver,
                                 (uint64_t)x->precision, (uint64_t)a->n_ref, (uint64_t)(uintptr_t)st,
                                 (uint64_t)(uintptr_t)a->z_dev, (uint64_t)(uintptr_t)a->mu_dev, (uint64_t)(uintptr_t)a->mask_dev,
                                 (uint64_t)(uintptr_t)a->sigmas_dev, (uint64_t)(uintptr_t)a->spk_dev, (uint64_t)(uintptr_t)a->sty_dev,
                                 (uint64_t)(uintptr_t)a->sty_lengths_dev, (uint64_t)(uintptr_t)a->out_dev, (uint64_t)(uintptr_t)a->workspace_dev,
                                 (uint64_t)(uintptr_t)(a->S_churn > 0.f ? a->noise_dev : nullptr)};
    key.push_back((uint64_t)g_xcd_gen.load(std::memory_order_acquire));          // placement rule of the cluster hand-offs (see g_xcd_map)
    for (float v : {a->S_churn, a->S_min, a->S_max, a->S_noise}) { uint32_t u; memcpy(&u, &v, 4); key.push_back(u); }
    for (int j = 0; j < a->n_ref; ++j) key.push_back((uint64_t)(uintptr_t)a->ref_skips_dev[j]);
    for (int v : knobs.v) key.push_back((uint64_t)(uint32_t)v);          // EVERY registered knob, as this call sees it
    DexCtx::GraphEntry* hit = nullptr;
    for (auto& g : x->graphs) if (g.key == key) { hit = &g; break; }
    if (!hit) {
        constexpr size_t MAX_GRAPHS = 8;
        if (x->graphs.size() >= MAX_GRAPHS) {                 // evict the least recently used graph
            size_t lru = 0;
            for (size_t i = 1; i < x->graphs.size(); ++i) if (x->graphs[i].stamp < x->graphs[lru].stamp) lru = i;
            hipGraphExecDestroy(x->graphs[lru].exec);
            x->graphs.erase(x->graphs.begin() + lru);
        }
        hipGraph_t graph = nullptr;
        HIPCHK(x, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = enqueue();
        hipError_t ec = hipStreamEndCapture(st, &graph);          // always leave capture mode
        if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
        if (ec != hipSuccess || !graph) return x->fail(DEX_ERR_HIP, "stream capture failed: %s", hipGetErrorString(ec));
        hipGraphExec_t exec = nullptr;
        ec = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ec != hipSuccess) return x->fail(DEX_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ec));
        x->graphs.push_back({key, exec, 0, x->last_xerr});        // (the Runner set last_xerr while its launches were captured)
        hit = &x->graphs.back();
    }
    x->last_xerr = hit->xerr;           // a replay enqueues nothing on the host: dex_call_status reads the word the CAPTURED launches write
    hit->stamp = ++x->graph_clock;
    HIPCHK(x, hipGraphLaunch(hit->exec, st));
    HIPCHK(x, hipGetLastError());
    return DEX_OK;
}

// ---- taps ---------------------------------------------------------------------------------------
int dex_num_taps(const DexCtx* x) { return x ? (int)x->taps.size() : 0; }
const char* dex_tap_name(const DexCtx* x, int i) { return (x && i >= 0 && i < (int)x->taps.size()) ? x->taps[i].name.c_str() : nullptr; }
int dex_tap_info(const DexCtx* x, const char* name, int64_t shape[4], int* ndim) {
    if (!x || !name) return DEX_ERR_ARG;
    for (const auto& t : x->taps)
        if (t.name == name) { shape[0] = t.shape[0]; shape[1] = t.shape[1]; if (ndim) *ndim = 2; return DEX_OK; }
    return DEX_ERR_ARG;
}
int dex_tap_copy(DexCtx* x, const char* name, float* dst, size_t dst_bytes, dex_stream_t stream) {
    if (!x || !name || !dst) return DEX_ERR_ARG;
    for (const auto& t : x->taps)
        if (t.name == name) {
            const size_t rows = (size_t)t.shape[0], C = (size_t)t.shape[1], ld = (size_t)t.shape[2];
            if (dst_bytes < rows * C * 4) return x->fail(DEX_ERR_ARG, "tap '%s' needs %zu bytes", name, rows * C * 4);
            HIPCHK(x, hipMemcpy2DAsync(dst, C * 4, t.p, ld * 4, C * 4, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream));
            return DEX_OK;
        }
    return x->fail(DEX_ERR_ARG, "unknown tap '%s'", name);
}

// ---- profiling ----------------------------------------------------------------------------------
int dex_profile_enable(DexCtx* x, int on) { if (!x) return DEX_ERR_ARG; x->prof_on = on != 0; return DEX_OK; }
static void prof_aggregate(DexCtx* x) {
    if (!x->prof_agg.empty() || x->prof.empty()) return;
    hipEventSynchronize(x->prof.back().b);
    std::map<std::string, size_t> idx;
    for (auto& pr : x->prof) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, pr.a, pr.b);
        auto it = idx.find(pr.name);
        if (it == idx.end()) { idx[pr.name] = x->prof_agg.size(); x->prof_agg.push_back({pr.name, 0, 0.0, 0.0, 0.0}); it = idx.find(pr.name); }
        ProfAgg& g = x->prof_agg[it->second];
        g.calls += 1; g.ms += ms; g.flops += pr.flops; g.bytes += pr.bytes;
    }
}
int dex_profile_num(const DexCtx* x) { if (!x) return 0; prof_aggregate(const_cast<DexCtx*>(x)); return (int)x->prof_agg.size(); }
int dex_profile_get(const DexCtx* x, int i, const char** name, int* calls, double* total_ms, double* flops, double* bytes) {
    if (!x) return DEX_ERR_ARG;
    prof_aggregate(const_cast<DexCtx*>(x));
    if (i < 0 || i >= (int)x->prof_agg.size()) return DEX_ERR_ARG;
    const ProfAgg& g = x->prof_agg[i];
    if (name) *name = g.name.c_str(); if (calls) *calls = g.calls; if (total_ms) *total_ms = g.ms;
    if (flops) *flops = g.flops; if (bytes) *bytes = g.bytes;
    return DEX_OK;
}

// ---- STFT / mel front-end -------------------------------------------------------------------------
int dex_mel_frames(int n_samples) { return n_samples / 256 + 1; }

// Status of the last dex_sample / dex_denoise_once call of this context on `stream`: waits for the stream, reads the call's hand-off
// word.  DEX_OK, or DEX_ERR_HANDOFF when an in-launch hand-off of the cluster row chain timed out (1) or met a peer on another XCD
// (2) - the call's outputs are NaN in that case.  An XCC mismatch switches the XCD-local form off for the device (all contexts:
// the graph-cache generation moves), so simply repeating the call takes the placement-independent form.
static int handoff_verdict(DexCtx* x, int v) {
    if (v == 0) return DEX_OK;
    if (v == 2 && !knob_set("DEX_DEBUG_DROP_HANDOFF")) { g_xcd_map.store(0); x->drop_graphs(); }
    if (v == 2) return x->fail(DEX_ERR_HANDOFF_XCD, "a cluster hand-off met its peer on another XCD (outputs poisoned); the XCD-local form is now off for this device - repeat the call");
    return x->fail(DEX_ERR_HANDOFF, "a cluster hand-off timed out (outputs poisoned)");
}
int dex_call_status(DexCtx* x, dex_stream_t stream) {
    if (!x) return DEX_ERR_ARG;
    if (!x->last_xerr) return DEX_OK;
    int v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&v, x->last_xerr, sizeof v, hipMemcpyDeviceToHost) != hipSuccess)
        return x->fail(DEX_ERR_HIP, "dex_call_status: could not read the hand-off word");
    return handoff_verdict(x, v);
}
// The same check without blocking the host (the call is asynchronous): _begin enqueues, behind the call on its
// stream, a copy of the hand-off word into a pinned host word of the context and records an event; _poll reads the verdict once that
// event has passed (wait = 0: DEX_PENDING while it has not; wait != 0: waits for THAT event only, not for the stream).
int dex_call_status_begin(DexCtx* x, dex_stream_t stream) {
    if (!x) return DEX_ERR_ARG;
    if (x->st_pending) { const int rc = dex_call_status_poll(x, 1); if (rc != DEX_OK) return rc; }      // (one check in flight per context)
    if (!x->last_xerr) return DEX_OK;
    if (!x->st_host) {
        if (hipHostMalloc(reinterpret_cast<void**>(&x->st_host), sizeof(int), hipHostMallocDefault) != hipSuccess) return x->fail(DEX_ERR_HIP, "dex_call_status_begin: hipHostMalloc failed");
        if (hipEventCreateWithFlags(&x->st_ev, hipEventDisableTiming) != hipSuccess) return x->fail(DEX_ERR_HIP, "dex_call_status_begin: hipEventCreate failed");
    }
    HIPCHK(x, hipMemcpyAsync(x->st_host, x->last_xerr, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(x, hipEventRecord(x->st_ev, (hipStream_t)stream));
    x->st_pending = true;
    return DEX_OK;
}
int dex_call_status_poll(DexCtx* x, int wait) {
    if (!x) return DEX_ERR_ARG;
    if (!x->st_pending) return DEX_OK;
    if (wait) { if (hipEventSynchronize(x->st_ev) != hipSuccess) return x->fail(DEX_ERR_HIP, "dex_call_status_poll: event wait failed"); }
    else {
        const hipError_t q = hipEventQuery(x->st_ev);
        if (q == hipErrorNotReady) return DEX_PENDING;
        if (q != hipSuccess) return x->fail(DEX_ERR_HIP, "dex_call_status_poll: event query failed");
    }
    x->st_pending = false;
    return handoff_verdict(x, *static_cast<volatile int*>(x->st_host));
}

int dex_debug_handoff_timeouts(DexCtx* x, dex_stream_t stream) {
    if (!x) return DEX_ERR_ARG;
    if (!x->last_xerr) return 0;
    int v = 0;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&v, x->last_xerr, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (v == 2 && !knob_set("DEX_DEBUG_DROP_HANDOFF")) { g_xcd_map.store(0); x->drop_graphs(); }        // members of a cluster met on different XCDs: the XCD-local form is off from here on
    return v;
}

// standalone mel context: the DFT basis and the Slaney filterbank, nothing else (a preprocess job needs no score network)
struct DexMel { float* basis = nullptr; float* filt = nullptr; std::string err; };

int dex_mel_create(DexMel** out) {
    if (!out) return DEX_ERR_ARG;
    DexMel* m = new DexMel();
    std::vector<float> basis, filt;
    build_mel_constants(basis, filt);
    if (hipMalloc((void**)&m->basis, basis.size() * sizeof(float)) != hipSuccess || hipMalloc((void**)&m->filt, filt.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(m->basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->filt, filt.data(), filt.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (m->basis) hipFree(m->basis);
        if (m->filt) hipFree(m->filt);
        delete m;
        return DEX_ERR_HIP;
    }
    *out = m;
    return DEX_OK;
}
void dex_mel_destroy(DexMel* m) { if (m) { hipFree(m->basis); hipFree(m->filt); delete m; } }
const char* dex_mel_last_error(const DexMel* m) { return m ? m->err.c_str() : "null mel context"; }

static size_t mel_pad_stride(int n) { return ((size_t)(dex_mel_frames(n) - 1) * 256 + 1024 + 256 + 63) & ~size_t(63); }
size_t dex_mel_workspace_bytes(int B, int n) {
    if (B < 1 || n < 1) return 0;
    return ((size_t)B * mel_pad_stride(n) + (size_t)B * dex_mel_frames(n) * 1152) * sizeof(float) + 512;
}

// the whole front-end for B equally long rows: pad/clip -> windowed DFT as ONE batched implicit GEMM -> magnitude / mel / log
static int mel_enqueue(const float* basis, const float* filt, const float* wav_dev, int B, int n, float* mel_dev, float* energy_dev, void* ws, hipStream_t st) {
    const int frames = dex_mel_frames(n), NP = 1152;
    const size_t pstride = mel_pad_stride(n);
    float* ypad = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* spec = ypad + (size_t)B * pstride;
    launch_wav_pad(wav_dev, n, 512, ypad, (int)pstride, st, B, (long)pstride);
    // per utterance: frames x 1152 = (frames x 1024 overlapping-row view, row stride = hop) x basis[1024][1152]
    IGemmP g{};
    g.A = ypad; g.lda = 256; g.a_bstride = (long)pstride; g.a_coff = 0; g.Hi = 1; g.Wi = frames; g.Cin = 1024;
    g.KH = 1; g.KW = 1; g.sh = 1; g.sw = 1; g.step_h = 1; g.step_w = 1; g.Ho = 1; g.Wo = frames;
    g.W = basis; g.N = NP; g.K = 1024; g.ksplit = 1; g.groups = 1;
    g.C = spec; g.ldc = NP; g.c_bstride = (long)frames * NP; g.OHf = 1; g.OWf = frames; g.osh = 1; g.osw = 1; g.gate_nstride = 1; g.B = B;
    launch_igemm(g, DEX_PREC_FP32, st);
    MagMelP m{spec, NP, 576, frames, 513, filt, 80, mel_dev, energy_dev};
    launch_magmel(m, st, B);
    return hipGetLastError() == hipSuccess ? DEX_OK : DEX_ERR_HIP;
}

int dex_mel_spectrogram(DexMel* m, const float* wav_dev, int B, int n, float* mel_dev, float* energy_dev, void* workspace_dev,
                        size_t workspace_bytes, dex_stream_t stream) {
    if (!m) return DEX_ERR_ARG;
    auto fail = [&](const char* msg) { m->err = msg; return DEX_ERR_ARG; };
    if (!wav_dev || !mel_dev || !energy_dev || !workspace_dev) return fail("null pointer argument");
    if (B < 1) return fail("B must be >= 1");
    if (n < 513) return fail("reflect padding needs at least 513 samples per row");
    if (workspace_bytes < dex_mel_workspace_bytes(B, n)) return fail("workspace too small (dex_mel_workspace_bytes)");
    const int rc = mel_enqueue(m->basis, m->filt, wav_dev, B, n, mel_dev, energy_dev, workspace_dev, (hipStream_t)stream);
    if (rc) m->err = "kernel launch failed";
    return rc;
}

int dex_lf0_normalize(const float* f0_dev, const int* lengths_dev, int B, int T, float* lf0_dev, dex_stream_t stream) {
    if (!f0_dev || !lf0_dev || B < 1 || T < 1 || T > 16382) return DEX_ERR_ARG;       // (the voiced frames of an utterance are staged in LDS: 4 T + 8 bytes must fit the 64 KB a launch gets without an attribute)
    launch_lf0_normalize(f0_dev, lengths_dev, B, T, lf0_dev, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? DEX_OK : DEX_ERR_HIP;
}

int dex_mel_from_wav(DexCtx* x, const float* wav_dev, int n, float* mel_dev, float* energy_dev, dex_stream_t stream) {
    if (!x || !wav_dev || !mel_dev || !energy_dev) return DEX_ERR_ARG;
    if (n < 513) return x->fail(DEX_ERR_ARG, "reflect padding needs at least 513 samples (got %d)", n);
    hipStream_t st = (hipStream_t)stream;
    if (!x->mel_basis) {
        std::vector<float> basis, filt;
        build_mel_constants(basis, filt);
        HIPCHK(x, hipMalloc((void**)&x->mel_basis, basis.size() * sizeof(float)));
        HIPCHK(x, hipMalloc((void**)&x->mel_filt, filt.size() * sizeof(float)));
        HIPCHK(x, hipMemcpy(x->mel_basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(x, hipMemcpy(x->mel_filt, filt.data(), filt.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    const size_t need = dex_mel_workspace_bytes(1, n);
    if (need > x->mel_ws_bytes) {       // grow-only scratch, single-stream use (documented in include/dex_amd.h)
        if (x->mel_ws) { HIPCHK(x, hipStreamSynchronize(st)); hipFree(x->mel_ws); }
        x->mel_ws = nullptr; x->mel_ws_bytes = 0;
        HIPCHK(x, hipMalloc(&x->mel_ws, need));
        x->mel_ws_bytes = need;
    }
    if (mel_enqueue(x->mel_basis, x->mel_filt, wav_dev, 1, n, mel_dev, energy_dev, x->mel_ws, st)) return x->fail(DEX_ERR_HIP, "mel front-end launch failed");
    return DEX_OK;
}

}  // extern "C"
