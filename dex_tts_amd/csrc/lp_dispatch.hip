// lp_dispatch.hip — the reduced-precision kernels exist once per operand type (namespace dex::bf16 and dex::f16: the same
// sources compiled twice, lp_config.h); these are the precision-taking launch functions of kernels.h that pick one.
#include "kernels.h"

#include "kernels_lp.h"          // namespace dex::bf16
#undef DEX_LP_NS
#define DEX_LP_NS f16
#include "kernels_lp.h"          // namespace dex::f16
#undef DEX_LP_NS
#define DEX_LP_NS f16w
#include "kernels_lp.h"          // namespace dex::f16w (weights as hi + lo)
#undef DEX_LP_NS

namespace dex {

thread_local const char* g_last_symbol = nullptr;
thread_local bool g_lp_wsplit = false;

// shape predicates do not depend on the operand type - but the split-weight build (f16w) has its own: a kernel without a split form
// answers false there and the caller takes the next form down (generic implicit GEMM, exact fp32)
bool conv3x3_bf16_supported(int Cin, int Cout) { return (g_lp_wsplit ? f16w::conv3x3_bf16_supported(Cin, Cout) : bf16::conv3x3_bf16_supported(Cin, Cout)); }
bool conv3x3_bf16_tail_supported(int C) { return (g_lp_wsplit ? f16w::conv3x3_bf16_tail_supported(C) : bf16::conv3x3_bf16_tail_supported(C)); }
bool conv3x3_bf16_res_supported(int Cin, int Cout) { return (g_lp_wsplit ? f16w::conv3x3_bf16_res_supported(Cin, Cout) : bf16::conv3x3_bf16_res_supported(Cin, Cout)); }
bool conv3x3_bf16_xb_supported(int Cin, int Cout) { return (g_lp_wsplit ? f16w::conv3x3_bf16_xb_supported(Cin, Cout) : bf16::conv3x3_bf16_xb_supported(Cin, Cout)); }
bool conv3x3_plain_lp_in_supported(int H, int W, int B, int Cin, int Cout) { return (g_lp_wsplit ? f16w::conv3x3_plain_lp_in_supported(H, W, B, Cin, Cout) : bf16::conv3x3_plain_lp_in_supported(H, W, B, Cin, Cout)); }
bool linattn_out2_lp_out_supported(int npix, int B) { return (g_lp_wsplit ? f16w::linattn_out2_lp_out_supported(npix, B) : bf16::linattn_out2_lp_out_supported(npix, B)); }
bool linattn_fused_supported(int C) { return (g_lp_wsplit ? f16w::linattn_fused_supported(C) : bf16::linattn_fused_supported(C)); }
bool conv3x3_res2_form(int H, int W, int B) { return (g_lp_wsplit ? f16w::conv3x3_res2_form(H, W, B) : bf16::conv3x3_res2_form(H, W, B)); }
bool attention_lp_shared_form(int Nq, int heads, int B, int ksplit) { return (g_lp_wsplit ? f16w::attention_lp_shared_form(Nq, heads, B, ksplit) : bf16::attention_lp_shared_form(Nq, heads, B, ksplit)); }
bool igemm_nwalk_form(const IGemmP& p) { return (g_lp_wsplit ? f16w::igemm_nwalk_form(p) : bf16::igemm_nwalk_form(p)); }
bool conv3x3_cat_lp_in_supported(int H, int W, int B, int Cin, int Cout) { return (g_lp_wsplit ? f16w::conv3x3_cat_lp_in_supported(H, W, B, Cin, Cout) : bf16::conv3x3_cat_lp_in_supported(H, W, B, Cin, Cout)); }
bool conv3x3_strip_form(const Conv3P& p) { return (g_lp_wsplit ? f16w::conv3x3_stream_tiles(p) : bf16::conv3x3_stream_tiles(p)) != 0 || (g_lp_wsplit ? f16w::conv3x3_regw_form(p) : bf16::conv3x3_regw_form(p)); }
bool pos_conv_direct_supported(int hid, int groups, int kernel, int Hf) { return (g_lp_wsplit ? f16w::pos_conv_direct_supported(hid, groups, kernel, Hf) : bf16::pos_conv_direct_supported(hid, groups, kernel, Hf)); }
bool tv_chain_form(int npix, int C, int B) { return (g_lp_wsplit ? f16w::tv_chain_form(npix, C, B) : bf16::tv_chain_form(npix, C, B)); }
bool attention_direct_batch_regime(int N, int B) { return (g_lp_wsplit ? f16w::attention_direct_batch_regime(N, B) : bf16::attention_direct_batch_regime(N, B)); }
int attention_direct_ksplit(int N, int B) { return (g_lp_wsplit ? f16w::attention_direct_ksplit(N, B) : bf16::attention_direct_ksplit(N, B)); }
int attention_q64_ksplit(int N, int B, int max_split) { return (g_lp_wsplit ? f16w::attention_q64_ksplit(N, B, max_split) : bf16::attention_q64_ksplit(N, B, max_split)); }
void attention_q64_plan(int N, int B, int max_split, int* ks, int* tail_g, int* tail_ks) { (g_lp_wsplit ? f16w::attention_q64_plan(N, B, max_split, ks, tail_g, tail_ks) : bf16::attention_q64_plan(N, B, max_split, ks, tail_g, tail_ks)); }
bool conv_down_supported(int C, int H, int W, int ldx, int ldy, int x_coff) { return (g_lp_wsplit ? f16w::conv_down_supported(C, H, W, ldx, ldy, x_coff) : bf16::conv_down_supported(C, H, W, ldx, ldy, x_coff)); }
bool convt_up_supported(int C, int H, int W, int ldx, int ldy) { return (g_lp_wsplit ? f16w::convt_up_supported(C, H, W, ldx, ldy) : bf16::convt_up_supported(C, H, W, ldx, ldy)); }
bool patch_embed_fused_supported(int k, int C, int hid, long ntok) { return (g_lp_wsplit ? f16w::patch_embed_fused_supported(k, C, hid, ntok) : bf16::patch_embed_fused_supported(k, C, hid, ntok)); }
bool dit_rowchain_supported(int hidden, int mlp_hidden) { return (g_lp_wsplit ? f16w::dit_rowchain_supported(hidden, mlp_hidden) : bf16::dit_rowchain_supported(hidden, mlp_hidden)); }
bool dit_rowchain64_form(int rows_per_batch, int B, int attn_inline) { return (g_lp_wsplit ? f16w::dit_rowchain64_form(rows_per_batch, B, attn_inline) : bf16::dit_rowchain64_form(rows_per_batch, B, attn_inline)); }
bool dit_rowchain_cluster_form(int rows_per_batch, int B) { return (g_lp_wsplit ? f16w::dit_rowchain_cluster_form(rows_per_batch, B) : bf16::dit_rowchain_cluster_form(rows_per_batch, B)); }
int dit_rowchain_cluster_xcds(int rows_per_batch, int B) { return bf16::dit_rowchain_cluster_xcds(rows_per_batch, B); }
bool dit_rowchain_cluster_local_fits(int rows_per_batch, int B) { return (g_lp_wsplit ? f16w::dit_rowchain_cluster_local_fits(rows_per_batch, B) : bf16::dit_rowchain_cluster_local_fits(rows_per_batch, B)); }

#define DEX_LP_CALL(fn, ...) do { if (precision == PREC_FP16X2) f16w::fn(__VA_ARGS__); else if (precision == PREC_FP16) f16::fn(__VA_ARGS__); else bf16::fn(__VA_ARGS__); } while (0)

void launch_conv3x3_lp(const Conv3P& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_conv3x3_lp, p, st); }
void launch_igemm_lp(const IGemmP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_igemm_lp, p, st); }
void launch_convt_up(const ConvTUpP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_convt_up, p, st); }
void launch_conv_down(const ConvDownP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_conv_down, p, st); }
void launch_attention_lp(const AttnP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_attention_lp, p, st); }
void launch_tv_kv_prep(const TvKvPrepP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_tv_kv_prep, p, st); }
void launch_tv_vfrag_prep(const TvKvPrepP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_tv_vfrag_prep, p, st); }
void launch_tv_chain(const TvChainP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_tv_chain, p, st); }
void launch_attention_direct(const AttnDirectP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_attention_direct, p, st); }
bool attention_q64_half_plan(int N, int B, int* half_g, int* half_n) { return bf16::attention_q64_half_plan(N, B, half_g, half_n); }
void launch_attention_q64(const AttnDirectP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_attention_q64, p, st); }
void launch_dit_rowchain(const DitChainP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_dit_rowchain, p, st); }
void launch_pack_lp_frag(const float* src, void* dst, int K, int N, int precision, hipStream_t st) { DEX_LP_CALL(launch_pack_lp_frag, src, dst, K, N, st); }
void launch_pack_lp_frag_nk(const float* src, void* dst, int K, int N, int precision, hipStream_t st) { DEX_LP_CALL(launch_pack_lp_frag_nk, src, dst, K, N, st); }
void launch_linattn_kvctx(const LinKvCtxP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_linattn_kvctx, p, st); }
void launch_linattn_merge(const LinMergeP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_linattn_merge, p, st); }
void launch_linattn_out2(const LinOut2P& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_linattn_out2, p, st); }
void launch_pos_conv_direct(const PosConvP& p, int precision, hipStream_t st) { DEX_LP_CALL(launch_pos_conv_direct, p, st); }
void launch_patch_embed_fused(const DwConvP& p, const void* Wb, const float* bias, float* emb, int hid, int precision, hipStream_t st) { DEX_LP_CALL(launch_patch_embed_fused, p, Wb, bias, emb, hid, st); }

}  // namespace dex
