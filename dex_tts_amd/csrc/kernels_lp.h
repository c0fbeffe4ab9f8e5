// kernels_lp.h — launch interfaces of the low-precision (bf16 / fp16 operand) kernel files, declared inside
// namespace dex::DEX_LP_NS.  No include guard on purpose: lp_dispatch.hip includes it once per namespace.
// The parameter structs are the shared ones of kernels.h.
#include "kernels.h"
#include "lp_config.h"

namespace dex {
namespace DEX_LP_NS {

// Patch-staged 3x3/s1/p1 Block convolution (conv3x3_bf16.hip) and its strip-streaming form (conv3x3_stream.hip)
bool conv3x3_bf16_supported(int Cin, int Cout);
bool conv3x3_bf16_tail_supported(int C);              // pro_res form (Cin == Cout == C)
bool conv3x3_bf16_res_supported(int Cin, int Cout);   // res_w form (fused 1x1 shortcut)
bool conv3x3_bf16_xb_supported(int Cin, int Cout);    // x_bf16 form (low-precision input under a GroupNorm prologue)
void launch_conv3x3_lp(const Conv3P& p, hipStream_t st);
int conv3x3_stream_tiles(const Conv3P& p);            // iterations per workgroup of the throughput form, 0 = not applicable
void launch_conv3x3_stream(const Conv3P& p, int tiles_per_wg, hipStream_t st);
bool conv3x3_regw_form(const Conv3P& p);                 // weights-in-registers strip form (conv3x3_regw.hip) takes this launch
void launch_conv3x3_regw(const Conv3P& p, hipStream_t st);
bool linattn_out2_lp_out_supported(int npix, int B);
bool linattn_fused_supported(int C);
bool conv3x3_plain_lp_in_supported(int H, int W, int B, int Cin, int Cout);
bool conv3x3_res2_form(int H, int W, int B);          // the ping-pong strip form (the only one that implements Conv3P::res2_*) takes this grid
// implicit GEMM on the low-precision MFMA (igemm_bf16.hip)
void launch_igemm_lp(const IGemmP& p, hipStream_t st);
bool attention_lp_shared_form(int Nq, int heads, int B, int ksplit);
bool igemm_nwalk_form(const IGemmP& p);               // the column-walking unpatchify GEMM takes this launch (the form that implements c_lp for the scatter)
bool conv3x3_cat_lp_in_supported(int H, int W, int B, int Cin, int Cout);   // the fused-shortcut conv of the up path has a 16-bit-input form at this grid
// Upsample (ConvTranspose2d 4/2/1) strip kernel (convt_up.hip)
bool convt_up_supported(int C, int H, int W, int ldx, int ldy);
void launch_convt_up(const ConvTUpP& p, hipStream_t st);
// Downsample (Conv2d 3/2/1) strip kernel (conv_down.hip)
bool conv_down_supported(int C, int H, int W, int ldx, int ldy, int x_coff);
void launch_conv_down(const ConvDownP& p, hipStream_t st);
// softmax attention (attention_bf16.hip: fp32 q/k/v in HBM; attention_direct.hip: fragment-ordered operands)
void launch_attention_lp(const AttnP& p, hipStream_t st);
void launch_attention_direct(const AttnDirectP& p, hipStream_t st);
// the DEX TV adaptor as one launch (attention_bf16.hip)
bool tv_chain_form(int npix, int C, int B);
void launch_tv_kv_prep(const TvKvPrepP& p, hipStream_t st);
void launch_tv_vfrag_prep(const TvKvPrepP& p, hipStream_t st);
void launch_tv_chain(const TvChainP& p, hipStream_t st);
bool attention_direct_batch_regime(int N, int B);
int attention_direct_ksplit(int N, int B);
// 64-queries-per-wave form (attention_q64.hip): 4 waves x 64 queries, persistent work units, up to max_split key splits
void launch_attention_q64(const AttnDirectP& p, hipStream_t st);
int attention_q64_ksplit(int N, int B, int max_split);
void attention_q64_plan(int N, int B, int max_split, int* ks, int* tail_g, int* tail_ks);
bool attention_q64_half_plan(int N, int B, int* half_g, int* half_n);
// DiT row chain (dit_rowchain.hip) and its weight packing
bool dit_rowchain_supported(int hidden, int mlp_hidden);
bool dit_rowchain64_form(int rows_per_batch, int B, int attn_inline);
bool dit_rowchain_cluster_form(int rows_per_batch, int B);
bool dit_rowchain_cluster_local_fits(int rows_per_batch, int B);
int dit_rowchain_cluster_xcds(int rows_per_batch, int B);
void launch_dit_rowchain(const DitChainP& p, hipStream_t st);
void launch_pack_lp_frag(const float* src, void* dst, int K, int N, hipStream_t st);
void launch_pack_lp_frag_nk(const float* src, void* dst, int K, int N, hipStream_t st);   // source [N][K]
// fused linear attention (linattn_fused.hip)
void launch_linattn_kvctx(const LinKvCtxP& p, hipStream_t st);
void launch_linattn_merge(const LinMergeP& p, hipStream_t st);
void launch_linattn_out2(const LinOut2P& p, hipStream_t st);
// PatchEmbed2D as one launch at small grids (patch_embed.hip)
bool patch_embed_fused_supported(int k, int C, int hid, long ntok);
void launch_patch_embed_fused(const DwConvP& p, const void* Wb, const float* bias, float* emb, int hid, hipStream_t st);
// direct grouped positional convolution (pos_conv.hip)
bool pos_conv_direct_supported(int hid, int groups, int kernel, int Hf);
void launch_pos_conv_direct(const PosConvP& p, hipStream_t st);

}  // namespace DEX_LP_NS
}  // namespace dex
