// lp_config.h — which reduced-precision operand type a "low-precision" translation unit is compiled for.
// Every MFMA-heavy kernel file of the reduced-precision modes (conv3x3_bf16, conv3x3_stream, igemm_bf16, attention_bf16,
// attention_direct, dit_rowchain, linattn_fused, pos_conv) is compiled TWICE by dex_tts_amd/build.py: as is (operands
// bf16, v_mfma_f32_32x32x16_bf16, namespace dex::bf16) and with -DDEX_LP_F16 (operands fp16, v_mfma_f32_32x32x16_f16,
// namespace dex::f16).  Both have the same fragment layouts, one-instruction packed converts (v_cvt_pk_bf16_f32 /
// v_cvt_pk_f16_f32) and fp32 accumulation; lp_dispatch.hip picks the namespace from the context's precision mode.
#pragma once
#if defined(DEX_LP_NS_OVERRIDE)
// (bench tools that compile a kernel file of their own next to the library's copy: a namespace - and so a kernel symbol - of their own;
// two modules registering one kernel name resolve to whichever the runtime saw first)
#define DEX_LP_NS DEX_LP_NS_OVERRIDE
#elif defined(DEX_LP_F16)
#define DEX_LP_NS f16
#else
#define DEX_LP_NS bf16
#endif
