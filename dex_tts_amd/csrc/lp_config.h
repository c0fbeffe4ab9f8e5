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
#elif defined(DEX_LP_WSPLIT)
// third build: fp16 operands with every WEIGHT as hi + lo (w = fp16(w) + fp16(w - fp16(w)), two MFMAs per product, activations rounded
// once) - the "fp16x2" mode: the weight rounding is what separates the fp16 mode from the fp32 reference over a 50-step sampler
// (oracle/lowp_emulate.py).  Compiled with -DDEX_LP_F16 -DDEX_LP_WSPLIT; a packed weight twin is the hi pack followed by the lo pack
// in the same layout.  Kernels without a split form answer false from their shape predicates in this namespace.
#define DEX_LP_NS f16w
// the fused kernels that have a split-weight form (their shape predicates answer as in the other builds)
#define DEX_WS_HAVE_CONV3 1
#define DEX_WS_HAVE_CONV3_RES 1
#define DEX_WS_HAVE_LINATTN 1
#define DEX_WS_HAVE_RC 1
#define DEX_WS_HAVE_RC64 1
#define DEX_WS_HAVE_RCC 1
#define DEX_WS_HAVE_PE 1
#define DEX_WS_HAVE_POS 1
#define DEX_WS_HAVE_DOWN 1
#define DEX_WS_HAVE_UP 1
#define DEX_WS_HAVE_NWALK 1
#elif defined(DEX_LP_F16)
#define DEX_LP_NS f16
#else
#define DEX_LP_NS bf16
#endif
