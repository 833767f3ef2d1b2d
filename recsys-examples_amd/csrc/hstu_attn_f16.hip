// fp16 operands of the HSTU attention kernels: the second translation unit of hstu_attn.hip (see the note on HSTU_F16 at
// its head).  Same kernels, masks, tilings and knobs; v_mfma_f32_32x32x16_f16, v_cvt_pk_f16_f32 and an fp16 read of the
// bias instead of their bf16 counterparts; entry points mi355_hstu_attn_{fwd,fwd_kv,bwd,fwd_window,bwd_window,fwd_rab,
// bwd_rab}_f16 (include/recsys_amd.h).  Reference: hstu_api.cpp:359-366 (q_dtype == fp16 || bf16).
#define HSTU_F16 1
#include "hstu_attn.hip"
