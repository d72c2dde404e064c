// l1_regs_prims.h -- pieces shared by the register-resident first-stage kernels (k_layer1_regs: Silero v3.1, k_layer1_regs_v4: Silero v4):
// K = 16 products as two K = 32 matrix instructions, and the depthwise conv of one channel in two overlapping column tiles.
// Device code only; include inside a .hip translation unit after enc_regs_prims.h.
#pragma once
#include "enc_regs_prims.h"

namespace vadc {

typedef __attribute__((address_space(3))) void l1_lds_void_t;

// ---- K = 16 GEMMs of the transformer block -------------------------------------------------------------------------------------------------
// The three split terms of a K = 16 product as TWO v_mfma_f32_16x16x32_f16 instead of three v_mfma_f32_16x16x16_f16 (which occupy the pipe just as
// long): the instruction's 32 k slots are the 16 channels twice.  With operand B as (hi, lo) of a lane's four channels (HL),
//   A as (lo, hi) (LH) gives  a.lo . b.hi + a.hi . b.lo   -- both cross terms,
//   A as (hi, 0)  (H0) gives  a.hi . b.hi                 -- on the same accumulator, the same instruction back to back.
// (A K = 16 instruction for hi . hi behind the K = 32 one on the same accumulator returned stale accumulators: hipcc pads no wait states between
// the two shapes, and the hardware forwards an accumulator only between equal ones.)  Weights are stored in both forms (two 16-byte LDS reads per
// lane); an activation that is the A operand of a product (Q, V^T) is split into both, one that is the B operand into HL.
struct AOp { h8 lh, h0; };
__device__ __forceinline__ AOp lds_aop(const char *base, int lane)
{
   const h8 *p = reinterpret_cast<const h8 *>(base);
   AOp w;
   w.lh = p[lane];
   w.h0 = p[64 + lane];
   return w;
}
__device__ __forceinline__ h8 split4_hl(const f4 &u) { const Frag4 f = split4(u); return __builtin_shufflevector(f.hi, f.lo, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ AOp split4_a(const f4 &u)
{
   const Frag4 f = split4(u);
   const h4 z = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
   AOp r;
   r.lh = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
   r.h0 = __builtin_shufflevector(f.hi, z, 0, 1, 2, 3, 4, 5, 6, 7);
   return r;
}
// c += A . B
__device__ __forceinline__ f4 mm(const AOp &a, const h8 &b, f4 c)
{
   c = MFMA16(a.lh, b, c);
   return MFMA16(a.h0, b, c);
}
// c += X^T-style product with the ACTIVATION x (HL) as the A operand and the weight as B: x.hi . w.lo + x.lo . w.hi, then x.hi . w.hi + x.lo . 0
__device__ __forceinline__ f4 mmt(const h8 &x, const AOp &w, f4 c)
{
   c = MFMA16(x, w.lh, c);
   return MFMA16(x, w.h0, c);
}
__device__ __forceinline__ h8 keep(const h8 &v, bool k)
{
   typedef int i4 __attribute__((ext_vector_type(4)));
   i4 u = __builtin_bit_cast(i4, v);
#pragma unroll
   for (int i = 0; i < 4; ++i) u[i] = k ? u[i] : 0;
   return __builtin_bit_cast(h8, u);
}

// depthwise k = 5, zero pad 2, + bias (conv.c:17-53) of one channel in BOTH tiles: the time neighbours are row shifts of x itself, carried by the
// multiply-adds as their DPP operand (enc_regs_prims.h: dw5); the two tiles' chains alternate.  The centre tap goes first, as a three-address
// multiply-add onto the bias (the taps come out of LDS as float4s: accumulating into the bias's own register costs a copy per channel), then taps
// -2, -1, +1, +2; the centre tap's two instructions also are the VALU-write -> DPP-read wait states of x.
__device__ __forceinline__ void dw5x2(float x0, float x1, float k0, float k1, float k2, float k3, float k4, float bias, float &d0, float &d1)
{
   asm("v_fma_f32 %0, %2, %6, %9\n\t"
       "v_fma_f32 %1, %3, %6, %9\n\t"
       "v_fmac_f32_dpp %0, %2, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %1, %3, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %0, %2, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %1, %3, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %0, %2, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %1, %3, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %0, %2, %8 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
       "v_fmac_f32_dpp %1, %3, %8 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
       : "=&v"(d0), "=&v"(d1) : "v"(x0), "v"(x1), "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(k4), "v"(bias));
}
}  // namespace vadc
