// kernels_frontend.hip -- reflect pad + STFT (strided conv with the [258,1,256] basis, hop 64) +
// magnitude + log1p(2^20 x) + per-frame bin mean, for gfx950.
//
// Replaces, per chunk (reference file:line):
//   tensor_reflect_pad_last_dim_lr  tensor.h:912-958
//   my_stft_                        stft.c:15-224   (AVX2 loop :82-190, magnitude :194-213)
//   first two loops of adaptive_audio_normalization_inplace  misc.c:40-63
//
// NUMERICS.  The parity of the whole engine is decided here: the reference's
// 256-tap dot product is a fixed fp32 tree with separately rounded products and sums
//     tap t = 64 i + 8 j + l ;  g_i[l] = ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7))  over j
//     v[l] = (g_0[l] + g_1[l]) + (g_2[l] + g_3[l]) ;  y = ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7))
// and this kernel evaluates exactly that expression: this file is compiled with -ffp-contract=off and
// carries `#pragma clang fp contract(off)`; no FMA, no MFMA, no reassociation.  The output magnitudes are
// bit-identical to the reference's (tests/test_gpu_parity.py::test_stft_magnitude_bit_exact).
//
// MAPPING (MI355X-first, not the reference's loop nest).  hop == 64 == the tree's group size, so with the
// padded chunk cut into 28 blocks of 64 samples, frame n group i multiplies block n+i with basis taps
// [64i, 64i+64).  One LANE owns one block: its 64 samples stay in 64 VGPRs for the whole kernel, and for
// every filter it evaluates G_i[l] for i = 0..3 (4 x (64 mul + 56 add)).  The basis value is the same for
// every lane of the wave, so it is fetched with SCALAR loads (s_load_dwordx8 from the L2-resident, host-
// permuted basis) and used as the SGPR operand of v_mul_f32: the inner loop has no vector memory and no
// LDS traffic at all.  Frame n then needs G_0(n), G_1(n+1), G_2(n+2), G_3(n+3): three wave shifts.
//     t01 = G_0 + shift1(G_1) ; t23 = G_2 + shift1(G_3) ; v = t01 + shift2(t23)
// Waves overlap by 3 halo lanes (61 producing lanes per wave), blocks 25..27 of a chunk only feed their
// neighbours: lane efficiency 61/64 * 25/28 = 85 %.  ~100 VGPRs => 4-5 waves/SIMD to cover scalar-load
// latency.  Algorithmic work: 258*25*(256 mul + 255 add) = 3.30 M VALU lane-ops per chunk, which bounds this
// kernel at 1/2 of the FMA peak by construction (2 instructions per MAC).
// The shipped instantiation is PK = 2: products two at a time with v_pk_mul_f32 and an SGPR pair, and the tree adds of tree
// lanes (l, l+1) two at a time with v_pk_add_f32 (component-wise the same IEEE operations in the same order): every VALU
// instruction costs ~4.6 cycles per wave64 here whether packed or not (PMC), so instruction count is what matters.
#include "common.h"
#include <type_traits>

#pragma clang fp contract(off)

namespace vadc {

constexpr int kLanesOut = 61;   // producing lanes per wave; lanes 61..63 are halo for the shifts

__device__ __forceinline__ float sample_to_f32(float v) { return v; }
__device__ __forceinline__ float sample_to_f32(int16_t v) { return (float)v * (1.0f / 32768.0f); }  // exact

typedef float f16v __attribute__((ext_vector_type(16)));

// Software-pipelined scalar loads of the basis.  hipcc serialises `s_load -> s_waitcnt -> use` through one
// SGPR tuple when left alone, exposing the full L2 latency every 32 VALU instructions; here the next 32 taps
// are requested (two s_load_dwordx16 into the OTHER buffer) before the current 32 are consumed, and the
// wait sits behind the arithmetic.  SMEM returns out of order, so the wait is always lgkmcnt(0); tying the
// buffers to the wait statement ("+s") keeps every consumer behind it (hipcc does not track asm loads).
#define VADC_SLOAD32(A, B, BASE, BYTEOFF)                                               \
   asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"              \
                : "=&s"(A), "=&s"(B)                                                     \
                : "s"(BASE), "n"(BYTEOFF), "n"((BYTEOFF) + 64));                          \
   __builtin_amdgcn_sched_barrier(0)   /* keep the consumer arithmetic BEHIND the request */
#define VADC_SWAIT(A, B)                                                                \
   asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(A), "+s"(B));                             \
   __builtin_amdgcn_sched_barrier(0)

// G[l] = ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)),  p_j = x[8j+l] * k[j]   (stft.c:141-160)
// KV/O: the 8 basis taps of this (i,l) in j order inside a 16-float SGPR tuple.
// ABL (tools/fe_bench.hip only; 0 in the product): 1 = multiply by a VGPR instead of the SGPR tap, 2 = products only
// (no tree adds), 3 = tree adds only (no products) -- timing ablations, results are garbage for ABL != 0
#define VADC_PROD(J, L, KV, O) (ABL == 1 ? x[(J) * 8 + (L)] * vk : (ABL == 3 ? x[(J) * 8 + (L)] : x[(J) * 8 + (L)] * KV[(O) + (J)]))
#define VADC_TREE8(dst, L, KV, O)                                                       \
   {                                                                                    \
      const float p0 = VADC_PROD(0, L, KV, O), p1 = VADC_PROD(1, L, KV, O);             \
      const float p2 = VADC_PROD(2, L, KV, O), p3 = VADC_PROD(3, L, KV, O);             \
      const float p4 = VADC_PROD(4, L, KV, O), p5 = VADC_PROD(5, L, KV, O);             \
      const float p6 = VADC_PROD(6, L, KV, O), p7 = VADC_PROD(7, L, KV, O);             \
      if (ABL == 2) {                                                                   \
         asm volatile("" ::"v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)); \
         dst = p0;                                                                      \
      } else {                                                                          \
         const float p01 = p0 + p1, p23 = p2 + p3, p45 = p4 + p5, p67 = p6 + p7;        \
         const float p0123 = p01 + p23, p4567 = p45 + p67;                              \
         dst = p0123 + p4567;                                                           \
      }                                                                                 \
   }

// PK = 1: the two products of tree lanes (l, l+1) for the same j come from ONE v_pk_mul_f32 whose second operand is
// an even-aligned SGPR pair.  Measured on gfx950 (tools/valu_rate.hip, tools/fe_bench.hip): v_mul_f32 with an SGPR
// operand issues at ~4.3 cycles, v_pk_mul_f32 with an SGPR pair at ~4.7 for TWO products, VGPR-only v_add at ~2; the
// 256 products per output therefore cost 128 x 4.7 instead of 256 x 4.3 cycles.  Each product is still one IEEE
// fp32 multiply and the tree is unchanged, so the result is bit-identical.  Basis layout for PK: [i][l/2][j][l%2].
typedef float f2v __attribute__((ext_vector_type(2)));
#define VADC_TREE8_PK(dstA, dstB, LP, KV)                                                        \
   {                                                                                             \
      const f2v q0 = x2[0 * 4 + (LP)] * (f2v){KV[0], KV[1]},   q1 = x2[1 * 4 + (LP)] * (f2v){KV[2], KV[3]};   \
      const f2v q2 = x2[2 * 4 + (LP)] * (f2v){KV[4], KV[5]},   q3 = x2[3 * 4 + (LP)] * (f2v){KV[6], KV[7]};   \
      const f2v q4 = x2[4 * 4 + (LP)] * (f2v){KV[8], KV[9]},   q5 = x2[5 * 4 + (LP)] * (f2v){KV[10], KV[11]}; \
      const f2v q6 = x2[6 * 4 + (LP)] * (f2v){KV[12], KV[13]}, q7 = x2[7 * 4 + (LP)] * (f2v){KV[14], KV[15]}; \
      {                                                                                          \
         const float a01 = q0.x + q1.x, a23 = q2.x + q3.x, a45 = q4.x + q5.x, a67 = q6.x + q7.x; \
         const float a0123 = a01 + a23, a4567 = a45 + a67;                                       \
         dstA = a0123 + a4567;                                                                   \
      }                                                                                          \
      {                                                                                          \
         const float b01 = q0.y + q1.y, b23 = q2.y + q3.y, b45 = q4.y + q5.y, b67 = q6.y + q7.y; \
         const float b0123 = b01 + b23, b4567 = b45 + b67;                                       \
         dstB = b0123 + b4567;                                                                   \
      }                                                                                          \
   }
#define VADC_STAGE_PK(G, LB, CA, CB_)                                                            \
   VADC_TREE8_PK(G[(LB) + 0], G[(LB) + 1], (LB) / 2, CA) VADC_TREE8_PK(G[(LB) + 2], G[(LB) + 3], (LB) / 2 + 1, CB_)
// The same two trees with the seven adds per level kept PACKED as well (v_pk_add_f32 on the (l, l+1) pair): component-wise
// identical arithmetic, half the add instructions.  G2[i] = (G[2i], G[2i+1]).
#define VADC_TREE8_PK2(dst2, LP, KV)                                                             \
   {                                                                                             \
      const f2v q0 = x2[0 * 4 + (LP)] * (f2v){KV[0], KV[1]},   q1 = x2[1 * 4 + (LP)] * (f2v){KV[2], KV[3]};   \
      const f2v q2 = x2[2 * 4 + (LP)] * (f2v){KV[4], KV[5]},   q3 = x2[3 * 4 + (LP)] * (f2v){KV[6], KV[7]};   \
      const f2v q4 = x2[4 * 4 + (LP)] * (f2v){KV[8], KV[9]},   q5 = x2[5 * 4 + (LP)] * (f2v){KV[10], KV[11]}; \
      const f2v q6 = x2[6 * 4 + (LP)] * (f2v){KV[12], KV[13]}, q7 = x2[7 * 4 + (LP)] * (f2v){KV[14], KV[15]}; \
      const f2v a01 = q0 + q1, a23 = q2 + q3, a45 = q4 + q5, a67 = q6 + q7;                      \
      const f2v a0123 = a01 + a23, a4567 = a45 + a67;                                            \
      dst2 = a0123 + a4567;                                                                      \
   }
#define VADC_STAGE_PK2(G2, LB, CA, CB_)                                                          \
   VADC_TREE8_PK2(G2[(LB) / 2], (LB) / 2, CA) VADC_TREE8_PK2(G2[(LB) / 2 + 1], (LB) / 2 + 1, CB_)

// one pipeline stage = 32 taps = lanes-of-the-tree l in [LB, LB+4) of one 64-tap group
#define VADC_STAGE(G, LB, CA, CB_)                                                      \
   VADC_TREE8(G[(LB) + 0], (LB) + 0, CA, 0) VADC_TREE8(G[(LB) + 1], (LB) + 1, CA, 8)    \
   VADC_TREE8(G[(LB) + 2], (LB) + 2, CB_, 0) VADC_TREE8(G[(LB) + 3], (LB) + 3, CB_, 8)

// lane i <- lane i+1 (0 shifted in at lane 63).  SHIFT = 0: ds_bpermute (__shfl_down); SHIFT = 1: DPP
// wave_shl:1 folded into the consuming v_add_f32 (no LDS-queue traffic, no lgkmcnt wait).
template <int SHIFT>
__device__ __forceinline__ float lane_up1(float v)
{
   if (SHIFT == 0) return __shfl_down(v, 1);
   return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
template <int SHIFT>
__device__ __forceinline__ float lane_up2(float v)
{
   if (SHIFT == 0) return __shfl_down(v, 2);
   return lane_up1<1>(lane_up1<1>(v));
}

// One filter for this lane's frame.  Taps at kf + BASEOFF bytes, 256 floats in order [i = 3,2,1,0][l][j].
// On entry (pa,pb) hold the first 32 taps (already waited for); on exit they hold the first 32 taps found at
// next_base + NEXTOFF (the following filter), so the pipeline never drains.
template <int BASEOFF, int NEXTOFF, int SHIFT, int ABL = 0, int PK = 0>
__device__ __forceinline__ float stft_filter(const float (&x)[64], const float *kf, const float *next_base,
                                             f16v &pa, f16v &pb, float vk = 1.0f)
{
   f16v qa, qb;
   const f2v *x2 = reinterpret_cast<const f2v *>(&x[0]);   // x[8 j + l], x[8 j + l + 1]  ->  x2[4 j + l / 2]
   (void)x2;
   if constexpr (PK == 2) {
      // everything on (l, l+1) pairs: products v_pk_mul_f32, tree adds and the cross-lane combination v_pk_add_f32
      f2v ga[4], gb[4], t23[4], v[4];
      VADC_SLOAD32(qa, qb, kf, BASEOFF + 1 * 128);  VADC_STAGE_PK2(ga, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_3, l 0..3
      VADC_SLOAD32(pa, pb, kf, BASEOFF + 2 * 128);  VADC_STAGE_PK2(ga, 4, qa, qb)  VADC_SWAIT(pa, pb);   // G_3, l 4..7
      VADC_SLOAD32(qa, qb, kf, BASEOFF + 3 * 128);  VADC_STAGE_PK2(gb, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_2
      VADC_SLOAD32(pa, pb, kf, BASEOFF + 4 * 128);  VADC_STAGE_PK2(gb, 4, qa, qb)  VADC_SWAIT(pa, pb);
#pragma unroll
      for (int i = 0; i < 4; ++i) t23[i] = gb[i] + (f2v){lane_up1<SHIFT>(ga[i].x), lane_up1<SHIFT>(ga[i].y)};   // g_2 + g_3   (stft.c:166)
      VADC_SLOAD32(qa, qb, kf, BASEOFF + 5 * 128);  VADC_STAGE_PK2(ga, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_1
      VADC_SLOAD32(pa, pb, kf, BASEOFF + 6 * 128);  VADC_STAGE_PK2(ga, 4, qa, qb)  VADC_SWAIT(pa, pb);
      VADC_SLOAD32(qa, qb, kf, BASEOFF + 7 * 128);  VADC_STAGE_PK2(gb, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_0
      VADC_SLOAD32(pa, pb, next_base, NEXTOFF);     VADC_STAGE_PK2(gb, 4, qa, qb)  VADC_SWAIT(pa, pb);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
         const f2v t01 = gb[i] + (f2v){lane_up1<SHIFT>(ga[i].x), lane_up1<SHIFT>(ga[i].y)};               // g_0 + g_1   (stft.c:165)
         v[i] = t01 + (f2v){lane_up2<SHIFT>(t23[i].x), lane_up2<SHIFT>(t23[i].y)};                        // stft.c:167
      }
      const float s01 = v[0].x + v[0].y, s23 = v[1].x + v[1].y, s45 = v[2].x + v[2].y, s67 = v[3].x + v[3].y;   // stft.c:176-184
      const float s0123 = s01 + s23, s4567 = s45 + s67;
      return s0123 + s4567;
   } else {
   float ga[8], gb[8], t23[8], v[8];
#define VADC_STG(G, LB, CA, CB_) if (PK) { VADC_STAGE_PK(G, LB, CA, CB_) } else { VADC_STAGE(G, LB, CA, CB_) }
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 1 * 128);  VADC_STG(ga, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_3, l 0..3
   VADC_SLOAD32(pa, pb, kf, BASEOFF + 2 * 128);  VADC_STG(ga, 4, qa, qb)  VADC_SWAIT(pa, pb);   // G_3, l 4..7
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 3 * 128);  VADC_STG(gb, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_2
   VADC_SLOAD32(pa, pb, kf, BASEOFF + 4 * 128);  VADC_STG(gb, 4, qa, qb)  VADC_SWAIT(pa, pb);
#pragma unroll
   for (int l = 0; l < 8; ++l) t23[l] = gb[l] + lane_up1<SHIFT>(ga[l]);     // g_2 + g_3   (stft.c:166)
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 5 * 128);  VADC_STG(ga, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_1
   VADC_SLOAD32(pa, pb, kf, BASEOFF + 6 * 128);  VADC_STG(ga, 4, qa, qb)  VADC_SWAIT(pa, pb);
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 7 * 128);  VADC_STG(gb, 0, pa, pb)  VADC_SWAIT(qa, qb);   // G_0
   VADC_SLOAD32(pa, pb, next_base, NEXTOFF);     VADC_STG(gb, 4, qa, qb)  VADC_SWAIT(pa, pb);
#pragma unroll
   for (int l = 0; l < 8; ++l) {
      const float t01 = gb[l] + lane_up1<SHIFT>(ga[l]);                    // g_0 + g_1   (stft.c:165)
      v[l] = t01 + lane_up2<SHIFT>(t23[l]);                                // stft.c:167
   }
   const float s01 = v[0] + v[1], s23 = v[2] + v[3], s45 = v[4] + v[5], s67 = v[6] + v[7];   // stft.c:176-184
   const float s0123 = s01 + s23, s4567 = s45 + s67;
#undef VADC_STG
   return s0123 + s4567;
   }
}

// ---- PIPE = 1: mul/add interleaved software pipeline -------------------------------------------------
// Measured on MI355X (tools/valu_rate.hip): v_mul_f32/v_add_f32 with VGPR operands issue at 2 cycles per
// wave64 instruction, but any VALU op with an SGPR operand takes 4 -- unless SGPR-operand and VGPR-only
// instructions ALTERNATE, in which case the pair costs ~4.8 (2.4 each).  The tree needs 256 SGPR-operand muls
// and 255 VGPR-only adds per output, so the stream is arranged as (mul of tree t+1, add of tree t) pairs:
// each "slot" multiplies the 8 products of a new tree while it reduces the 8 products of the previous one.
// sched_barrier(0) after every pair pins the order (the expression, hence every rounding, is unchanged).
#define VADC_SB __builtin_amdgcn_sched_barrier(0)
#define VADC_SLOT(DST, LN, KV, O)                                                                     \
   {                                                                                                  \
      const float q0 = x[0 * 8 + (LN)] * KV[(O) + 0]; const float a01 = p[0] + p[1]; VADC_SB;         \
      const float q1 = x[1 * 8 + (LN)] * KV[(O) + 1]; const float a23 = p[2] + p[3]; VADC_SB;         \
      const float q2 = x[2 * 8 + (LN)] * KV[(O) + 2]; const float a45 = p[4] + p[5]; VADC_SB;         \
      const float q3 = x[3 * 8 + (LN)] * KV[(O) + 3]; const float a67 = p[6] + p[7]; VADC_SB;         \
      const float q4 = x[4 * 8 + (LN)] * KV[(O) + 4]; const float b03 = a01 + a23; VADC_SB;           \
      const float q5 = x[5 * 8 + (LN)] * KV[(O) + 5]; const float b47 = a45 + a67; VADC_SB;           \
      const float q6 = x[6 * 8 + (LN)] * KV[(O) + 6]; DST = b03 + b47; VADC_SB;                       \
      const float q7 = x[7 * 8 + (LN)] * KV[(O) + 7]; VADC_SB;                                        \
      p[0] = q0; p[1] = q1; p[2] = q2; p[3] = q3; p[4] = q4; p[5] = q5; p[6] = q6; p[7] = q7;         \
   }
// four trees: three from the current buffer (C*), then -- after the wait -- the first tree of the next one (N*)
#define VADC_PSTAGE(G, LB, CA, CB_, NA, NB, GN, LNEXT)                                                \
   VADC_SLOT(G[(LB) + 0], (LB) + 1, CA, 8) VADC_SLOT(G[(LB) + 1], (LB) + 2, CB_, 0)                   \
   VADC_SLOT(G[(LB) + 2], (LB) + 3, CB_, 8) VADC_SWAIT(NA, NB); VADC_SLOT(G[(LB) + 3], LNEXT, NA, 0)

// Same contract as stft_filter plus the carry p[8]: on entry the products of this filter's first tree, on
// exit the products of the next filter's first tree.
template <int BASEOFF, int NEXTOFF, int SHIFT>
__device__ __forceinline__ float stft_filter_pipe(const float (&x)[64], const float *kf, const float *next_base,
                                                  f16v &pa, f16v &pb, float (&p)[8])
{
   f16v qa, qb;
   float ga[8], gb[8], t23[8], v[8];
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 1 * 128);  VADC_PSTAGE(ga, 0, pa, pb, qa, qb, ga, 4)   // G_3
   VADC_SLOAD32(pa, pb, kf, BASEOFF + 2 * 128);  VADC_PSTAGE(ga, 4, qa, qb, pa, pb, gb, 0)
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 3 * 128);  VADC_PSTAGE(gb, 0, pa, pb, qa, qb, gb, 4)   // G_2
   VADC_SLOAD32(pa, pb, kf, BASEOFF + 4 * 128);  VADC_PSTAGE(gb, 4, qa, qb, pa, pb, ga, 0)
#pragma unroll
   for (int l = 0; l < 8; ++l) t23[l] = gb[l] + lane_up1<SHIFT>(ga[l]);     // g_2 + g_3   (stft.c:166)
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 5 * 128);  VADC_PSTAGE(ga, 0, pa, pb, qa, qb, ga, 4)   // G_1
   VADC_SLOAD32(pa, pb, kf, BASEOFF + 6 * 128);  VADC_PSTAGE(ga, 4, qa, qb, pa, pb, gb, 0)
   VADC_SLOAD32(qa, qb, kf, BASEOFF + 7 * 128);  VADC_PSTAGE(gb, 0, pa, pb, qa, qb, gb, 4)   // G_0
   VADC_SLOAD32(pa, pb, next_base, NEXTOFF);     VADC_PSTAGE(gb, 4, qa, qb, pa, pb, ga, 0)   // + next filter's tree 0
#pragma unroll
   for (int l = 0; l < 8; ++l) {
      const float t01 = gb[l] + lane_up1<SHIFT>(ga[l]);                    // g_0 + g_1   (stft.c:165)
      v[l] = t01 + lane_up2<SHIFT>(t23[l]);                                // stft.c:167
   }
   const float s01 = v[0] + v[1], s23 = v[2] + v[3], s45 = v[4] + v[5], s67 = v[6] + v[7];   // stft.c:176-184
   const float s0123 = s01 + s23, s4567 = s45 + s67;
   return s0123 + s4567;
}

// MODE 0: Y = log1p(2^20 * magnitude), FM = mean over the 129 bins (the engine's normal path)
// MODE 1: Y = magnitude (stage tap for the bit-exact STFT parity test), FM untouched
// Tuning knobs (tools/fe_bench.hip sweeps them on the device):
//   NT       threads per workgroup              MINW   min waves per SIMD (register cap)
//   SHIFT    0 ds_bpermute / 1 DPP wave shift   LOCK   1: one barrier per filter pair keeps the workgroup's waves
//            on the same basis rows so that they share scalar-cache lines
//   ABL      timing ablations for tools/fe_bench.hip (0 in the product)
// GEO: 0 = Silero v3.1 (reflect pad 128 -> 28 blocks, 25 frames), 1 = Silero v4 (reflect pad 96 -> 27 blocks, 24 frames;
//      silero_vad.py:26 to_pad = (n_fft - stride) / 2).  MODE 2 (v4): Y = log1p(2^20 m) AND MAG = m (the v4 encoder takes both).
template <typename T, int MODE, int NT = 256, int MINW = 3, int SHIFT = 0, int LOCK = 0, int ABL = 0, int PIPE = 0, int PK = 0, int GEO = 0>
__global__ __launch_bounds__(NT, MINW) void k_frontend(const T *__restrict__ pcm,          // [n_chunks][1536]
                                                  const float *__restrict__ basis,    // [258][256] permuted
                                                  float *__restrict__ Y,              // [n_chunks][129][frames]
                                                  float *__restrict__ FM,             // [kBinSplit][fm_stride] partial bin sums
                                                  int n_chunks, ItemMap map, size_t fm_stride, float *__restrict__ MAG = nullptr)
{
   constexpr int kPad = GEO == 0 ? vadc::kPad : 96;
   constexpr int kBlocks = GEO == 0 ? vadc::kBlocks : 27;
   constexpr int kFrames = GEO == 0 ? vadc::kFrames : 24;
   constexpr int kFirstFull = (kPad + 63) / 64;                        // first block without mirrored samples
   constexpr int kLastFull = (kChunk + kPad - 64) / 64;                // last block without mirrored samples
   const int lane = threadIdx.x & 63;
   const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const long total_slots = (long)n_chunks * kBlocks;
   const long slot0 = (long)wave * kLanesOut;
   if (!LOCK && slot0 >= total_slots) return;           // wave-uniform (with LOCK every wave must reach the barriers)
   const long slot = slot0 + lane;
   const bool live = slot < total_slots;
   const int item = live ? (int)(slot / kBlocks) : n_chunks - 1;
   const int m = live ? (int)(slot - (long)item * kBlocks) : kBlocks - 1;
   const int chunk = map(item);                         // position in the stream-major buffers

   // this lane's block of the reflect-padded chunk (tensor.h:931-954): padded index 64m+k, source
   // index s = 64m + k - 128, mirrored at both ends without repeating the edge sample.
   __attribute__((aligned(8))) float x[64];
   const T *src = pcm + (size_t)chunk * kChunk;
   if (m >= kFirstFull && m <= kLastFull) {
      const T *p = src + (64 * m - kPad);
#pragma unroll
      for (int k = 0; k < 64; ++k) x[k] = sample_to_f32(p[k]);
   } else {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
         int s = 64 * m + k - kPad;
         s = s < 0 ? -s : s;
         s = s >= kChunk ? 2 * (kChunk - 1) - s : s;
         x[k] = sample_to_f32(src[s]);
      }
   }

   const bool writer = live && lane < kLanesOut && m < kFrames;
   float *yout = Y + (size_t)chunk * (kBins * kFrames) + m;
   float bin_sum = 0.0f;
   constexpr int kImOff = kBins * kFilterLen * 4;       // byte offset from filter f to filter f+129
   // blockIdx.y selects a third of the 129 bins: 3x more, 3x shorter waves (small launches fill the chip, short
   // tails); each split writes its own partial bin sums, added in fixed order by the consumer (deterministic).
   const int f_start = blockIdx.y * kBinsPerSplit;
   const int f_end = min(f_start + kBinsPerSplit, kBins);
   f16v pa, pb;
   {
      const float *k0 = basis + (size_t)f_start * kFilterLen;
      VADC_SLOAD32(pa, pb, k0, 0);
      VADC_SWAIT(pa, pb);
   }
   float p[8];                                          // PIPE: products of the first tree of the first filter
#pragma unroll
   for (int j = 0; j < 8; ++j) p[j] = PIPE ? x[j * 8] * pa[j] : 0.0f;
   for (int f = f_start; f < f_end; ++f) {
      const int fn = f + 1;
      const float *kf = basis + (size_t)f * kFilterLen; // wave-uniform
      const float *kn = basis + (size_t)fn * kFilterLen;
      if (LOCK) __syncthreads();
      const float re = PIPE ? stft_filter_pipe<0, kImOff, SHIFT>(x, kf, kf, pa, pb, p)
                            : stft_filter<0, kImOff, SHIFT, ABL, PK>(x, kf, kf, pa, pb, x[63]);
      // the prefetch issued by the last stage of `im` reads the next filter's first taps (row 129 = im of
      // bin 0 when f == 128 and there is no stagger: in bounds, unused)
      const float im = PIPE ? stft_filter_pipe<kImOff, 0, SHIFT>(x, kf, kn, pa, pb, p)
                            : stft_filter<kImOff, 0, SHIFT, ABL, PK>(x, kf, kn, pa, pb, x[63]);
      const float re2 = re * re, im2 = im * im;
      const float mag = sqrtf(re2 + im2);                                  // stft.c:209
      float val;
      if (MODE == 0 || MODE == 2) {
         val = log1p_hw(mag * 1048576.0f);                                   // misc.c:42-45
         bin_sum += val;                                                   // misc.c:55-59 (channel order)
      } else {
         val = mag;
      }
      if (writer) yout[f * kFrames] = val;
      if (MODE == 2 && writer) MAG[(size_t)chunk * (kBins * kFrames) + m + f * kFrames] = mag;
   }
   if ((MODE == 0 || MODE == 2) && writer) FM[blockIdx.y * fm_stride + (size_t)chunk * kFrames + m] = bin_sum;   // /129 by the reader (misc.c:60)
}



// =====================================================================================================
// k_frontend_fl -- the same bit-exact tree with one LANE per FRAME ("frame-lane"), no halo, no wave shifts
// =====================================================================================================
// k_frontend gives a lane one 64-sample block and moves group sums between lanes: 3 halo lanes per wave and the 3 blocks of
// a chunk that only feed their neighbours leave 85 % of the lanes producing outputs, and every filter pair costs 48
// ds_bpermute.  Here a lane owns one (chunk, frame) POSITION and evaluates the whole 256-tap tree of stft.c:115-184 itself:
//     for l-pair LP (tree lanes 2LP, 2LP+1), group i:  g_i = tree8_j( x[64 (n+i) + 8 j + l] * k[64 i + 8 j + l] )   (packed pair)
//     v = (g_0 + g_1) + (g_2 + g_3) ;  y = ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7))  accumulated as the l-pairs complete
// -- the same IEEE operations in the same order, so Y and FM are bit-identical to k_frontend's.  The 16 samples a step
// (LP, i) needs are four ds_read_b128 from the workgroup's tile of reflect-padded chunks (block pitch 68 floats: the
// 16 lanes of a read phase hit disjoint banks; inside a block sample 8 j + l sits at 16 (l/2) + 2 j + l%2, so an l-pair's
// taps for j, j+1 are one aligned 16-byte read = two v_pk operands), prefetched one step ahead into the other register
// set, and are shared by the 2 NB filters of a bin batch, whose partial trees stay in registers (6 VGPRs per filter).
// The basis taps stay wave-uniform SCALAR loads from the same permuted basis k_frontend uses (one s_load_dwordx16 = the
// 16 taps of one (filter, i, LP)), software-pipelined one stage (= re + im of one bin) ahead.
// A workgroup = NPS x 64 consecutive positions (their chunks staged once) x 4 waves = the 4 bin splits of common.h, so FM
// keeps the partial-sum order the first encoder layer expects.  Every lane produces an output: lane efficiency ~100 %.
// What bounds it (tools/fe_bench.hip ablations, 16,384 chunks, kernel alone on the chip): 1.03 ms as shipped (NPS = 1); 0.89 ms
// with every tap load hitting the scalar cache (the 264 KB basis streams through a 16 KB scalar cache); the packed-instruction
// issue floor of the 518 VALU instructions per (bin, wave) is ~0.79 ms (4.3 cycles per v_pk_*_f32 at the ~2.25 GHz the chip
// sustains under this load, tools/pk_rate.hip).  NPS = 2 (512 threads: the two waves of a bin split share their tap lines)
// runs 0.90 ms alone -- but inside the engine, next to the LSTM chain on its own CUs, it is 5 % SLOWER than NPS = 1 (1.14 vs
// 1.08 ms, rocprofv3): the chip is power-limited under this kernel (clocks 2.0-2.4 GHz depending on the instruction mix,
// 32 CUs fewer cost 7 % not 14 %), so the in-engine A/B decides, and NPS = 1 ships (option "fe_nps").
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kFlBlockPitch = 68;
constexpr int fl_chunks(int nps) { return (kFrames - 1 + 64 * nps - 1) / kFrames + 1; }   // 64 nps consecutive positions span at most this many chunks

#define VADC_FL_SLOAD2(A, B, BASE, OFFA, OFFB)                                          \
   asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"              \
                : "=&s"(A), "=&s"(B)                                                     \
                : "s"(BASE), "n"(OFFA), "n"(OFFB));                                       \
   __builtin_amdgcn_sched_barrier(0)
#define VADC_FL_LDS16(X, ADDR, OFF)                                                     \
   asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\t"    \
                "ds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"        \
                : "=&v"(X[0]), "=&v"(X[1]), "=&v"(X[2]), "=&v"(X[3])                     \
                : "v"(ADDR), "n"(OFF), "n"((OFF) + 16), "n"((OFF) + 32), "n"((OFF) + 48)); \
   __builtin_amdgcn_sched_barrier(0)

// g = ((q0+q1)+(q2+q3)) + ((q4+q5)+(q6+q7)),  q_j = (x[8j+l], x[8j+l+1]) * (k[j][l], k[j][l+1])   (stft.c:141-160)
template <int ABL = 0>
__device__ __forceinline__ f2v fl_tree8(const f4v (&xq)[4], const f16v &kv_)
{
   f4v kvv[4] = {xq[1], xq[2], xq[3], xq[0]};                    // ABL & 4 (tools/fe_bench.hip only): VGPR operands instead of the SGPR taps
   const float *kx = reinterpret_cast<const float *>(&kvv[0]);
   float kv[16];
#pragma unroll
   for (int e = 0; e < 16; ++e) kv[e] = (ABL & 4) ? kx[e] : kv_[e];
   const f2v q0 = __builtin_shufflevector(xq[0], xq[0], 0, 1) * (f2v){kv[0], kv[1]};
   const f2v q1 = __builtin_shufflevector(xq[0], xq[0], 2, 3) * (f2v){kv[2], kv[3]};
   const f2v q2 = __builtin_shufflevector(xq[1], xq[1], 0, 1) * (f2v){kv[4], kv[5]};
   const f2v q3 = __builtin_shufflevector(xq[1], xq[1], 2, 3) * (f2v){kv[6], kv[7]};
   const f2v q4 = __builtin_shufflevector(xq[2], xq[2], 0, 1) * (f2v){kv[8], kv[9]};
   const f2v q5 = __builtin_shufflevector(xq[2], xq[2], 2, 3) * (f2v){kv[10], kv[11]};
   const f2v q6 = __builtin_shufflevector(xq[3], xq[3], 0, 1) * (f2v){kv[12], kv[13]};
   const f2v q7 = __builtin_shufflevector(xq[3], xq[3], 2, 3) * (f2v){kv[14], kv[15]};
   const f2v a01 = q0 + q1, a23 = q2 + q3, a45 = q4 + q5, a67 = q6 + q7;
   const f2v a0123 = a01 + a23, a4567 = a45 + a67;
   return a0123 + a4567;
}

// per-filter partial trees of a bin batch (filter 2b = re of bin b, 2b+1 = im)
template <int NB>
struct FlState {
   f2v ta[2 * NB], tb[2 * NB];      // g_0 (+ g_1), g_2 (+ g_3) of the current l-pair
   float sa[2 * NB], sb[2 * NB];    // (v0+v1) (+ (v2+v3)),  (v4+v5)
   float y[2 * NB];
};

template <int I, int LP, int NB>
__device__ __forceinline__ void fl_accumulate(FlState<NB> &st, int fi, f2v g)
{
   if (I == 0) st.ta[fi] = g;
   else if (I == 1) st.ta[fi] = st.ta[fi] + g;                   // g_0 + g_1   (stft.c:165)
   else if (I == 2) st.tb[fi] = g;
   else {
      const f2v t23 = st.tb[fi] + g;                             // g_2 + g_3   (stft.c:166)
      const f2v v = st.ta[fi] + t23;                             // stft.c:167
      const float s = v.x + v.y;                                 // stft.c:176-184, as the l-pairs complete
      if (LP == 0) st.sa[fi] = s;
      else if (LP == 1) st.sa[fi] = st.sa[fi] + s;
      else if (LP == 2) st.sb[fi] = s;
      else { const float s4567 = st.sb[fi] + s; st.y[fi] = st.sa[fi] + s4567; }
   }
}

// byte offset of the 16 taps of (bin b of the batch, group i, l-pair lp) from the batch's first filter; basis layout
// [f][ii = 3 - i][lp][j][l % 2] (engine_weights.hip), re and im rows of a bin are kBins rows apart
constexpr int fl_tap_off(int b, int i, int lp) { return b * 1024 + (3 - i) * 256 + lp * 64; }

// Stage K of a batch = (step S = K / NB = 4 LP + I, bin B = K % NB).  On entry ca/cb hold the taps of this stage (waited
// for) and xc the samples of this step; the stage requests the next stage's taps into na/nb and, at B == 0, the next
// step's samples into xn.  After the last stage ca/cb and xc are again "current" for the next batch (16 NB stages and 16
// steps are even counts, so the buffers end where they started).
// ABL (timing ablations for tools/fe_bench.hip, 0 in the product; results are garbage otherwise): 1 = no tap loads / waits,
// 2 = no sample loads, 4 = products by VGPR pairs instead of SGPR pairs, 8 = every batch reads the taps of the first one
template <int NB, int K, int ABL = 0>
struct FlStages {
   static __device__ __forceinline__ void run(FlState<NB> &st, const float *kf, unsigned xaddr, f16v &ca, f16v &cb, f16v &na, f16v &nb,
                                              f4v (&xc)[4], f4v (&xn)[4])
   {
      constexpr int S = K / NB, B = K % NB, LP = S / 4, I = S % 4;
      constexpr int Kn = K + 1;
      constexpr int Sn = (Kn / NB) % 16, Bn = Kn % NB, LPn = Sn / 4, In = Sn % 4;
      constexpr int noff = (Kn == 16 * NB ? NB * 1024 : 0) + fl_tap_off(Bn, In, LPn);
      constexpr int kImOffB = kBins * kFilterLen * 4;
      if constexpr (B == 0 && !(ABL & 2)) {
         constexpr int S1 = (S + 1) % 16;
         VADC_FL_LDS16(xn, xaddr, ((S1 % 4) * kFlBlockPitch + (S1 / 4) * 16) * 4);
      }
      if constexpr (!(ABL & 1)) { VADC_FL_SLOAD2(na, nb, kf, noff, noff + kImOffB); }
      fl_accumulate<I, LP, NB>(st, 2 * B, fl_tree8<ABL>(xc, ca));
      fl_accumulate<I, LP, NB>(st, 2 * B + 1, fl_tree8<ABL>(xc, cb));
      if constexpr (B == NB - 1) {
         // two statements: an asm with SGPR and VGPR outputs counts as a divergent source and its taps would be copied to VGPRs
         if constexpr (!(ABL & 1)) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb));
         else if constexpr (!(ABL & 2)) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb));
         else asm volatile("" : "+s"(na), "+s"(nb));
         asm volatile("" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3]));
         __builtin_amdgcn_sched_barrier(0);
         FlStages<NB, K + 1, ABL>::run(st, kf, xaddr, na, nb, ca, cb, xn, xc);
      } else {
         if constexpr (!(ABL & 1)) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb));
         else asm volatile("" : "+s"(na), "+s"(nb));
         __builtin_amdgcn_sched_barrier(0);
         FlStages<NB, K + 1, ABL>::run(st, kf, xaddr, na, nb, ca, cb, xc, xn);
      }
   }
};
template <int NB, int ABL>
struct FlStages<NB, 16 * NB, ABL> {
   static __device__ __forceinline__ void run(FlState<NB> &, const float *, unsigned, f16v &, f16v &, f16v &, f16v &, f4v (&)[4], f4v (&)[4]) {}
};

// Workgroups of a launch are dealt to the 8 XCDs in turn ((start + blockIdx) % 8, start unknown: tools/xcd_map_probe.hip), so neighbours in blockIdx sit behind two different
// L2s.  The exact-tree front ends give neighbouring workgroups neighbouring blocks of 64 positions, and two such blocks share a chunk: they write the two parts of the same
// 128-byte lines of Y.  XCD-major order hands the workgroups of one residue (= one XCD) CONSECUTIVE blocks, so a line's parts meet in one L2 before it is written back:
// tools/fe_bench FE_POWER, 24,576 chunks on 224 CUs: 0.4312 -> 0.4197 ms (without any Y store: 0.4055).  A permutation of which workgroup computes which block: no bit changes.
__device__ __forceinline__ unsigned xcd_major_block(unsigned bid, unsigned n_blocks)
{
   const unsigned q = n_blocks >> 3, r = n_blocks & 7, x = bid & 7, i = bid >> 3;
   return x * q + (x < r ? x : r) + i;
}

// MODE as k_frontend (0: Y = log1p(2^20 m) + FM partial bin sums; 1: Y = magnitude).  NB = bins per batch; kBinsPerSplit = 33 and
// the last split's 30 bins are both multiples of 3.
// NPS = position sets (of 64) per workgroup: the NPS waves that work on the same bin split start together and read the same
// taps at about the same time, so all but the first of them hit the scalar cache.
// PERSIST (option "fe_persist"): the grid is as large as the chip (slots), every workgroup draws its units of 64 NPS positions from
// `work_counter` (zeroed by the host before the launch) until they run out.  Alone on the chip this is 8-12 % faster (16,384 chunks:
// 1.05 -> 0.96 ms, on a stream masked to 248 CUs 1.16 -> 1.03; 65,536 chunks 4.12 -> 3.60: tools/fe_bench.hip) -- but inside the
// engine the plain grid already runs at that rate (3.53 ms per 65,536 chunks, 1.07 per 16,384 next to the LSTM chain) and the
// persistent one is no faster (3.57 / 1.08), so the plain grid stays the default.
template <typename T, int MODE, int NB = 3, int MINW = 4, int ABL = 0, int NPS = 1, bool PERSIST = false>
__global__ __launch_bounds__(256 * NPS, MINW) void k_frontend_fl(const T *__restrict__ pcm,          // [n_chunks][1536]
                                                           const float *__restrict__ basis,    // [258][256] permuted (k_frontend's)
                                                           float *__restrict__ Y,              // [n_chunks][129][25]
                                                           float *__restrict__ FM,             // [kBinSplit][fm_stride] partial bin sums
                                                           int n_chunks, ItemMap map, size_t fm_stride, int *__restrict__ work_counter = nullptr)
{
   static_assert(kBinsPerSplit % NB == 0 && (kBins - 3 * kBinsPerSplit) % NB == 0, "bin batches must tile every split");
   static_assert((16 * NB) % 2 == 0, "tap buffers must end where they started");
   constexpr int kChunkPitch = kBlocks * kFlBlockPitch;          // 1904 floats per chunk
   constexpr int kFlChunks = fl_chunks(NPS);
   __shared__ __attribute__((aligned(16))) float xs[kFlChunks * kChunkPitch];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;          // bin split
   const int pset = __builtin_amdgcn_readfirstlane(tid >> 8);              // position set
   const long total_pos = (long)n_chunks * kFrames;
   __shared__ int unit_s;
   const int n_units = (int)((total_pos + 64 * NPS - 1) / (64 * NPS));
#pragma unroll 1
   for (int round = 0;; ++round) {
   int unit = PERSIST ? 0 : (int)xcd_major_block(blockIdx.x, gridDim.x);      // neighbouring units share a chunk's lines of Y: one XCD's workgroups take consecutive units
   if (PERSIST) {
      if (round) __syncthreads();                          // the previous unit's readers of xs / unit_s are done
      if (tid == 0) unit_s = atomicAdd(work_counter, 1);
      __syncthreads();
      unit = unit_s;
      if (unit >= n_units) break;
   } else if (round) break;
   const long p0 = (long)unit * (64 * NPS);
   const int item0 = (int)(p0 / kFrames);

   // stage the (up to) 4 chunks these 64 positions touch: reflect pad (tensor.h:931-954), l-pair-major inside a block
   for (int c = 0; c < kFlChunks; ++c) {
      const int it = min(item0 + c, n_chunks - 1);
      const T *src = pcm + (size_t)map(it) * kChunk;
      for (int idx = tid; idx < kPadded; idx += 256 * NPS) {
         int s = idx - kPad;
         s = s < 0 ? -s : s;
         s = s >= kChunk ? 2 * (kChunk - 1) - s : s;
         const int k = idx & 63, j = k >> 3, l = k & 7;
         xs[c * kChunkPitch + (idx >> 6) * kFlBlockPitch + (l >> 1) * 16 + j * 2 + (l & 1)] = sample_to_f32(src[s]);
      }
   }
   __syncthreads();

   const long pe = p0 + pset * 64 + lane;
   const bool writer = pe < total_pos;
   const long pa_ = writer ? pe : total_pos - 1;
   const int item = (int)(pa_ / kFrames), n = (int)(pa_ - (long)item * kFrames);
   const int chunk = map(item);
   typedef __attribute__((address_space(3))) float lds_f;
   const unsigned xaddr = (unsigned)(uintptr_t)(lds_f *)(xs + (item - item0) * kChunkPitch + kFlBlockPitch * n);

   const int f_start = wave * kBinsPerSplit;
   const int f_end = min(f_start + kBinsPerSplit, kBins);
   float *yout = Y + (size_t)chunk * (kBins * kFrames) + n;
   float bin_sum = 0.0f;
   constexpr int kImOffB = kBins * kFilterLen * 4;

   f16v ca, cb, na, nb;
   f4v xc[4], xn[4];
   {
      const float *k0 = basis + (size_t)f_start * kFilterLen;
      VADC_FL_LDS16(xc, xaddr, 0);
      VADC_FL_SLOAD2(ca, cb, k0, fl_tap_off(0, 0, 0), fl_tap_off(0, 0, 0) + kImOffB);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca), "+s"(cb));
      asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]));
      __builtin_amdgcn_sched_barrier(0);
   }
#pragma unroll 1
   for (int f = f_start; f < f_end; f += NB) {
      const float *kf = basis + (size_t)((ABL & 8) ? 0 : (ABL & 16) ? f - f_start : f) * kFilterLen;          // wave-uniform  (ABL & 8: every batch reads the same taps = scalar-cache hits)
      FlState<NB> st;
      if (ABL & 1) { na = ca; nb = cb; }
      if (ABL & 2) { xn[0] = xc[0]; xn[1] = xc[1]; xn[2] = xc[2]; xn[3] = xc[3]; }
      FlStages<NB, 0, ABL>::run(st, kf, xaddr, ca, cb, na, nb, xc, xn);
      // pin the trees here: with the stores below under `if (writer)`, machine sinking would otherwise move the arithmetic
      // of a whole bin into the epilogue's blocks, far below the taps it consumes (spilling every tap on the way)
#pragma unroll
      for (int b = 0; b < 2 * NB; ++b) asm volatile("" : "+v"(st.y[b]));
#pragma unroll
      for (int b = 0; b < NB; ++b) {
         const float re = st.y[2 * b], im = st.y[2 * b + 1];
         const float re2 = re * re, im2 = im * im;
         const float mag = sqrtf(re2 + im2);                                  // stft.c:209
         float val;
         if (MODE == 0) {
            val = log1p_hw(mag * 1048576.0f);                                 // misc.c:42-45
            bin_sum += val;                                                   // misc.c:55-59 (channel order)
         } else {
            val = mag;
         }
         if (writer) yout[(f + b) * kFrames] = val;
      }
   }
   if (MODE == 0 && writer) FM[wave * fm_stride + (size_t)chunk * kFrames + n] = bin_sum;   // /129 by the reader (misc.c:60)
   }
}

// =====================================================================================================
// k_frontend_sym -- the bit-exact tree evaluated for 33 of the 129 bins; the other 96 follow from the basis' symmetries
// =====================================================================================================
// The reference's basis (row k < 129: w[n] cos(2 pi k n / 256), row 129 + k: -w[n] sin(2 pi k n / 256), w = periodic Hann) satisfies,
// BIT FOR BIT (the engine verifies it on the loaded tensor at create time, engine_weights.hip basis_has_dft_symmetries; otherwise
// k_frontend_fl runs):
//     re[128-b][n] = (-1)^n re[b][n]                      im[128-b][n] = -(-1)^n im[b][n]
//     re[64-b][n]  = {re, -im, -re, im}[b][n]  by n % 4   im[64-b][n]  = {-im, -re, im, re}[b][n]  by n % 4
// A tap is t = 64 i + 8 j + l, so n % 4 = l % 4: inside one tree lane l every tap of a derived row is +-(the same tap of row re[b] or
// im[b]).  IEEE multiplication and addition commute with negation (round-to-nearest is symmetric), so the whole per-lane part of the
// reference's tree -- products, the j tree, the group sums (stft.c:141-167) -- of a derived row equals +-v[l] of a base row EXACTLY:
//     v'[l] = s(l) v_re[l]  or  s(l) v_im[l]
// and only the last 7 additions over the tree lanes (stft.c:176-184) have to be redone per derived row, on sign-flipped operands:
// with (x, y) = (v[2 LP], v[2 LP + 1]) of an l-pair LP, the pair sums of the 8 rows that base bin b yields are
//     re b: rx + ry        im b: ix + iy        re 128-b: rx - ry     im 128-b: ix - iy     (same sign for every l-pair)
//     re 64-b: rx - iy     im 64-b: ix + ry     re 64+b: rx + iy      im 64+b: ix - ry      (sign alternates with the l-pair)
// (global signs dropped: only re^2 + im^2 is used; a zero may come out with the other sign, which squares away as well), and
//     y = (e0 + e1) + (e2 + e3)  for the first four,   y = (e0 - e1) + (e2 - e3)  for the alternating ones.
// Base bins 0..32 give all 129 bins: {b, 128-b, 64-b, 64+b}, with 64 -+ 0 and 64 -+ 32 coinciding with rows already there.  Work per
// position: 33 x 2 trees of 511 operations + 56 additions per base bin instead of 258 trees -- 3.9x less of what bounds k_frontend_fl --
// and magnitudes that are still the reference's bits (tests/test_gpu_parity.py::test_stft_magnitude_bit_exact_*; the derivation itself is
// checked step by step in float32 numpy, tests/test_stft_symmetry.py).
// Mapping as k_frontend_fl (one lane per (chunk, frame) position, samples from the workgroup's LDS tile, taps by scalar loads, NB base bins
// per batch share the sample reads); the 4 waves of a workgroup split the base bins 9 / 9 / 9 / 6, and with them the partial bin sums.
// MODE 0 takes v_sqrt_f32 (1 ulp) for the magnitude under the logarithm: Y moves by <= 6e-8, below log1p_hw's own error; the
// magnitude tap (MODE 1) keeps the correctly rounded sqrtf of stft.c:209.
constexpr int kSymBase = 33;                                     // base bins 0..32
constexpr int kSymChunkPitch = 1956;                             // floats per staged chunk: == 25 * 68 (mod 64), so a lane's tile address is linear in
                                                                 // its position across chunk boundaries and every ds_read_b128 phase stays conflict-free
__host__ __device__ constexpr int sym_first_bin(int wave, int nb) { return nb == 3 ? 9 * wave : (wave == 0 ? 0 : 8 * wave + 1); }
__host__ __device__ constexpr int sym_end_bin(int wave, int nb) { return wave == 3 ? kSymBase : sym_first_bin(wave + 1, nb); }

template <int NB>
struct SymState {
   f2v ta[2 * NB], tb[2 * NB];      // g_0 (+ g_1), g_2 of the current l-pair; filter 2 B = re, 2 B + 1 = im of base bin B
   float sa[8 * NB], sb[8 * NB];    // partial lane trees of the 8 rows of a base bin: e0 (+- e1), e2; sa ends up as the row's y
};

template <int LP, int NB>
__device__ __forceinline__ void sym_rows(SymState<NB> &st, int B, f2v vr, f2v vi)
{
   float e[8];
   e[0] = vr.x + vr.y; e[1] = vi.x + vi.y; e[2] = vr.x - vr.y; e[3] = vi.x - vi.y;
   e[4] = vr.x - vi.y; e[5] = vi.x + vr.y; e[6] = vr.x + vi.y; e[7] = vi.x - vr.y;
#pragma unroll
   for (int k = 0; k < 8; ++k) {
      float &sa = st.sa[8 * B + k], &sb = st.sb[8 * B + k];
      if (LP == 0) sa = e[k];
      else if (LP == 1) sa = (k < 4) ? sa + e[k] : sa - e[k];
      else if (LP == 2) sb = e[k];
      else { const float t = (k < 4) ? sb + e[k] : sb - e[k]; sa = sa + t; }
   }
}

// Stage K of a batch = (step S = K / NB = 4 LP + I, base bin B = K % NB), pipelined exactly like FlStages
// ZI: the im row of the batch's (only) base bin is identically zero (bin 0 of a DFT basis: -w[n] sin(0)): its products are +-0, its tree sums +-0, and a
// zero of either sign adds and squares to the same bits downstream -- the tree is skipped, not approximated
template <int NB, int K, bool ZI = false>
struct SymStages {
   static __device__ __forceinline__ void run(SymState<NB> &st, const float *kf, unsigned xaddr, f16v &ca, f16v &cb, f16v &na, f16v &nb,
                                              f4v (&xc)[4], f4v (&xn)[4])
   {
      constexpr int S = K / NB, B = K % NB, LP = S / 4, I = S % 4;
      constexpr int Kn = K + 1;
      constexpr int Sn = (Kn / NB) % 16, Bn = Kn % NB, LPn = Sn / 4, In = Sn % 4;
      constexpr int noff = (Kn == 16 * NB ? NB * 1024 : 0) + fl_tap_off(Bn, In, LPn);
      constexpr int kImOffB = kBins * kFilterLen * 4;
      if constexpr (B == 0) {
         constexpr int S1 = (S + 1) % 16;
         VADC_FL_LDS16(xn, xaddr, ((S1 % 4) * kFlBlockPitch + (S1 / 4) * 16) * 4);
      }
      VADC_FL_SLOAD2(na, nb, kf, noff, noff + kImOffB);
      const f2v gr = fl_tree8<0>(xc, ca), gi = ZI ? (f2v){0.0f, 0.0f} : fl_tree8<0>(xc, cb);
      if constexpr (I == 0) { st.ta[2 * B] = gr; st.ta[2 * B + 1] = gi; }
      else if constexpr (I == 1) { st.ta[2 * B] = st.ta[2 * B] + gr; st.ta[2 * B + 1] = st.ta[2 * B + 1] + gi; }       // g_0 + g_1   (stft.c:165)
      else if constexpr (I == 2) { st.tb[2 * B] = gr; st.tb[2 * B + 1] = gi; }
      else {
         const f2v t23r = st.tb[2 * B] + gr, t23i = st.tb[2 * B + 1] + gi;                                                // g_2 + g_3   (stft.c:166)
         sym_rows<LP, NB>(st, B, st.ta[2 * B] + t23r, st.ta[2 * B + 1] + t23i);                                            // stft.c:167, :176-184
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb));
      if constexpr (B == NB - 1) {
         asm volatile("" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3]));
         __builtin_amdgcn_sched_barrier(0);
         SymStages<NB, K + 1, ZI>::run(st, kf, xaddr, na, nb, ca, cb, xn, xc);
      } else {
         __builtin_amdgcn_sched_barrier(0);
         SymStages<NB, K + 1, ZI>::run(st, kf, xaddr, na, nb, ca, cb, xc, xn);
      }
   }
};
template <int NB, bool ZI>
struct SymStages<NB, 16 * NB, ZI> {
   static __device__ __forceinline__ void run(SymState<NB> &, const float *, unsigned, f16v &, f16v &, f16v &, f16v &, f4v (&)[4], f4v (&)[4]) {}
};

// one 8-sample octet of the reflect-padded chunk -> the tile (l-pair-major inside a block: sample 8 j + l at 16 (l / 2) + 2 j + l % 2)
__device__ __forceinline__ void sym_store_octet(float *blk, int j, const float (&v)[8])
{
#pragma unroll
   for (int lp = 0; lp < 4; ++lp) *reinterpret_cast<float2 *>(blk + lp * 16 + 2 * j) = make_float2(v[2 * lp], v[2 * lp + 1]);
}
__device__ __forceinline__ void sym_load_octet(const int16_t *p, float (&v)[8])
{
   const int4 q = *reinterpret_cast<const int4 *>(p);          // 16-byte aligned: chunk bases and octet offsets are multiples of 16 bytes
   const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
   for (int k = 0; k < 4; ++k) {
      v[2 * k] = (float)(int16_t)(w[k] & 0xffff) * (1.0f / 32768.0f);
      v[2 * k + 1] = (float)(w[k] >> 16) * (1.0f / 32768.0f);
   }
}
__device__ __forceinline__ void sym_load_octet(const float *p, float (&v)[8])
{
   const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
   v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// stage the (up to) 4 chunks a workgroup's 64 positions touch: reflect pad (tensor.h:931-954); interior octets with one 16-byte load
template <typename T>
__device__ __forceinline__ void sym_stage_chunks(float *xs, const T *__restrict__ pcm, const ItemMap &map, int item0, int n_chunks, int tid)
{
   constexpr int kFlChunks = fl_chunks(1);
   constexpr int kOctets = kPadded / 8;                                      // 224 per chunk
   for (int o = tid; o < kFlChunks * kOctets; o += 256) {
      const int c = o / kOctets, oc = o - c * kOctets;
      const int it = min(item0 + c, n_chunks - 1);
      const T *src = pcm + (size_t)map(it) * kChunk;
      const int idx = oc * 8;                                                // padded index of the octet's first sample
      float v[8];
      if (idx >= kPad && idx < kPad + kChunk) sym_load_octet(src + (idx - kPad), v);
      else {
#pragma unroll
         for (int k = 0; k < 8; ++k) {
            int sidx = idx + k - kPad;
            sidx = sidx < 0 ? -sidx : sidx;
            sidx = sidx >= kChunk ? 2 * (kChunk - 1) - sidx : sidx;
            v[k] = sample_to_f32(src[sidx]);
         }
      }
      sym_store_octet(xs + c * kSymChunkPitch + (idx >> 6) * kFlBlockPitch, (idx >> 3) & 7, v);
   }
}

// OPT (bit mask; NB = 2 only): 1 = the base-bin split a wave serves rotates with the workgroup -- the 9-bin split (17 - 18 trees against 16) then loads every
// SIMD in turn instead of always the one wave 0 lands on; 2 = split 0 runs bin 0 as its batch of one and, when `zero_im0` (the loaded basis' im row of bin 0 is
// all +-0, checked by the engine at create), without that row's tree and without the emit of the row pair that bin 0 does not have.  Both keep every bit
// (magnitudes, Y, partial sums).  Measured (round 4, tools/fe_bench FE_POWER: 2,000 launches per variant, four rounds in turn, 24,576 chunks on 224 CUs, shader
// clock 2.39 GHz throughout): OPT 0 0.4370 ms, OPT 3 0.4354 (-0.4 %) -- the tree is 1 of 66 and the kernel's time is set by the SIMDs' issue rate, not by one wave.
// (Also measured and NOT kept: log1p without its Newton step -- 0.4316 ms, -1.2 %, but max |dp| over the 25,600-chunk parity sweep 3.3e-5 -> 6.3e-5.)
// YP (tools/fe_bench.hip only; the product keeps kFrames): row pitch of Y in floats -- 32 = rows on 128-byte boundaries (round 4's measurement of VERDICT r03 item 8)
template <typename T, int MODE, int NB = 3, int MINW = 4, int OPT = 0, int YP = kFrames>
__global__ __launch_bounds__(256, MINW) void k_frontend_sym(const T *__restrict__ pcm,          // [n_chunks][1536], 16-byte aligned
                                                           const float *__restrict__ basis,    // [258][256] permuted (k_frontend's)
                                                           float *__restrict__ Y,              // [n_chunks][129][25]
                                                           float *__restrict__ FM,             // [kBinSplit][fm_stride] partial bin sums
                                                           int n_chunks, ItemMap map, size_t fm_stride, int zero_im0 = 0)
{
   static_assert(NB == 3 || NB == 2, "base-bin split tables exist for NB = 2, 3");
   constexpr int kFlChunks = fl_chunks(1);
   __shared__ __attribute__((aligned(16))) float xs[kFlChunks * kSymChunkPitch];
   const int tid = threadIdx.x, lane = tid & 63;
   const unsigned bid = (OPT & 8) ? xcd_major_block(blockIdx.x, gridDim.x) : blockIdx.x;      // OPT 8: XCD-major block order (xcd_major_block)
   const int wave = (OPT & 1) ? ((__builtin_amdgcn_readfirstlane(tid >> 6) + (int)bid) & 3) : __builtin_amdgcn_readfirstlane(tid >> 6);   // base-bin split
   const long total_pos = (long)n_chunks * kFrames;
   const long p0 = (long)bid * 64;
   const int item0 = (int)(p0 / kFrames);

   sym_stage_chunks<T>(xs, pcm, map, item0, n_chunks, tid);
   __syncthreads();

   const long pe = p0 + lane;
   const bool writer = pe < total_pos;
   const long pa_ = writer ? pe : total_pos - 1;
   const int item = (int)(pa_ / kFrames), n = (int)(pa_ - (long)item * kFrames);
   const int chunk = map(item);
   typedef __attribute__((address_space(3))) float lds_f;
   const unsigned xaddr = (unsigned)(uintptr_t)(lds_f *)(xs + (item - item0) * kSymChunkPitch + kFlBlockPitch * n);

   const int f_start = sym_first_bin(wave, NB), f_end = sym_end_bin(wave, NB);
   float *yout = Y + (size_t)chunk * (kBins * YP) + n;
   float bin_sum = 0.0f;
   constexpr int kImOffB = kBins * kFilterLen * 4;

   f16v ca, cb, na, nb;
   f4v xc[4], xn[4];
   {
      const float *k0 = basis + (size_t)f_start * kFilterLen;
      VADC_FL_LDS16(xc, xaddr, 0);
      VADC_FL_SLOAD2(ca, cb, k0, fl_tap_off(0, 0, 0), fl_tap_off(0, 0, 0) + kImOffB);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca), "+s"(cb));
      asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]));
      __builtin_amdgcn_sched_barrier(0);
   }
   // rows of base bin b: (b, 128 - b, 64 - b, 64 + b); 64 -+ 0 coincide, and 64 -+ 32 are rows 32 and 96 again
   auto emit_row = [&](int bin, bool counted, float re, float im) {
      const float re2 = re * re, im2 = im * im;
      const float p2 = re2 + im2;
      float val;
      if (MODE == 0) {
         const float x = __builtin_amdgcn_sqrtf(p2) * 1048576.0f;
         val = log1p_hw(x);                                                     // misc.c:42-45
         if (counted) bin_sum += val;                                          // misc.c:55-59
      } else {
         val = sqrtf(p2);                                                      // stft.c:209
      }
      if (writer && counted && !(OPT & 4)) yout[bin * YP] = val;          // OPT 4 (tools/fe_bench only): no Y at all -- what the stores cost, the bin sums keep every value alive
   };
   auto emit = [&](int b, const float *y8) {
      const int bins[4] = {b, 128 - b, 64 - b, 64 + b};
      const bool ok[4] = {true, true, b < 32, b > 0 && b < 32};
#pragma unroll
      for (int q = 0; q < 4; ++q) emit_row(bins[q], ok[q], y8[2 * q], y8[2 * q + 1]);
   };
   // a batch of one base bin from the current pipeline state (ca / cb / xc hold its first stage)
   auto single = [&](int f, auto zi_tag, bool is_bin0) {
      constexpr bool ZI = decltype(zi_tag)::value;
      const float *kf = basis + (size_t)f * kFilterLen;
      SymState<1> st;
      SymStages<1, 0, ZI>::run(st, kf, xaddr, ca, cb, na, nb, xc, xn);
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(st.sa[k]));
      if (is_bin0) {                                                           // rows 0, 128, 64 (64 - 0 and 64 + 0 coincide)
         emit_row(0, true, st.sa[0], st.sa[1]); emit_row(128, true, st.sa[2], st.sa[3]); emit_row(64, true, st.sa[4], st.sa[5]);
      } else emit(f, &st.sa[0]);
   };
   int f_loop = f_start;
   if constexpr (NB == 2 && (OPT & 2)) {
      if (wave == 0) {                                                         // split 0 = bin 0 alone, then bins 1..8 in pairs (same bin order: same partial sums)
         if (zero_im0) single(0, std::integral_constant<bool, true>{}, true);
         else          single(0, std::integral_constant<bool, false>{}, true);
         f_loop = 1;
      }
   }
#pragma unroll 1
   for (int f = f_loop; f + NB <= f_end; f += NB) {
      const float *kf = basis + (size_t)f * kFilterLen;                        // wave-uniform
      SymState<NB> st;
      SymStages<NB, 0>::run(st, kf, xaddr, ca, cb, na, nb, xc, xn);
#pragma unroll
      for (int k = 0; k < 8 * NB; ++k) asm volatile("" : "+v"(st.sa[k]));     // pin the trees above the epilogue (see k_frontend_fl)
#pragma unroll
      for (int B = 0; B < NB; ++B) emit(f + B, &st.sa[8 * B]);
   }
   if constexpr (NB == 2 && !(OPT & 2)) {                                      // NB = 2: wave 0 owns 9 base bins, the odd one as a batch of its own
      if (wave == 0) {
         const int f = f_end - 1;
         const float *kf = basis + (size_t)f * kFilterLen;
         VADC_FL_LDS16(xc, xaddr, 0);
         VADC_FL_SLOAD2(ca, cb, kf, fl_tap_off(0, 0, 0), fl_tap_off(0, 0, 0) + kImOffB);
         asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca), "+s"(cb));
         asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]));
         __builtin_amdgcn_sched_barrier(0);
         single(f, std::integral_constant<bool, false>{}, false);
      }
   }
   if (MODE == 0 && writer) FM[wave * fm_stride + (size_t)chunk * kFrames + n] = bin_sum;   // /129 by the reader (misc.c:60)
}

// Stage tap only: normalized[n][129][25] = Y - mean_t(smooth7(reflect3(FM)))   (misc.c:65-96).
// The engine's normal path folds this subtraction into the first encoder layer.
template <int kFrames>
__device__ __forceinline__ float norm_offset(const float *__restrict__ fmp, size_t fm_stride)
{
   float fm[kFrames];
   for (int q = 0; q < kFrames; ++q) fm[q] = ((fmp[q] + fmp[fm_stride + q]) + (fmp[2 * fm_stride + q] + fmp[3 * fm_stride + q])) / 129.0f;
   const float filt[7] = {0.03663284704089164733887f, 0.11128076165914535522461f, 0.21674531698226928710938f,
                          0.27068215608596801757812f, 0.21674531698226928710938f, 0.11128076165914535522461f,
                          0.03663284704089164733887f};
   float total = 0.0f;
   for (int t = 0; t < kFrames; ++t) {
      float r = 0.0f;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
         int q = t + i - 3;                              // reflect pad 3, no edge repeat
         q = q < 0 ? -q : q;
         q = q >= kFrames ? 2 * (kFrames - 1) - q : q;
         const float pv = fm[q] * filt[i];
         r += pv;
      }
      total += r;
   }
   return total / (float)kFrames;
}

template <int kFrames>
__global__ void k_normalize_tap(const float *__restrict__ Y, const float *__restrict__ FM, float *__restrict__ out, int n_chunks, size_t fm_stride)
{
   const int chunk = blockIdx.x;
   if (chunk >= n_chunks) return;
   const float mm = norm_offset<kFrames>(FM + (size_t)chunk * kFrames, fm_stride);
   for (int i = threadIdx.x; i < kBins * kFrames; i += blockDim.x)
      out[(size_t)chunk * kBins * kFrames + i] = Y[(size_t)chunk * kBins * kFrames + i] - mm;
}

// Stage tap only: inverse of the above for feeding a NORMALIZED or MAGNITUDE tensor into the encoder:
// from magnitudes compute Y and FM with the reference's element order.
__global__ void k_lognorm_from_magnitude(const float *__restrict__ mag, float *__restrict__ Y, float *__restrict__ FM, int n_chunks, size_t fm_stride)
{
   const int chunk = blockIdx.x;
   const int t = threadIdx.x;
   if (chunk >= n_chunks || t >= kFrames) return;
   float s = 0.0f;
   for (int f = 0; f < kBins; ++f) {
      const size_t idx = (size_t)chunk * kBins * kFrames + f * kFrames + t;
      const float v = log1p_hw(mag[idx] * 1048576.0f);
      Y[idx] = v;
      s += v;
   }
   FM[(size_t)chunk * kFrames + t] = s;
   FM[fm_stride + (size_t)chunk * kFrames + t] = 0.0f;
   FM[2 * fm_stride + (size_t)chunk * kFrames + t] = 0.0f;
   FM[3 * fm_stride + (size_t)chunk * kFrames + t] = 0.0f;
}

// n = number of items in this launch (= n_streams * map.cg)
// k_frontend_fl: what runs when the loaded basis lacks the DFT symmetries k_frontend_sym needs, or the input is not 16-byte aligned
void launch_frontend_fl_f32(const float *pcm, const float *basis, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int mode, hipStream_t st)
{
   const dim3 grid((unsigned)(((long)n * kFrames + 63) / 64));
   if (mode == 0) hipLaunchKernelGGL((k_frontend_fl<float, 0, 3, 4, 0, 1>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, nullptr);
   else           hipLaunchKernelGGL((k_frontend_fl<float, 1, 3, 4, 0, 1>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, nullptr);
}

void launch_frontend_fl_s16(const int16_t *pcm, const float *basis, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int mode, hipStream_t st)
{
   const dim3 grid((unsigned)(((long)n * kFrames + 63) / 64));
   (void)mode;                                       // (the magnitude tap -- MODE 1 -- takes f32 samples: vadc_amd_debug_stage_from_samples)
   hipLaunchKernelGGL((k_frontend_fl<int16_t, 0, 3, 4, 0, 1>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, nullptr);
}

// k_frontend_sym: the default v3.1 front end (basis symmetries verified by the engine, pcm 16-byte aligned)
// The kernel's OPT mask is fixed at 3 (rotating splits + bin 0 as a batch of one, its im tree skipped when zero_im0), + 8 = XCD-major block order (xcd_major_block):
// the engine's option "fe_xcd", default on.  (Round 3's kernel -- OPT 0 -- and the (re, im)-packed form k_frontend_ri -- option "fe_opt" = 11: 8 % fewer vector
// instructions, the same time, the same bits -- were options until round 5; the latter is kept under tools/study/ beside its reproducer.)
constexpr int kSymNB = 2;   // tools/fe_bench sym, 16,384 chunks: NB = 2 0.300 ms, NB = 3 0.329 ms (k_frontend_fl: 1.07 ms)
template <typename T>
static void launch_frontend_sym(const T *pcm, const float *basis, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int mode, hipStream_t st, int xcd, int zero_im0)
{
   const dim3 grid((unsigned)(((long)n * kFrames + 63) / 64));
   if constexpr (sizeof(T) == 4) {                   // (the magnitude tap -- MODE 1 -- takes f32 samples: vadc_amd_debug_stage_from_samples)
      if (mode != 0) { hipLaunchKernelGGL((k_frontend_sym<T, 1, kSymNB, 4, 3>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, zero_im0); return; }
   }
   if (xcd)  hipLaunchKernelGGL((k_frontend_sym<T, 0, kSymNB, 4, 11>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, zero_im0);
   else           hipLaunchKernelGGL((k_frontend_sym<T, 0, kSymNB, 4, 3>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, zero_im0);
}
void launch_frontend_sym_f32(const float *pcm, const float *basis, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int mode, hipStream_t st, int xcd, int zero_im0)
{
   launch_frontend_sym<float>(pcm, basis, Y, FM, fm_stride, n, map, mode, st, xcd, zero_im0);
}
void launch_frontend_sym_s16(const int16_t *pcm, const float *basis, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int mode, hipStream_t st, int xcd, int zero_im0)
{
   launch_frontend_sym<int16_t>(pcm, basis, Y, FM, fm_stride, n, map, mode, st, xcd, zero_im0);
}

// Silero v4 geometry (reflect pad 96, 24 frames): Y = log1p(2^20 m), MAG = m, FM = partial bin sums with frame stride 24
void launch_frontend_v4_f32(const float *pcm, const float *basis, float *Y, float *MAG, float *FM, size_t fm_stride, int n, ItemMap map, hipStream_t st)
{
   const long waves = ((long)n * 27 + kLanesOut - 1) / kLanesOut;
   const dim3 grid((unsigned)((waves + 3) / 4), kBinSplit);
   hipLaunchKernelGGL((k_frontend<float, 2, 256, 4, 0, 0, 0, 0, 2, 1>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, MAG);
}

void launch_frontend_v4_s16(const int16_t *pcm, const float *basis, float *Y, float *MAG, float *FM, size_t fm_stride, int n, ItemMap map, hipStream_t st)
{
   const long waves = ((long)n * 27 + kLanesOut - 1) / kLanesOut;
   const dim3 grid((unsigned)((waves + 3) / 4), kBinSplit);
   hipLaunchKernelGGL((k_frontend<int16_t, 2, 256, 4, 0, 0, 0, 0, 2, 1>), grid, dim3(256), 0, st, pcm, basis, Y, FM, n, map, fm_stride, MAG);
}

void launch_normalize_tap(const float *Y, const float *FM, size_t fm_stride, float *out, int n, hipStream_t st, int frames)
{
   if (frames == 24) hipLaunchKernelGGL(k_normalize_tap<24>, dim3(n), dim3(256), 0, st, Y, FM, out, n, fm_stride);
   else              hipLaunchKernelGGL(k_normalize_tap<kFrames>, dim3(n), dim3(256), 0, st, Y, FM, out, n, fm_stride);
}

void launch_lognorm_from_magnitude(const float *mag, float *Y, float *FM, size_t fm_stride, int n, hipStream_t st)
{
   hipLaunchKernelGGL(k_lognorm_from_magnitude, dim3(n), dim3(64), 0, st, mag, Y, FM, n, fm_stride);
}


}  // namespace vadc
