// enc_regs_prims.h -- the register-resident encoder kernels' common pieces (k_enc_fused: layers 2-4, k_layer1_regs: layer 1): split-fp16 MFMA operands
// made straight from accumulator tiles, LDS weight fragments, lane-quad reductions, LayerNorm and the depthwise conv in the accumulator layout.
// Device code only; include inside a .hip translation unit after common.h.
#pragma once
#include "common.h"
#include "enc_fused_layout.h"

namespace vadc {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Frag { h8 hi, lo; };       // one MFMA operand (8 k elements per lane), split: x = hi + lo to 22 significant bits

__device__ __forceinline__ void split2(float a, float b, h2 &hi, h2 &lo)
{
#ifdef VADC_ENC_ABL_NOSPLIT     // timing-only ablation: hi only, lo = hi (one instruction per pair instead of four)
   { const f2 ab_ = {a, b}; hi = __builtin_convertvector(ab_, h2); lo = hi; return; }
#endif
   const f2 ab = {a, b};
   hi = __builtin_convertvector(ab, h2);                    // v_cvt_pk_f16_f32
   // residual a - (float)hi straight from the packed halves (v_fma_mix_f32: f16 source selected by op_sel): one instruction per element where
   // hipcc emits v_cvt_f32_f16 + half a v_pk_add_f32.  Both asm inputs come out of the v_cvt_pk above, never directly out of an MFMA.
   float ra, rb;
   asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
   asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
   const f2 r = {ra, rb};
   lo = __builtin_convertvector(r, h2);
}
__device__ __forceinline__ Frag split8(const f4 &u, const f4 &v)
{
   h2 hi[4], lo[4];
   split2(u[0], u[1], hi[0], lo[0]); split2(u[2], u[3], hi[1], lo[1]);
   split2(v[0], v[1], hi[2], lo[2]); split2(v[2], v[3], hi[3], lo[3]);
   Frag f;
   f.hi = h8{hi[0][0], hi[0][1], hi[1][0], hi[1][1], hi[2][0], hi[2][1], hi[3][0], hi[3][1]};
   f.lo = h8{lo[0][0], lo[0][1], lo[1][0], lo[1][1], lo[2][0], lo[2][1], lo[3][0], lo[3][1]};
   return f;
}
// K = 16 operand (v_mfma_f32_16x16x16_f16: lane (q, row) holds k = 4 q + e, e < 4 -- the four registers of one accumulator tile)
struct Frag4 { h4 hi, lo; };
__device__ __forceinline__ Frag4 split4(const f4 &u)
{
   h2 hi[2], lo[2];
   split2(u[0], u[1], hi[0], lo[0]); split2(u[2], u[3], hi[1], lo[1]);
   Frag4 f;
   f.hi = h4{hi[0][0], hi[0][1], hi[1][0], hi[1][1]};
   f.lo = h4{lo[0][0], lo[0][1], lo[1][0], lo[1][1]};
   return f;
}

// max(x, 0) as ONE instruction (fmaxf costs a canonicalising v_max_f32 x, x in front of the v_max_f32)
__device__ __forceinline__ float relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.0e38f); }

// precision study (tests/reports/enc_terms_report.py): which of the two correction terms of W . X ~= Wl . Xh + Wh . Xl + Wh . Xh the parity bar needs
#ifndef VADC_ENC_WLO
#define VADC_ENC_WLO 1
#endif
#ifndef VADC_ENC_XLO
#define VADC_ENC_XLO 1
#endif
#ifdef VADC_ENC_ABL_NOMFMA      // timing-only ablation (tools/enc_ablate.sh): every MFMA replaced by one dependent vector add per operand pair -- results are wrong
__device__ __forceinline__ f4 fake_mfma(const h8 &a, const h8 &b, f4 c) { c[0] += (float)a[0] * (float)b[0]; return c; }
__device__ __forceinline__ f4 fake_mfma(const h4 &a, const h4 &b, f4 c) { c[0] += (float)a[0] * (float)b[0]; return c; }
#define MFMA16(a, b, c) fake_mfma((a), (b), (c))
#define MFMA16K16(a, b, c) fake_mfma((a), (b), (c))
#else
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define MFMA16K16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16((a), (b), (c), 0, 0, 0)
#endif

// one weight fragment (M tile, k block) of a GEMM from the LDS image
__device__ __forceinline__ Frag lds_frag(const char *base, int idx, int lane)
{
   const h8 *p = reinterpret_cast<const h8 *>(base + (size_t)idx * kFragBytes);
   Frag f;
   f.hi = p[lane];
   f.lo = p[64 + lane];
   return f;
}
__device__ __forceinline__ f4 lds_vec4(const float *v, int off) { return *reinterpret_cast<const f4 *>(v + off); }

// The first two fragments of a GEMM, requested ahead of the vector work (operand split, LayerNorm, softmax) that precedes it: LDS answers
// after 100-200 cycles with twelve waves reading, and a wave has nothing else to issue meanwhile
struct Pre { Frag a0, a1; };
template <int N>
__device__ __forceinline__ Pre prefetch(const char *wf, int idx0, int lane)
{
   Pre p;
   p.a0 = lds_frag(wf, idx0, lane);
   p.a1 = N > 1 ? lds_frag(wf, idx0 + 1, lane) : p.a0;
   return p;
}

// acc[nt][mt] += W[16 mt .., :] . B[nt]      W: MT x KB fragments at `wf` (first M tile = mt0), B: KB operands per tile
// Software pipeline, two fragments ahead (`pre` = fragments mt0 * KB + 0, 1, already on their way); the scheduling barriers keep hipcc from hoisting
// every fragment read of the (fully unrolled) layer to the top of the kernel (it did: 950 spilled registers).  A single accumulation chain of
// v_mfma_f32_16x16x32_f16 issues back to back at the pipe's rate, so the three terms of a tile need no interleaving with other tiles.
template <int NT, int MT, int KB, int MTA>
__device__ __forceinline__ void gemm(f4 (&acc)[NT][MTA], const char *wf, int mt0, const Frag (&b)[NT][KB], int lane, Pre pre)
{
   constexpr int N = MT * KB;
   Frag a0 = pre.a0, a1 = pre.a1;
#pragma unroll
   for (int i = 0; i < N; ++i) {
      const int mt = i / KB, kb = i % KB;
      Frag a2 = a1;
      if (i + 2 < N) a2 = lds_frag(wf, mt0 * KB + i + 2, lane);
#if VADC_ENC_WLO
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = MFMA16(a0.lo, b[nt][kb].hi, acc[nt][mt]);
#endif
#if VADC_ENC_XLO
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = MFMA16(a0.hi, b[nt][kb].lo, acc[nt][mt]);
#endif
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = MFMA16(a0.hi, b[nt][kb].hi, acc[nt][mt]);
      __builtin_amdgcn_sched_barrier(0);
      a0 = a1; a1 = a2;
   }
}
template <int NT, int MT, int KB, int MTA>
__device__ __forceinline__ void gemm(f4 (&acc)[NT][MTA], const char *wf, int mt0, const Frag (&b)[NT][KB], int lane)
{
   gemm<NT, MT, KB, MTA>(acc, wf, mt0, b, lane, prefetch<MT * KB>(wf, mt0 * KB, lane));
}

// MT float4 rows of a vector (this lane's accumulator rows), requested ahead of use
template <int MT>
struct Vec { f4 v[MT]; };
template <int MT>
__device__ __forceinline__ Vec<MT> load_vec(const float *p, int q)
{
   Vec<MT> r;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) r.v[mt] = lds_vec4(p, 16 * mt + 4 * q);
   return r;
}

template <int NT, int MT>
__device__ __forceinline__ void init_bias(f4 (&acc)[NT][MT], const float *bias, int q)
{
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) {
      const f4 b = lds_vec4(bias, 16 * mt + 4 * q);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = b;
   }
}

// sum / max over the four lane-quads (lanes l, l ^ 16, l ^ 32, l ^ 48) through the LDS crossbar (no LDS memory): ds_swizzle for l ^ 16 (bit mode,
// xor mask 0x10 inside each half of the wave), ds_bpermute for l ^ 32 -- two vector instructions per reduction.  (gfx950's v_permlane16_swap /
// v_permlane32_swap do it without LDS in six: a copy, the swap and the add, twice; this kernel is bound by vector-instruction issue, the LDS pipe
// is a quarter busy.  Through __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) ties both operands to one register or folds the two results into
// one: tools/enc_prims_test.hip keeps the inline-asm form that works.)
__device__ __forceinline__ float max2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.0e38f); }   // one v_med3_f32, no canonicalising v_max_f32 x, x
template <bool MAX>
__device__ __forceinline__ float quads_reduce(float v, int lane)
{
   const float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
   v = MAX ? max2(v, a) : v + a;
   const float b = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __builtin_bit_cast(int, v)));
   return MAX ? max2(v, b) : v + b;
}
// N independent reductions in lockstep: the N swizzles, then the N permutes are in flight together -- one LDS-crossbar round trip per step instead of N
template <bool MAX, int N>
__device__ __forceinline__ void quads_reduce_n(float (&v)[N], int lane)
{
   float a[N];
#pragma unroll
   for (int i = 0; i < N; ++i) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v[i]), 0x401F));
#pragma unroll
   for (int i = 0; i < N; ++i) v[i] = MAX ? max2(v[i], a[i]) : v[i] + a[i];
#pragma unroll
   for (int i = 0; i < N; ++i) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __builtin_bit_cast(int, v[i])));
#pragma unroll
   for (int i = 0; i < N; ++i) v[i] = MAX ? max2(v[i], a[i]) : v[i] + a[i];
}
__device__ __forceinline__ float quads_sum(float v, int lane) { return quads_reduce<false>(v, lane); }
__device__ __forceinline__ float quads_max(float v, int lane) { return quads_reduce<true>(v, lane); }

// LayerNorm over the D = 16 MT channels of each column in the accumulator layout (misc.c:143-210: biased variance, eps 1e-5)
// AFFINE = false: the scale and shift live in the next GEMM's weights and bias (LayerNorm 2 feeds the strided conv only: folded by the host)
template <int MT, bool AFFINE = true>
__device__ __forceinline__ void layer_norm(f4 (&x)[MT], const Vec<MT> &w, const Vec<MT> &b, int lane)
{
   constexpr int D = 16 * MT;
   float s = 0.0f;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) s += (x[mt][0] + x[mt][1]) + (x[mt][2] + x[mt][3]);
   s = quads_sum(s, lane);
   const float mean = s * (1.0f / D);
   float vs = 0.0f;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = x[mt][r] - mean; vs = fmaf(d, d, vs); }
   vs = quads_sum(vs, lane);
   const float rstd = __builtin_amdgcn_rsqf(vs * (1.0f / D) + 1e-5f);
   const float mr = mean * rstd;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) {
      const f4 w4 = w.v[mt], b4 = b.v[mt];
#pragma unroll
      for (int r = 0; r < 4; ++r) x[mt][r] = AFFINE ? fmaf(fmaf(x[mt][r], rstd, -mr), w4[r], b4[r]) : fmaf(x[mt][r], rstd, -mr);
   }
}

// LayerNorm of N tiles with D = 16 in lockstep (the same arithmetic per tile as layer_norm<1>): the tiles' reductions share their round trips
template <int N, bool AFFINE>
__device__ __forceinline__ void layer_norm16_n(f4 (&x)[N], const f4 &w4, const f4 &b4, int lane)
{
   float s[N];
#pragma unroll
   for (int i = 0; i < N; ++i) s[i] = 0.0f + ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3]));
   quads_reduce_n<false, N>(s, lane);
   float mean[N], vs[N];
#pragma unroll
   for (int i = 0; i < N; ++i) {
      mean[i] = s[i] * (1.0f / 16);
      vs[i] = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = x[i][r] - mean[i]; vs[i] = fmaf(d, d, vs[i]); }
   }
   quads_reduce_n<false, N>(vs, lane);
#pragma unroll
   for (int i = 0; i < N; ++i) {
      const float rstd = __builtin_amdgcn_rsqf(vs[i] * (1.0f / 16) + 1e-5f);
      const float mr = mean[i] * rstd;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[i][r] = AFFINE ? fmaf(fmaf(x[i][r], rstd, -mr), w4[r], b4[r]) : fmaf(x[i][r], rstd, -mr);
   }
}

// The pair layout of layers 3 / 4: two 7-step chunks in a 16-column tile at columns 0..6 and 8..14, columns 7 and 15 dead (zero).  The second chunk sits
// EXACTLY 8 columns behind the first: the attention's MFMAs and the softmax sum over the tile's 16 positions in a fixed internal grouping, and a chunk's
// bits must not depend on whether it is the first or the second of its pair (a stream's results are bit-identical for any split into calls:
// test_backend_run_shape_and_batch_invariance -- columns 0..6 / 9..15 with two dead columns between them failed exactly that).
__device__ __forceinline__ int pair_chunk(int lc) { return lc >> 3; }
__device__ __forceinline__ int pair_step(int lc) { return lc & 7; }                             // 7: the dead column
__device__ __forceinline__ bool pair_live(int lc) { return (lc & 7) != 7; }
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v)
{
   return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// depthwise k = 5, zero pad 2, + bias, ReLU (conv.c:17-53) of one register: x at this lane's step, neighbours by row shifts of x itself, carried by the
// multiply-adds as their DPP operand (hipcc does not fold a row shift into v_fmac: inline asm; the s_nop covers the VALU-write -> DPP-read wait states
// for x, which the caller has just written).  Columns outside a chunk hold zeros, so the zero padding comes with the shift -- except, in the pair layout
// (one dead column between the chunks), for the distance-2 taps of columns 6 and 8, which would reach the other chunk: those two taps are shifted into
// a register and masked.  The same order of the five products as the reference's loop.
template <bool PAIR>
__device__ __forceinline__ float dw5(float x, float k0, float k1, float k2, float k3, float k4, float bias, int lc)
{
   float dv = bias;
   if (PAIR) {
      const float sm2 = dpp_row<0x112>(x);                       // row_shr:2; shifted FIRST, by every lane: inside `lc == 8 ? 0 : shift` the shift is
      const float xm2 = lc == 8 ? 0.0f : sm2;                    // the arm of a branch, and a lane the branch has switched off reads as 0 to its neighbours
      dv = fmaf(xm2, k0, dv);
      asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(x), "v"(k1));   // (behind the shift, the select and the multiply-add above)
   } else {
      asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(x), "v"(k0));
      asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(x), "v"(k1));
   }
   dv = fmaf(x, k2, dv);
   asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(x), "v"(k3));
   if (PAIR) {
      const float sp2 = dpp_row<0x102>(x);                       // row_shl:2
      const float xp2 = lc == 6 ? 0.0f : sp2;
      dv = fmaf(xp2, k4, dv);
   } else {
      asm("v_fmac_f32_dpp %0, %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(x), "v"(k4));
   }
   return relu(dv);
}

}  // namespace vadc
