// kernels_layer1_regs.hip -- encoder layer 1 of Silero v3.1 (129 log-magnitude bins x 25 frames -> 16 channels x 13 steps), every activation in registers.
//
// Replaces (reference file:line): adaptive_audio_normalization_inplace misc.c:65-96 (its last step: the per-chunk offset); conv_block conv.c:761-814
// (dw :17-113, pw / proj :532-589); transformer_block transformer.c:13-234 (tensor_linear tensor.h:675-723, softmax :751-784, layer_norm
// misc.c:143-210); conv k = 1 stride 2 + BatchNorm + ReLU transformer.c:237-295 (conv.c:597-709, misc.c:221-258).
//
// Same arithmetic as k_enc_fused (kernels_encoder_fused.hip): W . X ~= Wl . Xh + Wh . Xl + Wh . Xh on the fp16 matrix pipe with fp32 accumulation,
// operands made from accumulator tiles by v_cvt_pk_f16_f32, weights as split-fp16 fragments in LDS, one WAVE per chunk from the front end's output to
// the layer's output with no workgroup barrier in between.  What is particular to this layer:
//
// * 25 steps = TWO 16-column MFMA tiles that OVERLAP: tile 0 carries steps 0..15 and owns 0..12, tile 1 carries steps 9..24 and owns 13..24.  Every
//   owned column has its four depthwise-conv neighbours in its own tile, the chunk's ends coincide with row ends (a DPP row shift brings in the zero
//   padding), so the five taps ride on the multiply-adds as DPP operands with no masks; the columns a tile does not own compute finite garbage that the
//   attention masks out and the store skips.
// * The input (12.9 KB per chunk, the largest array of the whole step) comes in by LDS-DMA: 13 global_load_lds_dwordx4 per chunk copy the chunk's
//   bytes as they lie -- from the 16-byte boundary below its first byte -- into the wave's own LDS buffer, issued for the NEXT chunk as soon as the
//   conv block has read the current one, so HBM sees whole 1 KB requests and the copy hides behind the transformer block.  A B operand's 8 channels of
//   k block kb are channels l1_channel(kb, q, e): conflict-free 4-byte LDS reads (enc_fused_layout.h).
// * K = 258 = relu(dw(x)) of 129 channels | x of 129 channels: 4 + 4 k blocks of 32 and one K = 16 MFMA for the Nyquist channel.
// * D = 16: every GEMM of the transformer block is one v_mfma_f32_16x16x16_f16 M tile (K = 16: an accumulator tile's four registers are the operand).
//   Attention over 25 steps = 2 x 2 score tiles per head; the heads (8 channels each = two lane quads) are separated by zeroing the other head's quads of
//   the K operand, the softmax of a column runs over its 8 registers (two query tiles) and the four lane quads.
#include "common.h"
#include "enc_fused_layout.h"
#include "enc_regs_prims.h"
#include <algorithm>

namespace vadc {

typedef L1Layout LL;
typedef __attribute__((address_space(3))) void l1_lds_void_t;

__device__ __forceinline__ Frag4 lds_frag4(const char *base, int lane)
{
   const h4 *p = reinterpret_cast<const h4 *>(base);
   Frag4 f;
   f.hi = p[lane];
   f.lo = p[64 + lane];
   return f;
}
// c += A . B in the three split terms, in k_enc_fused's order (A = weights: lo . hi, hi . lo, hi . hi)
__device__ __forceinline__ f4 mm3(const Frag4 &a, const Frag4 &b, f4 c)
{
   c = MFMA16K16(a.lo, b.hi, c);
   c = MFMA16K16(a.hi, b.lo, c);
   c = MFMA16K16(a.hi, b.hi, c);
   return c;
}
// the same with the ACTIVATION as the A operand (transposed output): lo term of the activation first, as k_enc_fused's V^T
__device__ __forceinline__ f4 mm3t(const Frag4 &x, const Frag4 &w, f4 c)
{
   c = MFMA16K16(x.lo, w.hi, c);
   c = MFMA16K16(x.hi, w.lo, c);
   c = MFMA16K16(x.hi, w.hi, c);
   return c;
}
__device__ __forceinline__ f4 mm3(const Frag &a, const Frag &b, f4 c)
{
   c = MFMA16(a.lo, b.hi, c);
   c = MFMA16(a.hi, b.lo, c);
   c = MFMA16(a.hi, b.hi, c);
   return c;
}
__device__ __forceinline__ h4 keep(const h4 &v, bool k)
{
   typedef int i2 __attribute__((ext_vector_type(2)));
   i2 u = __builtin_bit_cast(i2, v);
   u[0] = k ? u[0] : 0; u[1] = k ? u[1] : 0;
   return __builtin_bit_cast(h4, u);
}

// four input channels of a k block (half of a B operand): x at this lane's step in both tiles, and the channels' depthwise taps
struct L1Stage { float x0[4], x1[4]; f4 ka[4], kb[4]; };
__device__ __forceinline__ L1Stage l1_load_stage(const float *xb, const float *tp, int kb, int half)
{
   L1Stage s;
#pragma unroll
   for (int c = 0; c < 4; ++c) {
      const int e = 4 * half + c;
      s.x0[c] = xb[(32 * kb + e) * 25];
      s.x1[c] = xb[(32 * kb + e) * 25 + 9];
      s.ka[c] = lds_vec4(tp, (kb * 32 + e) * 8);
      s.kb[c] = lds_vec4(tp, (kb * 32 + e) * 8 + 4);
   }
   return s;
}
// x - offset (misc.c:84-96), relu(dw(x) + bias) (conv.c:17-53) for the stage's four channels in both tiles
__device__ __forceinline__ void l1_stage_math(const L1Stage &s, float off, int lc, f4 &x0, f4 &x1, f4 &d0, f4 &d1)
{
#pragma unroll
   for (int c = 0; c < 4; ++c) {
      x0[c] = s.x0[c] - off;
      x1[c] = s.x1[c] - off;
   }
#pragma unroll
   for (int c = 0; c < 4; ++c) {
      d0[c] = dw5<false>(x0[c], s.ka[c][0], s.ka[c][1], s.ka[c][2], s.ka[c][3], s.kb[c][0], s.kb[c][1], lc);
      d1[c] = dw5<false>(x1[c], s.ka[c][0], s.ka[c][1], s.ka[c][2], s.ka[c][3], s.kb[c][0], s.kb[c][2], lc);
   }
}

// step of column lc of tile t, and whether the tile owns it
__device__ __forceinline__ int l1_step(int t, int lc) { return t == 0 ? lc : 9 + lc; }
__device__ __forceinline__ bool l1_owns(int t, int lc) { return t == 0 ? lc <= 12 : lc >= 4; }

// ---- transformer block + 1x1 conv (BatchNorm and LayerNorm 2 folded) + ReLU on the chunk's two tiles -------------------------------------------
// y: the conv block's output (residual stream) on entry, relu(conv(LN2(...))) on return.  TAP (stage taps for the reference's op-level fixtures):
// 1 = y becomes the attention's result through the out projection (transformer.c:13-153) and the function returns, 2 = the transformer block's result
// (:160-234, LayerNorm 2 with its own scale and shift).
template <int TAP>
__device__ __forceinline__ void l1_block(f4 (&y)[2], const char *img, const float *vec, int lane)
{
   const int q = lane >> 4;
   const Frag4 wq = lds_frag4(img + LL::f_qkv, lane), wk = lds_frag4(img + LL::f_qkv + kFrag4Bytes, lane), wv = lds_frag4(img + LL::f_qkv + 2 * kFrag4Bytes, lane);
   const f4 bq = lds_vec4(vec, LL::v_q_b + 4 * q), bk = lds_vec4(vec, LL::v_k_b + 4 * q);
   Frag4 qf[2], kf[2], vf[2];
#pragma unroll
   for (int t = 0; t < 2; ++t) {
      const Frag4 yf = split4(y[t]);
      const f4 Q = mm3(wq, yf, bq), K = mm3(wk, yf, bk);
      // V^T (lane = channel, registers = steps) by swapping the operands; no V bias: a softmax row sums to 1, the host adds Wo . bv to the out bias
      const f4 VT = mm3t(yf, wv, f4{0.0f, 0.0f, 0.0f, 0.0f});
      qf[t] = split4(Q); kf[t] = split4(K); vf[t] = split4(VT);
   }
   const Frag4 wo = lds_frag4(img + LL::f_out, lane);
   const f4 bo = lds_vec4(vec, LL::v_out_b + 4 * q);
   // softmax mask of this lane's j = 4 q + r in query tile 0 / 1 (transformer.c:104-113: a_i = softmax_j(k_i . q_j)): columns a tile does not own
   f4 smask[2];
#pragma unroll
   for (int r = 0; r < 4; ++r) { smask[0][r] = l1_owns(0, 4 * q + r) ? 0.0f : -1.0e30f; smask[1][r] = l1_owns(1, 4 * q + r) ? 0.0f : -1.0e30f; }
   f4 att[2];
#pragma unroll
   for (int h = 0; h < 2; ++h) {
      const bool mine = (q >> 1) == h;                         // channels 4 q + e of head h live in quads 2 h, 2 h + 1
#pragma unroll
      for (int b = 0; b < 2; ++b) {
         Frag4 kh;
         kh.hi = keep(kf[b].hi, mine); kh.lo = keep(kf[b].lo, mine);
         // S^T[j][i] = sum_c Q[c][j] K[c][i]: lane (q, i) holds s[i][j = 4 q + r] of both query tiles, already scaled by log2(e) / sqrt(hd)
         f4 s0 = mm3(qf[0], kh, smask[0]);
         f4 s1 = mm3(qf[1], kh, smask[1]);
         float m = max2(max2(max2(s0[0], s0[1]), max2(s0[2], s0[3])), max2(max2(s1[0], s1[1]), max2(s1[2], s1[3])));
         m = quads_max(m, lane);
#pragma unroll
         for (int r = 0; r < 4; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r] - m); s1[r] = __builtin_amdgcn_exp2f(s1[r] - m); }      // tensor.h:751-784
         float sum = ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s1[0] + s1[1]) + (s1[2] + s1[3]));
         sum = quads_sum(sum, lane);
         const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
         for (int r = 0; r < 4; ++r) { s0[r] *= inv; s1[r] *= inv; }
         // att[c][i] = sum_j V[c][j] a[i][j]: A = V^T registers (lane = channel, k = step 4 q + r of tile a), B = a (lane = column i)
         f4 o = mm3(vf[0], split4(s0), f4{0.0f, 0.0f, 0.0f, 0.0f});
         o = mm3(vf[1], split4(s1), o);
         if (h == 0) att[b] = o;
         else {
#pragma unroll
            for (int r = 0; r < 4; ++r) att[b][r] = mine ? o[r] : att[b][r];
         }
      }
   }
   // out projection + residual, LN1, FFN, residual, LN2          transformer.c:202-220
   const Frag4 w1 = lds_frag4(img + LL::f_l1, lane), w2 = lds_frag4(img + LL::f_l2, lane);
   const f4 b1 = lds_vec4(vec, LL::v_l1_b + 4 * q), b2 = lds_vec4(vec, LL::v_l2_b + 4 * q);
   Vec<1> n1w = load_vec<1>(vec + LL::v_n1_w, q), n1b = load_vec<1>(vec + LL::v_n1_b, q);
#pragma unroll
   for (int b = 0; b < 2; ++b) {
      const f4 p = mm3(wo, split4(att[b]), bo);
      if (TAP == 1) y[b] = p; else y[b] += p;
   }
   if (TAP == 1) return;
   f4 yy[2][1] = {{y[0]}, {y[1]}};
#pragma unroll
   for (int b = 0; b < 2; ++b) layer_norm<1>(yy[b], n1w, n1b, lane);
   const Frag4 wc = lds_frag4(img + LL::f_cv, lane);
   const f4 bc = lds_vec4(vec, LL::v_cv_b + 4 * q);
   Vec<1> n2w = {}, n2b = {};
   if (TAP == 2) { n2w = load_vec<1>(vec + LL::v_n2_w, q); n2b = load_vec<1>(vec + LL::v_n2_b, q); }
#pragma unroll
   for (int b = 0; b < 2; ++b) {
      f4 f = mm3(w1, split4(yy[b][0]), b1);
#pragma unroll
      for (int r = 0; r < 4; ++r) f[r] = relu(f[r]);
      yy[b][0] += mm3(w2, split4(f), b2);
   }
#pragma unroll
   for (int b = 0; b < 2; ++b) {
      if (TAP == 2) { layer_norm<1, true>(yy[b], n2w, n2b, lane); y[b] = yy[b][0]; }
      else layer_norm<1, false>(yy[b], n2w, n2b, lane);       // scale and shift: folded into the conv below by the host
   }
   if (TAP == 2) return;
   // conv k = 1 (+ folded BatchNorm) -> ReLU, every step (the store keeps the even ones)      transformer.c:279-290
#pragma unroll
   for (int b = 0; b < 2; ++b) {
      f4 z = mm3(wc, split4(yy[b][0]), bc);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[b][r] = relu(z[r]);
   }
}

// NW waves per workgroup, one workgroup per CU (persistent: waves take chunks round-robin); TAP != 0: entered behind the conv block (see l1_block)
template <int NW, int TAP>
__global__ __launch_bounds__(64 * NW) void k_layer1_regs(L1RegsArgs a)
{
   __shared__ __attribute__((aligned(16))) char lds[kL1ImgBytes + (TAP ? 16 : NW * kL1BufBytes)];
   const int tid = threadIdx.x;
   const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, lc = lane & 15;
   const int slot = blockIdx.x * NW + wave, nslots = gridDim.x * NW;
   const char *img = lds;
   const float *vec = reinterpret_cast<const float *>(lds + LL::f_end);
   char *buf = lds + kL1ImgBytes + (TAP ? 0 : wave * kL1BufBytes);

   // the chunk's bytes, from the 16-byte boundary below its first one, into this wave's buffer (13 x 1 KB, the last piece 40 lanes), and the
   // four partial bin sums of its frames (lanes 0..24)
   float fmv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   auto issue_chunk = [&](int item) {
      const int n = a.map(item);
      const size_t cb = (size_t)n * (kL1ChunkFloats * 4);
      const char *g = reinterpret_cast<const char *>(a.y) + (cb & ~(size_t)15) + lane * 16;
      const unsigned dst = (unsigned)(uintptr_t)(l1_lds_void_t *)buf;
#pragma unroll
      for (int j = 0; j < 12; ++j)
         asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst + j * 1024), "v"(g + j * 1024) : "memory");
      if (lane < 40) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst + 12 * 1024), "v"(g + 12 * 1024) : "memory");
      const float *fmp = a.fm + (size_t)n * kFrames + (lane < kFrames ? lane : 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) fmv[k] = fmp[k * a.fm_stride];
   };
   if (!TAP && slot < a.n_chunks) issue_chunk(slot);
   {  // image -> LDS, 8 loads in flight per thread
      const uint4 *src = reinterpret_cast<const uint4 *>(a.img);
      uint4 *dst = reinterpret_cast<uint4 *>(lds);
      constexpr int n = kL1ImgBytes / 16;
      for (int i0 = 0; i0 < n; i0 += 8 * 64 * NW) {
         uint4 v[8];
#pragma unroll
         for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 * NW + tid; v[u] = src[i < n ? i : 0]; }
#pragma unroll
         for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 * NW + tid; if (i < n) dst[i] = v[u]; }
      }
   }
   __syncthreads();

   for (int item = slot; item < a.n_chunks; item += nslots) {
      f4 y[2];
      const int n = a.map(item);
      if constexpr (TAP != 0) {
#pragma unroll
         for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][r] = a.y[(size_t)n * (16 * kFrames) + (4 * q + r) * kFrames + l1_step(t, lc)];
         if (TAP == 3) {                                       // LayerNorm 1 alone (misc.c:143-210)
            const Vec<1> n1w = load_vec<1>(vec + LL::v_n1_w, q), n1b = load_vec<1>(vec + LL::v_n1_b, q);
#pragma unroll
            for (int t = 0; t < 2; ++t) { f4 yy[1] = {y[t]}; layer_norm<1>(yy, n1w, n1b, lane); y[t] = yy[0]; }
         } else l1_block<TAP>(y, img, vec, lane);
#pragma unroll
         for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
               if (l1_owns(t, lc)) a.out[(size_t)n * (16 * kFrames) + (4 * q + r) * kFrames + l1_step(t, lc)] = y[t][r];
         continue;
      } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this chunk's image and partial sums have landed
      // ---- adaptive normalization offset of the chunk (misc.c:65-82): frame means, 7-tap smoothing with reflect padding, mean over the frames ----
      float off;
      {
         const float fms = ((fmv[0] + fmv[1]) + (fmv[2] + fmv[3])) / 129.0f;
         const float filt[7] = {0.03663284704089164733887f, 0.11128076165914535522461f, 0.21674531698226928710938f,
                                0.27068215608596801757812f, 0.21674531698226928710938f, 0.11128076165914535522461f,
                                0.03663284704089164733887f};
         const int t = lane < kFrames ? lane : 0;
         float r = 0.0f;
#pragma unroll
         for (int i = 0; i < 7; ++i) {
            int qq = t + i - 3;                                 // reflect pad 3, no edge repeat
            qq = qq < 0 ? -qq : qq;
            qq = qq >= kFrames ? 2 * (kFrames - 1) - qq : qq;
            r += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * qq, __builtin_bit_cast(int, fms))) * filt[i];
         }
         float total = 0.0f;
#pragma unroll
         for (int tt = 0; tt < kFrames; ++tt) total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), tt));   // the reference's order
         off = total / (float)kFrames;
      }
      // ---- conv block: y = relu(pw(relu(dw(x))) + proj(x))        conv.c:761-814 ----
      const int lead = (int)(((size_t)n * (kL1ChunkFloats * 4)) & 15);
      const float *xb = reinterpret_cast<const float *>(buf + lead) + (16 * (q & 1) + 8 * (q >> 1)) * 25 + lc;
      const float *tp = vec + LL::v_taps + q * 64;
      f4 acc[2];
      acc[0] = acc[1] = lds_vec4(vec, LL::v_cb_b + 4 * q);
      L1Stage sa = l1_load_stage(xb, tp, 0, 0), sb;
      Frag wd = lds_frag(img + LL::f_conv, 0, lane), wx = lds_frag(img + LL::f_conv, 4, lane);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         sb = l1_load_stage(xb, tp, kb, 1);
         __builtin_amdgcn_sched_barrier(0);
         f4 xl0, xl1, dl0, dl1, xh0, xh1, dh0, dh1;
         l1_stage_math(sa, off, lc, xl0, xl1, dl0, dl1);
         Frag wdn = wd, wxn = wx;
         if (kb < 3) {
            sa = l1_load_stage(xb, tp, kb + 1, 0);
            wdn = lds_frag(img + LL::f_conv, kb + 1, lane); wxn = lds_frag(img + LL::f_conv, 4 + kb + 1, lane);
         } else {
            // the Nyquist channel (128), for every lane: k = 0 relu(dw(x)), k = 1 x; only quad 0's weights are not zero
            const float *xt = reinterpret_cast<const float *>(buf + lead) + 128 * 25 + lc;
            sa.x0[0] = xt[0]; sa.x1[0] = xt[9];
            sa.ka[0] = lds_vec4(vec, LL::v_tail); sa.kb[0] = lds_vec4(vec, LL::v_tail + 4);
         }
         __builtin_amdgcn_sched_barrier(0);
         l1_stage_math(sb, off, lc, xh0, xh1, dh0, dh1);
         {
            const Frag df0 = split8(dl0, dh0), xf0 = split8(xl0, xh0);
            const Frag df1 = split8(dl1, dh1), xf1 = split8(xl1, xh1);
            acc[0] = MFMA16(wd.lo, df0.hi, acc[0]); acc[1] = MFMA16(wd.lo, df1.hi, acc[1]);
            acc[0] = MFMA16(wd.hi, df0.lo, acc[0]); acc[1] = MFMA16(wd.hi, df1.lo, acc[1]);
            acc[0] = MFMA16(wd.hi, df0.hi, acc[0]); acc[1] = MFMA16(wd.hi, df1.hi, acc[1]);
            acc[0] = MFMA16(wx.lo, xf0.hi, acc[0]); acc[1] = MFMA16(wx.lo, xf1.hi, acc[1]);
            acc[0] = MFMA16(wx.hi, xf0.lo, acc[0]); acc[1] = MFMA16(wx.hi, xf1.lo, acc[1]);
            acc[0] = MFMA16(wx.hi, xf0.hi, acc[0]); acc[1] = MFMA16(wx.hi, xf1.hi, acc[1]);
         }
         __builtin_amdgcn_sched_barrier(0);
         wd = wdn; wx = wxn;
      }
      {
         const Frag4 wt = lds_frag4(img + LL::f_tail, lane);
         const float x0 = sa.x0[0] - off, x1 = sa.x1[0] - off;
         const float d0 = dw5<false>(x0, sa.ka[0][0], sa.ka[0][1], sa.ka[0][2], sa.ka[0][3], sa.kb[0][0], sa.kb[0][1], lc);
         const float d1 = dw5<false>(x1, sa.ka[0][0], sa.ka[0][1], sa.ka[0][2], sa.ka[0][3], sa.kb[0][0], sa.kb[0][2], lc);
         const Frag4 b0 = split4(f4{d0, x0, 0.0f, 0.0f}), b1 = split4(f4{d1, x1, 0.0f, 0.0f});
         acc[0] = mm3(wt, b0, acc[0]);
         acc[1] = mm3(wt, b1, acc[1]);
      }
      // the conv block has read the chunk: the next one may land in the buffer while the transformer block runs
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (item + nslots < a.n_chunks) issue_chunk(item + nslots);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
         for (int r = 0; r < 4; ++r) y[t][r] = relu(acc[t][r]);
      l1_block<0>(y, img, vec, lane);
      // stride 2: the even steps -- tile 0 lanes 0, 2, .. 12 (steps 0..12), tile 1 lanes 5, 7, .. 15 (steps 14..24)
      float *op = a.out + (size_t)n * (16 * 13) + (4 * q) * 13;
      if ((lc & 1) == 0 && lc <= 12) {
#pragma unroll
         for (int r = 0; r < 4; ++r) op[r * 13 + (lc >> 1)] = y[0][r];
      }
      if ((lc & 1) == 1 && lc >= 5) {
#pragma unroll
         for (int r = 0; r < 4; ++r) op[r * 13 + ((9 + lc) >> 1)] = y[1][r];
      }
      }
   }
}

// max_wgs: workgroups the grid may use (CUs not held by the LSTM chain).  waves: 8 or 10 per workgroup.
void launch_layer1_regs(const L1RegsArgs &a, int max_wgs, int waves, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   if (waves == 10) {
      const int g = std::min(max_wgs, (a.n_chunks + 9) / 10);
      hipLaunchKernelGGL((k_layer1_regs<10, 0>), dim3(g), dim3(640), 0, st, a);
   } else {
      const int g = std::min(max_wgs, (a.n_chunks + 7) / 8);
      hipLaunchKernelGGL((k_layer1_regs<8, 0>), dim3(g), dim3(512), 0, st, a);
   }
}

// stage taps for the op-level fixtures (vadc_amd_debug_layer1_block): a.y = [n][16][25], a.out = [n][16][25]
void launch_layer1_regs_tap(int what, const L1RegsArgs &a, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   const int g = std::min(256, (a.n_chunks + 3) / 4);
   if (what == 1)      hipLaunchKernelGGL((k_layer1_regs<4, 1>), dim3(g), dim3(256), 0, st, a);
   else if (what == 2) hipLaunchKernelGGL((k_layer1_regs<4, 2>), dim3(g), dim3(256), 0, st, a);
   else                hipLaunchKernelGGL((k_layer1_regs<4, 3>), dim3(g), dim3(256), 0, st, a);
}

}  // namespace vadc
