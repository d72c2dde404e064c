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
//   The three split terms of such a product are two matrix instructions (see mm).  Attention over 25 steps = 2 x 2 score tiles per head; the heads (8 channels each = two lane quads) are separated by zeroing the other head's quads of
//   the K operand, the softmax of a column runs over its 8 registers (two query tiles) and the four lane quads.
#include "common.h"
#include "enc_fused_layout.h"
#include "enc_regs_prims.h"
#include "l1_regs_prims.h"
#include <algorithm>

namespace vadc {

typedef L1Layout LL;

#ifdef VADC_L1R_PHASE_PROF      // cycle stamps of one wave per workgroup (tools/l1r_phases.sh): where a chunk's time goes
__device__ unsigned long long g_l1r_phase[8];
#define L1R_PH(i) do { if (ph_on) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_l1r_phase[i], t_ - ph_t); ph_t = t_; } } while (0)
#else
#define L1R_PH(i) do { } while (0)
#endif

// one input channel of a k block on its way out of LDS: x at this lane's step in both tiles, and the channel's depthwise taps.  xb = the k block's slab
// + the chunk's lead + this lane's (quad, column) part; tp = the taps of this lane's quad
struct L1Chan { f2 x; f4 ka, kb; };      // x = (tile 0, tile 1)
__device__ __forceinline__ L1Chan l1_load_chan(const float *xb, const float *tp, int kb, int e)
{
   L1Chan s;
   s.x = f2{xb[e * 25], xb[e * 25 + 9]};
   s.ka = lds_vec4(tp, (kb * 8 + e) * 32);
   s.kb = lds_vec4(tp, (kb * 8 + e) * 32 + 4);
   return s;
}
// x - offset (misc.c:84-96), relu(dw(x) + bias) of the channel in both tiles; (x0, x1) are one register pair as they come out of the LDS read: the
// subtraction is one packed instruction
__device__ __forceinline__ void l1_channel_math(const L1Chan &s, float off, float &x0, float &x1, float &d0, float &d1)
{
   const f2 xp = s.x - f2{off, off};
   x0 = xp[0]; x1 = xp[1];
   float a, b;
   dw5x2(xp[0], xp[1], s.ka[0], s.ka[1], s.ka[2], s.ka[3], s.kb[0], s.kb[1], a, b);
   d0 = relu(a); d1 = relu(b);
}

// step of column lc of tile t, and whether the tile owns it
__device__ __forceinline__ int l1_step(int t, int lc) { return t == 0 ? lc : 9 + lc; }
__device__ __forceinline__ bool l1_owns(int t, int lc) { return t == 0 ? lc <= 12 : lc >= 4; }

// ---- transformer block + 1x1 conv (BatchNorm and LayerNorm 2 folded) + ReLU on the chunk's two tiles -------------------------------------------
// y: the conv block's output (residual stream) on entry, relu(conv(LN2(...))) on return.  TAP (stage taps for the reference's op-level fixtures):
// 1 = y becomes the attention's result through the out projection (transformer.c:13-153) and the function returns, 2 = the transformer block's result
// (:160-234, LayerNorm 2 with its own scale and shift), 5 = only the tail runs: y is what the strided conv consumes (transformer.c:279-290; batch_norm misc.c:98-141).
template <int TAP>
__device__ __forceinline__ void l1_block(f4 (&y)[2], const char *img, const float *vec, int lane)
{
   const int q = lane >> 4;
   if constexpr (TAP == 5) {                                    // the layer's tail alone: conv k = 1 with BatchNorm (and LayerNorm 2's affine) folded in, ReLU
      const AOp wc = lds_aop(img + LL::f_cv, lane);
      const f4 bc = lds_vec4(vec, LL::v_cv_b + 4 * q);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
         const f4 z = mm(wc, split4_hl(y[b]), bc);
#pragma unroll
         for (int r = 0; r < 4; ++r) y[b][r] = relu(z[r]);
      }
      return;
   }
   const AOp wq = lds_aop(img + LL::f_qkv, lane), wk = lds_aop(img + LL::f_qkv + kFrag4Bytes, lane), wv = lds_aop(img + LL::f_qkv + 2 * kFrag4Bytes, lane);
   const f4 bq = lds_vec4(vec, LL::v_q_b + 4 * q), bk = lds_vec4(vec, LL::v_k_b + 4 * q);
   AOp qf[2], vf[2];
   h8 kf[2];
#pragma unroll
   for (int t = 0; t < 2; ++t) {
      const h8 yf = split4_hl(y[t]);
      const f4 Q = mm(wq, yf, bq), K = mm(wk, yf, bk);
      // V^T (lane = channel, registers = steps) by swapping the operands; no V bias: a softmax row sums to 1, the host adds Wo . bv to the out bias
      const f4 VT = mmt(yf, wv, f4{0.0f, 0.0f, 0.0f, 0.0f});
      qf[t] = split4_a(Q); kf[t] = split4_hl(K); vf[t] = split4_a(VT);
   }
   const AOp wo = lds_aop(img + LL::f_out, lane);
   const f4 bo = lds_vec4(vec, LL::v_out_b + 4 * q);
   // softmax mask of this lane's j = 4 q + r in query tile 0 / 1 (transformer.c:104-113: a_i = softmax_j(k_i . q_j)): columns a tile does not own
   f4 smask[2];
#pragma unroll
   for (int r = 0; r < 4; ++r) { smask[0][r] = l1_owns(0, 4 * q + r) ? 0.0f : -1.0e30f; smask[1][r] = l1_owns(1, 4 * q + r) ? 0.0f : -1.0e30f; }
   // The four (head, key tile) softmaxes run in LOCKSTEP: each is a chain of four LDS-crossbar round trips (max and sum over the lane quads) that a
   // wave can only wait for; side by side the four chains share their waits.
   f4 s[4][2];                                                  // [2 h + b][query tile]
#pragma unroll
   for (int hb = 0; hb < 4; ++hb) {
      const h8 kh = keep(kf[hb & 1], (q >> 1) == (hb >> 1));    // channels 4 q + e of head h live in quads 2 h, 2 h + 1
      // S^T[j][i] = sum_c Q[c][j] K[c][i]: lane (q, i) holds s[i][j = 4 q + r] of both query tiles, already scaled by log2(e) / sqrt(hd)
      s[hb][0] = mm(qf[0], kh, smask[0]);
      s[hb][1] = mm(qf[1], kh, smask[1]);
   }
   float m[4];
#pragma unroll
   for (int hb = 0; hb < 4; ++hb)
      m[hb] = max2(max2(max2(s[hb][0][0], s[hb][0][1]), max2(s[hb][0][2], s[hb][0][3])), max2(max2(s[hb][1][0], s[hb][1][1]), max2(s[hb][1][2], s[hb][1][3])));
   quads_reduce_n<true, 4>(m, lane);
   float sum[4];
#pragma unroll
   for (int hb = 0; hb < 4; ++hb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[hb][0][r] = __builtin_amdgcn_exp2f(s[hb][0][r] - m[hb]); s[hb][1][r] = __builtin_amdgcn_exp2f(s[hb][1][r] - m[hb]); }      // tensor.h:751-784
      sum[hb] = ((s[hb][0][0] + s[hb][0][1]) + (s[hb][0][2] + s[hb][0][3])) + ((s[hb][1][0] + s[hb][1][1]) + (s[hb][1][2] + s[hb][1][3]));
   }
   quads_reduce_n<false, 4>(sum, lane);
   f4 att[2];
#pragma unroll
   for (int hb = 0; hb < 4; ++hb) {
      const int h = hb >> 1, b = hb & 1;
      const bool mine = (q >> 1) == h;
      const float inv = __builtin_amdgcn_rcpf(sum[hb]);
      // att[c][i] = sum_j V[c][j] a[i][j]: A = V^T registers (lane = channel, k = step 4 q + r of tile a), B = exp2(s - max) (lane = column i);
      // the division by the row sum on the four results instead of the eight weights
      f4 o = mm(vf[0], split4_hl(s[hb][0]), f4{0.0f, 0.0f, 0.0f, 0.0f});
      o = mm(vf[1], split4_hl(s[hb][1]), o);
#pragma unroll
      for (int r = 0; r < 4; ++r) att[b][r] = (h == 0 || mine) ? o[r] * inv : att[b][r];
   }
   // out projection + residual, LN1, FFN, residual, LN2          transformer.c:202-220
   const AOp w1 = lds_aop(img + LL::f_l1, lane), w2 = lds_aop(img + LL::f_l2, lane);
   const f4 b1 = lds_vec4(vec, LL::v_l1_b + 4 * q), b2 = lds_vec4(vec, LL::v_l2_b + 4 * q);
   Vec<1> n1w = load_vec<1>(vec + LL::v_n1_w, q), n1b = load_vec<1>(vec + LL::v_n1_b, q);
#pragma unroll
   for (int b = 0; b < 2; ++b) {
      const f4 p = mm(wo, split4_hl(att[b]), bo);
      if (TAP == 1) y[b] = p; else y[b] += p;
   }
   if (TAP == 1) return;
   f4 yy[2] = {y[0], y[1]};
   layer_norm16_n<2, true>(yy, n1w.v[0], n1b.v[0], lane);
   const AOp wc = lds_aop(img + LL::f_cv, lane);
   const f4 bc = lds_vec4(vec, LL::v_cv_b + 4 * q);
   Vec<1> n2w = {}, n2b = {};
   if (TAP == 2) { n2w = load_vec<1>(vec + LL::v_n2_w, q); n2b = load_vec<1>(vec + LL::v_n2_b, q); }
   {
      f4 f[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
         f[b] = mm(w1, split4_hl(yy[b]), b1);
#pragma unroll
         for (int r = 0; r < 4; ++r) f[b][r] = relu(f[b][r]);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) yy[b] += mm(w2, split4_hl(f[b]), b2);
   }
   if (TAP == 2) { layer_norm16_n<2, true>(yy, n2w.v[0], n2b.v[0], lane); y[0] = yy[0]; y[1] = yy[1]; }
   else layer_norm16_n<2, false>(yy, n2w.v[0], n2b.v[0], lane);      // scale and shift: folded into the conv below by the host
   if (TAP == 2) return;
   // conv k = 1 (+ folded BatchNorm) -> ReLU, every step (the store keeps the even ones)      transformer.c:279-290
#pragma unroll
   for (int b = 0; b < 2; ++b) {
      const f4 z = mm(wc, split4_hl(yy[b]), bc);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[b][r] = relu(z[r]);
   }
}

// NW waves per workgroup, one workgroup per CU (persistent: waves take chunks round-robin); TAP 1, 2, 3, 5: entered behind the conv block (see l1_block) with
// a.y = [n][16][25]; TAP 4: the whole input path and the conv block as in the product (a.y = the chunk's [129][25], a.fm its partial sums), left behind the
// conv block's ReLU (conv.c:761-814) with every step of the chunk written as [n][16][25]
template <int NW, int TAP>
__global__ __launch_bounds__(64 * NW) void k_layer1_regs(L1RegsArgs a)
{
   constexpr bool ENTER = TAP != 0 && TAP != 4;                 // the chunk enters behind the conv block: no input pipeline
   __shared__ __attribute__((aligned(16))) char lds[kL1ImgBytes + (ENTER ? 16 : NW * kL1RingBytes)];
   const int tid = threadIdx.x;
   const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, lc = lane & 15;
   // wave-major: the last, partial round of the grid's (waves x workgroups) slots is then one or two waves on EVERY CU -- a wave alone on its SIMD runs
   // twice as fast -- instead of full workgroups on a few CUs beside idle ones (24,576 chunks on 224 CUs x 12 waves: 9.14 rounds)
   const int slot = wave * gridDim.x + blockIdx.x, nslots = gridDim.x * NW;
   const char *img = lds;
   const float *vec = reinterpret_cast<const float *>(lds + LL::f_end);
   char *buf = lds + kL1ImgBytes + (ENTER ? 0 : wave * kL1RingBytes);

   // ---- the input pipeline -----------------------------------------------------------------------------------------------------------------
   // A chunk travels in four GROUPS, one per k block: k block kb reads bytes [3200 kb, 3200 (kb + 1)) of the chunk (kb = 3: and the Nyquist channel's
   // 100 behind them), and group kb is the 208 16-byte units from the 16-byte boundary below byte 3200 kb -- four LDS-DMA instructions, the last with 16
   // lanes -- copied as they lie into one SLAB (3,328 B) of the wave's ring of three: group j of the wave's sequence (j = 4 * iteration + kb) lives in
   // slab j mod 3, and group j + 3 is issued as soon as k block j's last LDS read has returned.  A wave so holds 10 KB of LDS instead of a whole chunk
   // (12.9 KB: round 4), which lets a third wave per SIMD in -- 12 x 10 KB + the 35-KB weight image = 155 of 160 KB -- and every group has two to
   // three k blocks' time to arrive.  The four partial bin sums of the chunk's frames (lanes 0..24) are loaded by inline asm as well, ahead of the
   // group that follows k block 0: hipcc does not count the DMA instructions, and its own s_waitcnt for a load it knows would wait for all of them.
   //
   // The waits count LOADS only.  Loads complete in issue order among themselves, but a store may complete before an older load: a wait that counted
   // on the four stores of an iteration still being outstanding could pass with its load in flight.  Each wait below allows as many outstanding
   // operations as there are loads YOUNGER than the one it needs; stores still in flight then make it wait a little longer than it must, never less.
   // Issue order per iteration i: [W0] k block 0 (W1 inside, before its 7th channel) -> sums(i + 1), G(i, 3) -> k block 1 (W2) -> G(i + 1, 0) ->
   // k block 2 (W3) -> G(i + 1, 1) -> k block 3 -> G(i + 1, 2) -> 4 stores.  Younger loads at W0: G(i, 1), G(i, 2) = 8; W1: G(i, 2) = 4;
   // W2: sums(i + 1), G(i, 3) = 8 (last iteration: 4); W3: G(i + 1, 0) = 4 (last iteration: 0).
   float fmv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   const int lo16 = lane * 16;
   auto chunk_base = [&](int nn) -> const char * {
      const size_t cb = (size_t)nn * (kL1ChunkFloats * 4);
      return reinterpret_cast<const char *>(a.y) + (cb & ~(size_t)15);              // wave-uniform: scalar base + lane offset, no vector address arithmetic
   };
   auto issue_sums = [&](int nn) {                             // nn = the chunk's index map(item)
      const float *fmp = a.fm + (size_t)nn * kFrames;
      const int lo4 = (lane < kFrames ? lane : 0) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("global_load_dword %0, %1, %2" : "=v"(fmv[k]) : "v"(lo4), "s"(fmp + k * a.fm_stride) : "memory");
   };
   int ring = 0;                                              // slab of k block 0 of the current chunk (wave-uniform): (4 * iteration) mod 3
   auto slab_of = [&](int kb) { const int r = ring + kb; return r >= 3 ? r - 3 : r; };      // kb <= 3, ring <= 2: r <= 5
   auto issue_group = [&](int nn, int kb, int slab) {
      const char *src = chunk_base(nn) + 3200 * kb;
      const unsigned dst = (unsigned)(uintptr_t)(l1_lds_void_t *)buf + (unsigned)slab * kL1SlabBytes;
#ifndef VADC_L1R_ABL_NODMA        // (timing-only ablations, tools/l1r_ablate.sh: results are wrong)
#pragma unroll
      for (int j = 0; j < 3; ++j)
         asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + j * 1024), "v"(lo16), "s"(src + j * 1024) : "memory");
      if (lane < 16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + 3 * 1024), "v"(lo16), "s"(src + 3 * 1024) : "memory");
#else
      asm volatile("" :: "s"(dst), "s"(src), "v"(lo16));
#endif
   };
// wait until at most N of the wave's vector-memory operations are outstanding; `more`: whether this iteration issues for a next chunk
#define L1R_WAIT(more, n_more, n_last) do { if (more) asm volatile("s_waitcnt vmcnt(" #n_more ")" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #n_last ")" ::: "memory"); } while (0)
   if (!ENTER && slot < a.n_chunks) {
      const int n0 = a.map(slot);
      issue_sums(n0);
#pragma unroll
      for (int g = 0; g < 3; ++g) issue_group(n0, g, g);
   }
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
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#ifdef VADC_L1R_PHASE_PROF
   const bool ph_on = lane == 0 && wave == 3;
   unsigned long long ph_t = __builtin_readcyclecounter();
#endif
   for (int item = slot; item < a.n_chunks; item += nslots) {
      f4 y[2];
      const int n = a.map(item);
      if constexpr (ENTER) {
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
      const bool more = item + nslots < a.n_chunks;           // wave-uniform
      const int nnext = more ? a.map(item + nslots) : n;      // (the index arithmetic of the next chunk once per iteration, in scalar registers)
      L1R_PH(0);
      // W0: the chunk's sums and group 0 (younger loads: groups 1 and 2)
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(fmv[0]), "+v"(fmv[1]), "+v"(fmv[2]), "+v"(fmv[3]) :: "memory");
      L1R_PH(1);
      // ---- adaptive normalization offset of the chunk (misc.c:65-82): frame means, 7-tap smoothing with reflect padding, mean over the frames ----
      float off;
#ifdef VADC_L1R_ABL_NONORM
      off = fmv[0];
#else
      {
         const float fms = ((fmv[0] + fmv[1]) + (fmv[2] + fmv[3])) / 129.0f;
         const float filt[7] = {0.03663284704089164733887f, 0.11128076165914535522461f, 0.21674531698226928710938f,
                                0.27068215608596801757812f, 0.21674531698226928710938f, 0.11128076165914535522461f,
                                0.03663284704089164733887f};
         const int t = lane < kFrames ? lane : 0;
         float nb[7];
#pragma unroll
         for (int i = 0; i < 7; ++i) {
            int qq = t + i - 3;                                 // reflect pad 3, no edge repeat
            qq = qq < 0 ? -qq : qq;
            qq = qq >= kFrames ? 2 * (kFrames - 1) - qq : qq;
            nb[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * qq, __builtin_bit_cast(int, fms)));
         }
         float r = 0.0f;
#pragma unroll
         for (int i = 0; i < 7; ++i) r += nb[i] * filt[i];
         r = lane < kFrames ? r : 0.0f;
         // the sum of the 25 smoothed means as a tree over the lanes (row shifts, then the two rows): the reference adds them one after the other,
         // which as 25 v_readlane + add pairs took a fifth of a wave's time (each a trip through the scalar registers); the two orders differ by
         // a few units in the last place of an offset that is subtracted from values of its own size
         r += dpp_row<0x111>(r); r += dpp_row<0x112>(r); r += dpp_row<0x114>(r); r += dpp_row<0x118>(r);     // row_shr:1, 2, 4, 8: lane 15 of a row = its sum
         const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 15)) +
                             __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 31));
         off = total / (float)kFrames;
      }
#endif
      L1R_PH(2);
      // ---- conv block: y = relu(pw(relu(dw(x))) + proj(x))        conv.c:761-814 ----
      const int lead = (int)(((size_t)n * (kL1ChunkFloats * 4)) & 15);
      const int lane_part = lead + ((16 * (q & 1) + 8 * (q >> 1)) * 25 + lc) * 4;      // this lane's (quad, column) inside a slab
      auto slab_ptr = [&](int kb) { return reinterpret_cast<const float *>(buf + slab_of(kb) * kL1SlabBytes + lane_part); };
      const float *tp = vec + LL::v_taps + q * 8;
      f4 acc[2];
      acc[0] = acc[1] = lds_vec4(vec, LL::v_cb_b + 4 * q);
      // The channels come out of LDS two ahead of the one being worked on (a window of three: 30 registers -- two stages of four channels each held 80,
      // and with the weights of three k blocks the kernel stood at 208 registers, two waves per SIMD); the 12 MFMAs of k block kb - 1 are issued ONE AT
      // A TIME between the channels of k block kb (about 14 vector instructions each), not in a clump behind their operands' splits, and their weights
      // are fetched when the previous k block's have been used.  The scheduling barriers pin that order.
      L1Chan ch[3];
      const float *xb = slab_ptr(0);
      ch[0] = l1_load_chan(xb, tp, 0, 0);
      ch[1] = l1_load_chan(xb, tp, 0, 1);
      Frag pd0, pd1, px0, px1, pwd, pwx;                         // pending: operands and weights of the previous k block
      auto pending_mfma = [&](int i) {
         const int t = i & 1, term = i >> 1;
         const Frag &w = term < 3 ? pwd : pwx;
         const Frag &o = term < 3 ? (t ? pd1 : pd0) : (t ? px1 : px0);
         const int k = term % 3;
         acc[t] = k == 0 ? MFMA16(w.lo, o.hi, acc[t]) : (k == 1 ? MFMA16(w.hi, o.lo, acc[t]) : MFMA16(w.hi, o.hi, acc[t]));
      };
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         const bool have = kb > 0;
         f4 xl0, xl1, dl0, dl1, xh0, xh1, dh0, dh1;
         const float *xbn = xb;
#pragma unroll
         for (int c = 0; c < 8; ++c) {
            const int j = 8 * kb + c;                           // the channel's place in the chunk's sequence of 33
            if (c == 6 && kb < 3) {                             // the next k block's first channels are read from here on: its group must have landed
               if (kb == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // W1
               if (kb == 1) L1R_WAIT(more, 8, 4);                                 // W2
               if (kb == 2) L1R_WAIT(more, 4, 0);                                 // W3
               xbn = slab_ptr(kb + 1);
            }
            if (j + 2 < 32) ch[(j + 2) % 3] = l1_load_chan(c + 2 < 8 ? xb : xbn, tp, (j + 2) >> 3, (j + 2) & 7);
            else if (j + 2 == 32) {
               // the Nyquist channel (128), for every lane: k = 0 relu(dw(x)), k = 1 x; only quad 0's weights are not zero.  It lies behind k block 3's 32
               // channels in the same slab
               const float *xt = reinterpret_cast<const float *>(buf + slab_of(3) * kL1SlabBytes + lead) + 32 * 25 + lc;
               ch[2].x = f2{xt[0], xt[9]};
               ch[2].ka = lds_vec4(vec, LL::v_tail); ch[2].kb = lds_vec4(vec, LL::v_tail + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
               float x0, x1, d0, d1;
               l1_channel_math(ch[j % 3], off, x0, x1, d0, d1);
               if (c < 4) { xl0[c] = x0; xl1[c] = x1; dl0[c] = d0; dl1[c] = d1; }
               else       { xh0[c - 4] = x0; xh1[c - 4] = x1; dh0[c - 4] = d0; dh1[c - 4] = d1; }
            }
            if (have) pending_mfma(c);
            __builtin_amdgcn_sched_barrier(0);
         }
         // k block kb has been read (every value of its channels has been used): the group three further on may overwrite its slab
         if (kb == 0) {
            if (more) issue_sums(nnext);
            issue_group(n, 3, slab_of(0));
         } else if (more && kb < 3) issue_group(nnext, kb - 1, slab_of(kb));      // (k block 3's slab also holds the Nyquist channel: behind the tail's read)
         const Frag df0 = split8(dl0, dh0);
         if (have) pending_mfma(8);
         __builtin_amdgcn_sched_barrier(0);
         const Frag df1 = split8(dl1, dh1);
         if (have) pending_mfma(9);
         __builtin_amdgcn_sched_barrier(0);
         const Frag xf0 = split8(xl0, xh0);
         if (have) pending_mfma(10);
         __builtin_amdgcn_sched_barrier(0);
         const Frag xf1 = split8(xl1, xh1);
         if (have) pending_mfma(11);
         __builtin_amdgcn_sched_barrier(0);
         pd0 = df0; pd1 = df1; px0 = xf0; px1 = xf1;
         pwd = lds_frag(img + LL::f_conv, kb, lane); pwx = lds_frag(img + LL::f_conv, 4 + kb, lane);
         xb = xbn;
      }
      {
         const AOp wt = lds_aop(img + LL::f_tail, lane);
         float x0, x1, d0, d1;
         l1_channel_math(ch[2], off, x0, x1, d0, d1);           // (uses the last LDS read of the chunk)
         if (more) issue_group(nnext, 2, slab_of(3));
#pragma unroll
         for (int i = 0; i < 6; ++i) pending_mfma(i);
         const h8 b0 = split4_hl(f4{d0, x0, 0.0f, 0.0f}), b1 = split4_hl(f4{d1, x1, 0.0f, 0.0f});
#pragma unroll
         for (int i = 6; i < 12; ++i) pending_mfma(i);
         acc[0] = mm(wt, b0, acc[0]);
         acc[1] = mm(wt, b1, acc[1]);
      }
      ring = slab_of(1);                                         // four groups on: (ring + 4) mod 3
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
         for (int r = 0; r < 4; ++r) y[t][r] = relu(acc[t][r]);
#ifdef VADC_L1R_PHASE_PROF
      asm volatile("" :: "v"(y[0][0]), "v"(y[1][3]));
#endif
      L1R_PH(3);
      if constexpr (TAP == 4) {                                // the conv block's result, every step (at least four store instructions per iteration, as the waits count)
#pragma unroll
         for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
               if (l1_owns(t, lc)) a.out[(size_t)n * (16 * kFrames) + (4 * q + r) * kFrames + l1_step(t, lc)] = y[t][r];
         continue;
      }
#ifndef VADC_L1R_ABL_NOBLOCK
      l1_block<0>(y, img, vec, lane);
#endif
#ifdef VADC_L1R_PHASE_PROF
      asm volatile("" :: "v"(y[0][0]), "v"(y[1][3]));
#endif
      L1R_PH(4);
      // stride 2: the even steps -- tile 0 lanes 0, 2, .. 12 (steps 0..12), tile 1 lanes 5, 7, .. 15 (steps 14..24).  A lane owns at most one of the
      // two: FOUR stores per iteration (the waits above count them), in one branch that every wave takes
      {
         const bool odd = lc & 1;
         const int step = odd ? 9 + lc : lc;
         float *op = a.out + (size_t)n * (16 * 13) + (4 * q) * 13 + (step >> 1);
         if (odd ? lc >= 5 : lc <= 12) {
#pragma unroll
            for (int r = 0; r < 4; ++r) op[r * 13] = odd ? y[1][r] : y[0][r];
         }
      }
      L1R_PH(5);
      }
   }
}

#ifdef VADC_L1R_PHASE_PROF
extern "C" int vadc_amd_debug_l1r_phases(unsigned long long *out, int reset)
{
   if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_l1r_phase), sizeof(g_l1r_phase)) != hipSuccess) return -1;
   if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_l1r_phase), z, sizeof(z)) != hipSuccess) return -1; }
   return 0;
}
#endif

// max_wgs: workgroups the grid may use (CUs not held by the LSTM chain).  12 waves per workgroup: 3 per SIMD, up to 168 registers each -- the counted
// waits of the input pipeline rely on a register allocation without spills (a scratch access is a vector-memory operation the waits do not count:
// tools/kernel_regs.py checks the build).
void launch_layer1_regs(const L1RegsArgs &a, int max_wgs, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   const int g = std::min(max_wgs, (a.n_chunks + kL1Waves - 1) / kL1Waves);
   hipLaunchKernelGGL((k_layer1_regs<kL1Waves, 0>), dim3(g), dim3(64 * kL1Waves), 0, st, a);
}

// stage taps for the op-level fixtures (vadc_amd_debug_layer1_block): a.y = [n][16][25] (what = 4: [n][129][25] + a.fm), a.out = [n][16][25]
void launch_layer1_regs_tap(int what, const L1RegsArgs &a, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   const int g = std::min(256, (a.n_chunks + 3) / 4);
   if (what == 1)      hipLaunchKernelGGL((k_layer1_regs<4, 1>), dim3(g), dim3(256), 0, st, a);
   else if (what == 2) hipLaunchKernelGGL((k_layer1_regs<4, 2>), dim3(g), dim3(256), 0, st, a);
   else if (what == 3) hipLaunchKernelGGL((k_layer1_regs<4, 3>), dim3(g), dim3(256), 0, st, a);
   else if (what == 4) hipLaunchKernelGGL((k_layer1_regs<kL1Waves, 4>), dim3(std::min(256, (a.n_chunks + kL1Waves - 1) / kL1Waves)), dim3(64 * kL1Waves), 0, st, a);      // the product's shape and input pipeline
   else                hipLaunchKernelGGL((k_layer1_regs<4, 5>), dim3(g), dim3(256), 0, st, a);
}

}  // namespace vadc
