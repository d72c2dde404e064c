// kernels_encoder_fused.hip -- encoder layers 2, 3 and 4 of Silero v3.1 in ONE launch, every activation in registers.
//
// Replaces (reference file:line) for layers 2-4: conv_block conv.c:761-814 (dw :17-113, pw/proj :532-589); transformer_block
// transformer.c:13-234 (tensor_linear tensor.h:675-723, softmax :751-784, layer_norm misc.c:143-210); conv k=1 stride s + BatchNorm
// + ReLU transformer.c:237-295 (conv.c:597-709, misc.c:221-258).  Same arithmetic as k_layer_mfma's split-fp16 form
// (W . X ~= Wl . Xh + Wh . Xl + Wh . Xh on v_mfma_f32_16x16x32_f16, fp32 accumulation), re-mapped:
//
// * A WAVE owns its chunks from the layer-1 output to the LSTM hand-off and never meets another wave: no workgroup barrier, no
//   activation in LDS.  A 16-column MFMA N tile holds ONE chunk at 13 steps (layer 2: lanes 13..15 idle) or TWO chunks at 7 steps
//   (layers 3, 4: columns 0..6 and 8..14, 7 and 15 dead), so a chunk's steps are lanes of one DPP row: the depthwise conv's time
//   neighbours are row shifts riding on the multiply-adds as DPP operands (zero fill at the row ends and the dead columns = the conv's zero padding).
// * An accumulator tile IS the next GEMM's B operand: lane (q, column) holds rows 16 mt + 4 q + r, and the host stores every weight's
//   k in that order (enc_fused_layout.h: enc_sigma), so registers go from MFMA to split (v_cvt_pk_f16_f32) to MFMA.
// * Attention on the matrix cores, per head and tile: S^T = Q^T K (A = Q registers, B = K registers: lane (q, i) gets
//   s[i][j = 4 q + r], so the softmax over j runs over a lane's 4 registers and the 4 lane-quads), V is produced TRANSPOSED by swapping
//   the operands of its projection MFMA (lane = channel, registers = steps), and att = V . a^T lands in the accumulator layout again.
//   Cross-chunk and idle columns are masked in the softmax (k . q^T order and 1/sqrt(hd): transformer.c:104-114; the scale and log2(e)
//   are folded into the Q rows of the weight by the host).
// * Weights live in LDS, copied once per workgroup: the kernel is PERSISTENT (one 8-wave workgroup per CU, waves take batches of
//   chunks round-robin).  192 KB of split weights do not fit 160 KB, so it runs in two phases -- layers 2 + 3 of all its batches with
//   image A (69 KB), then layer 4 with image B (132 KB) -- the layer-3 output of a batch waits in a scratch buffer in its register
//   order (written and read back by the same wave).  A fragments are conflict-free 16-byte LDS reads (hi and lo blocks lane-linear).
#include "common.h"
#include "enc_fused_layout.h"
#include "enc_regs_prims.h"

namespace vadc {

// ---- transformer block + strided 1x1 conv (BatchNorm folded) + ReLU on NT column tiles --------------------------------------------
// acc: y = the conv block's output (residual stream) on entry, z = relu(conv(LN2(...))) on return.
// PAIR: column layout (false: one 13-step chunk per tile, true: two 7-step chunks) -- decides the softmax mask only.
template <typename L, int D, int NT, bool PAIR>
__device__ __forceinline__ void tf_block(f4 (&acc)[NT][D / 16], const char *lf, const float *lv, int lane)
{
   constexpr int MT = D / 16, KB = D / 32, HT = D / 32;      // HT = M tiles per head (hd = D / 2)
   constexpr int NQ = 3 * HT * KB;                           // fragments of one head's Q, K, V rows
   const int q = lane >> 4, lc = lane & 15;
   // fragment index of step i of a head's Q | K | V sequence
   auto qkv_idx = [](int h, int i) { const int part = i / (HT * KB), j = (i / KB) % HT, kb = i % KB; return (part * MT + h * HT + j) * KB + kb; };
   Pre pq = {lds_frag(lf + L::f_qkv, qkv_idx(0, 0), lane), lds_frag(lf + L::f_qkv, qkv_idx(0, 1), lane)};
   f4 bqk[2][HT];                                            // Q and K bias rows of the head in work
#pragma unroll
   for (int j = 0; j < HT; ++j) { bqk[0][j] = lds_vec4(lv, L::v_qkv_b + 16 * j + 4 * q); bqk[1][j] = lds_vec4(lv, L::v_qkv_b + D + 16 * j + 4 * q); }
   Frag yf[NT][KB];
#pragma unroll
   for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) yf[nt][kb] = split8(acc[nt][2 * kb], acc[nt][2 * kb + 1]);
   // softmax mask of this lane's four j = 4 q + r (transformer.c:104-113: a_i = softmax_j(k_i . q_j)): the score MFMA starts from 0 or -1e30,
   // so masked scores never win the max and their exp2 is 0
   f4 smask;
#pragma unroll
   for (int r = 0; r < 4; ++r) smask[r] = (PAIR ? (pair_live(4 * q + r) && pair_chunk(4 * q + r) == pair_chunk(lc)) : (4 * q + r < 13)) ? 0.0f : -1.0e30f;
   f4 att[NT][MT];
   Pre po;                                                   // first fragments of the out projection, requested under the last head's attention
   Vec<MT> b_out;
#pragma unroll
   for (int h = 0; h < 2; ++h) {
      // Q, K (rows = channels) and V^T (the same fragments with the operands swapped: lane = channel, registers = steps) of head h as ONE
      // pipelined fragment sequence.  No V bias: a softmax row sums to 1, so bv passes through the attention unchanged -- the host adds
      // Wo . bv to the out-projection bias.
      f4 QKV[3][NT][HT];
#pragma unroll
      for (int j = 0; j < HT; ++j)
#pragma unroll
         for (int nt = 0; nt < NT; ++nt) { QKV[0][nt][j] = bqk[0][j]; QKV[1][nt][j] = bqk[1][j]; QKV[2][nt][j] = f4{0.0f, 0.0f, 0.0f, 0.0f}; }
      {
         Frag a0 = pq.a0, a1 = pq.a1;
#pragma unroll
         for (int i = 0; i < NQ; ++i) {
            const int part = i / (HT * KB), j = (i / KB) % HT, kb = i % KB;
            Frag a2 = a1;
            if (i + 2 < NQ) a2 = lds_frag(lf + L::f_qkv, qkv_idx(h, i + 2), lane);
            else if (h == 0) a2 = lds_frag(lf + L::f_qkv, qkv_idx(1, i + 2 - NQ), lane);      // the next head's first fragments
            else if (i + 2 == NQ) { po = prefetch<MT * KB>(lf + L::f_out, 0, lane); b_out = load_vec<MT>(lv + L::v_out_b, q); }
            if (part < 2) {
#if VADC_ENC_WLO
#pragma unroll
               for (int nt = 0; nt < NT; ++nt) QKV[part][nt][j] = MFMA16(a0.lo, yf[nt][kb].hi, QKV[part][nt][j]);
#endif
#if VADC_ENC_XLO
#pragma unroll
               for (int nt = 0; nt < NT; ++nt) QKV[part][nt][j] = MFMA16(a0.hi, yf[nt][kb].lo, QKV[part][nt][j]);
#endif
#pragma unroll
               for (int nt = 0; nt < NT; ++nt) QKV[part][nt][j] = MFMA16(a0.hi, yf[nt][kb].hi, QKV[part][nt][j]);
            } else {
#if VADC_ENC_XLO
#pragma unroll
               for (int nt = 0; nt < NT; ++nt) QKV[2][nt][j] = MFMA16(yf[nt][kb].lo, a0.hi, QKV[2][nt][j]);
#endif
#if VADC_ENC_WLO
#pragma unroll
               for (int nt = 0; nt < NT; ++nt) QKV[2][nt][j] = MFMA16(yf[nt][kb].hi, a0.lo, QKV[2][nt][j]);
#endif
#pragma unroll
               for (int nt = 0; nt < NT; ++nt) QKV[2][nt][j] = MFMA16(yf[nt][kb].hi, a0.hi, QKV[2][nt][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1; a1 = a2;
         }
         pq.a0 = a0; pq.a1 = a1;                             // head 0: the next head's first two fragments
      }
      if (h == 0) {
#pragma unroll
         for (int j = 0; j < HT; ++j) { bqk[0][j] = lds_vec4(lv, L::v_qkv_b + 16 * (HT + j) + 4 * q); bqk[1][j] = lds_vec4(lv, L::v_qkv_b + D + 16 * (HT + j) + 4 * q); }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
         // S^T[j][i] = sum_c Q[c][j] K[c][i]: lane (q, i) holds s[i][j = 4 q + r], already scaled by log2(e) / sqrt(hd)
         f4 s = smask;
         if constexpr (HT == 2) {
            const Frag qa = split8(QKV[0][nt][0], QKV[0][nt][1]), kf = split8(QKV[1][nt][0], QKV[1][nt][1]);
            s = MFMA16(qa.lo, kf.hi, s); s = MFMA16(qa.hi, kf.lo, s); s = MFMA16(qa.hi, kf.hi, s);
         } else {
            const Frag4 qa = split4(QKV[0][nt][0]), kf = split4(QKV[1][nt][0]);
            s = MFMA16K16(qa.lo, kf.hi, s); s = MFMA16K16(qa.hi, kf.lo, s); s = MFMA16K16(qa.hi, kf.hi, s);
         }
         float m = max2(max2(s[0], s[1]), max2(s[2], s[3]));
         m = quads_max(m, lane);
         f4 p;
#pragma unroll
         for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] - m);      // tensor.h:751-784
         float sum = (p[0] + p[1]) + (p[2] + p[3]);
         sum = quads_sum(sum, lane);
         const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
         for (int r = 0; r < 4; ++r) p[r] *= inv;
         // att[c][i] = sum_j V[c][j] a[i][j]: A = V^T registers (lane = channel, k = step j = 4 q + r), B = a (lane = column i)
         const Frag4 af = split4(p);
#pragma unroll
         for (int j = 0; j < HT; ++j) {
            const Frag4 vf = split4(QKV[2][nt][j]);
            f4 o = {0.0f, 0.0f, 0.0f, 0.0f};
            o = MFMA16K16(vf.lo, af.hi, o); o = MFMA16K16(vf.hi, af.lo, o); o = MFMA16K16(vf.hi, af.hi, o);
            att[nt][h * HT + j] = o;
         }
      }
   }
   // out projection + residual, LN1, FFN, residual, LN2          transformer.c:202-220
   // every stage requests the NEXT stage's first fragments and vectors before its own MFMAs
   Vec<MT> n1w = load_vec<MT>(lv + L::v_n1_w, q), n1b = load_vec<MT>(lv + L::v_n1_b, q);
   Pre p1;
   Vec<MT> b_l1;
   {
      Frag af[NT][KB];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int kb = 0; kb < KB; ++kb) af[nt][kb] = split8(att[nt][2 * kb], att[nt][2 * kb + 1]);
      f4 p[NT][MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt) p[nt][mt] = b_out.v[mt];
      p1 = prefetch<MT * KB>(lf + L::f_l1, 0, lane);
      b_l1 = load_vec<MT>(lv + L::v_l1_b, q);
      gemm<NT, MT, KB, MT>(p, lf + L::f_out, 0, af, lane, po);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt) acc[nt][mt] += p[nt][mt];
   }
#pragma unroll
   for (int nt = 0; nt < NT; ++nt) layer_norm<MT>(acc[nt], n1w, n1b, lane);
   Vec<MT> n2w = {}, n2b = {}, b_cv;           // (LayerNorm 2's scale and shift are folded into the strided conv)
   Pre pc;
   {
      Frag xf[NT][KB];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int kb = 0; kb < KB; ++kb) xf[nt][kb] = split8(acc[nt][2 * kb], acc[nt][2 * kb + 1]);
      f4 f[NT][MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt) f[nt][mt] = b_l1.v[mt];
      const Pre p2 = prefetch<MT * KB>(lf + L::f_l2, 0, lane);
      const Vec<MT> b_l2 = load_vec<MT>(lv + L::v_l2_b, q);
      gemm<NT, MT, KB, MT>(f, lf + L::f_l1, 0, xf, lane, p1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) f[nt][mt][r] = relu(f[nt][mt][r]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int kb = 0; kb < KB; ++kb) xf[nt][kb] = split8(f[nt][2 * kb], f[nt][2 * kb + 1]);
      f4 g[NT][MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt) g[nt][mt] = b_l2.v[mt];
      pc = prefetch<MT * KB>(lf + L::f_cv, 0, lane);
      b_cv = load_vec<MT>(lv + L::v_cv_b, q);
      gemm<NT, MT, KB, MT>(g, lf + L::f_l2, 0, xf, lane, p2);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt) acc[nt][mt] += g[nt][mt];
   }
#pragma unroll
   for (int nt = 0; nt < NT; ++nt) layer_norm<MT, false>(acc[nt], n2w, n2b, lane);
   // conv k = 1 (+ folded BatchNorm) -> ReLU, every step (the caller keeps the surviving ones)      transformer.c:279-290
   {
      Frag xf[NT][KB];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int kb = 0; kb < KB; ++kb) xf[nt][kb] = split8(acc[nt][2 * kb], acc[nt][2 * kb + 1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b_cv.v[mt];
      gemm<NT, MT, KB, MT>(acc, lf + L::f_cv, 0, xf, lane, pc);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
         for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][mt][r] = relu(acc[nt][mt][r]);
   }
}

// ---- conv block of layers 3 / 4 on register inputs: y = relu(pw(relu(dw(x))) + (proj(x) | x))        conv.c:761-814 ---------------
// x: [NT][2] accumulator-layout tiles of the 32 input channels (idle columns 7, 15 zeroed here); y: [NT][D / 16]
template <typename L, int D, int NT, bool PROJ>
__device__ __forceinline__ void conv_block_regs(f4 (&x)[NT][2], f4 (&y)[NT][D / 16], const char *lf, const float *lv, int lane)
{
   constexpr int MT = D / 16;
   const int q = lane >> 4, lc = lane & 15;
   const bool idle = !pair_live(lc);
   Frag xf[NT][1], df[NT][1];
   f4 k[6][2];
#pragma unroll
   for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) k[t][mt] = lds_vec4(lv, L::v_dw + t * 32 + 16 * mt + 4 * q);
#pragma unroll
   for (int nt = 0; nt < NT; ++nt) {
      f4 d[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
         for (int r = 0; r < 4; ++r) {
            x[nt][mt][r] = idle ? 0.0f : x[nt][mt][r];
            d[mt][r] = dw5<true>(x[nt][mt][r], k[0][mt][r], k[1][mt][r], k[2][mt][r], k[3][mt][r], k[4][mt][r], k[5][mt][r], lc);
         }
      df[nt][0] = split8(d[0], d[1]);
      if (PROJ) xf[nt][0] = split8(x[nt][0], x[nt][1]);
   }
   init_bias<NT, MT>(y, lv + L::v_cb_b, q);
   gemm<NT, MT, 1, MT>(y, lf + L::f_pw, 0, df, lane);
   if (PROJ) gemm<NT, MT, 1, MT>(y, lf + L::f_pj, 0, xf, lane);
#pragma unroll
   for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
         for (int r = 0; r < 4; ++r) {
            if (!PROJ) y[nt][mt][r] += x[nt][mt < 2 ? mt : 0][r];            // identity residual (32 -> 32)
            y[nt][mt][r] = relu(y[nt][mt][r]);
         }
}

// [n][D][7] fp32 <-> the pair layout (lane (q, lc): chunk 2 tile + pair_chunk(lc), step pair_step(lc), channels 16 mt + 4 q + r)
template <int MT>
__device__ __forceinline__ void load_chw7(f4 (&x)[MT], const float *in, int item, int n_chunks, const ItemMap &map, int lane)
{
   const int q = lane >> 4, lc = lane & 15, t = pair_step(lc);
   const bool ok = pair_live(lc) && item + pair_chunk(lc) < n_chunks;
   const float *p = in + (size_t)map(ok ? item + pair_chunk(lc) : 0) * (16 * MT * 7) + (ok ? t : 0);
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) x[mt][r] = ok ? p[(16 * mt + 4 * q + r) * 7] : 0.0f;
}
template <int MT>
__device__ __forceinline__ void store_chw7(const f4 (&x)[MT], float *out, int item, int n_chunks, const ItemMap &map, int lane)
{
   const int q = lane >> 4, lc = lane & 15, t = pair_step(lc);
   if (!(pair_live(lc) && item + pair_chunk(lc) < n_chunks)) return;
   float *p = out + (size_t)map(item + pair_chunk(lc)) * (16 * MT * 7) + t;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[(16 * mt + 4 * q + r) * 7] = x[mt][r];
}

// A wave-uniform pointer, said so: both halves through readfirstlane, so that it lives in an SGPR pair and the per-lane part of an address stays a 32-bit offset
// (global_load / global_store saddr form).  Without it hipcc strength-reduces the batch loop's addresses into per-lane 64-bit induction pointers, and at 168
// registers (12 waves) spilled 17 of them into scratch: 9 scratch loads + 4 stores per batch (round 5's listing).
template <typename P>
__device__ __forceinline__ P *uniform_ptr(P *p)
{
   const unsigned long long v = reinterpret_cast<unsigned long long>(p);
   const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
   return reinterpret_cast<P *>(((unsigned long long)hi << 32) | lo);
}

// NP = column tiles per batch in layers 3 / 4 = pairs of chunks; a batch = 2 NP chunks
// NW = waves per workgroup (one workgroup per CU: the LDS image allows no second one)
// HOT: the product's call -- layers 2, 3 and 4, no stage taps -- with everything else compiled out: the tap stores' and the partial forms' loop-invariant per-lane
// offsets (and an integer division) were hoisted in front of the batch loops and, at 168 registers, spilled: 17 registers, 9 scratch loads + 4 stores per batch in
// round 5's listing of the one instantiation that ships.  HOT = false serves the stage taps and the partial ranges (first / last).
template <int NP, int NW, bool HOT>
__global__ __launch_bounds__(64 * NW) void k_enc_fused(EncFusedArgs a_)
{
   EncFusedArgs a = a_;
   if (HOT) { a.first = 2; a.last = 4; a.tap2 = nullptr; a.tap3 = nullptr; a.tap4 = nullptr; }
   __shared__ __attribute__((aligned(16))) char lds[kEncLdsBytes];
   const int tid = threadIdx.x;
   const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, lc = lane & 15;
   // wave-major slots: the last, partial round of batches is then a few waves on EVERY CU (each with more of its SIMD) instead of full workgroups on some CUs beside
   // idle ones (12,288 batches on 224 x 12 slots: 4.57 rounds) -- as in k_layer1_regs.  Both phases use the same map, so a batch meets its own scratch again
   const int slot = wave * gridDim.x + blockIdx.x, nslots = gridDim.x * NW;
   const int nb = (a.n_chunks + 2 * NP - 1) / (2 * NP);

   // image -> LDS, 8 loads in flight per thread (one load per trip made 6 + 11 dependent round trips to L2 per workgroup)
   auto copy_image = [&](const void *img, int bytes) {
      const uint4 *src = reinterpret_cast<const uint4 *>(img);
      uint4 *dst = reinterpret_cast<uint4 *>(lds);
      const int n = bytes / 16;
      for (int i0 = 0; i0 < n; i0 += 8 * 64 * NW) {
         uint4 v[8];
#pragma unroll
         for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 * NW + tid; v[u] = src[i < n ? i : 0]; }
#pragma unroll
         for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 * NW + tid; if (i < n) dst[i] = v[u]; }
      }
   };

   if (a.first <= 3) {
      copy_image(a.imgA, kEncA_Bytes);
      __syncthreads();
      const char *f2 = lds + kEncA_L2F, *f3 = lds + kEncA_L3F;
      const float *v2 = reinterpret_cast<const float *>(lds + kEncA_V2), *v3 = reinterpret_cast<const float *>(lds + kEncA_V3);
      for (int b = slot; b < nb; b += nslots) {
         const int item0 = b * 2 * NP;
         f4 x3[NP][2];
         if (a.first == 2) {
            // ---- layer 2 on 2 tiles (one 13-step chunk each) per pass; the surviving even steps of a pass make ONE tile of layer 3 ----
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
               f4 y[2][2];
               {
                  // conv block, inputs from memory: lane (q, t) takes channels 8 (q & 1) + e; quads 0, 1 feed relu(dw(x)), quads 2, 3 feed x
                  // into the stacked [pointwise | projection] GEMM (K = 32)
                  Frag bf[2][1];
                  f4 k[6][2];
#pragma unroll
                  for (int t = 0; t < 6; ++t)
#pragma unroll
                     for (int e4 = 0; e4 < 2; ++e4) k[t][e4] = lds_vec4(v2, EncL2::v_dw + t * 16 + 8 * (q & 1) + 4 * e4);
#pragma unroll
                  for (int nt = 0; nt < 2; ++nt) {
                     const int item = item0 + 2 * p + nt;
                     const bool ok = lc < 13 && item < a.n_chunks;
                     const float *xb = uniform_ptr(a.in + (size_t)a.map(item < a.n_chunks ? item : 0) * (16 * 13));      // (the item is the wave's: uniform)
                     const int xo = (8 * (q & 1)) * 13 + lc;
                     f4 xv[2], d[2];
#pragma unroll
                     for (int e = 0; e < 8; ++e) xv[e >> 2][e & 3] = ok ? xb[xo + e * 13] : 0.0f;
#pragma unroll
                     for (int e = 0; e < 8; ++e)
                        d[e >> 2][e & 3] = dw5<false>(xv[e >> 2][e & 3], k[0][e >> 2][e & 3], k[1][e >> 2][e & 3], k[2][e >> 2][e & 3], k[3][e >> 2][e & 3],
                                                      k[4][e >> 2][e & 3], k[5][e >> 2][e & 3], lc);
                     const bool usex = q >= 2;
#pragma unroll
                     for (int e = 0; e < 8; ++e) d[e >> 2][e & 3] = usex ? xv[e >> 2][e & 3] : d[e >> 2][e & 3];
                     bf[nt][0] = split8(d[0], d[1]);
                  }
                  init_bias<2, 2>(y, v2 + EncL2::v_cb_b, q);
                  gemm<2, 2, 1, 2>(y, f2 + EncL2::f_pw, 0, bf, lane);
#pragma unroll
                  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                     for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[nt][mt][r] = relu(y[nt][mt][r]);
               }
               tf_block<EncL2, 32, 2, false>(y, f2, v2, lane);
               // stride 2: step 2 t' of the first / second tile -> column t' / 8 + t' of the pair tile; same quad, same registers
               const int src = 4 * (16 * q + 2 * (pair_live(lc) ? pair_step(lc) : 0));
               const bool second = pair_chunk(lc) == 1, idle = !pair_live(lc);
               f4 xn[2];
#pragma unroll
               for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                     const float y0 = y[0][mt][r], y1 = y[1][mt][r];     // (a bit_cast applied to the vector-element lvalue itself read element 0 for every r)
                     const int v0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, y0));
                     const int v1 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, y1));
                     xn[mt][r] = idle ? 0.0f : __builtin_bit_cast(float, second ? v1 : v0);
                  }
               if (a.tap2) store_chw7<2>(xn, a.tap2, item0 + 2 * p, a.n_chunks, a.map, lane);
               // x3[p] = xn with p a loop variable: select without dynamic register indexing
#pragma unroll
               for (int pp = 0; pp < NP; ++pp)
                  if (pp == p) { x3[pp][0] = xn[0]; x3[pp][1] = xn[1]; }
            }
         } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) load_chw7<2>(x3[p], a.in, item0 + 2 * p, a.n_chunks, a.map, lane);
         }
         if (a.last >= 3) {
            f4 y3[NP][2];
            conv_block_regs<EncL3, 32, NP, false>(x3, y3, f3, v3, lane);
            tf_block<EncL3, 32, NP, true>(y3, f3, v3, lane);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
               if (a.tap3) store_chw7<2>(y3[p], a.tap3, item0 + 2 * p, a.n_chunks, a.map, lane);
               float *sp = uniform_ptr(a.scratch + ((size_t)b * NP + p) * 512);
#pragma unroll
               for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                  for (int r = 0; r < 4; ++r) sp[lane + (4 * mt + r) * 64] = y3[p][mt][r];
            }
         }
      }
      __syncthreads();                                     // every wave is done with image A
   }
   if (a.last < 4) return;
   copy_image(a.imgB, kEncB_Bytes);
   __syncthreads();
   {
      const char *f4_ = lds + kEncB_L4F;
      const float *v4 = reinterpret_cast<const float *>(lds + kEncB_V4);
      for (int b = slot; b < nb; b += nslots) {
         const int item0 = b * 2 * NP;
         f4 x4[NP][2];
         if (a.first <= 3) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
               const float *sp = uniform_ptr(a.scratch + ((size_t)b * NP + p) * 512);
#pragma unroll
               for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                  for (int r = 0; r < 4; ++r) x4[p][mt][r] = __builtin_nontemporal_load(sp + lane + (4 * mt + r) * 64);
            }
         } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) load_chw7<2>(x4[p], a.in, item0 + 2 * p, a.n_chunks, a.map, lane);
         }
         f4 y4[NP][4];
         conv_block_regs<EncL4, 64, NP, true>(x4, y4, f4_, v4, lane);
         tf_block<EncL4, 64, NP, true>(y4, f4_, v4, lane);
#pragma unroll
         for (int p = 0; p < NP; ++p) {
            if (a.tap4) { store_chw7<4>(y4[p], a.tap4, item0 + 2 * p, a.n_chunks, a.map, lane); continue; }
            // split-fp16 LSTM-native tiles (common.h lstm_xh_index): a (chunk, step) row = 64 units x {hi, lo}; this lane owns units 16 mt + 4 q .. + 3
            const int item = item0 + 2 * p + pair_chunk(lc), t = pair_step(lc);
            if (pair_live(lc) && item < a.n_chunks) {
               int st_, ch_;
               a.map.split(item, st_, ch_);
               _Float16 *dst = reinterpret_cast<_Float16 *>(a.out) + lstm_xh_index(st_, ch_, a.map.C, t, 4 * q, 7);
#pragma unroll
               for (int mt = 0; mt < 4; ++mt) {
                  h2 hi[2], lo[2];
                  split2(y4[p][mt][0], y4[p][mt][1], hi[0], lo[0]); split2(y4[p][mt][2], y4[p][mt][3], hi[1], lo[1]);
                  *reinterpret_cast<h4 *>(dst + 16 * mt) = h4{hi[0][0], hi[0][1], hi[1][0], hi[1][1]};
                  *reinterpret_cast<h4 *>(dst + kLstmTile * 64 + 16 * mt) = h4{lo[0][0], lo[0][1], lo[1][0], lo[1][1]};
               }
            }
         }
      }
   }
}

// grid: one workgroup per CU the stream may use (`max_wgs`), never more than there are batches for its waves.
// 12 waves, one pair tile (two chunks) per batch -- 168 registers, three waves per SIMD.  (Round 3 also carried an 8-wave form with two pair tiles per batch -- half
// the LDS weight traffic per chunk, 212 registers: 0.117 / 0.326 ms per 24,576 / 65,536 chunks against this form's 0.110 / 0.329 -- and a 16-wave form that
// spilled; both are gone: option "encoder_batch" is no more.)
void launch_enc_fused(const EncFusedArgs &a, int max_wgs, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   const int nw = 12;
   const int nb = (a.n_chunks + 1) / 2;
   int g = (nb + nw - 1) / nw;
   if (g > max_wgs) g = max_wgs;
   if (g < 1) g = 1;
   const bool hot = a.first == 2 && a.last == 4 && !a.tap2 && !a.tap3 && !a.tap4;
   if (hot) hipLaunchKernelGGL((k_enc_fused<1, 12, true>), dim3(g), dim3(768), 0, st, a);
   else     hipLaunchKernelGGL((k_enc_fused<1, 12, false>), dim3(g), dim3(768), 0, st, a);
}

}  // namespace vadc
