// kernels_lstm.hip -- 2-layer LSTM(64) over the 7 encoder frames of every chunk with per-stream state
// carried on the device, fused with the decoder (ReLU -> 1x1 conv 64->2 -> mean over time -> sigmoid).
//
// Replaces, per chunk (reference file:line):
//   tensor_transpose_last_2d [64,7]->[7,64]      silero_v3.c:115
//   lstm_cell / lstm / lstm_seq / lstm_tensor_minibatched   lstm.c:31-341  (gate order i,f,g,o; W = [4H][x|h])
//   state write-back                              silero_v3.c:178-179
//   decoder                                       silero_v3.c:231-303, maths.h:352-400
//
// The reference runs 7*B steps sequentially with ONE state (its batch = consecutive chunks of one stream).
// Here the unit of parallelism is the STREAM: state h,c is [n_streams][2][64] in HBM, loaded once per call,
// kept on chip while the call's n_chunks chunks of the stream are consumed in order, stored once.
//
// The MFMA mapping: a workgroup = 16 streams.  Per (step, layer) the gate pre-activations are
//   G[256 x 16] = W[256 x 128] . [x ; h][128 x 16]
//   Wave w owns hidden units [16w,16w+16) for ALL four gates, so the i,f,g,o values of one (unit, stream) land in
//   the same lane/register of its four accumulators and the cell update is register-local.  The layer's weights
//   live in registers as MFMA A-fragments for the whole call; [x;h] is the B operand.
// Two kernels, one schedule (the two layers run concurrently one step apart, see k_lstm_wavefront_fused):
//   k_lstm_wavefront_h3    (default, engine variant 6): gate GEMMs on the fp16 matrix pipe with split-fp16 operands, fp32 accuracy
//   k_lstm_wavefront_fused (engine variant 3): fp32 MFMA; what runs when an LSTM weight does not fit fp16's range
// Template parameters TS / DEC select the steps per chunk and the decoder of Silero v3.1 (7, two-output mean-then-sigmoid)
// or v4 (3, one-output sigmoid-then-mean).  (Round 1 also carried a one-wave-per-stream bring-up kernel, a step-sequential MFMA
// kernel and forms with layer 0's input projection hoisted into a GEMM of its own; all measured slower -- DESIGN.md 4.3 -- and removed.)
#include "common.h"
#include <cstdint>

namespace vadc {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Hardware-transcendental forms for the recurrent inner loop (v_exp_f32 / v_rcp_f32, ~1 ulp each): absolute
// error <= ~3e-7 on values in (0,1) / (-1,1), two decades below what the 1e-4 probability bar needs, and the
// recurrence is contractive.
// (__frcp_rn is the correctly rounded reciprocal: a 10-instruction v_div_scale/fmas/fixup sequence; v_rcp_f32 is 1 ulp.)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f); }

// ------------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// LDS-DMA: 64 lanes x 16 B land at (wave-uniform LDS base) + lane*16 without touching VGPRs
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
__device__ __forceinline__ void dma16(const float4 *src_lane, float *lds_wave_base)
{
   __builtin_amdgcn_global_load_lds((gbl_void_t *)src_lane, (lds_void_t *)lds_wave_base, 16, 0, 0);
}

constexpr int kTileS = kLstmTile;    // streams per workgroup (= MFMA N)


// ------------------------------------------------------------------------------------------------
// k_lstm_wavefront_h3: the same recurrence with the gate GEMMs on the fp16 matrix pipe, at fp32 accuracy
// ------------------------------------------------------------------------------------------------
// k_lstm_wavefront is bound by 192 fp32 MFMAs per SIMD and slot at 32 cycles each.  An fp32
// product splits exactly into fp16 pieces, a = ah + al with ah = (half)a, al = (half)(a - ah) (22 significant bits), and
//     W . h  ~=  Wl . hh  +  Wh . hl  +  Wh . hh            (the dropped Wl . hl term is ~2^-22 relative)
// is three v_mfma_f32_16x16x32_f16 (K = 32, 16 cycles each, fp32 accumulation of exact fp16 x fp16 products) instead of
// eight v_mfma_f32_16x16x4_f32 (K = 4 x 8, 32 cycles each): 5.3x fewer matrix cycles.  Measured error of a K = 128 dot product
// against float64: 6.0e-7 for this form, 1.07e-6 for the plain fp32 FMA chain (tools/mfma_f16_probe.hip) -- it is not a
// reduced-precision mode.  Weights are split once per call into registers (same 128 VGPRs as the fp32 fragments); h is
// split by the wave that produces it and lives in LDS as two fp16 tiles [stream][unit] (pitch 80 halves = 10 slots of 16 bytes: a
// B-fragment read -- lane (stream s, quarter q) at slot 10 s + q -- is conflict-free in every lane group of ds_read_b128: the group's eight streams of quarter q
// take the even slots mod 16, its eight of quarter q + 1 the odd ones.  Pitch 72 (rounds 2-5) was 2-way conflicted in every group: 37-39 % of the kernels' LDS
// cycles, tools/lds_frag_probe.hip).  Operand layout (verified on the device): lane l holds A[l & 15][8 (l >> 4) + e], B[8 (l >> 4) + e][l & 15].
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
constexpr int kHPitch = 80;                     // fp16 elements per stream row of an h tile (64 units + 16 pad: see above)

// `xh` is the encoder output in split-fp16 LSTM-native tiles (common.h lstm_xh_index, written by the last encoder stage): layer 0 runs
// the full K = 128 = [x_t ; h0] like layer 1, its x fragments are fetched from global one slot ahead (four 16-byte loads per lane),
// the accumulators start at the bias.
template <int TS, int DEC>
__global__ __launch_bounds__(512, 2) void k_lstm_wavefront_h3(const float *__restrict__ xh,    // split-fp16 X tiles
                                                              LstmWeights w,
                                                              float *__restrict__ hs, float *__restrict__ cs,
                                                              float *__restrict__ probs,
                                                              int n_streams, int n_chunks, int c0, int cg)
{
   // [layer][parity][hi / lo][stream][unit]: the CURRENT h of each layer as split fp16
   __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][2][kTileS * kHPitch];
   __shared__ float pd[2][4][2][kTileS];
   __shared__ __attribute__((aligned(16))) float bl1[256];
   __shared__ __attribute__((aligned(16))) float bl0[256];   // fused gate biases of both layers (accumulator init)

   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = tid >> 6;
   const int L = wave >> 2;
   const int wv = wave & 3;
   const int col = lane & 15;
   const int quad = lane >> 4;
   const int s0 = blockIdx.x * kTileS;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = (s0 + col) < n_streams;

   // A fragments, split: k-block kb covers k in [32 kb, 32 kb + 32) of this layer's K = 128 ([x_t ; h0] / [h0 ; h1]);
   // lane holds row 16 wv + (lane & 15) of gate g, k = 32 kb + 8 quad + e
   constexpr int KB1 = 4;
   h8v ah[4][KB1], al[4][KB1];
#pragma unroll
   for (int g = 0; g < 4; ++g) {
      const float *row = w.w + ((size_t)L * 256 + g * 64 + 16 * wv + (lane & 15)) * 128 + 8 * quad;
#pragma unroll
      for (int kb = 0; kb < KB1; ++kb) {
#pragma unroll
         for (int e = 0; e < 8; ++e) {
            const float v = row[32 * kb + e];
            const _Float16 hi = (_Float16)v;
            ah[g][kb][e] = hi;
            al[g][kb][e] = (_Float16)(v - (float)hi);
         }
      }
   }
   float c[4], dw[2][4], hlast[4];
   if (tid < 256) { bl1[tid] = w.b[256 + tid]; bl0[tid] = w.b[tid]; }
   {
      h4v hi4, lo4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wv + 4 * quad + r;
         dw[0][r] = w.dec_w[u];
         dw[1][r] = w.dec_w[64 + u];
         c[r] = cs[(size_t)s_col * 128 + L * 64 + u];
         hlast[r] = hs[(size_t)s_col * 128 + L * 64 + u];
         hi4[r] = (_Float16)hlast[r];
         lo4[r] = (_Float16)(hlast[r] - (float)hi4[r]);
      }
      *reinterpret_cast<h4v *>(&hb[L][0][0][col * kHPitch + 16 * wv + 4 * quad]) = hi4;
      *reinterpret_cast<h4v *>(&hb[L][0][1][col * kHPitch + 16 * wv + 4 * quad]) = lo4;
   }
   int par0 = 0, par1 = 0;
   float rsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   // this lane's x fragments of the NEXT step: k-blocks 0,1 (units 32 kb + 8 quad .. +7 of stream col), hi and lo
   const _Float16 *xh_lane = reinterpret_cast<const _Float16 *>(xh) + (size_t)blockIdx.x * n_chunks * (TS * 2 * kTileS * 64) + col * 64 + 8 * quad;
   h8v xnh[2], xnl[2];
   if (L == 0) {
      const _Float16 *p = xh_lane + (size_t)c0 * TS * (2 * kTileS * 64);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
         xnh[kb] = *reinterpret_cast<const h8v *>(p + 32 * kb);
         xnl[kb] = *reinterpret_cast<const h8v *>(p + kTileS * 64 + 32 * kb);
      }
   }
   __syncthreads();

   const int total = TS * cg;
   float psum = 0.0f;
#pragma unroll 1      // one copy of the slot body (see k_lstm_layer): a step's bits must not depend on its position in the call
   for (int k = 0; k <= total; ++k) {
      const bool active = (L == 0) ? (k < total) : (k >= 1);
      const int step = (L == 0) ? k : k - 1;
      const int chi = step / TS, t = step - chi * TS;
      if (active) {
         f4v acc[4];
         h8v xch[2], xcl[2];
         {
            const float *bias = (L == 0) ? bl0 : bl1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
               const float4 b4 = *reinterpret_cast<const float4 *>(&bias[g * 64 + 16 * wv + 4 * quad]);
               acc[g][0] = b4.x; acc[g][1] = b4.y; acc[g][2] = b4.z; acc[g][3] = b4.w;
            }
            if (L == 0) {                       // take this step's x fragments, request the next step's
               xch[0] = xnh[0]; xch[1] = xnh[1]; xcl[0] = xnl[0]; xcl[1] = xnl[1];
               if (k + 1 < total) {
                  const _Float16 *p = xh_lane + ((size_t)c0 * TS + (k + 1)) * (2 * kTileS * 64);
#pragma unroll
                  for (int kb = 0; kb < 2; ++kb) {
                     xnh[kb] = *reinterpret_cast<const h8v *>(p + 32 * kb);
                     xnl[kb] = *reinterpret_cast<const h8v *>(p + kTileS * 64 + 32 * kb);
                  }
               }
            }
         }
         // B fragments.  Layer 1: k-blocks 0,1 = h0, 2,3 = h1.  Layer 0: 0,1 = x_t from global, 2,3 = h0.  A lane reads units
         // 32 kb' + 8 quad .. +7 of stream col.
#pragma unroll
         for (int kb = 0; kb < KB1; ++kb) {
            {
               h8v bh, bl;
               if (L == 0 && kb < 2) { bh = xch[kb]; bl = xcl[kb]; }
               else {
                  const bool from_h1 = (L == 1) && (kb >= 2);
                  const _Float16 *base = from_h1 ? hb[1][par1][0] : hb[0][par0][0];
                  const int off = col * kHPitch + 32 * (kb & 1) + 8 * quad;
                  bh = *reinterpret_cast<const h8v *>(base + off);
                  bl = *reinterpret_cast<const h8v *>(base + kTileS * kHPitch + off);
               }
#ifndef VADC_LSTM_ABL_NOMFMA
#pragma unroll
               for (int g = 0; g < 4; ++g) {
                  acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[g][kb], bh, acc[g], 0, 0, 0);
                  acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g][kb], bl, acc[g], 0, 0, 0);
                  acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g][kb], bh, acc[g], 0, 0, 0);
               }
#else
               acc[0][0] += (float)bh[0] + (float)bl[1];     // ablation (timing only): keep the fragment reads alive
#endif
            }
         }
         h4v hi4, lo4;
#pragma unroll
         for (int r = 0; r < 4; ++r) {
#ifndef VADC_LSTM_ABL_NOGATES
            const float ig = fast_sigmoid(acc[0][r]), fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]), og = fast_sigmoid(acc[3][r]);
            c[r] = fmaf(fg, c[r], ig * gg);                    // pinned contraction: k_lstm_layer rounds the same way (bit-identical state)
            const float hn = og * fast_tanh(c[r]);
#else
            c[r] = 0.5f * c[r] + 0.001f * (acc[0][r] + acc[1][r]);   // ablation (timing only)
            const float hn = 0.001f * (acc[2][r] + acc[3][r]) + 0.01f * c[r];
#endif
            hlast[r] = hn;
            hi4[r] = (_Float16)hn;
            lo4[r] = (_Float16)(hn - (float)hi4[r]);
            if (L == 1) rsum[r] = (DEC == 0 ? rsum[r] : 0.0f) + fmaxf(hn, 0.0f);
         }
         _Float16 *hout = (L == 0) ? hb[0][par0 ^ 1][0] : hb[1][par1 ^ 1][0];
         *reinterpret_cast<h4v *>(hout + col * kHPitch + 16 * wv + 4 * quad) = hi4;
         *reinterpret_cast<h4v *>(hout + kTileS * kHPitch + col * kHPitch + 16 * wv + 4 * quad) = lo4;
      }
      const bool chunk_done = (L == 1) && active && (t == TS - 1);
      if (DEC == 1 && L == 1 && active) {
         float d0 = 0.0f;
#pragma unroll
         for (int r = 0; r < 4; ++r) d0 = fmaf(dw[0][r], rsum[r], d0);
         d0 += __shfl_xor(d0, 16);
         d0 += __shfl_xor(d0, 32);
         if (quad == 0) pd[k & 1][wv][0][col] = d0;
      }
      if (DEC == 0 && chunk_done) {
         float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
         for (int r = 0; r < 4; ++r) { d0 = fmaf(dw[0][r], rsum[r], d0); d1 = fmaf(dw[1][r], rsum[r], d1); rsum[r] = 0.0f; }
         d0 += __shfl_xor(d0, 16); d1 += __shfl_xor(d1, 16);
         d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32);
         if (quad == 0) { pd[0][wv][0][col] = d0; pd[0][wv][1][col] = d1; }
      }
      __syncthreads();
      if (k < total) par0 ^= 1;
      if (k >= 1) par1 ^= 1;
      if (DEC == 0 && chunk_done && wv == 0 && lane < 2 * kTileS) {
         const int sc = lane & 15, f = lane >> 4;
         const float m = ((pd[0][0][f][sc] + pd[0][1][f][sc]) + (pd[0][2][f][sc] + pd[0][3][f][sc])) / (float)TS + w.dec_b[f];
         if (s0 + sc < n_streams) probs[((size_t)(s0 + sc) * n_chunks + (c0 + chi)) * 2 + f] = sigmoidf_(m);
      }
      if (DEC == 1 && L == 1 && active && wv == 0 && lane < kTileS) {
         const int q = k & 1;
         psum += sigmoidf_(((pd[q][0][0][lane] + pd[q][1][0][lane]) + (pd[q][2][0][lane] + pd[q][3][0][lane])) + w.dec_b[0]);
         if (t == TS - 1) {
            const float pr = psum / (float)TS;
            if (s0 + lane < n_streams) {
               probs[((size_t)(s0 + lane) * n_chunks + (c0 + chi)) * 2 + 0] = pr;
               probs[((size_t)(s0 + lane) * n_chunks + (c0 + chi)) * 2 + 1] = pr;
            }
            psum = 0.0f;
         }
      }
   }
   if (col_ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wv + 4 * quad + r;
         cs[(size_t)s_col * 128 + L * 64 + u] = c[r];
         hs[(size_t)s_col * 128 + L * 64 + u] = hlast[r];      // fp32 value of the last step (LDS holds only the split form)
      }
   }
}

// Same layer-wavefront schedule with layer 0's input projection done INSIDE the recurrence (x frames staged by
// LDS-DMA).  Used when there are many stream tiles (the LSTM is throughput- not latency-bound and the extra GX round
// trip of the hoisted form does not pay).
template <int TS, int DEC>
__global__ __launch_bounds__(512, 2) void k_lstm_wavefront_fused(const float *__restrict__ enc,   // LSTM-native tiles (common.h)
                                                           LstmWeights w,
                                                           float *__restrict__ hs, float *__restrict__ cs,
                                                           float *__restrict__ probs,
                                                           int n_streams, int n_chunks, int c0, int cg)
{
   constexpr int kXTile = TS * 64 * kTileS;     // floats of one (tile, chunk) block of the encoder output
   __shared__ __attribute__((aligned(16))) float xs[2][kXTile];   // [parity][t][unit][stream]
   __shared__ float hb0[2][64 * kTileS];        // layer-0 hidden state, double buffered: [parity][unit][stream]
   __shared__ float hb1[2][64 * kTileS];        // layer-1 hidden state
   __shared__ float pd[2][4][2][kTileS];        // decoder partial dots per layer-1 wave, double buffered over slots (DEC 1)
   __shared__ __attribute__((aligned(16))) float bl[2][256];

   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = tid >> 6;
   const int L = wave >> 2;                     // layer of this wave
   const int wv = wave & 3;                     // owns hidden units [16 wv, 16 wv + 16) of its layer
   const int col = lane & 15;
   const int quad = lane >> 4;
   const int s0 = blockIdx.x * kTileS;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = (s0 + col) < n_streams;
   const float4 *tile_base = reinterpret_cast<const float4 *>(enc + (size_t)blockIdx.x * n_chunks * kXTile);

   float a[4][32];                              // A fragments of this wave's layer
#pragma unroll
   for (int g = 0; g < 4; ++g) {
      const float *row = w.w + ((size_t)L * 256 + g * 64 + 16 * wv + (lane & 15)) * 128 + quad;
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) a[g][kk] = row[4 * kk];
   }
   float c[4], dw[2][4];
   for (int i = tid; i < 512; i += 512) bl[i >> 8][i & 255] = w.b[i];
   float *hmine = L == 0 ? hb0[0] : hb1[0];
#pragma unroll
   for (int r = 0; r < 4; ++r) {
      const int u = 16 * wv + 4 * quad + r;
      dw[0][r] = w.dec_w[u];
      dw[1][r] = w.dec_w[64 + u];
      c[r] = cs[(size_t)s_col * 128 + L * 64 + u];
      hmine[u * kTileS + col] = hs[(size_t)s_col * 128 + L * 64 + u];
   }
   for (int i = tid; i < kXTile / 4; i += 512)
      reinterpret_cast<float4 *>(xs[0])[i] = (tile_base + (size_t)c0 * (kXTile / 4))[i];
   int par0 = 0, par1 = 0;                      // buffers holding the CURRENT h0 / h1
   float rsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   __syncthreads();

   const int total = TS * cg;
   float psum = 0.0f;                           // DEC 1: sum over the chunk's steps of sigmoid(decoder dot), lanes 0..15 of wave 4
   for (int k = 0; k <= total; ++k) {
      const bool active = (L == 0) ? (k < total) : (k >= 1);
      const int step = (L == 0) ? k : k - 1;    // the step this wave computes in this slot
      const int chi = step / TS, t = step - chi * TS;
      if (L == 0 && active && t == 0 && (chi + 1) < cg) {
         // prefetch the next chunk's frames straight into the other xs buffer (LDS-DMA); its last readers
         // finished before the previous slot's barrier
         const float4 *src = tile_base + (size_t)(c0 + chi + 1) * (kXTile / 4);
#pragma unroll
         for (int i = 0; i < TS; ++i) dma16(src + (tid & 255) + 256 * i, xs[(chi + 1) & 1] + (256 * i + 64 * wv) * 4);
      }
      if (active) {
         const float *xin = (L == 0) ? (xs[chi & 1] + t * 64 * kTileS) : hb0[par0];
         const float *hin = (L == 0) ? hb0[par0] : hb1[par1];
         f4v acc[4];
#pragma unroll
         for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4 *>(&bl[L][g * 64 + 16 * wv + 4 * quad]);
            acc[g][0] = b4.x; acc[g][1] = b4.y; acc[g][2] = b4.z; acc[g][3] = b4.w;
         }
#pragma unroll
         for (int kk = 0; kk < 32; ++kk) {
            const int kr = 4 * kk + quad;
            const float bv = (kk < 16) ? xin[kr * kTileS + col] : hin[(kr - 64) * kTileS + col];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][kk], bv, acc[g], 0, 0, 0);
         }
         float *hout = (L == 0) ? hb0[par0 ^ 1] : hb1[par1 ^ 1];
#pragma unroll
         for (int r = 0; r < 4; ++r) {
            const float ig = fast_sigmoid(acc[0][r]), fg = fast_sigmoid(acc[1][r]);
            const float gg = fast_tanh(acc[2][r]), og = fast_sigmoid(acc[3][r]);
            c[r] = fg * c[r] + ig * gg;
            const float hn = og * fast_tanh(c[r]);
            hout[(16 * wv + 4 * quad + r) * kTileS + col] = hn;
            if (L == 1) rsum[r] = (DEC == 0 ? rsum[r] : 0.0f) + fmaxf(hn, 0.0f);
         }
      }
      const bool chunk_done = (L == 1) && active && (t == TS - 1);
      if (DEC == 1 && L == 1 && active) {
         // per step: partial dot of this wave's 16 units, finished after the barrier by wave 4
         float d0 = 0.0f;
#pragma unroll
         for (int r = 0; r < 4; ++r) d0 = fmaf(dw[0][r], rsum[r], d0);
         d0 += __shfl_xor(d0, 16);
         d0 += __shfl_xor(d0, 32);
         if (quad == 0) pd[k & 1][wv][0][col] = d0;
      }
      if (DEC == 0 && chunk_done) {
         // decoder, once per chunk: mean_t(w . relu(h_t) + b) = (w . sum_t relu(h_t)) / 7 + b   (silero_v3.c:231-303)
         float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
         for (int r = 0; r < 4; ++r) { d0 = fmaf(dw[0][r], rsum[r], d0); d1 = fmaf(dw[1][r], rsum[r], d1); rsum[r] = 0.0f; }
         d0 += __shfl_xor(d0, 16); d1 += __shfl_xor(d1, 16);
         d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32);
         if (quad == 0) { pd[0][wv][0][col] = d0; pd[0][wv][1][col] = d1; }
      }
      __syncthreads();                          // one barrier per slot
      if (k < total) par0 ^= 1;                 // layer 0 wrote a new h0 in this slot
      if (k >= 1) par1 ^= 1;                    // layer 1 wrote a new h1 in this slot
      if (DEC == 0 && chunk_done && wv == 0 && lane < 2 * kTileS) {
         const int sc = lane & 15, f = lane >> 4;
         const float m = ((pd[0][0][f][sc] + pd[0][1][f][sc]) + (pd[0][2][f][sc] + pd[0][3][f][sc])) / (float)TS + w.dec_b[f];
         if (s0 + sc < n_streams) probs[((size_t)(s0 + sc) * n_chunks + (c0 + chi)) * 2 + f] = sigmoidf_(m);
      }
      if (DEC == 1 && L == 1 && active && wv == 0 && lane < kTileS) {
         const int q = k & 1;
         psum += sigmoidf_(((pd[q][0][0][lane] + pd[q][1][0][lane]) + (pd[q][2][0][lane] + pd[q][3][0][lane])) + w.dec_b[0]);
         if (t == TS - 1) {
            const float pr = psum / (float)TS;
            if (s0 + lane < n_streams) {
               probs[((size_t)(s0 + lane) * n_chunks + (c0 + chi)) * 2 + 0] = pr;
               probs[((size_t)(s0 + lane) * n_chunks + (c0 + chi)) * 2 + 1] = pr;
            }
            psum = 0.0f;
         }
      }
   }
   if (col_ok) {
      const float *hfin = (L == 0) ? hb0[par0] : hb1[par1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wv + 4 * quad + r;
         cs[(size_t)s_col * 128 + L * 64 + u] = c[r];
         hs[(size_t)s_col * 128 + L * 64 + u] = hfin[u * kTileS + col];
      }
   }
}

// ------------------------------------------------------------------------------------------------
// k_lstm_layer: ONE layer of the recurrence over all steps of the call; the two layers are two launches on two CU sets, pipelined over calls
// ------------------------------------------------------------------------------------------------
// With few stream tiles the recurrence is a latency chain (TS x chunks dependent slots per call) and k_lstm_wavefront_h3's slot is bound by what
// ONE CU's four matrix pipes have to issue: 8 waves x 48 split-fp16 MFMAs = 1536 pipe cycles per slot, before the gates.  Layer 1 at
// step s needs only h0_s, never anything newer: the layers are a producer / consumer pipeline, not a loop.  The recurrence is therefore also run
// LAYER-MAJOR (mathematically the reference's step-major order, lstm.c:128-144): launch L = 0 runs layer 0 over all steps of the call and writes
// every h0_s -- as the split-fp16 tile the next GEMM wants as its B operand, 4 KB per step, the layout of the encoder hand-off -- to global
// memory; launch L = 1 runs layer 1 + decoder over the same steps with that sequence as ITS input.  Both are the same code: "one LSTM layer over
// an input sequence of split-fp16 tiles".  The engine puts the two launches on two streams with CU sets of their own: layer 1 of call k runs
// beside layer 0 of call k+1 (and beside the front end + encoder of call k+2), so in steady state a call costs max(layer time), not their sum,
// and the hand-off needs nothing but the kernel boundary.
// All 8 waves of a workgroup work on one layer: a wave owns 8 hidden units x 4 gates = 2 MFMA row tiles (rows ordered [unit][gate], so that
// i, f, g, o of a unit are the 4 accumulator registers of one lane), 24 MFMAs and 2 cells per lane and slot -- half of the two-layer kernel's.
// Measured slot: 1.0 us against 1.54 us (tools: bench.py --opt lstm=6 / 7).
// TAPDEC (L = 1 only; stage tap for the reference's decoder fixture, test.c:170): the recurrence is bypassed -- h1 of step k is read from `tap_h`
// [stream][64][TS] fp32 -- and everything behind it (ReLU, the sum over the chunk's steps, the partial dots and their tree, mean, bias, sigmoid:
// silero_v3.c:231-303) runs as in the product; one chunk per stream, the streams' state is neither read nor written.
// TRAIL (round 4): the two layer launches of a call run AT THE SAME TIME, layer 1 a few slots behind layer 0, instead of one after the other -- a call's recurrence
// then takes one chain, not two (the last call of a run: 0.5 ms earlier; a single call's latency halves), with nothing but kernel boundaries BETWEEN calls as
// before.  Workgroup t of BOTH launches runs on XCD t % 8 (workgroups are dealt to the XCDs by their index; the engine verifies it on the device at create and
// the kernels check it again: layer 0 publishes its XCC id, layer 1 traps on a mismatch), so the pair shares ONE L2 and the hand-over needs no cache maintenance
// at all: layer 0's tile stores go through its write-through vector L1 into that L2 as always, and at the end of every BLOCK of 28 slots one lane publishes how
// many tiles are complete -- the counted vmcnt(4) of waves 0-3 in front of a slot's barrier is exactly the guarantee: everything older than the four newest
// vector-memory operations of a wave (store k-1, piece k+4, store k-2, piece k+3) has been acknowledged by the L2, i.e. tiles <= k-3.  Layer 1 fetches tiles
// (LDS-DMA, as always: its vector L1 was invalidated when the kernel started and sees every tile address for the first time) only below the published count:
// in front of a block it asks whether the block's pieces are there, and if not the workgroup polls -- wave 7 reads the word through the SCALAR cache
// (s_dcache_inv + s_load_dword: what the L2 holds now), hands it to the others through the LDS, between two barriers -- bounded: after ~2 s the kernel traps
// instead of hanging the device.  The slot loop's body is exactly the one without the hand-over (a check per slot, even every eighth, cost layer 1 0.11 - 0.16
// us of its 0.76-us slot, mostly through what it did to hipcc's schedule of the body: tools/trail_ablate.sh in the history).  `progress[tile]` = epoch << 20 |
// tiles complete; the epoch (one per launch pair) makes a value left by an earlier call read as zero.
// (Measured on the way: the same hand-over with device-scope accesses -- sc0 sc1 on the tile stores, the tile loads and the count -- is correct on any XCD
// placement but takes every access to the fabric: both chains 0.5 -> 1.0 ms.  An earlier attempt with agent-scope FENCES wrote back and invalidated the whole
// L2 at every hand-over and cost the front end 2 %: DESIGN.md 4.4.)
// FAIL-SAFE (round 6): a layer-1 workgroup whose hand-over failed -- its bounded wait ran out, or its pair sits on another XCD -- no longer leaves wrong state behind:
// it still runs on over whatever the hand-off buffer holds (no second exit from the slot loop, see wait_for), but it writes NEITHER the streams' state NOR its
// "done" word (progress[2 grid + tile] = epoch).  The engine launches the REDO form of layer 1 behind the pair, ordered behind layer 0's end by an event (a kernel
// boundary: plain loads see everything): a workgroup whose tile is marked done leaves at once (one load); one whose tile is not done AND whose layer 0 is complete
// (progress[tile] = the call's total) runs layer 1 for the tile from the untouched pre-call state over the complete h0 sequence -- the call's probabilities and
// state are then what the pair in turn would have produced, bit for bit -- and counts itself in `recov`; a tile whose layer 0 is NOT complete (a workgroup that
// never ran: ticket imbalance) cannot be recovered here: the fatal word (host-mapped, one plain store) is set, the layer-1 state stays as it was before the call,
// and the engine refuses further calls until the streams are reset.
template <int TS, int DEC, int L, bool TAPDEC = false, bool TRAIL = false, bool REDO = false>
__global__ __launch_bounds__(512, 4) void k_lstm_layer(const _Float16 *__restrict__ in_tiles,   // split-fp16 tiles [tile][n_chunks][TS][hi|lo][16][64]: encoder output (L = 0) / h0 sequence (L = 1)
                                                       _Float16 *__restrict__ h0seq,            // L = 0: the h0 sequence, same layout
                                                       LstmWeights w,
                                                       float *__restrict__ hs, float *__restrict__ cs,
                                                       float *__restrict__ probs,               // L = 1
                                                       int n_streams, int n_chunks, int c0, int cg, const float *__restrict__ tap_h = nullptr,
                                                       int *__restrict__ progress = nullptr, int epoch = 0, int *__restrict__ tickets = nullptr, int ticket_base = 0, int *__restrict__ err = nullptr,
                                                       int pgrid = 0, int *__restrict__ recov = nullptr, int wait_limit = 4000000)
{
   // [parity][hi / lo][stream][unit]: the CURRENT h of this layer as split fp16
   __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][kTileS * kHPitch];
   __shared__ __attribute__((aligned(16))) _Float16 xr[4][2][kTileS * 64];   // input ring: [step & 3][hi / lo][16-byte segment 0..7][stream] (LDS-DMA, see below)
   __shared__ float pd[2][8][2][kTileS];
   __shared__ __attribute__((aligned(8))) float dws[2][64];  // layer 1: the decoder's weights (read once per chunk: not worth four registers across the slot loop)
   __shared__ int avail_sync;                                // TRAIL, layer 1: the count a synchronous poll found

   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   int tile = blockIdx.x;
   if (TRAIL) {
      // Which XCD workgroup i of a launch lands on is (start + i) % 8 with a start that differs between queues and over time (tools/xcd_map_probe.hip), so the
      // pair of a tile is made by the XCD itself: the grid is a multiple of 8 (every XCD gets grid / 8 workgroups whatever the start), and a workgroup takes
      // tile xcc + 8 j with j its ticket from ITS XCD's counter of this layer (an L2-local atomic; `ticket_base` = what earlier launches have drawn) -- the
      // same rule in both launches, so tile t runs on XCD t % 8 in both.  Workgroups beyond the last tile leave.
      __shared__ int tile_s;
      if (tid == 0) {
         unsigned xcc;
         asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
         xcc &= 7u;
         const unsigned j = (unsigned)atomicAdd(tickets + L * 8 + (int)xcc, 1) - (unsigned)ticket_base;      // (both counts wrap together)
         tile_s = (int)xcc + 8 * (int)j;
         // an XCD that got more than its share of the grid (another then leaves tiles undone): this workgroup leaves -- no trap (a trap takes the whole HIP context,
         // every engine and stream of the process, with it) and no error word of its own: an undone tile has no "done" word (layer 1) or no final count (layer 0),
         // and the REDO launch behind the pair recovers the former and reports the latter
         if (j >= gridDim.x / 8) tile_s = -1;
      }
      __syncthreads();
      tile = tile_s;
      if (tile < 0 || tile >= (n_streams + kTileS - 1) / kTileS) return;
   }
   if (REDO) {                                             // (workgroup-uniform: every lane loads the same two words)
      if (progress[2 * pgrid + tile] == epoch) return;      // the pair's layer 1 finished this tile: nothing to do (the common case)
      const int raw0 = progress[tile];
      if ((raw0 >> 20) != epoch || (raw0 & 0xFFFFF) != TS * cg) {      // layer 0 never finished this tile: not recoverable from here
         if (tid == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(err), "v"(1) : "memory");
         return;
      }
   }
   const int col = lane & 15;
   const int quad = lane >> 4;
   const int s0 = tile * kTileS;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = (s0 + col) < n_streams;
   const int u0 = 8 * wave + 2 * quad;                    // this lane's cells: units u0 (row tile 0) and u0 + 1 (row tile 1), stream col
#ifdef VADC_LSTM_ABL_STOP          // (study of tools/study/pk_opsel_hazard.hip: the kernel ends after phase N -- results are garbage)
   if (VADC_LSTM_ABL_STOP == 1) return;
#endif

   // A fragments, split.  Row tile m, row 4 q + r of it = gate r of unit 8 wave + 2 q + m; lane holds row (lane & 15), k = 32 kb + 8 quad + e.
   h8v ah[2][4], al[2][4];
   {
      const int q = (lane & 15) >> 2, r = lane & 3;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
         const float *row = w.w + ((size_t)L * 256 + r * 64 + 8 * wave + 2 * q + m) * 128 + 8 * quad;
#pragma unroll
         for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
               const float v = row[32 * kb + e];
               const _Float16 hi = (_Float16)v;
               ah[m][kb][e] = hi;
               al[m][kb][e] = (_Float16)(v - (float)hi);
            }
      }
   }
   typedef _Float16 h2v __attribute__((ext_vector_type(2)));
   float c[2], hlast[2], bias_r[2][4];
#ifdef VADC_LSTM_ABL_STOP
   if (VADC_LSTM_ABL_STOP == 2) { if (ah[0][0][0] == (_Float16)123.0f && al[1][3][7] == (_Float16)77.0f) probs[0] = 1.0f; return; }      // behind the weight fragments
#endif
   if (L == 1 && tid < 128) dws[tid >> 6][tid & 63] = w.dec_w[tid];
#pragma unroll
   for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         bias_r[m][r] = w.b[L * 256 + r * 64 + u0 + m];
      }
   {
      _Float16 hi2[2], lo2[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
         c[m] = TAPDEC ? 0.0f : cs[(size_t)s_col * 128 + L * 64 + u0 + m];
         hlast[m] = TAPDEC ? 0.0f : hs[(size_t)s_col * 128 + L * 64 + u0 + m];
         hi2[m] = (_Float16)hlast[m];
         lo2[m] = (_Float16)(hlast[m] - (float)hi2[m]);
      }
      *reinterpret_cast<h2v *>(&hb[0][0][col * kHPitch + u0]) = (h2v){hi2[0], hi2[1]};
      *reinterpret_cast<h2v *>(&hb[0][1][col * kHPitch + u0]) = (h2v){lo2[0], lo2[1]};
   }
   // input sequence: 4 KB per step; a lane's B fragments of k-block kb: 16 bytes at (row col) + 32 kb + 8 quad of the hi and of the lo tile
   constexpr int kStepHalves = 2 * kTileS * 64;
   _Float16 *out_seq = h0seq + ((size_t)tile * n_chunks + c0) * TS * kStepHalves;
   const int total = TS * cg;
   // The input half of the gate GEMM (k-blocks 0, 1: W_x . x_t) does not depend on the recurrence: it is computed ONE SLOT AHEAD, into the
   // accumulators the next slot starts from (bias + W_x . x_{k+1}), in the shadow of this slot's gate arithmetic -- only the 12 MFMAs of
   // W_h . h_{k-1} per wave are left between the barrier and the gates.  The MFMA sequence on an accumulator is unchanged (bias, k-blocks
   // 0, 1, 2, 3, each al.bh, ah.bl, ah.bh), so the bits are k_lstm_wavefront_h3's.
   // The input tiles travel global -> LDS by LDS-DMA into a ring of four steps, FOUR slots ahead (fetched into registers one slot ahead, the
   // loop-carried register rotation made every slot wait for the load it had just issued: 0.27 of the 0.96 us slot).  Waves 0-3 issue one
   // 1 KB piece each per slot; lane l of piece h of a tile fetches row (l & 15), 16-byte segment 4 h + (l >> 4), so that the LDS image is
   // [segment][stream] and a B-fragment read (stream = lane & 15, segment = 4 kb + quad) is conflict-free.
   const _Float16 *in_seq = in_tiles + ((size_t)tile * n_chunks + c0) * TS * kStepHalves;
   auto issue_x = [&](int step, int slot) {                   // waves 0-3 only
      const int tl = wave >> 1, hf = wave & 1;
      const _Float16 *g = in_seq + (size_t)min(step, total - 1) * kStepHalves + tl * (kTileS * 64) + (lane & 15) * 64 + (4 * hf + (lane >> 4)) * 8;
      const unsigned dst = (unsigned)(uintptr_t)(lds_void_t *)&xr[slot & 3][tl][hf * 512];
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(g) : "memory");
   };
   auto xfrag = [&](int slot, int part, int kb) -> h8v { return *reinterpret_cast<const h8v *>(&xr[slot & 3][part][((4 * kb + quad) * 16 + col) * 8]); };
   // TRAIL, layer 1: tiles of layer 0 that may be fetched
   int avail = 0;
   auto decode = [&](int raw) { return ((raw >> 20) == epoch) ? (raw & 0xFFFFF) : 0; };
   // a plain store through the write-through vector L1 into the XCD's L2, and nothing behind it: a `volatile` store becomes a system-scope flat_store + s_waitcnt
   // vmcnt(0), which put a fabric round trip into wave 7's slot -- every slot (layer 0: 0.74 -> 1.13 us)
   auto publish = [&](int *p, int v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); };
   const int *flag = progress + (TRAIL ? tile : 0);            // wave-uniform
   auto read_flag_now = [&]() -> int {                         // through the scalar cache, invalidated first: what the L2 holds now
      int v;
      asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(flag) : "memory");
      return v;
   };
   // Layer 0 did not come within ~2 s (a time-sliced GPU, a tool that started serialising kernels after the create-time probe, a failed layer-0 launch): the
   // workgroup GIVES UP (avail = 2^30) and stops waiting -- it runs on over whatever the hand-off buffer holds (memory-safe: the engine's own allocation), writes
   // no state and no "done" word, and ends; the REDO launch behind the pair does the tile again.  Never a hang, and no trap (see the ticket above).  No early
   // return either: a second exit from the block loop made hipcc put an s_waitcnt vmcnt(0) at the head of the slot loop.
   auto wait_for = [&](int need) {                             // workgroup-uniform: every wave calls it with the same `need` and the same `avail`
      unsigned spins = 0;
      while (avail < need) {
         if (wave == 7) { const int v = read_flag_now(); if (lane == 0) avail_sync = decode(v); }
         __syncthreads();
         avail = max(avail, avail_sync);
         __syncthreads();
         if (avail < need) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > (unsigned)wait_limit) avail = 1 << 30;
         }
      }
   };
   if (TRAIL && L == 1) {
      wait_for(min(4, TS * cg));                             // the prologue's four pieces
      // the pair must share an L2: layer 0 publishes the XCC it runs on beside its count
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      int other = 0;
      for (int tries = 0; tries < 1000 && (other >> 20) != epoch; ++tries)       // (stored before the first count by the same lane; a few more looks cost nothing)
         asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(other) : "s"(progress + gridDim.x + tile) : "memory");
      if ((other >> 20) != epoch || (unsigned)(other & 0xf) != (xcc & 0xf)) avail = 1 << 30;      // the tiles may then be stale (another L2): give up, the REDO launch does the tile
   }
   if (TRAIL && L == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      if (tid == 448) publish(progress + gridDim.x + tile, (epoch << 20) | (int)(xcc & 0xf));
   }
   if (!TAPDEC && wave < 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue_x(i, i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
   }
   __syncthreads();
#ifdef VADC_LSTM_ABL_STOP
   if (VADC_LSTM_ABL_STOP == 3) { if (ah[0][0][0] == (_Float16)123.0f && c[0] == 5.0f) probs[0] = 1.0f; return; }      // behind the state loads, the LDS h tile and the prologue's four DMA pieces
#endif
   f4v accx[2];
#pragma unroll
   for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) accx[m][r] = bias_r[m][r];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
         const h8v bh = xfrag(0, 0, kb), bl = xfrag(0, 1, kb);
#if defined(VADC_LSTM_ABL_STOP) && defined(VADC_LSTM_ABL_PROLOGUE_NOMFMA)
         accx[m][0] += (float)bh[0] + (float)bl[1] + (float)ah[m][kb][2] + (float)al[m][kb][3];
#else
         accx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m][kb], bh, accx[m], 0, 0, 0);
         accx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], bl, accx[m], 0, 0, 0);
         accx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], bh, accx[m], 0, 0, 0);
#endif
      }
   }
   __syncthreads();                                           // ring slot 0 is free for step 4
#ifdef VADC_LSTM_ABL_STOP
   if (VADC_LSTM_ABL_STOP == 4) { if (accx[0][0] == 123.0f && accx[1][3] == 5.0f) probs[0] = 1.0f; return; }      // behind the twelve MFMAs of the first slot's input half
#endif

   int par = 0;
   float rsum[2] = {0.0f, 0.0f};
   float psum = 0.0f;
   // the decoder's bias of this lane's output (wave 0 finishes the decoder): fetched once -- as a load inside the loop it drained wave 0's LDS-DMA pieces
   // (the compiler's s_waitcnt vmcnt(0) for it) once per chunk
   const float dec_bias = (L == 1) ? w.dec_b[DEC == 0 ? ((lane >> 4) & 1) : 0] : 0.0f;
   // every value loaded before the loop is consumed HERE: hipcc otherwise waits for the carried state's loads at their first use inside the loop -- an
   // s_waitcnt vmcnt(0) in every slot, which also drains the LDS-DMA pieces and the tile store the counted waits leave in flight
   asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(hlast[0]), "+v"(hlast[1]));
   // one copy of the slot body: an unrolled loop may round its copies differently (fma contraction is decided per copy), and then a step's
   // bits would depend on its position in the call
   // TRAIL: the hand-over is looked after between BLOCKS of kTrailBlock slots, outside the slot loop, whose body stays what it is without it (as a check per
   // slot -- even every eighth -- the bookkeeping cost layer 1 0.11 - 0.16 us of its 0.76-us slot, most of it by what it did to the compiler's schedule of the
   // body: tools/trail_ablate.sh).  Layer 0 publishes its count at the end of a block, layer 1 asks in front of one whether its pieces are there.
   constexpr int kTrailBlock = 28;
   int k = 0;
   while (k < total) {
   int kend = total;
   if (TRAIL) kend = min(total, k + kTrailBlock);
   if (TRAIL && L == 1) {
      const int need = min(kend + 3, total - 1) + 1;          // the slots of this block issue the pieces of tiles up to kend + 3
      if (avail < need) wait_for(need);
   }
#pragma unroll 1
   for (; k < kend; ++k) {
      const int chi = k / TS, t = k - chi * TS;
      f4v acc[2];
      acc[0] = accx[0]; acc[1] = accx[1];
#ifndef VADC_LSTM_ABL_NOXLOAD     // (timing-only ablation: the slot without its input fetch)
      if (!TAPDEC && wave < 4) issue_x(k + 4, k);             // step k + 4 into the ring slot step k was read from (one slot ago)
#endif
      // layer 0: write the h tile of slot k - 1 (LDS parity `par`, complete since the last barrier) to the h0 sequence while this slot computes.
      // Issued AFTER the input loads: vector-memory operations retire in order, so a store in front of them would put its completion on the
      // path of the next slot's input (measured: 0.96 instead of 0.70 ms per 672 slots)
      if (L == 0 && k >= 1 && tid < 256) {
         __builtin_amdgcn_sched_barrier(0);
         const int rowi = tid >> 3, seg = tid & 7;           // 32 rows (16 hi + 16 lo) of 64 halves = 8 x 16 bytes
         const u4v tv = *reinterpret_cast<const u4v *>(&hb[par][rowi >> 4][(rowi & 15) * kHPitch + seg * 8]);
         u4v *tp = reinterpret_cast<u4v *>(out_seq + (size_t)(k - 1) * kStepHalves + rowi * 64 + seg * 8);
         *tp = tv;
         __builtin_amdgcn_sched_barrier(0);
      }
#ifndef VADC_LSTM_ABL_NOMFMA
      if constexpr (!TAPDEC) {
#pragma unroll
      for (int kb = 2; kb < 4; ++kb) {                        // the layer's own h (k-blocks 2, 3)
         const int off = col * kHPitch + 32 * (kb & 1) + 8 * quad;
         const h8v bh = *reinterpret_cast<const h8v *>(&hb[par][0][off]);
         const h8v bl = *reinterpret_cast<const h8v *>(&hb[par][1][off]);
#pragma unroll
         for (int m = 0; m < 2; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m][kb], bh, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], bl, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], bh, acc[m], 0, 0, 0);
         }
      }
      // bias + W_x . x_{k+1}: the next slot's starting accumulators (the last slot computes one nobody uses)
      h8v xbh[2], xbl[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) { xbh[kb] = xfrag(k + 1, 0, kb); xbl[kb] = xfrag(k + 1, 1, kb); }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
         for (int r = 0; r < 4; ++r) accx[m][r] = bias_r[m][r];
#pragma unroll
         for (int kb = 0; kb < 2; ++kb) {
            accx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m][kb], xbh[kb], accx[m], 0, 0, 0);
            accx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], xbl[kb], accx[m], 0, 0, 0);
            accx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], xbh[kb], accx[m], 0, 0, 0);
         }
      }
      }
#else
      { const int off = col * kHPitch + 8 * quad; const h8v bh = *reinterpret_cast<const h8v *>(&hb[par][0][off]); acc[0][0] += (float)bh[0] + (float)xfrag(k + 1, 0, 0)[0]; }
#endif
      _Float16 hi2[2], lo2[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#ifndef VADC_LSTM_ABL_NOGATES
         float hn;
         if constexpr (TAPDEC) hn = tap_h[((size_t)s_col * 64 + u0 + m) * TS + k];
         else {
         const float ig = fast_sigmoid(acc[m][0]), fg = fast_sigmoid(acc[m][1]);
         const float gg = fast_tanh(acc[m][2]), og = fast_sigmoid(acc[m][3]);
         c[m] = fmaf(fg, c[m], ig * gg);                       // as k_lstm_wavefront_h3
         hn = og * fast_tanh(c[m]);
         }
#else
         const float hn = (acc[m][0] + acc[m][1]) * 0.01f + (acc[m][2] + acc[m][3]) * 0.01f;
#endif
         hlast[m] = hn;
         hi2[m] = (_Float16)hn;
         lo2[m] = (_Float16)(hn - (float)hi2[m]);
         if (L == 1) rsum[m] = (DEC == 0 ? rsum[m] : 0.0f) + fmaxf(hn, 0.0f);
      }
      *reinterpret_cast<h2v *>(&hb[par ^ 1][0][col * kHPitch + u0]) = (h2v){hi2[0], hi2[1]};
      *reinterpret_cast<h2v *>(&hb[par ^ 1][1][col * kHPitch + u0]) = (h2v){lo2[0], lo2[1]};
      const bool chunk_done = (L == 1) && (t == TS - 1);
      if (L == 1 && (DEC == 1 || chunk_done)) {
         // decoder partial dots of this wave's 8 units: DEC 0 once per chunk (mean_t then sigmoid, two outputs: silero_v3.c:231-303),
         // DEC 1 every step (one output, sigmoid then mean_t: silero_vad.py:200-204,222)
         // The summation tree is k_lstm_wavefront_h3's, so that the probabilities do not depend on which of the two kernels served a call: there a
         // lane chains 4 consecutive units with fma, the 4 quads of a wave add as (q0 + q1) + (q2 + q3), the 4 waves as (w0 + w1) + (w2 + w3).
         // Here those 4 units are this lane's 2 (even quad) followed by the 2 of the lane 16 above (odd quad): the odd quad continues the chain.
         float d0, d1 = 0.0f;
         {
            const float2 w0 = *reinterpret_cast<const float2 *>(&dws[0][u0]), w1 = *reinterpret_cast<const float2 *>(&dws[1][u0]);
            const float dw[2][2] = {{w0.x, w0.y}, {w1.x, w1.y}};
            const float pe0 = fmaf(dw[0][1], rsum[1], fmaf(dw[0][0], rsum[0], 0.0f));
            const float up0 = __shfl_up(pe0, 16);
            d0 = fmaf(dw[0][1], rsum[1], fmaf(dw[0][0], rsum[0], up0));         // odd quads: the 4-unit chain of h3's lane
            d0 += __shfl_xor(d0, 32);                                           // quads 1 + 3 = h3's (q0 + q1) in even waves, (q2 + q3) in odd waves
            if (DEC == 0) {
               const float pe1 = fmaf(dw[1][1], rsum[1], fmaf(dw[1][0], rsum[0], 0.0f));
               const float up1 = __shfl_up(pe1, 16);
               d1 = fmaf(dw[1][1], rsum[1], fmaf(dw[1][0], rsum[0], up1));
               d1 += __shfl_xor(d1, 32);
               rsum[0] = rsum[1] = 0.0f;
            }
         }
         if (quad == 1) { pd[k & 1][wave][0][col] = d0; if (DEC == 0) pd[k & 1][wave][1][col] = d1; }
      }
      // waves 0-3: the piece of step k + 2 (issued two slots ago; read in the next slot) has landed -- newer operations may stay in flight: two
      // pieces (layer 1), plus layer 0's h0-tile stores (counted together, retired in order)
      if (!TAPDEC && wave < 4) { if (L == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      __syncthreads();                                        // one barrier per slot
      par ^= 1;

      if (L == 1) {
         if (DEC == 0 && chunk_done && wave == 0 && lane < 2 * kTileS) {
            const int sc = lane & 15, f = lane >> 4, q = k & 1;
            float m = ((pd[q][0][f][sc] + pd[q][1][f][sc]) + (pd[q][2][f][sc] + pd[q][3][f][sc])) + ((pd[q][4][f][sc] + pd[q][5][f][sc]) + (pd[q][6][f][sc] + pd[q][7][f][sc]));
            m = m / (float)TS + dec_bias;
            if (s0 + sc < n_streams) probs[((size_t)(s0 + sc) * n_chunks + (c0 + chi)) * 2 + f] = sigmoidf_(m);
         }
         if (DEC == 1 && wave == 0 && lane < kTileS) {
            const int q = k & 1;
            const float m = ((pd[q][0][0][lane] + pd[q][1][0][lane]) + (pd[q][2][0][lane] + pd[q][3][0][lane])) + ((pd[q][4][0][lane] + pd[q][5][0][lane]) + (pd[q][6][0][lane] + pd[q][7][0][lane]));
            psum += sigmoidf_(m + dec_bias);
            if (t == TS - 1) {
               const float pr = psum / (float)TS;
               if (s0 + lane < n_streams) {
                  probs[((size_t)(s0 + lane) * n_chunks + (c0 + chi)) * 2 + 0] = pr;
                  probs[((size_t)(s0 + lane) * n_chunks + (c0 + chi)) * 2 + 1] = pr;
               }
               psum = 0.0f;
            }
         }
      }
   }
   // the end of a block (its last slot was k - 1): every wave of 0-3 passed its vmcnt(4) in front of that slot's barrier, so tiles <= k - 4 are complete
   if (TRAIL && L == 0 && tid == 448 && k >= 4) publish(progress + tile, (epoch << 20) | (k - 3));
   }
   if (L == 0 && tid < 256) {                                 // the last step's tile
      const int rowi = tid >> 3, seg = tid & 7;
      const u4v tv = *reinterpret_cast<const u4v *>(&hb[par][rowi >> 4][(rowi & 15) * kHPitch + seg * 8]);
      u4v *tp = reinterpret_cast<u4v *>(out_seq + (size_t)(total - 1) * kStepHalves + rowi * 64 + seg * 8);
      *tp = tv;
   }
   if (TRAIL && L == 0) {                                     // every tile of the call is out: the final count
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 448) publish(progress + tile, (epoch << 20) | total);
   }
   const bool gave_up = TRAIL && L == 1 && avail >= (1 << 30);      // workgroup-uniform
   if (!TAPDEC && col_ok && !gave_up) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
         cs[(size_t)s_col * 128 + L * 64 + u0 + m] = c[m];
         hs[(size_t)s_col * 128 + L * 64 + u0 + m] = hlast[m];
      }
   }
   if (TRAIL && L == 1 && !gave_up && tid == 0) progress[2 * pgrid + tile] = epoch;      // this tile of layer 1 is done (read by the REDO launch, behind a kernel boundary)
   if (REDO && tid == 0) {                                    // counted (device memory), and flagged where the host sees it without a copy (err[1]: host-mapped, a plain store)
      atomicAdd(recov, 1);
      asm volatile("global_store_dword %0, %1, off" :: "v"(err + 1), "v"(1) : "memory");
   }
}

// one layer of the layer-major form (engine variant 7): layer 0 reads the encoder's split-fp16 tiles and writes h0seq, layer 1 reads h0seq
// steps = LSTM steps per chunk: 7 (Silero v3.1); 3 / 2 / 1 (Silero v4 with 1536- / 1024- / 512-sample windows)
// progress != nullptr: the TRAIL form (layer 1 may run beside layer 0 of the same call); epoch in [1, 2048); tickets / ticket_base: see the kernel
template <int TS, int DEC>
static void launch_layer_ts(int layer, const _Float16 *x, _Float16 *h, const LstmWeights &w, float *hs, float *cs, float *probs,
                            int n_streams, int n_chunks, int c0, int cg, hipStream_t st, int *progress, int epoch, int *tickets, int ticket_base, int *err, int *recov, int wait_limit)
{
   const int tiles = (n_streams + kTileS - 1) / kTileS;
   const float *no_tap = nullptr;
   if (progress) {
      const dim3 grid((tiles + 7) / 8 * 8), block(512);          // a multiple of 8: every XCD gets the same number of workgroups
      const int pgrid = (int)grid.x;
      if (layer == 0)      hipLaunchKernelGGL((k_lstm_layer<TS, DEC, 0, false, true>), grid, block, 0, st, x, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, no_tap, progress, epoch, tickets, ticket_base, err, pgrid, recov, wait_limit);
      else if (layer == 1) hipLaunchKernelGGL((k_lstm_layer<TS, DEC, 1, false, true>), grid, block, 0, st, h, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, no_tap, progress, epoch, tickets, ticket_base, err, pgrid, recov, wait_limit);
      else                 hipLaunchKernelGGL((k_lstm_layer<TS, DEC, 1, false, false, true>), dim3(tiles), block, 0, st, h, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, no_tap, progress, epoch, tickets, ticket_base, err, pgrid, recov, wait_limit);      // layer 2 = the REDO form of layer 1, behind the pair
      return;
   }
   const dim3 grid(tiles), block(512);
   if (layer == 0) hipLaunchKernelGGL((k_lstm_layer<TS, DEC, 0>), grid, block, 0, st, x, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, no_tap, progress, epoch, tickets, ticket_base, err, 0, recov, wait_limit);
   else            hipLaunchKernelGGL((k_lstm_layer<TS, DEC, 1>), grid, block, 0, st, h, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, no_tap, progress, epoch, tickets, ticket_base, err, 0, recov, wait_limit);
}
// layer: 0 / 1 = the two layers; 2 (TRAIL pairs only: progress != nullptr) = the REDO form of layer 1 behind the pair (see the kernel)
void launch_lstm_layer(int layer, const float *enc, float *h0seq, const LstmWeights &w, float *hs, float *cs, float *probs,
                       int n_streams, int n_chunks, int c0, int cg, hipStream_t st, int model, int steps, int *progress, int epoch, int *tickets, int ticket_base, int *err, int *recov, int wait_limit)
{
   const _Float16 *x = reinterpret_cast<const _Float16 *>(enc);
   _Float16 *h = reinterpret_cast<_Float16 *>(h0seq);
   if (model == 0)      launch_layer_ts<7, 0>(layer, x, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st, progress, epoch, tickets, ticket_base, err, recov, wait_limit);
   else if (steps == 3) launch_layer_ts<3, 1>(layer, x, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st, progress, epoch, tickets, ticket_base, err, recov, wait_limit);
   else if (steps == 2) launch_layer_ts<2, 1>(layer, x, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st, progress, epoch, tickets, ticket_base, err, recov, wait_limit);
   else                 launch_layer_ts<1, 1>(layer, x, h, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st, progress, epoch, tickets, ticket_base, err, recov, wait_limit);
}

// stage tap: the decoder of k_lstm_layer<.., 1> on n items of [64][steps] (one chunk each), probs [n][2]
void launch_lstm_decoder_tap(const float *tap_h, const LstmWeights &w, float *probs, int n, hipStream_t st, int model, int steps)
{
   const dim3 grid((n + kTileS - 1) / kTileS), block(512);
   const _Float16 *none = nullptr;
   if (model == 0)      hipLaunchKernelGGL((k_lstm_layer<7, 0, 1, true>), grid, block, 0, st, none, (_Float16 *)nullptr, w, (float *)nullptr, (float *)nullptr, probs, n, 1, 0, 1, tap_h, (int *)nullptr, 0, (int *)nullptr, 0);
   else if (steps == 3) hipLaunchKernelGGL((k_lstm_layer<3, 1, 1, true>), grid, block, 0, st, none, (_Float16 *)nullptr, w, (float *)nullptr, (float *)nullptr, probs, n, 1, 0, 1, tap_h, (int *)nullptr, 0, (int *)nullptr, 0);
   else if (steps == 2) hipLaunchKernelGGL((k_lstm_layer<2, 1, 1, true>), grid, block, 0, st, none, (_Float16 *)nullptr, w, (float *)nullptr, (float *)nullptr, probs, n, 1, 0, 1, tap_h, (int *)nullptr, 0, (int *)nullptr, 0);
   else                 hipLaunchKernelGGL((k_lstm_layer<1, 1, 1, true>), grid, block, 0, st, none, (_Float16 *)nullptr, w, (float *)nullptr, (float *)nullptr, probs, n, 1, 0, 1, tap_h, (int *)nullptr, 0, (int *)nullptr, 0);
}

// processes chunks [c0, c0 + cg) of every stream (n_chunks = chunks per stream in the buffers' layout)
// variant: 6 = k_lstm_wavefront_h3 (enc = split-fp16 tiles), 3 = k_lstm_wavefront_fused (enc = fp32 tiles); 7 is two launch_lstm_layer calls
// model: 0 = Silero v3.1 (7 steps per chunk), 1 = Silero v4 (`steps` = 3 / 2 / 1)
template <int TS, int DEC>
static void launch_lstm_ts(int variant, const float *enc, const LstmWeights &w, float *hs, float *cs, float *probs,
                           int n_streams, int n_chunks, int c0, int cg, hipStream_t st)
{
   const dim3 grid((n_streams + kTileS - 1) / kTileS), block(512);
   if (variant == 6) hipLaunchKernelGGL((k_lstm_wavefront_h3<TS, DEC>), grid, block, 0, st, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg);
   else              hipLaunchKernelGGL((k_lstm_wavefront_fused<TS, DEC>), grid, block, 0, st, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg);
}
void launch_lstm(int variant, const float *enc, const LstmWeights &w, float *hs, float *cs, float *probs,
                 int n_streams, int n_chunks, int c0, int cg, hipStream_t st, int model, int steps)
{
   if (model == 0)      launch_lstm_ts<7, 0>(variant, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st);
   else if (steps == 3) launch_lstm_ts<3, 1>(variant, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st);
   else if (steps == 2) launch_lstm_ts<2, 1>(variant, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st);
   else                 launch_lstm_ts<1, 1>(variant, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg, st);
}

}  // namespace vadc
