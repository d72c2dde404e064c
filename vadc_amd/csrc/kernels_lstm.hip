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
// split by the wave that produces it and lives in LDS as two fp16 tiles [stream][unit] (pitch 72: conflict-free 16-byte
// B-fragment reads).  Operand layout (verified on the device): lane l holds A[l & 15][8 (l >> 4) + e], B[8 (l >> 4) + e][l & 15].
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
constexpr int kHPitch = 72;                     // fp16 elements per stream row of an h tile (64 units + 8 pad)

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
            c[r] = fg * c[r] + ig * gg;
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
// k_lstm_pipe: the two layers of a 16-stream tile on TWO CUs, layer 1 trailing layer 0 through a step-by-step hand-off in global memory
// ------------------------------------------------------------------------------------------------
// With few stream tiles the recurrence is a latency chain (TS x chunks dependent slots per call) and k_lstm_wavefront_h3's slot is bound by what
// ONE CU's four matrix pipes have to issue: 8 waves x 48 split-fp16 MFMAs = 1536 pipe cycles per slot, before the gates.  Layer 1 at
// step s needs only h0_s, never anything newer, so the layers are a producer / consumer PIPELINE, not a loop: here a tile is served by two
// workgroups on two CUs.  Workgroup L = 0 runs layer 0 over all steps of the call and publishes every h0_s (as the split-fp16 tile the next GEMM
// wants as its B operand: 4 KB per step, the layout of the encoder hand-off) to global memory; workgroup L = 1 runs layer 1 + decoder over the
// same steps, reading h0_s as ITS input sequence.  Both are the same code: "one LSTM layer over an input sequence of split-fp16 tiles".  All 8
// waves of a workgroup work on one layer: a wave owns 8 hidden units x 4 gates = 2 MFMA row tiles (rows ordered [unit][gate], so that i, f, g, o
// of a unit are the 4 accumulator registers of one lane), 24 MFMAs and 2 cells per lane and slot -- half of the single-CU kernel's -- and the
// recurrent h stays in the workgroup's LDS exactly as before (one barrier per slot).
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): the producer copies the h tile of slot k from LDS to global with 16-byte `sc1`
// (write-through) stores during slot k+1, every storing wave drains them (s_waitcnt vmcnt(0)) and bumps a counter in LDS, the wave whose bump is
// the last publishes `flag[tile] = k + 1` with an agent-scope store.  The consumer reads the flag with agent-scope (sc1) loads, one slot ahead of
// need and off its critical path (the value is consumed at the end of the slot), and every load of handed-off bytes is an `sc1` load.  It only
// spins when it has caught up with the producer; the spin is bounded (kPipeSpinLimit polls ~ seconds): on expiry it raises `*error` and runs on
// with whatever it reads, so a lost producer can never hang the GPU.  The host zeroes the flags on the launch's stream before every launch.
// The engine uses this kernel while 2 x tiles workgroups fit the LSTM's CU partition one per CU (<= 1024 streams); beyond that the chain is
// throughput work and k_lstm_wavefront_h3 keeps both layers on one CU.
constexpr unsigned kPipeSpinLimit = 1u << 22;

typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(void *gptr, u4v v)
{
   asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(gptr), "v"(v) : "memory");
}
__device__ __forceinline__ u4v load16_sc1(const void *gptr)
{
   u4v v;
   asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(gptr) : "memory");
   return v;
}

template <int TS, int DEC>
__global__ __launch_bounds__(512, 1) void k_lstm_pipe(const float *__restrict__ xh,    // split-fp16 X tiles (encoder output)
                                                      _Float16 *__restrict__ h0seq,    // [tile][n_chunks][TS][hi|lo][16][64] halves: layer 0 -> layer 1
                                                      unsigned *__restrict__ flags,    // [tiles]: steps of h0seq published (zeroed by the host before the launch)
                                                      unsigned *__restrict__ error,    // raised when a consumer's bounded spin expires
                                                      LstmWeights w,
                                                      float *__restrict__ hs, float *__restrict__ cs,
                                                      float *__restrict__ probs,
                                                      int n_streams, int n_chunks, int c0, int cg, int tiles)
{
   // [parity][hi / lo][stream][unit]: the CURRENT h of this workgroup's layer as split fp16
   __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][kTileS * kHPitch];
   __shared__ float pd[2][8][2][kTileS];
   __shared__ unsigned drained_s, known_s;

   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int L = blockIdx.x >= tiles ? 1 : 0;            // layer-0 workgroups first: dispatched first
   const int tile = blockIdx.x - L * tiles;
   const int col = lane & 15;
   const int quad = lane >> 4;
   const int s0 = tile * kTileS;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = (s0 + col) < n_streams;
   const int u0 = 8 * wave + 2 * quad;                    // this lane's cells: units u0 (row tile 0) and u0 + 1 (row tile 1), stream col

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
   if (tid == 0) { drained_s = 0; known_s = 0; }
   float c[2], hlast[2], dw[2][2], bias_r[2][4];
#pragma unroll
   for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_r[m][r] = w.b[L * 256 + r * 64 + u0 + m];
   {
      _Float16 hi2[2], lo2[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
         dw[0][m] = w.dec_w[u0 + m];
         dw[1][m] = w.dec_w[64 + u0 + m];
         c[m] = cs[(size_t)s_col * 128 + L * 64 + u0 + m];
         hlast[m] = hs[(size_t)s_col * 128 + L * 64 + u0 + m];
         hi2[m] = (_Float16)hlast[m];
         lo2[m] = (_Float16)(hlast[m] - (float)hi2[m]);
      }
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
      *reinterpret_cast<h2v *>(&hb[0][0][col * kHPitch + u0]) = (h2v){hi2[0], hi2[1]};
      *reinterpret_cast<h2v *>(&hb[0][1][col * kHPitch + u0]) = (h2v){lo2[0], lo2[1]};
   }
   // input sequence of this layer: split-fp16 tiles, 4 KB per step; a lane's B fragments of k-block kb: 16 bytes at (row col) + 32 kb + 8 quad
   constexpr int kStepHalves = 2 * kTileS * 64;
   const _Float16 *in_seq = (L == 0 ? reinterpret_cast<const _Float16 *>(xh) : h0seq) + ((size_t)tile * n_chunks + c0) * TS * kStepHalves;
   const _Float16 *in_lane = in_seq + col * 64 + 8 * quad;
   _Float16 *out_seq = h0seq + ((size_t)tile * n_chunks + c0) * TS * kStepHalves;
   const int total = TS * cg;
   __syncthreads();

   // consumer: steps known to be published.  `need` steps must be visible before the fragments of step need - 1 are requested.
   unsigned known = (L == 0) ? 0xffffffffu : 0u;
   unsigned spins = 0;
   auto wait_for = [&](unsigned need) {
      while (known < need) {                                 // uniform over the workgroup: `known` comes from LDS behind a barrier
         if (tid == 0) {
            unsigned v = __hip_atomic_load(flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v < need) {
               __builtin_amdgcn_s_sleep(2);
               if (++spins > kPipeSpinLimit) { atomicOr(error, 1u); v = 0xffffffffu; }        // give up loudly, never hang
            }
            known_s = v;
         }
         __syncthreads();
         known = known_s;
         __syncthreads();
      }
   };
   auto load_frags = [&](int step, h8v (&fh)[2], h8v (&fl)[2]) {
      const _Float16 *p = in_lane + (size_t)step * kStepHalves;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
         if (L == 0) {
            fh[kb] = *reinterpret_cast<const h8v *>(p + 32 * kb);
            fl[kb] = *reinterpret_cast<const h8v *>(p + kTileS * 64 + 32 * kb);
         } else {                                             // handed-off bytes: sc1 loads only
            const u4v a = load16_sc1(p + 32 * kb), b = load16_sc1(p + kTileS * 64 + 32 * kb);
            fh[kb] = __builtin_bit_cast(h8v, a);
            fl[kb] = __builtin_bit_cast(h8v, b);
         }
      }
   };
   h8v xnh[2], xnl[2];
   // the consumer's fragment loads are inline asm (sc1): hipcc does not count them, so the wait is explicit and TIED to the registers
   // ("+v"), which keeps every use of them behind it
#define VADC_PIPE_WAIT_FRAGS() asm volatile("s_waitcnt vmcnt(0)" : "+v"(xnh[0]), "+v"(xnh[1]), "+v"(xnl[0]), "+v"(xnl[1])::"memory")
   wait_for(1);
   load_frags(0, xnh, xnl);
   if (L == 1) VADC_PIPE_WAIT_FRAGS();

   int par = 0;
   float rsum[2] = {0.0f, 0.0f};
   float psum = 0.0f;
   for (int k = 0; k < total; ++k) {
      const int chi = k / TS, t = k - chi * TS;
      // consumer: look at the flag early, use the value at the end of the slot (off the critical path)
      unsigned flag_seen = 0;
      if (L == 1 && tid == 0) flag_seen = __hip_atomic_load(flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // producer: publish the h tile of slot k - 1 (LDS parity `par`, complete since the last barrier) while this slot computes
      if (L == 0 && k >= 1 && tid < 256) {
         const int rowi = tid >> 3, seg = tid & 7;           // 32 rows (16 hi + 16 lo) of 64 halves = 8 x 16 bytes
         const u4v v = *reinterpret_cast<const u4v *>(&hb[par][rowi >> 4][(rowi & 15) * kHPitch + seg * 8]);
         store16_sc1(out_seq + (size_t)(k - 1) * kStepHalves + rowi * 64 + seg * 8, v);
      }
      f4v acc[2];
      h8v xch[2], xcl[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
         for (int r = 0; r < 4; ++r) acc[m][r] = bias_r[m][r];
      xch[0] = xnh[0]; xch[1] = xnh[1]; xcl[0] = xnl[0]; xcl[1] = xnl[1];
      if (k + 1 < total && known >= (unsigned)(k + 2)) load_frags(k + 1, xnh, xnl);       // next step's input, one slot ahead (if already published)
      const bool prefetched = (k + 1 < total) && known >= (unsigned)(k + 2);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         h8v bh, bl;
         if (kb < 2) { bh = xch[kb]; bl = xcl[kb]; }
         else {
            const int off = col * kHPitch + 32 * (kb & 1) + 8 * quad;
            bh = *reinterpret_cast<const h8v *>(&hb[par][0][off]);
            bl = *reinterpret_cast<const h8v *>(&hb[par][1][off]);
         }
#pragma unroll
         for (int m = 0; m < 2; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m][kb], bh, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], bl, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m][kb], bh, acc[m], 0, 0, 0);
         }
      }
      _Float16 hi2[2], lo2[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
         const float ig = fast_sigmoid(acc[m][0]), fg = fast_sigmoid(acc[m][1]);
         const float gg = fast_tanh(acc[m][2]), og = fast_sigmoid(acc[m][3]);
         c[m] = fg * c[m] + ig * gg;
         const float hn = og * fast_tanh(c[m]);
         hlast[m] = hn;
         hi2[m] = (_Float16)hn;
         lo2[m] = (_Float16)(hn - (float)hi2[m]);
         if (L == 1) rsum[m] = (DEC == 0 ? rsum[m] : 0.0f) + fmaxf(hn, 0.0f);
      }
      {
         typedef _Float16 h2v __attribute__((ext_vector_type(2)));
         *reinterpret_cast<h2v *>(&hb[par ^ 1][0][col * kHPitch + u0]) = (h2v){hi2[0], hi2[1]};
         *reinterpret_cast<h2v *>(&hb[par ^ 1][1][col * kHPitch + u0]) = (h2v){lo2[0], lo2[1]};
      }
      const bool chunk_done = (L == 1) && (t == TS - 1);
      if (L == 1 && (DEC == 1 || chunk_done)) {
         // decoder partial dots of this wave's 8 units: DEC 0 once per chunk (mean_t then sigmoid, two outputs), DEC 1 every step (one output)
         float d0 = fmaf(dw[0][1], rsum[1], dw[0][0] * rsum[0]), d1 = 0.0f;
         if (DEC == 0) { d1 = fmaf(dw[1][1], rsum[1], dw[1][0] * rsum[0]); rsum[0] = rsum[1] = 0.0f; }
         d0 += __shfl_xor(d0, 16); d0 += __shfl_xor(d0, 32);
         if (DEC == 0) { d1 += __shfl_xor(d1, 16); d1 += __shfl_xor(d1, 32); }
         if (quad == 0) { pd[k & 1][wave][0][col] = d0; if (DEC == 0) pd[k & 1][wave][1][col] = d1; }
      }
      // producer: the copy of slot k - 1's tile is drained -> count; the last of the 4 storing waves publishes step k - 1
      if (L == 0 && k >= 1 && tid < 256) {
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         unsigned old = 0;
         if (lane == 0) old = atomicAdd(&drained_s, 1u);
         old = __builtin_amdgcn_readfirstlane(old);
         if (lane == 0 && old == 4u * (unsigned)(k - 1) + 3u) __hip_atomic_store(flags + tile, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (L == 1) {
         if (prefetched) VADC_PIPE_WAIT_FRAGS();
         if (tid == 0) known_s = flag_seen;
      }
      __syncthreads();
      par ^= 1;
      if (L == 1) {
         known = max(known, known_s);
         if (k + 1 < total && !prefetched) {                 // caught up with the producer: wait (bounded), then fetch the next step's input now
            wait_for((unsigned)(k + 2));
            load_frags(k + 1, xnh, xnl);
            VADC_PIPE_WAIT_FRAGS();
         }
         if (DEC == 0 && chunk_done && wave == 0 && lane < 2 * kTileS) {
            const int sc = lane & 15, f = lane >> 4, q = k & 1;
            float m = 0.0f;
#pragma unroll
            for (int wv = 0; wv < 8; wv += 2) m += pd[q][wv][f][sc] + pd[q][wv + 1][f][sc];
            m = m / (float)TS + w.dec_b[f];
            if (s0 + sc < n_streams) probs[((size_t)(s0 + sc) * n_chunks + (c0 + chi)) * 2 + f] = sigmoidf_(m);
         }
         if (DEC == 1 && wave == 0 && lane < kTileS) {
            const int q = k & 1;
            float m = 0.0f;
#pragma unroll
            for (int wv = 0; wv < 8; wv += 2) m += pd[q][wv][0][lane] + pd[q][wv + 1][0][lane];
            psum += sigmoidf_(m + w.dec_b[0]);
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
   if (L == 0) {                                              // the last step's tile
      if (tid < 256) {
         const int rowi = tid >> 3, seg = tid & 7;
         const u4v v = *reinterpret_cast<const u4v *>(&hb[par][rowi >> 4][(rowi & 15) * kHPitch + seg * 8]);
         store16_sc1(out_seq + (size_t)(total - 1) * kStepHalves + rowi * 64 + seg * 8, v);
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         unsigned old = 0;
         if (lane == 0) old = atomicAdd(&drained_s, 1u);
         old = __builtin_amdgcn_readfirstlane(old);
         if (lane == 0 && old == 4u * (unsigned)(total - 1) + 3u) __hip_atomic_store(flags + tile, (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
   }
#undef VADC_PIPE_WAIT_FRAGS
   if (col_ok) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
         cs[(size_t)s_col * 128 + L * 64 + u0 + m] = c[m];
         hs[(size_t)s_col * 128 + L * 64 + u0 + m] = hlast[m];
      }
   }
}

// processes chunks [c0, c0 + cg) of every stream (n_chunks = chunks per stream in the buffers' layout)
// variant: 6 = k_lstm_wavefront_h3 (enc = split-fp16 tiles), 7 = k_lstm_pipe (the same tiles; two workgroups per stream tile, h0seq / flags /
// error are its hand-off buffers), 3 = k_lstm_wavefront_fused (enc = fp32 tiles)
// model: 0 = Silero v3.1 (7 steps per chunk), 1 = Silero v4 (3 steps)
void launch_lstm(int variant, const float *enc, const LstmWeights &w, float *hs, float *cs, float *probs,
                 int n_streams, int n_chunks, int c0, int cg, hipStream_t st, int model, float *h0seq, unsigned *flags, unsigned *error)
{
   const dim3 grid((n_streams + kTileS - 1) / kTileS), block(512);
   if (variant == 7) {
      const int tiles = (int)grid.x;
      (void)hipMemsetAsync(flags, 0, sizeof(unsigned) * tiles, st);
      if (model == 1) hipLaunchKernelGGL((k_lstm_pipe<3, 1>), dim3(2 * tiles), block, 0, st, enc, reinterpret_cast<_Float16 *>(h0seq), flags, error, w, hs, cs, probs, n_streams, n_chunks, c0, cg, tiles);
      else            hipLaunchKernelGGL((k_lstm_pipe<7, 0>), dim3(2 * tiles), block, 0, st, enc, reinterpret_cast<_Float16 *>(h0seq), flags, error, w, hs, cs, probs, n_streams, n_chunks, c0, cg, tiles);
   } else if (variant == 6) {
      if (model == 1) hipLaunchKernelGGL((k_lstm_wavefront_h3<3, 1>), grid, block, 0, st, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg);
      else            hipLaunchKernelGGL((k_lstm_wavefront_h3<7, 0>), grid, block, 0, st, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg);
   } else {
      if (model == 1) hipLaunchKernelGGL((k_lstm_wavefront_fused<3, 1>), grid, block, 0, st, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg);
      else            hipLaunchKernelGGL((k_lstm_wavefront_fused<7, 0>), grid, block, 0, st, enc, w, hs, cs, probs, n_streams, n_chunks, c0, cg);
   }
}

}  // namespace vadc
