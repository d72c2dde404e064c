// kernels_encoder_fused_v4.hip -- Silero v4 (every window, both sample-rate branches): encoder stages 2, 3 and 4 in ONE launch, every activation in registers.
//
// Reference arithmetic: silero_vad.py:191-236 (Silero_V4: four ConvBlock -> strided 1x1 conv (BatchNorm folded by the exporter) -> ReLU stages, no transformer
// blocks; ConvBlock = relu(pw(relu(dw(x))) + (proj(x) | x)), silero_vad.py:69-106; the reference itself reaches the graph through onnxruntime,
// onnx_helpers.c:532-549).  Stage shapes after the first stage at the default window: [16][12] -> [32][6] -> [32][3] -> [64][3] (strides 2, 2, 1), then the LSTM's input tiles;
// in general [16][t1] -> [32][t2] -> [32][ts] -> [64][ts] with t1 <= 12 valid steps (EncV4Args: the step counts are kernel arguments since round 6 -- the other windows ran three
// launches of k_layer_mfma: 0.12 - 0.28 ms per 65,536 chunks against 0.03 - 0.07 here), t2 = (t1 + 1) / 2, ts = (t2 + 1) / 2 or, in the 8 kHz branch (third conv of stride 1), t2.
//
// Round 4 ran these stages as three launches of k_layer_mfma (fp32 MFMA, activations in LDS, a workgroup barrier between every step): 0.069 + 0.048 + 0.066 ms per
// 65,536 chunks for 55 K MAC per chunk -- launch and barrier bound (profiles/r05/bench_v4_4096x16_pmc_compute.json: the matrix pipe < 10 % busy, waves waiting 43 %).
// This kernel is k_enc_fused's scheme (kernels_encoder_fused.hip) without its transformer blocks:
// * a WAVE owns a batch of two chunks from the first stage's output to the LSTM hand-off and never meets another wave: no barrier, no activation in LDS;
// * an accumulator tile IS the next GEMM's B operand (weights stored in the accumulator's row order, enc_sigma); GEMMs are split-fp16
//   (W . X ~= Wl . Xh + Wh . Xl + Wh . Xh on v_mfma_f32_16x16x32_f16, fp32 accumulation: 22-bit operands);
// * a chunk's steps are lanes of one DPP row: stage 2 one chunk per 16-column tile (12 steps, columns 12..15 zero), stages 3 / 4 both chunks in one tile at
//   columns 0.. and 8.. (6, then 3 steps: at least two dead columns between and behind them, so the depthwise conv's zero padding comes with the row shifts
//   and no tap reaches the other chunk); the stride-2 hand-overs are lane gathers inside a lane quad (ds_bpermute);
// * all weights (52 KB of split fragments and vectors) sit in LDS, copied once per workgroup; the kernel is persistent.
// A chunk's bits do not depend on its place in a batch or in the launch (MFMA columns are independent, the row shifts see the same zeros on both sides).
#include "common.h"
#include "enc_fused_layout.h"
#include "enc_regs_prims.h"

namespace vadc {

__device__ __forceinline__ int v4p_chunk(int lc) { return lc >> 3; }
__device__ __forceinline__ int v4p_step(int lc) { return lc & 7; }

// conv block on register inputs (32 input channels: two accumulator-layout tiles of the pair tile): y = relu(pw(relu(dw(x))) + (proj(x) | x))
template <typename L, int D, bool PROJ>
__device__ __forceinline__ void v4_conv_block(const f4 (&x)[2], f4 (&y)[D / 16], const char *lf, const float *lv, int lane)
{
   constexpr int MT = D / 16;
   const int q = lane >> 4, lc = lane & 15;
   f4 k[6][2];
#pragma unroll
   for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) k[t][mt] = lds_vec4(lv, L::v_dw + t * 32 + 16 * mt + 4 * q);
   f4 d[2];
#pragma unroll
   for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[mt][r] = dw5<false>(x[mt][r], k[0][mt][r], k[1][mt][r], k[2][mt][r], k[3][mt][r], k[4][mt][r], k[5][mt][r], lc);
   Frag df[1][1], xf[1][1];
   df[0][0] = split8(d[0], d[1]);
   if (PROJ) xf[0][0] = split8(x[0], x[1]);
   f4 acc[1][MT];
   init_bias<1, MT>(acc, lv + L::v_cb_b, q);
   gemm<1, MT, 1, MT>(acc, lf + L::f_pw, 0, df, lane);
   if (PROJ) gemm<1, MT, 1, MT>(acc, lf + L::f_pj, 0, xf, lane);
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         float v = acc[0][mt][r];
         if (!PROJ) v += x[mt < 2 ? mt : 0][r];                                // identity residual (32 -> 32)
         y[mt][r] = relu(v);
      }
}

// strided 1x1 conv (BatchNorm folded) + ReLU on every column: z = relu(W . y + b)
template <typename L, int D, int NT>
__device__ __forceinline__ void v4_conv1x1(const f4 (&y)[NT][D / 16], f4 (&z)[NT][D / 16], const char *lf, const float *lv, int lane)
{
   constexpr int MT = D / 16, KB = D / 32;
   const int q = lane >> 4;
   Frag yf[NT][KB];
#pragma unroll
   for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) yf[nt][kb] = split8(y[nt][2 * kb], y[nt][2 * kb + 1]);
   init_bias<NT, MT>(z, lv + L::v_cv_b, q);
   gemm<NT, MT, KB, MT>(z, lf + L::f_cv, 0, yf, lane);
#pragma unroll
   for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
         for (int r = 0; r < 4; ++r) z[nt][mt][r] = relu(z[nt][mt][r]);
}

// NW = waves per workgroup; a batch = two chunks
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_enc_fused_v4(EncV4Args a)
{
   __shared__ __attribute__((aligned(16))) char lds[kEncV4Bytes];
   const int tid = threadIdx.x;
   const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, lc = lane & 15;
   {
      const uint4 *src = reinterpret_cast<const uint4 *>(a.img);
      uint4 *dst = reinterpret_cast<uint4 *>(lds);
      for (int i = tid; i < kEncV4Bytes / 16; i += 64 * NW) dst[i] = src[i];
   }
   __syncthreads();
   const char *f2 = lds + kEncV4_L2F, *f3 = lds + kEncV4_L3F, *f4_ = lds + kEncV4_L4F;
   const float *v2 = reinterpret_cast<const float *>(lds + kEncV4_V2), *v3 = reinterpret_cast<const float *>(lds + kEncV4_V3),
               *v4 = reinterpret_cast<const float *>(lds + kEncV4_V4);
   const int nb = (a.n_chunks + 1) / 2;
   for (int b = wave * gridDim.x + blockIdx.x; b < nb; b += gridDim.x * NW) {      // wave-major slots: a partial last round is a few waves on every CU (k_enc_fused)
      const int item0 = 2 * b;
      // ---- stage 2 on two tiles (one 12-step chunk each): conv block with inputs from memory -- lane (q, t) takes channels 8 (q & 1) + e; quads 0, 1 feed
      //      relu(dw(x)), quads 2, 3 feed x into the stacked [pointwise | projection] GEMM (K = 32)
      f4 y2[2][2];
      {
         Frag bf[2][1];
         f4 k[6][2];
#pragma unroll
         for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int e4 = 0; e4 < 2; ++e4) k[t][e4] = lds_vec4(v2, EncV4L2::v_dw + t * 16 + 8 * (q & 1) + 4 * e4);
#pragma unroll
         for (int nt = 0; nt < 2; ++nt) {
            const int item = item0 + nt;
            const bool ok = lc < a.t1 && item < a.n_chunks;                  // (steps past the valid ones are zeros: the depthwise conv's padding)
            const float *xp = a.in + (size_t)a.map(item < a.n_chunks ? item : 0) * (16 * a.t1_pitch) + (8 * (q & 1)) * a.t1_pitch + (lc < a.t1 ? lc : 0);
            f4 xv[2], d[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e >> 2][e & 3] = ok ? xp[e * a.t1_pitch] : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
               d[e >> 2][e & 3] = dw5<false>(xv[e >> 2][e & 3], k[0][e >> 2][e & 3], k[1][e >> 2][e & 3], k[2][e >> 2][e & 3], k[3][e >> 2][e & 3],
                                             k[4][e >> 2][e & 3], k[5][e >> 2][e & 3], lc);
            const bool usex = q >= 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e >> 2][e & 3] = usex ? xv[e >> 2][e & 3] : d[e >> 2][e & 3];
            bf[nt][0] = split8(d[0], d[1]);
         }
         init_bias<2, 2>(y2, v2 + EncV4L2::v_cb_b, q);
         gemm<2, 2, 1, 2>(y2, f2 + EncV4L2::f_pw, 0, bf, lane);
#pragma unroll
         for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
               for (int r = 0; r < 4; ++r) y2[nt][mt][r] = relu(y2[nt][mt][r]);
      }
      f4 z2[2][2];
      v4_conv1x1<EncV4L2, 32, 2>(y2, z2, f2, v2, lane);
      // stride 2: step 2 t' of the first / second tile -> column t' / 8 + t' of the pair tile (t' < t2 <= 6); same quad, same registers
      const int t2 = (a.t1 + 1) >> 1, t3 = a.s3 == 2 ? (t2 + 1) >> 1 : t2;
      f4 x3[2];
      {
         const int t = v4p_step(lc);
         const bool live = t < t2;
         const int src = 4 * (16 * q + 2 * (live ? t : 0));
         const bool second = v4p_chunk(lc) == 1;
#pragma unroll
         for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
               const float a0 = z2[0][mt][r], a1 = z2[1][mt][r];
               const int g0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, a0));
               const int g1 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, a1));
               x3[mt][r] = live ? __builtin_bit_cast(float, second ? g1 : g0) : 0.0f;
            }
      }
      // ---- stage 3 on the pair tile (t2 steps per chunk)
      f4 y3[1][2], z3[1][2];
      v4_conv_block<EncV4L3, 32, false>(x3, y3[0], f3, v3, lane);
      v4_conv1x1<EncV4L3, 32, 1>(y3, z3, f3, v3, lane);
      // stride s3: step s3 t' -> column (chunk) 8 + t' (t' < t3 <= 3)
      f4 x4[2];
      {
         const int t = v4p_step(lc);
         const bool live = t < t3;
         const int src = 4 * (16 * q + 8 * v4p_chunk(lc) + a.s3 * (live ? t : 0));
#pragma unroll
         for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
               const float a0 = z3[0][mt][r];
               const int g0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, a0));
               x4[mt][r] = live ? __builtin_bit_cast(float, g0) : 0.0f;
            }
      }
      // ---- stage 4 on the pair tile (t3 steps per chunk), stride 1
      f4 y4[1][4], z4[1][4];
      v4_conv_block<EncV4L4, 64, true>(x4, y4[0], f4_, v4, lane);
      v4_conv1x1<EncV4L4, 64, 1>(y4, z4, f4_, v4, lane);
      // ---- split-fp16 LSTM-native tiles (common.h lstm_xh_index): a (chunk, step) row = 64 units x {hi, lo}; this lane owns units 16 mt + 4 q .. + 3
      {
         const int item = item0 + v4p_chunk(lc), t = v4p_step(lc);
         if (t < t3 && item < a.n_chunks) {
            int st_, ch_;
            a.map.split(item, st_, ch_);
            _Float16 *dst = reinterpret_cast<_Float16 *>(a.out) + lstm_xh_index(st_, ch_, a.map.C, t, 4 * q, a.ts);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
               h2 hi[2], lo[2];
               split2(z4[0][mt][0], z4[0][mt][1], hi[0], lo[0]); split2(z4[0][mt][2], z4[0][mt][3], hi[1], lo[1]);
               *reinterpret_cast<h4 *>(dst + 16 * mt) = h4{hi[0][0], hi[0][1], hi[1][0], hi[1][1]};
               *reinterpret_cast<h4 *>(dst + kLstmTile * 64 + 16 * mt) = h4{lo[0][0], lo[0][1], lo[1][0], lo[1][1]};
            }
         }
      }
   }
}

// grid: up to three 8-wave workgroups per CU the stream may use (52 KB of LDS each), never more than there are batches for its waves
void launch_enc_fused_v4(const EncV4Args &a, int max_cus, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   const int nb = (a.n_chunks + 1) / 2;
   int g = (nb + 7) / 8;
   if (g > 3 * max_cus) g = 3 * max_cus;
   if (g < 1) g = 1;
   hipLaunchKernelGGL((k_enc_fused_v4<8>), dim3(g), dim3(512), 0, st, a);
}

}  // namespace vadc
