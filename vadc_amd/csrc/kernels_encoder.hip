// kernels_encoder.hip -- one encoder ("transformer") layer per launch, for gfx950.
//
// Replaces, per chunk (reference file:line):
//   rest of adaptive_audio_normalization_inplace (smoothing, mean, subtract)   misc.c:65-96   (layer 1 only)
//   conv_block  (dw k5 pad2 -> ReLU -> pw 1x1 -> + proj(x)|x -> ReLU)           conv.c:761-814, :17-113, :532-589
//   transformer_block (2-head attention, post-norm, FFN)                        transformer.c:13-234,
//                                                                               tensor.h:675-784, misc.c:143-210
//   conv k=1 stride s + bias -> BatchNorm1d -> ReLU                             transformer.c:279-290,
//                                                                               conv.c:597-709, misc.c:221-258
//
// MAPPING.  A thread owns one (chunk, time-step) column for the whole layer and keeps that column's D
// channel values in registers; every weight it multiplies with is the same for all lanes of the wave, so
// weights are read through the scalar cache as SGPR operands of v_fmac and never touch LDS or VGPR
// loads.  LayerNorm (over D) and softmax (over T keys) are therefore purely in-thread reductions -- no
// cross-lane traffic at all.  The only exchange between threads of a chunk is attention's Q and V
// ([T][2D] per chunk, through LDS, one barrier).  The stride-s 1x1 convolution only needs the thread's own
// column (k = 1), so it is computed by the threads with t % s == 0 straight from registers; BatchNorm is
// folded into its weights at load time.  The depthwise conv reads its 5 taps straight from global memory
// (L1/L2-resident: every element is touched by 5 neighbouring lanes of the same wave).
//
// Numerics: fp32 with FMA contraction; the reference's summation orders are NOT reproduced here -- SURVEY.md
// Appendix F: everything after the normalization contributes ~1e-6 to the probability in any fp32 order.
#include "common.h"

namespace vadc {

__device__ __forceinline__ float norm_offset_fast(const float *__restrict__ fmp, size_t fm_stride)
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
         int q = t + i - 3;
         q = q < 0 ? -q : q;
         q = q >= kFrames ? 2 * (kFrames - 1) - q : q;
         r += fm[q] * filt[i];
      }
      total += r;
   }
   return total / 25.0f;
}

// y[o] = b[o] + sum_d W[o][d] x[d], four outputs at a time (independent FMA chains), W rows contiguous
template <int NOUT, int NIN>
__device__ __forceinline__ void matvec(const float *__restrict__ W, const float *__restrict__ b,
                                       const float (&x)[NIN], float (&y)[NOUT])
{
#pragma unroll
   for (int o = 0; o < NOUT; o += 4) {
      float a0 = b[o], a1 = b[o + 1], a2 = b[o + 2], a3 = b[o + 3];
#pragma unroll
      for (int d = 0; d < NIN; ++d) {
         a0 = fmaf(W[(o + 0) * NIN + d], x[d], a0);
         a1 = fmaf(W[(o + 1) * NIN + d], x[d], a1);
         a2 = fmaf(W[(o + 2) * NIN + d], x[d], a2);
         a3 = fmaf(W[(o + 3) * NIN + d], x[d], a3);
      }
      y[o] = a0; y[o + 1] = a1; y[o + 2] = a2; y[o + 3] = a3;
   }
}

// misc.c:143-210: biased variance, eps 1e-5, (x*rstd - mean*rstd)*w + b
template <int D>
__device__ __forceinline__ void layer_norm(float (&x)[D], const float *__restrict__ w, const float *__restrict__ b)
{
   float s = 0.0f;
#pragma unroll
   for (int i = 0; i < D; ++i) s += x[i];
   const float mean = s * (1.0f / D);
   float vs = 0.0f;
#pragma unroll
   for (int i = 0; i < D; ++i) { const float d = x[i] - mean; vs = fmaf(d, d, vs); }
   const float rstd = 1.0f / sqrtf(vs * (1.0f / D) + 1e-5f);
   const float mr = mean * rstd;
#pragma unroll
   for (int i = 0; i < D; ++i) x[i] = fmaf(fmaf(x[i], rstd, -mr), w[i], b[i]);
}

template <int CIN, int D, int T, int STRIDE, bool HAS_PROJ, bool FIRST, int CB, int NTHREADS, bool LSTM_OUT>
__global__ __launch_bounds__(NTHREADS) void k_layer(const float *__restrict__ in,   // [n][CIN][T]
                                                    const float *__restrict__ fm,   // [n][25] (FIRST) or null
                                                    LayerWeights w,
                                                    float *__restrict__ out,        // [n][D][TOUT]
                                                    int n_chunks, ItemMap map, size_t fm_stride)
{
   constexpr int TOUT = 1 + (T - 1) / STRIDE;
   constexpr int HD = D / 2;
   constexpr int PITCH = T * 2 * D + 4;                 // +4 floats: de-phase the chunks' LDS banks
   __shared__ float qv[CB * PITCH];

   const int tid = threadIdx.x;
   const int cb = tid / T;
   const int t = tid - cb * T;
   const int chunk_raw = blockIdx.x * CB + cb;
   const bool valid = (cb < CB) && (chunk_raw < n_chunks);
   const int chunk = map(valid ? chunk_raw : (n_chunks - 1));   // clamped: loads stay in bounds, stores are skipped
   const int cbs = cb < CB ? cb : CB - 1;

   const float *x_in = in + (size_t)chunk * CIN * T;
   float mm = 0.0f;
   if (FIRST) mm = norm_offset_fast(fm + (size_t)chunk * kFrames, fm_stride);         // misc.c:65-82

   // ---- conv block: y = relu(pw(relu(dw(x))) + proj(x) | x) -------------------------------------------
   float y[D];
   {
      float acc[D];
#pragma unroll
      for (int o = 0; o < D; ++o) acc[o] = w.pw_b[o] + (HAS_PROJ ? w.pj_b[o] : 0.0f);
      const bool l2 = t >= 2, l1 = t >= 1, r1 = t + 1 < T, r2 = t + 2 < T;
#pragma unroll 2
      for (int c = 0; c < CIN; ++c) {
         const float *row = x_in + c * T + t;
         const float x0 = row[0] - mm;                                     // misc.c:84-96
         const float xm2 = l2 ? row[-2] - mm : 0.0f, xm1 = l1 ? row[-1] - mm : 0.0f;
         const float xp1 = r1 ? row[1] - mm : 0.0f, xp2 = r2 ? row[2] - mm : 0.0f;
         const float *k = w.dw_w + c * 5;
         float dv = w.dw_b[c];
         dv = fmaf(xm2, k[0], dv); dv = fmaf(xm1, k[1], dv); dv = fmaf(x0, k[2], dv);
         dv = fmaf(xp1, k[3], dv); dv = fmaf(xp2, k[4], dv);
         dv = fmaxf(dv, 0.0f);
         const float *pw = w.pwT + c * D;
#pragma unroll
         for (int o = 0; o < D; ++o) acc[o] = fmaf(pw[o], dv, acc[o]);
         if (HAS_PROJ) {
            const float *pj = w.pjT + c * D;
#pragma unroll
            for (int o = 0; o < D; ++o) acc[o] = fmaf(pj[o], x0, acc[o]);
         }
      }
      if (!HAS_PROJ) {
#pragma unroll
         for (int o = 0; o < D; ++o) acc[o] += x_in[o * T + t];            // CIN == D: identity residual
      }
#pragma unroll
      for (int o = 0; o < D; ++o) y[o] = fmaxf(acc[o], 0.0f);
   }

   // ---- attention: QKV; Q and V of every time step go to LDS, K stays in registers --------------------
   float kk[D];
   {
      float *my = qv + cbs * PITCH + t * 2 * D;
      float q[D];
      matvec<D, D>(w.qkv_w, w.qkv_b, y, q);
      if (cb < CB) {
#pragma unroll
         for (int o = 0; o < D; o += 4) *reinterpret_cast<float4 *>(my + o) = make_float4(q[o], q[o + 1], q[o + 2], q[o + 3]);
      }
      matvec<D, D>(w.qkv_w + D * D, w.qkv_b + D, y, kk);
      matvec<D, D>(w.qkv_w + 2 * D * D, w.qkv_b + 2 * D, y, q);
      if (cb < CB) {
#pragma unroll
         for (int o = 0; o < D; o += 4) *reinterpret_cast<float4 *>(my + D + o) = make_float4(q[o], q[o + 1], q[o + 2], q[o + 3]);
      }
   }
   __syncthreads();

   float att[D];
   {
      const float scale = 1.0f / sqrtf((float)HD);                         // transformer.c:114
      const float *base = qv + cbs * PITCH;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
         float s[T];
         float mx = -3.0e38f;
#pragma unroll
         for (int j = 0; j < T; ++j) {                                     // a[i][j] = k_i . q_j  (transformer.c:104-105)
            const float *qj = base + j * 2 * D + h * HD;
            float a = 0.0f;
#pragma unroll
            for (int e = 0; e < HD; e += 4) {
               const float4 q4 = *reinterpret_cast<const float4 *>(qj + e);
               a = fmaf(kk[h * HD + e], q4.x, a); a = fmaf(kk[h * HD + e + 1], q4.y, a);
               a = fmaf(kk[h * HD + e + 2], q4.z, a); a = fmaf(kk[h * HD + e + 3], q4.w, a);
            }
            s[j] = a * scale;
            mx = fmaxf(mx, s[j]);
         }
         float sum = 0.0f;                                                 // tensor.h:751-784
#pragma unroll
         for (int j = 0; j < T; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
         const float inv = 1.0f / sum;
         float o[HD];
#pragma unroll
         for (int e = 0; e < HD; ++e) o[e] = 0.0f;
#pragma unroll
         for (int j = 0; j < T; ++j) {                                     // attn_i = sum_j a_ij v_j
            const float a = s[j] * inv;
            const float *vj = base + j * 2 * D + D + h * HD;
#pragma unroll
            for (int e = 0; e < HD; e += 4) {
               const float4 v4 = *reinterpret_cast<const float4 *>(vj + e);
               o[e] = fmaf(a, v4.x, o[e]); o[e + 1] = fmaf(a, v4.y, o[e + 1]);
               o[e + 2] = fmaf(a, v4.z, o[e + 2]); o[e + 3] = fmaf(a, v4.w, o[e + 3]);
            }
         }
#pragma unroll
         for (int e = 0; e < HD; ++e) att[h * HD + e] = o[e];
      }
   }

   // ---- out projection, residual, LN1, FFN, residual, LN2 (transformer.c:202-220) -----------------------
   {
      float p[D];
      matvec<D, D>(w.out_w, w.out_b, att, p);
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] += p[i];
      layer_norm<D>(y, w.n1_w, w.n1_b);
      matvec<D, D>(w.l1_w, w.l1_b, y, att);
#pragma unroll
      for (int i = 0; i < D; ++i) att[i] = fmaxf(att[i], 0.0f);
      matvec<D, D>(w.l2_w, w.l2_b, att, p);
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] += p[i];
      layer_norm<D>(y, w.n2_w, w.n2_b);
   }

   // ---- conv k=1 stride s (+ folded BatchNorm) -> ReLU; only the surviving time steps -------------------
   if (valid && (t % STRIDE) == 0) {
      // reference layout [n][D][TOUT] (element stride TOUT), or the LSTM-native tile layout (last layer)
      constexpr int ostride = LSTM_OUT ? kLstmTile : TOUT;
      float *dst;
      if (LSTM_OUT) {
         int st_, ch_;
         map.split(chunk_raw, st_, ch_);
         dst = out + lstm_x_index(st_, ch_, map.C, t / STRIDE, 0);
      } else {
         dst = out + (size_t)chunk * D * TOUT + t / STRIDE;
      }
#pragma unroll
      for (int o = 0; o < D; o += 4) {
         float a0 = w.cv_b[o], a1 = w.cv_b[o + 1], a2 = w.cv_b[o + 2], a3 = w.cv_b[o + 3];
#pragma unroll
         for (int d = 0; d < D; ++d) {
            a0 = fmaf(w.cv_w[(o + 0) * D + d], y[d], a0);
            a1 = fmaf(w.cv_w[(o + 1) * D + d], y[d], a1);
            a2 = fmaf(w.cv_w[(o + 2) * D + d], y[d], a2);
            a3 = fmaf(w.cv_w[(o + 3) * D + d], y[d], a3);
         }
         dst[(o + 0) * ostride] = fmaxf(a0, 0.0f);
         dst[(o + 1) * ostride] = fmaxf(a1, 0.0f);
         dst[(o + 2) * ostride] = fmaxf(a2, 0.0f);
         dst[(o + 3) * ostride] = fmaxf(a3, 0.0f);
      }
   }
}

// chunks per workgroup / threads per workgroup, chosen so that CB*T fills the waves
//   L1: T=25 -> 10 chunks = 250 of 256 lanes     L2: T=13 -> 9 chunks = 117 of 128 lanes
//   L3/L4: T=7 -> 9 chunks = 63 of 64 lanes
void launch_layer(int layer, const float *in, const float *fm, const LayerWeights &w, float *out, int n, ItemMap map,
                  int lstm_layout, size_t fm_stride, hipStream_t st)
{
   switch (layer) {
   case 0: hipLaunchKernelGGL((k_layer<129, 16, 25, 2, true, true, 10, 256, false>), dim3((n + 9) / 10), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride); break;
   case 1: hipLaunchKernelGGL((k_layer<16, 32, 13, 2, true, false, 9, 128, false>), dim3((n + 8) / 9), dim3(128), 0, st, in, fm, w, out, n, map, fm_stride); break;
   case 2: hipLaunchKernelGGL((k_layer<32, 32, 7, 1, false, false, 9, 64, false>), dim3((n + 8) / 9), dim3(64), 0, st, in, fm, w, out, n, map, fm_stride); break;
   case 3:
      if (lstm_layout) hipLaunchKernelGGL((k_layer<32, 64, 7, 1, true, false, 9, 64, true>), dim3((n + 8) / 9), dim3(64), 0, st, in, fm, w, out, n, map, fm_stride);
      else             hipLaunchKernelGGL((k_layer<32, 64, 7, 1, true, false, 9, 64, false>), dim3((n + 8) / 9), dim3(64), 0, st, in, fm, w, out, n, map, fm_stride);
      break;
   }
}

}  // namespace vadc
