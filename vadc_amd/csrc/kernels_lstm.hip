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
// k_lstm_mfma (default): a workgroup = 16 streams x 4 waves.  Per (step, layer) the gate pre-activations are
//   G[256 x 16] = W[256 x 128] . [x ; h][128 x 16]   on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
//   Wave w owns hidden units [16w,16w+16) for ALL four gates, so the i,f,g,o values of one (unit, stream) land in
//   the same lane/register of its four accumulators and the cell update is register-local.  Both layers' weights
//   live in registers as MFMA A-fragments (2 x 4 x 32 VGPRs) for the whole call; [x;h] is the B operand, read
//   from a small LDS tile.  4 barriers per step.
// k_lstm_simple: bring-up/reference variant (one wave per stream, weights streamed from L2), selectable with
//   vadc_amd_set_option(e, "lstm", 1); used by the tests to A/B the MFMA kernel on the device.
#include "common.h"

namespace vadc {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// ------------------------------------------------------------------------------------------------
// simple variant
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_lstm_simple(const float *__restrict__ enc,   // [S][C][64][7]
                                                    LstmWeights w,
                                                    float *__restrict__ hs,          // [S][2][64]
                                                    float *__restrict__ cs,          // [S][2][64]
                                                    float *__restrict__ probs,       // [S][C][2]
                                                    int n_streams, int n_chunks)
{
   __shared__ float xh[128];
   const int s = blockIdx.x;
   const int j = threadIdx.x;
   if (s >= n_streams) return;
   float h0 = hs[(size_t)s * 128 + j], h1 = hs[(size_t)s * 128 + 64 + j];
   float c0 = cs[(size_t)s * 128 + j], c1 = cs[(size_t)s * 128 + 64 + j];
   const float dw0 = w.dec_w[j], dw1 = w.dec_w[64 + j];

   for (int ch = 0; ch < n_chunks; ++ch) {
      const float *x = enc + ((size_t)s * n_chunks + ch) * 64 * 7;
      float d0 = 0.0f, d1 = 0.0f;                          // sum over t of (w . relu(h1_t))
      for (int t = 0; t < 7; ++t) {
#pragma unroll
         for (int l = 0; l < 2; ++l) {
            const float xin = (l == 0) ? x[j * 7 + t] : h0;
            __syncthreads();
            xh[j] = xin;
            xh[64 + j] = (l == 0) ? h0 : h1;
            __syncthreads();
            const float *wT = w.wT + (size_t)l * 128 * 256;
            float gi = w.b[l * 256 + j], gf = w.b[l * 256 + 64 + j], gg = w.b[l * 256 + 128 + j], go = w.b[l * 256 + 192 + j];
            for (int k = 0; k < 128; ++k) {
               const float v = xh[k];
               gi = fmaf(wT[k * 256 + j], v, gi);
               gf = fmaf(wT[k * 256 + 64 + j], v, gf);
               gg = fmaf(wT[k * 256 + 128 + j], v, gg);
               go = fmaf(wT[k * 256 + 192 + j], v, go);
            }
            const float ig = sigmoidf_(gi), fg = sigmoidf_(gf), g = tanhf(gg), og = sigmoidf_(go);
            if (l == 0) { c0 = fg * c0 + ig * g; h0 = og * tanhf(c0); }
            else        { c1 = fg * c1 + ig * g; h1 = og * tanhf(c1); }
         }
         const float r = fmaxf(h1, 0.0f);
         float p0 = dw0 * r, p1 = dw1 * r;
#pragma unroll
         for (int off = 32; off > 0; off >>= 1) { p0 += __shfl_xor(p0, off); p1 += __shfl_xor(p1, off); }
         d0 += p0 + w.dec_b[0];
         d1 += p1 + w.dec_b[1];
      }
      if (j == 0) {
         probs[((size_t)s * n_chunks + ch) * 2 + 0] = sigmoidf_(d0 / 7.0f);
         probs[((size_t)s * n_chunks + ch) * 2 + 1] = sigmoidf_(d1 / 7.0f);
      }
   }
   hs[(size_t)s * 128 + j] = h0; hs[(size_t)s * 128 + 64 + j] = h1;
   cs[(size_t)s * 128 + j] = c0; cs[(size_t)s * 128 + 64 + j] = c1;
}

// ------------------------------------------------------------------------------------------------
// MFMA variant
// ------------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kTileS = 16;          // streams per workgroup (= MFMA N)

// v_mfma_f32_16x16x4_f32 lane maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[row = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][col = l & 15]
//   D: lane l, reg r holds D[row = 4 (l >> 4) + r][col = l & 15]
__global__ __launch_bounds__(256, 1) void k_lstm_mfma(const float *__restrict__ enc,   // [S][C][64][7]
                                                      LstmWeights w,
                                                      float *__restrict__ hs, float *__restrict__ cs,
                                                      float *__restrict__ probs,
                                                      int n_streams, int n_chunks)
{
   __shared__ float xs[7 * 64 * kTileS];     // [t][unit][stream]  the chunk's encoder frames
   __shared__ float hb[2][64 * kTileS];      // [layer][unit][stream]  current hidden state
   __shared__ float dacc[2 * kTileS];

   const int tid = threadIdx.x;
   const int lane = tid & 63;
   const int wv = tid >> 6;                  // wave: owns hidden units [16 wv, 16 wv + 16)
   const int col = lane & 15;                // stream within the tile
   const int quad = lane >> 4;
   const int s0 = blockIdx.x * kTileS;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = (s0 + col) < n_streams;

   // A fragments: a[l][g][kk] = W[l][g*64 + 16 wv + (lane & 15)][4 kk + (lane >> 4)]
   float a[2][4][32];
#pragma unroll
   for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
         const float *row = w.w + ((size_t)l * 256 + g * 64 + 16 * wv + (lane & 15)) * 128 + quad;
#pragma unroll
         for (int kk = 0; kk < 32; ++kk) a[l][g][kk] = row[4 * kk];
      }
   // biases / cell state in the D layout: unit = 16 wv + 4 quad + r, stream = col
   float bias[2][4][4], c[2][4];
#pragma unroll
   for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wv + 4 * quad + r;
#pragma unroll
         for (int g = 0; g < 4; ++g) bias[l][g][r] = w.b[l * 256 + g * 64 + u];
         c[l][r] = cs[(size_t)s_col * 128 + l * 64 + u];
         hb[l][u * kTileS + col] = hs[(size_t)s_col * 128 + l * 64 + u];
      }

   for (int ch = 0; ch < n_chunks; ++ch) {
      __syncthreads();                       // previous chunk's readers of xs/dacc are done
      for (int i = tid; i < kTileS * 448; i += 256) {
         const int sc = i / 448, rem = i - sc * 448;          // rem = unit*7 + t  (coalesced over rem)
         const int u = rem / 7, t = rem - u * 7;
         const int ss = min(s0 + sc, n_streams - 1);
         xs[(t * 64 + u) * kTileS + sc] = enc[((size_t)ss * n_chunks + ch) * 448 + rem];
      }
      if (tid < 2 * kTileS) dacc[tid] = 0.0f;
      __syncthreads();

      for (int t = 0; t < 7; ++t) {
#pragma unroll
         for (int l = 0; l < 2; ++l) {
            const float *xin = (l == 0) ? (xs + t * 64 * kTileS) : hb[0];   // rows k < 64
            const float *hin = hb[l];                                        // rows k >= 64
            f4v acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc[g][0] = bias[l][g][0]; acc[g][1] = bias[l][g][1]; acc[g][2] = bias[l][g][2]; acc[g][3] = bias[l][g][3]; }
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
               const int k = 4 * kk + quad;
               const float bv = (kk < 16) ? xin[k * kTileS + col] : hin[(k - 64) * kTileS + col];
#pragma unroll
               for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l][g][kk], bv, acc[g], 0, 0, 0);
            }
            float hn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
               const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]);
               const float gg = tanhf(acc[2][r]), og = sigmoidf_(acc[3][r]);
               c[l][r] = fg * c[l][r] + ig * gg;
               hn[r] = og * tanhf(c[l][r]);
            }
            __syncthreads();                 // every wave has finished reading hb[l] (and hb[0] as input)
#pragma unroll
            for (int r = 0; r < 4; ++r) hb[l][(16 * wv + 4 * quad + r) * kTileS + col] = hn[r];
            __syncthreads();
         }
         // decoder partial for this step: 32 threads, (stream, output)
         if (tid < 2 * kTileS) {
            const int sc = tid & 15, f = tid >> 4;
            float d = w.dec_b[f];
            for (int u = 0; u < 64; ++u) d = fmaf(w.dec_w[f * 64 + u], fmaxf(hb[1][u * kTileS + sc], 0.0f), d);
            dacc[tid] += d;
         }
      }
      if (tid < 2 * kTileS) {
         const int sc = tid & 15, f = tid >> 4;
         if (s0 + sc < n_streams) probs[((size_t)(s0 + sc) * n_chunks + ch) * 2 + f] = sigmoidf_(dacc[tid] / 7.0f);
      }
   }
   __syncthreads();
   if (col_ok) {
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
         for (int r = 0; r < 4; ++r) {
            const int u = 16 * wv + 4 * quad + r;
            cs[(size_t)s_col * 128 + l * 64 + u] = c[l][r];
            hs[(size_t)s_col * 128 + l * 64 + u] = hb[l][u * kTileS + col];
         }
   }
}

void launch_lstm(int variant, const float *enc, const LstmWeights &w, float *hs, float *cs, float *probs,
                 int n_streams, int n_chunks, hipStream_t st)
{
   if (variant == 1)
      hipLaunchKernelGGL(k_lstm_simple, dim3(n_streams), dim3(64), 0, st, enc, w, hs, cs, probs, n_streams, n_chunks);
   else
      hipLaunchKernelGGL(k_lstm_mfma, dim3((n_streams + kTileS - 1) / kTileS), dim3(256), 0, st, enc, w, hs, cs, probs, n_streams, n_chunks);
}

}  // namespace vadc
