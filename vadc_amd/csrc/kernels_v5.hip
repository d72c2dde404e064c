// kernels_v5.hip -- Silero VAD v5 shapes (16 kHz): 64-sample context + 512-sample window per chunk.
//
// Replaces, per chunk (reference file:line): process_chunks_v5's context handling vadc.c:105-162; silero_vad.py::Silero_Vad_5 :367-434 =
// STFT_conv2 (:290-312: right reflect pad 64, basis [258,1,256], hop 128 -> 4 frames, magnitude), four MobileOneBlocks (:314-350: Conv1d k = 3,
// padding 1, strides 1 / 2 / 2 / 1, ReLU: [129,4] -> [128,4] -> [64,2] -> [64,1] -> [128,1]), LSTM(128, one layer) over the one step with carried
// state, decoder ReLU -> conv 128 -> 1 -> sigmoid (:326-341); the same arithmetic as the reference's C test silero_v5_test (test.c:2027-2196).
// The reference ships no v5 weights: parity is shape / arithmetic parity on seeded weights against its PyTorch class.
//
// MAPPING.  Every dense contraction is an fp32 MFMA GEMM (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains: the parity target is a framework
// evaluation in any fp32 order) over the (chunk, frame) columns of a workgroup:
//   k_v5_encoder: 8 chunks = 32 STFT columns per workgroup.  The frames are laid out im2col in LDS ([column][256 taps], pitch 260), the STFT is
//     [258 x 256] x [256 x 32]; a k = 3 conv is a GEMM with K = (tap, input channel) whose B operand is read from the previous stage's LDS tile with
//     the tap's time shift (zero outside the chunk); host-prepacked fragment-major weights (one coalesced 256-byte load per MFMA A operand);
//     the last stage is the LSTM's input projection W_ih . x + b (it does not depend on the recurrence), written as GX[item][512].
//   k_v5_lstm: one workgroup per 16 streams, 8 waves; wave w owns hidden units [16w, 16w + 16) of all four gates, W_hh in registers for the whole
//     call (128 VGPRs), h [128 x 16] in LDS double buffered, one barrier per chunk; accumulators start at GX; decoder folded in.
//   k_v5_context: the 64 samples that precede the next call's first window, kept per stream on the device (fp32).
#include "common.h"

namespace vadc {

typedef float f4v5 __attribute__((ext_vector_type(4)));

struct V5Weights {
   const float *stft_f;        // [17 m-tiles][64 k-steps][64 lanes]  A fragments of the basis (rows >= 258 zero)
   const float *conv_f[4];     // [CO / 16][3 taps * CIP / 4][64]     A fragments, K order (tap, input channel padded to a multiple of 4)
   const float *conv_b[4];     // [CO]
   const float *wih_f;         // [32][32][64]                        A fragments of W_ih (rows i, f, g, o x 128)
   const float *lstm_b;        // [512]  b_ih + b_hh
   const float *whh;           // [512][128] recurrent weights, row-major
   const _Float16 *whh_h;      // [32 m-tiles][4 k-blocks][64 lanes][hi 8 | lo 8]  split-fp16 A fragments of W_hh for v_mfma_f32_16x16x32_f16 (null: a weight
                               // does not fit fp16's range; k_v5_lstm runs)
   const float *dec_w;         // [128]
   const float *dec_b;         // [1]
};

constexpr int kV5Window = 512, kV5Context = 64, kV5Hidden = 128, kV5Gates = 512;
constexpr int kV5Chunks = 8;                     // chunks per encoder workgroup
constexpr int kV5Cols = 4 * kV5Chunks;           // STFT columns (chunk, frame)
constexpr int kV5XPitch = 260;                   // im2col row pitch (floats): 260 % 32 == 4 -> the 16 columns of a B fragment spread over the banks
constexpr int kV5P = kV5Cols + 4;                // pitch of the activation tiles [channel][column]

__device__ __forceinline__ float v5_sample(float v) { return v; }
__device__ __forceinline__ float v5_sample(int16_t v) { return (float)v * (1.0f / 32768.0f); }    // exact (vadc.c:883,898)

// One k = 3 / padding 1 conv stage as a GEMM on the matrix cores.  in: LDS tile [CI][kV5P] (columns = chunk * TIN + t), out: LDS tile [CO][kV5P]
// (columns = chunk * TOUT + t) after bias and ReLU.  Wave `wave` owns M-tiles [wave * NMW, (wave + 1) * NMW); NN = N-tiles of 16 output columns.
template <int CI, int CO, int TIN, int TOUT, int STRIDE, int NMW, int NN>
__device__ __forceinline__ void v5_conv(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ wf, const float *__restrict__ bias,
                                        int wave, int lane)
{
   constexpr int CIP = (CI + 3) / 4 * 4, KT = CIP / 4, KKW = 3 * KT;
   constexpr int NCOL = kV5Chunks * TOUT;
   const int lc = lane & 15, kq = lane >> 4;
   f4v5 acc[NMW][NN];
#pragma unroll
   for (int mi = 0; mi < NMW; ++mi) {
      const float4 b4 = *reinterpret_cast<const float4 *>(bias + 16 * (wave * NMW + mi) + 4 * kq);
#pragma unroll
      for (int ni = 0; ni < NN; ++ni) { acc[mi][ni][0] = b4.x; acc[mi][ni][1] = b4.y; acc[mi][ni][2] = b4.z; acc[mi][ni][3] = b4.w; }
   }
#pragma unroll
   for (int tap = 0; tap < 3; ++tap) {
      // this lane's input column for every N-tile: output column (chunk j, step t) reads input step q = t * STRIDE + tap - 1 of chunk j
      int icol[NN];
      bool ok[NN];
#pragma unroll
      for (int ni = 0; ni < NN; ++ni) {
         const int col = 16 * ni + lc, j = col / TOUT, t = col - j * TOUT, q = t * STRIDE + tap - 1;
         ok[ni] = col < NCOL && q >= 0 && q < TIN;
         icol[ni] = ok[ni] ? j * TIN + q : 0;
      }
      bool any = false;
#pragma unroll
      for (int ni = 0; ni < NN; ++ni) any |= ok[ni];
      if (!__any(any)) continue;                    // a tap that only ever sees padding (TIN = 1: taps 0 and 2)
#pragma unroll 4
      for (int kk = 0; kk < KT; ++kk) {
         const int ci = 4 * kk + kq;
         float a[NMW], b[NN];
#pragma unroll
         for (int mi = 0; mi < NMW; ++mi) a[mi] = wf[((size_t)(wave * NMW + mi) * KKW + tap * KT + kk) * 64 + lane];
#pragma unroll
         for (int ni = 0; ni < NN; ++ni) b[ni] = (ok[ni] && ci < CI) ? in[ci * kV5P + icol[ni]] : 0.0f;
#pragma unroll
         for (int mi = 0; mi < NMW; ++mi)
#pragma unroll
            for (int ni = 0; ni < NN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
   }
#pragma unroll
   for (int mi = 0; mi < NMW; ++mi)
#pragma unroll
      for (int ni = 0; ni < NN; ++ni) {
         const int col = 16 * ni + lc;
         if (col < NCOL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * (wave * NMW + mi) + 4 * kq + r) * kV5P + col] = fmaxf(acc[mi][ni][r], 0.0f);
         }
      }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void k_v5_encoder(const T *__restrict__ pcm,          // [S][C][512]
                                                       const float *__restrict__ ctx,      // [S][64] the samples before each stream's first window of this call
                                                       V5Weights w,
                                                       float *__restrict__ gx,             // [S * C][512]  W_ih . enc + b
                                                       int n_items, int n_chunks)
{
   // region A: im2col frames, then the STFT output, then C0 / C2 ; region B: magnitudes, then C1 / C3
   constexpr int kA = 258 * kV5P > kV5Cols * kV5XPitch ? 258 * kV5P : kV5Cols * kV5XPitch;
   __shared__ __attribute__((aligned(16))) float RA[kA];
   __shared__ __attribute__((aligned(16))) float RB[132 * kV5P];
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int lc = lane & 15, kq = lane >> 4;
   const int item0 = blockIdx.x * kV5Chunks;

   // ---- stage: [context 64 | window 512 | reflect 64] per chunk, written im2col: column (j, fr) holds samples [128 fr, 128 fr + 256) ----
   for (int i = tid; i < kV5Chunks * 640; i += 256) {
      const int j = i / 640, p = i - j * 640;
      const int item = min(item0 + j, n_items - 1);
      const int s = item / n_chunks, c = item - s * n_chunks;
      int q = p < 576 ? p : 2 * 575 - p;                               // F.pad(input, (0, 64), "reflect"): padded[576 + k] = input[574 - k]
      float v;
      if (q >= kV5Context) v = v5_sample(pcm[(size_t)item * kV5Window + (q - kV5Context)]);
      else if (c > 0)      v = v5_sample(pcm[(size_t)(item - 1) * kV5Window + (kV5Window - kV5Context + q)]);     // the previous window's tail (vadc.c:132-135)
      else                 v = ctx[(size_t)s * kV5Context + q];                                                  // carried from the previous call (vadc.c:124)
#pragma unroll
      for (int fr = 0; fr < 4; ++fr) {
         const int k = p - 128 * fr;
         if (k >= 0 && k < 256) RA[(4 * j + fr) * kV5XPitch + k] = v;
      }
   }
   __syncthreads();

   // ---- STFT: [272 x 256] x [256 x 32]; wave w owns M-tiles w, w + 4, w + 8, w + 12 and (wave 0) 16 ----
   {
      f4v5 acc[5][2];
#pragma unroll
      for (int mi = 0; mi < 5; ++mi)
#pragma unroll
         for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (f4v5){0.0f, 0.0f, 0.0f, 0.0f};
      const int m4 = wave == 0 ? 16 : wave + 12;                      // waves 1..3 repeat their last tile (discarded)
#pragma unroll 4
      for (int kk = 0; kk < 64; ++kk) {
         float a[5], b[2];
#pragma unroll
         for (int mi = 0; mi < 4; ++mi) a[mi] = w.stft_f[((size_t)(wave + 4 * mi) * 64 + kk) * 64 + lane];
         a[4] = w.stft_f[((size_t)m4 * 64 + kk) * 64 + lane];
#pragma unroll
         for (int ni = 0; ni < 2; ++ni) b[ni] = RA[(16 * ni + lc) * kV5XPitch + 4 * kk + kq];
#pragma unroll
         for (int mi = 0; mi < 5; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();                                                // every wave is done reading the frames: region A becomes the STFT output
#pragma unroll
      for (int mi = 0; mi < 5; ++mi) {
         const int mt = mi < 4 ? wave + 4 * mi : 16;
         if (mi == 4 && wave != 0) continue;
#pragma unroll
         for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
               const int row = 16 * mt + 4 * kq + r;
               if (row < 258) RA[row * kV5P + 16 * ni + lc] = acc[mi][ni][r];
            }
      }
   }
   __syncthreads();
   for (int i = tid; i < 132 * kV5Cols; i += 256) {                    // magnitude (silero_vad.py:308-311); rows 129..131 are the zero padding of K
      const int f = i / kV5Cols, col = i - f * kV5Cols;
      float m = 0.0f;
      if (f < 129) { const float re = RA[f * kV5P + col], im = RA[(129 + f) * kV5P + col]; m = sqrtf(re * re + im * im); }
      RB[f * kV5P + col] = m;
   }
   __syncthreads();
   v5_conv<129, 128, 4, 4, 1, 2, 2>(RB, RA, w.conv_f[0], w.conv_b[0], wave, lane);      // [129,4] -> [128,4]
   __syncthreads();
   v5_conv<128, 64, 4, 2, 2, 1, 1>(RA, RB, w.conv_f[1], w.conv_b[1], wave, lane);       // -> [64,2]
   __syncthreads();
   v5_conv<64, 64, 2, 1, 2, 1, 1>(RB, RA, w.conv_f[2], w.conv_b[2], wave, lane);        // -> [64,1]
   __syncthreads();
   v5_conv<64, 128, 1, 1, 1, 2, 1>(RA, RB, w.conv_f[3], w.conv_b[3], wave, lane);       // -> [128,1]
   __syncthreads();
   // ---- LSTM input projection: GX[item][512] = W_ih [512 x 128] . enc [128 x 8 chunks] + (b_ih + b_hh); wave w owns M-tiles 8w .. 8w + 7 ----
   {
      f4v5 acc[8];
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
         const float4 b4 = *reinterpret_cast<const float4 *>(w.lstm_b + 16 * (8 * wave + mi) + 4 * kq);
         acc[mi] = (f4v5){b4.x, b4.y, b4.z, b4.w};
      }
#pragma unroll 4
      for (int kk = 0; kk < 32; ++kk) {
         const float b = lc < kV5Chunks ? RB[(4 * kk + kq) * kV5P + lc] : 0.0f;
#pragma unroll
         for (int mi = 0; mi < 8; ++mi)
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.wih_f[((size_t)(8 * wave + mi) * 32 + kk) * 64 + lane], b, acc[mi], 0, 0, 0);
      }
      if (lc < kV5Chunks && item0 + lc < n_items) {
#pragma unroll
         for (int mi = 0; mi < 8; ++mi)
            *reinterpret_cast<float4 *>(gx + (size_t)(item0 + lc) * kV5Gates + 16 * (8 * wave + mi) + 4 * kq) = make_float4(acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]);
      }
   }
}

__device__ __forceinline__ float v5_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float v5_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v) + 1.0f); }

__global__ __launch_bounds__(512, 1) void k_v5_lstm(const float *__restrict__ gx,          // [S * C][512]
                                                    V5Weights w,
                                                    float *__restrict__ hs, float *__restrict__ cs,     // [S][128]
                                                    float *__restrict__ probs,                          // [S][C][2]
                                                    int n_streams, int n_chunks)
{
   __shared__ float hb[2][kV5Hidden * 16];           // [parity][unit][stream]
   __shared__ float pd[2][8][16];
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int col = lane & 15, quad = lane >> 4;
   const int s0 = blockIdx.x * 16;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = s0 + col < n_streams;
   // recurrent weights: gate g, rows g * 128 + 16 wave + (lane & 15), k = 4 kk + quad
   float a[4][32];
#pragma unroll
   for (int g = 0; g < 4; ++g) {
      const float *row = w.whh + (size_t)(g * kV5Hidden + 16 * wave + (lane & 15)) * kV5Hidden + quad;
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) a[g][kk] = row[4 * kk];
   }
   float c[4], dw[4];
#pragma unroll
   for (int r = 0; r < 4; ++r) {
      const int u = 16 * wave + 4 * quad + r;
      dw[r] = w.dec_w[u];
      c[r] = cs[(size_t)s_col * kV5Hidden + u];
      hb[0][u * 16 + col] = hs[(size_t)s_col * kV5Hidden + u];
   }
   const float *gx_lane = gx + (size_t)s_col * n_chunks * kV5Gates + 16 * wave + 4 * quad;
   float4 gn[4];
#pragma unroll
   for (int g = 0; g < 4; ++g) gn[g] = *reinterpret_cast<const float4 *>(gx_lane + g * kV5Hidden);
   __syncthreads();
   int par = 0;
   float hlast[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
   for (int ch = 0; ch < n_chunks; ++ch) {
      f4v5 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = (f4v5){gn[g].x, gn[g].y, gn[g].z, gn[g].w};
      if (ch + 1 < n_chunks) {
#pragma unroll
         for (int g = 0; g < 4; ++g) gn[g] = *reinterpret_cast<const float4 *>(gx_lane + (size_t)(ch + 1) * kV5Gates + g * kV5Hidden);
      }
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
         const float bv = hb[par][(4 * kk + quad) * 16 + col];
#pragma unroll
         for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][kk], bv, acc[g], 0, 0, 0);
      }
      float d = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const float ig = v5_sigmoid(acc[0][r]), fg = v5_sigmoid(acc[1][r]), gg = v5_tanh(acc[2][r]), og = v5_sigmoid(acc[3][r]);
         c[r] = fmaf(fg, c[r], ig * gg);
         const float hn = og * v5_tanh(c[r]);
         hlast[r] = hn;
         hb[par ^ 1][(16 * wave + 4 * quad + r) * 16 + col] = hn;
         d = fmaf(dw[r], fmaxf(hn, 0.0f), d);                         // decoder: ReLU -> conv 128 -> 1 (silero_vad.py:335-338)
      }
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      if (quad == 0) pd[ch & 1][wave][col] = d;
      __syncthreads();
      par ^= 1;
      if (wave == 0 && lane < 16 && s0 + lane < n_streams) {
         const float *p = &pd[ch & 1][0][lane];
         const float m = ((p[0] + p[16]) + (p[32] + p[48])) + ((p[64] + p[80]) + (p[96] + p[112])) + w.dec_b[0];
         const float pr = 1.0f / (1.0f + expf(-m));                   // sigmoid; the mean over the one step is the value itself (:412)
         probs[((size_t)(s0 + lane) * n_chunks + ch) * 2 + 0] = pr;
         probs[((size_t)(s0 + lane) * n_chunks + ch) * 2 + 1] = pr;
      }
   }
   if (col_ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wave + 4 * quad + r;
         cs[(size_t)s_col * kV5Hidden + u] = c[r];
         if (n_chunks > 0) hs[(size_t)s_col * kV5Hidden + u] = hlast[r];
      }
   }
}

// The same recurrence with W_hh h as split-fp16 MFMAs (W h ~ Wl hh + Wh hl + Wh hh, fp32 accumulation: the form the v3.1 / v4 LSTM kernels use,
// kernels_lstm.hip): 48 v_mfma_f32_16x16x32_f16 per wave and slot instead of 128 fp32 16x16x4 -- 5.0 -> 1.3 us per slot.  h travels through LDS as
// [stream][unit] halves (hi and lo tiles, pitch 136): a lane's B fragment of a k-block is one 16-byte read per tile, its four new units one
// 8-byte write per tile.
typedef _Float16 v5h8 __attribute__((ext_vector_type(8)));
typedef _Float16 v5h4 __attribute__((ext_vector_type(4)));
constexpr int kV5HP = kV5Hidden + 8;                 // halves per stream row

__global__ __launch_bounds__(512, 1) void k_v5_lstm_h3(const float *__restrict__ gx,       // [S * C][512]
                                                       V5Weights w,
                                                       float *__restrict__ hs, float *__restrict__ cs,  // [S][128]
                                                       float *__restrict__ probs,                       // [S][C][2]
                                                       int n_streams, int n_chunks)
{
   __shared__ __attribute__((aligned(16))) _Float16 hh[2][16 * kV5HP], hl[2][16 * kV5HP];   // [parity][stream][unit]: hi / lo halves of h
   __shared__ float pd[2][8][16];
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int col = lane & 15, quad = lane >> 4;
   const int s0 = blockIdx.x * 16;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = s0 + col < n_streams;
   // recurrent weights of this wave's 16 units: gate g = m-tile 8 g + wave; k-block kb: lane holds k = 32 kb + 8 quad + e
   v5h8 ah[4][4], al[4][4];
#pragma unroll
   for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         const v5h8 *pp = reinterpret_cast<const v5h8 *>(w.whh_h + (((size_t)(8 * g + wave) * 4 + kb) * 64 + lane) * 16);
         ah[g][kb] = pp[0]; al[g][kb] = pp[1];
      }
   float c[4], dw[4], hlast[4];
   {
      v5h4 h4, l4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wave + 4 * quad + r;
         dw[r] = w.dec_w[u];
         c[r] = cs[(size_t)s_col * kV5Hidden + u];
         hlast[r] = hs[(size_t)s_col * kV5Hidden + u];
         h4[r] = (_Float16)hlast[r]; l4[r] = (_Float16)(hlast[r] - (float)h4[r]);
      }
      *reinterpret_cast<v5h4 *>(&hh[0][col * kV5HP + 16 * wave + 4 * quad]) = h4;
      *reinterpret_cast<v5h4 *>(&hl[0][col * kV5HP + 16 * wave + 4 * quad]) = l4;
   }
   const float *gx_lane = gx + (size_t)s_col * n_chunks * kV5Gates + 16 * wave + 4 * quad;
   float4 gn[4];
#pragma unroll
   for (int g = 0; g < 4; ++g) gn[g] = *reinterpret_cast<const float4 *>(gx_lane + g * kV5Hidden);
   __syncthreads();
   int par = 0;
#pragma unroll 1
   for (int ch = 0; ch < n_chunks; ++ch) {
      f4v5 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = (f4v5){gn[g].x, gn[g].y, gn[g].z, gn[g].w};
      if (ch + 1 < n_chunks) {
#pragma unroll
         for (int g = 0; g < 4; ++g) gn[g] = *reinterpret_cast<const float4 *>(gx_lane + (size_t)(ch + 1) * kV5Gates + g * kV5Hidden);
      }
      v5h8 bh[4], bl[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         bh[kb] = *reinterpret_cast<const v5h8 *>(&hh[par][col * kV5HP + 32 * kb + 8 * quad]);
         bl[kb] = *reinterpret_cast<const v5h8 *>(&hl[par][col * kV5HP + 32 * kb + 8 * quad]);
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
         for (int g = 0; g < 4; ++g) {
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[g][kb], bh[kb], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g][kb], bl[kb], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g][kb], bh[kb], acc[g], 0, 0, 0);
         }
      float d = 0.0f;
      v5h4 h4, l4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const float ig = v5_sigmoid(acc[0][r]), fg = v5_sigmoid(acc[1][r]), gg = v5_tanh(acc[2][r]), og = v5_sigmoid(acc[3][r]);
         c[r] = fmaf(fg, c[r], ig * gg);
         float hn = og * v5_tanh(c[r]);
         asm volatile("" : "+v"(hn));                                 // the ROUNDED h is what is split (and what a later call re-splits from the state):
         hlast[r] = hn;                                               // without this the product is contracted into the subtraction below
         h4[r] = (_Float16)hn; l4[r] = (_Float16)(hn - (float)h4[r]);
         d = fmaf(dw[r], fmaxf(hn, 0.0f), d);                         // decoder: ReLU -> conv 128 -> 1 (silero_vad.py:335-338)
      }
      *reinterpret_cast<v5h4 *>(&hh[par ^ 1][col * kV5HP + 16 * wave + 4 * quad]) = h4;
      *reinterpret_cast<v5h4 *>(&hl[par ^ 1][col * kV5HP + 16 * wave + 4 * quad]) = l4;
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      if (quad == 0) pd[ch & 1][wave][col] = d;
      __syncthreads();
      par ^= 1;
      if (wave == 0 && lane < 16 && s0 + lane < n_streams) {
         const float *p = &pd[ch & 1][0][lane];
         const float m = ((p[0] + p[16]) + (p[32] + p[48])) + ((p[64] + p[80]) + (p[96] + p[112])) + w.dec_b[0];
         const float pr = 1.0f / (1.0f + expf(-m));                   // sigmoid; the mean over the one step is the value itself (:412)
         probs[((size_t)(s0 + lane) * n_chunks + ch) * 2 + 0] = pr;
         probs[((size_t)(s0 + lane) * n_chunks + ch) * 2 + 1] = pr;
      }
   }
   if (col_ok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int u = 16 * wave + 4 * quad + r;
         cs[(size_t)s_col * kV5Hidden + u] = c[r];
         hs[(size_t)s_col * kV5Hidden + u] = hlast[r];                // fp32: the state a later call (or the caller) sees is not rounded to halves
      }
   }
}

template <typename T>
__global__ void k_v5_context(const T *__restrict__ pcm, float *__restrict__ ctx, int n_streams, int n_chunks)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_streams * kV5Context) return;
   const int s = i / kV5Context, j = i - s * kV5Context;
   ctx[i] = v5_sample(pcm[((size_t)s * n_chunks + n_chunks - 1) * kV5Window + kV5Window - kV5Context + j]);
}

// front half of a call: encoder + LSTM input projection -> gx, then the streams' new context (read by the next call's encoder on the same stream)
template <typename T>
static void launch_v5_enc_t(const T *pcm, float *ctx, const V5Weights &w, float *gx, int n_streams, int n_chunks, hipStream_t st)
{
   const int n_items = n_streams * n_chunks;
   hipLaunchKernelGGL((k_v5_encoder<T>), dim3((n_items + kV5Chunks - 1) / kV5Chunks), dim3(256), 0, st, pcm, ctx, w, gx, n_items, n_chunks);
   hipLaunchKernelGGL((k_v5_context<T>), dim3((n_streams * kV5Context + 255) / 256), dim3(256), 0, st, pcm, ctx, n_streams, n_chunks);
}
void launch_v5_encoder_f32(const float *pcm, float *ctx, const V5Weights &w, float *gx, int n_streams, int n_chunks, hipStream_t st)
{
   launch_v5_enc_t<float>(pcm, ctx, w, gx, n_streams, n_chunks, st);
}
void launch_v5_encoder_s16(const int16_t *pcm, float *ctx, const V5Weights &w, float *gx, int n_streams, int n_chunks, hipStream_t st)
{
   launch_v5_enc_t<int16_t>(pcm, ctx, w, gx, n_streams, n_chunks, st);
}
// back half: the recurrence + decoder over the call's chunks.  fp32 = true (or no split-fp16 weights): W_hh h as fp32 MFMAs
void launch_v5_lstm(const V5Weights &w, const float *gx, float *hs, float *cs, float *probs, int n_streams, int n_chunks, bool fp32, hipStream_t st)
{
   if (fp32 || !w.whh_h) hipLaunchKernelGGL(k_v5_lstm, dim3((n_streams + 15) / 16), dim3(512), 0, st, gx, w, hs, cs, probs, n_streams, n_chunks);
   else                  hipLaunchKernelGGL(k_v5_lstm_h3, dim3((n_streams + 15) / 16), dim3(512), 0, st, gx, w, hs, cs, probs, n_streams, n_chunks);
}

}  // namespace vadc
