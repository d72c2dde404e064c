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
   // k_v5_encoder_h3 (round 6): every A operand as split-fp16 fragments for v_mfma_f32_16x16x32_f16, x 256 (a power of two: exact), [m-tile][k-block][hi | lo][lane][8]:
   // lane l holds row 16 mt + (l & 15), k = 32 kb + 8 (l >> 4) + e.  All null when a weight x 256 leaves fp16's range or the basis lacks the real-DFT fold symmetries
   // (then k_v5_encoder, fp32 MFMA, serves).
   const _Float16 *h_stft;     // [16][4]: m-tile t < 8 = re of bins 16 t + r, t >= 8 = im of bins 16 (t - 8) + r; k = fold slot (slot j pairs taps j + 1 and 255 - j)
   const _Float16 *h_conv[4];  // conv 0: [8][13]  k-blocks 0..11 = (tap, channel block of 32 of channels 0..127), 12 = channel 128 of tap (l >> 4) in element 0
                               // conv 1: [4][12]  (tap, 128 channels); conv 2: [4][4] taps 1, 2 x 64 channels (tap 0 only ever meets padding); conv 3: [8][2] tap 1
   const _Float16 *h_wih;      // [32][4]
   const float *wny;           // [128] bin 128's re weights by fold slot (unscaled fp32: that bin is a vector dot product in the fold)
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

// GATE ROW ORDER of GX, the bias and both W matrices: row 16 (u / 4) + 4 (u % 4) + g for gate g (i, f, g, o) of unit u (permuted once, on the host: engine_weights.hip
// build_weights_v5) -- an MFMA row tile = four units x four gates, the four gates of a unit = the four accumulator registers of one lane.  Wave w owns row tiles
// 4 w .. 4 w + 3 = units 16 w .. 16 w + 15; lane (column = stream, quad q) of tile mi holds unit 16 w + 4 mi + q.
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
   // recurrent weights: row tile 4 wave + mi, rows + (lane & 15), k = 4 kk + quad
   float a[4][32];
#pragma unroll
   for (int mi = 0; mi < 4; ++mi) {
      const float *row = w.whh + (size_t)(16 * (4 * wave + mi) + (lane & 15)) * kV5Hidden + quad;
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) a[mi][kk] = row[4 * kk];
   }
   float c[4], dw[4];
#pragma unroll
   for (int mi = 0; mi < 4; ++mi) {
      const int u = 16 * wave + 4 * mi + quad;
      dw[mi] = w.dec_w[u];
      c[mi] = cs[(size_t)s_col * kV5Hidden + u];
      hb[0][u * 16 + col] = hs[(size_t)s_col * kV5Hidden + u];
   }
   const float *gx_lane = gx + (size_t)s_col * n_chunks * kV5Gates + 64 * wave + 4 * quad;
   float4 gn[4];
#pragma unroll
   for (int mi = 0; mi < 4; ++mi) gn[mi] = *reinterpret_cast<const float4 *>(gx_lane + 16 * mi);
   __syncthreads();
   int par = 0;
   float hlast[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
   for (int ch = 0; ch < n_chunks; ++ch) {
      f4v5 acc[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[mi] = (f4v5){gn[mi].x, gn[mi].y, gn[mi].z, gn[mi].w};
      if (ch + 1 < n_chunks) {
#pragma unroll
         for (int mi = 0; mi < 4; ++mi) gn[mi] = *reinterpret_cast<const float4 *>(gx_lane + (size_t)(ch + 1) * kV5Gates + 16 * mi);
      }
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
         const float bv = hb[par][(4 * kk + quad) * 16 + col];
#pragma unroll
         for (int mi = 0; mi < 4; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][kk], bv, acc[mi], 0, 0, 0);
      }
      float d = 0.0f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
         const float ig = v5_sigmoid(acc[mi][0]), fg = v5_sigmoid(acc[mi][1]), gg = v5_tanh(acc[mi][2]), og = v5_sigmoid(acc[mi][3]);
         c[mi] = fmaf(fg, c[mi], ig * gg);
         const float hn = og * v5_tanh(c[mi]);
         hlast[mi] = hn;
         hb[par ^ 1][(16 * wave + 4 * mi + quad) * 16 + col] = hn;
         d = fmaf(dw[mi], fmaxf(hn, 0.0f), d);                        // decoder: ReLU -> conv 128 -> 1 (silero_vad.py:335-338)
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
      for (int mi = 0; mi < 4; ++mi) {
         const int u = 16 * wave + 4 * mi + quad;
         cs[(size_t)s_col * kV5Hidden + u] = c[mi];
         if (n_chunks > 0) hs[(size_t)s_col * kV5Hidden + u] = hlast[mi];
      }
   }
}

// The same recurrence with W_hh h as split-fp16 MFMAs (W h ~ Wl hh + Wh hl + Wh hh, fp32 accumulation: the form the v3.1 / v4 LSTM kernels use,
// kernels_lstm.hip): 48 v_mfma_f32_16x16x32_f16 per wave and slot instead of 128 fp32 16x16x4.  h travels through LDS as [stream][unit] halves (hi and lo tiles,
// pitch 144: conflict-free fragment reads): a lane's B fragment of a k-block is one 16-byte read per tile.
// Round 6: a slot was 2.0 us, of which the 48 MFMAs are 0.35 us of one wave and the gates -- ten transcendentals per cell, four cells per lane -- about as much
// again, one AFTER the other: with rows ordered [gate][unit] a lane's cells needed all four row tiles.  With the row order above a row tile is COMPLETE cells: the
// gates of tile mi run under the twelve MFMAs of tile mi + 1 (the MFMAs are issued tile by tile; hipcc schedules the independent vector work between them).
#ifndef VADC_V5_LSTM_WAVES
#define VADC_V5_LSTM_WAVES 16
#endif
typedef _Float16 v5h8 __attribute__((ext_vector_type(8)));
typedef _Float16 v5h4 __attribute__((ext_vector_type(4)));
constexpr int kV5HP = kV5Hidden + 16;                // halves per stream row: 18 slots of 16 bytes -- conflict-free B-fragment reads (136 was 2-way conflicted: tools/lds_frag_probe.hip)

// NW waves per workgroup (16: four per SIMD), wave w owns row tiles (32 / NW) w .. : with 8 waves a SIMD's two waves met at every slot's barrier with nothing to
// cover the dependent MFMA chains (1.9 -> 1.7 us per slot from the row order alone); 16 waves of half the work each hide them.
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_v5_lstm_h3(const float *__restrict__ gx,       // [S * C][512]
                                                           V5Weights w,
                                                           float *__restrict__ hs, float *__restrict__ cs,  // [S][128]
                                                           float *__restrict__ probs,                       // [S][C][2]
                                                           int n_streams, int n_chunks)
{
   constexpr int MW = 32 / NW;                          // row tiles (= groups of four units) per wave
   __shared__ __attribute__((aligned(16))) _Float16 hh[2][16 * kV5HP], hl[2][16 * kV5HP];   // [parity][stream][unit]: hi / lo halves of h
   __shared__ float pd[2][NW][16];
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int col = lane & 15, quad = lane >> 4;
   const int s0 = blockIdx.x * 16;
   const int s_col = min(s0 + col, n_streams - 1);
   const bool col_ok = s0 + col < n_streams;
   // recurrent weights of this wave's 4 MW units: row tile MW wave + mi; k-block kb: lane holds k = 32 kb + 8 quad + e
   v5h8 ah[MW][4], al[MW][4];
#pragma unroll
   for (int mi = 0; mi < MW; ++mi)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         const v5h8 *pp = reinterpret_cast<const v5h8 *>(w.whh_h + (((size_t)(MW * wave + mi) * 4 + kb) * 64 + lane) * 16);
         ah[mi][kb] = pp[0]; al[mi][kb] = pp[1];
      }
   float c[MW], dw[MW], hlast[MW];
#pragma unroll
   for (int mi = 0; mi < MW; ++mi) {
      const int u = 4 * (MW * wave + mi) + quad;
      dw[mi] = w.dec_w[u];
      c[mi] = cs[(size_t)s_col * kV5Hidden + u];
      hlast[mi] = hs[(size_t)s_col * kV5Hidden + u];
      const _Float16 hi = (_Float16)hlast[mi];
      hh[0][col * kV5HP + u] = hi;
      hl[0][col * kV5HP + u] = (_Float16)(hlast[mi] - (float)hi);
   }
   const float *gx_lane = gx + (size_t)s_col * n_chunks * kV5Gates + 16 * MW * wave + 4 * quad;
   float4 gn[MW];
#pragma unroll
   for (int mi = 0; mi < MW; ++mi) gn[mi] = *reinterpret_cast<const float4 *>(gx_lane + 16 * mi);
   __syncthreads();
   int par = 0;
#pragma unroll 1
   for (int ch = 0; ch < n_chunks; ++ch) {
      f4v5 acc[MW];
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) acc[mi] = (f4v5){gn[mi].x, gn[mi].y, gn[mi].z, gn[mi].w};
      if (ch + 1 < n_chunks) {
#pragma unroll
         for (int mi = 0; mi < MW; ++mi) gn[mi] = *reinterpret_cast<const float4 *>(gx_lane + (size_t)(ch + 1) * kV5Gates + 16 * mi);
      }
      v5h8 bh[4], bl[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         bh[kb] = *reinterpret_cast<const v5h8 *>(&hh[par][col * kV5HP + 32 * kb + 8 * quad]);
         bl[kb] = *reinterpret_cast<const v5h8 *>(&hl[par][col * kV5HP + 32 * kb + 8 * quad]);
      }
      float d = 0.0f;
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) {
#pragma unroll
         for (int kb = 0; kb < 4; ++kb) {
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi][kb], bh[kb], acc[mi], 0, 0, 0);
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi][kb], bl[kb], acc[mi], 0, 0, 0);
            acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi][kb], bh[kb], acc[mi], 0, 0, 0);
         }
      }
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) {
         const float ig = v5_sigmoid(acc[mi][0]), fg = v5_sigmoid(acc[mi][1]), gg = v5_tanh(acc[mi][2]), og = v5_sigmoid(acc[mi][3]);
         c[mi] = fmaf(fg, c[mi], ig * gg);
         float hn = og * v5_tanh(c[mi]);
         asm volatile("" : "+v"(hn));                                 // the ROUNDED h is what is split (and what a later call re-splits from the state):
         hlast[mi] = hn;                                              // without this the product is contracted into the subtraction below
         const _Float16 hi = (_Float16)hn;
         const int u = 4 * (MW * wave + mi) + quad;
         hh[par ^ 1][col * kV5HP + u] = hi;
         hl[par ^ 1][col * kV5HP + u] = (_Float16)(hn - (float)hi);
         d = fmaf(dw[mi], fmaxf(hn, 0.0f), d);                        // decoder: ReLU -> conv 128 -> 1 (silero_vad.py:335-338)
      }
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      if (quad == 0) pd[ch & 1][wave][col] = d;
      __syncthreads();
      par ^= 1;
      if (wave == 0 && lane < 16 && s0 + lane < n_streams) {
         const float *p = &pd[ch & 1][0][lane];
         float m = 0.0f;
#pragma unroll
         for (int g8 = 0; g8 < NW; g8 += 8)                            // (a fixed tree: the probabilities do not depend on timing)
            m += ((p[16 * g8] + p[16 * g8 + 16]) + (p[16 * g8 + 32] + p[16 * g8 + 48])) + ((p[16 * g8 + 64] + p[16 * g8 + 80]) + (p[16 * g8 + 96] + p[16 * g8 + 112]));
         m += w.dec_b[0];
         const float pr = 1.0f / (1.0f + expf(-m));                   // sigmoid; the mean over the one step is the value itself (:412)
         probs[((size_t)(s0 + lane) * n_chunks + ch) * 2 + 0] = pr;
         probs[((size_t)(s0 + lane) * n_chunks + ch) * 2 + 1] = pr;
      }
   }
   if (col_ok) {
#pragma unroll
      for (int mi = 0; mi < MW; ++mi) {
         const int u = 4 * (MW * wave + mi) + quad;
         cs[(size_t)s_col * kV5Hidden + u] = c[mi];
         hs[(size_t)s_col * kV5Hidden + u] = hlast[mi];               // fp32: the state a later call (or the caller) sees is not rounded to halves
      }
   }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// k_v5_encoder_h3 (round 6): the same stages on the fp16 matrix pipe at fp32 accuracy -- every contraction as three v_mfma_f32_16x16x32_f16 on split operands
// (a = hi + lo: Al.Bh + Ah.Bl + Ah.Bh, fp32 accumulation; the form the v3.1 / v4 kernels use), 16 chunks per 512-thread workgroup, TWO workgroups per CU (<= 80 KB
// of LDS and <= 128 registers for s16 input), so that one workgroup's vector / LDS phases run under the other's matrix phases.
//  * STFT: REAL-INPUT FOLD as in the v4 front end (kernels_frontend_gemm2.hip): slot j = 0 .. 127 pairs taps j + 1 and 255 - j, s_j = x[j + 1] + x[255 - j] feeds the
//    re rows, d_j = x[j + 1] - x[255 - j] the im rows (slot 127 pairs the centre tap with itself: re weight halved, im weight zero; tap 0's weight is zero) -- K = 128
//    instead of 256.  Samples stay INTEGERS (s16 as they come, in LDS as s16; f32 x 32768 as fp32), so a fold value of s16 input is a 17-bit integer: hi = round
//    toward zero, lo = the exact rest.  Wave w owns the re AND the im rows of bins 16 w .. 16 w + 15 (its A fragments stay in 64 registers over the four column
//    passes of 16 columns): the magnitude is formed in registers.  Bin 128 (im row identically zero) is a 128-term fp32 dot product per column, done by the fold.
//  * Activations live in LDS PIECE-MAJOR: plane [hi | lo][k quarter kq][channel block kb][column][8 halves], channel = 32 kb + 8 kq + e -- the 16-byte B fragment of
//    lane (column lc, quarter kq) of a k-block; the 16 lanes of a ds_read_b128 group (two quarters x 8 columns each, MI355X_MICROARCH.md LDS table) then read 16
//    consecutive 16-byte pieces of one or two piece planes whose distance is a multiple of 256 B: conflict-free.  ([column][channel] with pitch 136 -- the LSTM
//    kernels' h tile -- is 2-way conflicted in every group: 48 % of this kernel's LDS cycles in its first form.)  Magnitudes are kept x 256 (bounded by 2^15:
//    |x| <= 1 and the window sums to 128), conv outputs unscaled.
//  * A k = 3 conv is a GEMM with K = (tap, channel) whose B fragment is read with the tap's column shift (zero outside the chunk); the weights' fragments stream from
//    L2 once per workgroup (39 KB per chunk) through a rolling window of four k-blocks of registers.
// Scales are powers of two: A x 256 everywhere, magnitudes x 256 => one fma(acc, 2^-8 or 2^-16, bias) in each epilogue.
// ------------------------------------------------------------------------------------------------------------------------------------------------
typedef _Float16 v5h2 __attribute__((ext_vector_type(2)));
typedef float v5f2 __attribute__((ext_vector_type(2)));
constexpr int kH3Chunks = 16;                        // chunks per workgroup
constexpr int kH3Pre = 4;                            // k-blocks of A fragments in flight ahead of their MFMAs
constexpr int kV5X3Plane = 4 * 4 * 16 * 8;           // halves per plane of a tile of conv 3's output (piece-major, 4 channel blocks x 16 columns)

struct V5Frag { v5h8 hi, lo; };
__device__ __forceinline__ V5Frag v5_afrag(const _Float16 *base, int mt, int kbs, int kb, int lane)
{
   const v5h8 *p = reinterpret_cast<const v5h8 *>(base + ((size_t)(mt * kbs + kb) * 2) * 512 + lane * 8);
   return V5Frag{p[0], p[64]};
}
__device__ __forceinline__ f4v5 v5_mfma3(const V5Frag &a, const v5h8 &bh, const v5h8 &bl, f4v5 acc)
{
   acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, bh, acc, 0, 0, 0);
   acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, bl, acc, 0, 0, 0);
   return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, bh, acc, 0, 0, 0);
}
// two values -> packed (hi, lo) halves: hi = round toward zero (never overflows), lo = v - hi
__device__ __forceinline__ void v5_split2(float a, float b, v5h2 &hi, v5h2 &lo)
{
   hi = __builtin_bit_cast(v5h2, __builtin_amdgcn_cvt_pkrtz(a, b));
   const v5f2 r = {a - (float)hi[0], b - (float)hi[1]};
   lo = __builtin_convertvector(r, v5h2);
}
__device__ __forceinline__ void v5_split4_store(const float (&v)[4], _Float16 *hi_p, _Float16 *lo_p)
{
   v5h2 h0, l0, h1, l1;
   v5_split2(v[0], v[1], h0, l0);
   v5_split2(v[2], v[3], h1, l1);
   *reinterpret_cast<v5h4 *>(hi_p) = (v5h4){h0[0], h0[1], h1[0], h1[1]};
   *reinterpret_cast<v5h4 *>(lo_p) = (v5h4){l0[0], l0[1], l1[0], l1[1]};
}
// piece-major activation plane of CBS channel blocks x COLS columns: halves offset of the 8-half piece (quarter kq, block kb, column col)
template <int CBS, int COLS> __device__ __forceinline__ int v5_piece(int kq, int kb, int col) { return ((kq * CBS + kb) * COLS + col) * 8; }
// The fold planes are WRITTEN by lanes that differ in (kq, kb) at one column -- 8 pieces a multiple of 256 B apart: 8-way conflicted as they stand (43 % of the
// kernel's LDS cycles) -- so their column index is XOR-swizzled by a value from {0..3, 12..15} chosen by (kq, kb & 1): distinct mod 8 (the 8 pieces of a
// ds_write_b64 group land on 8 different bank quads), and bits 3 and 2 equal, which keeps a column inside its half of a ds_read_b128 lane group
// ({0-3, 12-15} / {4-11}) -- the MFMA's fragment reads stay conflict-free.
__device__ __forceinline__ int v5_bf_piece(int kq, int kb, int col)
{
   const int i = kq + 4 * (kb & 1);
   return ((kq * 4 + kb) * 16 + (col ^ (i < 4 ? i : i + 8))) * 8;
}
// ... and of the four output channels 16 mt + 4 q4 .. + 3 of an accumulator (m-tile mt, accumulator quad q4) at column col
template <int CBS, int COLS> __device__ __forceinline__ int v5_out4(int mt, int q4, int col) { return v5_piece<CBS, COLS>((2 * mt + (q4 >> 1)) & 3, mt >> 1, col) + 4 * (q4 & 1); }

// One conv stage (or the LSTM input projection) for ONE m-tile of 16 output channels and NN n-tiles of 16 columns starting at n-tile nt0.
//   in: planes [hi | lo] piece-major, CB channel blocks x ICOLS columns; K = NT taps (the first is tap TAP0) x CB blocks of 32 channels, + (EXTRA) one block carrying
//       channel 128 (x128: [hi | lo][ICOLS] halves) of tap (lane >> 4)
//   OUTF = false: bias, ReLU, split, planes [hi | lo] piece-major OCB x OCOLS;  OUTF = true: bias, fp32 rows of gx (the LSTM input projection)
template <int CB, int NT, int TAP0, int TIN, int TOUT, int STRIDE, int NN, int OCB, bool EXTRA, bool OUTF>
__device__ __forceinline__ void v5_conv_h3(const _Float16 *__restrict__ in, const _Float16 *__restrict__ x128, _Float16 *__restrict__ out,
                                           const _Float16 *__restrict__ wfrag, const float *__restrict__ bias, float scale, int mt, int nt0, int lane,
                                           float *__restrict__ gx, int item0, int n_items)
{
   constexpr int KB = NT * CB + (EXTRA ? 1 : 0);
   constexpr int ICOLS = kH3Chunks * TIN, OCOLS = kH3Chunks * TOUT;
   constexpr int in_plane = 4 * CB * ICOLS * 8, out_plane = 4 * OCB * OCOLS * 8;
   constexpr int PRE = KB < kH3Pre ? KB : kH3Pre;
   const int lc = lane & 15, kq = lane >> 4;
   V5Frag a[KB];                                        // (fully unrolled below: at most PRE of them are live at a time)
#pragma unroll
   for (int kb = 0; kb < PRE; ++kb) a[kb] = v5_afrag(wfrag, mt, KB, kb, lane);
   const float4 b4 = *reinterpret_cast<const float4 *>(bias + 16 * mt + 4 * kq);
   f4v5 acc[NN];
   int icol[NN][NT];                                    // input column of this lane's output column per tap, -1 = padding
#pragma unroll
   for (int ni = 0; ni < NN; ++ni) {
      acc[ni] = (f4v5){0.0f, 0.0f, 0.0f, 0.0f};
      const int col = 16 * (nt0 + ni) + lc, j = col / TOUT, t = col - j * TOUT;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) { const int q = t * STRIDE + TAP0 + ti - 1; icol[ni][ti] = (q >= 0 && q < TIN) ? j * TIN + q : -1; }
   }
   const v5h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
   for (int kb = 0; kb < KB; ++kb) {
      if (kb < NT * CB) {
         const int ti = kb / CB, cb = kb - ti * CB;
#pragma unroll
         for (int ni = 0; ni < NN; ++ni) {
            const int ic = icol[ni][ti];
            const _Float16 *src = in + v5_piece<CB, ICOLS>(kq, cb, ic < 0 ? 0 : ic);
            v5h8 bh = *reinterpret_cast<const v5h8 *>(src), bl = *reinterpret_cast<const v5h8 *>(src + in_plane);
            if (ic < 0) { bh = zero; bl = zero; }
            acc[ni] = v5_mfma3(a[kb], bh, bl, acc[ni]);
         }
      } else {                                          // channel 128 (bin 128): k slot 8 tap + 0 = this lane's tap kq (kq = 3: nothing)
#pragma unroll
         for (int ni = 0; ni < NN; ++ni) {
            const int col = 16 * (nt0 + ni) + lc, j = col / TOUT, t = col - j * TOUT, q = t * STRIDE + kq - 1;
            const bool ok = kq < 3 && q >= 0 && q < TIN;
            const int ic = ok ? j * TIN + q : 0;
            v5h8 bh = zero, bl = zero;
            bh[0] = ok ? x128[ic] : (_Float16)0; bl[0] = ok ? x128[ICOLS + ic] : (_Float16)0;
            acc[ni] = v5_mfma3(a[kb], bh, bl, acc[ni]);
         }
      }
      if (kb + PRE < KB) a[kb + PRE] = v5_afrag(wfrag, mt, KB, kb + PRE, lane);
   }
#pragma unroll
   for (int ni = 0; ni < NN; ++ni) {
      const int col = 16 * (nt0 + ni) + lc;
      float v[4] = {fmaf(acc[ni][0], scale, b4.x), fmaf(acc[ni][1], scale, b4.y), fmaf(acc[ni][2], scale, b4.z), fmaf(acc[ni][3], scale, b4.w)};
      if (OUTF) {
         if (item0 + col < n_items) *reinterpret_cast<float4 *>(gx + (size_t)(item0 + col) * kV5Gates + 16 * mt + 4 * kq) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
         for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
         _Float16 *o = out + v5_out4<OCB, OCOLS>(mt, kq, col);
         v5_split4_store(v, o, o + out_plane);
      }
   }
}

template <typename T> struct V5X;                       // how the staged samples are kept in LDS
template <> struct V5X<int16_t> { typedef int16_t type; };
template <> struct V5X<float> { typedef float type; };

template <typename T>
__global__ __launch_bounds__(512, 2) void k_v5_encoder_h3(const T *__restrict__ pcm,       // [S][C][512]
                                                          const float *__restrict__ ctx,   // [S][64]
                                                          V5Weights w,
                                                          _Float16 *__restrict__ x3,       // [tiles of 16 chunks][hi | lo] piece-major 4 x 16: conv 3's output, k_v5_wih's B operand
                                                          int n_items, int n_chunks)
{
   typedef typename V5X<T>::type XT;
   // region A: staged samples (integers) + the fold planes of one column pass; later conv 0's output.  region B: magnitudes; later conv 1 / 2 / 3's outputs
   constexpr int kXBytes = kH3Chunks * 640 * (int)sizeof(XT);            // 20,480 (s16) / 40,960 (f32)
   constexpr int kBFPlane = 4 * 4 * 16 * 8;                              // halves per fold plane: piece-major, 4 k-blocks x 16 columns
   constexpr int kMGPlane = 4 * 4 * 64 * 8;                              // halves per magnitude / conv 0 plane: 4 channel blocks x 64 columns
   constexpr int kABytes = (kXBytes + 4 * kBFPlane * 2) > 2 * kMGPlane * 2 ? (kXBytes + 4 * kBFPlane * 2) : 2 * kMGPlane * 2;      // 36,864 (s16) / 57,344 (f32)
   constexpr int kBBytes = 2 * kMGPlane * 2;                             // 32,768
   constexpr int kC1Plane = 4 * 2 * 32 * 8, kC2Plane = 4 * 2 * 16 * 8;
   static_assert(2 * (kC1Plane + kC2Plane) * 2 <= kBBytes, "conv 1-2 outputs fit the magnitude region");
   __shared__ __attribute__((aligned(16))) unsigned char RA[kABytes];
   __shared__ __attribute__((aligned(16))) unsigned char RB[kBBytes];
   __shared__ __attribute__((aligned(16))) _Float16 M128[2][64];        // bin 128's magnitude x 256 by column: [hi | lo]
   __shared__ float NYs[16];
   XT *X = reinterpret_cast<XT *>(RA);                                  // [16 chunks][640]
   _Float16 *BF = reinterpret_cast<_Float16 *>(RA + kXBytes);           // [s hi, s lo, d hi, d lo] piece-major 4 x 16
   _Float16 *C0 = reinterpret_cast<_Float16 *>(RA);                     // [hi | lo] piece-major 4 x 64
   _Float16 *MG = reinterpret_cast<_Float16 *>(RB);                     // [hi | lo] piece-major 4 x 64 (bins 0..127)
   _Float16 *C1 = reinterpret_cast<_Float16 *>(RB);                     // [hi | lo] piece-major 2 x 32
   _Float16 *C2 = C1 + 2 * kC1Plane;                                    // [hi | lo] piece-major 2 x 16
   const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int lc = lane & 15, kq = lane >> 4;
   const int item0 = blockIdx.x * kH3Chunks;

   // this wave's STFT A fragments: re tile `wave`, im tile 8 + `wave`, 4 k-blocks each: 64 registers, used by all four column passes
   V5Frag are[4], aim[4];
#pragma unroll
   for (int kb = 0; kb < 4; ++kb) { are[kb] = v5_afrag(w.h_stft, wave, 4, kb, lane); aim[kb] = v5_afrag(w.h_stft, 8 + wave, 4, kb, lane); }

   // ---- stage: [context 64 | window 512 | reflect 64] per chunk as integers (s16 as they are, f32 x 32768) ----
   constexpr float kToInt = sizeof(T) == 2 ? 1.0f : 32768.0f;
   for (int i = tid; i < kH3Chunks * 80; i += 512) {                     // 8-sample pieces: 8 of context, 64 of window, 8 of the reflected tail
      const int j = i / 80, pc = i - j * 80;
      const int item = min(item0 + j, n_items - 1);
      const int s = item / n_chunks, c = item - s * n_chunks;
      float v[8];
      if (pc >= 8 && pc < 72) {
         const T *src = pcm + (size_t)item * kV5Window + 8 * (pc - 8);
#pragma unroll
         for (int e = 0; e < 8; ++e) v[e] = (float)src[e] * kToInt;
      } else if (pc < 8) {
         if (c > 0) {                                                    // the previous window's tail (vadc.c:132-135)
            const T *src = pcm + (size_t)(item - 1) * kV5Window + (kV5Window - kV5Context) + 8 * pc;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)src[e] * kToInt;
         } else {                                                        // carried from the previous call (vadc.c:124): kept / 32768 for either type
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ctx[(size_t)s * kV5Context + 8 * pc + e] * 32768.0f;
         }
      } else {                                                           // F.pad(input, (0, 64), "reflect"): padded[576 + k] = input[574 - k] = window[510 - k]
         const T *src = pcm + (size_t)item * kV5Window + 510 - 8 * (pc - 72);
#pragma unroll
         for (int e = 0; e < 8; ++e) v[e] = (float)src[-e] * kToInt;
      }
      XT *dst = X + j * 640 + 8 * pc;
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = (XT)v[e];
   }
   __syncthreads();

   // the fold: thread = (column of the pass fcol, slot group fq: slots 8 fq .. 8 fq + 7, half fh: its slots 4 fh .. 4 fh + 3); a column's 32 threads are half a wave
   const int fcol = tid >> 5, fq = (tid >> 1) & 15, fh = tid & 1;
   float wny[4];
#pragma unroll
   for (int e = 0; e < 4; ++e) wny[e] = w.wny[8 * fq + 4 * fh + e];
#pragma unroll 1
   for (int pass = 0; pass < 4; ++pass) {
      {
         const int gcol = 16 * pass + fcol, j = gcol >> 2, fr = gcol & 3;
         const XT *x = X + j * 640 + 128 * fr;
         const int d0 = 8 * fq + 4 * fh;                                  // direct taps x[d0 + 1 + e], mirrored taps x[255 - d0 - e]
         typedef XT xt4 __attribute__((ext_vector_type(4)));
         const xt4 dq = *reinterpret_cast<const xt4 *>(x + d0), mq = *reinterpret_cast<const xt4 *>(x + 252 - d0);      // x[d0 .. d0 + 3], x[252 - d0 .. 255 - d0]: aligned
         const XT d4 = x[d0 + 4];
         const float dv[4] = {(float)dq[1], (float)dq[2], (float)dq[3], (float)d4}, mv[4] = {(float)mq[3], (float)mq[2], (float)mq[1], (float)mq[0]};
         float sv[4], dd[4];
#pragma unroll
         for (int e = 0; e < 4; ++e) { sv[e] = dv[e] + mv[e]; dd[e] = dv[e] - mv[e]; }
         _Float16 *bp = BF + v5_bf_piece(fq & 3, fq >> 2, fcol) + 4 * fh;
         v5_split4_store(sv, bp, bp + kBFPlane);
         v5_split4_store(dd, bp + 2 * kBFPlane, bp + 3 * kBFPlane);
         float ny = wny[0] * sv[0];
#pragma unroll
         for (int e = 1; e < 4; ++e) ny = fmaf(wny[e], sv[e], ny);
         ny += __shfl_xor(ny, 1); ny += __shfl_xor(ny, 2); ny += __shfl_xor(ny, 4); ny += __shfl_xor(ny, 8); ny += __shfl_xor(ny, 16);
         if ((tid & 31) == 0) NYs[fcol] = ny;                            // 2^15 x re of bin 128
      }
      __syncthreads();
      {
         f4v5 ar = (f4v5){0.0f, 0.0f, 0.0f, 0.0f}, ai = ar;
#pragma unroll
         for (int kb = 0; kb < 4; ++kb) {
            const _Float16 *bp = BF + v5_bf_piece(kq, kb, lc);
            const v5h8 s_h = *reinterpret_cast<const v5h8 *>(bp), s_l = *reinterpret_cast<const v5h8 *>(bp + kBFPlane);
            const v5h8 d_h = *reinterpret_cast<const v5h8 *>(bp + 2 * kBFPlane), d_l = *reinterpret_cast<const v5h8 *>(bp + 3 * kBFPlane);
            ar = v5_mfma3(are[kb], s_h, s_l, ar);
            ai = v5_mfma3(aim[kb], d_h, d_l, ai);
         }
         // accumulators = 2^23 x (re, im); magnitudes are kept x 2^8: sqrt(.) x 2^-15
         float m[4];
#pragma unroll
         for (int r = 0; r < 4; ++r) m[r] = __builtin_amdgcn_sqrtf(fmaf(ar[r], ar[r], ai[r] * ai[r])) * 3.0517578125e-05f;
         _Float16 *o = MG + v5_out4<4, 64>(wave, kq, 16 * pass + lc);
         v5_split4_store(m, o, o + kMGPlane);
         if (tid < 16) {                                                  // bin 128: 2^8 x |re| = |NYs| x 2^-7
            v5h2 hi, lo;
            v5_split2(fabsf(NYs[tid]) * 0.0078125f, 0.0f, hi, lo);
            M128[0][16 * pass + tid] = hi[0]; M128[1][16 * pass + tid] = lo[0];
         }
      }
      __syncthreads();
   }
   // conv 0: [129, 4] -> [128, 4]: wave = m-tile, four n-tiles (64 columns); A x 256, magnitudes x 256 => 2^-16
   v5_conv_h3<4, 3, 0, 4, 4, 1, 4, 4, true, false>(MG, &M128[0][0], C0, w.h_conv[0], w.conv_b[0], 1.52587890625e-05f, wave, 0, lane, nullptr, 0, 0);
   __syncthreads();
   // conv 1: [128, 4] -> [64, 2], stride 2: wave = (m-tile wave & 3, n-tile wave >> 2)
   v5_conv_h3<4, 3, 0, 4, 2, 2, 1, 2, false, false>(C0, nullptr, C1, w.h_conv[1], w.conv_b[1], 0.00390625f, wave & 3, wave >> 2, lane, nullptr, 0, 0);
   __syncthreads();
   // conv 2: [64, 2] -> [64, 1], stride 2: taps 1 and 2 (tap 0 reads step -1: padding); waves 0-3
   if (wave < 4) v5_conv_h3<2, 2, 1, 2, 1, 2, 1, 2, false, false>(C1, nullptr, C2, w.h_conv[2], w.conv_b[2], 0.00390625f, wave, 0, lane, nullptr, 0, 0);
   __syncthreads();
   // conv 3: [64, 1] -> [128, 1]: tap 1 only.  Its output leaves the kernel (8 KB per workgroup, in the fragment order of the next GEMM): the LSTM's input projection
   // W_ih [512 x 128] has 16 columns per workgroup here -- every one of its 262 KB of fragments would be fetched from L2 for ONE MFMA triple, 42 % of this kernel's
   // weight stream, and the kernel is bound by that stream (55 of the ~70 GB/s a CU gets from L2) -- so it runs as k_v5_wih, with W_ih resident in registers.
   v5_conv_h3<2, 1, 1, 1, 1, 1, 1, 4, false, false>(C2, nullptr, x3 + (size_t)blockIdx.x * (2 * kV5X3Plane), w.h_conv[3], w.conv_b[3], 0.00390625f, wave, 0, lane, nullptr, 0, 0);
}

// LSTM input projection: GX[item][512] = W_ih [512 x 128] . c3 [128 x items] + (b_ih + b_hh), persistent: a workgroup keeps ALL of W_ih as split-fp16 A fragments in
// registers (wave w: m-tiles 4 w .. 4 w + 3, 128 registers) and walks over tiles of 16 chunks; the B fragments are 16-byte pieces of k_v5_encoder_h3's output, one
// coalesced load per lane and plane, fetched one tile ahead.
__global__ __launch_bounds__(512, 1) void k_v5_wih(const _Float16 *__restrict__ x3, V5Weights w, float *__restrict__ gx, int n_items)
{
   const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
   const int lc = lane & 15, kq = lane >> 4;
   const int n_tiles = (n_items + kH3Chunks - 1) / kH3Chunks;
   V5Frag a[4][4];
#pragma unroll
   for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) a[mi][kb] = v5_afrag(w.h_wih, 4 * wave + mi, 4, kb, lane);
   float4 b4[4];
#pragma unroll
   for (int mi = 0; mi < 4; ++mi) b4[mi] = *reinterpret_cast<const float4 *>(w.lstm_b + 16 * (4 * wave + mi) + 4 * kq);
   auto load = [&](int t, v5h8 (&bh)[4], v5h8 (&bl)[4]) {
      const _Float16 *p = x3 + (size_t)min(t, n_tiles - 1) * (2 * kV5X3Plane);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         bh[kb] = *reinterpret_cast<const v5h8 *>(p + v5_piece<4, 16>(kq, kb, lc));
         bl[kb] = *reinterpret_cast<const v5h8 *>(p + kV5X3Plane + v5_piece<4, 16>(kq, kb, lc));
      }
   };
   v5h8 bh[4], bl[4], nh[4], nl[4];
   int t = blockIdx.x;
   if (t >= n_tiles) return;
   load(t, bh, bl);
#pragma unroll 1
   for (; t < n_tiles; t += gridDim.x) {
      load(t + (int)gridDim.x, nh, nl);
      const int item = t * kH3Chunks + lc;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
         f4v5 acc = (f4v5){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
         for (int kb = 0; kb < 4; ++kb) acc = v5_mfma3(a[mi][kb], bh[kb], bl[kb], acc);
         if (item < n_items)
            *reinterpret_cast<float4 *>(gx + (size_t)item * kV5Gates + 16 * (4 * wave + mi) + 4 * kq) =
               make_float4(fmaf(acc[0], 0.00390625f, b4[mi].x), fmaf(acc[1], 0.00390625f, b4[mi].y), fmaf(acc[2], 0.00390625f, b4[mi].z), fmaf(acc[3], 0.00390625f, b4[mi].w));
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) { bh[kb] = nh[kb]; bl[kb] = nl[kb]; }
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
static void launch_v5_enc_t(const T *pcm, float *ctx, const V5Weights &w, float *gx, float *x3, int n_streams, int n_chunks, bool fp32, int n_cus, hipStream_t st)
{
   const int n_items = n_streams * n_chunks;
   if (!fp32 && w.h_stft) {
      const int tiles = (n_items + kH3Chunks - 1) / kH3Chunks;
      hipLaunchKernelGGL((k_v5_encoder_h3<T>), dim3(tiles), dim3(512), 0, st, pcm, ctx, w, reinterpret_cast<_Float16 *>(x3), n_items, n_chunks);
      hipLaunchKernelGGL(k_v5_wih, dim3(tiles < n_cus ? tiles : n_cus), dim3(512), 0, st, reinterpret_cast<const _Float16 *>(x3), w, gx, n_items);
   } else                   hipLaunchKernelGGL((k_v5_encoder<T>), dim3((n_items + kV5Chunks - 1) / kV5Chunks), dim3(256), 0, st, pcm, ctx, w, gx, n_items, n_chunks);
   hipLaunchKernelGGL((k_v5_context<T>), dim3((n_streams * kV5Context + 255) / 256), dim3(256), 0, st, pcm, ctx, n_streams, n_chunks);
}
// fp32 = true (or no split-fp16 operands: a weight outside fp16's range, a basis without the fold symmetries): k_v5_encoder, fp32 MFMA
// x3: the hand-off between k_v5_encoder_h3 and k_v5_wih, 8 KB per 16 chunks (the engine's d_x35)
void launch_v5_encoder_f32(const float *pcm, float *ctx, const V5Weights &w, float *gx, float *x3, int n_streams, int n_chunks, bool fp32, int n_cus, hipStream_t st)
{
   launch_v5_enc_t<float>(pcm, ctx, w, gx, x3, n_streams, n_chunks, fp32, n_cus, st);
}
void launch_v5_encoder_s16(const int16_t *pcm, float *ctx, const V5Weights &w, float *gx, float *x3, int n_streams, int n_chunks, bool fp32, int n_cus, hipStream_t st)
{
   launch_v5_enc_t<int16_t>(pcm, ctx, w, gx, x3, n_streams, n_chunks, fp32, n_cus, st);
}
// back half: the recurrence + decoder over the call's chunks.  fp32 = true (or no split-fp16 weights): W_hh h as fp32 MFMAs
void launch_v5_lstm(const V5Weights &w, const float *gx, float *hs, float *cs, float *probs, int n_streams, int n_chunks, bool fp32, hipStream_t st)
{
   if (fp32 || !w.whh_h) hipLaunchKernelGGL(k_v5_lstm, dim3((n_streams + 15) / 16), dim3(512), 0, st, gx, w, hs, cs, probs, n_streams, n_chunks);
   else                  hipLaunchKernelGGL(k_v5_lstm_h3<VADC_V5_LSTM_WAVES>, dim3((n_streams + 15) / 16), dim3(64 * VADC_V5_LSTM_WAVES), 0, st, gx, w, hs, cs, probs, n_streams, n_chunks);
}

}  // namespace vadc
