// kernels_encoder_mfma.hip -- one encoder ("transformer") layer per launch on the gfx950 matrix cores.
//
// Same arithmetic as kernels_encoder.hip (the VALU bring-up variant kept for A/B tests), re-mapped so that every
// dense contraction of the layer -- pointwise + projection conv, QKV, attention out-projection, both FFN linears
// and the strided 1x1 conv -- runs on the matrix instructions:
//
//   Y[Cout x cols] = W[Cout x K] . X[K x cols]        cols = (chunk, time-step) of the workgroup's NCH chunks
//
// Replaces (reference file:line): conv_block conv.c:761-814 (dw :17-113, pw/proj :532-589); transformer_block
// transformer.c:13-234 (tensor_linear tensor.h:675-723, softmax :751-784, layer_norm misc.c:143-210); conv k=1 stride s
// + BatchNorm + ReLU transformer.c:279-290 (conv.c:597-709, misc.c:221-258); layer 1 also finishes
// adaptive_audio_normalization_inplace misc.c:65-96.
//
// MAPPING.  A workgroup (4 waves) owns 64 columns; WAVE w owns the 16-column MFMA N-tile [16w,16w+16) for the whole
// layer; accumulators start at the bias; epilogues (ReLU, residual) are applied in the accumulator layout.  LayerNorm
// reduces over channels = over the accumulator registers and the 4 lane-quads of the wave (two shuffles), never across
// waves.  Only the depthwise conv (time neighbours) and attention (all 7..25 steps of a chunk) look across columns, so those
// are the only phases that need workgroup barriers; after the attention a wave works on its own 16 columns alone.
//   * GEMMs of the 32- and 64-channel layers (2-4): split-fp16 form (H3) -- three v_mfma_f32_16x16x32_f16 per k-block on
//     host-split weights, activations in LDS as one pair of fp16 tiles [column][k] reused in place, fp32 accuracy.
//   * GEMMs of the 16-channel first layer (and option "encoder" = 3): v_mfma_f32_16x16x4_f32 on fragment-major fp32 weights
//     (one coalesced 256-byte load per MFMA), activations in LDS as [channel][column] fp32 (pitch 80).
//   * conv block of layers 2-4: the input tile is staged once through LDS (16-byte loads); of the first layer (129 / 258
//     input channels): the K = 1 form -- one lane per column, one input channel per v_mfma_f32_16x16x1_4b_f32, inputs straight
//     from global along the frames (or, option "encoder" = 2, the LDS slab path).
//   * attention: a lane pair per (head, column), Q / V rows padded per chunk for 16-byte reads, fp32 on the vector ALU.
#include "common.h"
#include <algorithm>
#include <cstdio>

namespace vadc {

// The first stage (K = 1 form, Silero v3.1) is compiled for SIX waves per SIMD = six workgroups per CU: 80 registers (the input ring holds 8 channels
// ahead instead of all 33) and 25 KB of LDS.  Measured in the step (tools/l1_occupancy.sh): 5 workgroups / whole input up front 0.225 ms, 6 / ring of 16
// 0.224, 6 / ring of 8 0.218, 6 / ring of 24 0.279 (spills).
#ifndef VADC_L1_XR
#define VADC_L1_XR 8
#endif
#ifndef VADC_L1_WAVES
#define VADC_L1_WAVES 6
#endif

#ifdef VADC_PHASE_PROF
__device__ unsigned long long g_phase[4][16];      // [layer][phase] accumulated cycles of (workgroup 0 mod 64, thread 0)
__device__ unsigned int g_phase_n[4];
#define PH(i) do { if (ph_on) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_phase[ph_layer][i], t_ - ph_t); ph_t = t_; } } while (0)
#else
#define PH(i) do { } while (0)
#endif

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kPitch = 80;    // LDS row pitch in floats (80 % 32 == 16: rows k..k+3 of a B fragment hit disjoint banks)


struct LayerWeightsM {
   const float *dw_w, *dw_b;          // [cin][5], [cin]
   const float *pw_f, *pj_f;          // fragment-major [D/16][cinp/4][64]
   const float *cb_b;                 // [D]   pw bias + proj bias
   const float *qkv_f, *qkv_b;        // [3D/16][D/4][64], [3D]
   const float *out_f, *out_b;
   const float *n1_w, *n1_b;
   const float *l1_f, *l1_b;
   const float *l2_f, *l2_b;
   const float *n2_w, *n2_b;
   const float *cv_f, *cv_b;          // strided conv with BatchNorm folded
   const float *pwj_k1;               // first stage only: [cin][pw 0..15 | proj 0..15] (K = 1 MFMA form)
   const _Float16 *qkv_h, *out_h, *l1_h, *l2_h, *cv_h;   // H3: host-split fp16 fragments [M / 16][K / 32][64 lanes][hi 8 | lo 8] (or null)
   const _Float16 *pw_h, *pj_h;                          // H3, 32 input channels: the conv block's pointwise / projection weights in the same form
};

template <int kFrames>
__device__ __forceinline__ float norm_offset_m(const float *__restrict__ fmp, size_t fm_stride)
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
   return total / (float)kFrames;
}

// full-wave lane shifts (gfx9 DPP wave_shr / wave_shl): lane i <- lane i-1 / lane i+1, 0 shifted in at the ends
__device__ __forceinline__ float dpp_wave_shr1(float v)
{
   return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));   // bound_ctrl: 0 shifted in, no v_mov of the old value
}
__device__ __forceinline__ float dpp_wave_shl1(float v)
{
   return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// acc[mt] (+)= W[16mt.., :] . X[:, wave's 16 columns]   for mt < MT, K = 4*KK rows of X starting at xrow0.
// wf: fragment-major weights [MT][KKW][64] (KKW = k-steps per M-tile in memory), kk0 = first k-step to use.
template <int MT, int KK, int P = kPitch>
__device__ __forceinline__ void gemm_acc(f4v (&acc)[MT], const float *__restrict__ wf, int KKW, int kk0,
                                         const float *X, int lane, int wave)
{
   const int quad = lane >> 4, lc = lane & 15;
   float b[KK];
#pragma unroll
   for (int kk = 0; kk < KK; ++kk) b[kk] = X[(4 * kk + quad) * P + 16 * wave + lc];
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) {
      const float *wp = wf + ((size_t)mt * KKW + kk0) * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[kk * 64], b[kk], acc[mt], 0, 0, 0);
   }
}

// ---- split-fp16 form of the layer GEMMs (H3; the LSTM's trick, kernels_lstm.hip): W . X ~= Wl . Xh + Wh . Xl + Wh . Xh as three
// v_mfma_f32_16x16x32_f16 (K = 32, 16 cycles, fp32 accumulation of exact fp16 x fp16 products) instead of eight
// v_mfma_f32_16x16x4_f32 (32 cycles each, on the vector ALU's lanes): 5.3x fewer matrix cycles at fp32 accuracy (22 significant
// bits per operand).  Activations live in LDS as two fp16 tiles [column][k] (k contiguous, pitch K + 8 halves: a wave's 16-byte
// B-fragment reads and 8-byte accumulator stores are conflict-free within a 16-lane phase); a lane's B fragment of k-block kb is the
// 16 bytes at k = 32 kb + 8 (lane >> 4) of column 16 wave + (lane & 15).  Weights come host-split: [M tile][kb][lane][hi 8 | lo 8].
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

template <int KB>
__device__ __forceinline__ void load_b_h3(h8v (&bh)[KB], h8v (&bl)[KB], const _Float16 *Sh, const _Float16 *Sl, int pitch, int lane, int wave)
{
   const int off = (16 * wave + (lane & 15)) * pitch + 8 * (lane >> 4);
#pragma unroll
   for (int kb = 0; kb < KB; ++kb) {
      bh[kb] = *reinterpret_cast<const h8v *>(Sh + off + 32 * kb);
      bl[kb] = *reinterpret_cast<const h8v *>(Sl + off + 32 * kb);
   }
}

template <int MT, int KB>
__device__ __forceinline__ void gemm_acc_h3(f4v (&acc)[MT], const _Float16 *__restrict__ wf, const h8v (&bh)[KB], const h8v (&bl)[KB], int lane)
{
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
         const h8v *p = reinterpret_cast<const h8v *>(wf + (((size_t)mt * KB + kb) * 64 + lane) * 16);
         const h8v ah = p[0], al = p[1];
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kb], acc[mt], 0, 0, 0);
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kb], acc[mt], 0, 0, 0);
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kb], acc[mt], 0, 0, 0);
      }
}

// accumulator layout -> split tiles: lane (quad, lc), acc[mt][r]  ->  column 16 wave + lc, k = 16 mt + 4 quad + r
template <int MT>
__device__ __forceinline__ void acc_store_h3(const f4v (&acc)[MT], _Float16 *Sh, _Float16 *Sl, int pitch, int lane, int wave)
{
   const int off = (16 * wave + (lane & 15)) * pitch + 4 * (lane >> 4);
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) {
      h4v hi, lo;
#pragma unroll
      for (int r = 0; r < 4; ++r) { hi[r] = (_Float16)acc[mt][r]; lo[r] = (_Float16)(acc[mt][r] - (float)hi[r]); }
      *reinterpret_cast<h4v *>(Sh + off + 16 * mt) = hi;
      *reinterpret_cast<h4v *>(Sl + off + 16 * mt) = lo;
   }
}

template <int MT>
__device__ __forceinline__ void acc_init(f4v (&acc)[MT], const float *__restrict__ bias, int lane)
{
   const int quad = lane >> 4;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) {
      const float4 b4 = *reinterpret_cast<const float4 *>(bias + 16 * mt + 4 * quad);
      acc[mt][0] = b4.x; acc[mt][1] = b4.y; acc[mt][2] = b4.z; acc[mt][3] = b4.w;
   }
}

// accumulator layout <-> LDS [row][col]: lane (quad, lc), reg r  <->  row 16 mt + 4 quad + r, col 16 wave + lc
template <int MT, int P = kPitch>
__device__ __forceinline__ void acc_store(const f4v (&acc)[MT], float *Y, int lane, int wave)
{
   const int quad = lane >> 4, lc = lane & 15;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[(16 * mt + 4 * quad + r) * P + 16 * wave + lc] = acc[mt][r];
}

// LayerNorm over the D = 16*MT channels of each column, in the accumulator layout (misc.c:143-210)
template <int MT>
__device__ __forceinline__ void layer_norm_acc(f4v (&x)[MT], const float *__restrict__ w, const float *__restrict__ b, int lane)
{
   constexpr int D = 16 * MT;
   const int quad = lane >> 4;
   float s = 0.0f;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) s += (x[mt][0] + x[mt][1]) + (x[mt][2] + x[mt][3]);
   s += __shfl_xor(s, 16);
   s += __shfl_xor(s, 32);
   const float mean = s * (1.0f / D);
   float vs = 0.0f;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = x[mt][r] - mean; vs = fmaf(d, d, vs); }
   vs += __shfl_xor(vs, 16);
   vs += __shfl_xor(vs, 32);
   const float rstd = __builtin_amdgcn_rsqf(vs * (1.0f / D) + 1e-5f);      // v_rsq_f32 (1 ulp)
   const float mr = mean * rstd;
#pragma unroll
   for (int mt = 0; mt < MT; ++mt) {
      const float4 w4 = *reinterpret_cast<const float4 *>(w + 16 * mt + 4 * quad);
      const float4 b4 = *reinterpret_cast<const float4 *>(b + 16 * mt + 4 * quad);
      x[mt][0] = fmaf(fmaf(x[mt][0], rstd, -mr), w4.x, b4.x);
      x[mt][1] = fmaf(fmaf(x[mt][1], rstd, -mr), w4.y, b4.y);
      x[mt][2] = fmaf(fmaf(x[mt][2], rstd, -mr), w4.z, b4.z);
      x[mt][3] = fmaf(fmaf(x[mt][3], rstd, -mr), w4.w, b4.w);
   }
}

// FIRST: 0 = plain input [n][CIN][T]; 1 = Silero v3.1 first layer (input = Y - mm, finishing the adaptive normalization);
//        2 = Silero v4 first block: input = concat(magnitude `in2` [n][129][T], Y - mm) = 258 channels (silero_vad.py:212).
//        3 = the same (K = 1 form only), but the magnitude half is RECOVERED from Y: m = (e^Y - 1) 2^-20 (Y = log(1 + 2^20 m)), so that the front end
//            does not write, and this stage does not read, a second 0.8 GB array per 65,536 chunks; absolute error <= 2^-43 + 8e-7 m.
// HAS_TF: false = Silero v4 encoder stage (ConvBlock -> strided 1x1 conv with folded BatchNorm -> ReLU, no transformer
//        block; silero_vad.py:157-189 with is_v4).
// LSTM_OUT: 0 = [n][D][TOUT] (next layer / stage taps); 1 = fp32 LSTM-native tiles (common.h lstm_x_index); 2 = split-fp16
//        LSTM-native tiles (common.h lstm_xh_index): what k_lstm_wavefront_h3f reads as MFMA B fragments, written through an
//        LDS transpose so that every (chunk, step) leaves as two contiguous 128-byte rows (hi, lo) -- D == 64 only.
// K1 (with !DIRECT): first stage in the K = 1 MFMA form instead of the LDS slab path (option "encoder" = 2 selects the slab path).
// Measured per 24,576 / 65,536 chunks: v3.1 0.21 / 0.52 ms (K = 1) vs 0.23 / 0.56 (slab); v4 0.39 / 0.78 vs 0.43 / 0.85.  Both read the
// 211 MB (v4: 406 MB) per 16,384 chunks of hand-off at 1.5-2 TB/s; the read alone takes 0.033 ms from the infinity cache and 0.077 ms
// from HBM (tools/yread_probe.hip).
// H3: the transformer block's GEMMs and the strided conv in the split-fp16 form above (D = 32 / 64 layers; option "encoder" = 3 keeps fp32 MFMA)
// WAVES = 8 (Silero v4's first stage with 24 steps per chunk): 128 lanes hold 5 chunks -- 2 chunks fill only 48 of 64 lanes.  A wave's lanes must
//        hold whole runs of a chunk's steps (the depthwise conv's neighbours are wave shifts), so the middle chunk is cut in two pieces that
//        OVERLAP by four steps: lanes 48..63 of the first half carry its steps 0..15 (owning 0..13), lanes 0..11 of the second half its steps
//        12..23 (owning 14..23); the two extra steps on either side are read again from memory and only feed their neighbours.  124 of 128
//        lanes carry data, 120 own an output column: a fifth less matrix and vector work per chunk.  LDS row pitch 144.
// TAP (stage taps for the reference's op-level fixtures, vadc_amd_debug_layer1_block): the first layer's OWN instantiation entered behind the conv block --
//        `in` is y [n][16][25] -- and left after 1 = the attention incl. its out projection (transformer.c:13-153), 2 = the transformer block
//        (:160-234), 3 = LayerNorm 1 alone (misc.c:143-210); `out` is [n][16][25].  TAP = 0 compiles to the hot kernel unchanged.
template <int CIN, int D, int T, int STRIDE, bool HAS_PROJ, int FIRST, int LSTM_OUT, int NCH, bool DIRECT, bool HAS_TF = true, bool K1 = false, bool H3 = false,
          int WAVES = 4, int TAP = 0>
__global__ __launch_bounds__(64 * WAVES, (K1 && CIN < 200) ? VADC_L1_WAVES : 1) void k_layer_mfma(const float *__restrict__ in,   // [n][CIN][T]
                                                    const float *__restrict__ fm,   // [4][fm_stride] partial bin sums (FIRST) or null
                                                    LayerWeightsM w,
                                                    float *__restrict__ out,
                                                    int n_chunks, ItemMap map, size_t fm_stride,
                                                    const float *__restrict__ in2 = nullptr, int lstm_layout = 0, int tv = T)
{
   // tv <= T: the VALID input steps of every chunk (Silero v4 at a window that is no multiple of 256 samples runs the next larger built geometry: T is that geometry's
   // step count, steps tv .. T - 1 of the input hold the front end's surplus frames / the previous stage's surplus outputs).  They enter the stage as ZEROS -- what the
   // depthwise conv's zero padding behind the last valid step is (conv.c:17-53) --, the normalization offset is taken over the tv valid frames (misc.c:65-82), and the
   // outputs past the valid ones are the next stage's to ignore.  tv = T (default) changes nothing.
   // LSTM_OUT = 3: the layout of the last stage's output is the kernel ARGUMENT lstm_layout (0 / 1 / 2 as above) -- one instantiation per geometry instead of three
   // (these per-layer kernels are fallbacks and the non-default windows' path: a uniform branch in their epilogue costs nothing that matters)
   const int lstm_out = LSTM_OUT == 3 ? lstm_layout : LSTM_OUT;
   static_assert(WAVES == 4 || (WAVES == 8 && K1 && !HAS_TF && NCH == 5 && 64 - 2 * T >= 4 && 5 * T - 60 <= 64),
                 "8 waves: the K = 1 first stage without a transformer block, 5 chunks whose middle one is split with a 4-step overlap");
   constexpr int kPitchG = kPitch;                        // the 4-wave geometry (file scope)
   constexpr int kCol = 16 * WAVES;                       // columns per workgroup (one MFMA N tile per wave)
   constexpr int kPitch = WAVES == 4 ? kPitchG : 144;     // shadows the file-scope pitch from here on; 144 % 32 == 16 as well
   constexpr int NT = 64 * WAVES;                         // threads
   // TS = column stride of a chunk.  K = 1 first stage, 4 waves: TWO DEAD COLUMNS between the chunks (T + 2; they hold zeros), so that the depthwise
   // conv's wave shifts bring in the zero padding at both ends of a chunk by themselves -- no per-tap select -- and the shifted taps can ride on the
   // multiply-adds as DPP operands (9 vector instructions per channel instead of 19)
   constexpr int TS = (K1 && WAVES == 4 && NCH * (T + 2) - 2 <= kCol) ? T + 2 : T;
   constexpr int NCOLV = WAVES == 4 ? NCH * TS : kCol;    // columns that may carry data (8 waves: see colmap; TS > T: the dead columns among them do not)
   static_assert(NCH * T <= kCol, "too many chunks per workgroup");
   // column -> (chunk slot, step); returns whether the column carries data; `owner`: whether it owns an output (8 waves: the overlap lanes do not)
   constexpr int P0 = 64 - 2 * T;                         // 8 waves: lanes of the first half that carry the middle chunk's steps 0 .. P0-1
   auto colmap = [](int col, int &cb, int &t, bool &owner) -> bool {
      if constexpr (WAVES == 4) { cb = col / TS; t = col - cb * TS; owner = cb < NCH && t < T; return owner; }
      else {
         const int g = col >> 6, l = col & 63;
         if (g == 0) {
            if (l < 2 * T) { cb = l / T; t = l - cb * T; owner = true; return true; }
            cb = 2; t = l - 2 * T; owner = t < P0 - 2; return true;
         }
         constexpr int T0 = P0 - 4, L1 = T - T0;           // second half: the middle chunk's steps T0 .. T-1 on lanes 0 .. L1-1
         if (l < L1) { cb = 2; t = T0 + l; owner = l >= 2; return true; }
         const int m = l - L1;
         cb = 3 + m / T; t = m - (m / T) * T; owner = cb < NCH;
         return owner;
      }
   };
   constexpr int TOUT = 1 + (T - 1) / STRIDE;
   constexpr int HD = D / 2;
   constexpr int TP = (T + 3) / 4 * 4;                  // chunk stride of the Q / V rows in LDS (attention layout)
   static_assert(!HAS_TF || (TS > T ? (NCH - 1) * TP + (T - 1) : ((kCol - 1) / T) * TP + ((kCol - 1) % T)) < kPitch, "padded Q / V rows must fit the LDS row pitch");
   constexpr int MT = D / 16;
   constexpr int CINP = (CIN + 3) / 4 * 4;
   constexpr int KKW = CINP / 4;                        // k-steps of the pw / proj weights
   constexpr int kSlab = 32;                            // input channels per slab (slab path)
   constexpr int NSLAB = (CINP + kSlab - 1) / kSlab;
   // LDS: Yb (D rows) + Bb (3D rows).  The attention output ATT reuses Yb (y is dead once QKV is formed: the residual lives
   // in the accumulators) and relu(lin1) reuses the Q rows, so layer 4 (D = 64) needs 80 KB instead of 100 KB and TWO
   // workgroups fit a CU's 160 KB -- with one, its single wave per SIMD had nothing to hide MFMA/LDS latency behind.
   // stages without a transformer block (v4) only keep the conv block's input tile (<= 35 rows) or the LSTM-native transpose (52 rows) here:
   // 56 rows instead of 3 D (stage 4: 80 KB -> 38 KB of LDS, 2 -> 4 workgroups per CU)
   // PHD (D = 64 transformer layer, split-fp16): Q / K / V and the attention one HEAD at a time -- 96 rows instead of 192 (61 -> 30 KB), so that THREE
   // workgroups fit a CU instead of two; the layer is latency-bound (two waves per SIMD, every GEMM waiting for its weight fragments)
#ifdef VADC_NO_PHD
   constexpr bool PHD = false;
#else
   constexpr bool PHD = H3 && HAS_TF && D == 64;
#endif
   // (K = 1 first stage with 4 waves: 52 rows = the 16 KB of the partial-sum exchange; with 64 the sixth workgroup did not fit a CU's LDS)
   constexpr int ROWS_B = (DIRECT && !HAS_TF) ? 56 : (PHD ? 3 * HD : ((K1 && WAVES == 4 && 3 * D <= 52) ? 52 : ((DIRECT || 3 * D > 2 * kSlab) ? 3 * D : 2 * kSlab)));
   static_assert(!H3 || (DIRECT && HAS_TF && D % 32 == 0), "split-fp16 layer GEMMs: transformer layers with D = 32 / 64");
   constexpr int HP = D + 8, KB = H3 ? D / 32 : 1;       // H3: pitch (halves) of the split activation tiles, k-blocks per GEMM
   __shared__ __attribute__((aligned(16))) _Float16 SH[H3 ? 2 * kCol * HP : 8];   // H3: [hi | lo][column][k]: the B operand of every GEMM of the block, reused in place
   _Float16 *Sh = SH, *Sl = SH + (H3 ? kCol * HP : 0);
   __shared__ __attribute__((aligned(16))) float Yb[H3 ? 4 : D * kPitch];   // conv-block output / residual stream / ATT (fp32 MFMA form)
   __shared__ __attribute__((aligned(16))) float Bb[ROWS_B * kPitch];     // Q, K, V rows; then relu(lin1)
   float *QKV = Bb, *ATT = Yb, *FFN = Bb;
   float *XS = Bb, *DWR = Bb + kSlab * kPitch;          // slab path only (aliases Q/K/V)

   const int tid = threadIdx.x;
   const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: per-row weights become scalar loads
   const int quad = lane >> 4, lc = lane & 15;
#ifdef VADC_PHASE_PROF
   const bool ph_on = tid == 0 && (blockIdx.x & 63) == 7;
   const int ph_layer = CIN > 100 ? 0 : (D == 32 && CIN == 16 ? 1 : (D == 32 ? 2 : 3));
   unsigned long long ph_t = __builtin_readcyclecounter();
   if (ph_on) atomicAdd(&g_phase_n[ph_layer], 1u);
#endif

   // K = 1 form: this wave's input channels are requested from memory FIRST -- before the normalization prologue and its
   // barrier, which do not depend on them -- so that the whole first-stage input of the workgroup (26 KB) is in flight at
   // once; with a group of 8 channels per wave in flight the stage ran at 1.2 TB/s of a 2.7 (HBM) .. 6.4 (infinity cache) TB/s
   // read ceiling (tools/yread_probe.hip).
   constexpr int CPW = K1 ? (CIN + 3) / 4 : 1;           // input channels per wave
   // x travels through a ring of XR registers: the first XR channels are requested here, channel i + XR when channel i has been
   // consumed (v3.1: XR = CPW = 33, everything up front; v4's 65 channels per wave would cost 65 VGPRs and an occupancy step)
   constexpr int XR = K1 ? ((CPW <= 40 && VADC_L1_WAVES <= 5) ? CPW : (CPW <= 40 ? VADC_L1_XR : 24)) : 1;
   float xv[XR];
   const float *xa = in, *xb = in;
   if constexpr (K1 && !TAP) {
      int cb0, t0; bool own0;
      const bool cm0 = colmap(64 * (wave >> 2) + lane, cb0, t0, own0);   // 8 waves: waves 0-3 = lanes 0..63 of the layout, 4-7 = lanes 64..127; wave & 3 = channel quarter
      const int item0 = blockIdx.x * NCH + cb0;
      const bool cv0 = cm0 && (item0 < n_chunks);
      const int chunk0 = map(cv0 ? item0 : min(blockIdx.x * NCH, n_chunks - 1));
      xa = in + (size_t)chunk0 * ((FIRST >= 2) ? kBins : CIN) * T + t0;
      xb = FIRST == 2 ? in2 + (size_t)chunk0 * kBins * T + t0 : xa;             // FIRST 3: the magnitude half is read from Y as well
      const int c0 = (wave & 3) * CPW, c1 = min(c0 + CPW, CIN);
#pragma unroll
      for (int i = 0; i < XR; ++i) {
         const int ch = min(c0 + i, c1 - 1);               // wave-uniform; channels past the range repeat the last one (zero weights)
         const bool first_half = (FIRST >= 2) && ch < kBins;  // magnitude half of the v4 input
#ifdef VADC_L1_ABL_NOLOAD
         xv[i] = (float)(ch + lane) * 0.01f; (void)first_half;
#else
         xv[i] = first_half ? xb[(size_t)ch * T] : xa[(size_t)((FIRST >= 2) ? ch - kBins : ch) * T];
#endif
      }
   }
   __shared__ float mm_s[FIRST ? NCH : 1];
   // K = 1 form: depthwise weights [ch][k0..k4, bias] in LDS, so that a channel's six values are broadcast LDS reads that sit in
   // the same batch as its global loads (as scalar loads they cost one exposed scalar-cache round trip per channel)
   __shared__ __attribute__((aligned(8))) float dws[K1 ? 4 * CPW * 6 : 2];
   if (K1 && !TAP) {
      for (int i = tid; i < 4 * CPW; i += NT) {
#pragma unroll
         for (int j = 0; j < 5; ++j) dws[i * 6 + j] = i < CIN ? w.dw_w[i * 5 + j] : 0.0f;
         dws[i * 6 + 5] = i < CIN ? w.dw_b[i] : 0.0f;
      }
      if (!FIRST) __syncthreads();
   }
   if (FIRST && !TAP) {
      // adaptive normalization offset mm per chunk (misc.c:65-82), spread over the first wave: lane = (chunk, frame) computes its
      // frame mean and its smoothed value; one lane per chunk adds the T smoothed values in the reference's order
      __shared__ float fms[NCH * T], rs[NCH * T];
      constexpr int NVAL = NCH * T;                        // one thread per (chunk, frame)
      if (tid < NVAL) {
         const int cbp = tid / T, q = tid - cbp * T;
         const int it = blockIdx.x * NCH + cbp;
         const float *fmp = fm + (size_t)map(it < n_chunks ? it : n_chunks - 1) * T + q;
         fms[tid] = ((fmp[0] + fmp[fm_stride]) + (fmp[2 * fm_stride] + fmp[3 * fm_stride])) / 129.0f;
      }
      if (NVAL > 64) __syncthreads();                     // the (chunk, frame) threads span two waves: LDS order no longer does it
      if (tid < NVAL) {
         const int cbp = tid / T, q = tid - cbp * T;
         const float filt[7] = {0.03663284704089164733887f, 0.11128076165914535522461f, 0.21674531698226928710938f,
                                0.27068215608596801757812f, 0.21674531698226928710938f, 0.11128076165914535522461f,
                                0.03663284704089164733887f};
         float r = 0.0f;
#pragma unroll
         for (int i = 0; i < 7; ++i) {
            int qq = q + i - 3;                           // reflect pad 3, no edge repeat
            qq = qq < 0 ? -qq : qq;
            qq = qq >= tv ? 2 * (tv - 1) - qq : qq;
            r += fms[cbp * T + min(qq, T - 1)] * filt[i]; // same wave: LDS is in order
         }
         rs[tid] = r;
      }
      if (NVAL > 64) __syncthreads();
      if (tid < NVAL) {
         const int cbp = tid / T, q = tid - cbp * T;
         if (q == 0) {
            float total = 0.0f;
            for (int tt = 0; tt < tv; ++tt) total += rs[cbp * T + tt];
            mm_s[cbp] = total / (float)tv;
         }
      }
      __syncthreads();
   }
   PH(0);
   f4v acc[MT];
   acc_init<MT>(acc, w.cb_b, lane);
   // this lane's column in the accumulator layout and its place in the [n][D][T] tap tensors
   [[maybe_unused]] int tap_off = -1;
   if constexpr (TAP != 0) {
      static_assert(!TAP || (K1 && WAVES == 4 && D == 16 && HAS_TF), "taps enter the first layer's K = 1 instantiation");
      const int tcol = 16 * wave + lc;
      int tcb, tt; bool town;
      const bool tv = colmap(tcol, tcb, tt, town) && (blockIdx.x * NCH + tcb < n_chunks);
      if (tv) tap_off = map(blockIdx.x * NCH + tcb) * (D * T) + tt;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[0][r] = tap_off >= 0 ? in[tap_off + (4 * quad + r) * T] : 0.0f;     // y, as the conv block would have left it
      if constexpr (TAP == 3) {
         layer_norm_acc<MT>(acc, w.n1_w, w.n1_b, lane);
#pragma unroll
         for (int r = 0; r < 4; ++r) if (tap_off >= 0) out[tap_off + (4 * quad + r) * T] = acc[0][r];
         return;
      }
      acc_store<MT, kPitch>(acc, Yb, lane, wave);
      __syncthreads();
   } else
   if constexpr (DIRECT) {
   // ---- conv block: y = relu(pw(relu(dw(x))) + proj(x) | x)        conv.c:761-814 ---------------------------
   // The workgroup's input tile -- NCH chunks x CIN channels x T steps, one CONTIGUOUS run of CIN T floats per chunk -- is staged once
   // with 16-byte loads into LDS as [channel][column] (the rows of the Q / K / V area, not live yet); a lane's B-fragment element for k-step
   // kk is channel 4 kk + quad at its own column, and x with its 4 time neighbours for the depthwise conv are five LDS reads.  x feeds the
   // projection MFMA, relu(dw(x)) the pointwise MFMA.  (Loaded straight from global, every lane fetched x and its four neighbours itself:
   // 40-48 load instructions per lane over 28-byte segments, and the texture path, not memory latency, made the block the longest phase of
   // layers 2-4: 7 / 13 / 15 K cycles per workgroup.)
   static_assert((CIN * T) % 4 == 0, "a chunk's input tile must be whole 16-byte pieces");
   // H3C: the conv block's two GEMMs in the split-fp16 form as well (32 input channels = one k-block): a lane then owns 8 CONSECUTIVE
   // channels of its column (the B fragment's k = 8 (lane >> 4) + e), so the tile's row pitch is 66: rows 8 apart land 16 banks apart
   constexpr bool H3C = H3 && CIN == 32;
   constexpr int XP = H3C ? 66 : kPitch;
   float *XT = Bb;                                        // [CIN][XP]
   float *DWS = Bb + CIN * XP;                            // [CIN][6]
   static_assert(CIN * XP + CIN * 6 <= ROWS_B * kPitch, "input tile + depthwise weights must fit the Q / K / V area");
   {
      constexpr int Q4 = CIN * T / 4;                     // 16-byte pieces per chunk
      for (int i = tid; i < NCH * Q4; i += NT) {
         const int cbs = i / Q4, q = i - cbs * Q4;
         const int its = blockIdx.x * NCH + cbs;
         float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
         if (its < n_chunks) v = *reinterpret_cast<const float4 *>(in + (size_t)map(its) * (CIN * T) + 4 * q);
         const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
         for (int e = 0; e < 4; ++e) {
            const int idx = 4 * q + e, ch = idx / T, tt = idx - ch * T;
            XT[ch * XP + cbs * T + tt] = ve[e];
         }
      }
      // depthwise weights [ch][k0..k4, bias] next to the tile (rows CIN.. of the same area): six LDS reads per k-step instead of six
      // per-lane global loads
      for (int i = tid; i < CIN; i += NT) {
#pragma unroll
         for (int j = 0; j < 5; ++j) DWS[i * 6 + j] = w.dw_w[i * 5 + j];
         DWS[i * 6 + 5] = w.dw_b[i];
      }
   }
   __syncthreads();
   const int mcol = 16 * wave + lc;                       // this lane's column in every MFMA phase
   const int mcb = mcol / T, mt_ = mcol - mcb * T;
   const int mitem = blockIdx.x * NCH + mcb;
   const bool mvalid = (mcol < NCOLV) && (mitem < n_chunks) && mt_ < tv;
   const float mmm = FIRST ? mm_s[mcb < NCH ? mcb : 0] : 0.0f;
   const bool tl2 = mvalid && mt_ >= 2, tl1 = mvalid && mt_ >= 1, tr1 = mvalid && mt_ + 1 < tv, tr2 = mvalid && mt_ + 2 < tv;
   const int ccol = mvalid ? mcol : 0;                    // invalid columns read a valid slot and are zeroed
   const int om2 = tl2 ? -2 : 0, om1 = tl1 ? -1 : 0, op1 = tr1 ? 1 : 0, op2 = tr2 ? 2 : 0;
   if constexpr (H3C) {
      float dvv[8], xvv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
         const int ch = 8 * quad + e;
         const float *xr = XT + ch * XP + ccol;
         const float x0 = mvalid ? xr[0] : 0.0f;
         const float xm2 = tl2 ? xr[om2] : 0.0f, xm1 = tl1 ? xr[om1] : 0.0f, xp1 = tr1 ? xr[op1] : 0.0f, xp2 = tr2 ? xr[op2] : 0.0f;
         const float *k5 = DWS + ch * 6;
         const float2 k01 = *reinterpret_cast<const float2 *>(k5), k23 = *reinterpret_cast<const float2 *>(k5 + 2), k45 = *reinterpret_cast<const float2 *>(k5 + 4);
         float dv = k45.y;                                               // conv.c:17-53
         dv = fmaf(xm2, k01.x, dv); dv = fmaf(xm1, k01.y, dv); dv = fmaf(x0, k23.x, dv);
         dv = fmaf(xp1, k23.y, dv); dv = fmaf(xp2, k45.x, dv);
         dvv[e] = mvalid ? fmaxf(dv, 0.0f) : 0.0f;
         xvv[e] = x0;
      }
      h8v dh, dl, xh, xl;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
         dh[e] = (_Float16)dvv[e]; dl[e] = (_Float16)(dvv[e] - (float)dh[e]);
         xh[e] = (_Float16)xvv[e]; xl[e] = (_Float16)(xvv[e] - (float)xh[e]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
         const h8v *pp = reinterpret_cast<const h8v *>(w.pw_h + ((size_t)mt * 64 + lane) * 16);
         const h8v ah = pp[0], al = pp[1];
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, dh, acc[mt], 0, 0, 0);
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, dl, acc[mt], 0, 0, 0);
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, dh, acc[mt], 0, 0, 0);
         if (HAS_PROJ) {
            const h8v *pq = reinterpret_cast<const h8v *>(w.pj_h + ((size_t)mt * 64 + lane) * 16);
            const h8v bh_ = pq[0], bl_ = pq[1];
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl_, xh, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh_, xl, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh_, xh, acc[mt], 0, 0, 0);
         }
      }
   } else {
#pragma unroll
   for (int kk = 0; kk < KKW; ++kk) {
      const int ch = 4 * kk + quad;
      const bool chv = ch < CIN;
      const float *xr = XT + (chv ? ch : 0) * XP + ccol;
      const float x0 = (mvalid && chv) ? xr[0] - mmm : 0.0f;             // misc.c:84-96
      const float xm2 = (tl2 && chv) ? xr[om2] - mmm : 0.0f, xm1 = (tl1 && chv) ? xr[om1] - mmm : 0.0f;
      const float xp1 = (tr1 && chv) ? xr[op1] - mmm : 0.0f, xp2 = (tr2 && chv) ? xr[op2] - mmm : 0.0f;
      const float *k5 = DWS + (chv ? ch : 0) * 6;
      const float2 k01 = *reinterpret_cast<const float2 *>(k5), k23 = *reinterpret_cast<const float2 *>(k5 + 2), k45 = *reinterpret_cast<const float2 *>(k5 + 4);
      float dv = k45.y;                                                  // conv.c:17-53
      dv = fmaf(xm2, k01.x, dv); dv = fmaf(xm1, k01.y, dv); dv = fmaf(x0, k23.x, dv);
      dv = fmaf(xp1, k23.y, dv); dv = fmaf(xp2, k45.x, dv);
      dv = (mvalid && chv) ? fmaxf(dv, 0.0f) : 0.0f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
         acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.pw_f[((size_t)mt * KKW + kk) * 64 + lane], dv, acc[mt], 0, 0, 0);
         if (HAS_PROJ) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.pj_f[((size_t)mt * KKW + kk) * 64 + lane], x0, acc[mt], 0, 0, 0);
      }
   }
   }
   if (!HAS_PROJ) {                                       // identity residual (CIN == D): + x
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
         for (int r = 0; r < 4; ++r) acc[mt][r] += mvalid ? XT[(16 * mt + 4 * quad + r) * XP + ccol] : 0.0f;
   }
   __syncthreads();                                       // every wave is done with the input tile before Q / K / V rows are written
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][r] = fmaxf(acc[mt][r], 0.0f);
#ifdef VADC_PHASE_PROF
   asm volatile("" :: "v"(acc[0][0]), "v"(acc[MT - 1][3]));
   PH(9);
#endif
   if constexpr (H3) acc_store_h3<MT>(acc, Sh, Sl, HP, lane, wave);
   else acc_store<MT, kPitch>(acc, Yb, lane, wave);               // y: B operand of QKV; also kept in acc as the residual
   PH(10);
   __syncthreads();

   } else if constexpr (K1) {
   // ---- first stage (129 / 258 input channels -> 16), K = 1 form: one lane per COLUMN, one input channel per MFMA --------
   // v_mfma_f32_16x16x1_4b_f32 multiplies, in each of its 4 blocks, a 16x1 column of A with a 1x16 row of B: with the same 16
   // weights W[0..15][ch] in every block and lane l's value as B of (block l / 16, column l % 16), ONE instruction adds
   // channel ch's contribution for all 64 columns of the workgroup.  So x[ch] is loaded straight from global along the frames
   // (coalesced), its time neighbours for the depthwise conv are wave shifts of that register (a wave's 64 lanes are the 64
   // columns), relu(dw(x)) feeds the pointwise MFMA and x the projection MFMA -- no LDS, no barriers, 32 channels in flight.
   // The 4 waves split the input channels; their partial accumulators meet once in LDS (alias of the Q/K/V rows).
   static_assert(!K1 || (D == 16 && HAS_PROJ), "K = 1 form: 16 output channels with projection");
   typedef float f16acc __attribute__((ext_vector_type(16)));
   int cb, t; bool cown;
   const bool cmv = colmap(64 * (wave >> 2) + lane, cb, t, cown);
   const int item_raw = blockIdx.x * NCH + cb;
   const bool cvalid = cmv && (item_raw < n_chunks) && t < tv;
   const float mm = FIRST ? mm_s[cb < NCH ? cb : 0] : 0.0f;
   const bool l2 = t >= 2, l1 = t >= 1, r1 = t + 1 < tv, r2 = t + 2 < tv;
   const int ch0 = (wave & 3) * CPW, ch1 = min(ch0 + CPW, CIN);
   f16acc P = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef VADC_PHASE_PROF
   { float sink = 0; for (int i = 0; i < XR; ++i) sink += xv[i]; asm volatile("" :: "v"(sink)); }
   PH(8);
#endif
   // Everything a channel needs besides x -- its two weight rows (global) and its six depthwise values (LDS broadcast) -- travels
   // through rings of registers, requested WR channels ahead, and the pointwise and projection products accumulate in two independent
   // accumulators: left to the compiler each channel's loads sat right in front of its two dependent MFMAs (600 cycles per channel,
   // 20 K of the workgroup's 45 K cycles: VADC_PHASE_PROF).
   constexpr int WR = 4;
   float wra[WR], wrb[WR];
   float2 k01r[WR], k23r[WR], k45r[WR];
   auto request = [&](int i, int slot) {
      const int ch = ch0 + i;                              // < 4 CPW: pwj_k1 and dws are zero-padded to that many channels
      wra[slot] = w.pwj_k1[ch * 32 + lc]; wrb[slot] = w.pwj_k1[ch * 32 + 16 + lc];
      k01r[slot] = *reinterpret_cast<const float2 *>(&dws[ch * 6]); k23r[slot] = *reinterpret_cast<const float2 *>(&dws[ch * 6 + 2]);
      k45r[slot] = *reinterpret_cast<const float2 *>(&dws[ch * 6 + 4]);
   };
#pragma unroll
   for (int i = 0; i < WR; ++i) request(i, i);
   f16acc P2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
   for (int i = 0; i < CPW; ++i) {
      const int ch = min(ch0 + i, ch1 - 1);                // wave-uniform
      const bool first_half = (FIRST >= 2) && ch < kBins;    // magnitude half of the v4 input: no mean removed
      const float xraw = xv[i % XR];
      if (i + XR < CPW) {                                  // the ring slot is free: request channel i + XR
         const int chn = min(ch0 + i + XR, ch1 - 1);
         const bool fh = (FIRST >= 2) && chn < kBins;
         xv[i % XR] = fh ? xb[(size_t)chn * T] : xa[(size_t)((FIRST >= 2) ? chn - kBins : chn) * T];
      }
      const float xfh = FIRST == 3 ? fmaf(__builtin_amdgcn_exp2f(xraw * 1.44269504088896340736f), 0x1p-20f, -0x1p-20f) : xraw;   // magnitude from log1p(2^20 m)
      const float x = cvalid ? (first_half ? xfh : xraw - mm) : 0.0f;                                 // misc.c:84-96
      const float2 k01 = k01r[i % WR], k23 = k23r[i % WR], k45 = k45r[i % WR];
      const float a = wra[i % WR], b = wrb[i % WR];
      if (i + WR < CPW) request(i + WR, i % WR);
      float dv = k45.y;                                    // conv.c:17-53
#ifdef VADC_L1_ABL_NODW
      dv = fmaf(x, k23.x, dv);
      if constexpr (false) {
#else
      if constexpr (TS > T) {
#endif
         // two dead (zero) columns on either side of every chunk: the shifted registers ARE the zero-padded taps, and the distance-2 taps ride on
         // the multiply-add as its DPP operand (a wave shift of the distance-1 register).  Same order of the five products as below: same bits.
         // (inline asm: hipcc does not fold a wave shift into v_fmac; the s_nop covers the VALU-write -> DPP-read wait states it would pad)
         const float xm1 = dpp_wave_shr1(x);
         asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(xm1), "v"(k01.x));
         dv = fmaf(xm1, k01.y, dv);
         dv = fmaf(x, k23.x, dv);
         const float xp1 = dpp_wave_shl1(x);
         dv = fmaf(xp1, k23.y, dv);
         asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(dv) : "v"(xp1), "v"(k45.x));
         dv = __builtin_amdgcn_fmed3f(dv, 0.0f, 3.0e38f);  // max(dv, 0) in one instruction
      } else if constexpr (TS == T) {
         const float xm1 = dpp_wave_shr1(x), xm2 = dpp_wave_shr1(xm1);
         const float xp1 = dpp_wave_shl1(x), xp2 = dpp_wave_shl1(xp1);
         dv = fmaf(l2 ? xm2 : 0.0f, k01.x, dv);
         dv = fmaf(l1 ? xm1 : 0.0f, k01.y, dv);
         dv = fmaf(x, k23.x, dv);
         dv = fmaf(r1 ? xp1 : 0.0f, k23.y, dv);
         dv = fmaf(r2 ? xp2 : 0.0f, k45.x, dv);
         dv = fmaxf(dv, 0.0f);       // not masked for invalid columns: columns never mix outside a chunk's attention, dead columns stay dead
      }
#ifdef VADC_L1_ABL_NOMFMA       // timing-only ablations of the first stage's channel loop (tools/l1_ablate.sh; results wrong)
      P[0] += a * dv; P2[0] += b * x;
#else
      P = __builtin_amdgcn_mfma_f32_16x16x1f32(a, dv, P, 0, 0, 0);      // channels past CIN: zero weight rows (host padding)
      P2 = __builtin_amdgcn_mfma_f32_16x16x1f32(b, x, P2, 0, 0, 0);
#endif
   }
#pragma unroll
   for (int e = 0; e < 16; ++e) P[e] += P2[e];
#ifdef VADC_PHASE_PROF
   asm volatile("" :: "v"(P[0]), "v"(P[15]));
   PH(9);
#endif
   float *PB = Bb;                                        // WAVES x 16 x 64 floats = 16 / 32 KB (Q/K/V rows are not live yet)
   static_assert(!K1 || ROWS_B * kPitch >= WAVES * 16 * 64, "partial buffer must fit the Q/K/V rows");
#pragma unroll
   for (int e = 0; e < 16; ++e) PB[(wave * 16 + e) * 64 + lane] = P[e];
   __syncthreads();
   // wave w owns the N tile of columns [16 w, 16 w + 16) = block w & 3 of lane group w >> 2: the four channel quarters of that group add up
#pragma unroll
   for (int r = 0; r < 4; ++r) {
      float v = acc[0][r];                                // bias (acc_init)
#pragma unroll
      for (int p = 0; p < 4; ++p) v += PB[((4 * (wave >> 2) + p) * 16 + 4 * (wave & 3) + r) * 64 + lane];
      acc[0][r] = fmaxf(v, 0.0f);
   }
   __syncthreads();                                       // partial buffer consumed before anyone writes Q/K/V
   acc_store<MT, kPitch>(acc, Yb, lane, wave);                    // y: B operand of QKV; also kept in acc as the residual
   __syncthreads();

   } else {
      // slab path (layer 1: 129 input channels): x and relu(dw(x)) staged through LDS 32 channels at a time
   // column owned by this thread in the element-wise phases of the slab path: col = lane, row group = wave
   const int col = lane;
   const int cb = col / T, t = col - cb * T;
   const int item_raw = blockIdx.x * NCH + cb;
   const bool cvalid = (col < NCOLV) && (item_raw < n_chunks);
   const int chunk = map(cvalid ? item_raw : min(blockIdx.x * NCH, n_chunks - 1));
   const float mm = FIRST ? mm_s[cb < NCH ? cb : 0] : 0.0f;
   // ---- conv block: y = relu(pw(relu(dw(x))) + proj(x) | x)        conv.c:761-814 ---------------------------
   const float *x_in = in + (size_t)chunk * ((FIRST >= 2) ? kBins : CIN) * T + t;
   const float *x_in2 = (FIRST >= 2) ? in2 + (size_t)chunk * kBins * T + t : nullptr;
   auto load_x = [&](int ch) -> float {
      if (!(cvalid && ch < CIN)) return 0.0f;
      if ((FIRST >= 2)) return ch < kBins ? x_in2[(size_t)ch * T] : x_in[(size_t)(ch - kBins) * T];
      return x_in[(size_t)ch * T];
   };
   // x slab: 8 rows per wave; rows >= CIN and invalid columns are zero.  The NEXT slab's rows are requested from
   // HBM as soon as the current ones are in LDS, so their latency hides behind the depthwise conv and the MFMAs.
   float xv[8];
#pragma unroll
   for (int i = 0; i < 8; ++i) xv[i] = load_x(wave * 8 + i);
#pragma unroll 1
   for (int s = 0; s < NSLAB; ++s) {
      const int c0 = s * kSlab;
      if (s > 0) __syncthreads();                        // previous slab fully consumed
      // In the slab path a wave's 64 lanes ARE the 64 columns, so the depthwise conv's time neighbours are the adjacent
      // lanes: x(t-1), x(t-2), x(t+1), x(t+2) come from DPP wave shifts of the value that is in a register anyway, not from
      // an LDS round trip behind a barrier.  depthwise k5 pad2 + ReLU: conv.c:17-53.
      float xc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
         const int r = wave * 8 + i, ch = c0 + r;
         xc[i] = (cvalid && ch < CIN) ? xv[i] - (((FIRST >= 2) && ch < kBins) ? 0.0f : mm) : 0.0f;  // misc.c:84-96
         XS[r * kPitch + col] = xc[i];
      }
      if (s + 1 < NSLAB) {
#pragma unroll
         for (int i = 0; i < 8; ++i) xv[i] = load_x(c0 + kSlab + wave * 8 + i);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
         const int r = wave * 8 + i, ch = c0 + r;
         const float xm1 = dpp_wave_shr1(xc[i]), xm2 = dpp_wave_shr1(xm1);
         const float xp1 = dpp_wave_shl1(xc[i]), xp2 = dpp_wave_shl1(xp1);
         float dv = 0.0f;
         if (ch < CIN) {                                 // wave-uniform
            const float *k = w.dw_w + ch * 5;
            dv = w.dw_b[ch];
            dv = fmaf(t >= 2 ? xm2 : 0.0f, k[0], dv);
            dv = fmaf(t >= 1 ? xm1 : 0.0f, k[1], dv);
            dv = fmaf(xc[i], k[2], dv);
            dv = fmaf(t + 1 < T ? xp1 : 0.0f, k[3], dv);
            dv = fmaf(t + 2 < T ? xp2 : 0.0f, k[4], dv);
            dv = fmaxf(dv, 0.0f);
         }
         DWR[r * kPitch + col] = cvalid ? dv : 0.0f;
      }
      __syncthreads();
      constexpr int KK_FULL = kSlab / 4;
      if (c0 + kSlab <= CINP) {
         gemm_acc<MT, KK_FULL, kPitch>(acc, w.pw_f, KKW, c0 / 4, DWR, lane, wave);
         if (HAS_PROJ) gemm_acc<MT, KK_FULL, kPitch>(acc, w.pj_f, KKW, c0 / 4, XS, lane, wave);
      } else {
         constexpr int KK_TAIL = (CINP % kSlab) / 4 > 0 ? (CINP % kSlab) / 4 : 1;
         gemm_acc<MT, KK_TAIL, kPitch>(acc, w.pw_f, KKW, c0 / 4, DWR, lane, wave);
         if (HAS_PROJ) gemm_acc<MT, KK_TAIL, kPitch>(acc, w.pj_f, KKW, c0 / 4, XS, lane, wave);
      }
   }
   if (!HAS_PROJ) {                                       // identity residual (CIN == D, single slab): + x
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
         for (int r = 0; r < 4; ++r) acc[mt][r] += XS[(16 * mt + 4 * quad + r) * kPitch + 16 * wave + lc];
   }
#pragma unroll
   for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][r] = fmaxf(acc[mt][r], 0.0f);
   acc_store<MT, kPitch>(acc, Yb, lane, wave);                    // y: B operand of QKV; also kept in acc as the residual
   __syncthreads();                                       // slab buffers free, Yb visible

   }
   PH(1);
   if constexpr (HAS_TF) {
   if constexpr (PHD) {
   // ---- QKV and attention, one head at a time (same arithmetic per element as below) --------------------------------------------
   constexpr int MTH = HD / 16;
   h8v ybh[KB], ybl[KB];
   load_b_h3<KB>(ybh, ybl, Sh, Sl, HP, lane, wave);        // y of this wave's columns: in registers for both heads (the attention output overwrites the tiles)
   const int qcol = 16 * wave + lc;
   const int qcb = qcol / TS;
   const int qcolp = qcb * TP + (qcol - qcb * TS);
#pragma unroll
   for (int h = 0; h < 2; ++h) {
      f4v q[3 * MTH];
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
         for (int j = 0; j < MTH; ++j) {
            const int mtg = part * MT + h * MTH + j;       // M tile of the full QKV weight
            const float4 b4 = *reinterpret_cast<const float4 *>(w.qkv_b + 16 * mtg + 4 * quad);
            f4v a4 = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
               const h8v *pw = reinterpret_cast<const h8v *>(w.qkv_h + (((size_t)mtg * KB + kb) * 64 + lane) * 16);
               const h8v ah = pw[0], al = pw[1];
               a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, ybh[kb], a4, 0, 0, 0);
               a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ybl[kb], a4, 0, 0, 0);
               a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ybh[kb], a4, 0, 0, 0);
            }
            q[part * MTH + j] = a4;
         }
      if (h > 0) __syncthreads();                          // the previous head's attention is done with the Q / K / V rows
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
         for (int j = 0; j < MTH; ++j) {
            const int cdst = part == 1 ? qcol : qcolp;     // K rows keep the column layout, Q and V rows the padded-chunk layout (see below)
#pragma unroll
            for (int r = 0; r < 4; ++r) QKV[(part * HD + 16 * j + 4 * quad + r) * kPitch + cdst] = q[part * MTH + j][r];
         }
      __syncthreads();
      {
         // FOUR adjacent lanes share one column's task of this head: lane p takes a quarter of the head's hd dimensions
         const int p = tid & 3, i = tid >> 2;
         const int icb = i / TS;
         constexpr int HH = HD / 4, TQ = TP / 4;
         static_assert(HH == 8, "one 16-byte store of halves per lane");
         _Float16 *dsh = Sh + i * HP + h * HD + p * HH, *dsl = Sl + i * HP + h * HD + p * HH;
         float ov[HH];
         if (i < NCOLV) {
            const float *Q = QKV + (p * HH) * kPitch + icb * TP, *K = QKV + (HD + p * HH) * kPitch + i;
            const float *V = QKV + (2 * HD + p * HH) * kPitch + icb * TP;
            float sc[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) sc[j] = 0.0f;
#pragma unroll
            for (int e = 0; e < HH; ++e) {
               const float ke = K[e * kPitch];
#pragma unroll
               for (int q4 = 0; q4 < TQ; ++q4) {
                  const float4 qv = *reinterpret_cast<const float4 *>(Q + e * kPitch + 4 * q4);
                  sc[4 * q4 + 0] = fmaf(ke, qv.x, sc[4 * q4 + 0]); sc[4 * q4 + 1] = fmaf(ke, qv.y, sc[4 * q4 + 1]);
                  sc[4 * q4 + 2] = fmaf(ke, qv.z, sc[4 * q4 + 2]); sc[4 * q4 + 3] = fmaf(ke, qv.w, sc[4 * q4 + 3]);
               }
            }
            const float scale = 1.0f / sqrtf((float)HD);      // transformer.c:114
            float mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < T; ++j) {
               float v = sc[j] + __shfl_xor(sc[j], 1);
               v += __shfl_xor(v, 2);
               sc[j] = v * scale;
               mx = fmaxf(mx, sc[j]);
            }
            float sum = 0.0f;                                 // tensor.h:751-784
#pragma unroll
            for (int j = 0; j < T; ++j) { sc[j] = __expf(sc[j] - mx); sum += sc[j]; }
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int e = 0; e < HH; ++e) {
               float o = 0.0f;
#pragma unroll
               for (int q4 = 0; q4 < TQ; ++q4) {
                  const float4 vv = *reinterpret_cast<const float4 *>(V + e * kPitch + 4 * q4);
                  if (4 * q4 + 0 < T) o = fmaf(sc[4 * q4 + 0], vv.x, o);
                  if (4 * q4 + 1 < T) o = fmaf(sc[4 * q4 + 1], vv.y, o);
                  if (4 * q4 + 2 < T) o = fmaf(sc[4 * q4 + 2], vv.z, o);
                  if (4 * q4 + 3 < T) o = fmaf(sc[4 * q4 + 3], vv.w, o);
               }
               ov[e] = o * inv;
            }
         } else {
#pragma unroll
            for (int e = 0; e < HH; ++e) ov[e] = 0.0f;
         }
         h8v hi, lo;
#pragma unroll
         for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)ov[e]; lo[e] = (_Float16)(ov[e] - (float)hi[e]); }
         *reinterpret_cast<h8v *>(dsh) = hi;
         *reinterpret_cast<h8v *>(dsl) = lo;
      }
   }
   __syncthreads();
   PH(3);
   } else {
   // ---- QKV = W y + b  -> LDS rows [0,D) Q, [D,2D) K, [2D,3D) V      transformer.c:69-99 -----------------
   {
      f4v q[3 * MT];
      acc_init<3 * MT>(q, w.qkv_b, lane);
      if constexpr (H3) {
         h8v bh[KB], bl[KB];
         load_b_h3<KB>(bh, bl, Sh, Sl, HP, lane, wave);
         gemm_acc_h3<3 * MT, KB>(q, w.qkv_h, bh, bl, lane);
      } else gemm_acc<3 * MT, D / 4, kPitch>(q, w.qkv_f, D / 4, 0, Yb, lane, wave);
      // K rows keep the column layout (attention reads K at its own column).  Q and V rows are stored with every chunk's T steps
      // padded to TP = a multiple of 4 (the row pitch has room: NCH TP <= 72), so that the attention below fetches a row's T
      // values of one chunk with T/4 aligned 16-byte reads instead of T scalar ones -- that phase is bound by LDS instructions.
      const int qcol = 16 * wave + lc;
      const int qcb = qcol / TS;
      const int qcolp = min(qcb * TP + (qcol - qcb * TS), kPitch - 1);      // (dead columns past the last chunk stay inside the row)
#pragma unroll
      for (int mt = 0; mt < 3 * MT; ++mt) {
         const int cdst = (mt >= MT && mt < 2 * MT) ? qcol : qcolp;
#pragma unroll
         for (int r = 0; r < 4; ++r) QKV[(16 * mt + 4 * quad + r) * kPitch + cdst] = q[mt][r];
      }
   }
   __syncthreads();
   PH(2);

   // ---- attention per (column i, head h): a = softmax_j(k_i . q_j / sqrt(hd)), att_i = sum_j a_j v_j ----------
   //      (K Q^T, not Q K^T: transformer.c:104-105)
   // All 256 threads: a PAIR of adjacent lanes shares one (head, column) task -- lane p takes half of the head's hd dimensions:
   // its part of the k.q dot products (summed across the pair with a quad-permute shuffle) and its hd/2 output rows.  Q and V rows
   // are read as whole padded chunks (16-byte reads).  One thread per task with scalar reads made 2 T hd LDS instructions per task
   // on two of the four waves: the longest phase of the transformer block.
   {
      const int p = tid & 1, task = tid >> 1;
      const int h = task / kCol, i = task - h * kCol;
      const int icb = i / TS;
      constexpr int HH = HD / 2, TQ = TP / 4;
      // which HH of the head's HD dimensions lane p takes.  H3: the consecutive half p HH .. (its outputs are one 16-byte store of halves).  fp32 form
      // (the first layer): the INTERLEAVED ones p, p + 2, ... -- the pair's rows are then one row apart instead of HH: HH kPitch is a multiple of the 64
      // banks, so with halves both lanes of a pair hit the same bank at different addresses on every K, Q and V read (2-way conflicts on all of them)
      constexpr int RS = H3 ? 1 : 2;                        // row stride of a lane's dimensions
      const int r0 = H3 ? p * HH : p;                       // its first one
      float *dst = ATT + (h * HD + r0) * kPitch + i;
      _Float16 *dsh = Sh + i * HP + h * HD + p * HH, *dsl = Sl + i * HP + h * HD + p * HH;    // H3: this lane's HH consecutive k of column i
      float ov[HH];
      if (i < NCOLV && i - icb * TS < T) {
         const float *Q = QKV + (h * HD + r0) * kPitch + icb * TP, *K = QKV + (D + h * HD + r0) * kPitch + i;
         const float *V = QKV + (2 * D + h * HD + r0) * kPitch + icb * TP;
         float sc[TP];
#pragma unroll
         for (int j = 0; j < TP; ++j) sc[j] = 0.0f;
#pragma unroll
         for (int e = 0; e < HH; ++e) {
            const float ke = K[e * RS * kPitch];
#pragma unroll
            for (int q4 = 0; q4 < TQ; ++q4) {
               const float4 qv = *reinterpret_cast<const float4 *>(Q + e * RS * kPitch + 4 * q4);
               sc[4 * q4 + 0] = fmaf(ke, qv.x, sc[4 * q4 + 0]); sc[4 * q4 + 1] = fmaf(ke, qv.y, sc[4 * q4 + 1]);
               sc[4 * q4 + 2] = fmaf(ke, qv.z, sc[4 * q4 + 2]); sc[4 * q4 + 3] = fmaf(ke, qv.w, sc[4 * q4 + 3]);
            }
         }
         const float scale = 1.0f / sqrtf((float)HD);      // transformer.c:114
         float mx = -3.0e38f;
#pragma unroll
         for (int j = 0; j < T; ++j) {
            sc[j] = (sc[j] + __shfl_xor(sc[j], 1)) * scale;
            mx = fmaxf(mx, sc[j]);
         }
         float sum = 0.0f;                                 // tensor.h:751-784
#pragma unroll
         for (int j = 0; j < T; ++j) { sc[j] = __expf(sc[j] - mx); sum += sc[j]; }   // v_exp_f32 (1 ulp); arguments <= 0
         const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
         for (int e = 0; e < HH; ++e) {
            float o = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < TQ; ++q4) {
               const float4 vv = *reinterpret_cast<const float4 *>(V + e * RS * kPitch + 4 * q4);
               if (4 * q4 + 0 < T) o = fmaf(sc[4 * q4 + 0], vv.x, o);
               if (4 * q4 + 1 < T) o = fmaf(sc[4 * q4 + 1], vv.y, o);
               if (4 * q4 + 2 < T) o = fmaf(sc[4 * q4 + 2], vv.z, o);
               if (4 * q4 + 3 < T) o = fmaf(sc[4 * q4 + 3], vv.w, o);
            }
            if constexpr (H3) ov[e] = o * inv; else dst[e * RS * kPitch] = o * inv;
         }
      } else {
#pragma unroll
         for (int e = 0; e < HH; ++e) { if constexpr (H3) ov[e] = 0.0f; else dst[e * RS * kPitch] = 0.0f; }
      }
      if constexpr (H3) {
#pragma unroll
         for (int e8 = 0; e8 < HH / 8; ++e8) {
            h8v hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)ov[8 * e8 + e]; lo[e] = (_Float16)(ov[8 * e8 + e] - (float)hi[e]); }
            *reinterpret_cast<h8v *>(dsh + 8 * e8) = hi;
            *reinterpret_cast<h8v *>(dsl + 8 * e8) = lo;
         }
      }
   }
   __syncthreads();
   PH(3);
   }  // !PHD

   // ---- out projection + residual, LN1, FFN, residual, LN2          transformer.c:202-220 -------------------
   {
      f4v p[MT];
      acc_init<MT>(p, w.out_b, lane);
      if constexpr (H3) {
         h8v bh[KB], bl[KB];
         load_b_h3<KB>(bh, bl, Sh, Sl, HP, lane, wave);
         gemm_acc_h3<MT, KB>(p, w.out_h, bh, bl, lane);
      } else gemm_acc<MT, D / 4, kPitch>(p, w.out_f, D / 4, 0, ATT, lane, wave);
      if constexpr (TAP == 1) {                               // dual_head_attention's result: softmax(k q^T) v through the out projection
#pragma unroll
         for (int r = 0; r < 4; ++r) if (tap_off >= 0) out[tap_off + (4 * quad + r) * T] = p[0][r];
         return;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] += p[mt];
   }
   layer_norm_acc<MT>(acc, w.n1_w, w.n1_b, lane);
   PH(4);
   // The wave's own 16 columns only from here on: a WAVE-level LDS round trip turns the accumulator layout into the next GEMM's B
   // fragments.  A wave's LDS instructions execute in order, every element a wave reads below was written by the same wave (rows
   // of its own columns), and no other wave touches those columns after the attention barrier -- so no workgroup barrier is
   // needed until the epilogue, and the four waves drift apart instead of meeting five times.
   // (H3: one pair of split tiles serves the whole chain in place -- a wave holds ALL of its B fragments of a GEMM in registers
   // before it stores that GEMM's output over them)
   if constexpr (H3) acc_store_h3<MT>(acc, Sh, Sl, HP, lane, wave); else acc_store<MT, kPitch>(acc, Yb, lane, wave);
   {
      f4v f[MT];
      acc_init<MT>(f, w.l1_b, lane);
      if constexpr (H3) {
         h8v bh[KB], bl[KB];
         load_b_h3<KB>(bh, bl, Sh, Sl, HP, lane, wave);
         gemm_acc_h3<MT, KB>(f, w.l1_h, bh, bl, lane);
      } else gemm_acc<MT, D / 4, kPitch>(f, w.l1_f, D / 4, 0, Yb, lane, wave);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
         for (int r = 0; r < 4; ++r) f[mt][r] = fmaxf(f[mt][r], 0.0f);
      if constexpr (H3) acc_store_h3<MT>(f, Sh, Sl, HP, lane, wave);
      else acc_store<MT, kPitch>(f, FFN, lane, wave);             // relu(lin1) goes to the (dead) Q rows
   }
   PH(5);
   {
      f4v g[MT];
      acc_init<MT>(g, w.l2_b, lane);
      if constexpr (H3) {
         h8v bh[KB], bl[KB];
         load_b_h3<KB>(bh, bl, Sh, Sl, HP, lane, wave);
         gemm_acc_h3<MT, KB>(g, w.l2_h, bh, bl, lane);
      } else gemm_acc<MT, D / 4, kPitch>(g, w.l2_f, D / 4, 0, FFN, lane, wave);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] += g[mt];
   }
   layer_norm_acc<MT>(acc, w.n2_w, w.n2_b, lane);
   if constexpr (TAP == 2) {                                  // transformer_block's result
#pragma unroll
      for (int r = 0; r < 4; ++r) if (tap_off >= 0) out[tap_off + (4 * quad + r) * T] = acc[0][r];
      return;
   }
   if constexpr (H3) acc_store_h3<MT>(acc, Sh, Sl, HP, lane, wave); else acc_store<MT, kPitch>(acc, Yb, lane, wave);
   PH(6);
   }  // HAS_TF

   // ---- conv k=1 stride s (+ folded BatchNorm) -> ReLU; only surviving time steps are stored -----------------
   {
      f4v z[MT];
      acc_init<MT>(z, w.cv_b, lane);
      if constexpr (H3) {
         h8v bh[KB], bl[KB];
         load_b_h3<KB>(bh, bl, Sh, Sl, HP, lane, wave);
         gemm_acc_h3<MT, KB>(z, w.cv_h, bh, bl, lane);
      } else gemm_acc<MT, D / 4, kPitch>(z, w.cv_f, D / 4, 0, Yb, lane, wave);
      // this lane's column in the accumulator layout
      const int ocol = 16 * wave + lc;
      int ocb, ot; bool oown;
      (void)colmap(ocol, ocb, ot, oown);
      const int oitem = blockIdx.x * NCH + ocb;
      if ((LSTM_OUT == 2 || LSTM_OUT == 3) && lstm_out == 2) {
         static_assert(D == 64 || (LSTM_OUT != 2 && LSTM_OUT != 3), "the LSTM hand-off is the last stage (64 units)");
         // transpose through LDS (Bb is dead here): Zs[column][unit], then every thread converts 16 consecutive units of one
         // column and stores 32 contiguous bytes of the hi row and of the lo row
         float *Zs = Bb;                                   // 64 columns x pitch 65
         __syncthreads();                                  // all waves are past their last read of Bb
#pragma unroll
         for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Zs[ocol * 65 + 16 * mt + 4 * quad + r] = fmaxf(z[mt][r], 0.0f);
         __syncthreads();
         const int tcol = tid >> 2, u0 = 16 * (tid & 3);
         const int tcb = tcol / T, tt = tcol - tcb * T;
         const int titem = blockIdx.x * NCH + tcb;
         if (tcol < NCOLV && titem < n_chunks && (tt % STRIDE) == 0) {
            int st_, ch_;
            map.split(titem, st_, ch_);
            typedef _Float16 h8o __attribute__((ext_vector_type(8)));
            _Float16 *dsth = reinterpret_cast<_Float16 *>(out) + lstm_xh_index(st_, ch_, map.C, tt / STRIDE, u0, TOUT);
            h8o hi[2], lo[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
               const float v = Zs[tcol * 65 + u0 + e];
               const _Float16 h = (_Float16)v;
               hi[e >> 3][e & 7] = h;
               lo[e >> 3][e & 7] = (_Float16)(v - (float)h);
            }
            *reinterpret_cast<h8o *>(dsth) = hi[0]; *reinterpret_cast<h8o *>(dsth + 8) = hi[1];
            *reinterpret_cast<h8o *>(dsth + kLstmTile * 64) = lo[0]; *reinterpret_cast<h8o *>(dsth + kLstmTile * 64 + 8) = lo[1];
         }
      } else
      if (oown && oitem < n_chunks && (ot % STRIDE) == 0) {
         const int ostride = lstm_out ? kLstmTile : TOUT;
         float *dst;
         if (lstm_out) {
            int st_, ch_;
            map.split(oitem, st_, ch_);
            dst = out + lstm_x_index(st_, ch_, map.C, ot / STRIDE, 0, TOUT);
         } else {
            dst = out + (size_t)map(oitem) * D * TOUT + ot / STRIDE;
         }
#pragma unroll
         for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)(16 * mt + 4 * quad + r) * ostride] = fmaxf(z[mt][r], 0.0f);
      }
   }
   PH(7);
}

#ifdef VADC_PHASE_PROF
extern "C" void vadc_phase_report(void)
{
   unsigned long long h[4][16]; unsigned int n[4];
   (void)hipDeviceSynchronize();
   (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h));
   (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_phase_n), sizeof(n));
   const char *names[8] = {"prologue", "conv block", "QKV", "attention", "out-proj+LN1", "FFN lin1", "lin2+LN2+store", "strided conv+out"};
   for (int l = 0; l < 4; ++l) {
      if (!n[l]) continue;
      printf("layer %d (%u workgroups sampled):", l + 1, n[l]);
      for (int i = 0; i < 8; ++i) printf("  %s %.0f", names[i], (double)h[l][i] / n[l]);
      printf("  | conv block: inputs arrived %.0f, MFMAs + residual %.0f, acc_store %.0f (barrier wait is the rest)", (double)h[l][8] / n[l], (double)h[l][9] / n[l], (double)h[l][10] / n[l]);
      printf("   [cycles of s_memtime/readcyclecounter]\n");
   }
}
#endif

// chunks per workgroup: L1 T=25 -> 2 (50 of 64 columns), L2 T=13 -> 4 (52), L3/L4 T=7 -> 9 (63)
// One launch per layer with fp32 MFMA: the FALLBACK of the register-resident kernels (a weight outside fp16's range, a failed self-check), option "encoder" = 3 /
// "layer1" = 1, and the literal-fp32 configuration of bench.py.  (Until round 5 this launcher also carried the per-layer split-fp16 forms -- option "encoder" = 5, round
// 2's hot path -- and the first layer's LDS slab path -- "encoder" = 2: experiments, neither a default nor a fallback.)
void launch_layer_mfma(int layer, const float *in, const float *fm, const LayerWeightsM &w, float *out, int n, ItemMap map,
                       int lstm_layout, size_t fm_stride, hipStream_t st)
{
   switch (layer) {
   case 0:
#ifdef VADC_L1_ABL_NOTF
      hipLaunchKernelGGL((k_layer_mfma<129, 16, 25, 2, true, true, false, 2, false, false, true>), dim3((n + 1) / 2), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride);
#else
      hipLaunchKernelGGL((k_layer_mfma<129, 16, 25, 2, true, true, false, 2, false, true, true>), dim3((n + 1) / 2), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride);
#endif
      break;
   case 1: hipLaunchKernelGGL((k_layer_mfma<16, 32, 13, 2, true, false, false, 4, true>), dim3((n + 3) / 4), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride); break;
   case 2: hipLaunchKernelGGL((k_layer_mfma<32, 32, 7, 1, false, false, false, 9, true>), dim3((n + 8) / 9), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride); break;
   case 3:
      hipLaunchKernelGGL((k_layer_mfma<32, 64, 7, 1, true, 0, 3, 9, true>), dim3((n + 8) / 9), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride, (const float *)nullptr, lstm_layout);
      break;
   }
}

// stage taps of the first layer (vadc_amd_debug_layer1_block): y [n][16][25] -> [n][16][25]
void launch_layer1_tap(int what, const float *y, const LayerWeightsM &w, float *out, int n, ItemMap map, hipStream_t st)
{
   const dim3 grid((n + 1) / 2), block(256);
   switch (what) {
   case 1: hipLaunchKernelGGL((k_layer_mfma<129, 16, 25, 2, true, true, false, 2, false, true, true, false, 4, 1>), grid, block, 0, st, y, nullptr, w, out, n, map, 0); break;
   case 2: hipLaunchKernelGGL((k_layer_mfma<129, 16, 25, 2, true, true, false, 2, false, true, true, false, 4, 2>), grid, block, 0, st, y, nullptr, w, out, n, map, 0); break;
   default: hipLaunchKernelGGL((k_layer_mfma<129, 16, 25, 2, true, true, false, 2, false, true, true, false, 4, 3>), grid, block, 0, st, y, nullptr, w, out, n, map, 0); break;
   }
}

// Silero v4 encoder stages (silero_vad.py:157-189, is_v4, strides 2, 2, 2, 1).  T0 = frames of the window = samples / 64 (onnx_helpers.c:164-170
// lets the v4 graph take 512 ... 1536 samples): 24 -> 12 -> 6 -> 3 -> 3 (1536), 20 -> 10 -> 5 -> 3 -> 3 (1280), 16 -> 8 -> 4 -> 2 -> 2 (1024), 12 -> 6 -> 3 -> 2 -> 2 (768),
// 8 -> 4 -> 2 -> 1 -> 1 (512).
// chunks per workgroup fill the 64 columns: T0 = 24: 2 / 5 / 10 / 21;  16: 4 / 8 / 16 / 32;  8: 8 / 16 / 32 / 64.  S3 = stride of the third strided conv.
// The first stage in its K = 1 form takes the magnitude half of its input from Y (FIRSTK = 3): no magnitude array.  (Until round 5: the LDS slab path -- "encoder" = 2 --,
// the magnitudes from a second array -- "v4_mag" = 1 -- and a 4-wave form at 24 frames -- "encoder" = 4: experiments.)
template <int T0, int NCH>
static void launch_v4_first(const float *in, const float *fm, const LayerWeightsM &w, float *out, int n, ItemMap map, size_t fm_stride, hipStream_t st, int tv)
{
   const float *no_in2 = nullptr;
   if constexpr (T0 == 24) {                                // 8 waves: 5 chunks per workgroup (120 of 128 lanes own a column instead of 48 of 64)
      hipLaunchKernelGGL((k_layer_mfma<258, 16, T0, 2, true, 3, false, 5, false, false, true, false, 8>), dim3((n + 4) / 5), dim3(512), 0, st, in, fm, w, out, n, map, fm_stride, no_in2, 0, tv);
      return;
   }
   hipLaunchKernelGGL((k_layer_mfma<258, 16, T0, 2, true, 3, false, NCH, false, false, true>), dim3((n + NCH - 1) / NCH), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride, no_in2, 0, tv);
}
// tv0: the valid STFT frames of a chunk (= T0 at the built windows; fewer at a window in between, which runs the next larger built geometry: see the kernel's `tv`)
template <int T0, int S3>
static void launch_v4_t(int layer, const float *in, const float *fm, const LayerWeightsM &w, float *out, int n, ItemMap map,
                        int lstm_layout, size_t fm_stride, hipStream_t st, int tv0)
{
   constexpr int T1 = (T0 + 1) / 2, T2 = (T1 + 1) / 2, T3 = S3 == 2 ? (T2 + 1) / 2 : T2;      // a k = 1 conv of stride 2 keeps 1 + (T - 1) / 2 steps (12 -> 6 -> 3 -> 2, 20 -> 10 -> 5 -> 3)
   constexpr int N0 = 64 / T0, N1 = 64 / T1, N2 = 64 / T2, N3 = 64 / T3;
   const int tv1 = (tv0 + 1) / 2, tv2 = (tv1 + 1) / 2, tv3 = S3 == 2 ? (tv2 + 1) / 2 : tv2;
   const float *in2 = nullptr;
   switch (layer) {
   case 0: launch_v4_first<T0, N0>(in, fm, w, out, n, map, fm_stride, st, tv0); break;
   case 1: hipLaunchKernelGGL((k_layer_mfma<16, 32, T1, 2, true, 0, false, N1, true, false>), dim3((n + N1 - 1) / N1), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride, in2, 0, tv1); break;
   case 2: hipLaunchKernelGGL((k_layer_mfma<32, 32, T2, S3, false, 0, false, N2, true, false>), dim3((n + N2 - 1) / N2), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride, in2, 0, tv2); break;
   case 3:
      hipLaunchKernelGGL((k_layer_mfma<32, 64, T3, 1, true, 0, 3, N3, true, false>), dim3((n + N3 - 1) / N3), dim3(256), 0, st, in, fm, w, out, n, map, fm_stride, in2, lstm_layout, tv3);
      break;
   }
}

// frames = STFT frames per chunk of the BUILT geometry the window runs in (multiples of 4); frames_valid <= frames: the window's own (a window that is no multiple of
// 256 samples).  stride3 = stride of the third strided conv: 2 in the 16 kHz branch (frames 24 / 20 / 16 / 12 / 8), 1 in the 8 kHz branch (silero_vad.py:178-181;
// frames 12 / 8 / 4: 12 -> 6 -> 3 -> 3 -> 3, ...).  The LSTM steps of frames and frames_valid are the same for every frames_valid in (frames - 4, frames].
void launch_layer_v4(int layer, const float *in, const float *fm, const LayerWeightsM &w, float *out, int n, ItemMap map,
                     int lstm_layout, size_t fm_stride, hipStream_t st, int frames, int stride3, int frames_valid)
{
   const int fv = frames_valid > 0 && frames_valid < frames ? frames_valid : frames;
   if (stride3 == 1) {
      if (frames == 8)      launch_v4_t<8, 1>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);
      else if (frames == 4) launch_v4_t<4, 1>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);
      else                  launch_v4_t<12, 1>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);
      return;
   }
   if (frames == 16)      launch_v4_t<16, 2>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);
   else if (frames == 12) launch_v4_t<12, 2>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);      // 768-sample window (round 5)
   else if (frames == 20) launch_v4_t<20, 2>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);      // 1280-sample window
   else if (frames == 8)  launch_v4_t<8, 2>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);
   else                   launch_v4_t<24, 2>(layer, in, fm, w, out, n, map, lstm_layout, fm_stride, st, fv);
}

}  // namespace vadc
