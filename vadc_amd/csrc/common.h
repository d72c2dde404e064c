// common.h -- shapes and device-side weight views shared by the kernels and the engine.
// Silero v3.1 / 16 kHz / 1536-sample chunk (reference tensor.h:154-191).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vadc {

constexpr int kChunk     = 1536;
constexpr int kPad       = 128;
constexpr int kPadded    = 1792;
constexpr int kBlocks    = 28;      // 64-sample blocks in the reflect-padded chunk
constexpr int kFilterLen = 256;
constexpr int kBins      = 129;
constexpr int kFilters   = 258;
constexpr int kFrames    = 25;
constexpr int kHidden    = 64;
constexpr int kBinSplit      = 4;    // partial bin sums: k_frontend splits the 129 bins over gridDim.y, k_frontend_mx over its 4 waves
constexpr int kBinsPerSplit  = 33;

// One encoder layer's weights on the device.  Matrices keep the reference's [out][in] layout
// (rows contiguous over `in`) except the conv-block pointwise/proj weights which are stored
// transposed [in][out] because the conv block walks input channels in its outer loop.
struct LayerWeights {
   const float *dw_w;     // [cin][5]
   const float *dw_b;     // [cin]
   const float *pwT;      // [cin][d]
   const float *pw_b;     // [d]
   const float *pjT;      // [cin][d] or nullptr
   const float *pj_b;     // [d]      or nullptr
   const float *qkv_w;    // [3d][d]
   const float *qkv_b;    // [3d]
   const float *out_w;    // [d][d]
   const float *out_b;    // [d]
   const float *n1_w, *n1_b;
   const float *l1_w, *l1_b;
   const float *l2_w, *l2_b;
   const float *n2_w, *n2_b;
   const float *cv_w;     // [d][d]  strided 1x1 conv with BatchNorm folded in at load time
   const float *cv_b;     // [d]
};

// A launch may cover only chunks [c0, c0+cg) of every stream (the engine pipelines chunk groups so that the
// latency-bound LSTM of group g overlaps the front end / encoder of group g+1).  Buffers keep the full
// stream-major layout item n = stream * C + chunk; launch-local item i maps to n as below.
struct ItemMap {
   int C, c0, cg;
   __host__ __device__ __forceinline__ int operator()(int i) const
   {
      const int s = i / cg;
      return s * C + c0 + (i - s * cg);
   }
   __host__ __device__ __forceinline__ void split(int i, int &stream, int &chunk) const
   {
      stream = i / cg;
      chunk = c0 + (i - stream * cg);
   }
};

// Encoder -> LSTM hand-off layout ("LSTM-native"): streams are tiled by 16 (the MFMA N dimension) and one
// (tile, chunk) block holds the chunk's 7 frames as [t][unit][stream-in-tile], i.e. exactly the LDS image
// k_lstm_mfma uses as its B operand, so the LSTM stages a chunk with coalesced 16-byte loads.
//   X[((tile * C + chunk) * steps + t) * 64 + unit) * 16 + stream % 16]
constexpr int kLstmTile = 16;
// steps = LSTM steps per chunk: 7 (Silero v3.1) or 3 (Silero v4)
__host__ __device__ __forceinline__ size_t lstm_x_index(int stream, int chunk, int C, int t, int unit, int steps = 7)
{
   return ((((size_t)(stream / kLstmTile) * C + chunk) * steps + t) * 64 + unit) * kLstmTile + (stream % kLstmTile);
}

// log1p(x) for x >= 0 on the hardware transcendental unit instead of libm's ~50-instruction log1pf:
//   u = fl(1 + x), c = x - (u - 1) the exact rounding error of u;  y0 = ln2 * v_log_f32(u)  (a few ulp);
//   one Newton step on exp(y) = u:  y1 = y0 + (u * exp(-y0) - 1), the residual formed with one fma (absolute error ~6e-8);
//   log1p(x) = y1 + c / u, with 1 / u taken from the step's own E = exp(-y0) (= (1 + 1e-6) / u; c / u <= 6e-8: no reciprocal needed).
// Absolute error <= ~1e-7 over the whole range (Y <= 16), i.e. libm-grade for what the path needs (Y enters as Y - mean);
// neither this nor libm is bit-identical to the reference's CRT log1pf (misc.c:42-45).
__device__ __forceinline__ float log1p_hw(float x)
{
   const float u = 1.0f + x;
   const float c = x - (u - 1.0f);
   float y = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
   const float E = __builtin_amdgcn_exp2f(y * -1.4426950408889634f);
   y += fmaf(u, E, -1.0f);
   return fmaf(c, E, y);
}

// The same without the Newton step: y = ln2 * v_log_f32(u) + c / u.  v_log_f32 is good to ~1 ulp of log2(u), i.e. an absolute error of
// up to ~1e-6 at Y ~ 16 (ten times log1p_hw's).  Used by the GEMM front end only (Silero v4, SPLIT16 mode), whose own deviation from a
// correctly rounded STFT moves Y by up to 4e-2 in near-silent bins: one transcendental and three more instructions less per output.
__device__ __forceinline__ float log1p_hw_fast(float x)
{
   const float u = 1.0f + x;
   const float c = x - (u - 1.0f);
   return fmaf(c, __builtin_amdgcn_rcpf(u), __builtin_amdgcn_logf(u) * 0.6931471805599453f);
}

// Split-fp16 hand-off ("h3-native"): one (tile, chunk, step) block = [hi | lo][stream-in-tile 16][unit 64] halves (4 KB, the size
// of the fp32 block), i.e. a stream's 64 units are one 128-byte row: the B fragment of v_mfma_f32_16x16x32_f16 for
// k-block kb is the 16 bytes at unit 32 kb + 8 (lane >> 4).  Index in HALVES of the hi row; the lo row is + 16 * 64.
__host__ __device__ __forceinline__ size_t lstm_xh_index(int stream, int chunk, int C, int t, int unit, int steps)
{
   return ((((size_t)(stream / kLstmTile) * C + chunk) * steps + t) * 2 * kLstmTile + (stream % kLstmTile)) * 64 + unit;
}

struct LstmWeights {
   const float *w;        // [2][256][128]  reference layout: [layer][gate*64+unit][x(64) | h(64)]
   const float *wT;       // [2][128][256]  k-major copy for the simple kernel
   const float *b;        // [2][256]
   const float *dec_w;    // [2][64]
   const float *dec_b;    // [2]
};

}  // namespace vadc
