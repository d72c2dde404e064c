// kernels_frontend_gemm.hip -- Silero v4 front end: reflect pad 96 + STFT + magnitude + log1p + bin means as an fp32 GEMM on
// v_mfma_f32_16x16x4_f32.
//
// Replaces, for the v4 model (reference arithmetic: silero_vad.py:22-66, STFT_conv with is_v4 + AdaptiveAudioNormalization;
// the reference itself runs it through onnxruntime, onnx_helpers.c:83-115): pad_reflect(96), conv1d with the [258,1,256]
// basis at hop 64 (24 frames per 1536-sample chunk), sqrt(re^2 + im^2), log1p(2^20 m) and the per-frame bin sums.
//
// Why a GEMM here and the bit-exact tree (kernels_frontend.hip) for v3.1: v3.1 parity is defined against the reference C
// backend, whose fp32 reduction tree is part of what the model sees (tools/stft_sensitivity.py).  v4
// has no C implementation in the reference -- its parity target is the PyTorch/onnxruntime convolution, any fp32 order --
// and its probabilities move by <= 1e-6 between fp32 and fp64 evaluation (tests/golden/gen_golden_v4_from_python_reference.py).
//
// REAL-INPUT FOLDING.  The basis rows are a windowed DFT: re rows are even about tap 128, im rows odd, tap 0 is zero (periodic
// Hann), and the im rows of bins 0 and 128 vanish.  vadc_amd_create VERIFIES these identities bit for bit on the loaded basis
// (engine.hip: v4_basis_is_symmetric) and otherwise keeps the tree kernel.  With them
//     re_k = sum_{n=0..127} Are[k][n] xs[n],   xs[n] = x[n] + x[256-n] (n >= 1),  xs[0] = x[128],  Are[k][0] = basis[k][128]
//     im_k = sum_{n=1..127} Aim[k][n] xd[n],   xd[n] = x[n] - x[256-n],           xd[0] = 0
// i.e. two K = 128 contractions instead of one K = 256: half the MACs of the dense conv.
//
// MAPPING.  Workgroup = 4 waves, persistent over groups of 4 chunks (96 positions = 6 MFMA column tiles).  Wave w keeps the
// A fragments of bins [32w, 32w+32) -- two re and two im row tiles, 128 VGPRs -- for the whole kernel.  The 4 chunks are
// staged twice in LDS (block pitch 68 floats: the 16 frames of a column tile hit disjoint banks): X0[i] = x[i] and X1[i] =
// x[i+1], so that both the direct taps x[p+32g+4q..+3] and the mirrored taps x[p+256-32g-4q-3..] are ALIGNED ds_read_b128.
// Lane (f = l & 15, g = l >> 4) supplies the B fragment of position f: taps 32 kb + 8 g + e of k-block kb.  re and im of
// one (bin, position) land in the same lane and register of their accumulators, so magnitude/log1p are register-local; the
// stores run along the frames of a chunk.  Bin 128 (re only) is one extra dot product on the vector ALU, taken by a different
// wave for every column tile.  Per-frame bin sums: one partial per wave = the 4 partials of FM (common.h kBinSplit).
#include "common.h"

namespace vadc {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// a = ah + al exactly to 22 bits; x = s16 / 32768 and x[n] +- x[256-n] (17 bits) split EXACTLY
__device__ __forceinline__ void split8(const float (&v)[8], h8v &hi, h8v &lo)
{
#pragma unroll
   for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
}

constexpr int kV4Pad = 96, kV4Frames = 24, kV4Padded = kChunk + 2 * kV4Pad;   // 1728 samples = 27 blocks of 64
constexpr int kGBlockPitch = 68;
constexpr int kGChunks = 4;                                                   // chunks per workgroup iteration
constexpr int kGChunkPitch = (kV4Padded / 64 + 1) * kGBlockPitch;             // 28 blocks (one spare: X1 reads sample 1728)
constexpr int kGTiles = kGChunks * kV4Frames / 16;                            // 6 column tiles of 16 positions

__device__ __forceinline__ float g_sample(float v) { return v; }
__device__ __forceinline__ float g_sample(int16_t v) { return (float)v * (1.0f / 32768.0f); }

// MATRIX PIPE.  The two K = 128 contractions run as v_mfma_f32_16x16x32_f16 with SPLIT-fp16 operands (a = ah + al, three MFMAs
// per k-block: al.bh + ah.bl + ah.bh, fp32 accumulation; see k_lstm_wavefront_h3 in kernels_lstm.hip): 48 matrix instructions
// of 16 cycles per column tile and wave instead of 128 fp32 MFMAs of 32 cycles, on the fp16 pipe that does not share lanes
// with the vector ALU.  s16 input and the folded sums split exactly; the basis keeps 22 of its 24 bits (error ~2e-7 relative,
// the level of fp32 accumulation itself).  Operand layout: lane l holds A[l & 15][8 (l >> 4) + e], B[8 (l >> 4) + e][l & 15].
// afrag: [tile 0..15 (0-7 re bins 16t.., 8-15 im)][kb 0..3][lane][8]  = A[16 t' + (lane & 15)][32 kb + 8 (lane >> 4) + e]  (fp32; split in-kernel)
// nyq:   [128] folded weights of bin 128 (re)
template <typename T>
__global__ __launch_bounds__(256, 2) void k_frontend_gemm_v4(const T *__restrict__ pcm, const float *__restrict__ afrag,
                                                             const float *__restrict__ nyq,
                                                             float *__restrict__ Y, float *__restrict__ MAG, float *__restrict__ FM,
                                                             int n_chunks, ItemMap map, size_t fm_stride)
{
   __shared__ __attribute__((aligned(16))) float X0[kGChunks * kGChunkPitch];
   __shared__ __attribute__((aligned(16))) float X1[kGChunks * kGChunkPitch];
   __shared__ __attribute__((aligned(16))) float nyq_s[128];
   __shared__ float bsum[4][kGChunks * kV4Frames];
   __shared__ float nyv[kGChunks * kV4Frames];          // log value of bin 128 per position (whichever wave computed it)
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int f = lane & 15, g = lane >> 4;

   // this wave's A fragments, split once: re tiles 2w, 2w+1 and im tiles 8+2w, 8+2w+1; [tile][kb] -> 8 halves hi + 8 halves lo
   h8v ah[4][4], al[4][4];
#pragma unroll
   for (int ti = 0; ti < 4; ++ti) {
      const int tile = (ti < 2 ? 0 : 8) + 2 * wave + (ti & 1);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         float v[8];
         const float *src = afrag + (((size_t)tile * 4 + kb) * 64 + lane) * 8;
#pragma unroll
         for (int e = 0; e < 8; ++e) v[e] = src[e];
         split8(v, ah[ti][kb], al[ti][kb]);
      }
   }
   if (tid < 128) nyq_s[tid] = nyq[tid];

   const int n_groups = (n_chunks + kGChunks - 1) / kGChunks;
#pragma unroll 1
   for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
      __syncthreads();                                   // previous iteration's readers are done
      // ---- stage 4 chunks: reflect pad 96 (no edge repeat), block pitch 68, X1 = X0 shifted by one sample ----
      for (int c = 0; c < kGChunks; ++c) {
         const int it = min(grp * kGChunks + c, n_chunks - 1);
         const T *src = pcm + (size_t)map(it) * kChunk;
         for (int idx = tid; idx < kV4Padded + 1; idx += 256) {
            int sidx = idx - kV4Pad;
            sidx = sidx < 0 ? -sidx : sidx;
            sidx = sidx >= kChunk ? 2 * (kChunk - 1) - sidx : sidx;
            const float v = (idx < kV4Padded) ? g_sample(src[sidx]) : 0.0f;
            if (idx < kV4Padded) X0[c * kGChunkPitch + (idx >> 6) * kGBlockPitch + (idx & 63)] = v;
            if (idx >= 1) { const int j = idx - 1; X1[c * kGChunkPitch + (j >> 6) * kGBlockPitch + (j & 63)] = v; }
         }
      }
      for (int i = tid; i < 4 * kGChunks * kV4Frames; i += 256) (&bsum[0][0])[i] = 0.0f;
      __syncthreads();

#pragma unroll 1
      for (int ct = 0; ct < kGTiles; ++ct) {
         // column tile ct: positions 16 ct + f of the group; position -> (chunk c, frame fr)
         const int pos = 16 * ct + f;
         const int c = pos / kV4Frames, fr = pos - c * kV4Frames;
         // k-block kb, lane group g, element e  <->  tap n = 32 kb + 8 g + e
         //   direct   x[64 fr + n]              = X0[block fr + (kb >> 1)][32 (kb & 1) + 8 g + e]
         //   mirrored x[64 fr + 256 - n]        = X1[block fr + 3 - (kb >> 1)][56 - 32 (kb & 1) - 8 g + (7 - e)]
         const float *pd = X0 + c * kGChunkPitch + fr * kGBlockPitch + 8 * g;
         const float *pm = X1 + c * kGChunkPitch + (fr + 3) * kGBlockPitch + 56 - 8 * g;
         const float center = X0[c * kGChunkPitch + (fr + 2) * kGBlockPitch];       // sample 64 fr + 128
         f4v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};       // re0, re1, im0, im1
         float ny = 0.0f;
         const bool do_ny = (ct & 3) == wave;
#pragma unroll
         for (int kb = 0; kb < 4; ++kb) {
            const float *pdk = pd + (kb >> 1) * kGBlockPitch + 32 * (kb & 1);
            const float *pmk = pm - (kb >> 1) * kGBlockPitch - 32 * (kb & 1);
            const float4 d0 = *reinterpret_cast<const float4 *>(pdk), d1 = *reinterpret_cast<const float4 *>(pdk + 4);
            const float4 m0 = *reinterpret_cast<const float4 *>(pmk), m1 = *reinterpret_cast<const float4 *>(pmk + 4);
            const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            const float mv[8] = {m1.w, m1.z, m1.y, m1.x, m0.w, m0.z, m0.y, m0.x};   // mirrored element of tap e is index 7 - e
            float xs[8], xd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xs[e] = dv[e] + mv[e]; xd[e] = dv[e] - mv[e]; }
            if (kb == 0 && g == 0) { xs[0] = center; xd[0] = 0.0f; }                // tap 0 carries the unpaired centre tap 128
            h8v sh, sl, dh, dl;
            split8(xs, sh, sl);
            split8(xd, dh, dl);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
               const h8v bh = ti < 2 ? sh : dh, bl = ti < 2 ? sl : dl;
               acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ti][kb], bh, acc[ti], 0, 0, 0);
               acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ti][kb], bl, acc[ti], 0, 0, 0);
               acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ti][kb], bh, acc[ti], 0, 0, 0);
            }
            if (do_ny) {                                   // bin 128 (re only) on the vector ALU, fp32
               const float4 w0 = *reinterpret_cast<const float4 *>(nyq_s + 32 * kb + 8 * g), w1 = *reinterpret_cast<const float4 *>(nyq_s + 32 * kb + 8 * g + 4);
               ny = fmaf(w0.x, xs[0], ny); ny = fmaf(w0.y, xs[1], ny); ny = fmaf(w0.z, xs[2], ny); ny = fmaf(w0.w, xs[3], ny);
               ny = fmaf(w1.x, xs[4], ny); ny = fmaf(w1.y, xs[5], ny); ny = fmaf(w1.z, xs[6], ny); ny = fmaf(w1.w, xs[7], ny);
            }
         }
         const f4v re0 = acc[0], re1 = acc[1], im0 = acc[2], im1 = acc[3];
         // ---- epilogue: D rows = bins 32 w + 16 j + 4 g + r, column = position f ----
         const int item = grp * kGChunks + c;
         const bool ok = item < n_chunks;
         const size_t ybase = (size_t)map(ok ? item : n_chunks - 1) * (kBins * kV4Frames) + fr;
         float part = 0.0f;
#pragma unroll
         for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
               const float re = j == 0 ? re0[r] : re1[r], im = j == 0 ? im0[r] : im1[r];
               const float mag = sqrtf(fmaf(re, re, im * im));
               const float val = log1p_hw(mag * 1048576.0f);
               const int bin = 32 * wave + 16 * j + 4 * g + r;
               if (ok) { Y[ybase + (size_t)bin * kV4Frames] = val; MAG[ybase + (size_t)bin * kV4Frames] = mag; }
               part += val;
            }
         if (do_ny) {                                      // bin 128: re only (its im row is identically zero)
            ny += __shfl_xor(ny, 16);
            ny += __shfl_xor(ny, 32);
            const float mag = fabsf(ny);
            const float val = log1p_hw(mag * 1048576.0f);
            if (g == 0) {
               if (ok) { Y[ybase + (size_t)128 * kV4Frames] = val; MAG[ybase + (size_t)128 * kV4Frames] = mag; }
               nyv[pos] = val;                             // added to partial 3 below: the sum must not depend on which wave took it
            }
         }
         part += __shfl_xor(part, 16);
         part += __shfl_xor(part, 32);
         if (g == 0) bsum[wave][pos] = part;               // one writer per (wave, position)
      }
      __syncthreads();
      // FM partial w = this wave's 32 (+ Nyquist) bins; the consumer adds the 4 partials in fixed order
      for (int i = tid; i < 4 * kGChunks * kV4Frames; i += 256) {
         const int wv = i / (kGChunks * kV4Frames), pos = i - wv * (kGChunks * kV4Frames);
         const int c = pos / kV4Frames, fr = pos - c * kV4Frames;
         const int item = grp * kGChunks + c;
         if (item < n_chunks) FM[wv * fm_stride + (size_t)map(item) * kV4Frames + fr] = (wv == 3) ? bsum[3][pos] + nyv[pos] : bsum[wv][pos];
      }
   }
}

void launch_frontend_gemm_v4_f32(const float *pcm, const float *afrag, const float *nyq, float *Y, float *MAG, float *FM, size_t fm_stride,
                                 int n, ItemMap map, int n_cus, hipStream_t st)
{
   const int groups = (n + kGChunks - 1) / kGChunks;
   const int grid = groups < 2 * n_cus ? groups : 2 * n_cus;
   hipLaunchKernelGGL(k_frontend_gemm_v4<float>, dim3(grid), dim3(256), 0, st, pcm, afrag, nyq, Y, MAG, FM, n, map, fm_stride);
}

void launch_frontend_gemm_v4_s16(const int16_t *pcm, const float *afrag, const float *nyq, float *Y, float *MAG, float *FM, size_t fm_stride,
                                 int n, ItemMap map, int n_cus, hipStream_t st)
{
   const int groups = (n + kGChunks - 1) / kGChunks;
   const int grid = groups < 2 * n_cus ? groups : 2 * n_cus;
   hipLaunchKernelGGL(k_frontend_gemm_v4<int16_t>, dim3(grid), dim3(256), 0, st, pcm, afrag, nyq, Y, MAG, FM, n, map, fm_stride);
}

}  // namespace vadc
