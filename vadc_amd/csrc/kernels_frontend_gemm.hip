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
// Lane (f = l & 15, g = l >> 4) supplies B[k = g][col = f] = xs/xd of position f at tap 32g + s for k-step s.  re and im of
// one (bin, position) land in the same lane and register of their accumulators, so magnitude/log1p are register-local; the
// stores run along the frames of a chunk.  Bin 128 (re only) is one extra dot product on the vector ALU, taken by a different
// wave for every column tile.  Per-frame bin sums: one partial per wave = the 4 partials of FM (common.h kBinSplit).
#include "common.h"

namespace vadc {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kV4Pad = 96, kV4Frames = 24, kV4Padded = kChunk + 2 * kV4Pad;   // 1728 samples = 27 blocks of 64
constexpr int kGBlockPitch = 68;
constexpr int kGChunks = 4;                                                   // chunks per workgroup iteration
constexpr int kGChunkPitch = (kV4Padded / 64 + 1) * kGBlockPitch;             // 28 blocks (one spare: X1 reads sample 1728)
constexpr int kGTiles = kGChunks * kV4Frames / 16;                            // 6 column tiles of 16 positions

__device__ __forceinline__ float g_sample(float v) { return v; }
__device__ __forceinline__ float g_sample(int16_t v) { return (float)v * (1.0f / 32768.0f); }

// afrag: [tile 0..15 (0-7 re bins 16t.., 8-15 im)][s 0..31][lane]   = A[16 t' + (lane & 15)][32 (lane >> 4) + s]
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
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int f = lane & 15, g = lane >> 4;

   // this wave's A fragments: re tiles 2w, 2w+1 and im tiles 8+2w, 8+2w+1
   float are0[32], are1[32], aim0[32], aim1[32];
#pragma unroll
   for (int s = 0; s < 32; ++s) {
      are0[s] = afrag[((size_t)(2 * wave) * 32 + s) * 64 + lane];
      are1[s] = afrag[((size_t)(2 * wave + 1) * 32 + s) * 64 + lane];
      aim0[s] = afrag[((size_t)(8 + 2 * wave) * 32 + s) * 64 + lane];
      aim1[s] = afrag[((size_t)(8 + 2 * wave + 1) * 32 + s) * 64 + lane];
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
         // direct taps 32 g + 4 q + {0..3}: sample 64 fr + 32 g + 4 q   -> block fr + (g >> 1), offset 32 (g & 1) + 4 q
         const float *pd = X0 + c * kGChunkPitch + (fr + (g >> 1)) * kGBlockPitch + 32 * (g & 1);
         // mirrored taps: samples 64 fr + 256 - 32 g - 4 q - {0..3} = X1[64 fr + 252 - 32 g - 4 q + {3,2,1,0}]
         //   252 - 32 g - 4 q stays inside block 3 - (g >> 1) for q = 0..7; offset inside the block 60 - 32 (g & 1) - 4 q
         const float *pm = X1 + c * kGChunkPitch + (fr + 3 - (g >> 1)) * kGBlockPitch + 60 - 32 * (g & 1);
         const float center = X0[c * kGChunkPitch + (fr + 2) * kGBlockPitch];       // sample 64 fr + 128
         f4v re0 = {0, 0, 0, 0}, re1 = {0, 0, 0, 0}, im0 = {0, 0, 0, 0}, im1 = {0, 0, 0, 0};
         float ny = 0.0f;
         const bool do_ny = (ct & 3) == wave;
         // operands of k-steps 4q..4q+3 are fetched one q ahead of the MFMAs that consume them
         float4 d = *reinterpret_cast<const float4 *>(pd), m = *reinterpret_cast<const float4 *>(pm);
#pragma unroll
         for (int q = 0; q < 8; ++q) {
            float xs[4], xd[4];
            xs[0] = d.x + m.w; xs[1] = d.y + m.z; xs[2] = d.z + m.y; xs[3] = d.w + m.x;
            xd[0] = d.x - m.w; xd[1] = d.y - m.z; xd[2] = d.z - m.y; xd[3] = d.w - m.x;
            if (q == 0 && g == 0) { xs[0] = center; xd[0] = 0.0f; }                 // tap 0 carries the unpaired centre tap 128
            if (q < 7) {
               d = *reinterpret_cast<const float4 *>(pd + 4 * (q + 1));
               m = *reinterpret_cast<const float4 *>(pm - 4 * (q + 1));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
               const int s = 4 * q + t;
               re0 = __builtin_amdgcn_mfma_f32_16x16x4f32(are0[s], xs[t], re0, 0, 0, 0);
               re1 = __builtin_amdgcn_mfma_f32_16x16x4f32(are1[s], xs[t], re1, 0, 0, 0);
               im0 = __builtin_amdgcn_mfma_f32_16x16x4f32(aim0[s], xd[t], im0, 0, 0, 0);
               im1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aim1[s], xd[t], im1, 0, 0, 0);
            }
         }
         if (do_ny) {                                      // bin 128 on the vector ALU: its own pass over the operands, so that its
#pragma unroll                                             // LDS reads do not serialise the MFMA loop's prefetch
            for (int q = 0; q < 8; ++q) {
               const float4 dq = *reinterpret_cast<const float4 *>(pd + 4 * q), mq = *reinterpret_cast<const float4 *>(pm - 4 * q);
               const float4 wq = *reinterpret_cast<const float4 *>(nyq_s + 32 * g + 4 * q);
               const float x0 = (q == 0 && g == 0) ? center : dq.x + mq.w;
               ny = fmaf(wq.x, x0, ny); ny = fmaf(wq.y, dq.y + mq.z, ny); ny = fmaf(wq.z, dq.z + mq.y, ny); ny = fmaf(wq.w, dq.w + mq.x, ny);
            }
         }
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
               part += val;
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
         if (item < n_chunks) FM[wv * fm_stride + (size_t)map(item) * kV4Frames + fr] = bsum[wv][pos];
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
