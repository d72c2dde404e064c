// kernels_frontend_gemm.hip -- the STFT front end as a folded GEMM on the fp16 matrix pipe with split-fp16 operands:
// reflect pad + STFT + magnitude + log1p + per-frame bin sums.
//
// Used for (a) the Silero v4 model (default front end there; reference arithmetic: silero_vad.py:22-66, STFT_conv with is_v4 +
// AdaptiveAudioNormalization; the reference itself runs it through onnxruntime, onnx_helpers.c:83-115): pad_reflect(96), conv1d
// with the [258,1,256] basis at hop 64 (24 frames per 1536-sample chunk), sqrt(re^2 + im^2), log1p(2^20 m), bin sums; and
// (b) Silero v3.1 in the engine's FAST_STFT precision mode (throughput mode; replaces tensor.h:912-958, stft.c:15-224,
// misc.c:40-63 with pad 128 and 25 frames).
//
// Why a GEMM is the default for v4 but an opt-in precision mode for v3.1: v3.1 parity is defined against the reference C
// backend, whose fp32 reduction tree is part of what the model sees (tests/reports/stft_sensitivity.py), so the
// fp32 mode keeps the bit-exact tree (kernels_frontend.hip).  v4 has no C implementation in the reference -- its parity target
// is the PyTorch/onnxruntime convolution, any fp32 order -- and its probabilities move by <= 1e-6 between fp32 and fp64
// evaluation (tests/golden/gen_golden_v4_from_python_reference.py).
//
// REAL-INPUT FOLDING.  The basis rows are a windowed DFT: re rows are even about tap 128, im rows odd, tap 0 is zero (periodic
// Hann), and the im rows of bins 0 and 128 vanish.  vadc_amd_create VERIFIES these identities bit for bit on the loaded basis
// (engine_weights.hip) and otherwise keeps the tree kernel.  With them
//     re_k = sum_{n=0..127} Are[k][n] xs[n],   xs[n] = x[n] + x[256-n] (n >= 1),  xs[0] = x[128],  Are[k][0] = basis[k][128]
//     im_k = sum_{n=1..127} Aim[k][n] xd[n],   xd[n] = x[n] - x[256-n],           xd[0] = 0
// i.e. two K = 128 contractions instead of one K = 256: half the MACs of the dense conv.
//
// MATRIX PIPE.  The contractions run as v_mfma_f32_16x16x32_f16 with SPLIT-fp16 operands (a = ah + al, three MFMAs per k-block:
// al.bh + ah.bl + ah.bh, fp32 accumulation; see k_lstm_wavefront_h3 in kernels_lstm.hip) on the fp16 pipe, which does not
// share lanes with the vector ALU.  s16 input and the folded sums split exactly; the basis keeps 22 of its 24 bits (error
// ~2e-7 relative, the level of fp32 accumulation itself).  Operand layout: lane l holds A[l & 15][8 (l >> 4) + e],
// B[8 (l >> 4) + e][l & 15].
//
// MAPPING.  Workgroup = 8 waves, persistent over groups of G chunks (v4: 4 x 24 positions = 6 column tiles of 16; v3.1: 5 x 25 = 8 tiles).
// Wave w keeps the split A fragments of bins [16w, 16w+16) -- one re and one im row tile, 64 VGPRs -- for the whole kernel
// (4 waves with 32 bins each need 128 VGPRs and leave two waves per SIMD; measured the two layouts are within 4 % of each
// other -- the kernel is bound by the vector-ALU epilogue, ~20 instructions per output with four transcendentals, not by
// latency).
// The group's chunks are staged once in LDS as fp32 (16-byte global loads, block pitch 68 floats: the 16 frames of a column tile
// hit disjoint banks).  The B operand of a column tile -- folded sums/differences, split into fp16 hi/lo -- is the same for all
// waves, so it is PREPARED ONCE: wave w folds k-block w % 4 of the next tile -- waves 0-3 the sums (re operand), waves 4-7 the
// differences (im operand) -- and splits it into an LDS fragment buffer (double buffered, one barrier per tile) while the matrix
// pipe works on the current one, and every wave then fetches its 16 fragments with conflict-free ds_read_b128.  re and im of one (bin, position) land in the same lane and register of their
// accumulators, so magnitude/log1p are register-local; the stores run along the frames of a chunk.  Bin 128 (re only) is a
// K = 128 dot product on the vector ALU: the wave that prepares the sums of k-block kb accumulates that block's share (reduced
// over the four lane groups with two shuffles), a rotating wave adds the four shares in fixed order.  Per-frame bin sums: one
// partial per wave; waves 2p and 2p+1 make up partial p of FM (common.h kBinSplit), added in that order.
#include "common.h"
#include <cstdio>

namespace vadc {

#ifdef VADC_PHASE_PROF
__device__ unsigned long long g_gemm_phase[8];
__device__ unsigned int g_gemm_groups;
#define GPH(i) do { if (gph_on) { const unsigned long long t_ = __builtin_readcyclecounter(); g_gemm_phase[i] += t_ - gph_t; gph_t = t_; } } while (0)
#else
#define GPH(i) do { } while (0)
#endif

typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// a = ah + al exactly to 22 bits; x = s16 / 32768 and x[n] +- x[256-n] (17 bits) split EXACTLY
__device__ __forceinline__ void split8(const float (&v)[8], h8v &hi, h8v &lo)
{
#pragma unroll
   for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
}

constexpr int kGBlockPitch = 68;

// GEO 0: Silero v3.1 (reflect pad 128, 25 frames, 28 blocks; Y + FM);  GEO 1: Silero v4 (pad 96, 24 frames, 27 blocks; Y + MAG + FM)
template <int GEO> struct GemmGeo;
// chunks = chunks per workgroup iteration: 5 x 25 = 125 positions fill 8 column tiles of 16 to 98 % (4 x 25 = 100 fill 7 to 89 %), 4 x 24 = 96 fill 6
template <> struct GemmGeo<0> { static constexpr int samples = 1536, pad = 128, frames = 25, blocks = 28, chunks = 5; static constexpr bool mag = false; };
template <> struct GemmGeo<1> { static constexpr int samples = 1536, pad = 96, frames = 24, blocks = 27, chunks = 4; static constexpr bool mag = true; };
// Silero v4 with 1024- and 512-sample windows (onnx_helpers.c:164-170: the v4 graph takes 512 ... 1536 samples; frames = samples / 64)
template <> struct GemmGeo<2> { static constexpr int samples = 1024, pad = 96, frames = 16, blocks = 19, chunks = 6; static constexpr bool mag = true; };
template <> struct GemmGeo<3> { static constexpr int samples = 512, pad = 96, frames = 8, blocks = 11, chunks = 12; static constexpr bool mag = true; };
// the 8 kHz branch of the v4 graph (same basis, same hop): 768- and 256-sample windows (512 is GEO 3)
template <> struct GemmGeo<4> { static constexpr int samples = 768, pad = 96, frames = 12, blocks = 15, chunks = 8; static constexpr bool mag = true; };
template <> struct GemmGeo<5> { static constexpr int samples = 256, pad = 96, frames = 4, blocks = 7, chunks = 16; static constexpr bool mag = true; };
// Silero v4, 16 kHz, the 1280-sample window (round 5; its 768-sample window is GEO 4): 4 x 20 = 80 positions = 5 column tiles of 16
template <> struct GemmGeo<6> { static constexpr int samples = 1280, pad = 96, frames = 20, blocks = 23, chunks = 4; static constexpr bool mag = true; };

__device__ __forceinline__ void g_stage8(const float *src, float (&v)[8])
{
   const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
   v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void g_stage8(const int16_t *src, float (&v)[8])
{
   const uint4 r = *reinterpret_cast<const uint4 *>(src);                       // 8 samples; chunks are 3072-byte aligned
   const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
   for (int i = 0; i < 4; ++i) {
      v[2 * i] = (float)(int16_t)(w[i] & 0xffffu) * (1.0f / 32768.0f);          // exact
      v[2 * i + 1] = (float)(int16_t)(w[i] >> 16) * (1.0f / 32768.0f);
   }
}
__device__ __forceinline__ float g_sample(float v) { return v; }
__device__ __forceinline__ float g_sample(int16_t v) { return (float)v * (1.0f / 32768.0f); }

// afrag: [tile 0..15 (0-7 re bins 16t.., 8-15 im)][kb 0..3][lane][8]  = A[16 t' + (lane & 15)][32 kb + 8 (lane >> 4) + e]  (fp32; split in-kernel)
// nyq:   [128] folded weights of bin 128 (re)
template <typename T, int GEO>
__global__ __launch_bounds__(512, 4) void k_frontend_gemm(const T *__restrict__ pcm, const float *__restrict__ afrag,
                                                          const float *__restrict__ nyq,
                                                          float *__restrict__ Y, float *__restrict__ MAG, float *__restrict__ FM,
                                                          int n_chunks, ItemMap map, size_t fm_stride, int nrt)
{
   // nrt <= Geo::samples: the samples a chunk REALLY has (= its stride in pcm).  Silero v4 at a window that is no multiple of 256 samples runs the next larger built
   // geometry: the chunk's samples are staged where they always are, the right reflect pad goes behind sample nrt - 1 (silero_vad.py:30), and frames 0 .. nrt / 64 - 1
   // are that window's frames bit for bit; what lies behind in the staging buffer is stale, the surplus frames it feeds are masked by the stages (k_layer_mfma `tv`).
   typedef GemmGeo<GEO> Geo;
   constexpr int kPadG = Geo::pad, kFr = Geo::frames, kBlk = Geo::blocks, kGChunks = Geo::chunks;
   constexpr int kChunk = Geo::samples;                                         // samples per chunk (shadows the v3.1 constant of common.h)
   constexpr int kPaddedG = kChunk + 2 * kPadG;
   constexpr int kChunkPitch = (kBlk + 1) * kGBlockPitch;                       // one spare block: the mirror of tap 0 is read (unused)
   constexpr int kPos = kGChunks * kFr;                                         // positions per group (96 | 125)
   constexpr int kTiles = (kPos + 15) / 16;                                     // 6 | 8 column tiles
   constexpr int kPosPad = kTiles * 16;

   __shared__ __attribute__((aligned(16))) float X0[kGChunks * kChunkPitch];
   __shared__ __attribute__((aligned(16))) _Float16 Bf[2][4][4][64][8];         // [buffer][kb][sh, sl, dh, dl][lane][8]: 32 KB
   __shared__ __attribute__((aligned(16))) float nyq_s[128];
   __shared__ __attribute__((aligned(16))) float nyp[2][16][20];               // bin 128: [buffer][position][4 kb + lane group] shares (row pitch 20: conflict-free)
   __shared__ float bsum[8][kPosPad];
   __shared__ float nyv[kPosPad];                                              // log value of bin 128 per position
   __shared__ int2 ptab[kPosPad];                                              // position -> {offset of its first block in X0, chunk | frame << 8 | valid << 16}
   __shared__ int crow[kGChunks];                                              // this group's chunks -> row of the outputs (map), -1 past the end
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   const int f = lane & 15, g = lane >> 4;

   // this wave's A fragments, split once: re tile w and im tile 8 + w; [tile][kb] -> 8 halves hi + 8 halves lo
   h8v ah[2][4], al[2][4];
#pragma unroll
   for (int ti = 0; ti < 2; ++ti) {
      const int tile = (ti < 1 ? 0 : 8) + wave;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
         float v[8];
         g_stage8(afrag + (((size_t)tile * 4 + kb) * 64 + lane) * 8, v);
         split8(v, ah[ti][kb], al[ti][kb]);
      }
   }
   if (tid < 128) nyq_s[tid] = nyq[tid];
   if (tid < kPosPad) {
      const int pos = min(tid, kPos - 1), c = pos / kFr, fr = pos - c * kFr;
      ptab[tid] = make_int2(c * kChunkPitch + fr * kGBlockPitch, c | (fr << 8) | (tid < kPos ? 1 << 16 : 0));
   }
   if (tid < kGChunks) X0[tid * kChunkPitch + kBlk * kGBlockPitch] = 0.0f;       // the spare sample (index kPaddedG) that tap 0's mirror touches

   // fold + split k-block `wave & 3` of column tile ct into fragment buffer ct & 1: waves 0-3 the sums, waves 4-7 the differences
   auto prepare = [&](int ct) {
      const int kb = wave & 3;
      const bool sums = wave < 4;
      // tap n = 32 kb + 8 g + e:  direct x[64 fr + n] = block fr + (kb >> 1), offset 32 (kb & 1) + 8 g + e;  mirrored x[64 fr + 256 - n]
      const float *row = X0 + ptab[16 * ct + f].x;
      const float *pd = row + (kb >> 1) * kGBlockPitch + 32 * (kb & 1) + 8 * g;
      const float4 d0 = *reinterpret_cast<const float4 *>(pd), d1 = *reinterpret_cast<const float4 *>(pd + 4);
      const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      // mirrored taps x[m0 - e], m0 = 256 - 32 kb - 8 g (a multiple of 8): x[m0 - 8 .. m0 - 1] is one aligned 8-float group inside a block
      // (two 16-byte reads), x[m0] one more float that may sit in the next block
      const int m0 = 256 - 32 * kb - 8 * g;
      const float *pm = row + ((m0 - 8) >> 6) * kGBlockPitch + ((m0 - 8) & 63);
      const float4 ma = *reinterpret_cast<const float4 *>(pm), mb = *reinterpret_cast<const float4 *>(pm + 4);
      const float mv[8] = {row[(m0 >> 6) * kGBlockPitch + (m0 & 63)], mb.w, mb.z, mb.y, mb.x, ma.w, ma.z, ma.y};
      const float sgn = sums ? 1.0f : -1.0f;                                    // wave-uniform; mv * +-1 is exact, so the fma is the plain sum / difference
      float xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = fmaf(mv[e], sgn, dv[e]);
      {                                                                         // tap 0 carries the unpaired centre tap 128 (branch-free)
         const float ctr = row[2 * kGBlockPitch];
         xv[0] = (kb == 0 && g == 0) ? (sums ? ctr : 0.0f) : xv[0];
      }
      h8v vh, vl;
      split8(xv, vh, vl);
      const int buf = ct & 1;
      *reinterpret_cast<h8v *>(&Bf[buf][kb][sums ? 0 : 2][lane][0]) = vh;
      *reinterpret_cast<h8v *>(&Bf[buf][kb][sums ? 1 : 3][lane][0]) = vl;
      if (sums) {                                                              // wave-uniform
         // bin 128 (re only): this k-block's share, fp32 on the vector ALU
         const float4 w0 = *reinterpret_cast<const float4 *>(nyq_s + 32 * kb + 8 * g), w1 = *reinterpret_cast<const float4 *>(nyq_s + 32 * kb + 8 * g + 4);
         float ny = w0.x * xv[0];
         ny = fmaf(w0.y, xv[1], ny); ny = fmaf(w0.z, xv[2], ny); ny = fmaf(w0.w, xv[3], ny);
         ny = fmaf(w1.x, xv[4], ny); ny = fmaf(w1.y, xv[5], ny); ny = fmaf(w1.z, xv[6], ny); ny = fmaf(w1.w, xv[7], ny);
         nyp[buf][f][4 * kb + g] = ny;
      }
   };

   const int n_groups = (n_chunks + kGChunks - 1) / kGChunks;
#ifdef VADC_PHASE_PROF
   const bool gph_on = blockIdx.x == 3 && tid == 0;
   unsigned long long gph_t = __builtin_readcyclecounter();
#endif
#pragma unroll 1
   for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
#ifdef VADC_PHASE_PROF
      if (gph_on) g_gemm_groups += 1;
#endif
      GPH(0);
      __syncthreads();                                   // previous iteration's readers are done
      // ---- stage the group's chunks: reflect pad (no edge repeat), block pitch 68 ----
      for (int i = tid; i < kGChunks * (kChunk / 8); i += 512) {
         const int c = i / (kChunk / 8), q = i - c * (kChunk / 8);
         const int it = min(grp * kGChunks + c, n_chunks - 1);
         if (8 * q >= nrt) continue;
         float v[8];
         g_stage8(pcm + (size_t)map(it) * nrt + 8 * q, v);
         const int p = kPadG + 8 * q;
         float *dst = X0 + c * kChunkPitch + (p >> 6) * kGBlockPitch + (p & 63);
         *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
         *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      if (tid < kGChunks) crow[tid] = grp * kGChunks + tid < n_chunks ? (int)map(grp * kGChunks + tid) : -1;
      for (int i = tid; i < kGChunks * 2 * kPadG; i += 512) {
         const int c = i / (2 * kPadG), j = i - c * (2 * kPadG);
         const int it = min(grp * kGChunks + c, n_chunks - 1);
         const T *src = pcm + (size_t)map(it) * nrt;
         const int p = j < kPadG ? j : nrt + j;                                // padded index: left pad | right pad
         const int sidx = j < kPadG ? kPadG - j : 2 * (nrt - 1) - (p - kPadG);
         X0[c * kChunkPitch + (p >> 6) * kGBlockPitch + (p & 63)] = g_sample(src[sidx]);
      }
      static_assert(kPaddedG == kBlk * 64, "padded chunk must be whole blocks");
      __syncthreads();
      GPH(1);
      prepare(0);
      __syncthreads();
      GPH(2);

#pragma unroll 1
      for (int ct = 0; ct < kTiles; ++ct) {
         GPH(3);
         // column tile ct: positions 16 ct + f of the group; position -> (chunk c, frame fr)
         const int buf = ct & 1;
         const int pos = 16 * ct + f;
         f4v acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};       // re, im
#pragma unroll
         for (int kb = 0; kb < 4; ++kb) {
            const h8v sh = *reinterpret_cast<const h8v *>(&Bf[buf][kb][0][lane][0]), sl = *reinterpret_cast<const h8v *>(&Bf[buf][kb][1][lane][0]);
            const h8v dh = *reinterpret_cast<const h8v *>(&Bf[buf][kb][2][lane][0]), dl = *reinterpret_cast<const h8v *>(&Bf[buf][kb][3][lane][0]);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
               const h8v bh = ti < 1 ? sh : dh, bl = ti < 1 ? sl : dl;
               acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ti][kb], bh, acc[ti], 0, 0, 0);
               acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ti][kb], bl, acc[ti], 0, 0, 0);
               acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ti][kb], bh, acc[ti], 0, 0, 0);
            }
         }
         if (ct + 1 < kTiles) prepare(ct + 1);           // into the other buffer (its last readers passed the previous barrier), while the matrix pipe works
#ifdef VADC_PHASE_PROF
         asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][3]));
#endif
         GPH(4);
         // ---- epilogue: D rows = bins 16 w + 4 g + r, column = position f ----
         const int pcf = ptab[pos].y, orow = crow[pcf & 255];
         const bool ok = (pcf >> 16) != 0 && orow >= 0;
         const size_t ybase = (size_t)max(orow, 0) * (kBins * kFr) + ((pcf >> 8) & 31);
         float mag[4], val[4];
#pragma unroll
         for (int r = 0; r < 4; ++r) {
            const float re = acc[0][r], im = acc[1][r];
            mag[r] = __builtin_amdgcn_sqrtf(fmaf(re, re, im * im));             // v_sqrt_f32 (1 ulp): this front end is not the bit-exact one
            val[r] = log1p_hw_fast(mag[r] * 1048576.0f);
         }
         if (ok) {                                                              // rows = bins 16 wave + 4 g + r: one address, constant offsets
            float *yp = Y + ybase + (size_t)((16 * wave + 4 * g) * kFr);
#pragma unroll
            for (int r = 0; r < 4; ++r) yp[r * kFr] = val[r];
            if (Geo::mag && MAG) {
               float *mp = MAG + ybase + (size_t)((16 * wave + 4 * g) * kFr);
#pragma unroll
               for (int r = 0; r < 4; ++r) mp[r * kFr] = mag[r];
            }
         }
         float part = ((val[0] + val[1]) + val[2]) + val[3];
         if (4 + (ct & 3) == wave && g == 0) {             // bin 128: re only (its im row is identically zero); the 16 shares added in fixed order by one of the
            const float4 *np = reinterpret_cast<const float4 *>(&nyp[buf][f][0]);   // waves that fold differences (they have no share to compute)
            const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            const float ny = (((n0.x + n0.y) + (n0.z + n0.w)) + ((n1.x + n1.y) + (n1.z + n1.w))) +
                             (((n2.x + n2.y) + (n2.z + n2.w)) + ((n3.x + n3.y) + (n3.z + n3.w)));
            const float nmag = fabsf(ny);
            const float nval = log1p_hw_fast(nmag * 1048576.0f);
            if (ok) {
               Y[ybase + (size_t)128 * kFr] = nval;
               if (Geo::mag && MAG) MAG[ybase + (size_t)128 * kFr] = nmag;
            }
            nyv[pos] = nval;                               // added to partial 3 below: the sum must not depend on which wave took it
         }
         part += __shfl_xor(part, 16);
         part += __shfl_xor(part, 32);
         if (g == 0) bsum[wave][pos] = part;               // one writer per (wave, position)
         GPH(5);
         __syncthreads();
         GPH(6);                                  // fragment buffer ct & 1 and nyp[ct & 1] are free again; ct + 1 is ready
      }
      // FM partial p = the 32 bins of waves 2p, 2p+1 (+ Nyquist for p = 3); the consumer adds the 4 partials in fixed order
      for (int i = tid; i < 4 * kPos; i += 512) {
         const int wv = i / kPos, pos = i - wv * kPos;
         const int c = pos / kFr, fr = pos - c * kFr;
         const int orow = crow[c];
         const float v = bsum[2 * wv][pos] + bsum[2 * wv + 1][pos];
         if (orow >= 0) FM[wv * fm_stride + (size_t)orow * kFr + fr] = (wv == 3) ? v + nyv[pos] : v;
      }
   }
}

#ifdef VADC_PHASE_PROF
extern "C" void vadc_gemm_phase_report(void)
{
   unsigned long long h[8]; unsigned int n;
   (void)hipDeviceSynchronize();
   (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_phase), sizeof(h));
   (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_gemm_groups), sizeof(n));
   if (!n) return;
   const char *names[7] = {"FM write-out / loop", "staging", "prepare(0)+barrier", "prepare(ct+1)", "B reads + MFMAs", "epilogue", "barrier"};
   printf("k_frontend_gemm, %u groups (cycles per group):", n);
   for (int i = 0; i < 7; ++i) printf("  %s %.0f", names[i], (double)h[i] / n);
   printf("\n");
}
#endif

template <typename T, int GEO>
static void launch_gemm_geo(const T *pcm, const float *afrag, const float *nyq, float *Y, float *MAG, float *FM, size_t fm_stride, int n, ItemMap map,
                            int n_cus, hipStream_t st, int nrt)
{
   const int groups = (n + GemmGeo<GEO>::chunks - 1) / GemmGeo<GEO>::chunks;
   const int grid = groups < 2 * n_cus ? groups : 2 * n_cus;
   hipLaunchKernelGGL((k_frontend_gemm<T, GEO>), dim3(grid), dim3(512), 0, st, pcm, afrag, nyq, Y, MAG, FM, n, map, fm_stride,
                      (nrt > 0 && nrt < GemmGeo<GEO>::samples) ? nrt : GemmGeo<GEO>::samples);
}
template <typename T>
static void launch_gemm(const T *pcm, const float *afrag, const float *nyq, float *Y, float *MAG, float *FM, size_t fm_stride, int n, ItemMap map,
                        int n_cus, hipStream_t st, int geo, int nrt)
{
   switch (geo) {
   case 1:  launch_gemm_geo<T, 1>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, nrt); break;
   case 2:  launch_gemm_geo<T, 2>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, nrt); break;
   case 3:  launch_gemm_geo<T, 3>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, nrt); break;
   case 4:  launch_gemm_geo<T, 4>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, nrt); break;
   case 5:  launch_gemm_geo<T, 5>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, nrt); break;
   case 6:  launch_gemm_geo<T, 6>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, nrt); break;
   default: launch_gemm_geo<T, 0>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, 0); break;
   }
}

// geo: 0 = Silero v3.1 geometry (MAG unused), 1 / 2 / 3 / 4 / 5 = Silero v4 with 1536- / 1024- / 512- / 768- / 256-sample windows
// nrt: samples per chunk in pcm when fewer than the geometry's (0: the geometry's)
void launch_frontend_gemm_f32(const float *pcm, const float *afrag, const float *nyq, float *Y, float *MAG, float *FM, size_t fm_stride,
                              int n, ItemMap map, int n_cus, hipStream_t st, int geo, int nrt)
{
   launch_gemm<float>(pcm, afrag, nyq, Y, MAG, FM, fm_stride, n, map, n_cus, st, geo, nrt);
}

// (s16 input is k_frontend_gemm2's: kernels_frontend_gemm2.hip.  This form serves f32 input -- the host's backend_run samples, the stage taps.)

}  // namespace vadc
