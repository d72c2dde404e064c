// kernels_frontend_gemm2.hip -- the GEMM STFT front end, second form: v_mfma_f32_32x32x16_f16, one persistent 8-wave workgroup per CU, every stage of a
// column tile (fold + split of the input, the matrix products, magnitude / log1p / stores, per-frame bin sums) software-pipelined ACROSS tiles inside each wave.
//
// Same arithmetic contract as kernels_frontend_gemm.hip (reference: silero_vad.py:22-66 STFT_conv + AdaptiveAudioNormalization for Silero v4, reached by the
// reference through onnxruntime, onnx_helpers.c:83-115; for Silero v3.1 only in the FAST_STFT throughput mode, replacing tensor.h:912-958, stft.c:15-224,
// misc.c:40-63): reflect pad, conv1d with the [258,1,256] basis at hop 64, sqrt(re^2 + im^2), log1p(2^20 m), four partial per-frame bin sums.  s16 input only
// (the f32 entry points keep the first form).
//
// Why a second form.  Round 5's counters on the first form (profiles/r05/bench_v4_4096x16_pmc_compute.json): matrix pipe 28 % busy, waves parked in
// s_waitcnt / s_barrier 55 % of their cycles, 29 % of the LDS cycles bank conflicts.  Its phases -- fold, MFMA burst, epilogue, barrier -- run in lockstep over
// the eight waves of a workgroup, so the matrix pipe idles during the vector phases and the vector ALU during the bursts; and a 16x16x32 MFMA holds the SIMD's
// vector issue for 8 of its 16 cycles, which leaves too few issue slots for the ~110 vector instructions a wave needs per tile.  Here:
//
// * 32x32x16 MFMAs: the same FLOP per pipe cycle, half the instructions, 24 of 32 cycles free for vector issue.  A wave owns 32 rows of ONE kind (waves 0-3:
//   re of bins 32 w .. 32 w + 31, waves 4-7: im of the same bins; 64 registers of split A fragments) x 32 positions per tile: 24 MFMAs per tile on one accumulator.
// * REAL-INPUT FOLD as in the first form, but with slot j = tap j + 1:  s_j = x[j + 1] + x[255 - j],  d_j = x[j + 1] - x[255 - j],  j = 0 .. 127.  Slot 127 pairs
//   the centre tap with itself (its re weight is halved, its im weight is zero), tap 0's weight is zero (periodic Hann) -- no special cases in the kernel.
// * The input stays s16 in LDS (block pitch 72 halves, chunk pitch = 72 F mod 128: every 16-byte read conflict-free, also across chunk boundaries).  A fold value is a
//   17-bit integer: hi = fp16 round-toward-zero (never overflows: 65,536 -> 65,504 + 32), lo = value - hi exactly.  Scales are powers of two carried to the logarithm's
//   argument: A = 256 x basis, B = integer samples => accumulators = 2^23 x (re, im).
// * The wave pair (w, w + 4) exchanges HALF of its accumulators through LDS: each finalises 16 of the pair's 32 bins for all 32 positions (im rows are stored rotated
//   by 16 so that both keep registers 0-7 and send 8-15): magnitude, log1p, stores and bin sums are spread evenly over the 8 waves, and so is the fold (wave v folds
//   k block v of the NEXT tile for everybody).
// * Pipeline per iteration t of a wave (ONE barrier):  MFMAs of tile t  |  fold + split of tile t + 1  |  finalisation of tile t - 1  |  bin sums of tile t - 2  |
//   staging of a later group's samples (global -> registers -> LDS; its reflect pads LDS -> LDS).  Everything between two barriers is one basic block; masked
//   stores are buffer stores with an out-of-range offset.
#include "common.h"

#ifndef VADC_G2_VALU_PER_MFMA
#define VADC_G2_VALU_PER_MFMA 0
#endif

namespace vadc {

typedef float g2_f16v __attribute__((ext_vector_type(16)));
typedef float g2_f4v __attribute__((ext_vector_type(4)));
typedef _Float16 g2_h8v __attribute__((ext_vector_type(8)));
typedef _Float16 g2_h2v __attribute__((ext_vector_type(2)));
typedef float g2_f2v __attribute__((ext_vector_type(2)));
typedef unsigned g2_u4v __attribute__((ext_vector_type(4)));
typedef unsigned g2_u2v __attribute__((ext_vector_type(2)));

template <int GEO> struct G2Geo;
// GEO as in kernels_frontend_gemm.hip: 0 = Silero v3.1 (pad 128, 25 frames), 1 / 2 / 3 = Silero v4 16 kHz with 1536- / 1024- / 512-sample windows, 4 / 5 = the 8 kHz
// branch's 768- / 256-sample windows.  chunks = chunks per group: 96 positions = 3 column tiles of 32 (v3.1: 125 of 128).
template <> struct G2Geo<0> { static constexpr int samples = 1536, pad = 128, frames = 25, chunks = 5; };
template <> struct G2Geo<1> { static constexpr int samples = 1536, pad = 96, frames = 24, chunks = 4; };
template <> struct G2Geo<2> { static constexpr int samples = 1024, pad = 96, frames = 16, chunks = 6; };
template <> struct G2Geo<3> { static constexpr int samples = 512, pad = 96, frames = 8, chunks = 12; };
template <> struct G2Geo<4> { static constexpr int samples = 768, pad = 96, frames = 12, chunks = 8; };
template <> struct G2Geo<5> { static constexpr int samples = 256, pad = 96, frames = 4, chunks = 24; };
template <> struct G2Geo<6> { static constexpr int samples = 1280, pad = 96, frames = 20, chunks = 8; };      // Silero v4 16 kHz, 1280-sample window: 160 positions = 5 tiles

constexpr int kG2BlockPitch = 72;                 // halves per 64-sample block of the padded chunk
constexpr unsigned kG2Oob = 0x80000000u;          // OR-ed into the buffer-store offset of a masked lane (>= num_records: the store is dropped)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g2_rsrc(const void *p)
{
   return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void g2_store(float v, __amdgpu_buffer_rsrc_t r, unsigned off)
{
   __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
// x + (x of lane ^ 32) in every lane
__device__ __forceinline__ float g2_sum_halves(float x)
{
   // Inline asm, because __builtin_amdgcn_permlane32_swap(x, x) ties both operands to one register (enc_regs_prims.h) -- and therefore with its own wait states:
   // hipcc's hazard recogniser does not look into asm text, and a swap issued right behind the v_mov that makes its operand read the register's OLD value
   // (round 5: bin 128 of a workgroup's first tile came out wrong, where the copy and the swap were adjacent; two instructions apart they were right).
   float a = x, b = x;
   asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));      // a = {x[0..31], x[0..31]}, b = {x[32..63], x[32..63]}
   return a + b;
}
__device__ __forceinline__ int g2_sext_lo(unsigned w) { return (int)(short)(w & 0xffffu); }
__device__ __forceinline__ int g2_sext_hi(unsigned w) { return (int)w >> 16; }

// two fold values (17-bit integers as floats) -> packed (hi, lo) halves: hi = round toward zero, lo = v - hi (exact)
__device__ __forceinline__ void g2_split2(float a, float b, g2_h2v &hi, g2_h2v &lo)
{
   hi = __builtin_bit_cast(g2_h2v, __builtin_amdgcn_cvt_pkrtz(a, b));
   float ra, rb;
   asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
   asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
   const g2_f2v r = {ra, rb};
   lo = __builtin_convertvector(r, g2_h2v);
}
// A fragments keep 22 bits: hi = round to nearest, lo = the rest
__device__ __forceinline__ void g2_split8_rn(const float (&v)[8], g2_h8v &hi, g2_h8v &lo)
{
#pragma unroll
   for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
}

// afrag2: [tile 0..7 (0-3: re bins 32 t + r, 4-7: im bins 32 (t - 4) + (r + 16) % 32)][kb 0..7][lane][8] = 256 x A[row lane & 31][slot 16 kb + 8 (lane >> 5) + e]
// nyq2:   [128] folded weights of bin 128 (re) by slot
// ABL: timing-only ablations for tools/gemm_bench (results WRONG): 1 no Y stores, 2 no fold arithmetic, 4 no MFMAs, 8 no staging, 16 no sqrt / log, 32 no accumulator exchange
template <int GEO, bool WMAG, int ABL = 0>
__global__ __launch_bounds__(512, 2) void k_frontend_gemm2(const int16_t *__restrict__ pcm, const float *__restrict__ afrag2, const float *__restrict__ nyq2,
                                                           float *__restrict__ Y, float *__restrict__ MAG, float *__restrict__ FM,
                                                           int n_chunks, ItemMap map, size_t fm_stride, int nrt)
{
   // nrt <= Geo::samples: the samples a chunk really has (= its stride in pcm) -- a Silero v4 window between the built ones, run in the next larger geometry: see
   // k_frontend_gemm.  nrt = Geo::samples at the built windows: the address arithmetic below is what it was.
   typedef G2Geo<GEO> Geo;
   constexpr int S = Geo::samples, kPadG = Geo::pad, F = Geo::frames, G = Geo::chunks;
   constexpr int kPadded = S + 2 * kPadG, kBlk = kPadded / 64;
   static_assert(kPadded % 64 == 0 && kPadG % 8 == 0, "geometry");
   constexpr int kPos = G * F, kTiles = (kPos + 31) / 32;
   constexpr int kCPraw = kBlk * kG2BlockPitch;
   constexpr int kCP = kCPraw + ((((kG2BlockPitch * F - kCPraw) % 128) + 128) % 128);          // chunk pitch (halves) = 72 F mod 128: bank position linear in the position index
   static_assert(kCP % 8 == 0 && kTiles >= 2, "geometry");
   constexpr int kX0 = G * kCP;                                                              // halves per staging buffer
   constexpr int kMainU = G * (S / 8);                                                       // 16-byte pieces of a group's samples
   constexpr int kMainParts = kTiles - 1;
   constexpr int kPerPart = (kMainU + kMainParts - 1) / kMainParts;
   static_assert(kPerPart <= 512, "one 16-byte piece per thread and staging part");
   constexpr int kPads = G * 2 * kPadG;

   __shared__ __attribute__((aligned(16))) int16_t X0[2][kX0];
   __shared__ __attribute__((aligned(16))) _Float16 Bf[2][4][8][64][8];        // [buffer][s hi, s lo, d hi, d lo][kb][lane][8]: 2 x 32 KB
   __shared__ __attribute__((aligned(16))) float Ex[2][8][2][64][4];           // [buffer][wave][register quad][lane][4]: the accumulator half a wave hands to its partner
   __shared__ float Sx[2][4][32];                                              // im waves' bin sums of a tile, by position
   __shared__ float Ny[4][8][32];                                              // bin 128: the eight k blocks' shares of a tile, by position
   __shared__ int2 ptab[kTiles * 32];                                          // position of a group -> {offset of its window in a staging buffer, chunk | frame << 8 | valid << 16}
   __shared__ int crow[4][G];                                                  // group (mod 4) -> output rows of its chunks, -1 past the end

   const int tid = threadIdx.x, lane = tid & 63, v = __builtin_amdgcn_readfirstlane(tid >> 6);      // v is wave-uniform: say so (buffer resources built from it stay in SGPRs)
   const int role = v >> 2, w = v & 3, j = lane & 31, h = lane >> 5;

   // ---- this wave's A fragments, split once: 8 k blocks x (hi, lo) = 64 registers
   g2_h8v ah[8], al[8];
#pragma unroll
   for (int kb = 0; kb < 8; ++kb) {
      const float *p = afrag2 + (((size_t)v * 8 + kb) * 64 + lane) * 8;
      const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
      const float t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      g2_split8_rn(t, ah[kb], al[kb]);
   }
   float wny[8];                                                               // bin 128's weights of this lane's eight slots (fold of k block v)
#pragma unroll
   for (int e = 0; e < 8; ++e) wny[e] = nyq2[16 * v + 8 * h + e];
   for (int p = tid; p < kTiles * 32; p += 512) {
      const int pc = min(p, kPos - 1), c = pc / F, fr = pc - c * F;
      ptab[p] = make_int2(c * kCP + fr * kG2BlockPitch, c | (fr << 8) | (p < kPos ? 1 << 16 : 0));
   }

   const int n_groups = (n_chunks + G - 1) / G;
   const int nlg = ((int)blockIdx.x < n_groups) ? (n_groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;     // this workgroup's groups: blockIdx.x + gl * gridDim.x
   const int n_it = nlg * kTiles;
   if (n_it == 0) return;

   // fold geometry of this lane: slots 8 q .. 8 q + 7, q = 2 v + h; direct taps x[8 q + 1 + e] = group q elements 1..7 + group q + 1 element 0; mirrored taps x[255 - 8 q - e] = group 31 - q
   const int q = 2 * v + h;
   auto goff = [](int g) { return (g >> 3) * kG2BlockPitch + (g & 7) * 8; };   // halves
   const int offD = goff(q), offN = goff(q + 1), offM = goff(31 - q);

   // ------------------------------------------------------------------------------------------------ staging
   // part p < kTiles - 1 of local group tg: 16-byte pieces [p kPerPart, (p + 1) kPerPart) of its samples, global -> register -> LDS
   auto stage_load = [&](int tg, int part, g2_u4v &r) -> bool {
      const int u = part * kPerPart + tid;
      bool on = tg < nlg && tid < kPerPart && u < kMainU;
      if (on) {
         const int c = u / (S / 8), q8 = u - c * (S / 8);
         const int item = min(((int)blockIdx.x + tg * (int)gridDim.x) * G + c, n_chunks - 1);
         on = 8 * q8 < nrt;
         if (on) r = *reinterpret_cast<const g2_u4v *>(pcm + (size_t)map(item) * nrt + 8 * q8);
      }
      return on;
   };
   auto stage_store = [&](int tg, int part, const g2_u4v &r) {
      const int u = part * kPerPart + tid;
      const int c = u / (S / 8), q8 = u - c * (S / 8);
      const int P = kPadG + 8 * q8;
      *reinterpret_cast<g2_u4v *>(&X0[tg & 1][c * kCP + (P >> 6) * kG2BlockPitch + (P & 63)]) = r;
   };
   // last part: the reflect pads (no edge repeat), LDS -> LDS, and the group's output rows
   auto stage_pads = [&](int tg) {
      if (tg >= nlg) return;
      int16_t *x = X0[tg & 1];
      for (int i = tid; i < kPads; i += 512) {
         const int c = i / (2 * kPadG), jj = i - c * (2 * kPadG);
         const int dst = jj < kPadG ? jj : nrt + jj;
         const int src = jj < kPadG ? 2 * kPadG - jj : nrt + 2 * kPadG - 2 - jj;
         x[c * kCP + (dst >> 6) * kG2BlockPitch + (dst & 63)] = x[c * kCP + (src >> 6) * kG2BlockPitch + (src & 63)];
      }
      if (tid < G) {
         const int it = ((int)blockIdx.x + tg * (int)gridDim.x) * G + tid;
         crow[tg & 3][tid] = it < n_chunks ? (int)map(it) : -1;
      }
   };

   // ------------------------------------------------------------------------------------------------ fold + split of one column tile (k block v, all four planes)
   struct Fold { g2_h8v sh, sl, dh, dl; float ny; };
   auto fold_read = [&](int tile, g2_u4v &D, g2_u2v &N, g2_u4v &M) {
      const int gl = tile / kTiles, ti = tile - gl * kTiles;
      const int16_t *x = X0[gl & 1] + ptab[ti * 32 + j].x;
      D = *reinterpret_cast<const g2_u4v *>(x + offD);
      N = *reinterpret_cast<const g2_u2v *>(x + offN);
      M = *reinterpret_cast<const g2_u4v *>(x + offM);
   };
   auto fold_math = [&](const g2_u4v &D, const g2_u2v &N, const g2_u4v &M) -> Fold {
      // direct element e = x[8 q + 1 + e]: halves 1..7 of D, then half 0 of N;  mirrored element e = x[255 - 8 q - e]: half 7 - e of M
      int dv[8], mv[8];
      dv[0] = g2_sext_hi(D[0]); dv[1] = g2_sext_lo(D[1]); dv[2] = g2_sext_hi(D[1]); dv[3] = g2_sext_lo(D[2]);
      dv[4] = g2_sext_hi(D[2]); dv[5] = g2_sext_lo(D[3]); dv[6] = g2_sext_hi(D[3]); dv[7] = g2_sext_lo(N[0]);
      mv[0] = g2_sext_hi(M[3]); mv[1] = g2_sext_lo(M[3]); mv[2] = g2_sext_hi(M[2]); mv[3] = g2_sext_lo(M[2]);
      mv[4] = g2_sext_hi(M[1]); mv[5] = g2_sext_lo(M[1]); mv[6] = g2_sext_hi(M[0]); mv[7] = g2_sext_lo(M[0]);
      float sv[8], dd[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sv[e] = (float)(dv[e] + mv[e]); dd[e] = (float)(dv[e] - mv[e]); }
      Fold f;
      g2_h2v hi, lo;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
         g2_split2(sv[e], sv[e + 1], hi, lo);
         f.sh[e] = hi[0]; f.sh[e + 1] = hi[1]; f.sl[e] = lo[0]; f.sl[e + 1] = lo[1];
         g2_split2(dd[e], dd[e + 1], hi, lo);
         f.dh[e] = hi[0]; f.dh[e + 1] = hi[1]; f.dl[e] = lo[0]; f.dl[e + 1] = lo[1];
      }
      float ny = wny[0] * sv[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) ny = fmaf(wny[e], sv[e], ny);
      f.ny = g2_sum_halves(ny);                                                 // this k block's share of bin 128 (both lane halves hold it)
      return f;
   };
   // ------------------------------------------------------------------------------------------------ prologue: group 0 staged, part 0 of group 1, tile 0 folded
   {
      g2_u4v r;
#pragma unroll 1
      for (int p = 0; p < kMainParts; ++p)
         if (stage_load(0, p, r)) stage_store(0, p, r);
      __syncthreads();
      stage_pads(0);
      if (stage_load(1, 0, r)) stage_store(1, 0, r);
      __syncthreads();
      g2_u4v D, M; g2_u2v N;
      fold_read(0, D, N, M);
      const Fold f = fold_math(D, N, M);
      *reinterpret_cast<g2_h8v *>(&Bf[0][0][v][lane][0]) = f.sh;
      *reinterpret_cast<g2_h8v *>(&Bf[0][1][v][lane][0]) = f.sl;
      *reinterpret_cast<g2_h8v *>(&Bf[0][2][v][lane][0]) = f.dh;
      *reinterpret_cast<g2_h8v *>(&Bf[0][3][v][lane][0]) = f.dl;
      if (h == 0) Ny[0][v][j] = f.ny;
      __syncthreads();
   }

   constexpr unsigned kRow = 129u * F * 4u;                                     // bytes per output row (chunk)
   const int planeB = 2 * role;                                                 // re rows multiply the sums, im rows the differences
   const int binbase = 32 * w + 16 * role + 4 * h;                              // bin of register i (this wave finalises accumulator registers 0..7): binbase + 8 (i >> 2) + (i & 3)
   float carry = 0.0f, nyv = 0.0f;                                              // bin sum of this wave's 16 bins / bin 128's value, of the tile finalised in the previous iteration
   unsigned fm_off = kG2Oob;                                                    // ... and where its FM partial goes (offset from fm_row0's row; masked lanes carry the out-of-range bit)
   int fm_row0 = 0;

   // One iteration `it` of a wave, between two barriers (accumulators alternate between two register sets, so the loop below is unrolled by two):
   //   head     the accumulators of tile it - 1 (complete: their last MFMA was issued in front of the barrier): registers 8..15 go to the partner wave, registers 0..7 are squared
   //   chain    24 MFMAs of tile it on the other accumulator set, and in their shadow
   //              the fold + split of tile it + 1 (k block v) and its LDS writes,
   //              the finalisation of tile it - 2: partner's halves from LDS, magnitude, log1p, 8 stores, bin sums
   //   tail     bin 128 of tile it - 2 (wave 3), FM partial of tile it - 3 (re waves), staged samples / pads of a later group
   auto iteration = [&](const int it, g2_f16v &acc, const g2_f16v &accp, float (&keep_use)[8], float (&keep_make)[8]) __attribute__((always_inline)) {
      const int gl = it / kTiles, ti = it - gl * kTiles;
      // ---- staging: issue this iteration's global load
      const int stg = (ti == kTiles - 1) ? gl + 2 : gl + 1, spart = (ti == kTiles - 1) ? 0 : ti + 1;
      g2_u4v sreg;
      bool son = false;
      if (!(ABL & 8) && spart < kMainParts) son = stage_load(stg, spart, sreg);

      // ---- LDS reads that do not depend on this iteration's work
      const _Float16 *bb = &Bf[it & 1][planeB][0][lane][0];
      g2_h8v bh[8], bl[8];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) { bh[kb] = *reinterpret_cast<const g2_h8v *>(bb + kb * 512); bl[kb] = *reinterpret_cast<const g2_h8v *>(bb + 4096 + kb * 512); }
      const int tn = min(it + 1, n_it - 1);
      g2_u4v D, M; g2_u2v N;
      fold_read(tn, D, N, M);
      const int tp = it - 2;                                                    // the tile finalised in this iteration
      const int glp = max(tp, 0) / kTiles, tip = max(tp, 0) - glp * kTiles;
      const int2 ptp = ptab[tip * 32 + j];
      const int rowp = crow[glp & 3][ptp.y & 255], rowp0 = crow[glp & 3][0];
      const float *ep = &Ex[tp & 1][v ^ 4][0][lane][0];
      const g2_f4v p0 = *reinterpret_cast<const g2_f4v *>(ep), p1 = *reinterpret_cast<const g2_f4v *>(ep + 256);
      const float sim = Sx[(it - 3) & 1][w][j];

      // ---- head: hand half of tile it - 1's accumulators to the partner wave, keep the squares of the other half
      if (!(ABL & 32)) {
         float *eo = &Ex[(it - 1) & 1][v][0][lane][0];
         *reinterpret_cast<g2_f4v *>(eo) = g2_f4v{accp[8], accp[9], accp[10], accp[11]};
         *reinterpret_cast<g2_f4v *>(eo + 256) = g2_f4v{accp[12], accp[13], accp[14], accp[15]};
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) keep_make[i] = accp[i] * accp[i];

      // ---- the matrix products of tile `it` (stale operands in the drain iterations: harmless, nothing of them is stored)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
         if (kb + 2 < 8) { bh[kb + 2] = *reinterpret_cast<const g2_h8v *>(bb + (kb + 2) * 512); bl[kb + 2] = *reinterpret_cast<const g2_h8v *>(bb + 4096 + (kb + 2) * 512); }
         if (ABL & 4) { acc[kb] += (float)bh[kb][0] * (float)al[kb][0] + (float)bl[kb][1] * (float)ah[kb][1]; continue; }
         acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb], bh[kb], acc, 0, 0, 0);
         acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], bl[kb], acc, 0, 0, 0);
         acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], bh[kb], acc, 0, 0, 0);
      }

      // ---- fold + split of tile it + 1 (k block v), written as soon as it is there
      Fold fo;
      if (ABL & 2) {
         const g2_h8v z = __builtin_bit_cast(g2_h8v, D);
         fo.sh = z; fo.sl = __builtin_bit_cast(g2_h8v, M); fo.dh = z; fo.dl = z; fo.ny = __builtin_bit_cast(float, N[0]);
      } else fo = fold_math(D, N, M);
      {
         const int b = (it + 1) & 1;
         *reinterpret_cast<g2_h8v *>(&Bf[b][0][v][lane][0]) = fo.sh;
         *reinterpret_cast<g2_h8v *>(&Bf[b][1][v][lane][0]) = fo.sl;
         *reinterpret_cast<g2_h8v *>(&Bf[b][2][v][lane][0]) = fo.dh;
         *reinterpret_cast<g2_h8v *>(&Bf[b][3][v][lane][0]) = fo.dl;
      }

      // ---- finalisation of tile it - 2: this wave's 16 bins x 32 positions
      float part = 0.0f;
      const bool okp = tp >= 0 && tp < n_it && (ptp.y >> 16) != 0 && rowp >= 0;
      const unsigned maskp = okp ? 0u : kG2Oob;
      const unsigned relp = (unsigned)(rowp - rowp0), frp = (unsigned)((ptp.y >> 8) & 255);
      const int row0p = __builtin_amdgcn_readfirstlane(max(rowp0, 0));
      {
         const float pn[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
         const unsigned off = (relp * kRow + (unsigned)(binbase * F) * 4u + frp * 4u) | maskp;
         const __amdgpu_buffer_rsrc_t ry = g2_rsrc(reinterpret_cast<const char *>(Y) + (size_t)row0p * kRow);
         const __amdgpu_buffer_rsrc_t rm = g2_rsrc(reinterpret_cast<const char *>(WMAG ? MAG : Y) + (size_t)row0p * kRow);
#pragma unroll
         for (int i = 0; i < 8; ++i) {
            const float m = (ABL & 16) ? fmaf(pn[i], pn[i], keep_use[i]) : __builtin_amdgcn_sqrtf(fmaf(pn[i], pn[i], keep_use[i]));                    // 2^23 x magnitude
            const float val = (ABL & 16) ? fmaf(m, 0.125f, 1.0f) : __builtin_amdgcn_logf(fmaf(m, 0.125f, 1.0f)) * 0.6931471805599453f;   // log1p(2^20 magnitude); v_log_f32 is good to ~1 ulp of log2
            const unsigned o = off + (unsigned)((8 * (i >> 2) + (i & 3)) * F) * 4u;
            if (!(ABL & 1)) g2_store(val, ry, o);
            if (WMAG) g2_store(m * 1.1920928955078125e-07f, rm, o);                                 // 2^-23
            part += val;
         }
         part = g2_sum_halves(part);
      }
#if VADC_G2_VALU_PER_MFMA > 0
      // experiment: one matrix instruction, then N vector instructions, 24 times.  hipcc's own schedule already spaces the dependent MFMAs ~8 instructions apart
      // (the fold and the finalisation in between); with this directive it clumped them instead, so it is off by default
#pragma unroll
      for (int m = 0; m < 24; ++m) {
         __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
         __builtin_amdgcn_sched_group_barrier(0x002, VADC_G2_VALU_PER_MFMA, 0);
      }
#endif
      // ---- tail.  FM partial w of tile it - 3 = (re wave's 16 bins + im wave's 16 bins) [+ bin 128 for w = 3]: the sums and the address were carried from the previous iteration
      if (role == 0) {                                                          // wave-uniform
         const float fmv = (carry + sim) + nyv;                                 // nyv = 0 except in wave 3
         g2_store(fmv, g2_rsrc(reinterpret_cast<const char *>(FM) + ((size_t)w * fm_stride + (size_t)fm_row0 * F) * 4u), fm_off);
      }
      // bin 128 of tile it - 2 (wave 3: it carries partial 3 of FM)
      float nyval = 0.0f;
      if (v == 3) {                                                             // wave-uniform
         float nsh[8];
#pragma unroll
         for (int k8 = 0; k8 < 8; ++k8) nsh[k8] = Ny[tp & 3][k8][j];
         const float ny = ((nsh[0] + nsh[1]) + (nsh[2] + nsh[3])) + ((nsh[4] + nsh[5]) + (nsh[6] + nsh[7]));     // 2^15 x re of bin 128 (its im row is identically zero)
         const float nm = fabsf(ny);
         nyval = __builtin_amdgcn_logf(fmaf(nm, 32.0f, 1.0f)) * 0.6931471805599453f;
         const unsigned off = (relp * kRow + (unsigned)(128 * F) * 4u + frp * 4u) | maskp | (h == 0 ? 0u : kG2Oob);
         g2_store(nyval, g2_rsrc(reinterpret_cast<const char *>(Y) + (size_t)row0p * kRow), off);
         if (WMAG) g2_store(nm * 3.0517578125e-05f, g2_rsrc(reinterpret_cast<const char *>(MAG) + (size_t)row0p * kRow), off);   // 2^-15
      }
      carry = part; nyv = nyval;
      fm_off = ((relp * F + frp) * 4u) | maskp | (h == 0 ? 0u : kG2Oob);
      fm_row0 = row0p;
      if (h == 0) Ny[(it + 1) & 3][v][j] = fo.ny;
      if (role == 1 && h == 0) Sx[tp & 1][w][j] = part;
      if (son) stage_store(stg, spart, sreg);
      if (!(ABL & 8) && spart == kMainParts) stage_pads(stg);
      __syncthreads();
   };

   g2_f16v acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
   float keep0[8], keep1[8];
#pragma unroll
   for (int i = 0; i < 8; ++i) { keep0[i] = 0.0f; keep1[i] = 0.0f; }
   const int n_loop = (n_it + 3 + 1) & ~1;                                      // tiles + three drain iterations, rounded up to the unroll factor (a surplus iteration stores nothing)
#pragma unroll 1
   for (int it = 0; it < n_loop; it += 2) {
      iteration(it, acc0, acc1, keep0, keep1);
      iteration(it + 1, acc1, acc0, keep1, keep0);
   }
}

template <int ABL>
void launch_frontend_gemm2_abl(const int16_t *pcm, const float *afrag2, const float *nyq2, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int n_cus, hipStream_t st)
{
   const int groups = (n + G2Geo<1>::chunks - 1) / G2Geo<1>::chunks;
   hipLaunchKernelGGL((k_frontend_gemm2<1, false, ABL>), dim3(groups < n_cus ? groups : n_cus), dim3(512), 0, st, pcm, afrag2, nyq2, Y, nullptr, FM, n, map, fm_stride, G2Geo<1>::samples);
}

// (no magnitude array: the v4 first stage recovers the magnitudes from Y; the WMAG = true instantiations served option "v4_mag" = 1 until round 5)
void launch_frontend_gemm2_s16(const int16_t *pcm, const float *afrag2, const float *nyq2, float *Y, float *FM, size_t fm_stride,
                               int n, ItemMap map, int n_cus, hipStream_t st, int geo, int nrt)
{
   if (n <= 0) return;
#define VADC_G2_CASE(GEO) \
   case GEO: { \
      const int groups = (n + G2Geo<GEO>::chunks - 1) / G2Geo<GEO>::chunks; \
      const int grid = groups < n_cus ? groups : n_cus; \
      hipLaunchKernelGGL((k_frontend_gemm2<GEO, false>), dim3(grid), dim3(512), 0, st, pcm, afrag2, nyq2, Y, nullptr, FM, n, map, fm_stride, \
                         (nrt > 0 && nrt < G2Geo<GEO>::samples) ? nrt : G2Geo<GEO>::samples); \
   } break;
   switch (geo) {
   VADC_G2_CASE(1) VADC_G2_CASE(2) VADC_G2_CASE(3) VADC_G2_CASE(4) VADC_G2_CASE(5) VADC_G2_CASE(6)
   default: VADC_G2_CASE(0)
   }
#undef VADC_G2_CASE
}

}  // namespace vadc
