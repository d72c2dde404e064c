// NOT PART OF THE BUILD -- an experiment kept in history for DESIGN.md section 4.5 (no faster than k_frontend_gemm2; this last revision, with the vector waves' LDS reads
// hoisted, also has FM wrong in one frame of six: not debugged, the revision before it was bit-identical in Y and FM on every geometry).
// kernels_frontend_gemm4.hip -- the GEMM STFT front end, third form: k_frontend_gemm2's arithmetic (bit for bit) with the work of a column tile divided between
// two KINDS of waves of one persistent 16-wave workgroup per CU.
//
// Same contract as kernels_frontend_gemm2.hip (reference: silero_vad.py:22-66 STFT_conv + AdaptiveAudioNormalization for Silero v4, reached by the reference through
// onnxruntime, onnx_helpers.c:83-115; for Silero v3.1 only in the FAST_STFT throughput mode, replacing tensor.h:912-958, stft.c:15-224, misc.c:40-63): reflect pad,
// conv1d with the [258,1,256] basis at hop 64 as a folded real-input GEMM (slot j = tap j + 1), sqrt(re^2 + im^2), log1p(2^20 m), four partial per-frame bin sums.
// s16 input only.
//
// Why a third form.  In k_frontend_gemm2 every wave does everything -- 24 MFMAs, its share of the fold, of the finalisation, of the sums -- and the eight waves
// meet at one barrier per tile: its matrix work alone runs in 0.185 ms per 65,536 chunks, the whole kernel in 0.351, because the ~150 vector instructions a wave
// issues per tile do not hide behind its own MFMA chain (two waves per SIMD, both in the same phase).  Here
//   * waves 0-7 (two per SIMD) are MATRIX waves: the 32 rows x K = 128 of the split basis resident in registers, per tile 16 B-fragment reads, 24 MFMAs, and the
//     16 accumulator registers of the previous tile written to LDS -- nothing else;
//   * waves 8-15 (two per SIMD) are VECTOR waves: wave 8 + u folds k block u of the NEXT tile for everybody and finalises 16 bins (block u & 3, half u >> 2) of the
//     PREVIOUS tile for all 32 positions (re and im accumulators from LDS: magnitude, log1p, 8 stores per lane, the bin sums) -- exactly the share a wave of
//     k_frontend_gemm2 had, without its MFMAs; wave 11 also bin 128;
//   so the matrix pipe's two waves per SIMD never wait for vector work, and two more waves per SIMD fill the issue slots their MFMAs leave.  (One vector wave per
//   SIMD with twice the share -- the first version -- was no faster than k_frontend_gemm2: a single wave cannot hide its own latencies over ~450 instructions per tile.)
// Two barriers per tile, both LDS-only (s_waitcnt lgkmcnt(0) + s_barrier: nothing travels between waves through global memory, and a fence's vmcnt(0) would make
// the vector waves wait for their Y stores): A at the tile's end, B behind the matrix waves' accumulator write, which the vector waves pass at once.  The
// accumulator image in LDS is single-buffered: written behind A, read behind B.
// Bit-identical to k_frontend_gemm2 in Y, MAG and FM: the same MFMA sequence per accumulator, the same operation order in the magnitudes (bins 0..15 of a block:
// fma(im, im, re * re); bins 16..31: fma(re, re, im * im) -- which wave "owned" which half there) and in the sums.
#include "common.h"
#include "frontend_gemm2_common.h"

namespace vadc {

#ifdef VADC_G4_CLOCK_PROBE      // tools/gemm_bench: the shader clock the chip holds under this kernel (s_memtime against the constant 100-MHz counter, workgroup 0)
__device__ long long g_g4_clock[2];
#endif

// afrag2 / nyq2: as packed for k_frontend_gemm2 (gemm2_pack.h); its im tiles are stored rotated by 16 rows, which this kernel undoes when it loads them
// ABL (tools/gemm_bench, timing only, results WRONG): 1 = the matrix waves skip their MFMA chain, 2 = the vector waves skip finalisation and fold, 4 = no Y stores,
// 8 = no fold, 16 = one output per lane instead of eight
template <int GEO, bool WMAG, int ABL = 0>
__global__ __launch_bounds__(1024) void k_frontend_gemm4(const int16_t *__restrict__ pcm, const float *__restrict__ afrag2, const float *__restrict__ nyq2,
                                                        float *__restrict__ Y, float *__restrict__ MAG, float *__restrict__ FM,
                                                        int n_chunks, ItemMap map, size_t fm_stride)
{
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
   static_assert(kPerPart <= 512, "one 16-byte piece per vector-wave thread and staging part");
   constexpr int kPads = G * 2 * kPadG;

   __shared__ __attribute__((aligned(16))) int16_t X0[2][kX0];
   __shared__ __attribute__((aligned(16))) _Float16 Bf[2][4][8][64][8];        // [buffer][s hi, s lo, d hi, d lo][kb][lane][8]: 2 x 32 KB
   __shared__ __attribute__((aligned(16))) float Acc[8][4][64][4];             // [matrix wave][register quad][lane][4]: the accumulators of the tile before: 32 KB
   constexpr bool kWide = !WMAG && F % 4 == 0 && GEO != 5;                      // 16-byte Y stores (GEO 5's staging buffers leave no room for the transpose tile)
   __shared__ __attribute__((aligned(16))) float Yt[kWide ? 8 : 1][16][32];    // a vector wave's 16 bins x 32 positions of log-magnitudes on their way to 16-byte stores: 16 KB
   __shared__ float Sx[2][4][32];                                              // the upper halves' bin sums of a tile, by position
   __shared__ float Ny[4][8][32];                                              // bin 128: the eight k blocks' shares of a tile, by position
   __shared__ int crow[4][G];                                                  // group (mod 4) -> output rows of its chunks, -1 past the end
   __shared__ int crin[4][G];                                                  // group (mod 4) -> input rows of its chunks (clamped to the last chunk past the end)

   const int tid = threadIdx.x, lane = tid & 63, v = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform
   const bool matrix = v < 8;
   const int u = v - 8;                                                        // vector wave 0..7 (matrix waves: negative, unused)
   const int vt = tid - 512;                                                   // thread index among the vector waves
   const int role = (v >> 2) & 1, w = v & 3, j = lane & 31, h = lane >> 5;      // role: matrix waves re / im rows; vector waves lower / upper 16 bins of block w

   const int n_groups = (n_chunks + G - 1) / G;
   const int nlg = ((int)blockIdx.x < n_groups) ? (n_groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;     // this workgroup's groups: blockIdx.x + gl * gridDim.x
   const int n_it = nlg * kTiles;
   if (n_it == 0) return;
#ifdef VADC_G4_CLOCK_PROBE
   const long long probe_w0 = wall_clock64(), probe_c0 = clock64();
#endif
   // tiles + two drain iterations (finalisation, FM), in whole groups of kTiles: the loops below are unrolled over a group's tiles so that everything that depends on
   // the tile's place in its group -- which chunk and frame a lane's position is, which staging piece is due -- is a compile-time constant or a register made here
   const int n_gloop = (n_it + 2 + kTiles - 1) / kTiles;

#define G4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
   constexpr unsigned kRow = 129u * F * 4u;                                     // bytes per output row (chunk)

   if (matrix) {
      // =============================================================================================== matrix waves
      // this wave's A fragments, split once: 8 k blocks x (hi, lo) = 64 registers (im tiles: the packed rows are rotated by 16, taken back here)
      g2_h8v ah[8], al[8];
      {
         const int src_lane = role ? ((lane & 32) | ((j + 16) & 31)) : lane;
#pragma unroll
         for (int kb = 0; kb < 8; ++kb) {
            const float *p = afrag2 + (((size_t)v * 8 + kb) * 64 + src_lane) * 8;
            const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
            const float t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            g2_split8_rn(t, ah[kb], al[kb]);
         }
      }
      const int planeB = 2 * role;                                              // re rows multiply the sums, im rows the differences
      __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();       // the vector waves' prologue (headers, group 0, its pads, tile 0's fold)
      g2_f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
      for (int it = 0; it < n_gloop * kTiles; ++it) {
         {  // the accumulators of tile it - 1 (it = 0: zeros; nobody stores anything of them).  The barrier's lgkmcnt(0) has the writes done before the registers are reused
            float *ao = &Acc[v][0][lane][0];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) *reinterpret_cast<g2_f4v *>(ao + r4 * 256) = g2_f4v{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
         }
         G4_BARRIER();                                                         // B
         if (it < n_it && !(ABL & 1)) {
            const _Float16 *bb = &Bf[it & 1][planeB][0][lane][0];
            g2_h8v bh[8], bl[8];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) { bh[kb] = *reinterpret_cast<const g2_h8v *>(bb + kb * 512); bl[kb] = *reinterpret_cast<const g2_h8v *>(bb + 4096 + kb * 512); }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
               if (kb + 2 < 8) { bh[kb + 2] = *reinterpret_cast<const g2_h8v *>(bb + (kb + 2) * 512); bl[kb + 2] = *reinterpret_cast<const g2_h8v *>(bb + 4096 + (kb + 2) * 512); }
               acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kb], bh[kb], acc, 0, 0, 0);
               acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], bl[kb], acc, 0, 0, 0);
               acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kb], bh[kb], acc, 0, 0, 0);
            }
         }
         G4_BARRIER();                                                         // A
      }
   } else {
      // =============================================================================================== vector waves
      // ---- per-lane tables, made once.  Position p = 32 ti + j of a group: chunk c = p / F, frame fr = p - c F.
      int pos_x[kTiles];                                                       // offset (halves) of the position's window in a staging buffer
      unsigned pos_off[kTiles];                                                // byte offset of this wave's output 0 inside the group's rows, without the row term: (binbase F + fr) 4; kG2Oob past the group's positions
      int pos_c[kTiles];
      const int binbase = 32 * w + 16 * role + 4 * h;                          // bin of this wave's output i = 0..7: binbase + 8 (i >> 2) + (i & 3)  (accumulator register 8 role + i)
#pragma unroll
      for (int ti = 0; ti < kTiles; ++ti) {
         const int p = 32 * ti + j, pc = min(p, kPos - 1), c = pc / F, fr = pc - c * F;
         pos_x[ti] = c * kCP + fr * kG2BlockPitch;
         pos_c[ti] = c;
         pos_off[ti] = (unsigned)((binbase * F + fr) * 4) | (p < kPos ? 0u : kG2Oob);
      }
      // 16-byte stores (frames per chunk a multiple of 4, no magnitude array): lane L stores bins L / 8 and 8 + L / 8 of the wave's 16 at positions 4 (L % 8) .. + 3 of the
      // tile -- four consecutive frames of ONE chunk, 16-byte aligned.  A dword store costs the address unit a lane at a time whatever it carries: 72 store
      // instructions per tile and CU were a quarter of the kernel's time (tools/gemm_bench, "no Y stores")
      unsigned w4_off[kTiles];                                                 // ((32 w + 16 role + L / 8) F + fr) 4 of the lane's first store; kG2Oob past the group's positions
      int w4_c[kTiles];
#pragma unroll
      for (int ti = 0; ti < kTiles; ++ti) {
         const int p = 32 * ti + 4 * (lane & 7), pc = min(p, kPos - 1), c = pc / F, fr = pc - c * F;
         w4_c[ti] = c;
         w4_off[ti] = (unsigned)(((32 * w + 16 * role + (lane >> 3)) * F + fr) * 4) | (p < kPos ? 0u : kG2Oob);
      }
      // staging: one 16-byte piece per vector thread and part
      int st_lds[kMainParts], st_c[kMainParts];                                // where piece (part, vt) goes in a staging buffer (halves; -1: none), and its chunk
      unsigned st_src[kMainParts];                                             // its byte offset inside the chunk's samples
#pragma unroll
      for (int part = 0; part < kMainParts; ++part) {
         const int uu = part * kPerPart + vt;
         const bool on = vt < kPerPart && uu < kMainU;
         const int c = on ? uu / (S / 8) : 0, q8 = on ? uu - c * (S / 8) : 0, P = kPadG + 8 * q8;
         st_lds[part] = on ? c * kCP + (P >> 6) * kG2BlockPitch + (P & 63) : -1;
         st_c[part] = c; st_src[part] = (unsigned)(16 * q8);
      }
      float wny[8];                                                            // bin 128's weights of this lane's eight slots in the k block it folds
#pragma unroll
      for (int e = 0; e < 8; ++e) wny[e] = nyq2[16 * u + 8 * h + e];
      auto goff = [](int g) { return (g >> 3) * kG2BlockPitch + (g & 7) * 8; };   // halves
      const int offD = goff(2 * u + h), offN = goff(2 * u + h + 1), offM = goff(31 - (2 * u + h));

      // the rows of local group tg's chunks: input rows clamped to the last chunk, output rows -1 past the end (one wave-0 lane per chunk)
      auto group_header = [&](int tg) {
         if (vt < G && tg < nlg) {
            const int it0 = ((int)blockIdx.x + tg * (int)gridDim.x) * G + vt;
            crin[tg & 3][vt] = (int)map(min(it0, n_chunks - 1));
            crow[tg & 3][vt] = it0 < n_chunks ? (int)map(it0) : -1;
         }
      };
      auto stage_load = [&](int tg, int part, g2_u4v &r) -> bool {
         const bool on = tg < nlg && st_lds[part] >= 0;
         if (on) {
            const int row0 = __builtin_amdgcn_readfirstlane(crin[tg & 3][0]);
            const unsigned rel = (unsigned)(crin[tg & 3][st_c[part]] - row0);
            r = __builtin_bit_cast(g2_u4v, __builtin_amdgcn_raw_buffer_load_b128(g2_rsrc(reinterpret_cast<const char *>(pcm) + (size_t)row0 * (S * 2)), rel * (unsigned)(S * 2) + st_src[part], 0, 0));
         }
         return on;
      };
      auto stage_store = [&](int tg, int part, const g2_u4v &r) { *reinterpret_cast<g2_u4v *>(&X0[tg & 1][st_lds[part]]) = r; };
      // the reflect pads (no edge repeat), LDS -> LDS
      auto stage_pads = [&](int tg) {
         if (tg >= nlg) return;
         int16_t *x = X0[tg & 1];
         for (int i = vt; i < kPads; i += 512) {
            const int c = i / (2 * kPadG), jj = i - c * (2 * kPadG);
            const int dst = jj < kPadG ? jj : S + jj;
            const int src = jj < kPadG ? 2 * kPadG - jj : S + 2 * kPadG - 2 - jj;
            x[c * kCP + (dst >> 6) * kG2BlockPitch + (dst & 63)] = x[c * kCP + (src >> 6) * kG2BlockPitch + (src & 63)];
         }
      };

      // ---- fold + split of one column tile (tile ti of local group gl), k block u, all four planes.
      // slots 8 q .. 8 q + 7, q = 2 u + h; direct taps x[8 q + 1 + e] = group q elements 1..7 + group q + 1 element 0; mirrored taps x[255 - 8 q - e] = group 31 - q
      auto fold_rest = [&](const g2_u4v &D, const g2_u2v &N, const g2_u4v &M, int tile) {
         int dv[8], mv[8];
         dv[0] = g2_sext_hi(D[0]); dv[1] = g2_sext_lo(D[1]); dv[2] = g2_sext_hi(D[1]); dv[3] = g2_sext_lo(D[2]);
         dv[4] = g2_sext_hi(D[2]); dv[5] = g2_sext_lo(D[3]); dv[6] = g2_sext_hi(D[3]); dv[7] = g2_sext_lo(N[0]);
         mv[0] = g2_sext_hi(M[3]); mv[1] = g2_sext_lo(M[3]); mv[2] = g2_sext_hi(M[2]); mv[3] = g2_sext_lo(M[2]);
         mv[4] = g2_sext_hi(M[1]); mv[5] = g2_sext_lo(M[1]); mv[6] = g2_sext_hi(M[0]); mv[7] = g2_sext_lo(M[0]);
         float sv[8], dd[8];
#pragma unroll
         for (int e = 0; e < 8; ++e) { sv[e] = (float)(dv[e] + mv[e]); dd[e] = (float)(dv[e] - mv[e]); }
         g2_h8v sh, sl, dh, dl;
         g2_h2v hi, lo;
#pragma unroll
         for (int e = 0; e < 8; e += 2) {
            g2_split2(sv[e], sv[e + 1], hi, lo);
            sh[e] = hi[0]; sh[e + 1] = hi[1]; sl[e] = lo[0]; sl[e + 1] = lo[1];
            g2_split2(dd[e], dd[e + 1], hi, lo);
            dh[e] = hi[0]; dh[e + 1] = hi[1]; dl[e] = lo[0]; dl[e + 1] = lo[1];
         }
         float ny = wny[0] * sv[0];
#pragma unroll
         for (int e = 1; e < 8; ++e) ny = fmaf(wny[e], sv[e], ny);
         ny = g2_sum_halves(ny);                                                // this k block's share of bin 128 (both lane halves hold it)
         const int bb = tile & 1;
         *reinterpret_cast<g2_h8v *>(&Bf[bb][0][u][lane][0]) = sh;
         *reinterpret_cast<g2_h8v *>(&Bf[bb][1][u][lane][0]) = sl;
         *reinterpret_cast<g2_h8v *>(&Bf[bb][2][u][lane][0]) = dh;
         *reinterpret_cast<g2_h8v *>(&Bf[bb][3][u][lane][0]) = dl;
         if (h == 0) Ny[tile & 3][u][j] = ny;
      };
      auto fold_tile = [&](int gl, int ti_x, int tile) {
         const int16_t *x = X0[gl & 1] + ti_x;
         const g2_u4v D = *reinterpret_cast<const g2_u4v *>(x + offD);
         const g2_u2v N = *reinterpret_cast<const g2_u2v *>(x + offN);
         const g2_u4v M = *reinterpret_cast<const g2_u4v *>(x + offM);
         fold_rest(D, N, M, tile);
      };

      // ---- prologue: the first two groups' rows, group 0 staged, part 0 of group 1, tile 0 folded
      {
         group_header(0); group_header(1);
         __syncthreads();
         g2_u4v r;
#pragma unroll
         for (int part = 0; part < kMainParts; ++part)
            if (stage_load(0, part, r)) stage_store(0, part, r);
         __syncthreads();
         stage_pads(0);
         if (stage_load(1, 0, r)) stage_store(1, 0, r);
         __syncthreads();
         fold_tile(0, pos_x[0], 0);
         __syncthreads();
      }

      float carry = 0.0f, nyv = 0.0f;                                          // bin sum of this wave's 16 bins / bin 128's value, of the tile finalised in the previous iteration
      unsigned fm_off = kG2Oob;                                                // ... and where its FM partial goes
      int fm_row0 = 0;
#pragma unroll 1
      for (int gl = 0; gl < n_gloop; ++gl) {
#pragma unroll
         for (int ti = 0; ti < kTiles; ++ti) {
            const int it = gl * kTiles + ti;
            // ---- staging: this iteration's piece of a later group (the load in front of B, the LDS write behind the tile's work)
            constexpr int kLast = kTiles - 1;
            const int stg = (ti == kLast) ? gl + 2 : gl + 1;
            const int spart = (ti == kLast) ? 0 : ti + 1;                        // compile-time per ti
            g2_u4v sreg;
            bool son = false;
            if (spart < kMainParts) son = stage_load(stg, spart < kMainParts ? spart : 0, sreg);
            G4_BARRIER();                                                      // B
            if (ABL & 2) { if (son) stage_store(stg, spart < kMainParts ? spart : 0, sreg); if (spart == kMainParts) { stage_pads(stg); group_header(gl + 2); } G4_BARRIER(); continue; }
            // ---- every LDS read of the iteration first: the accumulators and rows of tile it - 1 (tile tip of group glp), the samples of tile it + 1's fold.
            // One basic block from here to the bin-128 branch: hipcc interleaves the finalisation's and the fold's arithmetic and their LDS round trips
            const int tp = it - 1;
            const int tip = (ti + kTiles - 1) % kTiles, glp = (ti == 0) ? gl - 1 : gl;
            const int gslot = glp & 3;
            const int rowp = crow[gslot][pos_c[tip]], rowp0 = crow[gslot][0];
            const int rw = crow[gslot][w4_c[tip]];
            g2_f4v ro[2], io[2];                                                 // accumulator registers 8 role .. 8 role + 7 of the re wave and of the im wave of block w
            {
               const float *ar = &Acc[w][2 * role][lane][0], *ai = &Acc[w + 4][2 * role][lane][0];
#pragma unroll
               for (int r4 = 0; r4 < 2; ++r4) { ro[r4] = *reinterpret_cast<const g2_f4v *>(ar + r4 * 256); io[r4] = *reinterpret_cast<const g2_f4v *>(ai + r4 * 256); }
            }
            const float sim = Sx[it & 1][w][j];                                 // written in iteration it - 1 for tile it - 2
            // (past the last tile the fold runs on stale samples into a buffer nobody reads: no branch)
            const int16_t *xf = X0[(ti == kLast ? gl + 1 : gl) & 1] + pos_x[(ti + 1) % kTiles];
            const g2_u4v fD = *reinterpret_cast<const g2_u4v *>(xf + offD);
            const g2_u2v fN = *reinterpret_cast<const g2_u2v *>(xf + offN);
            const g2_u4v fM = *reinterpret_cast<const g2_u4v *>(xf + offM);

            // ---- finalisation of tile it - 1: this wave's 16 bins x 32 positions
            const bool okp = tp >= 0 && tp < n_it && rowp >= 0;
            const unsigned maskp = okp ? 0u : kG2Oob;
            const unsigned relp = (unsigned)(rowp - rowp0);
            const int row0p = __builtin_amdgcn_readfirstlane(max(rowp0, 0));
            const __amdgpu_buffer_rsrc_t ry = g2_rsrc(reinterpret_cast<const char *>(Y) + (size_t)row0p * kRow);
            const __amdgpu_buffer_rsrc_t rm = g2_rsrc(reinterpret_cast<const char *>(WMAG ? MAG : Y) + (size_t)row0p * kRow);
            const unsigned off = (relp * kRow + pos_off[tip]) | maskp;
            float part = 0.0f;
#pragma unroll
            for (int i = 0; i < ((ABL & 16) ? 1 : 8); ++i) {
               const float a = ro[i >> 2][i & 3], b = io[i >> 2][i & 3];
               // k_frontend_gemm2's order: the wave that finalised a bin squared ITS OWN accumulator first -- the re wave the lower 16 bins of a block, the im wave the upper
               const float own = role ? b : a, oth = role ? a : b;
               const float keep = own * own;
               const float m = __builtin_amdgcn_sqrtf(fmaf(oth, oth, keep));                         // 2^23 x magnitude
               const float val = __builtin_amdgcn_logf(fmaf(m, 0.125f, 1.0f)) * 0.6931471805599453f;   // log1p(2^20 magnitude)
               const unsigned o = off + (unsigned)((8 * (i >> 2) + (i & 3)) * F) * 4u;
               if (kWide) Yt[kWide ? u : 0][4 * h + 8 * (i >> 2) + (i & 3)][j] = val;
               else if (!(ABL & 4)) g2_store(val, ry, o);
               if (WMAG) g2_store(m * 1.1920928955078125e-07f, rm, o);                                 // 2^-23
               part += val;
            }
            // ---- fold + split of tile it + 1, k block u (its buffer was last read by the matrix waves in iteration it - 1)
            if (!(ABL & 8)) fold_rest(fD, fN, fM, it + 1);
            if (kWide && !(ABL & 4)) {
               // the wave's own 2 KB back as rows of four positions (LDS serves a wave's accesses in order: no wait between its write and its read)
               const unsigned o4 = ((unsigned)(rw - rowp0) * kRow + w4_off[tip]) | ((tp >= 0 && tp < n_it && rw >= 0) ? 0u : kG2Oob);
#pragma unroll
               for (int k2 = 0; k2 < 2; ++k2) {
                  const g2_f4v y4 = *reinterpret_cast<const g2_f4v *>(&Yt[kWide ? u : 0][8 * k2 + (lane >> 3)][4 * (lane & 7)]);
                  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(g2_u4v, y4), ry, o4 + (unsigned)(k2 * 8 * F * 4), 0, 0);
               }
            }
            part = g2_sum_halves(part);
            // ---- FM partial w of tile it - 2 = (lower 16 bins + upper 16 bins) [+ bin 128 for w = 3]: the sums and the address were carried from the previous iteration
            // (the upper halves' waves issue the store masked: no branch)
            {
               const float fmv = (carry + sim) + nyv;                          // nyv = 0 except in vector wave 3
               g2_store(fmv, g2_rsrc(reinterpret_cast<const char *>(FM) + ((size_t)w * fm_stride + (size_t)fm_row0 * F) * 4u), fm_off | (role == 0 ? 0u : kG2Oob));
            }
            // the frame and validity bits of the position inside pos_off: (binbase F + fr) 4 | out-of-range bit
            const unsigned frp4 = (pos_off[tip] & ~kG2Oob) - (unsigned)(binbase * F * 4);
            const unsigned pmask = (pos_off[tip] & kG2Oob) | maskp;
            if (role == 1 && h == 0) Sx[(it + 1) & 1][w][j] = part;              // read in iteration it + 1
            // bin 128 of tile it - 1 (vector wave 3: it carries partial 3 of FM)
            float nyval = 0.0f;
            if (u == 3) {                                                       // wave-uniform
               float nsh[8];
#pragma unroll
               for (int k8 = 0; k8 < 8; ++k8) nsh[k8] = Ny[tp & 3][k8][j];
               const float ny = ((nsh[0] + nsh[1]) + (nsh[2] + nsh[3])) + ((nsh[4] + nsh[5]) + (nsh[6] + nsh[7]));     // 2^15 x re of bin 128 (its im row is identically zero)
               const float nm = fabsf(ny);
               nyval = __builtin_amdgcn_logf(fmaf(nm, 32.0f, 1.0f)) * 0.6931471805599453f;
               const unsigned offn = (relp * kRow + (unsigned)(128 * F) * 4u + frp4) | pmask | (h == 0 ? 0u : kG2Oob);
               g2_store(nyval, ry, offn);
               if (WMAG) g2_store(nm * 3.0517578125e-05f, rm, offn);             // 2^-15
            }
            carry = part; nyv = nyval;
            fm_off = (relp * (unsigned)(F * 4) + frp4) | pmask | (h == 0 ? 0u : kG2Oob);
            fm_row0 = row0p;
            if (son) stage_store(stg, spart < kMainParts ? spart : 0, sreg);
            if (spart == kMainParts) { stage_pads(stg); group_header(gl + 2); }  // (the rows of the group whose part 0 is loaded in the next iteration)
            G4_BARRIER();                                                      // A
         }
      }
   }
#undef G4_BARRIER
#ifdef VADC_G4_CLOCK_PROBE
   if (tid == 0 && blockIdx.x == 0) { g_g4_clock[0] = wall_clock64() - probe_w0; g_g4_clock[1] = clock64() - probe_c0; }
#endif
}

template <int ABL>
void launch_frontend_gemm4_abl(const int16_t *pcm, const float *afrag2, const float *nyq2, float *Y, float *FM, size_t fm_stride, int n, ItemMap map, int n_cus, hipStream_t st)
{
   const int groups = (n + G2Geo<1>::chunks - 1) / G2Geo<1>::chunks;
   hipLaunchKernelGGL((k_frontend_gemm4<1, false, ABL>), dim3(groups < n_cus ? groups : n_cus), dim3(1024), 0, st, pcm, afrag2, nyq2, Y, nullptr, FM, n, map, fm_stride);
}

void launch_frontend_gemm4_s16(const int16_t *pcm, const float *afrag2, const float *nyq2, float *Y, float *MAG, float *FM, size_t fm_stride,
                               int n, ItemMap map, int n_cus, hipStream_t st, int geo)
{
   if (n <= 0) return;
#define VADC_G4_CASE(GEO) \
   case GEO: { \
      const int groups = (n + G2Geo<GEO>::chunks - 1) / G2Geo<GEO>::chunks; \
      const int grid = groups < n_cus ? groups : n_cus; \
      if (MAG) hipLaunchKernelGGL((k_frontend_gemm4<GEO, true>), dim3(grid), dim3(1024), 0, st, pcm, afrag2, nyq2, Y, MAG, FM, n, map, fm_stride); \
      else     hipLaunchKernelGGL((k_frontend_gemm4<GEO, false>), dim3(grid), dim3(1024), 0, st, pcm, afrag2, nyq2, Y, MAG, FM, n, map, fm_stride); \
   } break;
   switch (geo) {
   VADC_G4_CASE(1) VADC_G4_CASE(2) VADC_G4_CASE(3) VADC_G4_CASE(4) VADC_G4_CASE(5)
   default: VADC_G4_CASE(0)
   }
#undef VADC_G4_CASE
}

}  // namespace vadc
