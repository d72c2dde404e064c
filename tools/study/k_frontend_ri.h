// tools/study/k_frontend_ri.h -- NOT part of the product library.  The exact-tree front end with the packed pair = (re, im) of one tree lane, as it was wired into the
// engine at commit b91f1ec (option "fe_opt" = 11) and taken out again: 8 % fewer vector instructions than k_frontend_sym, 1.5 % less time (an unpacked VGPR-only
// v_add_f32 issues in 2 cycles, so packing the derived rows' sums buys instruction count, not time), bit-identical to k_frontend_sym alone on the chip (tools/fe_bench
// sym: 400 launches of 10,240 chunks) -- but beside waves of k_lstm_layer on the same SIMDs one of the rows 64 -+ b changed its bits for lanes 48 .. 63 about once in
// 1.5e5 workgroups, whatever its occupancy (2 .. 5 waves per SIMD), with or without the inline-asm forms, with the epilogue's transcendentals one at a time, with the
// recurrence's streams at normal priority; k_frontend_sym never did (0 of 30,000 runs of the same harness).  Unexplained; DESIGN.md section 4.1, profiles/r04/ri_study/.
// Included by tools/fe_bench.hip (inside namespace vadc, behind kernels_frontend.hip) for the bit and timing comparison.
// stage the (up to) 4 chunks a workgroup's 64 positions touch: reflect pad (tensor.h:931-954); interior octets with one 16-byte load
template <typename T>
__device__ __forceinline__ void sym_stage_chunks(float *xs, const T *__restrict__ pcm, const ItemMap &map, int item0, int n_chunks, int tid)
{
   constexpr int kFlChunks = fl_chunks(1);
   constexpr int kOctets = kPadded / 8;                                      // 224 per chunk
   for (int o = tid; o < kFlChunks * kOctets; o += 256) {
      const int c = o / kOctets, oc = o - c * kOctets;
      const int it = min(item0 + c, n_chunks - 1);
      const T *src = pcm + (size_t)map(it) * kChunk;
      const int idx = oc * 8;                                                // padded index of the octet's first sample
      float v[8];
      if (idx >= kPad && idx < kPad + kChunk) sym_load_octet(src + (idx - kPad), v);
      else {
#pragma unroll
         for (int k = 0; k < 8; ++k) {
            int sidx = idx + k - kPad;
            sidx = sidx < 0 ? -sidx : sidx;
            sidx = sidx >= kChunk ? 2 * (kChunk - 1) - sidx : sidx;
            v[k] = sample_to_f32(src[sidx]);
         }
      }
      sym_store_octet(xs + c * kSymChunkPitch + (idx >> 6) * kFlBlockPitch, (idx >> 3) & 7, v);
   }
}

// =====================================================================================================
// k_frontend_ri -- k_frontend_sym with the packed pair = (re, im) of ONE tree lane instead of tree lanes (l, l + 1) of one row
// =====================================================================================================
// k_frontend_sym packs tree lanes (l, l + 1) of a row into one v_pk_*: its products and lane trees are packed, but everything behind them -- the sign-flipped
// pair sums of the 8 derived rows (56 additions per base bin), re^2 + im^2, the logarithm -- is one instruction per value: 24 M of its 214 M vector
// instructions per 24,576 chunks.  Here a packed register holds (re, im) of the SAME tree lane: the product is x[t] * (re[t], im[t]) -- the sample broadcast to
// both halves by the instruction's op_sel, the two taps an SGPR pair of a basis copy interleaved for it ([f][ii][lp][l % 2][j][re | im]) --, the lane trees are
// the same v_pk_add_f32 count as before, and at their end the tree lanes 2 LP and 2 LP + 1 arrive as P = (rx, ix), Q = (ry, iy): exactly the operands of
//     (re b, im b) = P + Q      (re 128-b, im 128-b) = P - Q      (re 64-b, im 64-b) = P + (-Q.hi, Q.lo)      (re 64+b, im 64+b) = P + (Q.hi, -Q.lo)
// -- four packed additions (op_sel swaps Q's halves, neg_lo / neg_hi negate one: a + (-b) IS a - b) instead of eight, and so for the sums over the l-pairs (28
// packed instead of 56 per base bin); re^2 and im^2 are one v_pk_mul_f32 of the row's pair with itself; and the epilogue -- magnitude, 2^20, log1p -- runs for two
// rows at once (v_pk_add / v_pk_mul / v_pk_fma around the four transcendentals).  Every component of every packed instruction is the IEEE operation the
// reference's expression has at that place, in the same order: the bits of Y, FM and the magnitudes are k_frontend_sym's (tools/fe_bench sym, tests).
// Bin 0 stays on k_frontend_sym's stages (a batch of one, without the tree of its all-zero im row when `zero_im0`).
constexpr int ri_tap_off(int b, int i, int lp) { return b * 2048 + (3 - i) * 512 + lp * 128; }      // bytes: 32 floats per (bin, group, l-pair) = [l % 2][j][re | im]

// g = ((q0+q1)+(q2+q3)) + ((q4+q5)+(q6+q7)),  q_j = x[8j + 2 LP + H] * (re, im)[j]   (stft.c:141-160); xq as fl_tree8, kv = the 8 (re, im) pairs of tree lane 2 LP + H
template <int H>
__device__ __forceinline__ f2v ri_tree8(const f4v (&xq)[4], const f16v &kv)
{
   f2v q[8];
#pragma unroll
   for (int j = 0; j < 8; ++j) {
      const f2v xp = (j & 1) ? __builtin_shufflevector(xq[j >> 1], xq[j >> 1], 2, 3) : __builtin_shufflevector(xq[j >> 1], xq[j >> 1], 0, 1);
      const f2v kp = {kv[2 * j], kv[2 * j + 1]};
      // (as `(f2v){x, x} * kp` hipcc folds only some of the broadcasts into op_sel and copies the others: 113 v_mov per kernel instead of 47;
      // tools/pk_opsel_probe.hip checks what the selector bits do)
#if defined(VADC_RI_ABL) && (VADC_RI_ABL & 1)
      { const float x = H ? xp.y : xp.x; q[j] = (f2v){x, x} * kp; }
#else
      if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(q[j]) : "v"(xp), "s"(kp));
      else        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(q[j]) : "v"(xp), "s"(kp));
#endif
   }
   const f2v a01 = q[0] + q[1], a23 = q[2] + q[3], a45 = q[4] + q[5], a67 = q[6] + q[7];
   const f2v a0123 = a01 + a23, a4567 = a45 + a67;
   return a0123 + a4567;
}

struct RiState {                    // a batch of two base bins
   f2v ta[4], tb[4];                // g_0 (+ g_1), g_2 of tree lane 2 LP + h of base bin B: [2 B + h], as (re, im)
   f2v sa[8], sb[8];                // the 4 row pairs of base bin B: [4 B + q] = (re, im) of row {b, 128 - b, 64 - b, 64 + b}[q]: e0 (+- e1), e2; sa ends up as the row's y
#ifdef VADC_RI_SHADOW
   float xa[8], xb[8];              // diagnostic: rows q = 2, 3 of both bins once more, scalar ([4 B + k], k = 0..3 = e4..e7), and every e they were fed
   float xe[2][4][4];               // [B][LP][k]
   f2v pe[2][4][2];                 // [B][LP][q - 2]: the packed E of the same
   f2v pq[2][2];                    // [B]: P, Q of LP 0
#endif
};

template <int LP>
__device__ __forceinline__ void ri_rows(RiState &st, int B, f2v P, f2v Q)
{
   f2v E[4];
   E[0] = P + Q;
   E[1] = P - Q;
#if defined(VADC_RI_ABL) && (VADC_RI_ABL & 2)
   E[2] = (f2v){P.x - Q.y, P.y + Q.x};
   E[3] = (f2v){P.x + Q.y, P.y - Q.x};
#else
   asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(E[2]) : "v"(P), "v"(Q));      // (rx - iy, ix + ry)
   asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=&v"(E[3]) : "v"(P), "v"(Q));      // (rx + iy, ix - ry)
#endif
#if defined(VADC_RI_DIAG) && (VADC_RI_DIAG & 2)
   asm volatile("s_nop 7" : "+v"(E[2]), "+v"(E[3]) : "v"(P), "v"(Q));
#endif
#ifdef VADC_RI_SHADOW
   {
      __builtin_amdgcn_sched_barrier(0);
      const float e[4] = {P.x - Q.y, P.y + Q.x, P.x + Q.y, P.y - Q.x};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
         float &sa = st.xa[4 * B + k], &sb = st.xb[4 * B + k];
         st.xe[B][LP][k] = e[k];
         if (LP == 0) sa = e[k];
         else if (LP == 1) sa = sa - e[k];
         else if (LP == 2) sb = e[k];
         else { const float t = sb - e[k]; sa = sa + t; }
      }
      st.pe[B][LP][0] = E[2]; st.pe[B][LP][1] = E[3];
      if (LP == 0) { st.pq[B][0] = P; st.pq[B][1] = Q; }
      __builtin_amdgcn_sched_barrier(0);
   }
#endif
#if defined(VADC_RI_ABL) && (VADC_RI_ABL & 8)
#pragma unroll
   for (int q = 0; q < 2; ++q) {
      f2v &sa = st.sa[4 * B + q], &sb = st.sb[4 * B + q];
      if (LP == 0) sa = E[q];
      else if (LP == 1) sa = sa + E[q];
      else if (LP == 2) sb = E[q];
      else { const f2v t = sb + E[q]; sa = sa + t; }
   }
   const float e[4] = {P.x - Q.y, P.y + Q.x, P.x + Q.y, P.y - Q.x};
#pragma unroll
   for (int k = 0; k < 4; ++k) {
      f2v &va = st.sa[4 * B + 2 + (k >> 1)], &vb = st.sb[4 * B + 2 + (k >> 1)];
      float sa = va[k & 1], sb = vb[k & 1];
      if (LP == 0) sa = e[k];
      else if (LP == 1) sa = sa - e[k];
      else if (LP == 2) sb = e[k];
      else { const float t = sb - e[k]; sa = sa + t; }
      va[k & 1] = sa; vb[k & 1] = sb;
   }
#else
#pragma unroll
   for (int q = 0; q < 4; ++q) {
      f2v &sa = st.sa[4 * B + q], &sb = st.sb[4 * B + q];
      if (LP == 0) sa = E[q];
      else if (LP == 1) sa = (q < 2) ? sa + E[q] : sa - E[q];
      else if (LP == 2) sb = E[q];
      else { const f2v t = (q < 2) ? sb + E[q] : sb - E[q]; sa = sa + t; }
   }
#endif
}

// Stage K of a batch of two = (step S = K / 2 = 4 LP + I, base bin B = K % 2), pipelined exactly like SymStages; ca / cb = the taps of tree lanes 2 LP / 2 LP + 1
template <int K>
struct RiStages {
   static __device__ __forceinline__ void run(RiState &st, const float *kf, unsigned xaddr, f16v &ca, f16v &cb, f16v &na, f16v &nb, f4v (&xc)[4], f4v (&xn)[4])
   {
      constexpr int S = K / 2, B = K % 2, LP = S / 4, I = S % 4;
      constexpr int Kn = K + 1;
      constexpr int Sn = (Kn / 2) % 16, Bn = Kn % 2, LPn = Sn / 4, In = Sn % 4;
      constexpr int noff = (Kn == 32 ? 2 * 2048 : 0) + ri_tap_off(Bn, In, LPn);
      if constexpr (B == 0) {
         constexpr int S1 = (S + 1) % 16;
         VADC_FL_LDS16(xn, xaddr, ((S1 % 4) * kFlBlockPitch + (S1 / 4) * 16) * 4);
      }
      VADC_FL_SLOAD2(na, nb, kf, noff, noff + 64);
      const f2v g0 = ri_tree8<0>(xc, ca), g1 = ri_tree8<1>(xc, cb);
      if constexpr (I == 0) { st.ta[2 * B] = g0; st.ta[2 * B + 1] = g1; }
      else if constexpr (I == 1) { st.ta[2 * B] = st.ta[2 * B] + g0; st.ta[2 * B + 1] = st.ta[2 * B + 1] + g1; }       // g_0 + g_1   (stft.c:165)
      else if constexpr (I == 2) { st.tb[2 * B] = g0; st.tb[2 * B + 1] = g1; }
      else {
         const f2v t0 = st.tb[2 * B] + g0, t1 = st.tb[2 * B + 1] + g1;                                                  // g_2 + g_3   (stft.c:166)
         ri_rows<LP>(st, B, st.ta[2 * B] + t0, st.ta[2 * B + 1] + t1);                                                   // stft.c:167, :176-184
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb));
      if constexpr (B == 1) {
         asm volatile("" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3]));
         __builtin_amdgcn_sched_barrier(0);
         RiStages<K + 1>::run(st, kf, xaddr, na, nb, ca, cb, xn, xc);
      } else {
         __builtin_amdgcn_sched_barrier(0);
         RiStages<K + 1>::run(st, kf, xaddr, na, nb, ca, cb, xc, xn);
      }
   }
};
template <>
struct RiStages<32> {
   static __device__ __forceinline__ void run(RiState &, const float *, unsigned, f16v &, f16v &, f16v &, f16v &, f4v (&)[4], f4v (&)[4]) {}
};

// log1p_hw of two values: component for component the same operations
// A transcendental and its first consumer as ONE statement (v_sqrt / v_log / v_exp, the wait state its result needs, the multiply or fma that reads it): a wave of
// this kernel never has two transcendentals in flight.  With two issued back to back -- the natural code for "two rows at once" -- lanes 48 .. 63 (the last of the
// four passes) of the FIRST one's result were, once in ~10^5 workgroups, not what the consumer behind the second one read -- only while waves of another,
// transcendental-heavy kernel (k_lstm_layer's gates) shared the SIMD, never alone on the chip, and no number of wait states behind the pair changed the rate
// (tools/ns_dbg3.py: an engine's front end beside another engine's recurrence; 38 of 6,000 runs).  k_frontend_sym's one-row epilogue never showed it.
__device__ __forceinline__ float ri_sqrt_scaled(float p2)        // v_sqrt_f32(p2) * 2^20
{
   float r;
   asm("v_sqrt_f32 %0, %1\n\ts_nop 0\n\tv_mul_f32 %0, 0x49800000, %0" : "=&v"(r) : "v"(p2));
   return r;
}
__device__ __forceinline__ float ri_ln(float u)                  // v_log_f32(u) * ln 2
{
   float r;
   asm("v_log_f32 %0, %1\n\ts_nop 0\n\tv_mul_f32 %0, 0x3f317218, %0" : "=&v"(r) : "v"(u));
   return r;
}
__device__ __forceinline__ void ri_exp_fma(float t, float u, float &E, float &w)   // E = v_exp_f32(t), w = fma(u, E, -1)
{
   asm("v_exp_f32 %0, %2\n\ts_nop 0\n\tv_fma_f32 %1, %3, %0, -1.0" : "=&v"(E), "=&v"(w) : "v"(t), "v"(u));
}
// log1p_hw of two values: component for component the same operations
__device__ __forceinline__ f2v log1p_hw2(f2v x)
{
   const f2v one = {1.0f, 1.0f};
   const f2v u = one + x;
   const f2v c = x - (u - one);
   f2v y = {ri_ln(u.x), ri_ln(u.y)};
   const f2v t = y * -1.4426950408889634f;
   float e0, e1, w0, w1;
   ri_exp_fma(t.x, u.x, e0, w0);
   ri_exp_fma(t.y, u.y, e1, w1);
   y += (f2v){w0, w1};
   return __builtin_elementwise_fma(c, (f2v){e0, e1}, y);
}

template <typename T, int MODE, int MINW = 4>
__global__ __launch_bounds__(256, MINW) void k_frontend_ri(const T *__restrict__ pcm,          // [n_chunks][1536], 16-byte aligned
                                                          const float *__restrict__ basis,    // [258][256] permuted (k_frontend's): bin 0
                                                          const float *__restrict__ basis_ri, // [33][4][4][2][8][2]: base bins, (re, im) interleaved
                                                          float *__restrict__ Y,              // [n_chunks][129][25]
                                                          float *__restrict__ FM,             // [kBinSplit][fm_stride] partial bin sums
                                                          int n_chunks, ItemMap map, size_t fm_stride, int zero_im0)
{
   constexpr int kFlChunks = fl_chunks(1);
   __shared__ __attribute__((aligned(16))) float xs[kFlChunks * kSymChunkPitch];
   const int tid = threadIdx.x, lane = tid & 63;
   const int wave = (__builtin_amdgcn_readfirstlane(tid >> 6) + (int)blockIdx.x) & 3;       // base-bin split, rotating with the workgroup (k_frontend_sym's OPT 1)
   const long total_pos = (long)n_chunks * kFrames;
   const long p0 = (long)blockIdx.x * 64;
   const int item0 = (int)(p0 / kFrames);
   sym_stage_chunks<T>(xs, pcm, map, item0, n_chunks, tid);
   __syncthreads();

   const long pe = p0 + lane;
   const bool writer = pe < total_pos;
   const long pa_ = writer ? pe : total_pos - 1;
   const int item = (int)(pa_ / kFrames), n = (int)(pa_ - (long)item * kFrames);
   const int chunk = map(item);
   typedef __attribute__((address_space(3))) float lds_f;
   const unsigned xaddr = (unsigned)(uintptr_t)(lds_f *)(xs + (item - item0) * kSymChunkPitch + kFlBlockPitch * n);

   const int f_start = sym_first_bin(wave, 2), f_end = sym_end_bin(wave, 2);
   float *yout = Y + (size_t)chunk * (kBins * kFrames) + n;
   float bin_sum = 0.0f;
   f16v ca, cb, na, nb;
   f4v xc[4], xn[4];
   VADC_FL_LDS16(xc, xaddr, 0);
   // the samples must have ARRIVED before the compiler may touch their registers (it does not know the asm's reads are in flight, and the branch below makes it
   // copy them): the wait is tied to them
   asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]));
   __builtin_amdgcn_sched_barrier(0);

   auto emit_row = [&](int bin, float re, float im) {                             // bin 0's three rows (k_frontend_sym's emit_row)
      const float re2 = re * re, im2 = im * im;
      const float p2 = re2 + im2;
      float val;
      if (MODE == 0) { val = log1p_hw(__builtin_amdgcn_sqrtf(p2) * 1048576.0f); bin_sum += val; }
      else val = sqrtf(p2);
      if (writer) yout[bin * kFrames] = val;
   };
   // two rows at once: (re, im) pairs ra, rb of bins ba, bb (in this order into the partial bin sum: misc.c:55-59)
   auto emit_pair = [&](int ba, int bb, f2v ra, f2v rb) {
      const f2v sqa = ra * ra, sqb = rb * rb;
      const f2v p2 = {sqa.x + sqa.y, sqb.x + sqb.y};
      f2v val;
      if (MODE == 0) {
         const f2v m = {ri_sqrt_scaled(p2.x), ri_sqrt_scaled(p2.y)};
         val = log1p_hw2(m);                                                     // misc.c:42-45
         bin_sum += val.x; bin_sum += val.y;
      } else val = (f2v){sqrtf(p2.x), sqrtf(p2.y)};                              // stft.c:209
      if (writer) { yout[ba * kFrames] = val.x; yout[bb * kFrames] = val.y; }
#if defined(VADC_RI_DIAG) && (VADC_RI_DIAG & 1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
   };
   constexpr int kImOffB = kBins * kFilterLen * 4;
   int f_loop = f_start;
   if (wave == 0) {                                                               // split 0: bin 0 alone on k_frontend_sym's stages, then bins 1..8 in pairs
      VADC_FL_SLOAD2(ca, cb, basis, fl_tap_off(0, 0, 0), fl_tap_off(0, 0, 0) + kImOffB);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca), "+s"(cb));
      __builtin_amdgcn_sched_barrier(0);
      SymState<1> s1;
      if (zero_im0) SymStages<1, 0, true>::run(s1, basis, xaddr, ca, cb, na, nb, xc, xn);
      else          SymStages<1, 0, false>::run(s1, basis, xaddr, ca, cb, na, nb, xc, xn);
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(s1.sa[k]));
      emit_row(0, s1.sa[0], s1.sa[1]); emit_row(128, s1.sa[2], s1.sa[3]); emit_row(64, s1.sa[4], s1.sa[5]);     // rows 0, 128, 64 (64 - 0 and 64 + 0 coincide)
      f_loop = 1;
   }
   {
      const float *k0 = basis_ri + (size_t)f_loop * 512;
      VADC_FL_SLOAD2(ca, cb, k0, ri_tap_off(0, 0, 0), ri_tap_off(0, 0, 0) + 64);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ca), "+s"(cb));
      __builtin_amdgcn_sched_barrier(0);
   }
#pragma unroll 1
   for (int f = f_loop; f + 2 <= f_end; f += 2) {
      const float *kf = basis_ri + (size_t)f * 512;                              // wave-uniform
      RiState st;
      RiStages<0>::run(st, kf, xaddr, ca, cb, na, nb, xc, xn);
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(st.sa[k]));            // pin the trees above the epilogue (see k_frontend_fl)
#pragma unroll
      for (int B = 0; B < 2; ++B) {
         const int b = f + B;                                                     // 1 .. 32: rows b, 128 - b and -- below 32 -- 64 - b, 64 + b (at 32 they are rows 32 and 96 again)
#ifdef VADC_RI_SHADOW
         for (int k = 0; k < 4; ++k) {
            const float pk_ = st.sa[4 * B + 2 + (k >> 1)][k & 1], sh = st.xa[4 * B + k];
            if (__float_as_uint(pk_) != __float_as_uint(sh)) {
               printf("RI-MISMATCH wg %d wave %d lane %d bin %d k %d | LP0: P %a %a Q %a %a  E2 %a %a E3 %a %a | scalar e %a %a %a %a\n", (int)blockIdx.x, wave, lane, b, k,
                      st.pq[B][0].x, st.pq[B][0].y, st.pq[B][1].x, st.pq[B][1].y, st.pe[B][0][0].x, st.pe[B][0][0].y, st.pe[B][0][1].x, st.pe[B][0][1].y,
                      st.xe[B][0][0], st.xe[B][0][1], st.xe[B][0][2], st.xe[B][0][3]);
            }
         }
#endif
#if defined(VADC_RI_ABL) && (VADC_RI_ABL & 16)
         emit_row(b, st.sa[4 * B].x, st.sa[4 * B].y); emit_row(128 - b, st.sa[4 * B + 1].x, st.sa[4 * B + 1].y);
         if (b < 32) { emit_row(64 - b, st.sa[4 * B + 2].x, st.sa[4 * B + 2].y); emit_row(64 + b, st.sa[4 * B + 3].x, st.sa[4 * B + 3].y); }
#elif defined(VADC_RI_ABL) && (VADC_RI_ABL & 32)
         emit_pair(b, 128 - b, st.sa[4 * B], st.sa[4 * B + 1]);
         emit_pair(min(64 - b, 63), max(64 + b, 65) > 96 ? 96 : max(64 + b, 65), st.sa[4 * B + 2], st.sa[4 * B + 3]);      // (timing probe: no branch around the second pair; b = 32 writes garbage rows)
#else
         emit_pair(b, 128 - b, st.sa[4 * B], st.sa[4 * B + 1]);
         if (b < 32) emit_pair(64 - b, 64 + b, st.sa[4 * B + 2], st.sa[4 * B + 3]);
#endif
      }
   }
   if (MODE == 0 && writer) FM[wave * fm_stride + (size_t)chunk * kFrames + n] = bin_sum;   // /129 by the reader (misc.c:60)
}

