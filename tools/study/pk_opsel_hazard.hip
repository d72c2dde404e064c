// tools/study/pk_opsel_hazard.hip -- a fully synthetic reproducer of the gfx950 behaviour behind tools/check_pk_opsel.py; no kernel of the product is needed.
//
//   VICTIM     waves that execute, again and again,   E = v_pk_add_f32(P, Q) op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]   (low result = P.lo - Q.HI: the low result takes the
//              SECOND source's HIGH dword) on fresh P, Q and compare both results with scalar v_sub_f32 / v_add_f32 of the same registers.
//   NEIGHBOUR  a kernel on another stream whose waves share the SIMDs: barrier + LDS + vector ALU, plus (kind 1) MFMAs on two accumulators in turn or (kind 2) six
//              BACK-TO-BACK DEPENDENT v_mfma_f32_16x16x32_f16 on ONE accumulator (each takes the previous one's result as its SrcC) -- what the inner loop of any GEMM is.
//
//   measured (MI355X, ROCm 7.2; 1.6e9 executions of the instruction per line):
//      alone on the chip                                   0 wrong results
//      beside kind 0 / kind 1                               0 / 0
//      beside kind 2                                        7,351,488 wrong results (1 in 220), ALL in lanes 48 .. 63, low result = P.lo alone (Q.hi read as zero), high result right
//      the same sum with the swapped pair as the FIRST source (v_pk_add_f32 E, Q, P op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]), beside kind 2:   0
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/study/pk_opsel_hazard tools/study/pk_opsel_hazard.hip ;  tools/study/pk_opsel_hazard [launches = 1000]
//   (-DWITH_LSTM: the neighbour is the product's k_lstm_layer pair instead -- 137,920 wrong results in 2.5e9; its first block of twelve MFMAs, six dependent per accumulator,
//    is what it takes: -DVADC_LSTM_ABL_STOP=1..4, -DVADC_LSTM_ABL_PROLOGUE_NOMFMA)
#ifdef WITH_LSTM          // -DWITH_LSTM: the neighbour is the product's k_lstm_layer pair (as in ri_repro.hip) instead of the synthetic one
#include "../../vadc_amd/csrc/kernels_lstm.hip"
#include <vector>
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>
typedef float f2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f16v __attribute__((ext_vector_type(16)));
template <int SRC0>
__global__ __launch_bounds__(256, 4) void k_victim(const float *seed, unsigned *bad, unsigned *bad_lane_hist, int iters)
{
   const int lane = threadIdx.x & 63;
   float a = seed[threadIdx.x] + blockIdx.x * 1e-3f, b = seed[256 + threadIdx.x], c = seed[512 + threadIdx.x], d = seed[768 + threadIdx.x];
   unsigned wrong = 0;
   // the surroundings the instruction has in k_frontend_ri: taps arrive by scalar loads issued right behind it, products take an SGPR pair and a broadcast sample
   f16v ta, tb;
   asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ta), "=&s"(tb) : "s"(seed));
   for (int i = 0; i < iters; ++i) {
      // fresh operands as in a tree: eight products x * (re, im) of an SGPR pair with the sample broadcast, summed pairwise
      f2v q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
         const f2v xp = {a + 0.03125f * j, b - 0.0625f * j}, kp = {(j & 1) ? tb[2 * (j >> 1)] : ta[2 * (j >> 1)], (j & 1) ? tb[2 * (j >> 1) + 1] : ta[2 * (j >> 1) + 1]};
         if (j < 4) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(q[j]) : "v"(xp), "s"(kp));
         else       asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(q[j]) : "v"(xp), "s"(kp));
      }
      f2v P = ((q[0] + q[1]) + (q[2] + q[3])) + (f2v){c, d};
      f2v Q = ((q[4] + q[5]) + (q[6] + q[7])) - (f2v){d, c};
      f2v E;
#if defined(VICTIM_OP) && VICTIM_OP == 1       // v_pk_mul_f32: low = P.lo * Q.hi, high = P.hi * Q.lo
      if (SRC0) asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(E) : "v"(P), "v"(Q));
      else      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(E) : "v"(P), "v"(Q));
#elif defined(VICTIM_OP) && VICTIM_OP == 2     // v_pk_fma_f32 with the swap on the THIRD source: low = P.lo * P.lo + Q.hi, high = P.hi * P.hi + Q.lo
      asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=&v"(E) : "v"(P), "v"(Q));
#else
      if (SRC0) asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=&v"(E) : "v"(P), "v"(Q));
      else      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(E) : "v"(P), "v"(Q));
#endif
      // the next stage's taps are requested right behind it (different tuples in turn, as the kernel's double buffer)
      if (i & 1) asm volatile("s_load_dwordx16 %0, %1, 0x80" : "=&s"(ta) : "s"(seed));
      else       asm volatile("s_load_dwordx16 %0, %1, 0xc0" : "=&s"(tb) : "s"(seed));
      float want_lo, want_hi;
#if defined(VICTIM_OP) && VICTIM_OP == 1
      asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %4, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(P.x), "v"(Q.y), "v"(P.y), "v"(Q.x));
#elif defined(VICTIM_OP) && VICTIM_OP == 2
      asm volatile("v_fma_f32 %0, %2, %2, %3\n\tv_fma_f32 %1, %4, %4, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(P.x), "v"(Q.y), "v"(P.y), "v"(Q.x));
#else
      asm volatile("v_sub_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(P.x), "v"(Q.y), "v"(P.y), "v"(Q.x));
#endif
      if (__float_as_uint(E.x) != __float_as_uint(want_lo) || __float_as_uint(E.y) != __float_as_uint(want_hi)) {
         if (!wrong && atomicAdd(bad + 66, 1u) == 0) { float *dbg = reinterpret_cast<float *>(bad + 67); dbg[0] = P.x; dbg[1] = P.y; dbg[2] = Q.x; dbg[3] = Q.y; dbg[4] = E.x; dbg[5] = E.y; dbg[6] = want_lo; dbg[7] = want_hi; bad[75] = lane; bad[76] = i; }
         ++wrong;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ta), "+s"(tb));
      // next iteration's values depend on this one's (no hoisting), stay bounded
      a = a * 0.75f + E.x * 0.125f + 0.01f; b = b * 0.75f - E.y * 0.125f; c = c * 0.5f + want_lo * 0.25f; d = d * 0.5f + 0.125f * want_hi + 0.02f;
   }
   if (wrong) { atomicAdd(bad, wrong); atomicAdd(bad_lane_hist + lane, 1u); }
   if (a + b + c + d == 12345.678f) bad[1] = 1;       // keep the chain alive
}

// the neighbour: workgroups of 8 waves that take turns at a barrier, exchange through the LDS and do some vector arithmetic -- what a recurrence kernel looks like to its SIMD.
// KIND 0: that alone; 1: + six MFMAs on each of two accumulators, the two chains in turn; 2: + six MFMAs BACK TO BACK on ONE accumulator, each taking the previous one's result
template <int KIND>
__global__ __launch_bounds__(512, 4) void k_neighbour(float *out, int iters)
{
   __shared__ float sh[512];
   float v = threadIdx.x * 0.001f;
   typedef _Float16 h8v __attribute__((ext_vector_type(8)));
   typedef float f4v __attribute__((ext_vector_type(4)));
   for (int i = 0; i < iters; ++i) {
      sh[threadIdx.x] = v;
      __syncthreads();
      if (KIND) {
         h8v a8, b8; for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(v + e); b8[e] = (_Float16)(v - e); }
         f4v acc = {v, v, v, v}, acc2 = {v, -v, v, -v};
#pragma unroll
         for (int r = 0; r < 6; ++r) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
            if (KIND == 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, acc2, 0, 0, 0);
         }
         v += (acc[0] + acc2[1]) * 1e-30f;
      }
      v = sh[(threadIdx.x + 37) & 511] * 0.5f + v * 0.25f + 0.1f;
      v = __builtin_amdgcn_rcpf(1.0f + v * v) + v * 0.125f;
      __syncthreads();
   }
   out[blockIdx.x * 512 + threadIdx.x] = v;
}

int main(int argc, char **argv)
{
   const int R = argc > 1 ? atoi(argv[1]) : 1000;
   const int n_iters = getenv("NEIGHBOUR_ITERS") ? atoi(getenv("NEIGHBOUR_ITERS")) : 400;     // length of a neighbour workgroup's life
   float h[1024]; srand(3); for (float &x : h) x = ((rand() % 2001) - 1000) / 997.0f;
   float *seed, *nout; unsigned *bad;
   CK(hipMalloc(&seed, sizeof h)); CK(hipMemcpy(seed, h, sizeof h, hipMemcpyHostToDevice));
   CK(hipMalloc(&nout, 2048 * 512 * 4));
   CK(hipHostMalloc(&bad, (2 + 64 + 16) * 4));
   hipStream_t sv, sn;
   CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sn, hipStreamNonBlocking));
#ifdef WITH_LSTM
   const int LS = 10240, ltiles = LS / 16, layers = getenv("LSTM_LAYERS") ? atoi(getenv("LSTM_LAYERS")) : 3;      // bit 0: layer 0's kernel, bit 1: layer 1's
   hipStream_t sn2; CK(hipStreamCreateWithFlags(&sn2, hipStreamNonBlocking));
   _Float16 *lx, *lh0; float *lwb, *lhs, *lcs, *lprobs;
   const size_t tile_halves = (size_t)ltiles * 7 * 2 * 16 * 64;
   CK(hipMalloc(&lx, tile_halves * 2)); CK(hipMalloc(&lh0, tile_halves * 2));
   { std::vector<_Float16> hx(tile_halves); for (auto &v : hx) v = (_Float16)(((rand() % 2001) - 1000) / 4000.0f); CK(hipMemcpy(lx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(lh0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); }
   { std::vector<float> hw((size_t)2 * 256 * 128 + 2 * 256 + 128 + 2); for (auto &v : hw) v = ((rand() % 2001) - 1000) / 8000.0f; CK(hipMalloc(&lwb, hw.size() * 4)); CK(hipMemcpy(lwb, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); }
   CK(hipMalloc(&lhs, (size_t)LS * 128 * 4)); CK(hipMalloc(&lcs, (size_t)LS * 128 * 4)); CK(hipMalloc(&lprobs, (size_t)LS * 2 * 4));
   CK(hipMemset(lhs, 0, (size_t)LS * 128 * 4)); CK(hipMemset(lcs, 0, (size_t)LS * 128 * 4));
   vadc::LstmWeights lw; lw.w = lwb; lw.wT = lwb; lw.b = lwb + 2 * 256 * 128; lw.dec_w = lw.b + 2 * 256; lw.dec_b = lw.dec_w + 128;
#endif
#ifdef WITH_LSTM
   const int kinds = 2;      // 0 = alone, 1 = beside k_lstm_layer x 2
#else
   const int kinds = 4;      // 0 = alone, 1 .. 3 = beside k_neighbour<0 .. 2>
#endif
   static const char *names[4] = {"alone on the chip                                      ",
#ifdef WITH_LSTM
                                  "beside the product's k_lstm_layer pair                  ", "", ""};
#else
                                  "beside a neighbour without MFMAs                        ", "beside a neighbour with MFMAs on two accumulators in turn", "beside a neighbour with DEPENDENT back-to-back MFMAs    "};
#endif
   for (int neighbour = 0; neighbour < kinds; ++neighbour)
      for (int src0 = 0; src0 < 2; ++src0) {
         std::atomic<bool> stop{false};
#ifdef WITH_LSTM
         std::thread th([&] { if (!neighbour) return; (void)hipSetDevice(0); while (!stop) { for (int k = 0; k < 8; ++k) {
               if (layers & 1) vadc::launch_lstm_layer(0, reinterpret_cast<const float *>(lx), reinterpret_cast<float *>(lh0), lw, lhs, lcs, lprobs, LS, 1, 0, 1, sn, 0, 7, nullptr, 0, nullptr, 0);
               if (layers & 2) vadc::launch_lstm_layer(1, reinterpret_cast<const float *>(lx), reinterpret_cast<float *>(lh0), lw, lhs, lcs, lprobs, LS, 1, 0, 1, sn2, 0, 7, nullptr, 0, nullptr, 0); }
            (void)hipStreamSynchronize(sn); (void)hipStreamSynchronize(sn2); } });
#else
         std::thread th([&] { if (!neighbour) return; (void)hipSetDevice(0); while (!stop) { for (int k = 0; k < 8; ++k) {
               if (neighbour == 1) hipLaunchKernelGGL(k_neighbour<0>, dim3(1024), dim3(512), 0, sn, nout, n_iters);
               if (neighbour == 2) hipLaunchKernelGGL(k_neighbour<1>, dim3(1024), dim3(512), 0, sn, nout, n_iters);
               if (neighbour == 3) hipLaunchKernelGGL(k_neighbour<2>, dim3(1024), dim3(512), 0, sn, nout, n_iters); }
            (void)hipStreamSynchronize(sn); } });
#endif
         for (int i = 0; i < 82; ++i) bad[i] = 0;
         int bad_launches = 0; unsigned long long executions = 0;
         for (int r = 0; r < R; ++r) {
            const unsigned before = bad[0];
            if (src0) hipLaunchKernelGGL(k_victim<1>, dim3(1024), dim3(256), 0, sv, seed, bad, bad + 2, 400);
            else      hipLaunchKernelGGL(k_victim<0>, dim3(1024), dim3(256), 0, sv, seed, bad, bad + 2, 400);
            CK(hipStreamSynchronize(sv));
            executions += 1024ull * 4 * 400;
            if (bad[0] != before) ++bad_launches;
         }
         stop = true; th.join();
         if (bad[66]) { const float *dbg = reinterpret_cast<const float *>(bad + 67); printf("   first wrong result: lane %u iteration %u: P (%a, %a) Q (%a, %a) -> E (%a, %a), scalar (%a, %a)\n", bad[75], bad[76], dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]); }
         unsigned lo = 0, hi = 0; for (int l = 0; l < 64; ++l) (l < 48 ? lo : hi) += bad[2 + l];
         printf("swapped pair as the %s source, %s: %u wrong results in %.2e executions (%d of %d launches; waves with a wrong lane 0..47: %u, lane 48..63: %u)\n",
                src0 ? "FIRST " : "SECOND", names[neighbour], bad[0], (double)executions, bad_launches, R, lo, hi);
      }
   return 0;
}
