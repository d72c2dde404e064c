// fe_bench.hip -- device micro-benchmark for k_frontend tuning knobs (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/fe_bench.hip -o tools/fe_bench
//   tools/fe_bench [n_chunks]
// Times every variant with HIP events on random s16 input and checks that all variants produce the same
// bits (magnitudes in MODE 1; MODE 0 outputs except the rotated-order mean of STAGGER variants).
#include "../vadc_amd/csrc/kernels_frontend.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <functional>
#include <atomic>
#include <thread>
#include <unistd.h>

using namespace vadc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int NT, int MINW, int SHIFT, int LOCK, int ABL, int PIPE = 0, int PK = 0>
static float run(const char *name, const int16_t *pcm, const float *basis, float *Y, float *FM, int n, int reps)
{
   const long waves = ((long)n * kBlocks + kLanesOut - 1) / kLanesOut;
   const int wpb = NT / 64;
   const dim3 blocks((unsigned)((waves + wpb - 1) / wpb), kBinSplit);
   const size_t fm_stride = (size_t)n * kFrames;
   const ItemMap map{n, 0, n};
   hipEvent_t a, b;
   CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   hipLaunchKernelGGL((k_frontend<int16_t, MODE, NT, MINW, SHIFT, LOCK, ABL, PIPE, PK>), blocks, dim3(NT), 0, 0, pcm, basis, Y, FM, n, map, fm_stride);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(a, 0));
   for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL((k_frontend<int16_t, MODE, NT, MINW, SHIFT, LOCK, ABL, PIPE, PK>), blocks, dim3(NT), 0, 0, pcm, basis, Y, FM, n, map, fm_stride);
   CK(hipEventRecord(b, 0));
   CK(hipEventSynchronize(b));
   float ms = 0;
   CK(hipEventElapsedTime(&ms, a, b));
   ms /= reps;
   printf("%-44s mode %d  %8.4f ms  %7.2f Mchunks/s\n", name, MODE, ms, n / ms / 1e3);
   return ms;
}

// "lstm": a latency-bound MFMA + LDS chain on 16 workgroups of 512 threads on CUs 0..7, concurrently with every timed launch
__global__ __launch_bounds__(512) void k_dummy_chain(float *out, int iters)
{
   __shared__ float sh[8192];
   typedef _Float16 h8 __attribute__((ext_vector_type(8)));
   typedef float f4 __attribute__((ext_vector_type(4)));
   f4 acc = {0, 0, 0, 0};
   h8 a, b;
   for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)0.5f; }
   for (int i = threadIdx.x; i < 8192; i += 512) sh[i] = i;
   __syncthreads();
   for (int it = 0; it < iters; ++it) {
      for (int j = 0; j < 12; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
      sh[(threadIdx.x * 4 + it) & 8191] = acc[0];
      __syncthreads();
      acc[1] += sh[(threadIdx.x * 7 + it) & 8191];
   }
   out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1];
}
static hipStream_t g_stb = 0;
static float *g_dummy_out = nullptr;
static int g_dummy_iters = 0;
static int g_removed = 0;
static hipStream_t g_st = 0;                     // "mask": a stream restricted to CUs 8..255 like the engine's stream A
static bool g_cold = false;
static void *g_scratch = nullptr;
static int *g_counter = nullptr;
static int g_slots = 0;                          // persistent variants: workgroups in the grid
template <int MODE, int NB, int MINW, int ABL = 0, int NPS = 1, bool PERSIST = false>
static float run_fl(const char *name, const int16_t *pcm, const float *basis, float *Y, float *FM, int n, int reps, int dyn_lds = 0)
{
   const dim3 blocks((unsigned)(((long)n * kFrames + 64 * NPS - 1) / (64 * NPS)));
   const size_t fm_stride = (size_t)n * kFrames;
   const ItemMap map{n, 0, n};
   hipEvent_t a, b;
   CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   hipLaunchKernelGGL((k_frontend_fl<int16_t, MODE, NB, MINW, ABL, NPS, PERSIST>), PERSIST ? dim3(g_slots / NPS) : blocks, dim3(256 * NPS), dyn_lds, 0, pcm, basis, Y, FM, n, map, fm_stride, g_counter);
   CK(hipDeviceSynchronize());
   float ms = 0;
   if (g_cold) {                                  // single launches, each after an unrelated 256 MB memset (cold caches, no back-to-back overlap)
      for (int r = 0; r < reps; ++r) {
         CK(hipMemsetAsync(g_scratch, r, (size_t)256 << 20, g_st));
         CK(hipStreamSynchronize(g_st));
         if (g_dummy_iters) hipLaunchKernelGGL(k_dummy_chain, dim3(16), dim3(512), 0, g_stb, g_dummy_out, g_dummy_iters);
         CK(hipEventRecord(a, g_st));
         if (PERSIST) CK(hipMemsetAsync(g_counter, 0, 4, g_st));
         hipLaunchKernelGGL((k_frontend_fl<int16_t, MODE, NB, MINW, ABL, NPS, PERSIST>), PERSIST ? dim3(g_slots / NPS) : blocks, dim3(256 * NPS), dyn_lds, g_st, pcm, basis, Y, FM, n, map, fm_stride, g_counter);
         CK(hipEventRecord(b, g_st));
         CK(hipEventSynchronize(b));
         float t = 0;
         CK(hipEventElapsedTime(&t, a, b));
         ms += t;
      }
   } else {
      CK(hipEventRecord(a, g_st));
      for (int r = 0; r < reps; ++r) {
         if (PERSIST) CK(hipMemsetAsync(g_counter, 0, 4, g_st));
         hipLaunchKernelGGL((k_frontend_fl<int16_t, MODE, NB, MINW, ABL, NPS, PERSIST>), PERSIST ? dim3(g_slots / NPS) : blocks, dim3(256 * NPS), dyn_lds, g_st, pcm, basis, Y, FM, n, map, fm_stride, g_counter);
      }
      CK(hipEventRecord(b, g_st));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
   }
   ms /= reps;
   printf("%-44s mode %d  %8.4f ms  %7.2f Mchunks/s\n", name, MODE, ms, n / ms / 1e3);
   return ms;
}

int main(int argc, char **argv)
{
   const int n = argc > 1 ? atoi(argv[1]) : 16384;
   const int reps = 5;
   std::vector<int16_t> h_pcm((size_t)n * kChunk);
   srand(1);
   for (auto &v : h_pcm) v = (int16_t)((rand() % 20001) - 10000);
   std::vector<float> h_basis((size_t)kFilters * kFilterLen + 1024);   // slack: the pipelines prefetch past the last filter
   for (auto &v : h_basis) v = (float)((rand() % 2001) - 1000) / 1000.0f;
   if (argc > 4 && !strcmp(argv[4], "real")) {     // a windowed-DFT basis in the engine's permuted layout and speech-like input (low level, silences)
      for (int f = 0; f < kFilters; ++f)
         for (int ii = 0; ii < 4; ++ii)
            for (int lp = 0; lp < 4; ++lp)
               for (int j = 0; j < 8; ++j)
                  for (int b = 0; b < 2; ++b) {
                     const int t = 64 * (3 - ii) + 8 * j + (2 * lp + b), bin = f % kBins;
                     const double w = 0.5 - 0.5 * cos(2.0 * M_PI * t / 256.0), ph = 2.0 * M_PI * bin * t / 256.0;
                     h_basis[(size_t)f * 256 + ii * 64 + lp * 16 + j * 2 + b] = (float)(f < kBins ? w * cos(ph) : -w * sin(ph));
                  }
      for (size_t i = 0; i < h_pcm.size(); ++i) {
         const size_t seg = i / 8000;
         const double amp = (seg % 3 == 0) ? 3.0 : 2500.0;
         h_pcm[i] = (int16_t)(amp * sin(0.02 * (double)(i % 16000)) + amp * 0.3 * ((rand() % 2001) - 1000) / 1000.0);
      }
      printf("real-like basis and input\n");
   }
   int16_t *pcm; float *basis, *Y0, *Y1, *FM;
   CK(hipMalloc(&pcm, h_pcm.size() * 2)); CK(hipMalloc(&basis, h_basis.size() * 4));
   CK(hipMalloc(&Y0, (size_t)n * kBins * kFrames * 4)); CK(hipMalloc(&Y1, (size_t)n * kBins * kFrames * 4));
   CK(hipMalloc(&FM, (size_t)kBinSplit * n * kFrames * 4));
   CK(hipMemcpy(pcm, h_pcm.data(), h_pcm.size() * 2, hipMemcpyHostToDevice));
   CK(hipMemcpy(basis, h_basis.data(), h_basis.size() * 4, hipMemcpyHostToDevice));
   std::vector<float> ref((size_t)n * kBins * kFrames), got(ref.size());
   auto check = [&](const char *name, float *Y) {
      CK(hipMemcpy(got.data(), Y, got.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < got.size(); ++i) if (memcmp(&got[i], &ref[i], 4)) ++bad;
      printf("   %-41s %s (%zu mismatching words)\n", name, bad ? "MISMATCH" : "bit-identical to baseline", bad);
   };
   printf("n_chunks = %d\n", n);
   if (argc > 5 && !strncmp(argv[5], "mask", 4)) {  // maskN: the first N CUs (mask bit order) are taken away, "mask" = 8
      const int removed = argv[5][4] ? atoi(argv[5] + 4) : 8;
      g_removed = removed;
      uint32_t m[8];
      for (int w = 0; w < 8; ++w) m[w] = 0xffffffffu;
      for (int cu = 0; cu < removed; ++cu) m[cu / 32] &= ~(1u << (cu % 32));
      CK(hipExtStreamCreateWithCUMask(&g_st, 8, m));
      printf("stream masked to CUs %d..255\n", removed);
   }
   if (argc > 6 && !strcmp(argv[6], "lstm")) {
      uint32_t m[8] = {0xffu, 0, 0, 0, 0, 0, 0, 0};
      CK(hipExtStreamCreateWithCUMask(&g_stb, 8, m));
      CK(hipMalloc(&g_dummy_out, 16 * 512 * 4));
      g_dummy_iters = 1000;
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipLaunchKernelGGL(k_dummy_chain, dim3(16), dim3(512), 0, g_stb, g_dummy_out, 100);
      CK(hipEventRecord(a, g_stb));
      hipLaunchKernelGGL(k_dummy_chain, dim3(16), dim3(512), 0, g_stb, g_dummy_out, g_dummy_iters);
      CK(hipEventRecord(b, g_stb)); CK(hipEventSynchronize(b));
      float t; CK(hipEventElapsedTime(&t, a, b));
      g_dummy_iters = (int)(g_dummy_iters * 1.5f / t);           // ~1.5 ms
      printf("concurrent chain kernel on CUs 0..7: %d iterations (~1.5 ms)\n", g_dummy_iters);
   }
   if (argc > 3 && !strcmp(argv[3], "cold")) { g_cold = true; CK(hipMalloc(&g_scratch, (size_t)256 << 20)); }
   if (argc > 2 && !strcmp(argv[2], "sym")) {         // k_frontend_sym (33 base bins + symmetries) against k_frontend_fl (all 129 bins): bits and time
      // a windowed-DFT basis with the reference's symmetries BY CONSTRUCTION: cos / sin from one quadrant table
      double qc[65];
      for (int k = 0; k <= 64; ++k) qc[k] = cos(2.0 * M_PI * k / 256.0);
      qc[64] = 0.0;                                      // sin(0) = 0 exactly: the im row of bin 0 is all zeros, as in the shipped tensor
      auto cosi = [&](int m) { m &= 255; if (m > 128) m = 256 - m; return m <= 64 ? qc[m] : -qc[128 - m]; };
      auto sini = [&](int m) { return cosi(m - 64); };
      std::vector<float> nat((size_t)kFilters * 256);
      for (int f = 0; f < kFilters; ++f)
         for (int t = 0; t < 256; ++t) {
            const int bin = f % kBins;
            const double w = 0.5 - 0.5 * cosi(t);
            nat[(size_t)f * 256 + t] = (float)(f < kBins ? w * cosi(bin * t) : -w * sini(bin * t));
         }
      for (int f = 0; f < kFilters; ++f)
         for (int ii = 0; ii < 4; ++ii)
            for (int lp = 0; lp < 4; ++lp)
               for (int j = 0; j < 8; ++j)
                  for (int b = 0; b < 2; ++b)
                     h_basis[(size_t)f * 256 + ii * 64 + lp * 16 + j * 2 + b] = nat[(size_t)f * 256 + 64 * (3 - ii) + 8 * j + (2 * lp + b)];
      CK(hipMemcpy(basis, h_basis.data(), h_basis.size() * 4, hipMemcpyHostToDevice));
      // k_frontend_ri's copy: base bins 0..32, (re, im) of a tap side by side ([f][ii][lp][l % 2][j][re | im], engine.hip)
      std::vector<float> h_ri((size_t)35 * 512 + 64, 0.0f);
      for (int f = 0; f < 33; ++f)
         for (int ii = 0; ii < 4; ++ii)
            for (int lp = 0; lp < 4; ++lp)
               for (int h = 0; h < 2; ++h)
                  for (int j = 0; j < 8; ++j)
                     for (int c = 0; c < 2; ++c)
                        h_ri[(size_t)f * 512 + ii * 128 + lp * 32 + h * 16 + j * 2 + c] = nat[(size_t)(c ? kBins + f : f) * 256 + 64 * (3 - ii) + 8 * j + (2 * lp + h)];
      float *basis_ri;
      CK(hipMalloc(&basis_ri, h_ri.size() * 4));
      CK(hipMemcpy(basis_ri, h_ri.data(), h_ri.size() * 4, hipMemcpyHostToDevice));
      float *FM1;
      CK(hipMalloc(&FM1, (size_t)kBinSplit * n * kFrames * 4));
      const size_t fm_stride = (size_t)n * kFrames;
      const ItemMap map{n, 0, n};
      const dim3 grid((unsigned)(((long)n * kFrames + 63) / 64));
      auto time_it = [&](const char *name, auto launch) {
         hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
         launch(); CK(hipDeviceSynchronize());
         CK(hipEventRecord(a, g_st));
         for (int r = 0; r < reps; ++r) launch();
         CK(hipEventRecord(b, g_st)); CK(hipEventSynchronize(b));
         float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
         printf("%-44s %8.4f ms  %7.2f Mchunks/s\n", name, ms, n / ms / 1e3);
      };
      time_it("fl nb3 (all 129 bins), magnitude", [&] { hipLaunchKernelGGL((k_frontend_fl<int16_t, 1, 3, 4, 0, 1>), grid, dim3(256), 0, g_st, pcm, basis, Y0, FM, n, map, fm_stride, nullptr); });
      CK(hipMemcpy(ref.data(), Y0, ref.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb3 w4, magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 3, 4>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      check("sym nb3 w4", Y1);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb2 w4, magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      check("sym nb2 w4", Y1);
      // OPT variants of the NB = 2 kernel: 1 = split rotation, 2 = bin 0 alone without its zero im tree
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb2 OPT1, magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 1>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
      check("sym nb2 OPT1", Y1);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb2 OPT2, magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 2>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
      check("sym nb2 OPT2", Y1);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb2 OPT2 (zero_im0 = 0), magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 2>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 0); });
      check("sym nb2 OPT2 z0", Y1);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb2 OPT3, magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 3>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
      check("sym nb2 OPT3", Y1);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("sym nb2 OPT11 (XCD-major blocks), magnitude", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 11>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
      check("sym nb2 OPT11", Y1);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("ri (re, im pairs), magnitude", [&] { hipLaunchKernelGGL((k_frontend_ri<int16_t, 1, 4>), grid, dim3(256), 0, g_st, pcm, basis, basis_ri, Y1, FM1, n, map, fm_stride, 1); });
      check("ri", Y1);
      if (getenv("FE_RI_DEBUG")) {                        // mismatching words by bin
         std::vector<long> by_bin(kBins, 0);
         for (size_t i = 0; i < got.size(); ++i) if (memcmp(&got[i], &ref[i], 4)) ++by_bin[(i / kFrames) % kBins];
         for (int b = 0; b < kBins; ++b) if (by_bin[b]) printf("      bin %3d: %ld\n", b, by_bin[b]);
         printf("      first words of bin 1: got %g %g %g want %g %g %g\n", got[25], got[26], got[27], ref[25], ref[26], ref[27]);
         return 0;
      }
      CK(hipMemset(Y1, 0, ref.size() * 4));
      time_it("ri (zero_im0 = 0), magnitude", [&] { hipLaunchKernelGGL((k_frontend_ri<int16_t, 1, 4>), grid, dim3(256), 0, g_st, pcm, basis, basis_ri, Y1, FM1, n, map, fm_stride, 0); });
      check("ri z0", Y1);
      {  // log mode: Y and the partial bin sums of k_frontend_ri against k_frontend_sym OPT 3, bit for bit
         std::vector<float> ya(ref.size()), yb(ref.size()), fa((size_t)kBinSplit * n * kFrames), fb(fa.size());
         float *FM2; CK(hipMalloc(&FM2, fa.size() * 4));
         hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 3>), grid, dim3(256), 0, g_st, pcm, basis, Y0, FM1, n, map, fm_stride, 1);
         hipLaunchKernelGGL((k_frontend_ri<int16_t, 0, 4>), grid, dim3(256), 0, g_st, pcm, basis, basis_ri, Y1, FM2, n, map, fm_stride, 1);
         CK(hipDeviceSynchronize());
         CK(hipMemcpy(ya.data(), Y0, ya.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(yb.data(), Y1, yb.size() * 4, hipMemcpyDeviceToHost));
         CK(hipMemcpy(fa.data(), FM1, fa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(fb.data(), FM2, fb.size() * 4, hipMemcpyDeviceToHost));
         printf("   ri log mode against sym OPT3: Y %s, FM %s\n", memcmp(ya.data(), yb.data(), ya.size() * 4) ? "MISMATCH" : "bit-identical", memcmp(fa.data(), fb.data(), fa.size() * 4) ? "MISMATCH" : "bit-identical");
         CK(hipFree(FM2));
         CK(hipMemcpy(ref.data(), Y0, 0, hipMemcpyDeviceToHost));
      }
      if (getenv("FE_RI_REPEAT")) {                       // k_frontend_ri against ITSELF and against k_frontend_sym, launch after launch: a result that depends on timing shows here
         const int R = atoi(getenv("FE_RI_REPEAT"));
         std::vector<float> y0(ref.size()), y1(ref.size()), f0((size_t)kBinSplit * n * kFrames), f1(f0.size());
         hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 3>), grid, dim3(256), 0, g_st, pcm, basis, Y0, FM, n, map, fm_stride, 1);
         CK(hipDeviceSynchronize());
         CK(hipMemcpy(y0.data(), Y0, y0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f0.data(), FM, f0.size() * 4, hipMemcpyDeviceToHost));
         long bad_runs = 0;
         for (int r = 0; r < R; ++r) {
            CK(hipMemsetAsync(Y1, 0xff, y1.size() * 4, g_st));
            hipLaunchKernelGGL((k_frontend_ri<int16_t, 0, 4>), grid, dim3(256), 0, g_st, pcm, basis, basis_ri, Y1, FM1, n, map, fm_stride, 1);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(y1.data(), Y1, y1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f1.data(), FM1, f1.size() * 4, hipMemcpyDeviceToHost));
            size_t by = 0, bf = 0; long first = -1;
            for (size_t i = 0; i < y1.size(); ++i) if (memcmp(&y1[i], &y0[i], 4)) { ++by; if (first < 0) first = (long)i; }
            for (size_t i = 0; i < f1.size(); ++i) if (memcmp(&f1[i], &f0[i], 4)) ++bf;
            if (by || bf) { ++bad_runs; printf("   run %d: %zu words of Y, %zu of FM differ; first: chunk %ld bin %ld frame %ld\n", r, by, bf, first / (kBins * kFrames), (first / kFrames) % kBins, first % kFrames); }
         }
         printf("ri x %d at %d chunks: %ld runs with differences\n", R, n, bad_runs);
         return 0;
      }
      for (int rep = 0; rep < 2; ++rep) {
         time_it("ri, log mode", [&] { hipLaunchKernelGGL((k_frontend_ri<int16_t, 0, 4>), grid, dim3(256), 0, g_st, pcm, basis, basis_ri, Y1, FM1, n, map, fm_stride, 1); });
         time_it("sym nb2 OPT0, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 0>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
         time_it("sym nb2 OPT1, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 1>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
         time_it("sym nb2 OPT2, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 2>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
         time_it("sym nb2 OPT3, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 3>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); });
      }
      float *Y32 = nullptr;
      if (getenv("FE_POWER")) CK(hipMalloc(&Y32, (size_t)n * kBins * 32 * 4));
      if (getenv("FE_POWER")) {
         // 1 s of back-to-back launches per variant, 4 rounds in turn, with the shader clock and the socket power sampled every 5 ms from sysfs
         // (whichever card's pp_dpm_sclk / hwmon power1_* is readable and above idle): is the kernel's rate set by its instruction count or by the power cap?
         std::vector<std::string> sclk, pwr;
         for (int c = 0; c < 16; ++c) {
            char b[256];
            snprintf(b, sizeof b, "/sys/class/drm/card%d/device/pp_dpm_sclk", c); if (FILE *f = fopen(b, "r")) { fclose(f); sclk.push_back(b); }
            for (int h = 0; h < 12; ++h)
               for (const char *leaf : {"power1_average", "power1_input"}) {
                  snprintf(b, sizeof b, "/sys/class/drm/card%d/device/hwmon/hwmon%d/%s", c, h, leaf); if (FILE *f = fopen(b, "r")) { fclose(f); pwr.push_back(b); }
               }
         }
         printf("sysfs: %zu sclk files, %zu power files\n", sclk.size(), pwr.size());
         auto read_sclk = [&](const std::string &p) { double v = 0; if (FILE *f = fopen(p.c_str(), "r")) { char l[128]; while (fgets(l, sizeof l, f)) if (strchr(l, '*')) { const char *q = strchr(l, ':'); if (q) v = atof(q + 1); } fclose(f); } return v; };
         auto read_num = [&](const std::string &p) { double v = 0; if (FILE *f = fopen(p.c_str(), "r")) { if (fscanf(f, "%lf", &v) != 1) v = 0; fclose(f); } return v; };
         struct Var { const char *name; std::function<void()> launch; };
         std::vector<Var> vars = {
            {"OPT0", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 0>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); }},
            {"OPT3", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 3>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); }},
            {"RI", [&] { hipLaunchKernelGGL((k_frontend_ri<int16_t, 0, 4>), grid, dim3(256), 0, g_st, pcm, basis, basis_ri, Y1, FM1, n, map, fm_stride, 1); }},
            // rows of Y on 128-byte boundaries (pitch 32 floats instead of 25: + 28 % bytes, no row straddles a 128-byte line)
            // no Y stores at all (OPT bit 4; the partial bin sums keep every value alive): the ceiling of what a front end fused into the first stage could gain HERE
            {"NOY", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 7>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); }},
            // XCD-major block order (OPT bit 8): the two workgroups that share a chunk write its lines behind the same L2
            {"XCD", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 11>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride, 1); }},
            {"YP32", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4, 3, 32>), grid, dim3(256), 0, g_st, pcm, basis, Y32, FM1, n, map, fm_stride, 1); }},
         };
         if (getenv("FE_ONLY")) {                      // one variant, 200 launches: for a rocprofv3 --pmc pass (WRITE_SIZE per launch)
            for (auto &v : vars) if (!strcmp(v.name, getenv("FE_ONLY"))) { for (int r = 0; r < 200; ++r) v.launch(); CK(hipDeviceSynchronize()); printf("%s x 200\n", v.name); }
            return 0;
         }
         for (int round = 0; round < 4; ++round)
            for (auto &v : vars) {
               hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
               std::atomic<bool> stop{false};
               double s_clk = 0, s_pw = 0; int ns = 0; double max_pw = 0;
               std::thread th([&] {
                  while (!stop) {
                     double c = 0, pw = 0;
                     for (auto &p : sclk) c = fmax(c, read_sclk(p));
                     for (auto &p : pwr) pw = fmax(pw, read_num(p));
                     s_clk += c; s_pw += pw; ++ns; max_pw = fmax(max_pw, pw);
                     usleep(5000);
                  }
               });
               CK(hipEventRecord(a, g_st));
               const int L = 2000;
               for (int r = 0; r < L; ++r) v.launch();
               CK(hipEventRecord(b, g_st)); CK(hipEventSynchronize(b));
               stop = true; th.join();
               float ms; CK(hipEventElapsedTime(&ms, a, b));
               printf("round %d %-5s %8.4f ms/launch   sclk %6.0f MHz   power avg %6.1f W max %6.1f W  (%d samples)\n", round, v.name, ms / L, ns ? s_clk / ns : 0, ns ? s_pw / ns * 1e-6 : 0, max_pw * 1e-6, ns);
               fflush(stdout);
            }
         return 0;
      }
      if (getenv("FE_OPT_ONLY")) return 0;
      time_it("fl nb3 (all 129 bins), log mode", [&] { hipLaunchKernelGGL((k_frontend_fl<int16_t, 0, 3, 4, 0, 1>), grid, dim3(256), 0, g_st, pcm, basis, Y0, FM, n, map, fm_stride, nullptr); });
      time_it("sym nb3 w4, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 3, 4>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb2 w4, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb2 w5, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 5>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb3 w3, log mode", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 3, 3>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      // occupancy probes: unused dynamic LDS caps the workgroups per CU (31 KB static + pad): 3, 2 and 1 workgroups = 3, 2, 1 waves per SIMD
      time_it("sym nb2 w4, log mode, 3 waves/SIMD", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4>), grid, dim3(256), 16 << 10, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb2 w4, log mode, 2 waves/SIMD", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4>), grid, dim3(256), 40 << 10, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb3 w3, log mode, 2 waves/SIMD", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 3, 3>), grid, dim3(256), 40 << 10, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb3 w2, log mode, 2 waves/SIMD", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 3, 2>), grid, dim3(256), 40 << 10, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      time_it("sym nb2 w4, log mode, 1 wave/SIMD", [&] { hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 2, 4>), grid, dim3(256), 60 << 10, g_st, pcm, basis, Y1, FM1, n, map, fm_stride); });
      {  // log values: sym takes v_sqrt_f32 under the logarithm (<= 1 ulp of the magnitude): report the largest difference against fl
         std::vector<float> a(ref.size()), b(ref.size());
         CK(hipMemcpy(a.data(), Y0, a.size() * 4, hipMemcpyDeviceToHost));
         hipLaunchKernelGGL((k_frontend_sym<int16_t, 0, 3, 4>), grid, dim3(256), 0, g_st, pcm, basis, Y1, FM1, n, map, fm_stride);
         CK(hipDeviceSynchronize());
         CK(hipMemcpy(b.data(), Y1, b.size() * 4, hipMemcpyDeviceToHost));
         double mx = 0; for (size_t i = 0; i < a.size(); ++i) mx = fmax(mx, fabs((double)a[i] - b[i]));
         printf("   max |Y_sym - Y_fl| (log mode) = %.3e\n", mx);
      }
      return 0;
   }
   if (argc > 2 && !strcmp(argv[2], "fl")) {          // frame-lane kernel against the shipped k_frontend: magnitudes, log values and FM
      float *FM1;
      CK(hipMalloc(&FM1, (size_t)kBinSplit * n * kFrames * 4));
      std::vector<float> fref((size_t)kBinSplit * n * kFrames), fgot(fref.size());
      auto check_fm = [&](const char *name) {
         CK(hipMemcpy(fgot.data(), FM1, fgot.size() * 4, hipMemcpyDeviceToHost));
         size_t bad = 0;
         for (size_t i = 0; i < fgot.size(); ++i) if (memcmp(&fgot[i], &fref[i], 4)) ++bad;
         printf("   %-41s FM %s (%zu mismatching words)\n", name, bad ? "MISMATCH" : "bit-identical to baseline", bad);
      };
      run<1, 256, 4, 0, 0, 0, 0, 2>("k_frontend PK2 w4 (shipped), magnitude", pcm, basis, Y0, FM, n, reps);
      CK(hipMemcpy(ref.data(), Y0, ref.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(Y1, 0, ref.size() * 4));
      run_fl<1, 3, 4>("fl nb3 w4, magnitude", pcm, basis, Y1, FM1, n, reps);          check("fl nb3 w4", Y1);
      run_fl<1, 3, 3>("fl nb3 w3, magnitude", pcm, basis, Y1, FM1, n, reps);          check("fl nb3 w3", Y1);
      run_fl<1, 3, 2>("fl nb3 w2, magnitude", pcm, basis, Y1, FM1, n, reps);          check("fl nb3 w2", Y1);
      run_fl<1, 1, 4>("fl nb1 w4, magnitude", pcm, basis, Y1, FM1, n, reps);          check("fl nb1 w4", Y1);
      run_fl<1, 1, 5>("fl nb1 w5, magnitude", pcm, basis, Y1, FM1, n, reps);          check("fl nb1 w5", Y1);
      run<0, 256, 4, 0, 0, 0, 0, 2>("k_frontend PK2 w4 (shipped), log mode", pcm, basis, Y0, FM, n, reps);
      CK(hipMemcpy(ref.data(), Y0, ref.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(fref.data(), FM, fref.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(Y1, 0, ref.size() * 4));
      run_fl<0, 3, 4>("fl nb3 w4, log mode", pcm, basis, Y1, FM1, n, reps);           check("fl nb3 w4 log", Y1); check_fm("fl nb3 w4 log");
      run_fl<0, 3, 3>("fl nb3 w3, log mode", pcm, basis, Y1, FM1, n, reps);           check("fl nb3 w3 log", Y1); check_fm("fl nb3 w3 log");
      run_fl<0, 1, 4>("fl nb1 w4, log mode", pcm, basis, Y1, FM1, n, reps);           check("fl nb1 w4 log", Y1); check_fm("fl nb1 w4 log");
      run_fl<0, 3, 4, 1>("fl ABL1: no tap loads/waits", pcm, basis, Y1, FM1, n, reps);
      run_fl<0, 3, 4, 2>("fl ABL2: no sample loads", pcm, basis, Y1, FM1, n, reps);
      run_fl<0, 3, 4, 3>("fl ABL3: neither", pcm, basis, Y1, FM1, n, reps);
      run_fl<0, 3, 4, 8>("fl ABL8: same 6 KB of taps for every batch", pcm, basis, Y1, FM1, n, reps);
      run_fl<0, 3, 4, 10>("fl ABL10: same taps, no sample loads", pcm, basis, Y1, FM1, n, reps);
      run_fl<0, 3, 4, 16>("fl ABL16: the 4 waves of a WG read the same taps", pcm, basis, Y1, FM1, n, reps);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      CK(hipMalloc(&g_counter, 4)); CK(hipMemset(g_counter, 0, 4));
      g_slots = 4 * (256 - g_removed);
      CK(hipMemset(Y1, 0, ref.size() * 4));
      run_fl<0, 3, 4, 0, 1, true>("fl persistent, work counter", pcm, basis, Y1, FM1, n, reps);   check("fl persistent", Y1); check_fm("fl persistent");
      g_slots = 5 * (256 - g_removed);
      run_fl<0, 3, 4, 0, 1, true>("fl persistent, 5 per CU", pcm, basis, Y1, FM1, n, reps);
      g_slots = 4 * (256 - g_removed);
      run_fl<0, 3, 4, 0, 2>("fl nps2 (512 threads)", pcm, basis, Y1, FM1, n, reps);   check("fl nps2", Y1); check_fm("fl nps2");
      CK(hipMemset(Y1, 0, ref.size() * 4));
      run_fl<0, 3, 4, 0, 4>("fl nps4 (1024 threads)", pcm, basis, Y1, FM1, n, reps);  check("fl nps4", Y1); check_fm("fl nps4");
      run_fl<0, 3, 4, 0>("fl +23 KB LDS: 3 WG/CU", pcm, basis, Y1, FM1, n, reps, 23 * 1024);
      run_fl<0, 3, 4, 0>("fl +50 KB LDS: 2 WG/CU", pcm, basis, Y1, FM1, n, reps, 50 * 1024);
      run_fl<0, 3, 4, 16>("fl ABL16 +23 KB LDS: 3 WG/CU", pcm, basis, Y1, FM1, n, reps, 23 * 1024);
      run_fl<0, 3, 4, 16>("fl ABL16 +50 KB LDS: 2 WG/CU", pcm, basis, Y1, FM1, n, reps, 50 * 1024);
      run_fl<0, 3, 4, 8>("fl ABL8 +50 KB LDS: 2 WG/CU", pcm, basis, Y1, FM1, n, reps, 50 * 1024);
      return 0;
   }
   //            MODE NT  MINW SHIFT LOCK STAG
   run<1, 256, 3, 0, 0, 0>("baseline nt256 w3 bperm", pcm, basis, Y0, FM, n, reps);
   CK(hipMemcpy(ref.data(), Y0, ref.size() * 4, hipMemcpyDeviceToHost));
   run<1, 256, 3, 1, 0, 0>("dpp", pcm, basis, Y1, FM, n, reps);                     check("dpp", Y1);
   run<1, 256, 3, 0, 0, 1>("ABL1: v_mul by a VGPR (no SGPR operand)", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 3, 0, 0, 2>("ABL2: products only (no tree adds)", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 3, 0, 0, 3>("ABL3: tree adds only (no products)", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 3, 0, 0, 0, 0, 1>("PK w3 (v_pk_mul products)", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 4, 0, 0, 0, 0, 1>("PK w4", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 2, 0, 0, 0, 0, 1>("PK w2", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 4, 0, 0, 0, 0, 2>("PK2 w4 (adds packed too)", pcm, basis, Y1, FM, n, reps);     check("PK2 w4", Y1);
   run<1, 256, 3, 0, 0, 0, 0, 2>("PK2 w3", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 5, 0, 0, 0, 0, 2>("PK2 w5", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 4, 1, 0, 0, 0, 2>("PK2 w4 dpp", pcm, basis, Y1, FM, n, reps);                   check("PK2 w4 dpp", Y1);
   run<0, 256, 4, 0, 0, 0, 0, 2>("PK2 w4 (log mode)", pcm, basis, Y1, FM, n, reps);
   run<0, 256, 4, 1, 0, 0, 0, 2>("PK2 w4 dpp (log mode)", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 3, 1, 0, 0, 0, 1>("PK w3 dpp", pcm, basis, Y1, FM, n, reps);
   run<0, 256, 3, 0, 0, 0, 0, 1>("PK w3 (log mode)", pcm, basis, Y1, FM, n, reps);
   run<0, 256, 4, 0, 0, 0, 0, 1>("PK w4 (log mode)", pcm, basis, Y1, FM, n, reps);
   run<0, 256, 3, 0, 0, 0, 0, 0>("baseline w3 (log mode)", pcm, basis, Y1, FM, n, reps);
   run<0, 256, 4, 0, 0, 0, 0, 0>("baseline w4 (log mode)", pcm, basis, Y1, FM, n, reps);
   run<0, 256, 5, 0, 0, 0, 0, 1>("PK w5 (log mode)", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 4, 0, 0, 0>("baseline w4", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 4, 0, 0, 1>("ABL1 w4", pcm, basis, Y1, FM, n, reps);
   run<1, 256, 4, 0, 0, 3>("ABL3 w4", pcm, basis, Y1, FM, n, reps);
   return 0;
}
