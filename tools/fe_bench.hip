// fe_bench.hip -- device micro-benchmark for k_frontend tuning knobs (not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/fe_bench.hip -o tools/fe_bench
//   tools/fe_bench [n_chunks]
// Times every variant with HIP events on random s16 input and checks that all variants produce the same
// bits (magnitudes in MODE 1; MODE 0 outputs except the rotated-order mean of STAGGER variants).
#include "../vadc_amd/csrc/kernels_frontend.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

template <int MODE, int NB, int MINW>
static float run_fl(const char *name, const int16_t *pcm, const float *basis, float *Y, float *FM, int n, int reps)
{
   const dim3 blocks((unsigned)(((long)n * kFrames + 63) / 64));
   const size_t fm_stride = (size_t)n * kFrames;
   const ItemMap map{n, 0, n};
   hipEvent_t a, b;
   CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   hipLaunchKernelGGL((k_frontend_fl<int16_t, MODE, NB, MINW>), blocks, dim3(256), 0, 0, pcm, basis, Y, FM, n, map, fm_stride);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(a, 0));
   for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL((k_frontend_fl<int16_t, MODE, NB, MINW>), blocks, dim3(256), 0, 0, pcm, basis, Y, FM, n, map, fm_stride);
   CK(hipEventRecord(b, 0));
   CK(hipEventSynchronize(b));
   float ms = 0;
   CK(hipEventElapsedTime(&ms, a, b));
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
