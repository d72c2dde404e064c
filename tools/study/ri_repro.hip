// tools/study/ri_repro.hip -- NOT part of the product.  The reproducer behind DESIGN.md 4.1 (d) and tools/check_pk_opsel.py: k_frontend_ri (the product's kernel)
// launched again and again on one stream while the layer kernels of the recurrence (k_lstm_layer, the product's) run on two other streams over dummy data; every
// launch's magnitudes are compared ON THE DEVICE with k_frontend_sym's (computed alone, before).
//   victim "ri"      the shipped form: the half-swapped pair is the FIRST source of the two v_pk_add_f32 per l-pair            0 of 12,000 launches differ
//   victim "ri_src1" the natural form: `v_pk_add_f32 d, P, Q op_sel:[0,1] op_sel_hi:[1,0]`, low result = P.lo + Q.HI         164 of 6,000: always 16 words, one
//                    row 64 -+ b, lanes 48 .. 63 of one workgroup, the low result = P.lo alone (the values occur nowhere else in the output)
//   victim "sym"     k_frontend_sym                                                                                             0 of 6,000
//   neighbour "none" any victim alone on the chip                                                                               0 of 3,000
// Timing-only ablations of the NEIGHBOUR through the product's macros (no MFMAs / no transcendentals / no LDS-DMA: 2,444 / 60 / 513 of 6,000) say that no instruction
// class of it is needed -- only that its waves share the SIMDs:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize [-DVADC_LSTM_ABL_NOMFMA | -DVADC_LSTM_ABL_NOGATES | -DVADC_LSTM_ABL_NOXLOAD] tools/study/ri_repro.hip -o tools/study/ri_repro
//   tools/study/ri_repro [launches = 6000] [victim: ri | ri_src1 | sym] [neighbour: lstm | none]
#include "../../vadc_amd/csrc/kernels_frontend.hip"
#include "../../vadc_amd/csrc/kernels_lstm.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>
#include <algorithm>
using namespace vadc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void k_count_diff(const unsigned *a, const unsigned *b, size_t n, unsigned *count, unsigned *first)
{
   for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
      if (a[i] != b[i]) { if (atomicAdd(count, 1u) == 0) *first = (unsigned)i; }
}
int main(int argc, char **argv)
{
   const int R = argc > 1 ? atoi(argv[1]) : 6000;
   const char *victim = argc > 2 ? argv[2] : "ri";
   const bool victim_ri = !strcmp(victim, "ri"), victim_src1 = !strcmp(victim, "ri_src1");
   const bool neighbour = !(argc > 3 && !strcmp(argv[3], "none"));
   const int n = 2048;                                      // chunks per front-end launch: 800 workgroups
   // basis with the DFT symmetries by construction (tools/fe_bench.hip), speech-like input
   double qc[65];
   for (int k = 0; k <= 64; ++k) qc[k] = cos(2.0 * M_PI * k / 256.0);
   qc[64] = 0.0;
   auto cosi = [&](int m) { m &= 255; if (m > 128) m = 256 - m; return m <= 64 ? qc[m] : -qc[128 - m]; };
   auto sini = [&](int m) { return cosi(m - 64); };
   std::vector<float> nat((size_t)kFilters * 256), h_basis((size_t)kFilters * kFilterLen + 1024, 0.0f), h_ri((size_t)35 * 512 + 64, 0.0f);
   for (int f = 0; f < kFilters; ++f)
      for (int t = 0; t < 256; ++t) { const int bin = f % kBins; const double w = 0.5 - 0.5 * cosi(t); nat[(size_t)f * 256 + t] = (float)(f < kBins ? w * cosi(bin * t) : -w * sini(bin * t)); }
   for (int f = 0; f < kFilters; ++f) for (int ii = 0; ii < 4; ++ii) for (int lp = 0; lp < 4; ++lp) for (int j = 0; j < 8; ++j) for (int b = 0; b < 2; ++b)
      h_basis[(size_t)f * 256 + ii * 64 + lp * 16 + j * 2 + b] = nat[(size_t)f * 256 + 64 * (3 - ii) + 8 * j + (2 * lp + b)];
   for (int f = 0; f < 33; ++f) for (int ii = 0; ii < 4; ++ii) for (int lp = 0; lp < 4; ++lp) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) for (int c = 0; c < 2; ++c)
      h_ri[(size_t)f * 512 + ii * 128 + lp * 32 + h * 16 + j * 2 + c] = nat[(size_t)(c ? kBins + f : f) * 256 + 64 * (3 - ii) + 8 * j + (2 * lp + h)];
   std::vector<int16_t> h_pcm((size_t)n * kChunk);
   srand(7);
   for (size_t i = 0; i < h_pcm.size(); ++i) { const double amp = ((i / 8000) % 3 == 0) ? 3.0 : 2500.0; h_pcm[i] = (int16_t)(amp * sin(0.02 * (double)(i % 16000)) + amp * 0.3 * ((rand() % 2001) - 1000) / 1000.0); }
   int16_t *pcm; float *basis, *basis_ri, *Y0, *Y1, *FM;
   CK(hipMalloc(&pcm, h_pcm.size() * 2)); CK(hipMalloc(&basis, h_basis.size() * 4)); CK(hipMalloc(&basis_ri, h_ri.size() * 4));
   const size_t ny = (size_t)n * kBins * kFrames;
   CK(hipMalloc(&Y0, ny * 4)); CK(hipMalloc(&Y1, ny * 4)); CK(hipMalloc(&FM, (size_t)kBinSplit * n * kFrames * 4));
   CK(hipMemcpy(pcm, h_pcm.data(), h_pcm.size() * 2, hipMemcpyHostToDevice));
   CK(hipMemcpy(basis, h_basis.data(), h_basis.size() * 4, hipMemcpyHostToDevice));
   CK(hipMemcpy(basis_ri, h_ri.data(), h_ri.size() * 4, hipMemcpyHostToDevice));
   const size_t fm_stride = (size_t)n * kFrames;
   const ItemMap map{n, 0, n};
   const dim3 grid((unsigned)(((long)n * kFrames + 63) / 64));
   hipStream_t sv, s0, s1;
   CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
   hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 3>), grid, dim3(256), 0, sv, pcm, basis, Y0, FM, n, map, fm_stride, 1);
   CK(hipStreamSynchronize(sv));
   // the neighbour: both layer kernels over 10,240 streams x 1 chunk (640 tiles of 16 streams), dummy tiles and small weights
   const int S = 10240, tiles = S / 16;
   _Float16 *x, *h0; float *w, *b, *dw, *db, *hs, *cs, *probs;
   const size_t tile_halves = (size_t)tiles * 1 * 7 * 2 * 16 * 64;
   CK(hipMalloc(&x, tile_halves * 2)); CK(hipMalloc(&h0, tile_halves * 2));
   { std::vector<_Float16> hx(tile_halves); for (auto &v : hx) v = (_Float16)(((rand() % 2001) - 1000) / 4000.0f); CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(h0, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); }
   { std::vector<float> hw((size_t)2 * 256 * 128 + 2 * 256 + 128 + 2); for (auto &v : hw) v = ((rand() % 2001) - 1000) / 8000.0f;
     CK(hipMalloc(&w, hw.size() * 4)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); b = w + 2 * 256 * 128; dw = b + 2 * 256; db = dw + 128; }
   CK(hipMalloc(&hs, (size_t)S * 128 * 4)); CK(hipMalloc(&cs, (size_t)S * 128 * 4)); CK(hipMalloc(&probs, (size_t)S * 2 * 4));
   CK(hipMemset(hs, 0, (size_t)S * 128 * 4)); CK(hipMemset(cs, 0, (size_t)S * 128 * 4));
   LstmWeights lw; lw.w = w; lw.wT = w; lw.b = b; lw.dec_w = dw; lw.dec_b = db;
   std::atomic<bool> stop{false};
   std::thread th([&] {
      if (!neighbour) return;
      (void)hipSetDevice(0);
      while (!stop) {
         for (int k = 0; k < 8; ++k) {
            launch_lstm_layer(0, reinterpret_cast<const float *>(x), reinterpret_cast<float *>(h0), lw, hs, cs, probs, S, 1, 0, 1, s0, 0, 7, nullptr, 0, nullptr, 0);
            launch_lstm_layer(1, reinterpret_cast<const float *>(x), reinterpret_cast<float *>(h0), lw, hs, cs, probs, S, 1, 0, 1, s1, 0, 7, nullptr, 0, nullptr, 0);
         }
         (void)hipStreamSynchronize(s0); (void)hipStreamSynchronize(s1);
      }
   });
   unsigned *cnt; CK(hipHostMalloc(&cnt, 8));
   int bad = 0;
   for (int r = 0; r < R; ++r) {
      cnt[0] = 0; cnt[1] = 0;
      if (victim_ri)        hipLaunchKernelGGL((k_frontend_ri<int16_t, 1, 4, false>), grid, dim3(256), 0, sv, pcm, basis, basis_ri, Y1, FM, n, map, fm_stride, 1);
      else if (victim_src1) hipLaunchKernelGGL((k_frontend_ri<int16_t, 1, 4, true>), grid, dim3(256), 0, sv, pcm, basis, basis_ri, Y1, FM, n, map, fm_stride, 1);
      else                  hipLaunchKernelGGL((k_frontend_sym<int16_t, 1, 2, 4, 3>), grid, dim3(256), 0, sv, pcm, basis, Y1, FM, n, map, fm_stride, 1);
      hipLaunchKernelGGL(k_count_diff, dim3(256), dim3(256), 0, sv, (const unsigned *)Y0, (const unsigned *)Y1, ny, cnt, cnt + 1);
      CK(hipStreamSynchronize(sv));
      if (cnt[0] && bad < 4) {                              // where do the wrong values come from?  search k_frontend_sym's output for the same bits near by
         static std::vector<unsigned> a, bb;
         a.resize(ny); bb.resize(ny);
         CK(hipMemcpy(a.data(), Y0, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bb.data(), Y1, ny * 4, hipMemcpyDeviceToHost));
         int shown = 0;
         for (size_t i = 0; i < ny && shown < 16; ++i) if (a[i] != bb[i]) {
            const long chunk = i / (kBins * kFrames), bin = (i / kFrames) % kBins, fr = i % kFrames, pos = chunk * kFrames + fr;
            float got, want; memcpy(&got, &bb[i], 4); memcpy(&want, &a[i], 4);
            printf("      chunk %ld bin %ld frame %ld (lane %ld): want %a got %a;", chunk, bin, fr, pos % 64, want, got);
            // the same bits anywhere within +-3 chunks?
            int found = 0;
            for (long c2 = std::max(0l, chunk - 3); c2 <= std::min((long)n - 1, chunk + 3) && found < 3; ++c2)
               for (long e2 = 0; e2 < kBins * kFrames && found < 3; ++e2)
                  if (a[(size_t)c2 * kBins * kFrames + e2] == bb[i]) { printf(" = sym's (chunk %ld bin %ld frame %ld)", c2, e2 / kFrames, e2 % kFrames); ++found; }
            printf("%s\n", found ? "" : " (nowhere near)");
            ++shown;
         }
      }
      if (cnt[0]) { ++bad; if (bad <= 6) { const size_t i = cnt[1]; printf("   launch %d: %u words differ; first: chunk %zu bin %zu frame %zu (position %zu -> lane %zu)\n", r, cnt[0], i / (kBins * kFrames), (i / kFrames) % kBins, i % kFrames, (i / (kBins * kFrames)) * kFrames + i % kFrames, ((i / (kBins * kFrames)) * kFrames + i % kFrames) % 64); } }
   }
   stop = true; th.join();
   printf("%s beside %s: %d of %d launches differ from k_frontend_sym alone\n", victim_ri ? "k_frontend_ri (swapped pair first)" : victim_src1 ? "k_frontend_ri<SRC1> (swapped pair second)" : "k_frontend_sym", neighbour ? "k_lstm_layer (both layers, 640 tiles)" : "nothing", bad, R);
   return 0;
}
