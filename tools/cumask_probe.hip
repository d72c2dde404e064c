// Which (XCD, shader engine, CU) a bit of hipExtStreamCreateWithCUMask selects, and how a persistent one-workgroup-per-CU grid lands on a masked stream.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask_probe tools/cumask_probe.hip && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void k_where(unsigned *out)
{
   unsigned xcc, hw;
   asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
   asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
   if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
// a workgroup that fills a CU's LDS (so that only one fits) and stays for a while
__global__ void k_hold(unsigned *out, int spin)
{
   __shared__ char big[140 * 1024];
   unsigned xcc, hw;
   asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
   asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
   big[threadIdx.x] = (char)xcc;
   unsigned long long t0 = __builtin_readcyclecounter();
   while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
   if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw + big[7] * 0; }
}
int main()
{
   int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
   printf("CUs %d\n", ncu);
   unsigned *d; hipMalloc(&d, 4096 * 8); std::vector<unsigned> h(4096 * 2);
   const int words = (ncu + 31) / 32;
   for (int bit : {0, 1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 33, 47, 48, 63, 64, 127, 128, 255}) {
      if (bit >= ncu) continue;
      std::vector<uint32_t> m(words, 0u); m[bit / 32] |= 1u << (bit % 32);
      hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, words, m.data()) != hipSuccess) { printf("bit %d: create failed\n", bit); continue; }
      hipLaunchKernelGGL(k_where, dim3(4), dim3(64), 0, s, d); hipStreamSynchronize(s);
      hipMemcpy(h.data(), d, 32, hipMemcpyDeviceToHost);
      printf("bit %3d -> xcc %u  hw_id 0x%08x (cu %u sh %u se %u)\n", bit, h[0] & 0xf, h[1], (h[1] >> 8) & 0xf, (h[1] >> 12) & 1, (h[1] >> 13) & 0x7);
      hipStreamDestroy(s);
   }
   // persistent grid on a stream that excludes the first `ex` mask bits: workgroups per XCD and the time of the launch
   for (int ex : {0, 32, 48, 64}) {
      std::vector<uint32_t> m(words, 0u);
      for (int cu = ex; cu < ncu; ++cu) m[cu / 32] |= 1u << (cu % 32);
      hipStream_t s; hipExtStreamCreateWithCUMask(&s, words, m.data());
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      const int grid = ncu - ex;
      hipLaunchKernelGGL(k_hold, dim3(grid), dim3(256), 0, s, d, 200000); hipStreamSynchronize(s);
      hipEventRecord(a, s); hipLaunchKernelGGL(k_hold, dim3(grid), dim3(256), 0, s, d, 200000); hipEventRecord(b, s); hipStreamSynchronize(s);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
      std::map<unsigned, int> per;
      for (int i = 0; i < grid; ++i) per[h[2 * i]]++;
      printf("first %2d bits excluded, grid %3d: %.3f ms; workgroups per XCD:", ex, grid, ms);
      for (auto &kv : per) printf(" %u:%d", kv.first, kv.second);
      printf("\n");
      hipStreamDestroy(s);
   }
   // the same with the excluded CUs spread over the mask: every 8th / every 16th bit ...
   for (int ex : {32, 48}) {
      std::vector<uint32_t> m(words, 0xffffffffu);
      int left = ex;
      for (int r = 0; left > 0 && r < 32; ++r)                 // bit = 32 * x + r for x = 0..7: r-th CU of every 32-bit word
         for (int x = 0; x < words && left > 0; ++x) { m[x] &= ~(1u << r); --left; }
      hipStream_t s; hipExtStreamCreateWithCUMask(&s, words, m.data());
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      const int grid = ncu - ex;
      hipLaunchKernelGGL(k_hold, dim3(grid), dim3(256), 0, s, d, 200000); hipStreamSynchronize(s);
      hipEventRecord(a, s); hipLaunchKernelGGL(k_hold, dim3(grid), dim3(256), 0, s, d, 200000); hipEventRecord(b, s); hipStreamSynchronize(s);
      float ms = 0; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
      std::map<unsigned, int> per;
      for (int i = 0; i < grid; ++i) per[h[2 * i]]++;
      printf("%2d bits excluded, spread over the words, grid %3d: %.3f ms; workgroups per XCD:", ex, grid, ms);
      for (auto &kv : per) printf(" %u:%d", kv.first, kv.second);
      printf("\n");
      hipStreamDestroy(s);
   }
   return 0;
}
