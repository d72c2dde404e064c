// Which XCD does workgroup i of a launch run on -- is it i % 8 for EVERY launch on a CU-masked stream, whatever was launched there before?
// (k_lstm_layer's TRAIL form needs workgroup i of two launches on two masked streams to share an XCD.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_map_probe tools/xcd_map_probe.hip && /tmp/xcd_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k_where(unsigned *out, int spin)
{
   unsigned xcc;
   asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
   const unsigned long long t0 = __builtin_readcyclecounter();
   while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
   if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
}
int main()
{
   unsigned *d; (void)hipMalloc(&d, 4096 * 4);
   std::vector<unsigned> h(4096);
   hipStream_t sb, sc;
   uint32_t mb[8] = {0x0000ffffu, 0, 0, 0, 0, 0, 0, 0}, mc[8] = {0xffff0000u, 0, 0, 0, 0, 0, 0, 0};
   (void)hipExtStreamCreateWithCUMask(&sb, 8, mb);
   (void)hipExtStreamCreateWithCUMask(&sc, 8, mc);
   const int grids[] = {16, 16, 7, 16, 3, 16, 20, 16, 16, 5, 5, 16};
   for (int which = 0; which < 2; ++which) {
      hipStream_t st = which ? sc : sb;
      printf("stream %c (mask bits %s):\n", which ? 'C' : 'B', which ? "16..31" : "0..15");
      for (int g : grids) {
         hipLaunchKernelGGL(k_where, dim3(g), dim3(512), 0, st, d, 20000);
         (void)hipStreamSynchronize(st);
         (void)hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
         printf("  grid %2d:", g);
         for (int i = 0; i < g; ++i) printf(" %u", h[i]);
         printf("\n");
      }
   }
   return 0;
}
