// clock_probe.hip -- what clock do the CUs actually run at under load?  Compares s_memtime (shader-clock counter) with
// wall_clock64() (constant 100 MHz) around a busy loop of dependent fp32 MFMAs / VALU ops on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k(float *out, long long *t, int iters, int mode)
{
   f4v acc = {0, 0, 0, 0};
   float a = threadIdx.x * 0.001f, b = 1.0001f, v = a;
   const long long w0 = wall_clock64(), c0 = clock64();
   for (int i = 0; i < iters; ++i) {
      if (mode == 0) {
#pragma unroll
         for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      } else {
#pragma unroll
         for (int j = 0; j < 64; ++j) v = v * b + a;
      }
   }
   const long long w1 = wall_clock64(), c1 = clock64();
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + v;
   if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}
int main()
{
   float *out; long long *t, h[2];
   hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&t, 16);
   for (int mode = 0; mode < 2; ++mode)
      for (int waves = 1; waves <= 4; waves *= 2) {
         const int iters = 20000;
         hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves), 0, 0, out, t, iters, mode);
         hipDeviceSynchronize();
         hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
         const double sec = h[0] / 100e6;
         const double n = (double)iters * (mode == 0 ? 16 : 64);
         printf("mode %s waves/SIMD %d: wall %.3f ms, clock64 delta %lld (%.1f MHz if shader clock), %.1f ns per op per wave\n",
                mode == 0 ? "mfma16x16x4 dependent chain" : "v_fma dependent chain", waves, sec * 1e3, h[1], h[1] / sec / 1e6, sec * 1e9 / n);
      }
   return 0;
}
