// yread_probe.hip -- how fast can 16,384 x [129][25] floats (211 MB, the front end -> first encoder stage hand-off) be READ at all?
//   hipcc --offload-arch=gfx950 -O3 tools/yread_probe.hip -o tools/yread_probe
// MODE 0: the first stage's pattern (workgroup = 2 chunks, lane = (chunk, frame) column, 4 waves split the channels, dword loads)
// MODE 1: flat coalesced float4 loads over the same bytes (the memory system's own ceiling for this size)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *__restrict__ Y, float *__restrict__ out, int n)
{
   float acc = 0;
   if (MODE == 0) {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int cb = lane / 25, t = lane - cb * 25;
      const int item = blockIdx.x * 2 + cb;
      const bool ok = lane < 50 && item < n;
      const float *x = Y + (size_t)(ok ? item : 0) * 129 * 25 + t;
      const int ch0 = wave * 33, ch1 = min(ch0 + 33, 129);
#pragma unroll 8
      for (int ch = ch0; ch < ch1; ++ch) acc += x[ch * 25];
   } else {
      const float4 *p = reinterpret_cast<const float4 *>(Y) + (size_t)blockIdx.x * (2 * 129 * 25 / 4) ;
      for (int i = threadIdx.x; i < 2 * 129 * 25 / 4; i += 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
   }
   if (acc == 12345.678f) out[0] = acc;
}
int main()
{
   const int n = 16384;
   float *Y, *out, *scratch;
   CK(hipMalloc(&Y, (size_t)n * 129 * 25 * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&scratch, (size_t)512 << 20));
   CK(hipMemset(Y, 0, (size_t)n * 129 * 25 * 4));
   hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   for (int mode = 0; mode < 2; ++mode)
      for (int cold = 0; cold < 2; ++cold) {
         float tot = 0;
         for (int r = 0; r < 6; ++r) {
            if (cold) CK(hipMemsetAsync(scratch, r, (size_t)512 << 20, 0));   // evicts L2 and the 256 MB infinity cache
            CK(hipEventRecord(a, 0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n / 2), dim3(256), 0, 0, Y, out, n);
            else           hipLaunchKernelGGL(k<1>, dim3(n / 2), dim3(256), 0, 0, Y, out, n);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float t; CK(hipEventElapsedTime(&t, a, b));
            if (r) tot += t;
         }
         printf("mode %d (%s) %s: %.4f ms  %.2f TB/s\n", mode, mode ? "flat float4" : "first-stage pattern", cold ? "cold" : "warm", tot / 5, 211.3e6 / (tot / 5 * 1e-3) / 1e12);
      }
   return 0;
}
