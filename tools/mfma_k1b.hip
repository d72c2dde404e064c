// mfma_k1b.hip -- throughput probe: K=1 MFMA products + VALU tree adds, structured like the planned kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// per iteration: 8 taps -> 8 MFMAs (C = 0) -> per output the 7-add tree + 1 accumulate = 8 adds x 16 outputs = 128 v_add
template <int MINW>
__global__ __launch_bounds__(256, MINW) void k(float *out, const float *xa, const float *kb, int iters)
{
   const int l = threadIdx.x & 63;
   const float *xp = xa + l, *kp = kb + (l & 15);
   f16v acc = {0};
   const f16v z = {0};
   for (int it = 0; it < iters; ++it) {
      f16v p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = __builtin_amdgcn_mfma_f32_16x16x1f32(xp[(it * 8 + j) * 64 % 4096], kp[(it * 8 + j) * 16 % 4096], z, 0, 0, 0);
      const f16v p01 = p[0] + p[1], p23 = p[2] + p[3], p45 = p[4] + p[5], p67 = p[6] + p[7];
      const f16v q0 = p01 + p23, q1 = p45 + p67;
      acc += q0 + q1;
   }
   float r = 0;
   for (int i = 0; i < 16; ++i) r += acc[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MINW> int bench(float *out, const float *xa, const float *kb, int wps)
{
   const int iters = 4000;
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   hipLaunchKernelGGL(k<MINW>, dim3(256 * wps), dim3(256), 0, 0, out, xa, kb, 50);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(e0, 0));
   hipLaunchKernelGGL(k<MINW>, dim3(256 * wps), dim3(256), 0, 0, out, xa, kb, iters);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1));
   printf("8 MFMA(K=1) + 128 v_add per iter, launch_bounds min %d, %d waves/SIMD: %.1f cycles per tap (MFMA) per SIMD @2.4GHz\n", MINW, wps,
          ms * 1e6 / ((double)iters * 8 * wps) * 2.4);
   return 0;
}

int main()
{
   float *out, *xa, *kb;
   CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&xa, 8192 * 4)); CK(hipMalloc(&kb, 8192 * 4));
   CK(hipMemset(xa, 0, 8192 * 4)); CK(hipMemset(kb, 0, 8192 * 4));
   for (int wps : {1, 2, 3, 4}) { bench<2>(out, xa, kb, wps); }
   for (int wps : {2, 4}) { bench<4>(out, xa, kb, wps); }
   return 0;
}
