// mfma_k1.hip -- probe for an MFMA-based exact-product STFT: v_mfma_f32_16x16x1_4b_f32 with C = 0 gives
// individually rounded products a_i * b_j (outer product, K = 1).  (1) layout dump, (2) bit-exactness vs v_mul_f32,
// (3) throughput of [1 MFMA + 16 dependent-free v_add] per iteration at 1/2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_layout(const float *a, const float *b, float *d)
{
   const int l = threadIdx.x;
   f16v c = {0};
   c = __builtin_amdgcn_mfma_f32_16x16x1f32(a[l], b[l], c, 0, 0, 0);
   for (int r = 0; r < 16; ++r) d[l * 16 + r] = c[r];
}

template <int NADD>
__global__ __launch_bounds__(256) void k_rate(float *out, const float *in, int iters)
{
   const int l = threadIdx.x & 63;
   float a = in[l], b = in[64 + l];
   f16v acc = {0}, p = {0};
   float s[16];
   for (int i = 0; i < 16; ++i) s[i] = 0.0f;
   for (int it = 0; it < iters; ++it) {
      f16v z = {0};
      p = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, z, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NADD; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[i & 15]) : "v"(acc[i & 15]));
      acc = p;
      a += 1.0f;
   }
   float r = 0;
   for (int i = 0; i < 16; ++i) r += s[i] + acc[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NADD> int bench(float *out, const float *in, int wps)
{
   const int iters = 20000;
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   hipLaunchKernelGGL(k_rate<NADD>, dim3(256 * wps), dim3(256), 0, 0, out, in, 100);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(e0, 0));
   hipLaunchKernelGGL(k_rate<NADD>, dim3(256 * wps), dim3(256), 0, 0, out, in, iters);
   CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
   float ms; CK(hipEventElapsedTime(&ms, e0, e1));
   printf("1 mfma_16x16x1 + %2d v_add per iter, %d waves/SIMD: %.2f cycles per iter per SIMD @2.4GHz\n", NADD, wps,
          ms * 1e6 / ((double)iters * wps) * 2.4);
   return 0;
}

int main()
{
   float ha[64], hb[64], hd[1024];
   for (int i = 0; i < 64; ++i) { ha[i] = 1000.0f + i; hb[i] = 1.0f + i * 0.001f; }
   float *a, *b, *d, *out;
   CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 4096)); CK(hipMalloc(&out, 256 * 8 * 256 * 4));
   CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
   hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, a, b, d);
   CK(hipMemcpy(hd, d, 4096, hipMemcpyDeviceToHost));
   // hypothesis: lane l, reg v: block = v / 4, row i = 4 * (l >> 4) + (v % 4), col j = l & 15;
   //             value = A[lane 16*block + i] * B[lane 16*block + j]
   int bad = 0, inexact = 0;
   for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 16; ++v) {
         const int blk = v / 4, i = 4 * (l >> 4) + (v % 4), j = l & 15;
         const float want = ha[16 * blk + i] * hb[16 * blk + j];
         if (memcmp(&want, &hd[l * 16 + v], 4)) { ++bad; if (bad < 5) printf("mismatch l=%d v=%d got %.9g want %.9g\n", l, v, hd[l * 16 + v], want); }
      }
   printf("layout hypothesis: %d mismatches of 1024 (0 => D[l][v] = A[16(v/4) + 4(l>>4) + v%%4] * B[16(v/4) + (l&15)], exact)\n", bad);
   float hin[128]; for (int i = 0; i < 128; ++i) hin[i] = 1.0f + i * 1e-3f;
   float *in; CK(hipMalloc(&in, 512)); CK(hipMemcpy(in, hin, 512, hipMemcpyHostToDevice));
   for (int wps : {1, 2, 3}) { bench<0>(out, in, wps); bench<8>(out, in, wps); bench<16>(out, in, wps); bench<24>(out, in, wps); }
   return 0;
}
