// mfma_f16_probe.hip -- operand layout and accuracy of v_mfma_f32_16x16x32_f16 on gfx950, and of the split-fp16 ("f16x3")
// evaluation of an fp32 product:  a*b ~= ah*bh + (ah*bl + al*bh)  with  ah = (half)a, al = (half)(a - ah).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// D[16x16] = A[16x32] . B[32x16];  candidate layout: lane l holds A[l & 15][8 (l >> 4) + e], B[8 (l >> 4) + e][l & 15], e = 0..7;
// D: lane l, reg r -> row 4 (l >> 4) + r, col l & 15
__global__ void k_layout(const float *A, const float *B, float *D)
{
   const int l = threadIdx.x;
   h8 a, b;
   for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + e]; b[e] = (_Float16)B[(8 * (l >> 4) + e) * 16 + (l & 15)]; }
   f4 acc = {0, 0, 0, 0};
   acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
   for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

// split evaluation of fp32 A . B with K = 128 (four k-blocks), three MFMAs per block
__global__ void k_split(const float *A, const float *B, float *D)
{
   const int l = threadIdx.x;
   f4 acc = {0, 0, 0, 0};
   for (int kb = 0; kb < 4; ++kb) {
      h8 ah, al, bh, bl;
      for (int e = 0; e < 8; ++e) {
         const float av = A[(l & 15) * 128 + 32 * kb + 8 * (l >> 4) + e], bv = B[(32 * kb + 8 * (l >> 4) + e) * 16 + (l & 15)];
         ah[e] = (_Float16)av; al[e] = (_Float16)(av - (float)ah[e]);
         bh[e] = (_Float16)bv; bl[e] = (_Float16)(bv - (float)bh[e]);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
   }
   for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

int main()
{
   std::vector<float> A(16 * 128), B(128 * 16), D(256);
   srand(1);
   for (auto &v : A) v = ((rand() % 2001) - 1000) / 1000.0f * 0.3f;          // LSTM-like weights
   for (auto &v : B) v = ((rand() % 2001) - 1000) / 1000.0f;                 // h in (-1, 1)
   float *dA, *dB, *dD;
   hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
   // layout check with K = 32 (first 32 columns of A rows re-packed)
   std::vector<float> A32(16 * 32), B32(32 * 16);
   for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A32[i * 32 + k] = (float)(_Float16)A[i * 128 + k];
   for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B32[k * 16 + j] = (float)(_Float16)B[k * 16 + j];
   hipMemcpy(dA, A32.data(), A32.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B32.data(), B32.size() * 4, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
   hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
   double worst = 0;
   for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int k = 0; k < 32; ++k) r += (double)A32[i * 32 + k] * B32[k * 16 + j]; worst = fmax(worst, fabs(r - D[i * 16 + j])); }
   printf("layout check (K=32, fp16-exact inputs): max |D - ref| = %.3e  (expect ~1e-7)\n", worst);
   hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dA, dB, dD);
   hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
   double w3 = 0, w32 = 0;
   for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double r = 0; float f = 0;
      for (int k = 0; k < 128; ++k) { r += (double)A[i * 128 + k] * B[k * 16 + j]; f = fmaf(A[i * 128 + k], B[k * 16 + j], f); }
      w3 = fmax(w3, fabs(r - D[i * 16 + j])); w32 = fmax(w32, fabs(r - f));
   }
   printf("K=128 fp32 data: split-fp16 (3 MFMA) max abs err %.3e ; plain fp32 fma chain max abs err %.3e\n", w3, w32);
   return 0;
}
