// enc_prims_test.hip -- checks, on the device, the lane-level assumptions k_enc_fused is built on (gfx950):
//   v_permlane16_swap / v_permlane32_swap as a four-quad all-reduce, DPP row shifts with zero fill, and a chain of two
//   v_mfma_f32_16x16x32_f16 GEMMs where the first accumulator tile is the second one's B operand in the enc_sigma k order,
//   plus the operand-swapped (transposed-output) form.
// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/enc_prims_test tools/enc_prims_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int enc_sigma(int kb, int q, int e) { return 32 * kb + 16 * (e >> 2) + 4 * q + (e & 3); }

__global__ void k_prims(float *out, const float *W1, const float *W2, const float *X, float *Y, float *Yt)
{
   const int lane = threadIdx.x, q = lane >> 4, lc = lane & 15;
   float v = (float)(lane * lane + 1);
   {
      // inline asm: through __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) either ties both operands to one register (same value twice) or
      // folds the two results into one (v_add v, v4, v4): measured here before this form
      float a = v, b = v;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
      float s = a + b;
      a = s; b = s;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
      out[lane] = a + b;
   }
   out[64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
   out[128 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x102, 0xf, 0xf, true));  // row_shl:2
   // GEMM 1: H[32][16] = W1[32][32] . X[32][16], hardware k order; GEMM 2: Y[16][16] = W2[16][32] . H with H's accumulators as the operand
   h8 xb;
   for (int e = 0; e < 8; ++e) xb[e] = (_Float16)X[(8 * q + e) * 16 + lc];
   f4 h[2];
   for (int mt = 0; mt < 2; ++mt) {
      h8 a;
      for (int e = 0; e < 8; ++e) a[e] = (_Float16)W1[(16 * mt + lc) * 32 + 8 * q + e];
      f4 c = {0, 0, 0, 0};
      h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb, c, 0, 0, 0);
   }
   h8 hb;
   for (int e = 0; e < 8; ++e) hb[e] = (_Float16)h[e >> 2][e & 3];
   h8 a2;
   for (int e = 0; e < 8; ++e) a2[e] = (_Float16)W2[lc * 32 + enc_sigma(0, q, e)];
   f4 c = {0, 0, 0, 0};
   const f4 y = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, hb, c, 0, 0, 0);
   for (int r = 0; r < 4; ++r) Y[(4 * q + r) * 16 + lc] = y[r];
   const f4 yt = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, a2, c, 0, 0, 0);      // swapped: D[col][row]
   for (int r = 0; r < 4; ++r) Yt[(4 * q + r) * 16 + lc] = yt[r];
}

int main()
{
   std::vector<float> W1(32 * 32), W2(16 * 32), X(32 * 16);
   for (size_t i = 0; i < W1.size(); ++i) W1[i] = (float)((int)(i * 7 % 13) - 6) / 8.0f;
   for (size_t i = 0; i < W2.size(); ++i) W2[i] = (float)((int)(i * 5 % 11) - 5) / 8.0f;
   for (size_t i = 0; i < X.size(); ++i) X[i] = (float)((int)(i * 3 % 7) - 3) / 4.0f;
   float *dW1, *dW2, *dX, *dY, *dYt, *dout;
   hipMalloc(&dW1, W1.size() * 4); hipMalloc(&dW2, W2.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, 256 * 4); hipMalloc(&dYt, 256 * 4); hipMalloc(&dout, 192 * 4);
   hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
   hipLaunchKernelGGL(k_prims, dim3(1), dim3(64), 0, 0, dout, dW1, dW2, dX, dY, dYt);
   std::vector<float> out(192), Y(256), Yt(256);
   hipMemcpy(out.data(), dout, 192 * 4, hipMemcpyDeviceToHost); hipMemcpy(Y.data(), dY, 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(Yt.data(), dYt, 256 * 4, hipMemcpyDeviceToHost);
   int bad = 0;
   for (int l = 0; l < 64; ++l) {
      auto f = [](int x) { return (float)(x * x + 1); };
      const float want = f(l) + f(l ^ 16) + f(l ^ 32) + f(l ^ 48);
      if (out[l] != want) { if (bad < 5) printf("quads_sum lane %d: %g want %g\n", l, out[l], want); ++bad; }
      const float w1 = (l & 15) >= 1 ? f(l - 1) : 0.0f, w2 = (l & 15) + 2 < 16 ? f(l + 2) : 0.0f;
      if (out[64 + l] != w1) { if (bad < 10) printf("row_shr1 lane %d: %g want %g\n", l, out[64 + l], w1); ++bad; }
      if (out[128 + l] != w2) { if (bad < 15) printf("row_shl2 lane %d: %g want %g\n", l, out[128 + l], w2); ++bad; }
   }
   printf("lane primitives: %s\n", bad ? "MISMATCH" : "ok");
   std::vector<double> H(32 * 16), Yr(256);
   for (int m = 0; m < 32; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < 32; ++k) s += (double)W1[m * 32 + k] * X[k * 16 + n]; H[m * 16 + n] = (double)(float)(_Float16)(float)s; }
   double e1 = 0, e2 = 0;
   for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
      double s = 0; for (int k = 0; k < 32; ++k) s += (double)W2[m * 32 + k] * H[k * 16 + n];
      e1 = fmax(e1, fabs(s - Y[m * 16 + n])); e2 = fmax(e2, fabs(s - Yt[n * 16 + m]));
   }
   printf("chained GEMM (accumulator as operand): max err %g; operand-swapped (transposed) form: max err %g\n", e1, e2);
   return 0;
}
