// valu_rate2.hip -- follow-up probes: why do v_mul(sgpr)/v_add pairs run at 4 cycles each inside k_frontend?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));

// OP 0: mix, same sgpr, in place (reference, 2.4 cyc)
// OP 1: mix, 16 distinct sgprs
// OP 2: mix, distinct sgprs, 3-address adds (dst != src)
// OP 3: like 2 but body unrolled 32x (big code footprint: ~16 KB)
// OP 4: tree shaped: 8 mul (sgpr) into q[], adds of previous p[] (exactly the k_frontend slot), in-loop
// OP 5: adds only, 3-address, 48 live registers
template <int OP, int REP>
__global__ __launch_bounds__(256) void k(float *out, const float *kin, int iters)
{
   float x[64], p[8], g = 0;
   for (int i = 0; i < 64; ++i) x[i] = threadIdx.x * 0.001f + i;
   for (int i = 0; i < 8; ++i) p[i] = x[i];
   f16v s = *(const f16v *)kin;   // uniform -> SGPRs
   asm volatile("" : "+s"(s));
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < REP; ++r) {
         if (OP == 0) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
               asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i + 1]) : "s"(s[0]));
               asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 2) & 15]));
            }
         } else if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
               asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i + 1]) : "s"(s[i]));
               asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 2) & 15]));
            }
         } else if (OP == 2 || OP == 3) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
               asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[16 + i]) : "s"(s[i]), "v"(x[32 + i]));
               asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[17 + i]) : "v"(x[48 + (i & 7)]), "v"(x[40 + ((i + 3) & 7)]));
            }
         } else if (OP == 4) {
            float q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
               asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q[j]) : "s"(s[j + 8 * (r & 1)]), "v"(x[8 * j + (r & 7)]));
               if (j == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[0]) : "v"(p[0]), "v"(p[1]));
               if (j == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[2]) : "v"(p[2]), "v"(p[3]));
               if (j == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[4]) : "v"(p[4]), "v"(p[5]));
               if (j == 3) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[6]) : "v"(p[6]), "v"(p[7]));
               if (j == 4) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[0]) : "v"(p[0]), "v"(p[2]));
               if (j == 5) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[4]) : "v"(p[4]), "v"(p[6]));
               if (j == 6) asm volatile("v_add_f32 %0, %1, %2" : "=v"(g) : "v"(p[0]), "v"(p[4]));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = q[j];
         } else if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
               asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[16 + i]) : "v"(x[48 + (i & 7)]), "v"(x[32 + ((i + 3) & 15)]));
         } else if (OP == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
               asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[16 + i]) : "v"(x[48 + (i & 7)]), "v"(x[32 + ((i + 3) & 15)]));
         }
      }
   }
   float acc = g;
   for (int i = 0; i < 64; ++i) acc += x[i];
   for (int i = 0; i < 8; ++i) acc += p[i];
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP, int REP> int bench(const char *name, float *out, const float *kin, int wps)
{
   const int iters = 8000 / REP;
   const int blocks = 256 * wps;
   hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   hipLaunchKernelGGL((k<OP, REP>), dim3(blocks), dim3(256), 0, 0, out, kin, 10);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(a, 0));
   hipLaunchKernelGGL((k<OP, REP>), dim3(blocks), dim3(256), 0, 0, out, kin, iters);
   CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
   float ms; CK(hipEventElapsedTime(&ms, a, b));
   const int per_rep = (OP == 4) ? 15 : 16;
   const double instr_per_simd = (double)iters * REP * per_rep * wps;
   printf("%-52s waves/SIMD %d  %.3f ms -> %.2f cycles/instr @2.4GHz\n", name, wps, ms, ms * 1e6 / instr_per_simd * 2.4);
   return 0;
}

int main()
{
   float *out, *kin; CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&kin, 4096));
   float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 1e-4f;
   CK(hipMemcpy(kin, h, sizeof(h), hipMemcpyHostToDevice));
   for (int wps : {2, 3, 4}) {
      bench<0, 4>("mix same sgpr in-place", out, kin, wps);
      bench<1, 4>("mix distinct sgprs in-place", out, kin, wps);
      bench<2, 4>("mix distinct sgprs 3-address", out, kin, wps);
      bench<3, 128>("mix distinct sgprs 3-address, 2048-instr body", out, kin, wps);
      bench<4, 8>("k_frontend slot shape (8 mul + 7 add)", out, kin, wps);
      bench<4, 128>("k_frontend slot shape, 1920-instr body", out, kin, wps);
      bench<5, 4>("add only 3-address", out, kin, wps);
      bench<6, 4>("mul only 3-address (vgpr)", out, kin, wps);
   }
   return 0;
}
