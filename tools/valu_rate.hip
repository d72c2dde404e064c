// valu_rate.hip -- measures the issue rate of plain vs packed fp32 VALU ops on gfx950 (tuning aid, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1, int iters)
{
   float a[16];
   f2 p[16];
   for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f2){a[i], a[i] + 0.5f}; }
   const f2 s = {s0, s1};
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
         for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
            if (OP == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 2) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "s"(s0));
            if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(s));
            if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "s"(s));
            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
            if (OP == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(a[(i + 1) & 15]));
            if (OP == 8) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 9) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 10) { if (i & 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0)); else asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15])); }
            if (OP == 11) { if (i & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(s)); else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15])); }
            if (OP == 12) { if (i & 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 2) & 15])); else asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15])); }
            if (OP == 13) { if (i & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "s"(s)); else { asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 2) & 15])); asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i+1]) : "v"(a[(i + 3) & 15])); } }
            if (OP == 14) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 15) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 16) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
            if (OP == 17) asm volatile("v_mul_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
            if (OP == 18) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
         }
   }
   float acc = 0;
   for (int i = 0; i < 16; ++i) acc += a[i] + p[i][0] + p[i][1];
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP> int bench(const char *name, float *out, int wps)
{
   const int iters = 2000;
   const int blocks = 256 * wps;     // wps waves per SIMD: 256 CUs x 4 SIMDs / 4 waves per block
   hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f, 10);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(a, 0));
   hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f, iters);
   CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
   float ms; CK(hipEventElapsedTime(&ms, a, b));
   const double instr_per_simd = (double)iters * 64 * wps;
   printf("%-28s waves/SIMD %d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
          ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
   return 0;
}

int main()
{
   float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
   for (int wps : {2, 4}) {
      bench<0>("v_mul_f32 (sgpr)", out, wps); bench<8>("v_mul_f32 (vgpr)", out, wps);
      bench<1>("v_add_f32 (vgpr)", out, wps); bench<16>("v_add_f32 (sgpr)", out, wps);
      bench<14>("v_sub_f32", out, wps); bench<15>("v_max_f32", out, wps);
      bench<2>("v_fma_f32 (sgpr)", out, wps); bench<9>("v_fma_f32 (vgpr)", out, wps);
      bench<3>("v_pk_mul_f32 (sgpr pair)", out, wps); bench<6>("v_pk_mul_f32 (vgpr)", out, wps);
      bench<4>("v_pk_add_f32", out, wps); bench<5>("v_pk_fma_f32", out, wps);
      bench<10>("mix mul(sgpr)+add", out, wps); bench<12>("mix mul(vgpr)+add", out, wps);
      bench<11>("mix pk_mul(sgpr)+pk_add", out, wps); bench<13>("mix pk_mul(sgpr)+2 add (3 instr/2 slots)", out, wps);
      bench<17>("v_mul_f32_dpp", out, wps); bench<18>("v_add_f32_dpp", out, wps);
   }
   return 0;
}
