// pk_rate.hip -- issue rate of the front end's instruction mix on the whole chip, with the shader clock measured alongside.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o tools/pk_rate
// Every variant runs 256 x 4 workgroups of 256 threads (4 waves/SIMD on every CU) for `iters` iterations of an unrolled body of
// 64 independent instructions; prints ns and shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

// OP 0: v_pk_mul_f32 v, v, v     1: v_pk_mul_f32 v, v, s[pair]   2: v_pk_add_f32 v, v, v   3: v_add_f32 v, v, v
// OP 4: the tree mix: 8 pk_mul (sgpr pair) + 7 pk_add (3-address, dependent as in the tree)
// OP 5: the tree mix with VGPR-pair taps       6: v_mul_f32 v, s, v      7: tree mix, unpacked (16 v_mul sgpr + 14 v_add)
template <int OP>
__global__ __launch_bounds__(256, 4) void k(float *out, const float *kin, int iters, long long *t)
{
   f2v x[16], q[8], g = {0, 0};
   for (int i = 0; i < 16; ++i) x[i] = (f2v){threadIdx.x * 0.001f + i, 1.0f + i};
   f16v s = *(const f16v *)kin;   // uniform -> SGPRs
   asm volatile("" : "+s"(s));
   const long long w0 = wall_clock64(), c0 = clock64();
   for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         if (OP == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q[i & 7]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
         } else if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q[i & 7]) : "v"(x[i]), "s"((f2v){s[2 * (i & 7)], s[2 * (i & 7) + 1]}));
         } else if (OP == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(q[i & 7]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
         } else if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(q[i & 7].x) : "v"(x[i].x), "v"(x[(i + 1) & 15].y));
         } else if (OP == 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q[i & 7].x) : "s"(s[i]), "v"(x[i].x));
         } else if (OP == 4 || OP == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
               if (OP == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q[i]) : "v"(x[i + 8 * (r & 1)]), "s"((f2v){s[2 * i], s[2 * i + 1]}));
               else         asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q[i]) : "v"(x[i + 8 * (r & 1)]), "v"(x[(i + 3) & 15]));
            }
            f2v a01, a23, a45, a67, b0, b1;
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a01) : "v"(q[0]), "v"(q[1]));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a23) : "v"(q[2]), "v"(q[3]));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a45) : "v"(q[4]), "v"(q[5]));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a67) : "v"(q[6]), "v"(q[7]));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(b0) : "v"(a01), "v"(a23));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(b1) : "v"(a45), "v"(a67));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(g) : "v"(b0), "v"(b1));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[r]) : "v"(g), "v"(x[r]));    // 16th instruction: keeps the count at 16 per r
         } else if (OP == 7) {
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[i]) : "s"(s[i]), "v"(x[i].x));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[2 * i]), "v"(p[2 * i + 1]));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[2 * i]), "v"(p[2 * i + 1]));
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[2 * i]), "v"(p[2 * i + 1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[r].y) : "v"(p[0]), "v"(p[1]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[r + 4].y) : "v"(p[0]), "v"(p[1]));   // 32 per r
         }
      }
   }
   const long long w1 = wall_clock64(), c1 = clock64();
   float acc = g.x + g.y;
   for (int i = 0; i < 8; ++i) acc += q[i].x + q[i].y;
   for (int i = 0; i < 16; ++i) acc += x[i].x + x[i].y;
   out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
   if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
}

template <int OP>
static int run(const char *name, float *out, float *kin, long long *t, int per_iter)
{
   const int iters = 20000;
   hipEvent_t a, b;
   CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
   hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, out, kin, 2000, t);
   CK(hipDeviceSynchronize());
   CK(hipEventRecord(a, 0));
   hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, out, kin, iters, t);
   CK(hipEventRecord(b, 0));
   CK(hipEventSynchronize(b));
   float ms = 0;
   CK(hipEventElapsedTime(&ms, a, b));
   long long h[2];
   CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
   const double wall_s = h[0] / 100e6;
   const double mhz = h[1] / wall_s / 1e6;                       // clock64 ticks per second (shader clock if s_memtime counts it)
   const double instr_per_simd = (double)iters * per_iter * 4;   // 4 waves per SIMD
   printf("%-46s %8.3f ms  %6.3f ns/instr/SIMD  clock64 %7.1f MHz  => %5.2f cycles @2.4GHz  %5.2f cycles @clock64\n", name, ms,
          ms * 1e6 / instr_per_simd, mhz, ms * 1e-3 * 2.4e9 / instr_per_simd, ms * 1e-3 * mhz * 1e6 / instr_per_simd);
   return 0;
}

int main()
{
   float *out, *kin; long long *t;
   CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&kin, 256)); CK(hipMalloc(&t, 16));
   CK(hipMemset(kin, 0, 256));
   run<0>("v_pk_mul_f32 v,v,v", out, kin, t, 64);
   run<1>("v_pk_mul_f32 v,v,s[pair]", out, kin, t, 64);
   run<2>("v_pk_add_f32 v,v,v", out, kin, t, 64);
   run<3>("v_add_f32 v,v,v", out, kin, t, 64);
   run<6>("v_mul_f32 v,s,v", out, kin, t, 64);
   run<4>("tree mix packed, SGPR-pair taps (8 mul + 8 add)", out, kin, t, 64);
   run<5>("tree mix packed, VGPR-pair taps", out, kin, t, 64);
   run<7>("tree mix unpacked (16 mul sgpr + 16 add)", out, kin, t, 128);
   return 0;
}
