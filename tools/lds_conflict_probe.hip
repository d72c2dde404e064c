// lds_conflict_probe.hip -- what a lane -> address pattern costs in the LDS pipe of gfx950, for the read shapes k_layer1_regs uses: 16 waves of one CU each issue 16
// independent reads back to back, 2000 times; the CU's cycles per read instruction = the pipe's time for the pattern (4 for a conflict-free ds_read_b128).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_conflict_probe tools/lds_conflict_probe.hip && tools/lds_conflict_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>      // 0: ds_read_b128, 1: ds_read2_b32 offset1 = 9 dwords, 2: ds_read_b32, 3: ds_read2_b32 offset1 = 64 dwords
__global__ __launch_bounds__(1024) void k_probe(const int *addr, unsigned long long *cycles, float *sink)
{
   __shared__ __attribute__((aligned(16))) float lds[16384];
   for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = (float)i;
   __syncthreads();
   const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds + (unsigned)addr[threadIdx.x & 63];
   f4 acc = {0, 0, 0, 0};
   const unsigned long long t0 = __builtin_readcyclecounter();
   for (int it = 0; it < 2000; ++it) {
      f4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
         if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(a));
         if (KIND == 1) { float x, y; asm volatile("ds_read2_b32 %0, %1 offset1:9" : "=v"(*(float __attribute__((ext_vector_type(2))) *)&v[k]) : "v"(a)); (void)x; (void)y; }
         if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(v[k][0]) : "v"(a));
         if (KIND == 3) asm volatile("ds_read2_b32 %0, %1 offset1:64" : "=v"(*(float __attribute__((ext_vector_type(2))) *)&v[k]) : "v"(a));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[0] += v[k][0];
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   if (threadIdx.x == 0) *cycles = t1 - t0;
   if (acc[0] == 12345.0f) sink[threadIdx.x] = acc[0];
}

int main()
{
   int *d_addr; unsigned long long *d_c; float *d_s;
   hipMalloc(&d_addr, 256); hipMalloc(&d_c, 8); hipMalloc(&d_s, 256);
   struct P { const char *name; int kind; int (*f)(int lane); };
   static const int qo[4] = {0, 16, 8, 24};
   P pats[] = {
      {"b128 lane*16 (contiguous)", 0, [](int l) { return l * 16; }},
      {"b128 one address per quad, quads 256 B apart (taps today)", 0, [](int l) { return (l >> 4) * 256; }},
      {"b128 one address per quad, quads 32 B apart", 0, [](int l) { return (l >> 4) * 32; }},
      {"b128 one address per quad, quads 16 B apart (biases)", 0, [](int l) { return (l >> 4) * 16; }},
      {"b128 one address per quad, quads 272 B apart", 0, [](int l) { return (l >> 4) * 272; }},
      {"b128 all lanes one address", 0, [](int) { return 0; }},
      {"b128 two fragments: lane*16 (+1 KB apart is another instruction)", 0, [](int l) { return 1024 + l * 16; }},
      {"read2_b32 +9: x today (quad channel offsets 0,16,8,24 x 25 floats)", 1, [](int l) { return (qo[l >> 4] * 25 + (l & 15)) * 4; }},
      {"read2_b32 +9: x today, chunk lead 12 bytes", 1, [](int l) { return (qo[l >> 4] * 25 + (l & 15)) * 4 + 12; }},
      {"read2_b32 +9: quad channel offsets 0,1,2,3", 1, [](int l) { return ((l >> 4) * 25 + (l & 15)) * 4; }},
      {"read2_b32 +9: quad channel offsets 0,1,16,17", 1, [](int l) { return ((((l >> 4) & 1) + 16 * (l >> 5)) * 25 + (l & 15)) * 4; }},
      {"read2_b32 +9: quad channel offsets 0,16,1,17", 1, [](int l) { return ((16 * ((l >> 4) & 1) + (l >> 5)) * 25 + (l & 15)) * 4; }},
      {"read2_b32 +9: quad channel offsets 0,2,4,6", 1, [](int l) { return ((l >> 4) * 50 + (l & 15)) * 4; }},
      {"read_b32: x today, first dword only", 2, [](int l) { return (qo[l >> 4] * 25 + (l & 15)) * 4; }},
      {"read_b32: lane*4", 2, [](int l) { return l * 4; }},
      {"read2_b32 +64: lane*4 (ideal read2)", 3, [](int l) { return l * 4; }},
   };
   for (const P &p : pats) {
      int h[64];
      for (int l = 0; l < 64; ++l) h[l] = p.f(l);
      hipMemcpy(d_addr, h, 256, hipMemcpyHostToDevice);
      unsigned long long c = 0;
      for (int rep = 0; rep < 2; ++rep) {
         if (p.kind == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
         if (p.kind == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
         if (p.kind == 2) hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
         if (p.kind == 3) hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
         hipDeviceSynchronize();
         hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
      }
      printf("%-75s %6.2f cycles per read\n", p.name, (double)c / (2000.0 * 16 * 16));
   }
   return 0;
}
