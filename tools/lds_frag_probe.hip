// lds_frag_probe.hip -- LDS-pipe cost of the MFMA fragment layouts of kernels_v5.hip / kernels_lstm.hip on gfx950 (companion of lds_conflict_probe.hip): 16 waves of
// one CU each issue 16 independent accesses back to back, 2000 times; cycles per instruction per CU = the pipe's time for the lane -> address pattern
// (4 for a conflict-free ds_read_b128, MI355X_MICROARCH.md LDS table).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_frag_probe tools/lds_frag_probe.hip && tools/lds_frag_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>      // 0: ds_read_b128, 1: ds_write_b64, 2: ds_read_b64, 3: ds_read_u16
__global__ __launch_bounds__(1024) void k_probe(const int *addr, unsigned long long *cycles, float *sink)
{
   __shared__ __attribute__((aligned(16))) float lds[16384];
   for (int i = threadIdx.x; i < 16384; i += 1024) lds[i] = (float)i;
   __syncthreads();
   const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds + (unsigned)addr[threadIdx.x & 63];
   f4 acc = {0, 0, 0, 0};
   const f2 w = {1.0f, 2.0f};
   const unsigned long long t0 = __builtin_readcyclecounter();
   for (int it = 0; it < 2000; ++it) {
      f4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
         if (KIND == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(a));
         if (KIND == 1) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(w) : "memory");
         if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(*(f2 *)&v[k]) : "v"(a));
         if (KIND == 3) asm volatile("ds_read_u16 %0, %1" : "=v"(v[k][0]) : "v"(a));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (KIND != 1) {
#pragma unroll
         for (int k = 0; k < 16; ++k) acc[0] += v[k][0];
      }
   }
   const unsigned long long t1 = __builtin_readcyclecounter();
   if (threadIdx.x == 0) *cycles = t1 - t0;
   if (acc[0] == 12345.0f) sink[threadIdx.x] = acc[0];
}

static int swz(int kq, int kb) { const int i = kq + 4 * (kb & 1); return i < 4 ? i : i + 8; }

int main()
{
   int *d_addr; unsigned long long *d_c; float *d_s;
   hipMalloc(&d_addr, 256); hipMalloc(&d_c, 8); hipMalloc(&d_s, 4096);
   struct P { const char *name; int kind; int (*f)(int lane); };
   P pats[] = {
      {"read b128  lane*16 (contiguous: the reference)", 0, [](int l) { return l * 16; }},
      {"read b128  [column][channel] pitch 136 halves: col*272 + kq*16  (k_lstm_* h tile, k_v5_lstm_h3)", 0, [](int l) { return (l & 15) * 272 + (l >> 4) * 16; }},
      {"read b128  [column][channel] pitch 144 halves: col*288 + kq*16", 0, [](int l) { return (l & 15) * 288 + (l >> 4) * 16; }},
      {"read b128  [column][channel] pitch 160 halves: col*320 + kq*16", 0, [](int l) { return (l & 15) * 320 + (l >> 4) * 16; }},
      {"read b128  [column][channel] pitch 72 halves: col*144 + kq*16  (k_lstm_* h tile, rounds 2-5)", 0, [](int l) { return (l & 15) * 144 + (l >> 4) * 16; }},
      {"read b128  [column][channel] pitch 80 halves: col*160 + kq*16  (k_lstm_* h tile, round 6)", 0, [](int l) { return (l & 15) * 160 + (l >> 4) * 16; }},
      {"read b128  piece-major: kq*4096 + col*16  (k_v5_encoder_h3 activations)", 0, [](int l) { return (l >> 4) * 4096 + (l & 15) * 16; }},
      {"read b128  piece-major, stride-2 columns: kq*4096 + col*32  (conv 1 / 2)", 0, [](int l) { return (l >> 4) * 4096 + (l & 15) * 32; }},
      {"read b128  piece-major swizzled (fold planes, kb = 1): (kq*4+1)*256 + (col ^ f)*16", 0, [](int l) { return ((l >> 4) * 4 + 1) * 256 + ((l & 15) ^ swz(l >> 4, 1)) * 16; }},
      {"read b128  [column][channel] XOR form: col*256 + ((kq ^ (col & 3)) * 16)  (4 quarters of one 64-byte k-block row... 64-B rows)", 0, [](int l) { return (l & 15) * 64 + (((l >> 4) ^ ((l & 15) & 3)) * 16); }},
      {"write b64  epilogue piece-major: (q4>>1)*4096 + col*16 + 8*(q4&1)", 1, [](int l) { return ((l >> 4) >> 1) * 4096 + (l & 15) * 16 + 8 * ((l >> 4) & 1); }},
      {"write b64  fold, unswizzled: lane = 2 fq + fh: (kq*4+kb)*256 + 8 fh  (first form: 8-way?)", 1, [](int l) { const int fq = (l >> 1) & 15, fh = l & 1; return ((fq & 3) * 4 + (fq >> 2)) * 256 + 8 * fh + (l >> 5) * 16; }},
      {"write b64  fold, swizzled", 1, [](int l) { const int fq = (l >> 1) & 15, fh = l & 1, col = l >> 5; return ((fq & 3) * 4 + (fq >> 2)) * 256 + ((col ^ swz(fq & 3, fq >> 2)) * 16) + 8 * fh; }},
      {"write b64  [column][channel] pitch 136: col*272 + 32*wave + 8*q4 (k_v5_lstm_h3 h write, wave 0)", 1, [](int l) { return (l & 15) * 272 + 8 * (l >> 4); }},
      {"read b64   fold X reads: lane*8", 2, [](int l) { return l * 8; }},
      {"read u16   fold X reads, first form: lane*8", 3, [](int l) { return l * 8; }},
   };
   for (auto &p : pats) {
      int h[64];
      for (int l = 0; l < 64; ++l) h[l] = p.f(l);
      hipMemcpy(d_addr, h, 256, hipMemcpyHostToDevice);
      if (p.kind == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
      if (p.kind == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
      if (p.kind == 2) hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
      if (p.kind == 3) hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(1024), 0, 0, d_addr, d_c, d_s);
      unsigned long long c = 0;
      hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
      printf("%7.2f cycles per instruction per CU   %s\n", (double)c / (2000.0 * 16 * 16), p.name);
   }
   return 0;
}
