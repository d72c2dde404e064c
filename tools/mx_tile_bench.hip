// mx_tile_bench.hip -- times the generated STFT tile (tools/gen_mx_asm.py) in isolation: 2 workgroups of 4 waves per CU,
// every wave runs `tiles` tiles back to back on a synthetic x tile in LDS and the real basis layout size in global memory.
// Build one binary per generator variant:  hipcc --offload-arch=gfx950 -O3 -DTILE_INC='"/path/variant.inc"' ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include TILE_INC
typedef float f16acc __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_float_t;

__global__ __launch_bounds__(256, 2) void k_bench(const float *__restrict__ bt, float *__restrict__ out, int tiles)
{
   __shared__ __attribute__((aligned(16))) float xs[4 * 1904];
   for (int i = threadIdx.x; i < 4 * 1904; i += 256) xs[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.0f / 65536.0f) - 0.5f;
   __syncthreads();
   const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const unsigned xaddr = (unsigned)(uintptr_t)(lds_float_t *)(xs + 68 * lane);
   const unsigned boff = (lane & 15) * 16;
   f16acc acc = {0};
#pragma unroll 1
   for (int t = 0; t < tiles; ++t) {
      const float *bbase = bt + (size_t)__builtin_amdgcn_readfirstlane((wave + 4 * t) % 17) * 4096;
      f16acc y;
      asm volatile(VADC_MX_TILE_ASM : VADC_MX_TILE_Y_CONSTRAINT(y) : [xaddr] "v"(xaddr), [boff] "v"(boff), [bbase] "s"(bbase) : VADC_MX_TILE_CLOBBERS);
      acc += y;
   }
   float s = 0;
   for (int e = 0; e < 16; ++e) s += acc[e];
   out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv)
{
   const int tiles = argc > 1 ? atoi(argv[1]) : 8, wgs = argc > 2 ? atoi(argv[2]) : 512;
   float *bt, *out;
   hipMalloc(&bt, 17 * 4096 * 4 + 65536); hipMalloc(&out, (size_t)wgs * 256 * 4);
   std::vector<float> h(17 * 4096 + 16384);
   for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() & 0xffff) / 65536.0f - 0.5f;
   hipMemcpy(bt, h.data(), h.size() * 4, hipMemcpyHostToDevice);
   hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
   hipLaunchKernelGGL(k_bench, dim3(wgs), dim3(256), 0, 0, bt, out, tiles);
   hipDeviceSynchronize();
   hipEventRecord(a);
   for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_bench, dim3(wgs), dim3(256), 0, 0, bt, out, tiles);
   hipEventRecord(b); hipEventSynchronize(b);
   float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
   // per tile per wave in cycles at 2.4 GHz, and the equivalent full-frontend time (6400 WGs x 17 tiles / (wgs x 4 waves))
   const double cyc_tile = ms * 1e-3 * 2.4e9 / tiles;
   printf("%s: %.3f ms, %.0f cycles/tile/wave (%.1f per tap), frontend-equivalent %.3f ms\n", TILE_INC, ms, cyc_tile, cyc_tile / 256,
          ms / tiles * (6400.0 * 17 / (wgs * 4.0)));
   float hs[4]; hipMemcpy(hs, out, 16, hipMemcpyDeviceToHost); printf("  check %g\n", hs[0]);
   return 0;
}
