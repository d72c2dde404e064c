// softmax_fixture.hip -- TEST program (not part of the product library): the row softmax of the register-resident encoder kernels
// (kernels_layer1_regs.hip l1_block / kernels_encoder_fused.hip: scores in the accumulator layout, maximum and sum over a lane's registers and the four
// lane quads through the LDS crossbar -- enc_regs_prims.h quads_reduce_n --, exp2 of the log2(e)-scaled scores, one v_rcp_f32 of the row sum) applied to
// rows of any length, so that the reference's softmax fixture (test.c:900: [100, 100], tensor.h:751-784) can meet the PRIMITIVES the product's attention
// is built from.  The product instantiates them for 16- and 2 x 16-column rows; here a row spans NT column tiles of 16.
//   softmax_fixture in.f32 rows cols out.f32        (row-major float32 files)
#include "../../vadc_amd/csrc/enc_regs_prims.h"

#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <vector>

using namespace vadc;

template <int NT>
__global__ __launch_bounds__(64) void k_softmax_rows(const float *__restrict__ x, float *__restrict__ out, int rows, int cols)
{
   const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
   const int row = blockIdx.x * 16 + i;
   // lane (q, i) holds s[i][j = 16 t + 4 q + r] -- the accumulator layout of S^T = Q^T . K with the query index j along the registers and quads
   f4 s[NT];
#pragma unroll
   for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int j = 16 * t + 4 * q + r;
         s[t][r] = (row < rows && j < cols) ? x[(size_t)row * cols + j] * 1.4426950408889634f : -1.0e30f;   // the product folds log2(e) / sqrt(hd) into the Q rows; masked like its idle columns
      }
   float m[1] = {s[0][0]};
#pragma unroll
   for (int t = 0; t < NT; ++t) m[0] = max2(m[0], max2(max2(s[t][0], s[t][1]), max2(s[t][2], s[t][3])));
   quads_reduce_n<true, 1>(m, lane);
   float sum[1] = {0.0f};
#pragma unroll
   for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m[0]);
      sum[0] += (s[t][0] + s[t][1]) + (s[t][2] + s[t][3]);
   }
   quads_reduce_n<false, 1>(sum, lane);
   const float inv = __builtin_amdgcn_rcpf(sum[0]);
#pragma unroll
   for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
         const int j = 16 * t + 4 * q + r;
         if (row < rows && j < cols) out[(size_t)row * cols + j] = s[t][r] * inv;
      }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char **argv)
{
   if (argc != 5) { fprintf(stderr, "usage: %s in.f32 rows cols out.f32\n", argv[0]); return 1; }
   const int rows = atoi(argv[2]), cols = atoi(argv[3]);
   if (rows <= 0 || cols <= 0 || cols > 16 * 8) { fprintf(stderr, "rows > 0, 0 < cols <= 128\n"); return 1; }
   std::vector<float> h((size_t)rows * cols);
   FILE *f = fopen(argv[1], "rb");
   if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
   fclose(f);
   float *dx, *dy;
   CK(hipMalloc(&dx, h.size() * 4)); CK(hipMalloc(&dy, h.size() * 4));
   CK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
   const dim3 grid((rows + 15) / 16), block(64);
   const int nt = (cols + 15) / 16;
   switch (nt) {
   case 1: hipLaunchKernelGGL(k_softmax_rows<1>, grid, block, 0, 0, dx, dy, rows, cols); break;
   case 2: hipLaunchKernelGGL(k_softmax_rows<2>, grid, block, 0, 0, dx, dy, rows, cols); break;
   case 3: hipLaunchKernelGGL(k_softmax_rows<3>, grid, block, 0, 0, dx, dy, rows, cols); break;
   case 4: hipLaunchKernelGGL(k_softmax_rows<4>, grid, block, 0, 0, dx, dy, rows, cols); break;
   case 5: hipLaunchKernelGGL(k_softmax_rows<5>, grid, block, 0, 0, dx, dy, rows, cols); break;
   case 6: hipLaunchKernelGGL(k_softmax_rows<6>, grid, block, 0, 0, dx, dy, rows, cols); break;
   case 7: hipLaunchKernelGGL(k_softmax_rows<7>, grid, block, 0, 0, dx, dy, rows, cols); break;
   default: hipLaunchKernelGGL(k_softmax_rows<8>, grid, block, 0, 0, dx, dy, rows, cols); break;
   }
   CK(hipGetLastError());
   CK(hipDeviceSynchronize());
   CK(hipMemcpy(h.data(), dy, h.size() * 4, hipMemcpyDeviceToHost));
   f = fopen(argv[4], "wb");
   if (!f || fwrite(h.data(), 4, h.size(), f) != h.size()) { fprintf(stderr, "cannot write %s\n", argv[4]); return 1; }
   fclose(f);
   fflush(NULL);
   _exit(0);      // a short-lived GPU process beside the test runner's context: not through the HIP runtime's exit handlers (host/vadc_hip.c)
}
