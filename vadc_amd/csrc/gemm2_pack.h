// gemm2_pack.h -- host-side operand packing of k_frontend_gemm2 (kernels_frontend_gemm2.hip); shared by the engine and the device bench.
#pragma once
#include <cstddef>
#include <vector>

namespace vadc {

// basis: the reference's STFT filter bank [258][256] (rows 0..128 re, 129..257 im; silero_vad.py:22-66, stft.c:15-224), already verified to have the real-DFT
// symmetries (build_gemm_frontend in engine_weights.hip).  Slot j = tap j + 1 (j = 0 .. 127): slots 0 .. 126 pair taps (j + 1, 255 - j), slot 127 pairs the centre tap
// with itself -- its re weight is halved, its im weight is zero.
//   af2: [tile 0..7][kb 0..7][lane 0..63][8] = 256 x A[row][slot 16 kb + 8 (lane >> 5) + e];  tile t < 4: row = re of bin 32 t + (lane & 31);
//        tile t >= 4: row = im of bin 32 (t - 4) + ((lane & 31) + 16) % 32   (im rows rotated by 16: see the kernel's accumulator exchange)
//   ny2: [128] bin 128's re weights by slot (unscaled)
inline void pack_gemm2_frontend(const std::vector<float> &basis, std::vector<float> &af2, std::vector<float> &ny2)
{
   auto B = [&](int row, int n) { return basis[(size_t)row * 256 + n]; };
   af2.assign((size_t)8 * 8 * 64 * 8, 0.0f);
   ny2.assign(128, 0.0f);
   for (int t = 0; t < 8; ++t)
      for (int kb = 0; kb < 8; ++kb)
         for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
               const int slot = 16 * kb + 8 * (l >> 5) + e, n = slot + 1;
               float v;
               if (t < 4) {
                  const int bin = 32 * t + (l & 31);
                  v = (slot == 127) ? 0.5f * B(bin, 128) : B(bin, n);
               } else {
                  const int bin = 32 * (t - 4) + (((l & 31) + 16) & 31);
                  v = (slot == 127) ? 0.0f : B(129 + bin, n);
               }
               af2[(((size_t)t * 8 + kb) * 64 + l) * 8 + e] = 256.0f * v;
            }
   for (int slot = 0; slot < 128; ++slot) ny2[slot] = (slot == 127) ? 0.5f * B(128, 128) : B(128, slot + 1);
}

}  // namespace vadc
