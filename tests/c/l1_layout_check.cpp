// The invariants k_layer1_regs (vadc_amd/csrc/kernels_layer1_regs.hip) builds on, checked at compile time on the host from the header both the
// kernel and the engine's image packer use.  g++ -std=c++17 -fsyntax-only (tests/test_abi.py).
#define __host__
#define __device__
#include <cstddef>
namespace vadc { struct ItemMap { int C, c0, cg; }; }
#include "enc_fused_layout.h"
using namespace vadc;

// every input channel 0..127 is the (kb, q, e) of exactly one B-operand slot
constexpr bool channels_are_a_bijection()
{
   bool seen[128] = {};
   for (int kb = 0; kb < 4; ++kb)
      for (int q = 0; q < 4; ++q)
         for (int e = 0; e < 8; ++e) {
            const int c = l1_channel(kb, q, e);
            if (c < 0 || c >= 128 || seen[c]) return false;
            seen[c] = true;
         }
   return true;
}
static_assert(channels_are_a_bijection(), "l1_channel");

// ds_read_b32 banks = (byte / 4) mod 32, conflicts inside a 32-lane half: the half's two lane quads read 16 consecutive floats each of a [channel][25]
// image -- their first floats must sit 16 banks apart, whatever the chunk's lead (a multiple of 4 bytes)
constexpr bool quads_of_a_half_hit_disjoint_banks()
{
   for (int kb = 0; kb < 4; ++kb)
      for (int e = 0; e < 8; ++e)
         for (int half = 0; half < 2; ++half) {
            const int a = l1_channel(kb, 2 * half, e) * 25, b = l1_channel(kb, 2 * half + 1, e) * 25;
            if (((b - a) % 32 + 32) % 32 != 16) return false;
         }
   return true;
}
static_assert(quads_of_a_half_hit_disjoint_banks(), "LDS banks");
// the same for Silero v4's [channel][24] image and its own map
constexpr bool v4_channels_are_a_bijection_on_disjoint_banks()
{
   bool seen[128] = {};
   for (int kb = 0; kb < 4; ++kb)
      for (int q = 0; q < 4; ++q)
         for (int e = 0; e < 8; ++e) {
            const int c = l1v4_channel(kb, q, e);
            if (c < 0 || c >= 128 || seen[c]) return false;
            seen[c] = true;
         }
   for (int e = 0; e < 8; ++e)
      for (int half = 0; half < 2; ++half) {
         const int a = l1v4_channel(0, 2 * half, e) * 24, b = l1v4_channel(0, 2 * half + 1, e) * 24;
         if (((b - a) % 32 + 32) % 32 != 16) return false;
      }
   return true;
}
static_assert(v4_channels_are_a_bijection_on_disjoint_banks(), "l1v4_channel");

// the chunk image: 12,900 bytes from the 16-byte boundary below the chunk's first byte (lead <= 12) fit 808 units of 16 bytes
static_assert(kL1ChunkFloats * 4 + 12 <= kL1BufBytes && kL1BufBytes == 808 * 16, "chunk image");
static_assert(kL1YSlackBytes >= kL1BufBytes - kL1ChunkFloats * 4, "the last chunk's copy stays inside the allocation");
// k_layer1_regs: group kb = 208 units (3 pieces of 64 lanes and one of 16) from unit 200 kb of that image.  It holds k block kb's bytes
// [lead + 3200 kb, lead + 3200 (kb + 1)) -- and, kb = 3, the Nyquist channel's 100 behind them -- whatever the lead, and group 3 ends where the image ends
constexpr bool dma_groups_hold_their_k_blocks()
{
   for (int kb = 0; kb < 4; ++kb)
      for (int lead = 0; lead <= 12; lead += 4) {
         const int first = lead + 3200 * kb, last = lead + 3200 * (kb + 1) + (kb == 3 ? 100 : 0);      // bytes of the image the k block reads: [first, last)
         if (first < 200 * kb * 16 || last > 200 * kb * 16 + kL1SlabBytes) return false;
      }
   return 3 * 64 + 16 == kL1SlabBytes / 16 && 200 * 3 * 16 + kL1SlabBytes == kL1BufBytes;
}
static_assert(dma_groups_hold_their_k_blocks(), "DMA groups");
// the taps of a channel: the four lane quads of one ds_read_b128 read 32 bytes apart (quads 256 bytes apart met in the same banks: tools/lds_conflict_probe.hip)
static_assert(L1Layout::v_taps % 4 == 0, "taps");
// the image and the waves' buffers fit a CU's 160 KB of LDS (twelve rings of three slabs);
// fragments and vectors are 16-byte aligned
static_assert(kL1ImgBytes + kL1Waves * kL1RingBytes + 64 <= 160 * 1024, "LDS budget");
static_assert(kL1V4ImgBytes + kL1Waves * kL1V4RingBytes + 64 <= 160 * 1024, "LDS budget (v4)");
static_assert(L1Layout::f_tail % 16 == 0 && L1Layout::f_qkv % 16 == 0 && L1Layout::f_end % 16 == 0 && L1Layout::v_tail % 4 == 0 && L1Layout::v_cb_b % 4 == 0 &&
              L1Layout::v_q_b % 4 == 0 && L1Layout::v_cv_b % 4 == 0, "alignment of 16-byte LDS reads");
// k_enc_fused's images: phase B is the larger one and fits beside nothing else
static_assert(kEncLdsBytes <= 160 * 1024 && kEncA_Bytes <= kEncLdsBytes, "k_enc_fused LDS");
int main() { return 0; }
