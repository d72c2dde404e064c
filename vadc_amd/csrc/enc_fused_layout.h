// enc_fused_layout.h -- LDS images of k_enc_fused (kernels_encoder_fused.hip): where the host packer (engine.hip) puts every weight
// fragment and vector of encoder layers 2-4, and where the kernel finds them.  Two images, copied verbatim into LDS:
//   image A = layers 2 and 3 (phase A of the kernel), image B = layer 4 (phase B).
//
// GEMM weights are split-fp16 A fragments of v_mfma_f32_16x16x32_f16, one 2 KB block per (M tile, k block):
//   [hi: 64 lanes x 8 halves][lo: 64 lanes x 8 halves]      lane l = (q = l >> 4, m = l & 15) holds W[16 mt + m][k(kb, q, e)], e = 0..7
// with the LOGICAL k order of an accumulator tile used as the next operand (the output channels of the GEMM before):
//   k(kb, q, e) = 32 kb + 16 (e >> 2) + 4 q + (e & 3)        (enc_sigma)
// because lane (q, column) of a 16x16 accumulator holds rows 16 mt + 4 q + r of its column: registers [2 kb][0..3], [2 kb + 1][0..3] of a
// lane ARE its 8 operand elements of k block kb -- no LDS round trip, no lane movement between the GEMMs of a layer.
// Layer 2's conv block reads its 16 input channels from memory instead, so its stacked [pointwise | projection] weight uses the
// hardware order k = 8 q + e over [relu(dw(x)) channels 0..15 | x channels 0..15].
// Vectors are fp32 in natural channel order (a lane reads the float4 at 16 mt + 4 q: its four accumulator rows).
#pragma once

namespace vadc {

__host__ __device__ constexpr int enc_sigma(int kb, int q, int e) { return 32 * kb + 16 * (e >> 2) + 4 * q + (e & 3); }

constexpr int kFragBytes = 2048;          // one (M tile, k block): hi 1 KB + lo 1 KB

// offsets of one transformer layer's fragments (bytes, relative to the layer's fragment base) and vectors (floats, relative to its vector base)
template <int CIN, int D, bool PROJ>
struct EncLayerLayout {
   static constexpr int MT = D / 16, KB = D / 32, KBC = 1;                  // conv block: K = 32 for every layer (layer 2: 16 + 16 stacked)
   static constexpr int f_pw  = 0;
   static constexpr int f_pj  = f_pw + MT * KBC * kFragBytes;               // layer 4 only (PROJ with 32 input channels); layer 2's projection is stacked into f_pw
   static constexpr int f_qkv = (PROJ && CIN == 32) ? f_pj + MT * KBC * kFragBytes : f_pj;
   static constexpr int f_out = f_qkv + 3 * MT * KB * kFragBytes;
   static constexpr int f_l1  = f_out + MT * KB * kFragBytes;
   static constexpr int f_l2  = f_l1 + MT * KB * kFragBytes;
   static constexpr int f_cv  = f_l2 + MT * KB * kFragBytes;
   static constexpr int f_end = f_cv + MT * KB * kFragBytes;
   // vectors (floats)
   static constexpr int v_dw   = 0;                    // [6][CIN]: taps 0..4, bias
   static constexpr int v_cb_b = v_dw + 6 * CIN;       // pointwise bias (+ projection bias)
   static constexpr int v_qkv_b = v_cb_b + D;          // [3 D], Q part pre-scaled
   static constexpr int v_out_b = v_qkv_b + 3 * D;
   static constexpr int v_n1_w = v_out_b + D, v_n1_b = v_n1_w + D;
   static constexpr int v_l1_b = v_n1_b + D, v_l2_b = v_l1_b + D;
   static constexpr int v_n2_w = v_l2_b + D, v_n2_b = v_n2_w + D;
   static constexpr int v_cv_b = v_n2_b + D;
   static constexpr int v_end  = (v_cv_b + D + 63) / 64 * 64;
};

typedef EncLayerLayout<16, 32, true>  EncL2;
typedef EncLayerLayout<32, 32, false> EncL3;
typedef EncLayerLayout<32, 64, true>  EncL4;

constexpr int kEncA_L2F = 0;
constexpr int kEncA_L3F = kEncA_L2F + EncL2::f_end;
constexpr int kEncA_V2  = kEncA_L3F + EncL3::f_end;                 // bytes
constexpr int kEncA_V3  = kEncA_V2 + EncL2::v_end * 4;
constexpr int kEncA_Bytes = kEncA_V3 + EncL3::v_end * 4;
constexpr int kEncB_L4F = 0;
constexpr int kEncB_V4  = kEncB_L4F + EncL4::f_end;
constexpr int kEncB_Bytes = kEncB_V4 + EncL4::v_end * 4;
constexpr int kEncLdsBytes = kEncA_Bytes > kEncB_Bytes ? kEncA_Bytes : kEncB_Bytes;
static_assert(EncL2::f_end == 32768 && EncL3::f_end == 32768 && EncL4::f_end == 131072, "fragment sizes");
static_assert(kEncA_Bytes % 16 == 0 && kEncB_Bytes % 16 == 0, "images are copied in 16-byte pieces");

// floats per chunk of the phase A -> phase B scratch (a batch's layer-3 output in register order: [tile][8 registers][64 lanes], 2 chunks per tile)
constexpr int kEncScratchPerChunk = 256;

struct EncFusedArgs {
   const float *in;          // first == 2: layer-1 output [n][16][13]; first == 3 / 4: [n][32][7]
   const void *imgA, *imgB;  // device copies of the two LDS images
   float *scratch;           // [batches][tile][8][64]
   void *out;                // last == 4: split-fp16 LSTM-native tiles (common.h lstm_xh_index), unless tap4
   float *tap2, *tap3, *tap4;   // optional [n][32][7], [n][32][7], [n][64][7] (stage taps; tap4 replaces the LSTM tiles)
   int n_chunks, first, last;
   ItemMap map;
};

}  // namespace vadc
