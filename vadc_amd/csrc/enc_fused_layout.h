// enc_fused_layout.h -- LDS images of k_enc_fused (kernels_encoder_fused.hip): where the host packer (engine_weights.hip) puts every weight
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

// ---- Silero v4, stages 2-4 (k_enc_fused_v4, kernels_encoder_fused_v4.hip): ONE image -----------------------------------------------------
// conv block + strided 1x1 conv per stage, no transformer block; the same fragment form and k orders as above (stage 2: stacked [pointwise | projection], hardware k order)
template <int CIN, int D, bool PROJ>
struct EncV4LayerLayout {
   static constexpr int MT = D / 16, KB = D / 32;
   static constexpr int f_pw  = 0;
   static constexpr int f_pj  = f_pw + MT * kFragBytes;                     // stage 4 only (PROJ with 32 input channels)
   static constexpr int f_cv  = (PROJ && CIN == 32) ? f_pj + MT * kFragBytes : f_pj;
   static constexpr int f_end = f_cv + MT * KB * kFragBytes;
   static constexpr int v_dw   = 0;                    // [6][CIN]: taps 0..4, bias
   static constexpr int v_cb_b = v_dw + 6 * CIN;       // pointwise bias (+ projection bias)
   static constexpr int v_cv_b = v_cb_b + D;
   static constexpr int v_end  = (v_cv_b + D + 3) / 4 * 4;
};
typedef EncV4LayerLayout<16, 32, true>  EncV4L2;
typedef EncV4LayerLayout<32, 32, false> EncV4L3;
typedef EncV4LayerLayout<32, 64, true>  EncV4L4;
constexpr int kEncV4_L2F = 0;
constexpr int kEncV4_L3F = kEncV4_L2F + EncV4L2::f_end;
constexpr int kEncV4_L4F = kEncV4_L3F + EncV4L3::f_end;
constexpr int kEncV4_V2  = kEncV4_L4F + EncV4L4::f_end;              // bytes
constexpr int kEncV4_V3  = kEncV4_V2 + EncV4L2::v_end * 4;
constexpr int kEncV4_V4  = kEncV4_V3 + EncV4L3::v_end * 4;
constexpr int kEncV4Bytes = kEncV4_V4 + EncV4L4::v_end * 4;
static_assert(EncV4L2::f_end == 4 * kFragBytes && EncV4L3::f_end == 4 * kFragBytes && EncV4L4::f_end == 16 * kFragBytes, "fragment sizes");
static_assert(kEncV4Bytes % 16 == 0, "the image is copied in 16-byte pieces");

struct EncV4Args {
   const float *in;          // first stage's output [n][16][t1_pitch]
   const void *img;          // device copy of the LDS image
   void *out;                // split-fp16 LSTM-native tiles, ts steps per chunk (common.h lstm_xh_index)
   int n_chunks;
   ItemMap map;
   // the window's geometry (round 6: every window, both branches): steps per chunk of `in` as the first stage laid it out (<= 12: the built geometry's), of which the first t1 are
   // VALID (a window between two built ones: the rest enter as the zeros the depthwise conv's padding is), the third strided conv's stride (2; the 8 kHz branch: 1), and the
   // LSTM steps per chunk = the steps that leave stage 4: t1 -> t2 = (t1 + 1) / 2 -> ts = s3 == 2 ? (t2 + 1) / 2 : t2
   int t1_pitch, t1, s3, ts;
};

// ---- layer 1 (k_layer1_regs, kernels_layer1_regs.hip): ONE image ------------------------------------------------------------------------
// conv block 129 -> 16: the [pointwise | projection] weights as K = 32 split-fp16 A fragments (2 KB each) over k blocks
//   0..3: relu(dw(x)) of channels 32 kb + ..,   4..7: x of channels 32 (kb - 4) + ..
// with lane quad q of the B operand taking channels l1_channel(kb, q, e), e = 0..7: quads 0 / 1 and 2 / 3 sit 16 channels apart, so that the two quads
// of a 32-lane half read their 16 consecutive steps of a [channel][25] chunk image from disjoint LDS banks (25 * 16 = 16 mod 32).  Channel 128 (the
// Nyquist bin) is a K = 16 fragment of its own (k = 0: relu(dw(x128)), k = 1: x128, the rest zero).  The transformer block's GEMMs (D = 16) are
// K = 16 fragments of 2 KB: lane (q, m) holds W[m][4 q + e], e < 4, twice -- as 16 bytes [lo x 4 | hi x 4] (block LH, 1 KB) and as [hi x 4 | 0 x 4]
// (block H0, 1 KB): kernels_layer1_regs.hip, mm() -- an accumulator tile's four registers are the other operand as they are.
__host__ __device__ constexpr int l1_channel(int kb, int q, int e) { return 32 * kb + 16 * (q & 1) + 8 * (q >> 1) + e; }
// Silero v4's first stage reads a [channel][24] image: 24 * 16 = 0 (mod 32) -- with the map above the two quads of a half would meet in the same banks (they did:
// SQ_LDS_BANK_CONFLICT 14.9 M of 71.6 M LDS cycles per launch of 65,536 chunks).  24 * d = 16 (mod 32) needs d = 2 (mod 4): the quads of a half sit TWO channels
// apart, the halves one, a lane's eight channels four apart.
__host__ __device__ constexpr int l1v4_channel(int kb, int q, int e) { return 32 * kb + 4 * e + 2 * (q & 1) + (q >> 1); }
constexpr int kFrag4Bytes = 2048;
struct L1Layout {
   static constexpr int f_conv = 0;                                  // 8 x 2 KB
   static constexpr int f_tail = f_conv + 8 * kFragBytes;
   static constexpr int f_qkv  = f_tail + kFrag4Bytes;               // Q (rows pre-scaled by log2(e) / sqrt(8)), K, V
   static constexpr int f_out  = f_qkv + 3 * kFrag4Bytes;
   static constexpr int f_l1   = f_out + kFrag4Bytes;
   static constexpr int f_l2   = f_l1 + kFrag4Bytes;
   static constexpr int f_cv   = f_l2 + kFrag4Bytes;                 // LayerNorm 2's scale folded into its columns
   static constexpr int f_end  = f_cv + kFrag4Bytes;
   // vectors (floats, relative to the vector base at byte f_end)
   static constexpr int v_taps = 0;                                  // [4 kb][8 e][4 q][8]: k0 k1 k2 k3 | k4 bias bias 0 -- the four quads of a read 32 B apart: quads 256 B apart (q outside e) met in the same banks, 2.63 against 1.84 cycles per ds_read_b128 (tools/lds_conflict_probe.hip)
   static constexpr int v_tail = v_taps + 4 * 4 * 8 * 8;             // channel 128: the same 8 floats
   static constexpr int v_cb_b = v_tail + 8;
   static constexpr int v_q_b  = v_cb_b + 16, v_k_b = v_q_b + 16;
   static constexpr int v_out_b = v_k_b + 16;                        // + Wo . bv (the kernel adds no V bias)
   static constexpr int v_n1_w = v_out_b + 16, v_n1_b = v_n1_w + 16;
   static constexpr int v_l1_b = v_n1_b + 16, v_l2_b = v_l1_b + 16;
   static constexpr int v_n2_w = v_l2_b + 16, v_n2_b = v_n2_w + 16;  // (stage tap 2 only: the hot path has them folded into the strided conv)
   static constexpr int v_cv_b = v_n2_b + 16;
   static constexpr int v_end  = (v_cv_b + 16 + 3) / 4 * 4;
};
constexpr int kL1ImgBytes = L1Layout::f_end + L1Layout::v_end * 4;
static_assert(kL1ImgBytes % 16 == 0, "the image is copied in 16-byte pieces");
constexpr int kL1ChunkFloats = 129 * 25;
constexpr int kL1BufBytes = 808 * 16;        // a chunk's image: 12,900 bytes from the 16-byte boundary below the chunk (<= 12 bytes of lead) (k_layer1_regs_v4: one per wave)
constexpr int kL1SlabBytes = 208 * 16;       // k_layer1_regs: one k block's 3,200 bytes (+ the Nyquist channel's 100 behind k block 3) from the 16-byte boundary below them
constexpr int kL1RingBytes = 3 * kL1SlabBytes;      // a wave's ring of three slabs
constexpr int kL1Waves = 12;                 // waves per workgroup of k_layer1_regs (three per SIMD)
static_assert(600 * 16 + kL1SlabBytes == kL1BufBytes, "group 3 ends where the whole-chunk image ended");
constexpr int kL1YSlackBytes = 64;           // the DMA of the last chunk reads up to 28 bytes past it

struct L1RegsArgs {
   const float *y;           // [n][129][25] log-magnitudes (tap: [n][16][25], the conv block's output)
   const float *fm;          // [4][fm_stride] partial per-frame bin sums
   size_t fm_stride;
   const void *img;
   float *out;               // [n][16][13] (tap: [n][16][25])
   int n_chunks;
   ItemMap map;
   int tv = 0;               // k_layer1_regs_v4 only: the VALID frames of a chunk when fewer than the geometry's 24 (a window of 1344 / 1408 / 1472 samples); 0 = all
};

// ---- Silero v4, first stage (k_layer1_regs_v4, kernels_layer1_regs_v4.hip) ---------------------------------------------------------------
// Input channels 0..128 = magnitude (recovered from the log-magnitudes), 129..257 = normalized log-magnitude, both made from the SAME element of Y:
// 8 "virtual" k blocks vb = 2 kb + which (which = 0 magnitude, 1 normalized) of 32 channels l1_channel(kb, q, e) each; fragment vb = relu(dw(.)) .
// pointwise, fragment 8 + vb = (.) . projection.  Bin 128 of both halves is a K = 16 fragment: k = 0 relu(dw(m128)), 1 relu(dw(n128)), 2 m128, 3 n128.
struct L1V4Layout {
   static constexpr int f_conv = 0;                                  // 16 x 2 KB
   static constexpr int f_tail = f_conv + 16 * kFragBytes;
   static constexpr int f_cv   = f_tail + kFrag4Bytes;               // strided 1x1 conv (BatchNorm folded by the exporter)
   static constexpr int f_end  = f_cv + kFrag4Bytes;
   static constexpr int v_taps = 0;                                  // [8 vb][8 e][4 q][8]: k0 k1 k2 k3 | k4 bias bias 0 (as L1Layout::v_taps)
   static constexpr int v_tail = v_taps + 8 * 4 * 8 * 8;             // bin 128: magnitude's 8 floats, normalized's 8 floats
   static constexpr int v_cb_b = v_tail + 16;
   static constexpr int v_cv_b = v_cb_b + 16;
   static constexpr int v_end  = v_cv_b + 16;
};
constexpr int kL1V4ImgBytes = L1V4Layout::f_end + L1V4Layout::v_end * 4;
static_assert(kL1V4ImgBytes % 16 == 0, "the image is copied in 16-byte pieces");
constexpr int kL1V4Frames = 24;
constexpr int kL1V4ChunkFloats = 129 * kL1V4Frames;                  // 12,384 bytes = 774 units of 16: a chunk always starts on a 16-byte boundary
constexpr int kL1V4BufBytes = 774 * 16;
constexpr int kL1V4SlabBytes = 198 * 16;     // k_layer1_regs_v4: a k block's 192 units (+ bin 128's 6 behind k block 3)
constexpr int kL1V4RingBytes = 3 * kL1V4SlabBytes;
static_assert(3 * 192 * 16 + kL1V4SlabBytes == kL1V4BufBytes, "group 3 ends where the chunk ends");
static_assert(kL1V4ChunkFloats * 4 == kL1V4BufBytes && 32 * kL1V4Frames * 4 == 192 * 16, "a k block's 32 channels are exactly one DMA group of 192 units");

}  // namespace vadc
