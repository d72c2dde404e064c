// frontend_gemm2_common.h -- pieces shared by the 32x32x16 GEMM front ends (kernels_frontend_gemm2.hip, kernels_frontend_gemm4.hip): geometries, the s16 fold's
// exact (hi, lo) split, masked buffer stores, the lane-half sum.  Device code only.
#pragma once
#include "common.h"

namespace vadc {

typedef float g2_f16v __attribute__((ext_vector_type(16)));
typedef float g2_f4v __attribute__((ext_vector_type(4)));
typedef _Float16 g2_h8v __attribute__((ext_vector_type(8)));
typedef _Float16 g2_h2v __attribute__((ext_vector_type(2)));
typedef float g2_f2v __attribute__((ext_vector_type(2)));
typedef unsigned g2_u4v __attribute__((ext_vector_type(4)));
typedef unsigned g2_u2v __attribute__((ext_vector_type(2)));

template <int GEO> struct G2Geo;
// GEO as in kernels_frontend_gemm.hip: 0 = Silero v3.1 (pad 128, 25 frames), 1 / 2 / 3 = Silero v4 16 kHz with 1536- / 1024- / 512-sample windows, 4 / 5 = the 8 kHz
// branch's 768- / 256-sample windows.  chunks = chunks per group: 96 positions = 3 column tiles of 32 (v3.1: 125 of 128).
template <> struct G2Geo<0> { static constexpr int samples = 1536, pad = 128, frames = 25, chunks = 5; };
template <> struct G2Geo<1> { static constexpr int samples = 1536, pad = 96, frames = 24, chunks = 4; };
template <> struct G2Geo<2> { static constexpr int samples = 1024, pad = 96, frames = 16, chunks = 6; };
template <> struct G2Geo<3> { static constexpr int samples = 512, pad = 96, frames = 8, chunks = 12; };
template <> struct G2Geo<4> { static constexpr int samples = 768, pad = 96, frames = 12, chunks = 8; };
template <> struct G2Geo<5> { static constexpr int samples = 256, pad = 96, frames = 4, chunks = 24; };

constexpr int kG2BlockPitch = 72;                 // halves per 64-sample block of the padded chunk
constexpr unsigned kG2Oob = 0x80000000u;          // OR-ed into the buffer-store offset of a masked lane (>= num_records: the store is dropped)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g2_rsrc(const void *p)
{
   return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void g2_store(float v, __amdgpu_buffer_rsrc_t r, unsigned off)
{
   __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
// x + (x of lane ^ 32) in every lane
__device__ __forceinline__ float g2_sum_halves(float x)
{
   // Inline asm, because __builtin_amdgcn_permlane32_swap(x, x) ties both operands to one register (enc_regs_prims.h) -- and therefore with its own wait states:
   // hipcc's hazard recogniser does not look into asm text, and a swap issued right behind the v_mov that makes its operand read the register's OLD value
   // (round 5: bin 128 of a workgroup's first tile came out wrong, where the copy and the swap were adjacent; two instructions apart they were right).
   float a = x, b = x;
   asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));      // a = {x[0..31], x[0..31]}, b = {x[32..63], x[32..63]}
   return a + b;
}
__device__ __forceinline__ int g2_sext_lo(unsigned w) { return (int)(short)(w & 0xffffu); }
__device__ __forceinline__ int g2_sext_hi(unsigned w) { return (int)w >> 16; }

// two fold values (17-bit integers as floats) -> packed (hi, lo) halves: hi = round toward zero, lo = v - hi (exact)
__device__ __forceinline__ void g2_split2(float a, float b, g2_h2v &hi, g2_h2v &lo)
{
   hi = __builtin_bit_cast(g2_h2v, __builtin_amdgcn_cvt_pkrtz(a, b));
   float ra, rb;
   asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
   asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
   const g2_f2v r = {ra, rb};
   lo = __builtin_convertvector(r, g2_h2v);
}
// A fragments keep 22 bits: hi = round to nearest, lo = the rest
__device__ __forceinline__ void g2_split8_rn(const float (&v)[8], g2_h8v &hi, g2_h8v &lo)
{
#pragma unroll
   for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
}

}  // namespace vadc
