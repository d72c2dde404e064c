// kernels_layer1_regs_v4.hip -- the first encoder stage of Silero v4 at its default window (258 channels x 24 frames -> 16 channels x 12 steps), every
// activation in registers: k_layer1_regs' conv block (kernels_layer1_regs.hip: two overlapping 16-column tiles per chunk, the chunk by LDS-DMA in four
// groups with counted waits, split-fp16 MFMAs, the MFMAs of a k block issued between the next block's channel groups) without a transformer block.
//
// Replaces (reference file:line): the first ConvBlock + strided conv + ReLU of silero_vad.py's v4 graph (conv_block: conv.c:761-814, dw :17-113,
// pw / proj :532-589; conv k = 1 stride 2: conv.c:597-709), the concat(magnitude, normalized) in front of it and the last step of the adaptive
// normalization (misc.c:65-96).  What is particular to this stage:
// * one element of Y = log1p(2^20 m) feeds TWO input channels: the magnitude m = (e^Y - 1) 2^-20 (bins 0..128 of the concat; engine option "v4_mag" = 0:
//   the front end writes no magnitude array) and Y - offset (bins 129..257).  A k block of 32 bins is processed twice from the same LDS bytes -- 8
//   virtual k blocks vb = 2 kb + which, K = 516 + 4.
// * 24 frames: tile 0 carries steps 0..15 and owns 0..11, tile 1 carries steps 8..23 and owns 12..23.  A chunk is 12,384 bytes = 774 units of 16: it
//   starts on a 16-byte boundary and a k block's 32 bins are exactly one DMA group of 192 units.
// * stride 2 keeps the even steps, which sit on even lanes in BOTH tiles: tile 1's results move one lane up (DPP) so that every lane owns at most one
//   output and the four stores per iteration that the waits count stay four.
#include "common.h"
#include "enc_fused_layout.h"
#include "enc_regs_prims.h"
#include "l1_regs_prims.h"
#include <algorithm>

namespace vadc {

typedef L1V4Layout L4;
constexpr int kT4 = kL1V4Frames;

struct L1V4Stage { f2 x[4]; f4 ka[4], kb[4]; };      // x[c] = Y at this lane's step in (tile 0, tile 1)
__device__ __forceinline__ L1V4Stage l1v4_load_stage(const float *xb, const float *tp, int vb, int half)
{
   L1V4Stage s;
   const int kb = vb >> 1;
#pragma unroll
   for (int c = 0; c < 4; ++c) {
      const int e = 4 * half + c;
      s.x[c] = f2{xb[(32 * kb + e) * kT4], xb[(32 * kb + e) * kT4 + 8]};
      s.ka[c] = lds_vec4(tp, (vb * 8 + e) * 32);
      s.kb[c] = lds_vec4(tp, (vb * 8 + e) * 32 + 4);
   }
   return s;
}
// magnitude from Y = log1p(2^20 m) (k_layer_mfma's form: one v_exp_f32 and one fma), or Y - offset (misc.c:84-96)
template <int WHICH>
__device__ __forceinline__ f2 l1v4_input(const f2 &y, float off)
{
   if (WHICH == 0)
      return f2{fmaf(__builtin_amdgcn_exp2f(y[0] * 1.44269504088896340736f), 0x1p-20f, -0x1p-20f), fmaf(__builtin_amdgcn_exp2f(y[1] * 1.44269504088896340736f), 0x1p-20f, -0x1p-20f)};
   return y - f2{off, off};
}
template <int WHICH>
__device__ __forceinline__ void l1v4_channel_math(const L1V4Stage &s, int c, float off, f4 &x0, f4 &x1, f4 &d0, f4 &d1)
{
   const f2 xp = l1v4_input<WHICH>(s.x[c], off);
   x0[c] = xp[0]; x1[c] = xp[1];
   float a, b;
   dw5x2(xp[0], xp[1], s.ka[c][0], s.ka[c][1], s.ka[c][2], s.ka[c][3], s.kb[c][0], s.kb[c][1], a, b);
   d0[c] = relu(a); d1[c] = relu(b);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_layer1_regs_v4(L1RegsArgs a)
{
   __shared__ __attribute__((aligned(16))) char lds[kL1V4ImgBytes + NW * kL1V4BufBytes];
   const int tid = threadIdx.x;
   const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
   const int q = lane >> 4, lc = lane & 15;
   const int slot = blockIdx.x * NW + wave, nslots = gridDim.x * NW;
   const char *img = lds;
   const float *vec = reinterpret_cast<const float *>(lds + L4::f_end);
   char *buf = lds + kL1V4ImgBytes + wave * kL1V4BufBytes;

   // the input pipeline of k_layer1_regs: per iteration 4 loads (partial sums), 3 + 3 + 3 + 4 DMA pieces (groups of 192 units = the four k blocks; the
   // last piece: bin 128, 6 units), 4 stores, in that order; counted waits below
   float fmv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   const int lo16 = lane * 16;
   auto issue_sums = [&](int nn) {
      const float *fmp = a.fm + (size_t)nn * kT4;
      const int lo4 = (lane < kT4 ? lane : 0) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("global_load_dword %0, %1, %2" : "=v"(fmv[k]) : "v"(lo4), "s"(fmp + k * a.fm_stride) : "memory");
   };
   auto issue_group = [&](int nn, int g) {
      const char *src = reinterpret_cast<const char *>(a.y) + (size_t)nn * kL1V4BufBytes;
      const unsigned dst = (unsigned)(uintptr_t)(l1_lds_void_t *)buf;
#pragma unroll
      for (int j = 3 * g; j < 3 * g + 3; ++j)
         asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + j * 1024), "v"(lo16), "s"(src + j * 1024) : "memory");
      if (g == 3 && lane < 6) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst + 12 * 1024), "v"(lo16), "s"(src + 12 * 1024) : "memory");
   };
#define L1V4_WAIT(more, n_more, n_last) do { if (more) asm volatile("s_waitcnt vmcnt(" #n_more ")" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" #n_last ")" ::: "memory"); } while (0)
   if (slot < a.n_chunks) {
      const int n0 = a.map(slot);
      issue_sums(n0);
#pragma unroll
      for (int g = 0; g < 4; ++g) issue_group(n0, g);
   }
   {  // image -> LDS, 8 loads in flight per thread
      const uint4 *src = reinterpret_cast<const uint4 *>(a.img);
      uint4 *dst = reinterpret_cast<uint4 *>(lds);
      constexpr int n = kL1V4ImgBytes / 16;
      for (int i0 = 0; i0 < n; i0 += 8 * 64 * NW) {
         uint4 v[8];
#pragma unroll
         for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 * NW + tid; v[u] = src[i < n ? i : 0]; }
#pragma unroll
         for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 * NW + tid; if (i < n) dst[i] = v[u]; }
      }
   }
   __syncthreads();
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

   for (int item = slot; item < a.n_chunks; item += nslots) {
      const int n = a.map(item);
      const bool more = item + nslots < a.n_chunks;           // wave-uniform
      const int nnext = more ? a.map(item + nslots) : n;
      // outstanding, oldest first: sums 4, groups 3 3 3 4, stores 4.  The sums have landed when 17 are left.
      asm volatile("s_waitcnt vmcnt(17)" : "+v"(fmv[0]), "+v"(fmv[1]), "+v"(fmv[2]), "+v"(fmv[3]) :: "memory");
      // ---- adaptive normalization offset of the chunk (misc.c:65-82) over its 24 frames ----
      float off;
      {
         const float fms = ((fmv[0] + fmv[1]) + (fmv[2] + fmv[3])) / 129.0f;
         const float filt[7] = {0.03663284704089164733887f, 0.11128076165914535522461f, 0.21674531698226928710938f,
                                0.27068215608596801757812f, 0.21674531698226928710938f, 0.11128076165914535522461f,
                                0.03663284704089164733887f};
         const int t = lane < kT4 ? lane : 0;
         float nb[7];
#pragma unroll
         for (int i = 0; i < 7; ++i) {
            int qq = t + i - 3;                                 // reflect pad 3, no edge repeat
            qq = qq < 0 ? -qq : qq;
            qq = qq >= kT4 ? 2 * (kT4 - 1) - qq : qq;
            nb[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * qq, __builtin_bit_cast(int, fms)));
         }
         float r = 0.0f;
#pragma unroll
         for (int i = 0; i < 7; ++i) r += nb[i] * filt[i];
         r = lane < kT4 ? r : 0.0f;
         r += dpp_row<0x111>(r); r += dpp_row<0x112>(r); r += dpp_row<0x114>(r); r += dpp_row<0x118>(r);     // row_shr:1, 2, 4, 8: lane 15 of a row = its sum
         const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 15)) +
                             __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 31));
         off = total / (float)kT4;
      }
      // ---- conv block: y = relu(pw(relu(dw(x))) + proj(x)), x = concat(magnitude, normalized) ----
      const float *xb = reinterpret_cast<const float *>(buf) + (16 * (q & 1) + 8 * (q >> 1)) * kT4 + lc;
      const float *tp = vec + L4::v_taps + q * 8;
      f4 acc[2];
      acc[0] = acc[1] = lds_vec4(vec, L4::v_cb_b + 4 * q);
      asm volatile("s_waitcnt vmcnt(13)" ::: "memory");        // k block 0: group 0 (and the first piece of group 1)
      L1V4Stage sa = l1v4_load_stage(xb, tp, 0, 0), sb;
      Frag wd = lds_frag(img + L4::f_conv, 0, lane), wx = lds_frag(img + L4::f_conv, 8, lane);
      Frag pd0, pd1, px0, px1, pwd, pwx;                         // pending: operands and weights of the previous virtual k block
      auto pending_mfma = [&](int i) {
         const int t = i & 1, term = i >> 1;
         const Frag &w = term < 3 ? pwd : pwx;
         const Frag &o = term < 3 ? (t ? pd1 : pd0) : (t ? px1 : px0);
         const int k = term % 3;
         acc[t] = k == 0 ? MFMA16(w.lo, o.hi, acc[t]) : (k == 1 ? MFMA16(w.hi, o.lo, acc[t]) : MFMA16(w.hi, o.hi, acc[t]));
      };
      f2 ytail = {0.0f, 0.0f};
#pragma unroll
      for (int vb = 0; vb < 8; ++vb) {
         const bool have = vb > 0;
         const int kb = vb >> 1;
         sb = l1v4_load_stage(xb, tp, vb, 1);
         __builtin_amdgcn_sched_barrier(0);
         f4 xl0, xl1, dl0, dl1, xh0, xh1, dh0, dh1;
#pragma unroll
         for (int c = 0; c < 4; ++c) {
            if (vb & 1) l1v4_channel_math<1>(sa, c, off, xl0, xl1, dl0, dl1); else l1v4_channel_math<0>(sa, c, off, xl0, xl1, dl0, dl1);
            if (have) pending_mfma(c);
            __builtin_amdgcn_sched_barrier(0);
         }
         Frag wdn = wd, wxn = wx;
         // the next virtual k block's first stage: k block kb + 1 needs its group (done so far must be 11 / 14 / 17 of the last iteration's 21
         // operations; this one has issued 0 / 7 / 10 more if it issues at all) -- the same k block's second half is already there
         if (vb == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
         if (vb == 3) L1V4_WAIT(more, 14, 7);
         if (vb == 5) L1V4_WAIT(more, 14, 4);
         if (vb < 7) {
            sa = l1v4_load_stage(xb, tp, vb + 1, 0);
            wdn = lds_frag(img + L4::f_conv, vb + 1, lane); wxn = lds_frag(img + L4::f_conv, 8 + vb + 1, lane);
         } else {
            const float *xt = reinterpret_cast<const float *>(buf) + 128 * kT4 + lc;      // bin 128
            ytail = f2{xt[0], xt[8]};
            sa.ka[0] = lds_vec4(vec, L4::v_tail); sa.kb[0] = lds_vec4(vec, L4::v_tail + 4);
            sa.ka[1] = lds_vec4(vec, L4::v_tail + 8); sa.kb[1] = lds_vec4(vec, L4::v_tail + 12);
         }
         __builtin_amdgcn_sched_barrier(0);
#pragma unroll
         for (int c = 0; c < 4; ++c) {
            if (vb & 1) l1v4_channel_math<1>(sb, c, off, xh0, xh1, dh0, dh1); else l1v4_channel_math<0>(sb, c, off, xh0, xh1, dh0, dh1);
            if (have) pending_mfma(4 + c);
            __builtin_amdgcn_sched_barrier(0);
         }
         // k block kb has been read by both of its halves: group kb of the next chunk may overwrite it
         if (more && (vb & 1) && kb < 3) {
            if (kb == 0) issue_sums(nnext);
            issue_group(nnext, kb);
         }
         const Frag df0 = split8(dl0, dh0);
         if (have) pending_mfma(8);
         __builtin_amdgcn_sched_barrier(0);
         const Frag df1 = split8(dl1, dh1);
         if (have) pending_mfma(9);
         __builtin_amdgcn_sched_barrier(0);
         const Frag xf0 = split8(xl0, xh0);
         if (have) pending_mfma(10);
         __builtin_amdgcn_sched_barrier(0);
         const Frag xf1 = split8(xl1, xh1);
         if (have) pending_mfma(11);
         __builtin_amdgcn_sched_barrier(0);
         pd0 = df0; pd1 = df1; px0 = xf0; px1 = xf1; pwd = wd; pwx = wx;
         wd = wdn; wx = wxn;
      }
      {
         const AOp wt = lds_aop(img + L4::f_tail, lane);
         const f2 m = l1v4_input<0>(ytail, off), nn2 = l1v4_input<1>(ytail, off);      // (uses the last LDS read of the chunk)
         if (more) issue_group(nnext, 3);
#pragma unroll
         for (int i = 0; i < 4; ++i) pending_mfma(i);
         float dm0, dm1, dn0, dn1;
         dw5x2(m[0], m[1], sa.ka[0][0], sa.ka[0][1], sa.ka[0][2], sa.ka[0][3], sa.kb[0][0], sa.kb[0][1], dm0, dm1);
         dw5x2(nn2[0], nn2[1], sa.ka[1][0], sa.ka[1][1], sa.ka[1][2], sa.ka[1][3], sa.kb[1][0], sa.kb[1][1], dn0, dn1);
#pragma unroll
         for (int i = 4; i < 8; ++i) pending_mfma(i);
         const h8 b0 = split4_hl(f4{relu(dm0), relu(dn0), m[0], nn2[0]}), b1 = split4_hl(f4{relu(dm1), relu(dn1), m[1], nn2[1]});
#pragma unroll
         for (int i = 8; i < 12; ++i) pending_mfma(i);
         acc[0] = mm(wt, b0, acc[0]);
         acc[1] = mm(wt, b1, acc[1]);
      }
      // ---- conv k = 1 stride 2 -> ReLU on every step; the store keeps the even ones ----
      const AOp wc = lds_aop(img + L4::f_cv, lane);
      const f4 bc = lds_vec4(vec, L4::v_cv_b + 4 * q);
      f4 z[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
         f4 y;
#pragma unroll
         for (int r = 0; r < 4; ++r) y[r] = relu(acc[t][r]);
         z[t] = mm(wc, split4_hl(y), bc);
#pragma unroll
         for (int r = 0; r < 4; ++r) z[t][r] = relu(z[t][r]);
      }
      // even steps: tile 0 lanes 0, 2, .. 10 (steps 0..10); tile 1 (steps 8 + lane) lanes 4, 6, .. 14 (steps 12..22), moved one lane up: odd lanes 5..15
      {
         const bool odd = lc & 1;
         float v[4];
#pragma unroll
         for (int r = 0; r < 4; ++r) { const float up = dpp_row<0x111>(z[1][r]); v[r] = odd ? up : z[0][r]; }
         const int step = odd ? 7 + lc : lc;
         float *op = a.out + (size_t)n * (16 * 12) + (4 * q) * 12 + (step >> 1);
         if (odd ? lc >= 5 : lc <= 10) {
#pragma unroll
            for (int r = 0; r < 4; ++r) op[r * 12] = v[r];
         }
      }
   }
}

// max_wgs: workgroups the grid may use (CUs not held by the LSTM chain); 8 waves per workgroup (see launch_layer1_regs)
void launch_layer1_regs_v4(const L1RegsArgs &a, int max_wgs, hipStream_t st)
{
   if (a.n_chunks <= 0) return;
   const int g = std::min(max_wgs, (a.n_chunks + 7) / 8);
   hipLaunchKernelGGL((k_layer1_regs_v4<8>), dim3(g), dim3(512), 0, st, a);
}

}  // namespace vadc
