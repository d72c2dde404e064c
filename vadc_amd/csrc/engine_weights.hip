// engine_weights.hip -- the weights container (.testtensor, tensor.h:97-253) and the packers that turn its tensors into the device images the kernels read:
// fragment-major MFMA operands, split-fp16 (hi, lo) pairs, LDS images of the persistent kernels, the folded DFT basis, and the checks that decide which kernels may serve
// an engine (basis symmetries, weights inside fp16's range).  Host code only; called once per engine by vadc_amd_create (engine.hip).
#include "engine_internal.h"

// ---------------------------------------------------------------------------------------------------
// weights container (tensor.h:97-102, 201-253) -> host tensors, positional wiring tensor.h:114-191
// ---------------------------------------------------------------------------------------------------

bool vadc::parse_testtensor(const unsigned char *p, size_t len, std::vector<HostTensor> &out)
{
   size_t off = 0;
   auto rd = [&](int32_t &v) { if (off + 4 > len) return false; memcpy(&v, p + off, 4); off += 4; return true; };
   int32_t version, count;
   if (!rd(version) || !rd(count) || version != 1 || count <= 0 || count > 4096) return false;
   for (int i = 0; i < count; ++i) {
      int32_t n;
      if (!rd(n) || n <= 0 || off + (size_t)n > len) return false;
      off += (size_t)n;
   }
   out.resize(count);
   for (int i = 0; i < count; ++i) {
      int32_t ndim, size, nbytes;
      if (!rd(ndim) || ndim < 0 || ndim > 8) return false;
      out[i].dims.resize(ndim);
      long prod = 1;
      for (int d = 0; d < ndim; ++d) { int32_t v; if (!rd(v) || v <= 0) return false; out[i].dims[d] = v; prod *= v; }
      if (!rd(size) || !rd(nbytes) || size <= 0 || nbytes != size * 4 || prod != size) return false;
      if (off + (size_t)nbytes > len) return false;
      out[i].data = reinterpret_cast<const float *>(p + off);   // may be unaligned: only memcpy'd below
      out[i].size = size;
      off += (size_t)nbytes;
   }
   return off == len;
}

static void copy_unaligned(std::vector<float> &dst, const HostTensor &t)
{
   dst.resize(t.size);
   memcpy(dst.data(), t.data, (size_t)t.size * 4);
}

// first encoder stage, K = 1 MFMA form (kernels_encoder_mfma.hip): per input channel the 16 pointwise and the 16 projection
// weights as one 128-byte row [ch][pw 0..15 | proj 0..15]
static std::vector<float> k1_pack(const std::vector<float> &pw, const std::vector<float> &pj, int D, int C)
{
   std::vector<float> r((size_t)((C + 3) / 4 * 4) * 2 * D, 0.0f);      // zero rows up to 4 x (channels per wave): the kernel's last wave reads them
   for (int c = 0; c < C; ++c)
      for (int o = 0; o < D; ++o) { r[((size_t)c * 2 + 0) * D + o] = pw[(size_t)o * C + c]; r[((size_t)c * 2 + 1) * D + o] = pj[(size_t)o * C + c]; }
   return r;
}

// k_frontend_sym (kernels_frontend.hip) evaluates the reference's tree for bins 0..32 only and derives the other 96 bins from
//     re[128-b][n] = (-1)^n re[b][n]     im[128-b][n] = -(-1)^n im[b][n]
//     re[64-b][n] = {re, -im, -re, im}[b][n] by n % 4     im[64-b][n] = {-im, -re, im, re}[b][n] by n % 4
// which hold BIT FOR BIT for the reference's forward_basis_buffer (a zero may carry either sign).  Checked here on the loaded
// tensor [258][256]; any other basis (a perturbed test tensor, a re-exported model) runs the full tree of k_frontend_fl instead.
static bool basis_has_dft_symmetries(const std::vector<float> &basis)
{
   auto RE = [&](int k, int n) { return basis[(size_t)k * 256 + n]; };
   auto IM = [&](int k, int n) { return basis[(size_t)(kBins + k) * 256 + n]; };
   auto same = [](float a, float b) { return a == b; };               // +0 == -0: the sign of a zero tap never reaches a magnitude
   for (int b = 0; b <= 64; ++b)
      for (int n = 0; n < 256; ++n) {
         const float sg = (n & 1) ? -1.0f : 1.0f;
         if (!same(RE(128 - b, n), sg * RE(b, n)) || !same(IM(128 - b, n), -sg * IM(b, n))) return false;
         float er, ei;
         switch (n & 3) {
         case 0:  er = RE(b, n);  ei = -IM(b, n); break;
         case 1:  er = -IM(b, n); ei = -RE(b, n); break;
         case 2:  er = -RE(b, n); ei = IM(b, n);  break;
         default: er = IM(b, n);  ei = RE(b, n);  break;
         }
         if (!same(RE(64 - b, n), er) || !same(IM(64 - b, n), ei)) return false;
      }
   for (float v : basis) if (!(fabsf(v) < 3.0e38f)) return false;      // NaN / inf never compare equal to their mirror, but be explicit
   return true;
}

// GEMM front end (kernels_frontend_gemm.hip): needs re rows even about tap 128, im rows odd, tap 0 zero, im rows of bins 0
// and 128 zero -- verified bit for bit on the loaded basis [258][256]; otherwise the tree kernel stays in charge.
static bool build_gemm_frontend(const std::vector<float> &basis, Packer &pk, size_t &off_afrag, size_t &off_nyq, size_t &off_afrag2, size_t &off_nyq2)
{
   auto B = [&](int row, int n) { return basis[(size_t)row * 256 + n]; };
   for (int k = 0; k < kBins; ++k) {
      if (B(k, 0) != 0.0f || B(kBins + k, 0) != 0.0f || B(kBins + k, 128) != 0.0f) return false;
      for (int n = 1; n < 128; ++n)
         if (B(k, n) != B(k, 256 - n) || B(kBins + k, n) != -B(kBins + k, 256 - n)) return false;
   }
   for (int n = 0; n < 256; ++n) if (B(kBins, n) != 0.0f || B(kBins + 128, n) != 0.0f) return false;
   // A fragments (v_mfma_f32_16x16x32_f16 operand order): tile t < 8: re bins 16 t + r; tile 8 + t: im bins 16 t + r;
   // k-block kb, lane l, element e: row l & 15, tap 32 kb + 8 (l >> 4) + e
   std::vector<float> af((size_t)16 * 4 * 64 * 8), ny(128);
   for (int t = 0; t < 16; ++t)
      for (int kb = 0; kb < 4; ++kb)
         for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 8; ++el) {
               const int bin = 16 * (t & 7) + (l & 15), n = 32 * kb + 8 * (l >> 4) + el;
               float v;
               if (t < 8) v = (n == 0) ? B(bin, 128) : B(bin, n);          // slot 0 carries the unpaired centre tap
               else       v = (n == 0) ? 0.0f : B(kBins + bin, n);
               af[(((size_t)t * 4 + kb) * 64 + l) * 8 + el] = v;
            }
   for (int n = 0; n < 128; ++n) ny[n] = (n == 0) ? B(128, 128) : B(128, n);
   off_afrag = pk.add(af.data(), af.size());
   off_nyq = pk.add(ny.data(), ny.size());
   std::vector<float> af2, ny2;                 // the second form's operands (gemm2_pack.h)
   pack_gemm2_frontend(basis, af2, ny2);
   off_afrag2 = pk.add(af2.data(), af2.size());
   off_nyq2 = pk.add(ny2.data(), ny2.size());
   return true;
}

int vadc::build_weights(vadc_amd_engine *e, const std::vector<HostTensor> &ts)
{
   if (ts.size() != 99) return fail(VADC_AMD_EWEIGHTS, "weights: expected 99 tensors, found %zu", ts.size());
   Packer pk;
   std::vector<float> tmp, tmp2;
   int idx = 0;
   auto need = [&](int i, int n) { return ts[i].size == n; };

   // [0] STFT basis [258,1,256] -> consumption order of k_frontend (PK form): [f][i = 3,2,1,0][l / 2][j][l % 2], so that
   // the taps of tree lanes (l, l+1) for one j are an even-aligned SGPR pair = second operand of one v_pk_mul_f32
   if (!need(idx, kFilters * kFilterLen)) return fail(VADC_AMD_EWEIGHTS, "weights: bad forward_basis_buffer");
   copy_unaligned(tmp, ts[idx++]);
   tmp2.resize(tmp.size());
   for (int f = 0; f < kFilters; ++f)
      for (int ii = 0; ii < 4; ++ii)
         for (int lp = 0; lp < 4; ++lp)
            for (int j = 0; j < 8; ++j)
               for (int b = 0; b < 2; ++b)
                  tmp2[(size_t)f * 256 + ii * 64 + lp * 16 + j * 2 + b] = tmp[(size_t)f * 256 + 64 * (3 - ii) + 8 * j + (2 * lp + b)];
   const size_t off_basis = pk.add(tmp2.data(), tmp2.size());
   size_t off_afrag = 0, off_nyq = 0, off_afrag2 = 0, off_nyq2 = 0;
   e->gemm_ok = build_gemm_frontend(tmp, pk, off_afrag, off_nyq, off_afrag2, off_nyq2);   // FAST_STFT precision mode
   e->sym_ok = basis_has_dft_symmetries(tmp);
   e->zero_im0 = true;
   for (int n = 0; n < 256; ++n) if (tmp[(size_t)kBins * 256 + n] != 0.0f) e->zero_im0 = false;      // (-0 == 0)
   pk.add(nullptr, 512);  // the tap pipelines' final prefetch reads up to 1 KB past the last im row (k_frontend_fl: one (group, l-pair) block of "filter 258"): keep slack

   struct LOff { size_t dw_w, dw_b, pwT, pw_b, pjT, pj_b, qkv_w, qkv_b, out_w, out_b, n1_w, n1_b, l1_w, l1_b, l2_w, l2_b, n2_w, n2_b, cv_w, cv_b;
                 size_t pw_f, pj_f, cb_b, qkv_f, out_f, l1_f, l2_f, cv_f, pwj_k1, qkv_h, out_h, l1_h, l2_h, cv_h, pw_h, pj_h; } lo[4];
   // split-fp16 A fragments for v_mfma_f32_16x16x32_f16 (kernels_encoder_mfma.hip, H3): [m-tile][k-block][lane][hi 8 | lo 8] halves, lane l holds
   // W[16 mt + (l & 15)][32 kb + 8 (l >> 4) + e]; two halves per float slot of the packer.  `h3_ok` = every weight fits fp16's range.
   bool h3_ok = true;
   auto frag_h3 = [&h3_ok](const std::vector<float> &W, int M, int K) {
      const int KB = K / 32;
      std::vector<_Float16> h((size_t)(M / 16) * KB * 64 * 16);
      for (int mt = 0; mt < M / 16; ++mt)
         for (int kb = 0; kb < KB; ++kb)
            for (int l = 0; l < 64; ++l)
               for (int el = 0; el < 8; ++el) {
                  const float v = W[(size_t)(16 * mt + (l & 15)) * K + 32 * kb + 8 * (l >> 4) + el];
                  if (!(fabsf(v) < 60000.0f)) h3_ok = false;
                  const _Float16 hi = (_Float16)v;
                  h[(((size_t)mt * KB + kb) * 64 + l) * 16 + el] = hi;
                  h[(((size_t)mt * KB + kb) * 64 + l) * 16 + 8 + el] = (_Float16)(v - (float)hi);
               }
      std::vector<float> f(h.size() / 2);
      memcpy(f.data(), h.data(), h.size() * sizeof(_Float16));
      return f;
   };
   // MFMA A-fragment order for v_mfma_f32_16x16x4_f32: [m-tile][k-step][lane], lane l holds W[16mt + (l&15)][4kk + (l>>4)]
   auto frag = [](const std::vector<float> &W, int M, int K) {
      const int KKW = (K + 3) / 4;
      std::vector<float> f((size_t)(M / 16) * KKW * 64, 0.0f);
      for (int mt = 0; mt < M / 16; ++mt)
         for (int kk = 0; kk < KKW; ++kk)
            for (int l = 0; l < 64; ++l) {
               const int k = 4 * kk + (l >> 4);
               if (k < K) f[((size_t)mt * KKW + kk) * 64 + l] = W[(size_t)(16 * mt + (l & 15)) * K + k];
            }
      return f;
   };
   struct RawLayer { std::vector<float> dw_w, dw_b, pw, pw_b, pj, pj_b, qkv_w, qkv_b, out_w, out_b, n1_w, n1_b, l1_w, l1_b, l2_w, l2_b, n2_w, n2_b, cv_w, cv_b; } raw[4];   // for the fused kernel's LDS images (cv: BatchNorm folded)
   for (int l = 0; l < 4; ++l) {
      const LayerShape &s = kLayers[l];
      const int D = s.d, C = s.cin;
      auto take = [&](int n, std::vector<float> &v) -> bool { if (!need(idx, n)) return false; copy_unaligned(v, ts[idx++]); return true; };
      auto transposed = [&](const std::vector<float> &src) { std::vector<float> r((size_t)C * D); for (int o = 0; o < D; ++o) for (int c = 0; c < C; ++c) r[(size_t)c * D + o] = src[(size_t)o * C + c]; return r; };
      std::vector<float> v;
      if (!take(C * 5, v)) goto bad; lo[l].dw_w = pk.add(v.data(), v.size()); raw[l].dw_w = v;
      if (!take(C, v)) goto bad;     lo[l].dw_b = pk.add(v.data(), v.size()); raw[l].dw_b = v;
      std::vector<float> cbb;
      std::vector<float> pwm;
      if (!take(D * C, v)) goto bad; raw[l].pw = v; { auto tr = transposed(v); lo[l].pwT = pk.add(tr.data(), tr.size()); auto f = frag(v, D, C); lo[l].pw_f = pk.add(f.data(), f.size()); pwm = v; if (C == 32) { auto h = frag_h3(v, D, C); lo[l].pw_h = pk.add(h.data(), h.size()); } }
      if (!take(D, v)) goto bad;     lo[l].pw_b = pk.add(v.data(), v.size()); cbb = v; raw[l].pw_b = v;
      lo[l].pjT = lo[l].pj_b = lo[l].pj_f = (size_t)-1;
      if (s.proj) {
         if (!take(D * C, v)) goto bad; raw[l].pj = v; { auto tr = transposed(v); lo[l].pjT = pk.add(tr.data(), tr.size()); auto f = frag(v, D, C); lo[l].pj_f = pk.add(f.data(), f.size()); if (C == 32) { auto h = frag_h3(v, D, C); lo[l].pj_h = pk.add(h.data(), h.size()); } }
         if (l == 0) { auto k1 = k1_pack(pwm, v, D, C); lo[l].pwj_k1 = pk.add(k1.data(), k1.size()); }
         if (!take(D, v)) goto bad;     lo[l].pj_b = pk.add(v.data(), v.size()); raw[l].pj_b = v;
         for (int o = 0; o < D; ++o) cbb[o] += v[o];
      }
      lo[l].cb_b = pk.add(cbb.data(), cbb.size());
      if (!take(3 * D * D, v)) goto bad; lo[l].qkv_w = pk.add(v.data(), v.size()); raw[l].qkv_w = v; { auto f = frag(v, 3 * D, D); lo[l].qkv_f = pk.add(f.data(), f.size()); if (D % 32 == 0) { auto h = frag_h3(v, 3 * D, D); lo[l].qkv_h = pk.add(h.data(), h.size()); } }
      if (!take(3 * D, v)) goto bad;     lo[l].qkv_b = pk.add(v.data(), v.size()); raw[l].qkv_b = v;
      if (!take(D * D, v)) goto bad;     lo[l].out_w = pk.add(v.data(), v.size()); raw[l].out_w = v; { auto f = frag(v, D, D); lo[l].out_f = pk.add(f.data(), f.size()); if (D % 32 == 0) { auto h = frag_h3(v, D, D); lo[l].out_h = pk.add(h.data(), h.size()); } }
      if (!take(D, v)) goto bad;         lo[l].out_b = pk.add(v.data(), v.size()); raw[l].out_b = v;
      if (!take(D, v)) goto bad;         lo[l].n1_w = pk.add(v.data(), v.size()); raw[l].n1_w = v;
      if (!take(D, v)) goto bad;         lo[l].n1_b = pk.add(v.data(), v.size()); raw[l].n1_b = v;
      if (!take(D * D, v)) goto bad;     lo[l].l1_w = pk.add(v.data(), v.size()); raw[l].l1_w = v; { auto f = frag(v, D, D); lo[l].l1_f = pk.add(f.data(), f.size()); if (D % 32 == 0) { auto h = frag_h3(v, D, D); lo[l].l1_h = pk.add(h.data(), h.size()); } }
      if (!take(D, v)) goto bad;         lo[l].l1_b = pk.add(v.data(), v.size()); raw[l].l1_b = v;
      if (!take(D * D, v)) goto bad;     lo[l].l2_w = pk.add(v.data(), v.size()); raw[l].l2_w = v; { auto f = frag(v, D, D); lo[l].l2_f = pk.add(f.data(), f.size()); if (D % 32 == 0) { auto h = frag_h3(v, D, D); lo[l].l2_h = pk.add(h.data(), h.size()); } }
      if (!take(D, v)) goto bad;         lo[l].l2_b = pk.add(v.data(), v.size()); raw[l].l2_b = v;
      if (!take(D, v)) goto bad;         lo[l].n2_w = pk.add(v.data(), v.size()); raw[l].n2_w = v;
      if (!take(D, v)) goto bad;         lo[l].n2_b = pk.add(v.data(), v.size()); raw[l].n2_b = v;
      {
         // strided 1x1 conv + BatchNorm1d (transformer.c:279-288, misc.c:221-258) folded:
         //   ((W z + b) - mean) / sqrt(var + eps) * gamma + beta  =  (W * s) z + ((b - mean) * s + beta)
         std::vector<float> cw, cb, g, be, mu, var;
         if (!take(D * D, cw) || !take(D, cb) || !take(D, g) || !take(D, be) || !take(D, mu) || !take(D, var)) goto bad;
         for (int o = 0; o < D; ++o) {
            const float sc = g[o] / sqrtf(var[o] + 1e-5f);
            for (int d = 0; d < D; ++d) cw[(size_t)o * D + d] *= sc;
            cb[o] = (cb[o] - mu[o]) * sc + be[o];
         }
         raw[l].cv_w = cw; raw[l].cv_b = cb;
         lo[l].cv_w = pk.add(cw.data(), cw.size());
         lo[l].cv_b = pk.add(cb.data(), cb.size());
         { auto f = frag(cw, D, D); lo[l].cv_f = pk.add(f.data(), f.size()); if (D % 32 == 0) { auto h = frag_h3(cw, D, D); lo[l].cv_h = pk.add(h.data(), h.size()); } }
      }
   }
   e->enc_h3_ok = h3_ok;
   if (h3_ok) {
      // ---- LDS images of k_enc_fused (enc_fused_layout.h): split-fp16 A fragments [hi 64 x 8 | lo 64 x 8] per (M tile, k block) with the k order
      // of an accumulator tile (enc_sigma), vectors in natural order; Q rows of the QKV weight and bias pre-scaled by log2(e) / sqrt(hd)
      // (the kernel's softmax is exp2(s - max): transformer.c:104-114, tensor.h:751-784)
      bool ok = true;
      auto put_frags = [&ok](unsigned char *dst, const std::vector<float> &W, int M, int K, bool sigma) {
         const int KB = K / 32;
         _Float16 *h = reinterpret_cast<_Float16 *>(dst);
         for (int mt = 0; mt < M / 16; ++mt)
            for (int kb = 0; kb < KB; ++kb)
               for (int l = 0; l < 64; ++l)
                  for (int el = 0; el < 8; ++el) {
                     const int q = l >> 4, k = sigma ? enc_sigma(kb, q, el) : 32 * kb + 8 * q + el;
                     const float v = W[(size_t)(16 * mt + (l & 15)) * K + k];
                     if (!(fabsf(v) < 60000.0f)) ok = false;
                     const _Float16 hi = (_Float16)v;
                     const size_t base = ((size_t)mt * KB + kb) * 1024;
                     h[base + l * 8 + el] = hi;
                     h[base + 512 + l * 8 + el] = (_Float16)(v - (float)hi);
                  }
      };
      auto put_vec = [](float *dst, const std::vector<float> &v) { memcpy(dst, v.data(), v.size() * sizeof(float)); };
      auto build_layer = [&](int l, unsigned char *fbase, float *vbase, auto L) {
         typedef decltype(L) LL;
         const RawLayer &r = raw[l];
         const int D = kLayers[l].d, C = kLayers[l].cin;
         if (C == 16) {                                        // layer 2: [pointwise | projection] stacked over K = 16 + 16, hardware k order
            std::vector<float> st((size_t)D * 32);
            for (int o = 0; o < D; ++o)
               for (int c = 0; c < 16; ++c) { st[(size_t)o * 32 + c] = r.pw[(size_t)o * 16 + c]; st[(size_t)o * 32 + 16 + c] = r.pj[(size_t)o * 16 + c]; }
            put_frags(fbase + LL::f_pw, st, D, 32, false);
         } else {
            put_frags(fbase + LL::f_pw, r.pw, D, 32, true);
            if (kLayers[l].proj) put_frags(fbase + LL::f_pj, r.pj, D, 32, true);
         }
         std::vector<float> qw = r.qkv_w, qb = r.qkv_b;
         const float sc = 1.4426950408889634f / sqrtf((float)(D / 2));
         for (int o = 0; o < D; ++o) { for (int c = 0; c < D; ++c) qw[(size_t)o * D + c] *= sc; qb[o] *= sc; }
         put_frags(fbase + LL::f_qkv, qw, 3 * D, D, true);
         put_frags(fbase + LL::f_out, r.out_w, D, D, true);
         put_frags(fbase + LL::f_l1, r.l1_w, D, D, true);
         put_frags(fbase + LL::f_l2, r.l2_w, D, D, true);
         // LayerNorm 2 feeds the strided conv only: its scale goes into the conv's columns, its shift into the conv's bias (the kernel normalises without them)
         std::vector<float> cvw = r.cv_w, cvb = r.cv_b;
         for (int o = 0; o < D; ++o) {
            double acc = 0.0;
            for (int c = 0; c < D; ++c) { acc += (double)r.cv_w[(size_t)o * D + c] * (double)r.n2_b[c]; cvw[(size_t)o * D + c] = r.cv_w[(size_t)o * D + c] * r.n2_w[c]; }
            cvb[o] += (float)acc;
         }
         put_frags(fbase + LL::f_cv, cvw, D, D, true);
         for (int t = 0; t < 5; ++t) for (int c = 0; c < C; ++c) vbase[LL::v_dw + t * C + c] = r.dw_w[(size_t)c * 5 + t];
         for (int c = 0; c < C; ++c) vbase[LL::v_dw + 5 * C + c] = r.dw_b[c];
         for (int o = 0; o < D; ++o) vbase[LL::v_cb_b + o] = r.pw_b[o] + (kLayers[l].proj ? r.pj_b[o] : 0.0f);
         put_vec(vbase + LL::v_qkv_b, qb);
         {  // a softmax row sums to 1, so the V bias passes through the attention unchanged: out_b' = out_b + Wo . bv (the kernel adds no V bias)
            std::vector<float> ob = r.out_b;
            for (int o = 0; o < D; ++o) {
               double acc = 0.0;
               for (int c = 0; c < D; ++c) acc += (double)r.out_w[(size_t)o * D + c] * (double)r.qkv_b[2 * D + c];
               ob[o] += (float)acc;
            }
            put_vec(vbase + LL::v_out_b, ob);
         }
         put_vec(vbase + LL::v_n1_w, r.n1_w); put_vec(vbase + LL::v_n1_b, r.n1_b);
         put_vec(vbase + LL::v_l1_b, r.l1_b);   put_vec(vbase + LL::v_l2_b, r.l2_b);
         put_vec(vbase + LL::v_n2_w, r.n2_w);   put_vec(vbase + LL::v_n2_b, r.n2_b); put_vec(vbase + LL::v_cv_b, cvb);
      };
      e->h_encA.assign(kEncA_Bytes, 0); e->h_encB.assign(kEncB_Bytes, 0);
      build_layer(1, e->h_encA.data() + kEncA_L2F, reinterpret_cast<float *>(e->h_encA.data() + kEncA_V2), EncL2());
      build_layer(2, e->h_encA.data() + kEncA_L3F, reinterpret_cast<float *>(e->h_encA.data() + kEncA_V3), EncL3());
      build_layer(3, e->h_encB.data() + kEncB_L4F, reinterpret_cast<float *>(e->h_encB.data() + kEncB_V4), EncL4());
      if (!ok) { e->h_encA.clear(); e->h_encB.clear(); }       // a pre-scaled Q row left fp16's range: the per-layer kernels serve
   }
   {
      // ---- LDS image of k_layer1_regs (enc_fused_layout.h: L1Layout) ----
      bool ok = true;
      const RawLayer &r = raw[0];
      const int C = kBins, D = 16;
      auto put_h = [&ok](_Float16 *hi, _Float16 *lo, float v) {
         if (!(fabsf(v) < 60000.0f)) ok = false;
         *hi = (_Float16)v;
         *lo = (_Float16)(v - (float)*hi);
      };
      e->h_l1img.assign(kL1ImgBytes, 0);
      unsigned char *img = e->h_l1img.data();
      for (int kb = 0; kb < 8; ++kb) {                       // conv block: k blocks 0..3 relu(dw(x)) . pointwise, 4..7 x . projection
         _Float16 *h = reinterpret_cast<_Float16 *>(img + L1Layout::f_conv + kb * kFragBytes);
         const std::vector<float> &W = kb < 4 ? r.pw : r.pj;
         for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 8; ++el)
               put_h(&h[l * 8 + el], &h[512 + l * 8 + el], W[(size_t)(l & 15) * C + l1_channel(kb & 3, l >> 4, el)]);
      }
      auto put_frag4 = [&](int off, auto W) {                 // W(m, k): K = 16 fragment, lane (q, m) holds k = 4 q + e
         _Float16 *h = reinterpret_cast<_Float16 *>(img + off);
         for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 4; ++el) {                  // block LH: per lane [lo x 4 | hi x 4]; block H0: [hi x 4 | 0 x 4]
               put_h(&h[l * 8 + 4 + el], &h[l * 8 + el], W(l & 15, 4 * (l >> 4) + el));
               h[512 + l * 8 + el] = h[l * 8 + 4 + el];
            }
      };
      put_frag4(L1Layout::f_tail, [&](int m, int k) { return k == 0 ? r.pw[(size_t)m * C + 128] : (k == 1 ? r.pj[(size_t)m * C + 128] : 0.0f); });
      const float sc = 1.4426950408889634f / sqrtf(8.0f);    // log2(e) / sqrt(hd): the kernel's softmax is exp2(s - max)
      put_frag4(L1Layout::f_qkv, [&](int m, int k) { return r.qkv_w[(size_t)m * D + k] * sc; });
      put_frag4(L1Layout::f_qkv + kFrag4Bytes, [&](int m, int k) { return r.qkv_w[(size_t)(D + m) * D + k]; });
      put_frag4(L1Layout::f_qkv + 2 * kFrag4Bytes, [&](int m, int k) { return r.qkv_w[(size_t)(2 * D + m) * D + k]; });
      put_frag4(L1Layout::f_out, [&](int m, int k) { return r.out_w[(size_t)m * D + k]; });
      put_frag4(L1Layout::f_l1, [&](int m, int k) { return r.l1_w[(size_t)m * D + k]; });
      put_frag4(L1Layout::f_l2, [&](int m, int k) { return r.l2_w[(size_t)m * D + k]; });
      put_frag4(L1Layout::f_cv, [&](int m, int k) { return r.cv_w[(size_t)m * D + k] * r.n2_w[k]; });      // LayerNorm 2's scale
      float *v = reinterpret_cast<float *>(img + L1Layout::f_end);
      auto put_taps = [&](float *d, int ch) {
         for (int t = 0; t < 4; ++t) d[t] = r.dw_w[(size_t)ch * 5 + t];
         d[4] = r.dw_w[(size_t)ch * 5 + 4]; d[5] = d[6] = r.dw_b[ch]; d[7] = 0.0f;
      };
      for (int kb = 0; kb < 4; ++kb)
         for (int q = 0; q < 4; ++q)
            for (int el = 0; el < 8; ++el) put_taps(v + L1Layout::v_taps + ((kb * 8 + el) * 4 + q) * 8, l1_channel(kb, q, el));
      put_taps(v + L1Layout::v_tail, 128);
      for (int o = 0; o < D; ++o) {
         v[L1Layout::v_cb_b + o] = r.pw_b[o] + r.pj_b[o];
         v[L1Layout::v_q_b + o] = r.qkv_b[o] * sc;
         v[L1Layout::v_k_b + o] = r.qkv_b[D + o];
         double ob = 0.0, cb = 0.0;
         for (int c = 0; c < D; ++c) { ob += (double)r.out_w[(size_t)o * D + c] * (double)r.qkv_b[2 * D + c]; cb += (double)r.cv_w[(size_t)o * D + c] * (double)r.n2_b[c]; }
         v[L1Layout::v_out_b + o] = r.out_b[o] + (float)ob;   // a softmax row sums to 1: the V bias passes through the attention unchanged
         v[L1Layout::v_n1_w + o] = r.n1_w[o]; v[L1Layout::v_n1_b + o] = r.n1_b[o];
         v[L1Layout::v_l1_b + o] = r.l1_b[o]; v[L1Layout::v_l2_b + o] = r.l2_b[o];
         v[L1Layout::v_n2_w + o] = r.n2_w[o]; v[L1Layout::v_n2_b + o] = r.n2_b[o];
         v[L1Layout::v_cv_b + o] = r.cv_b[o] + (float)cb;     // LayerNorm 2's shift
      }
      if (!ok) e->h_l1img.clear();                            // a weight outside fp16's range: k_layer_mfma's fp32 form serves
   }
   {
      if (!need(idx, 2 * 256 * 128) || !need(idx + 1, 2 * 256) || !need(idx + 2, 128) || !need(idx + 3, 2)) goto bad;
      std::vector<float> W, B, dw, db;
      copy_unaligned(W, ts[idx]); copy_unaligned(B, ts[idx + 1]); copy_unaligned(dw, ts[idx + 2]); copy_unaligned(db, ts[idx + 3]);
      for (float v : W) if (!(fabsf(v) < 3.0e4f)) e->lstm_h3_ok = false;      // also catches NaN / inf
      std::vector<float> WT(W.size());
      for (int l = 0; l < 2; ++l)
         for (int r = 0; r < 256; ++r)
            for (int k = 0; k < 128; ++k) WT[((size_t)l * 128 + k) * 256 + r] = W[((size_t)l * 256 + r) * 128 + k];
      const size_t o_w = pk.add(W.data(), W.size()), o_wT = pk.add(WT.data(), WT.size());
      const size_t o_b = pk.add(B.data(), B.size()), o_dw = pk.add(dw.data(), dw.size()), o_db = pk.add(db.data(), db.size());

      HIP_TRY(hipMalloc(&e->d_weights, pk.buf.size() * sizeof(float)), VADC_AMD_ENOMEM);
      HIP_TRY(upload(e->d_weights, pk.buf.data(), pk.buf.size() * sizeof(float)), VADC_AMD_EHIP);
      const float *base = e->d_weights;
      e->d_basis = base + off_basis;
      if (e->gemm_ok) { e->d_afrag = base + off_afrag; e->d_nyq = base + off_nyq; e->d_afrag2 = base + off_afrag2; e->d_nyq2 = base + off_nyq2; }
      for (int l = 0; l < 4; ++l) {
         LayerWeights w;
         w.dw_w = base + lo[l].dw_w; w.dw_b = base + lo[l].dw_b; w.pwT = base + lo[l].pwT; w.pw_b = base + lo[l].pw_b;
         w.pjT = kLayers[l].proj ? base + lo[l].pjT : nullptr; w.pj_b = kLayers[l].proj ? base + lo[l].pj_b : nullptr;
         w.qkv_w = base + lo[l].qkv_w; w.qkv_b = base + lo[l].qkv_b; w.out_w = base + lo[l].out_w; w.out_b = base + lo[l].out_b;
         w.n1_w = base + lo[l].n1_w; w.n1_b = base + lo[l].n1_b; w.l1_w = base + lo[l].l1_w; w.l1_b = base + lo[l].l1_b;
         w.l2_w = base + lo[l].l2_w; w.l2_b = base + lo[l].l2_b; w.n2_w = base + lo[l].n2_w; w.n2_b = base + lo[l].n2_b;
         w.cv_w = base + lo[l].cv_w; w.cv_b = base + lo[l].cv_b;
         LayerWeightsM &m = e->lwm[l];
         m.dw_w = w.dw_w; m.dw_b = w.dw_b; m.pw_f = base + lo[l].pw_f; m.pj_f = kLayers[l].proj ? base + lo[l].pj_f : nullptr;
         m.pwj_k1 = (l == 0) ? base + lo[l].pwj_k1 : nullptr;
         if (kLayers[l].d % 32 == 0 && e->enc_h3_ok) {
            m.qkv_h = reinterpret_cast<const _Float16 *>(base + lo[l].qkv_h); m.out_h = reinterpret_cast<const _Float16 *>(base + lo[l].out_h);
            m.l1_h = reinterpret_cast<const _Float16 *>(base + lo[l].l1_h);   m.l2_h = reinterpret_cast<const _Float16 *>(base + lo[l].l2_h);
            m.cv_h = reinterpret_cast<const _Float16 *>(base + lo[l].cv_h);
            m.pw_h = kLayers[l].cin == 32 ? reinterpret_cast<const _Float16 *>(base + lo[l].pw_h) : nullptr;
            m.pj_h = (kLayers[l].cin == 32 && kLayers[l].proj) ? reinterpret_cast<const _Float16 *>(base + lo[l].pj_h) : nullptr;
         } else m.qkv_h = m.out_h = m.l1_h = m.l2_h = m.cv_h = m.pw_h = m.pj_h = nullptr;
         m.cb_b = base + lo[l].cb_b; m.qkv_f = base + lo[l].qkv_f; m.qkv_b = w.qkv_b; m.out_f = base + lo[l].out_f; m.out_b = w.out_b;
         m.n1_w = w.n1_w; m.n1_b = w.n1_b; m.l1_f = base + lo[l].l1_f; m.l1_b = w.l1_b; m.l2_f = base + lo[l].l2_f; m.l2_b = w.l2_b;
         m.n2_w = w.n2_w; m.n2_b = w.n2_b; m.cv_f = base + lo[l].cv_f; m.cv_b = w.cv_b;
      }
      e->lstm.w = base + o_w; e->lstm.wT = base + o_wT; e->lstm.b = base + o_b; e->lstm.dec_w = base + o_dw; e->lstm.dec_b = base + o_db;
   }
   return VADC_AMD_OK;
bad:
   return fail(VADC_AMD_EWEIGHTS, "weights: tensor %d has an unexpected size", idx);
}

// Silero v4 / 16 kHz: 36-tensor container written by vadc_amd/onnx_weights.py (order documented there)
int vadc::build_weights_v4(vadc_amd_engine *e, const std::vector<HostTensor> &ts)
{
   Packer pk;
   std::vector<float> tmp, tmp2;
   int idx = 0;
   auto need = [&](int i, int n) { return ts[i].size == n; };
   if (!need(idx, kFilters * kFilterLen)) return fail(VADC_AMD_EWEIGHTS, "weights: bad forward_basis_buffer");
   copy_unaligned(tmp, ts[idx++]);
   tmp2.resize(tmp.size());
   for (int f = 0; f < kFilters; ++f)                     // same consumption order as v3.1 (k_frontend, PK form)
      for (int ii = 0; ii < 4; ++ii)
         for (int lp = 0; lp < 4; ++lp)
            for (int j = 0; j < 8; ++j)
               for (int b = 0; b < 2; ++b)
                  tmp2[(size_t)f * 256 + ii * 64 + lp * 16 + j * 2 + b] = tmp[(size_t)f * 256 + 64 * (3 - ii) + 8 * j + (2 * lp + b)];
   const size_t off_basis = pk.add(tmp2.data(), tmp2.size());
   pk.add(nullptr, 64);
   size_t off_afrag = 0, off_nyq = 0, off_afrag2 = 0, off_nyq2 = 0;
   e->gemm_ok = build_gemm_frontend(tmp, pk, off_afrag, off_nyq, off_afrag2, off_nyq2);
   auto frag = [](const std::vector<float> &W, int M, int K) {
      const int KKW = (K + 3) / 4;
      std::vector<float> f((size_t)(M / 16) * KKW * 64, 0.0f);
      for (int mt = 0; mt < M / 16; ++mt)
         for (int kk = 0; kk < KKW; ++kk)
            for (int l = 0; l < 64; ++l) {
               const int k = 4 * kk + (l >> 4);
               if (k < K) f[((size_t)mt * KKW + kk) * 64 + l] = W[(size_t)(16 * mt + (l & 15)) * K + k];
            }
      return f;
   };
   struct LOff { size_t dw_w, dw_b, pw_f, pj_f, cb_b, cv_f, cv_b, pwj_k1; } lo[4];
   struct RawV4 { std::vector<float> dw_w, dw_b, pw, pj, cb_b, cv_w, cv_b; } r0, rl[4];      // the stages' weights as they come, for the LDS images of k_layer1_regs_v4 (r0) and k_enc_fused_v4 (rl[1..3])
   for (int l = 0; l < 4; ++l) {
      const LayerShape &s = kLayersV4[l];
      const int D = s.d, C = s.cin;
      auto take = [&](int n, std::vector<float> &v) -> bool { if (!need(idx, n)) return false; copy_unaligned(v, ts[idx++]); return true; };
      std::vector<float> v, cbb;
      if (!take(C * 5, v)) goto bad; lo[l].dw_w = pk.add(v.data(), v.size()); if (l == 0) r0.dw_w = v; rl[l].dw_w = v;
      if (!take(C, v)) goto bad;     lo[l].dw_b = pk.add(v.data(), v.size()); if (l == 0) r0.dw_b = v; rl[l].dw_b = v;
      std::vector<float> pwm;
      if (!take(D * C, v)) goto bad; { auto f = frag(v, D, C); lo[l].pw_f = pk.add(f.data(), f.size()); pwm = v; rl[l].pw = v; }
      if (!take(D, v)) goto bad;     cbb = v;
      lo[l].pj_f = (size_t)-1;
      if (s.proj) {
         if (!take(D * C, v)) goto bad; { auto f = frag(v, D, C); lo[l].pj_f = pk.add(f.data(), f.size()); rl[l].pj = v; }
         if (l == 0) { auto k1 = k1_pack(pwm, v, D, C); lo[l].pwj_k1 = pk.add(k1.data(), k1.size()); r0.pw = pwm; r0.pj = v; }
         if (!take(D, v)) goto bad;
         for (int o = 0; o < D; ++o) cbb[o] += v[o];
      }
      lo[l].cb_b = pk.add(cbb.data(), cbb.size());
      if (l == 0) r0.cb_b = cbb;
      rl[l].cb_b = cbb;
      if (!take(D * D, v)) goto bad; { auto f = frag(v, D, D); lo[l].cv_f = pk.add(f.data(), f.size()); }   // BatchNorm folded by the exporter
      if (l == 0) r0.cv_w = v;
      rl[l].cv_w = v;
      if (!take(D, v)) goto bad;     lo[l].cv_b = pk.add(v.data(), v.size());
      if (l == 0) r0.cv_b = v;
      rl[l].cv_b = v;
   }
   if (kLayersV4[0].cin == 2 * kBins && kLayersV4[0].d == 16 && r0.pj.size() == (size_t)16 * 2 * kBins) {
      // ---- LDS image of k_layer1_regs_v4 (enc_fused_layout.h: L1V4Layout) ----
      bool ok = true;
      const int C = 2 * kBins, D = 16;
      auto put_h = [&ok](_Float16 *hi, _Float16 *lo, float v) {
         if (!(fabsf(v) < 60000.0f)) ok = false;
         *hi = (_Float16)v;
         *lo = (_Float16)(v - (float)*hi);
      };
      e->h_l1img.assign(kL1V4ImgBytes, 0);
      unsigned char *img = e->h_l1img.data();
      for (int f = 0; f < 16; ++f) {                         // fragments 0..7: relu(dw(.)) . pointwise of virtual k block vb = f, 8..15: (.) . projection
         _Float16 *h = reinterpret_cast<_Float16 *>(img + L1V4Layout::f_conv + f * kFragBytes);
         const std::vector<float> &W = f < 8 ? r0.pw : r0.pj;
         const int vb = f & 7, kb = vb >> 1, which = vb & 1;
         for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 8; ++el)
               put_h(&h[l * 8 + el], &h[512 + l * 8 + el], W[(size_t)(l & 15) * C + which * kBins + l1v4_channel(kb, l >> 4, el)]);
      }
      auto put_frag4 = [&](int off, auto W) {                 // W(m, k): K = 16 fragment, lane (q, m) holds k = 4 q + e: block LH [lo x 4 | hi x 4], block H0 [hi x 4 | 0 x 4]
         _Float16 *h = reinterpret_cast<_Float16 *>(img + off);
         for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 4; ++el) {
               put_h(&h[l * 8 + 4 + el], &h[l * 8 + el], W(l & 15, 4 * (l >> 4) + el));
               h[512 + l * 8 + el] = h[l * 8 + 4 + el];
            }
      };
      put_frag4(L1V4Layout::f_tail, [&](int m, int k) {
         return k == 0 ? r0.pw[(size_t)m * C + 128] : (k == 1 ? r0.pw[(size_t)m * C + kBins + 128] : (k == 2 ? r0.pj[(size_t)m * C + 128] : (k == 3 ? r0.pj[(size_t)m * C + kBins + 128] : 0.0f)));
      });
      put_frag4(L1V4Layout::f_cv, [&](int m, int k) { return r0.cv_w[(size_t)m * D + k]; });
      float *v = reinterpret_cast<float *>(img + L1V4Layout::f_end);
      auto put_taps = [&](float *d, int ch) {
         for (int t = 0; t < 4; ++t) d[t] = r0.dw_w[(size_t)ch * 5 + t];
         d[4] = r0.dw_w[(size_t)ch * 5 + 4]; d[5] = d[6] = r0.dw_b[ch]; d[7] = 0.0f;
      };
      for (int vb = 0; vb < 8; ++vb)
         for (int q = 0; q < 4; ++q)
            for (int el = 0; el < 8; ++el) put_taps(v + L1V4Layout::v_taps + ((vb * 8 + el) * 4 + q) * 8, (vb & 1) * kBins + l1v4_channel(vb >> 1, q, el));
      put_taps(v + L1V4Layout::v_tail, 128);
      put_taps(v + L1V4Layout::v_tail + 8, kBins + 128);
      for (int o = 0; o < D; ++o) { v[L1V4Layout::v_cb_b + o] = r0.cb_b[o]; v[L1V4Layout::v_cv_b + o] = r0.cv_b[o]; }
      if (!ok) e->h_l1img.clear();                            // a weight outside fp16's range: k_layer_mfma's fp32 form serves
   }
   if (kLayersV4[1].cin == 16 && kLayersV4[1].d == 32 && kLayersV4[2].cin == 32 && kLayersV4[2].d == 32 && !kLayersV4[2].proj && kLayersV4[3].cin == 32 && kLayersV4[3].d == 64) {
      // ---- LDS image of k_enc_fused_v4 (enc_fused_layout.h: EncV4LayerLayout): split-fp16 A fragments [hi 64 x 8 | lo 64 x 8] per (M tile, k block) ----
      bool ok = true;
      auto put_frags = [&ok](unsigned char *dst, const std::vector<float> &W, int M, int K, bool sigma) {
         _Float16 *h = reinterpret_cast<_Float16 *>(dst);
         const int KB = K / 32;
         for (int mt = 0; mt < M / 16; ++mt)
            for (int kb = 0; kb < KB; ++kb)
               for (int l = 0; l < 64; ++l)
                  for (int el = 0; el < 8; ++el) {
                     const int q = l >> 4, k = sigma ? enc_sigma(kb, q, el) : 32 * kb + 8 * q + el;
                     const float v = W[(size_t)(16 * mt + (l & 15)) * K + k];
                     if (!(fabsf(v) < 60000.0f)) ok = false;
                     const _Float16 hi = (_Float16)v;
                     const size_t base = ((size_t)mt * KB + kb) * 1024;
                     h[base + l * 8 + el] = hi;
                     h[base + 512 + l * 8 + el] = (_Float16)(v - (float)hi);
                  }
      };
      e->h_encv4.assign(kEncV4Bytes, 0);
      auto build = [&](int l, unsigned char *fbase, float *vbase, auto L) {
         typedef decltype(L) LL;
         const RawV4 &r = rl[l];
         const int D = kLayersV4[l].d, C = kLayersV4[l].cin;
         if (C == 16) {                                        // stage 2: [pointwise | projection] stacked over K = 16 + 16, hardware k order
            std::vector<float> st((size_t)D * 32);
            for (int o = 0; o < D; ++o)
               for (int c = 0; c < 16; ++c) { st[(size_t)o * 32 + c] = r.pw[(size_t)o * 16 + c]; st[(size_t)o * 32 + 16 + c] = r.pj[(size_t)o * 16 + c]; }
            put_frags(fbase + LL::f_pw, st, D, 32, false);
         } else {
            put_frags(fbase + LL::f_pw, r.pw, D, 32, true);
            if (kLayersV4[l].proj) put_frags(fbase + LL::f_pj, r.pj, D, 32, true);
         }
         put_frags(fbase + LL::f_cv, r.cv_w, D, D, true);
         for (int t = 0; t < 5; ++t) for (int c = 0; c < C; ++c) vbase[LL::v_dw + t * C + c] = r.dw_w[(size_t)c * 5 + t];
         for (int c = 0; c < C; ++c) vbase[LL::v_dw + 5 * C + c] = r.dw_b[c];
         for (int o = 0; o < D; ++o) { vbase[LL::v_cb_b + o] = r.cb_b[o]; vbase[LL::v_cv_b + o] = r.cv_b[o]; }
      };
      unsigned char *img = e->h_encv4.data();
      build(1, img + kEncV4_L2F, reinterpret_cast<float *>(img + kEncV4_V2), EncV4L2());
      build(2, img + kEncV4_L3F, reinterpret_cast<float *>(img + kEncV4_V3), EncV4L3());
      build(3, img + kEncV4_L4F, reinterpret_cast<float *>(img + kEncV4_V4), EncV4L4());
      if (!ok) e->h_encv4.clear();                           // a weight outside fp16's range: the per-stage fp32 kernels serve
   }
   {
      if (!need(idx, 2 * 256 * 128) || !need(idx + 1, 2 * 256) || !need(idx + 2, 64) || !need(idx + 3, 1) || !need(idx + 4, 7)) goto bad;
      std::vector<float> W, B, dw, db;
      copy_unaligned(W, ts[idx]); copy_unaligned(B, ts[idx + 1]); copy_unaligned(dw, ts[idx + 2]); copy_unaligned(db, ts[idx + 3]);
      for (float v : W) if (!(fabsf(v) < 3.0e4f)) e->lstm_h3_ok = false;
      dw.resize(128, 0.0f); db.resize(2, 0.0f);            // LstmWeights carries room for the v3.1 two-output decoder
      std::vector<float> WT(W.size());
      for (int l = 0; l < 2; ++l)
         for (int r = 0; r < 256; ++r)
            for (int k = 0; k < 128; ++k) WT[((size_t)l * 128 + k) * 256 + r] = W[((size_t)l * 256 + r) * 128 + k];
      const size_t o_w = pk.add(W.data(), W.size()), o_wT = pk.add(WT.data(), WT.size());
      const size_t o_b = pk.add(B.data(), B.size()), o_dw = pk.add(dw.data(), dw.size()), o_db = pk.add(db.data(), db.size());
      HIP_TRY(hipMalloc(&e->d_weights, pk.buf.size() * sizeof(float)), VADC_AMD_ENOMEM);
      HIP_TRY(upload(e->d_weights, pk.buf.data(), pk.buf.size() * sizeof(float)), VADC_AMD_EHIP);
      const float *base = e->d_weights;
      e->d_basis = base + off_basis;
      if (e->gemm_ok) { e->d_afrag = base + off_afrag; e->d_nyq = base + off_nyq; e->d_afrag2 = base + off_afrag2; e->d_nyq2 = base + off_nyq2; }
      for (int l = 0; l < 4; ++l) {
         LayerWeightsM &m = e->lwm[l];
         m = LayerWeightsM{};
         m.dw_w = base + lo[l].dw_w; m.dw_b = base + lo[l].dw_b; m.pw_f = base + lo[l].pw_f;
         m.pj_f = kLayersV4[l].proj ? base + lo[l].pj_f : nullptr;
         m.pwj_k1 = (l == 0) ? base + lo[l].pwj_k1 : nullptr;
         m.cb_b = base + lo[l].cb_b; m.cv_f = base + lo[l].cv_f; m.cv_b = base + lo[l].cv_b;
      }
      e->lstm.w = base + o_w; e->lstm.wT = base + o_wT; e->lstm.b = base + o_b; e->lstm.dec_w = base + o_dw; e->lstm.dec_b = base + o_db;
   }
   return VADC_AMD_OK;
bad:
   return fail(VADC_AMD_EWEIGHTS, "weights (v4): tensor %d has an unexpected size", idx);
}

// Silero v5 shapes: 13-tensor container in the order of the reference's C test (test.c:2045-2068 without its input / expected-output tensors):
// basis [258,1,256]; reparam_conv_{0..3} weight [Co,Ci,3] + bias; lstm weights [1,512,256] = [i,f,g,o][x(128) | h(128)] (utils.py:93-97), lstm biases
// [1,512]; decoder weight [1,128,1], bias [1]
int vadc::build_weights_v5(vadc_amd_engine *e, const std::vector<HostTensor> &ts)
{
   static const int co[4] = {128, 64, 64, 128}, ci[4] = {129, 128, 64, 64};
   static const int expect[13] = {258 * 256, 128 * 129 * 3, 128, 64 * 128 * 3, 64, 64 * 64 * 3, 64, 128 * 64 * 3, 128, 512 * 256, 512, 128, 1};
   for (int i = 0; i < 13; ++i) if (ts[i].size != expect[i]) return fail(VADC_AMD_EWEIGHTS, "weights (v5): tensor %d has an unexpected size", i);
   Packer pk;
   std::vector<float> v;
   // MFMA A-fragment order for v_mfma_f32_16x16x4_f32: [m-tile][k-step][lane], lane l holds A[16 mt + (l & 15)][4 kk + (l >> 4)]
   copy_unaligned(v, ts[0]);
   std::vector<float> sf((size_t)17 * 64 * 64, 0.0f);
   for (int mt = 0; mt < 17; ++mt)
      for (int kk = 0; kk < 64; ++kk)
         for (int l = 0; l < 64; ++l) {
            const int row = 16 * mt + (l & 15), k = 4 * kk + (l >> 4);
            if (row < 258) sf[((size_t)mt * 64 + kk) * 64 + l] = v[(size_t)row * 256 + k];
         }
   const size_t o_stft = pk.add(sf.data(), sf.size());
   size_t o_cf[4], o_cb[4];
   for (int c = 0; c < 4; ++c) {
      copy_unaligned(v, ts[1 + 2 * c]);
      const int CI = ci[c], CO = co[c], KT = (CI + 3) / 4, KKW = 3 * KT;
      std::vector<float> f((size_t)(CO / 16) * KKW * 64, 0.0f);         // K order (tap, input channel padded to a multiple of 4)
      for (int mt = 0; mt < CO / 16; ++mt)
         for (int tap = 0; tap < 3; ++tap)
            for (int kk = 0; kk < KT; ++kk)
               for (int l = 0; l < 64; ++l) {
                  const int o = 16 * mt + (l & 15), i = 4 * kk + (l >> 4);
                  if (i < CI) f[((size_t)mt * KKW + tap * KT + kk) * 64 + l] = v[((size_t)o * CI + i) * 3 + tap];
               }
      o_cf[c] = pk.add(f.data(), f.size());
      copy_unaligned(v, ts[2 + 2 * c]);
      o_cb[c] = pk.add(v.data(), v.size());
   }
   // GATE ROW ORDER.  The container has PyTorch's [gate i, f, g, o][unit] (utils.py:93-101).  Every kernel here works on rows p(g, u) = 16 (u / 4) + 4 (u % 4) + g instead
   // -- an MFMA row tile of 16 = four units x four gates, so that the (i, f, g, o) of a unit are the four accumulator registers of ONE lane and a tile's cells can be
   // updated as soon as ITS twelve MFMAs are done, under the next tile's (k_v5_lstm_h3).  W_ih, the bias and W_hh are permuted here, once; GX[item][512] is in that
   // order whichever encoder wrote it, and the unit-indexed state h, c [128] is what it always was.
   std::vector<float> W0, W, lb0;
   copy_unaligned(W0, ts[9]);
   copy_unaligned(lb0, ts[10]);
   W.resize(W0.size());
   std::vector<float> lbp(512);
   for (int g = 0; g < 4; ++g)
      for (int u = 0; u < 128; ++u) {
         const int p = 16 * (u / 4) + 4 * (u % 4) + g;
         memcpy(&W[(size_t)p * 256], &W0[(size_t)(g * 128 + u) * 256], 256 * sizeof(float));
         lbp[p] = lb0[g * 128 + u];
      }
   std::vector<float> wih((size_t)32 * 32 * 64), whh((size_t)512 * 128);
   for (int mt = 0; mt < 32; ++mt)
      for (int kk = 0; kk < 32; ++kk)
         for (int l = 0; l < 64; ++l) wih[((size_t)mt * 32 + kk) * 64 + l] = W[(size_t)(16 * mt + (l & 15)) * 256 + 4 * kk + (l >> 4)];
   for (int r = 0; r < 512; ++r) memcpy(&whh[(size_t)r * 128], &W[(size_t)r * 256 + 128], 128 * sizeof(float));
   const size_t o_wih = pk.add(wih.data(), wih.size()), o_whh = pk.add(whh.data(), whh.size());
   // split-fp16 A fragments of W_hh for v_mfma_f32_16x16x32_f16: [m-tile][k-block][lane][hi 8 | lo 8], lane l holds W[16 mt + (l & 15)][32 kb + 8 (l >> 4) + e] (permuted rows)
   bool h3_ok = true;
   std::vector<_Float16> wh((size_t)32 * 4 * 64 * 16);
   for (int mt = 0; mt < 32; ++mt)
      for (int kb = 0; kb < 4; ++kb)
         for (int l = 0; l < 64; ++l)
            for (int el = 0; el < 8; ++el) {
               const float x = whh[(size_t)(16 * mt + (l & 15)) * 128 + 32 * kb + 8 * (l >> 4) + el];
               if (!(fabsf(x) < 60000.0f)) h3_ok = false;
               const _Float16 hi = (_Float16)x;
               wh[(((size_t)mt * 4 + kb) * 64 + l) * 16 + el] = hi;
               wh[(((size_t)mt * 4 + kb) * 64 + l) * 16 + 8 + el] = (_Float16)(x - (float)hi);
            }
   std::vector<float> whf(wh.size() / 2);
   memcpy(whf.data(), wh.data(), wh.size() * sizeof(_Float16));
   const size_t o_whh_h = pk.add(whf.data(), whf.size());
   e->lstm_h3_ok = h3_ok;
   // k_v5_encoder_h3: every A operand of the encoder as split-fp16 fragments x 256, [m-tile][k-block][hi | lo][lane][8]; lane l holds row 16 mt + (l & 15),
   // k = 32 kb + 8 (l >> 4) + e.  The STFT rows are FOLDED (kernels_v5.hip): that needs re rows even and im rows odd about tap 128, tap 0 zero, the im row of bin 128
   // zero -- checked bit for bit on the loaded basis; a basis without them, or a weight x 256 outside fp16's range, leaves the fp32-MFMA encoder in charge.
   bool enc_ok = true;
   std::vector<_Float16> eh;
   size_t oh_stft = 0, oh_conv[4] = {0, 0, 0, 0}, oh_wih = 0;
   std::vector<float> wny(128, 0.0f);
   {
      std::vector<float> basis;
      copy_unaligned(basis, ts[0]);
      auto B = [&](int row, int n) { return basis[(size_t)row * 256 + n]; };
      for (int k = 0; k < kBins && enc_ok; ++k) {
         if (B(k, 0) != 0.0f || B(kBins + k, 0) != 0.0f || B(kBins + k, 128) != 0.0f) enc_ok = false;
         for (int n = 1; n < 128 && enc_ok; ++n)
            if (B(k, n) != B(k, 256 - n) || B(kBins + k, n) != -B(kBins + k, 256 - n)) enc_ok = false;
      }
      for (int n = 0; n < 256 && enc_ok; ++n) if (B(kBins + 128, n) != 0.0f) enc_ok = false;
      auto push = [&](float x) {                                     // one weight -> (hi, lo) appended 8 halves apart by the caller's loop structure
         const float sx = 256.0f * x;
         if (!(fabsf(sx) < 60000.0f)) enc_ok = false;
         const _Float16 hi = (_Float16)sx;
         return std::pair<_Float16, _Float16>(hi, (_Float16)(sx - (float)hi));
      };
      // value(mt, kb, lane, e) by stage
      auto pack = [&](int MT, int KB, auto &&value) {
         const size_t off = eh.size();
         eh.resize(off + (size_t)MT * KB * 2 * 512);
         for (int mt = 0; mt < MT; ++mt)
            for (int kb = 0; kb < KB; ++kb)
               for (int l = 0; l < 64; ++l)
                  for (int el = 0; el < 8; ++el) {
                     const auto hl = push(value(16 * mt + (l & 15), kb, l >> 4, el));
                     eh[off + ((size_t)(mt * KB + kb) * 2 + 0) * 512 + l * 8 + el] = hl.first;
                     eh[off + ((size_t)(mt * KB + kb) * 2 + 1) * 512 + l * 8 + el] = hl.second;
                  }
         return off;
      };
      if (enc_ok) {
         oh_stft = pack(16, 4, [&](int row, int kb, int kq, int el) {
            const int slot = 32 * kb + 8 * kq + el, n = slot + 1;
            if (row < 128) return slot == 127 ? 0.5f * B(row, 128) : B(row, n);
            return slot == 127 ? 0.0f : B(kBins + (row - 128), n);
         });
         for (int slot = 0; slot < 128; ++slot) wny[slot] = slot == 127 ? 0.5f * B(128, 128) : B(128, slot + 1);
         std::vector<float> cw[4];
         for (int c = 0; c < 4; ++c) copy_unaligned(cw[c], ts[1 + 2 * c]);
         auto CW = [&](int c, int o, int i, int tap) { return cw[c][((size_t)o * ci[c] + i) * 3 + tap]; };
         oh_conv[0] = pack(8, 13, [&](int o, int kb, int kq, int el) {
            if (kb < 12) return CW(0, o, 32 * (kb & 3) + 8 * kq + el, kb >> 2);
            return (kq < 3 && el == 0) ? CW(0, o, 128, kq) : 0.0f;           // channel 128 of tap kq
         });
         oh_conv[1] = pack(4, 12, [&](int o, int kb, int kq, int el) { return CW(1, o, 32 * (kb & 3) + 8 * kq + el, kb >> 2); });
         oh_conv[2] = pack(4, 4, [&](int o, int kb, int kq, int el) { return CW(2, o, 32 * (kb & 1) + 8 * kq + el, 1 + (kb >> 1)); });      // taps 1, 2 (tap 0 only ever meets padding)
         oh_conv[3] = pack(8, 2, [&](int o, int kb, int kq, int el) { return CW(3, o, 32 * kb + 8 * kq + el, 1); });                        // tap 1 (one input step)
         oh_wih = pack(32, 4, [&](int r, int kb, int kq, int el) { return W[(size_t)r * 256 + 32 * kb + 8 * kq + el]; });
      }
   }
   size_t o_eh = 0, o_wny = 0;
   if (enc_ok) {
      std::vector<float> ehf((eh.size() + 1) / 2);
      memcpy(ehf.data(), eh.data(), eh.size() * sizeof(_Float16));
      o_eh = pk.add(ehf.data(), ehf.size());
      o_wny = pk.add(wny.data(), wny.size());
   }
   e->v5_enc_h3_ok = enc_ok;
   const size_t o_lb = pk.add(lbp.data(), lbp.size());
   copy_unaligned(v, ts[11]); const size_t o_dw = pk.add(v.data(), v.size());
   copy_unaligned(v, ts[12]); const size_t o_db = pk.add(v.data(), v.size());
   HIP_TRY(hipMalloc(&e->d_weights, pk.buf.size() * sizeof(float)), VADC_AMD_ENOMEM);
   HIP_TRY(upload(e->d_weights, pk.buf.data(), pk.buf.size() * sizeof(float)), VADC_AMD_EHIP);
   const float *base = e->d_weights;
   e->v5.stft_f = base + o_stft;
   for (int c = 0; c < 4; ++c) { e->v5.conv_f[c] = base + o_cf[c]; e->v5.conv_b[c] = base + o_cb[c]; }
   e->v5.whh_h = h3_ok ? reinterpret_cast<const _Float16 *>(base + o_whh_h) : nullptr;
   e->v5.wih_f = base + o_wih; e->v5.whh = base + o_whh; e->v5.lstm_b = base + o_lb; e->v5.dec_w = base + o_dw; e->v5.dec_b = base + o_db;
   {
      const _Float16 *hb = enc_ok ? reinterpret_cast<const _Float16 *>(base + o_eh) : nullptr;
      e->v5.h_stft = enc_ok ? hb + oh_stft : nullptr;
      for (int c = 0; c < 4; ++c) e->v5.h_conv[c] = enc_ok ? hb + oh_conv[c] : nullptr;
      e->v5.h_wih = enc_ok ? hb + oh_wih : nullptr;
      e->v5.wny = enc_ok ? base + o_wny : nullptr;
   }
   return VADC_AMD_OK;
}
