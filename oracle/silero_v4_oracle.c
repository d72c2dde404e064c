/* silero_v4_oracle.c -- see silero_v4_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "silero_v4_oracle.h"
#include "silero_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float *a4(size_t n) { return (float *)calloc(n ? n : 1, sizeof(float)); }

typedef struct rd4 { const unsigned char *p; size_t len, off; int ok; } rd4;
static int32_t rd4_i32(rd4 *r)
{
   int32_t v = 0;
   if (r->off + 4 > r->len) { r->ok = 0; return 0; }
   memcpy(&v, r->p + r->off, 4);
   r->off += 4;
   return v;
}

/* container format: tensor.h:97-102,201-253 */
so4_model *so4_model_from_bytes(const void *blob, size_t len)
{
   rd4 r = { (const unsigned char *)blob, len, 0, 1 };
   int32_t version = rd4_i32(&r), count = rd4_i32(&r);
   if (!r.ok || version != 1 || (count != SO4_TENSORS && count != SO4_TENSORS + 1)) return NULL;   /* + 1: the 8 kHz container's sample-rate marker */
   for (int i = 0; i < count; ++i) {
      int32_t n = rd4_i32(&r);
      if (!r.ok || n <= 0 || r.off + (size_t)n > len) return NULL;
      r.off += (size_t)n;
   }
   rd4 r2 = r;
   size_t total = 0;
   for (int i = 0; i < count; ++i) {
      int32_t ndim = rd4_i32(&r2);
      if (!r2.ok || ndim < 0 || ndim > 8) return NULL;
      for (int d = 0; d < ndim; ++d) rd4_i32(&r2);
      int32_t size = rd4_i32(&r2), nbytes = rd4_i32(&r2);
      if (!r2.ok || size <= 0 || nbytes != size * 4 || r2.off + (size_t)nbytes > len) return NULL;
      r2.off += (size_t)nbytes;
      total += (size_t)size;
   }
   if (r2.off != len) return NULL;
   so4_model *m = (so4_model *)calloc(1, sizeof(so4_model));
   m->storage = a4(total);
   const float *t[SO4_TENSORS + 1];
   int sizes[SO4_TENSORS + 1];
   size_t used = 0;
   for (int i = 0; i < count; ++i) {
      int32_t ndim = rd4_i32(&r);
      for (int d = 0; d < ndim; ++d) rd4_i32(&r);
      int32_t size = rd4_i32(&r), nbytes = rd4_i32(&r);
      memcpy(m->storage + used, r.p + r.off, (size_t)nbytes);
      r.off += (size_t)nbytes;
      t[i] = m->storage + used; sizes[i] = size; used += (size_t)size;
   }
   /* 8 kHz branch (silero_vad.py:178-181, `sr == 16000` else): the third strided conv has stride 1 */
   m->sample_rate = (count == SO4_TENSORS + 1) ? 8000 : 16000;
   static const int cin[4] = {258, 16, 32, 32}, cout[4] = {16, 32, 32, 64}, proj[4] = {1, 1, 0, 1};
   const int stride[4] = {2, 2, m->sample_rate == 8000 ? 1 : 2, 1};
   int idx = 0, ok = 1, t_in = SO4_FRAMES;
   m->basis = t[idx]; ok &= sizes[idx] == 258 * 256; idx++;
   for (int l = 0; l < 4; ++l) {
      so4_block *B = &m->block[l];
      B->cin = cin[l]; B->cout = cout[l]; B->stride = stride[l]; B->has_proj = proj[l];
      B->t_in = t_in; B->t_out = 1 + (t_in - 1) / stride[l]; t_in = B->t_out;
#define TAKE(field, expect) do { B->field = t[idx]; ok &= (sizes[idx] == (expect)); idx++; } while (0)
      TAKE(dw_w, cin[l] * 5); TAKE(dw_b, cin[l]);
      TAKE(pw_w, cout[l] * cin[l]); TAKE(pw_b, cout[l]);
      if (proj[l]) { TAKE(proj_w, cout[l] * cin[l]); TAKE(proj_b, cout[l]); }
      TAKE(conv_w, cout[l] * cout[l]); TAKE(conv_b, cout[l]);
#undef TAKE
   }
   m->lstm_w = t[idx]; ok &= sizes[idx] == 2 * 256 * 128; idx++;
   m->lstm_b = t[idx]; ok &= sizes[idx] == 2 * 256; idx++;
   m->dec_w = t[idx]; ok &= sizes[idx] == 64; idx++;
   m->dec_b = t[idx]; ok &= sizes[idx] == 1; idx++;
   ok &= sizes[idx] == 7; idx++;                       /* adaptive-normalization filter: so_adaptive_norm carries the constants */
   if (count == SO4_TENSORS + 1) { ok &= sizes[idx] == 1 && t[idx][0] == 8000.0f; idx++; }
   if (!ok || idx != count) { so4_model_free(m); return NULL; }
   return m;
}

void so4_model_free(so4_model *m)
{
   if (!m) return;
   free(m->storage);
   free(m);
}

/* ConvBlock (silero_vad.py:68-93) for any T >= 1: the v3.1 restatement so_dw_conv_k5 spells out the reference C edge
 * handling and needs T >= 5; v4's last block runs at T = 3. */
static void so4_conv_block(const float *in, const so4_block *B, int t, float *out)
{
   const int cin = B->cin, cout = B->cout;
   float *dw = a4((size_t)cin * t);
   for (int c = 0; c < cin; ++c)
      for (int i = 0; i < t; ++i) {
         float r = 0.0f;
         for (int j = 0; j < 5; ++j) {
            const int q = i + j - 2;
            if (q >= 0 && q < t) { float v = in[c * t + q] * B->dw_w[c * 5 + j]; r += v; }
         }
         r = B->dw_b[c] + r;
         dw[c * t + i] = r > 0.0f ? r : 0.0f;
      }
   so_conv_k1(dw, cin, t, B->pw_w, B->pw_b, cout, 1, out);
   if (B->has_proj) {
      float *pr = a4((size_t)cout * t);
      so_conv_k1(in, cin, t, B->proj_w, B->proj_b, cout, 1, pr);
      for (int i = 0; i < cout * t; ++i) out[i] += pr[i];
      free(pr);
   } else {
      for (int i = 0; i < cout * t; ++i) out[i] += in[i];
   }
   for (int i = 0; i < cout * t; ++i) out[i] = out[i] > 0.0f ? out[i] : 0.0f;
   free(dw);
}

/* silero_vad.py:191-236 for a window of n_samples (the class takes any length; the reference's onnxruntime path feeds 512 ... 1536 samples,
 * onnx_helpers.c:164-170).  frames = (n_samples + 2 * 96 - 256) / 64 + 1 (conv1d, stride 64, no padding: trailing samples that do not fill a
 * frame are ignored, as F.conv1d does); stage lengths 1 + (t - 1) / stride. */
float so4_forward_chunk_w(const so4_model *m, const float *samples, int n_samples, float *h, float *c, const so4_taps *taps)
{
   const int padded_len = n_samples + 2 * SO4_PAD, frames = (padded_len - 256) / 64 + 1;
   float *padded = a4((size_t)padded_len), *conv = a4((size_t)258 * frames), *x0 = a4((size_t)258 * frames);
   so_reflect_pad(samples, n_samples, SO4_PAD, SO4_PAD, padded);            /* :32,40 (to_pad = 96) */
   so_stft_conv(padded, padded_len, m->basis, conv);                        /* :45 */
   so_magnitude(conv, frames, x0);                                          /* :47-50, rows 0..128 */
   float *norm = x0 + (size_t)129 * frames;                                 /* rows 129..257: torch.cat([spect, normalized], 1) :212 */
   memcpy(norm, x0, sizeof(float) * 129 * frames);
   so_adaptive_norm(norm, 129, frames);                                     /* :57-66 */
   if (taps && taps->magnitude)  memcpy(taps->magnitude, x0, sizeof(float) * 129 * frames);
   if (taps && taps->normalized) memcpy(taps->normalized, norm, sizeof(float) * 129 * frames);
   float *cur = x0, *louts[4];
   int t_in = frames, t_outs[4];
   for (int l = 0; l < 4; ++l) {
      const so4_block *B = &m->block[l];
      const int t_out = 1 + (t_in - 1) / B->stride;
      float *cb = a4((size_t)B->cout * t_in);
      so4_conv_block(cur, B, t_in, cb);                                       /* :68-93 */
      louts[l] = a4((size_t)B->cout * t_out);
      so_conv_k1(cb, B->cout, t_in, B->conv_w, B->conv_b, B->cout, B->stride, louts[l]); /* :160-186, BN folded */
      for (int i = 0; i < B->cout * t_out; ++i) louts[l][i] = louts[l][i] > 0.0f ? louts[l][i] : 0.0f;
      free(cb);
      cur = louts[l];
      t_in = t_out; t_outs[l] = t_out;
   }
   const int steps = t_outs[3];
   if (taps && taps->l1) memcpy(taps->l1, louts[0], sizeof(float) * 16 * t_outs[0]);
   if (taps && taps->l2) memcpy(taps->l2, louts[1], sizeof(float) * 32 * t_outs[1]);
   if (taps && taps->l3) memcpy(taps->l3, louts[2], sizeof(float) * 32 * t_outs[2]);
   if (taps && taps->l4) memcpy(taps->l4, louts[3], sizeof(float) * 64 * t_outs[3]);
   float *seq = a4((size_t)steps * 64), *lout = a4((size_t)steps * 64);
   for (int t = 0; t < steps; ++t)
      for (int u = 0; u < 64; ++u) seq[t * 64 + u] = louts[3][u * steps + t];   /* permute [0,2,1] :215 */
   so_lstm_seq(seq, steps, m->lstm_w, m->lstm_b, 2, h, c, lout);             /* :217, :229-234 */
   if (taps && taps->lstm_out) memcpy(taps->lstm_out, lout, sizeof(float) * steps * 64);
   float acc = 0.0f;
   for (int t = 0; t < steps; ++t) {                                          /* ReLU -> conv 64->1 -> sigmoid :200-204, mean :222 */
      float d = 0.0f;
      for (int u = 0; u < 64; ++u) d += m->dec_w[u] * (lout[t * 64 + u] > 0.0f ? lout[t * 64 + u] : 0.0f);
      d += m->dec_b[0];
      acc += 1.0f / (1.0f + expf(-d));
   }
   for (int l = 0; l < 4; ++l) free(louts[l]);
   free(seq); free(lout);
   free(padded); free(conv); free(x0);
   return acc / (float)steps;
}

float so4_forward_chunk(const so4_model *m, const float *samples, float *h, float *c, const so4_taps *taps)
{
   return so4_forward_chunk_w(m, samples, 1536, h, c, taps);
}

void so4_forward_stream_f32_w(const so4_model *m, const float *x, int n_chunks, int window, float *h, float *c, float *probs)
{
   for (int i = 0; i < n_chunks; ++i) probs[i] = so4_forward_chunk_w(m, x + (size_t)i * window, window, h, c, NULL);
}

void so4_forward_stream_f32(const so4_model *m, const float *x, int n_chunks, float *h, float *c, float *probs)
{
   for (int i = 0; i < n_chunks; ++i) probs[i] = so4_forward_chunk(m, x + (size_t)i * 1536, h, c, NULL);
}

void so4_forward_stream_s16(const so4_model *m, const int16_t *pcm, int n_chunks, float *h, float *c, float *probs)
{
   float buf[1536];
   for (int i = 0; i < n_chunks; ++i) {
      for (int k = 0; k < 1536; ++k) buf[k] = (float)pcm[(size_t)i * 1536 + k] / 32768.0f;    /* vadc.c:883,898 */
      probs[i] = so4_forward_chunk(m, buf, h, c, NULL);
   }
}
