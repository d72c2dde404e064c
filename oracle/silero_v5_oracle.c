/* silero_v5_oracle.c -- see silero_v5_oracle.h.  Plain fp32 loops; the parity target is a PyTorch module in any fp32 order. */
#include "silero_v5_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const unsigned char *p; size_t len, off; int ok; } rd5;
static int32_t rd5_i32(rd5 *r)
{
   int32_t v = 0;
   if (r->off + 4 > r->len) { r->ok = 0; return 0; }
   memcpy(&v, r->p + r->off, 4);
   r->off += 4;
   return v;
}

/* container format: tensor.h:97-102,201-253 */
so5_model *so5_model_from_bytes(const void *blob, size_t len)
{
   rd5 r = { (const unsigned char *)blob, len, 0, 1 };
   int32_t version = rd5_i32(&r), count = rd5_i32(&r);
   if (!r.ok || version != 1 || count != SO5_TENSORS) return NULL;
   for (int i = 0; i < count; ++i) {
      int32_t n = rd5_i32(&r);
      if (!r.ok || n <= 0 || r.off + (size_t)n > len) return NULL;
      r.off += (size_t)n;
   }
   static const int expect[SO5_TENSORS] = {258 * 256, 128 * 129 * 3, 128, 64 * 128 * 3, 64, 64 * 64 * 3, 64, 128 * 64 * 3, 128, 512 * 256, 512, 128, 1};
   size_t total = 0;
   for (int i = 0; i < count; ++i) total += (size_t)expect[i];
   so5_model *m = (so5_model *)calloc(1, sizeof(so5_model));
   m->storage = (float *)malloc(total * sizeof(float));
   const float *t[SO5_TENSORS];
   size_t used = 0;
   for (int i = 0; i < count; ++i) {
      int32_t ndim = rd5_i32(&r);
      if (!r.ok || ndim < 0 || ndim > 8) { so5_model_free(m); return NULL; }
      for (int d = 0; d < ndim; ++d) rd5_i32(&r);
      int32_t size = rd5_i32(&r), nbytes = rd5_i32(&r);
      if (!r.ok || size != expect[i] || nbytes != size * 4 || r.off + (size_t)nbytes > len) { so5_model_free(m); return NULL; }
      memcpy(m->storage + used, r.p + r.off, (size_t)nbytes);
      r.off += (size_t)nbytes;
      t[i] = m->storage + used; used += (size_t)size;
   }
   if (r.off != len) { so5_model_free(m); return NULL; }
   m->basis = t[0];
   for (int l = 0; l < 4; ++l) { m->conv_w[l] = t[1 + 2 * l]; m->conv_b[l] = t[2 + 2 * l]; }
   m->lstm_w = t[9]; m->lstm_b = t[10]; m->dec_w = t[11]; m->dec_b = t[12];
   return m;
}

void so5_model_free(so5_model *m)
{
   if (!m) return;
   free(m->storage);
   free(m);
}

/* MobileOneBlock (silero_vad.py:314-320): Conv1d(cin, cout, 3, stride, padding 1) + ReLU; test.c:2128-2160 */
static void conv_k3(const float *in, int cin, int t_in, const float *w, const float *b, int cout, int stride, float *out, int t_out)
{
   for (int o = 0; o < cout; ++o)
      for (int t = 0; t < t_out; ++t) {
         float acc = b[o];
         for (int c = 0; c < cin; ++c)
            for (int k = 0; k < 3; ++k) {
               const int q = t * stride + k - 1;
               if (q >= 0 && q < t_in) acc += w[((size_t)o * cin + c) * 3 + k] * in[c * t_in + q];
            }
         out[o * t_out + t] = acc > 0.0f ? acc : 0.0f;
      }
}

static float sigmoidf5(float v) { return 1.0f / (1.0f + expf(-v)); }

float so5_forward_chunk(const so5_model *m, const float *window, float *context, float *h, float *c, const so5_taps *taps)
{
   enum { IN = SO5_CONTEXT + SO5_WINDOW, PADDED = IN + SO5_PAD_RIGHT };
   float x[PADDED];
   memcpy(x, context, sizeof(float) * SO5_CONTEXT);                            /* vadc.c:124-139: [context | window] */
   memcpy(x + SO5_CONTEXT, window, sizeof(float) * SO5_WINDOW);
   for (int j = 0; j < SO5_PAD_RIGHT; ++j) x[IN + j] = x[IN - 2 - j];           /* F.pad(input, (0, 64), "reflect") :301 */
   memcpy(context, window + SO5_WINDOW - SO5_CONTEXT, sizeof(float) * SO5_CONTEXT);
   float mag[129 * SO5_FRAMES];
   for (int f = 0; f < 129; ++f)                                              /* conv1d stride 128 :306, magnitude :308-311 */
      for (int t = 0; t < SO5_FRAMES; ++t) {
         float re = 0.0f, im = 0.0f;
         for (int k = 0; k < 256; ++k) {
            re += m->basis[(size_t)f * 256 + k] * x[t * SO5_HOP + k];
            im += m->basis[(size_t)(129 + f) * 256 + k] * x[t * SO5_HOP + k];
         }
         mag[f * SO5_FRAMES + t] = sqrtf(re * re + im * im);
      }
   float c0[128 * 4], c1[64 * 2], c2[64], c3[128];
   conv_k3(mag, 129, 4, m->conv_w[0], m->conv_b[0], 128, 1, c0, 4);            /* encoder_shapes :345-350 */
   conv_k3(c0, 128, 4, m->conv_w[1], m->conv_b[1], 64, 2, c1, 2);
   conv_k3(c1, 64, 2, m->conv_w[2], m->conv_b[2], 64, 2, c2, 1);
   conv_k3(c2, 64, 1, m->conv_w[3], m->conv_b[3], 128, 1, c3, 1);
   if (taps) {
      if (taps->magnitude) memcpy(taps->magnitude, mag, sizeof(mag));
      if (taps->c0) memcpy(taps->c0, c0, sizeof(c0));
      if (taps->c1) memcpy(taps->c1, c1, sizeof(c1));
      if (taps->c2) memcpy(taps->c2, c2, sizeof(c2));
      if (taps->c3) memcpy(taps->c3, c3, sizeof(c3));
   }
   /* LSTM(128, one layer), one step: gates i,f,g,o; W = [512][x(128) | h(128)] (lstm.c:31-95) */
   float g[512];
   for (int r = 0; r < 512; ++r) {
      float acc = m->lstm_b[r];
      for (int k = 0; k < 128; ++k) acc += m->lstm_w[(size_t)r * 256 + k] * c3[k];
      for (int k = 0; k < 128; ++k) acc += m->lstm_w[(size_t)r * 256 + 128 + k] * h[k];
      g[r] = acc;
   }
   float d = m->dec_b[0];
   for (int u = 0; u < 128; ++u) {
      const float ig = sigmoidf5(g[u]), fg = sigmoidf5(g[128 + u]), gg = tanhf(g[256 + u]), og = sigmoidf5(g[384 + u]);
      c[u] = fg * c[u] + ig * gg;
      h[u] = og * tanhf(c[u]);
      d += m->dec_w[u] * (h[u] > 0.0f ? h[u] : 0.0f);                           /* ReLU -> conv 128->1 :335-338 */
   }
   return sigmoidf5(d);                                                         /* sigmoid, mean over the one step :412 */
}

void so5_forward_stream_f32(const so5_model *m, const float *x, int n_chunks, float *context, float *h, float *c, float *probs)
{
   for (int i = 0; i < n_chunks; ++i) probs[i] = so5_forward_chunk(m, x + (size_t)i * SO5_WINDOW, context, h, c, NULL);
}
