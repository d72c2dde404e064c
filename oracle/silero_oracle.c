/*
 * silero_oracle.c -- see silero_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain scalar C.  Where the reference uses AVX2 lanes, the lane structure is spelled out with
 * small arrays so that every individual fp32 rounding happens in the same place as in the
 * reference's MSVC /fp:precise build (no FMA contraction: compile with -ffp-contract=off).
 */
#include "silero_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */

static float *so_alloc(size_t n)
{
   float *p = (float *)calloc(n ? n : 1, sizeof(float));
   if (!p) { fprintf(stderr, "silero_oracle: out of memory\n"); abort(); }
   return p;
}

static void so_relu(float *x, int n)                       /* maths.h:94-103 */
{
   for (int i = 0; i < n; ++i) if (x[i] < 0.0f) x[i] = 0.0f;
}

static float so_sigmoid(float v)                           /* maths.h:9-12, 327-334 */
{
   return 1.0f / (1.0f + expf(-v));
}

static void so_transpose(const float *in, int rows, int cols, float *out) /* tensor.h:636-673 */
{
   for (int x = 0; x < cols; ++x)
      for (int y = 0; y < rows; ++y)
         out[x * rows + y] = in[y * cols + x];
}

/* ------------------------------------------------------------------------------------------ */
/* dot product with the reference's AVX2 association -- maths.h:123-158 (dotproduct_simd)      */
/*   16 taps per iteration: ab = a[0..7]*b[0..7], cd = a[8..15]*b[8..15];                      */
/*   hadd(ab,cd) = {ab0+ab1, ab2+ab3, cd0+cd1, cd2+cd3, ab4+ab5, ab6+ab7, cd4+cd5, cd6+cd7};   */
/*   r += hadd; after the loop result = ((((0+r0)+r1)+...)+r7); scalar tail appended in order. */
/* ------------------------------------------------------------------------------------------ */
float so_dot(const float *a, const float *b, int n)
{
   float r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
   int wide = (n / 16) * 16;
   for (int i = 0; i < wide; i += 16) {
      float ab[8], cd[8], h[8];
      for (int k = 0; k < 8; ++k) { ab[k] = a[i + k] * b[i + k]; cd[k] = a[i + 8 + k] * b[i + 8 + k]; }
      h[0] = ab[0] + ab[1]; h[1] = ab[2] + ab[3]; h[2] = cd[0] + cd[1]; h[3] = cd[2] + cd[3];
      h[4] = ab[4] + ab[5]; h[5] = ab[6] + ab[7]; h[6] = cd[4] + cd[5]; h[7] = cd[6] + cd[7];
      for (int k = 0; k < 8; ++k) r[k] = r[k] + h[k];
   }
   float result = 0.0f;
   result = result + r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7];
   for (int i = wide; i < n; ++i) {
      float v = a[i] * b[i];
      result += v;
   }
   return result;
}

/* y = x W^T (+ b added afterwards as a separate pass) -- tensor.h:675-723, maths.h:265-300 */
void so_linear(const float *in, int rows, int k, const float *w, const float *b, int n_out, float *out)
{
   for (int i = 0; i < rows; ++i)
      for (int o = 0; o < n_out; ++o)
         out[i * n_out + o] = so_dot(in + i * k, w + o * k, k);
   if (b)
      for (int i = 0; i < rows; ++i)
         for (int o = 0; o < n_out; ++o)
            out[i * n_out + o] += b[o];
}

/* ------------------------------------------------------------------------------------------ */
/* front end                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* tensor.h:912-958: left pad j <- in[pad_l - j], right pad j <- in[n - 2 - j] (no edge repeat) */
void so_reflect_pad(const float *in, int n, int pad_l, int pad_r, float *out)
{
   memcpy(out + pad_l, in, (size_t)n * sizeof(float));
   for (int j = 0; j < pad_l; ++j) out[j] = in[pad_l - j];
   for (int j = 0; j < pad_r; ++j) out[pad_l + n + j] = in[n - 2 - j];
}

/* stft.c:82-190 -- strided convolution with the [258,1,256] basis, hop 64, fixed reduction tree:
 *   tap t = 64*i + 8*j + l  (i: 64-tap group, j: 8-lane vector within the group, l: lane)
 *   per lane l and group i:   g_i[l] = ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7))   over j   (:141-160)
 *   across groups:            v[l]   = (g_0[l]+g_1[l]) + (g_2[l]+g_3[l])                  (:165-167)
 *   across lanes:             y      = ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7))              (:176-184)
 * every product and every sum individually rounded to fp32. */
void so_stft_conv(const float *padded, int padded_len, const float *basis, float *out)
{
   int frames = 1 + (padded_len - SO_FILTER_LEN) / SO_HOP;
   for (int f = 0; f < SO_FILTERS; ++f) {
      const float *k = basis + (size_t)f * SO_FILTER_LEN;
      for (int n = 0; n < frames; ++n) {
         const float *x = padded + n * SO_HOP;
         float g[4][8];
         for (int i = 0; i < 4; ++i) {
            for (int l = 0; l < 8; ++l) {
               float p[8];
               for (int j = 0; j < 8; ++j) p[j] = x[64 * i + 8 * j + l] * k[64 * i + 8 * j + l];
               float p01 = p[0] + p[1], p23 = p[2] + p[3], p45 = p[4] + p[5], p67 = p[6] + p[7];
               float p0123 = p01 + p23, p4567 = p45 + p67;
               g[i][l] = p0123 + p4567;
            }
         }
         float v[8];
         for (int l = 0; l < 8; ++l) {
            float g01 = g[0][l] + g[1][l];
            float g23 = g[2][l] + g[3][l];
            v[l] = g01 + g23;
         }
         float s01 = v[0] + v[1], s23 = v[2] + v[3], s45 = v[4] + v[5], s67 = v[6] + v[7];
         float s0123 = s01 + s23, s4567 = s45 + s67;
         out[f * frames + n] = s0123 + s4567;
      }
   }
}

/* stft.c:194-213: rows 0..128 real, rows 129..257 imaginary; sqrtf(re*re + im*im), unfused */
void so_magnitude(const float *conv, int frames, float *mag)
{
   int half = SO_BINS * frames;
   for (int i = 0; i < half; ++i) {
      float re = conv[i], im = conv[half + i];
      float re2 = re * re, im2 = im * im;
      mag[i] = sqrtf(re2 + im2);
   }
}

/* misc.c:1-124 */
void so_adaptive_norm(float *x, int channels, int frames)
{
   static const float filter[7] = {
      0.03663284704089164733887f, 0.11128076165914535522461f, 0.21674531698226928710938f,
      0.27068215608596801757812f, 0.21674531698226928710938f, 0.11128076165914535522461f,
      0.03663284704089164733887f };
   const float million = (float)(1024 * 1024);

   for (int i = 0; i < channels * frames; ++i) x[i] = log1pf(x[i] * million);          /* :40-46 */

   float *mean = so_alloc((size_t)frames);
   for (int t = 0; t < frames; ++t) {                                                   /* :50-63 */
      float s = 0.0f;
      for (int c = 0; c < channels; ++c) s += x[c * frames + t];
      mean[t] = s / channels;
   }
   float *padded = so_alloc((size_t)frames + 6);
   so_reflect_pad(mean, frames, 3, 3, padded);                                          /* :65 */
   float *smooth = so_alloc((size_t)frames);
   for (int t = 0; t < frames; ++t) {                      /* :67 -> conv.c:597-709 general path */
      float r = 0.0f;
      for (int i = 0; i < 7; ++i) { float v = padded[t + i] * filter[i]; r += v; }
      smooth[t] = 0.0f + r;
   }
   float ms = 0.0f;                                                                     /* :71-82 */
   for (int t = 0; t < frames; ++t) ms += smooth[t];
   float mean_mean = ms / frames;
   for (int i = 0; i < channels * frames; ++i) x[i] = x[i] - mean_mean;                 /* :84-96 */
   free(mean); free(padded); free(smooth);
}

/* ------------------------------------------------------------------------------------------ */
/* convolutions                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* short dot (n<16): maths.h:150-154 tail only, sequential */
static float so_dot_short(const float *a, const float *b, int n)
{
   float r = 0.0f;
   for (int i = 0; i < n; ++i) { float v = a[i] * b[i]; r += v; }
   return r;
}

/* depthwise cross-correlation k=5, zero pad 2 -- conv.c:17-53, 60-113 */
void so_dw_conv_k5(const float *in, int channels, int t, const float *w, const float *b, float *out)
{
   for (int c = 0; c < channels; ++c) {
      const float *a = in + c * t;
      const float *k = w + c * 5;
      float *o = out + c * t;
      float bias = b[c];
      o[0] = bias + so_dot_short(a, k + 2, 3);
      o[1] = bias + so_dot_short(a, k + 1, 4);
      for (int i = 0; i < t - 4; ++i) o[2 + i] = bias + so_dot_short(a + i, k, 5);
      o[t - 2] = bias + so_dot_short(a + t - 4, k, 4);
      o[t - 1] = bias + so_dot_short(a + t - 3, k, 3);
   }
}

/* k=1 convolution.
 * stride 1 -> "variant E" conv.c:532-589: premultiply into temp[t][c], two 8-lane accumulators over
 *   16-channel steps, three hadd levels, then scalar tail, then bias.
 * stride>1 -> general path conv.c:597-709: out[f][i] += x[c][i*stride]*w[f][c] with c outermost,
 *   bias added afterwards. */
void so_conv_k1(const float *in, int cin, int t, const float *w, const float *b, int cout, int stride, float *out)
{
   if (stride == 1) {
      float *temp = so_alloc((size_t)cin * t);
      for (int f = 0; f < cout; ++f) {
         const float *k = w + (size_t)f * cin;
         for (int c = 0; c < cin; ++c)
            for (int i = 0; i < t; ++i)
               temp[i * cin + c] = in[c * t + i] * k[c];
         for (int i = 0; i < t; ++i) {
            const float *row = temp + i * cin;
            float r1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int j = 0;
            for (; j < cin - 15; j += 16)
               for (int q = 0; q < 8; ++q) { r1[q] = r1[q] + row[j + q]; r2[q] = r2[q] + row[j + 8 + q]; }
            float h0 = r1[0] + r1[1], h1 = r1[2] + r1[3], h2 = r2[0] + r2[1], h3 = r2[2] + r2[3];
            float h4 = r1[4] + r1[5], h5 = r1[6] + r1[7], h6 = r2[4] + r2[5], h7 = r2[6] + r2[7];
            float g0 = h0 + h1, g1 = h2 + h3, g4 = h4 + h5, g5 = h6 + h7;
            float f0 = g0 + g1, f4 = g4 + g5;
            float acc = 0.0f;
            acc += f0 + f4;
            for (; j < cin; ++j) acc += row[j];
            acc += b ? b[f] : 0.0f;
            out[f * t + i] = acc;
         }
      }
      free(temp);
   } else {
      int t_out = 1 + (t - 1) / stride;
      for (int i = 0; i < cout * t_out; ++i) out[i] = 0.0f;
      for (int c = 0; c < cin; ++c)
         for (int f = 0; f < cout; ++f)
            for (int i = 0; i < t_out; ++i) {
               float r = 0.0f;
               float v = in[c * t + i * stride] * w[(size_t)f * cin + c];
               r += v;
               out[f * t_out + i] += r;
            }
      if (b)
         for (int f = 0; f < cout; ++f)
            for (int i = 0; i < t_out; ++i) out[f * t_out + i] += b[f];
   }
}

/* conv.c:761-814: relu(pw(relu(dw(x))) + (proj(x) | x)) */
void so_conv_block(const float *in, int cin, int t, int cout,
                   const float *dw_w, const float *dw_b, const float *pw_w, const float *pw_b,
                   const float *proj_w, const float *proj_b, float *out)
{
   float *dw = so_alloc((size_t)cin * t);
   so_dw_conv_k5(in, cin, t, dw_w, dw_b, dw);
   so_relu(dw, cin * t);
   so_conv_k1(dw, cin, t, pw_w, pw_b, cout, 1, out);
   if (proj_w) {
      float *pr = so_alloc((size_t)cout * t);
      so_conv_k1(in, cin, t, proj_w, proj_b, cout, 1, pr);
      for (int i = 0; i < cout * t; ++i) out[i] += pr[i];
      free(pr);
   } else {
      for (int i = 0; i < cout * t; ++i) out[i] += in[i];
   }
   so_relu(out, cout * t);
   free(dw);
}

/* ------------------------------------------------------------------------------------------ */
/* transformer block                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* tensor.h:751-784 */
void so_softmax_rows(float *x, int rows, int cols)
{
   float *e = so_alloc((size_t)cols);
   for (int r = 0; r < rows; ++r) {
      float *row = x + r * cols;
      float mx = row[0];
      for (int i = 0; i < cols; ++i) if (row[i] > mx) mx = row[i];
      float sum = 0.0f;
      for (int i = 0; i < cols; ++i) { e[i] = expf(row[i] - mx); sum += e[i]; }
      float inv = 1.0f / sum;
      for (int i = 0; i < cols; ++i) row[i] = e[i] * inv;
   }
   free(e);
}

/* misc.c:143-210 */
void so_layer_norm(const float *in, int rows, int features, const float *w, const float *b, float *out)
{
   const float eps = 1e-5f;
   float inv_features = 1.0f / features;
   for (int r = 0; r < rows; ++r) {
      const float *x = in + r * features;
      float sum = 0.0f;
      for (int i = 0; i < features; ++i) sum += x[i];
      float mean = sum * inv_features;
      float vs = 0.0f;
      for (int i = 0; i < features; ++i) { float d = x[i] - mean; vs += d * d; }
      float variance = vs * inv_features;
      float std_dev = sqrtf(variance + eps);
      float rstd = 1.0f / std_dev;
      float mean_rstd = mean * rstd;
      for (int i = 0; i < features; ++i)
         out[r * features + i] = (x[i] * rstd - mean_rstd) * w[i] + b[i];
   }
}

/* misc.c:221-258 */
void so_batch_norm(const float *in, int channels, int t, const float *mean, const float *var,
                   const float *w, const float *b, float *out)
{
   const float eps = 1e-5f;
   for (int c = 0; c < channels; ++c) {
      float std_dev = sqrtf(var[c] + eps);
      for (int i = 0; i < t; ++i) {
         float nv = (in[c * t + i] - mean[c]) / std_dev;
         out[c * t + i] = nv * w[c] + b[c];
      }
   }
}

/* transformer.c:13-153.  QKV = x W^T + b; heads are contiguous halves of each of Q,K,V (:72-99);
 * a = softmax_j((k_i . q_j) * 1/sqrt(hd)) -- K Q^T, not Q K^T (:104-105,114-120);
 * attn_i = sum_j a_ij v_j (:128-129); concat heads; out projection (:146). */
void so_attention(const float *in, int t, int d, const float *qkv_w, const float *qkv_b,
                  const float *out_w, const float *out_b, float *out)
{
   const int heads = 2;
   int hd = d / heads;
   float *qkv = so_alloc((size_t)t * 3 * d);
   so_linear(in, t, d, qkv_w, qkv_b, 3 * d, qkv);
   float *cat = so_alloc((size_t)t * d);
   float *q = so_alloc((size_t)t * hd), *k = so_alloc((size_t)t * hd), *vT = so_alloc((size_t)hd * t);
   float *a = so_alloc((size_t)t * t), *o = so_alloc((size_t)t * hd);
   const float scale = 1.0f / sqrtf((float)hd);
   for (int h = 0; h < heads; ++h) {
      for (int i = 0; i < t; ++i)
         for (int e = 0; e < hd; ++e) {
            q[i * hd + e]  = qkv[i * 3 * d + 0 * d + h * hd + e];
            k[i * hd + e]  = qkv[i * 3 * d + 1 * d + h * hd + e];
            vT[e * t + i]  = qkv[i * 3 * d + 2 * d + h * hd + e];
         }
      so_linear(k, t, hd, q, NULL, t, a);               /* a[i][j] = k_i . q_j */
      for (int i = 0; i < t * t; ++i) a[i] *= scale;
      so_softmax_rows(a, t, t);
      so_linear(a, t, t, vT, NULL, hd, o);              /* o[i][e] = a[i,:] . v[:,e] */
      for (int i = 0; i < t; ++i)
         for (int e = 0; e < hd; ++e) cat[i * d + h * hd + e] = o[i * hd + e];
   }
   so_linear(cat, t, d, out_w, out_b, d, out);
   free(qkv); free(cat); free(q); free(k); free(vT); free(a); free(o);
}

/* transformer.c:160-234: x=[T,D]; x += attn(x); y = LN1(x); y += lin2(relu(lin1(y))); out = LN2(y)^T */
void so_transformer_block(const float *in, int d, int t, const so_layer *L, float *out)
{
   float *x = so_alloc((size_t)t * d), *att = so_alloc((size_t)t * d), *n1 = so_alloc((size_t)t * d);
   float *f1 = so_alloc((size_t)t * d), *f2 = so_alloc((size_t)t * d), *n2 = so_alloc((size_t)t * d);
   so_transpose(in, d, t, x);
   so_attention(x, t, d, L->qkv_w, L->qkv_b, L->out_w, L->out_b, att);
   for (int i = 0; i < t * d; ++i) x[i] += att[i];
   so_layer_norm(x, t, d, L->n1_w, L->n1_b, n1);
   so_linear(n1, t, d, L->l1_w, L->l1_b, d, f1);
   so_relu(f1, t * d);
   so_linear(f1, t, d, L->l2_w, L->l2_b, d, f2);
   for (int i = 0; i < t * d; ++i) n1[i] += f2[i];
   so_layer_norm(n1, t, d, L->n2_w, L->n2_b, n2);
   so_transpose(n2, t, d, out);
   free(x); free(att); free(n1); free(f1); free(f2); free(n2);
}

/* transformer.c:237-295: conv_block -> transformer_block -> conv k=1 stride s + bias -> BN -> ReLU */
void so_transformer_layer(const float *in, const so_layer *L, int t_in, float *out)
{
   int d = L->cout;
   int t_out = 1 + (t_in - 1) / L->stride;
   float *cb = so_alloc((size_t)d * t_in), *tb = so_alloc((size_t)d * t_in), *cv = so_alloc((size_t)d * t_out);
   so_conv_block(in, L->cin, t_in, d, L->dw_w, L->dw_b, L->pw_w, L->pw_b,
                 L->has_proj ? L->proj_w : NULL, L->has_proj ? L->proj_b : NULL, cb);
   so_transformer_block(cb, d, t_in, L, tb);
   so_conv_k1(tb, d, t_in, L->conv_w, L->conv_b, d, L->stride, cv);
   so_batch_norm(cv, d, t_out, L->bn_mean, L->bn_var, L->bn_w, L->bn_b, out);
   so_relu(out, d * t_out);
   free(cb); free(tb); free(cv);
}

/* ------------------------------------------------------------------------------------------ */
/* LSTM + decoder                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* lstm.c:31-95 (cell), :100-149 (layer stack), :156-218 (sequence).  W[l] is [4H, 2H] over the
 * concatenation [x ; h_prev]; fused bias added after the matvec; gates i,f,g,o;
 * c = f*c_prev + i*g (unfused); h = tanh(c) * o. */
void so_lstm_seq(const float *x, int steps, const float *w, const float *b, int layers,
                 float *h, float *c, float *out)
{
   enum { H = SO_HIDDEN };
   float xh[2 * H], gates[4 * H];
   for (int s = 0; s < steps; ++s) {
      const float *input = x + s * H;
      for (int l = 0; l < layers; ++l) {
         const float *wl = w + (size_t)l * (4 * H) * (2 * H);
         const float *bl = b + l * 4 * H;
         float *hl = h + l * H, *cl = c + l * H;
         memcpy(xh, input, H * sizeof(float));
         memcpy(xh + H, hl, H * sizeof(float));
         for (int g = 0; g < 4 * H; ++g) gates[g] = so_dot(xh, wl + (size_t)g * 2 * H, 2 * H);
         for (int g = 0; g < 4 * H; ++g) gates[g] += bl[g];
         for (int j = 0; j < H; ++j) {
            float ig = so_sigmoid(gates[j]);
            float fg = so_sigmoid(gates[H + j]);
            float gg = tanhf(gates[2 * H + j]);
            float og = so_sigmoid(gates[3 * H + j]);
            float fc = fg * cl[j];
            float igg = ig * gg;
            float cn = fc + igg;
            float hn = tanhf(cn);
            cl[j] = cn;
            hl[j] = hn * og;
         }
         input = hl;
      }
      memcpy(out + s * H, h + (layers - 1) * H, H * sizeof(float));
   }
}

/* silero_v3.c:231-303 + maths.h:352-400: relu -> out[f][t] = sum_c w[f][c]*x[c][t] (c outermost,
 * accumulating into a zeroed row) + bias -> mean over t (sequential, / (float)t) -> sigmoid */
void so_decoder(const float *in, int channels, int t, const float *w, const float *b, int n_out, float *out)
{
   float *r = so_alloc((size_t)channels * t);
   memcpy(r, in, (size_t)channels * t * sizeof(float));
   so_relu(r, channels * t);
   float *row = so_alloc((size_t)t);
   for (int f = 0; f < n_out; ++f) {
      for (int i = 0; i < t; ++i) row[i] = 0.0f;
      for (int c = 0; c < channels; ++c)
         for (int i = 0; i < t; ++i) row[i] += w[f * channels + c] * r[c * t + i];
      for (int i = 0; i < t; ++i) row[i] += b ? b[f] : 0.0f;
      float s = 0.0f;
      for (int i = 0; i < t; ++i) s += row[i];
      float mean = s / (float)t;
      out[f] = 1.0f / (1.0f + expf(-mean));
   }
   free(r); free(row);
}

/* ------------------------------------------------------------------------------------------ */
/* whole path -- silero_v3.c:72-215                                                            */
/* ------------------------------------------------------------------------------------------ */
void so_forward_chunk(const so_model *m, const float *samples, float *h, float *c, float out[2], const so_taps *taps)
{
   float *padded = so_alloc(SO_PADDED);
   float *conv = so_alloc((size_t)SO_FILTERS * SO_FRAMES);
   float *x0 = so_alloc((size_t)SO_BINS * SO_FRAMES);
   so_reflect_pad(samples, SO_CHUNK_SAMPLES, SO_PAD, SO_PAD, padded);
   so_stft_conv(padded, SO_PADDED, m->basis, conv);
   so_magnitude(conv, SO_FRAMES, x0);
   if (taps && taps->padded)    memcpy(taps->padded, padded, SO_PADDED * sizeof(float));
   if (taps && taps->stft_conv) memcpy(taps->stft_conv, conv, sizeof(float) * SO_FILTERS * SO_FRAMES);
   if (taps && taps->magnitude) memcpy(taps->magnitude, x0, sizeof(float) * SO_BINS * SO_FRAMES);
   so_adaptive_norm(x0, SO_BINS, SO_FRAMES);
   if (taps && taps->normalized) memcpy(taps->normalized, x0, sizeof(float) * SO_BINS * SO_FRAMES);

   float *cur = x0;
   int t = SO_FRAMES;
   float *louts[SO_N_LAYERS];
   for (int l = 0; l < SO_N_LAYERS; ++l) {
      const so_layer *L = &m->layer[l];
      int t_out = 1 + (t - 1) / L->stride;
      louts[l] = so_alloc((size_t)L->cout * t_out);
      so_transformer_layer(cur, L, t, louts[l]);
      cur = louts[l];
      t = t_out;
   }
   if (taps && taps->l1) memcpy(taps->l1, louts[0], sizeof(float) * 16 * 13);
   if (taps && taps->l2) memcpy(taps->l2, louts[1], sizeof(float) * 32 * 7);
   if (taps && taps->l3) memcpy(taps->l3, louts[2], sizeof(float) * 32 * 7);
   if (taps && taps->l4) memcpy(taps->l4, louts[3], sizeof(float) * 64 * 7);

   float *seq = so_alloc((size_t)t * SO_HIDDEN), *lout = so_alloc((size_t)t * SO_HIDDEN);
   float *dec_in = so_alloc((size_t)t * SO_HIDDEN);
   so_transpose(cur, SO_HIDDEN, t, seq);                                   /* [64,7] -> [7,64]  :115 */
   so_lstm_seq(seq, t, m->lstm_w, m->lstm_b, SO_LSTM_LAYERS, h, c, lout);  /* :169-179 */
   if (taps && taps->lstm_out) memcpy(taps->lstm_out, lout, sizeof(float) * t * SO_HIDDEN);
   so_transpose(lout, t, SO_HIDDEN, dec_in);                               /* :176 */
   so_decoder(dec_in, SO_HIDDEN, t, m->dec_w, m->dec_b, 2, out);           /* :192 */

   for (int l = 0; l < SO_N_LAYERS; ++l) free(louts[l]);
   free(padded); free(conv); free(x0); free(seq); free(lout); free(dec_in);
}

void so_forward_stream_f32(const so_model *m, const float *samples, int n_chunks, float *h, float *c, float *probs)
{
   for (int i = 0; i < n_chunks; ++i)
      so_forward_chunk(m, samples + (size_t)i * SO_CHUNK_SAMPLES, h, c, probs + 2 * i, NULL);
}

void so_forward_stream_s16(const so_model *m, const int16_t *pcm, int n_chunks, float *h, float *c, float *probs)
{
   float buf[SO_CHUNK_SAMPLES];
   for (int i = 0; i < n_chunks; ++i) {
      for (int k = 0; k < SO_CHUNK_SAMPLES; ++k) {                          /* vadc.c:873-900 */
         float v = (float)pcm[(size_t)i * SO_CHUNK_SAMPLES + k];
         buf[k] = v / 32768.0f;
      }
      so_forward_chunk(m, buf, h, c, probs + 2 * i, NULL);
   }
}

/* ------------------------------------------------------------------------------------------ */
/* weights container -- tensor.h:97-102,201-253 (format), :114-191 (positional wiring)         */
/* ------------------------------------------------------------------------------------------ */
typedef struct so_rd { const unsigned char *p; size_t len, off; int ok; } so_rd;

static int32_t so_rd_i32(so_rd *r)
{
   int32_t v = 0;
   if (r->off + 4 > r->len) { r->ok = 0; return 0; }
   memcpy(&v, r->p + r->off, 4);
   r->off += 4;
   return v;
}

so_model *so_model_from_bytes(const void *blob, size_t len)
{
   so_rd r = { (const unsigned char *)blob, len, 0, 1 };
   int32_t version = so_rd_i32(&r), count = so_rd_i32(&r);
   if (!r.ok || version != 1 || count != 99) return NULL;
   for (int i = 0; i < count; ++i) {
      int32_t n = so_rd_i32(&r);
      if (!r.ok || n <= 0 || r.off + (size_t)n > len) return NULL;
      r.off += (size_t)n;
   }
   /* first pass: total floats */
   so_rd r2 = r;
   size_t total = 0;
   for (int i = 0; i < count; ++i) {
      int32_t ndim = so_rd_i32(&r2);
      if (!r2.ok || ndim < 0 || ndim > 8) return NULL;
      for (int d = 0; d < ndim; ++d) so_rd_i32(&r2);
      int32_t size = so_rd_i32(&r2), nbytes = so_rd_i32(&r2);
      if (!r2.ok || size <= 0 || nbytes != size * 4 || r2.off + (size_t)nbytes > len) return NULL;
      r2.off += (size_t)nbytes;
      total += (size_t)size;
   }
   if (r2.off != len) return NULL;

   so_model *m = (so_model *)calloc(1, sizeof(so_model));
   m->storage = so_alloc(total);
   m->tensor_count = count;
   const float *t[99];
   int sizes[99];
   size_t used = 0;
   for (int i = 0; i < count; ++i) {
      int32_t ndim = so_rd_i32(&r);
      for (int d = 0; d < ndim; ++d) so_rd_i32(&r);
      int32_t size = so_rd_i32(&r), nbytes = so_rd_i32(&r);
      memcpy(m->storage + used, r.p + r.off, (size_t)nbytes);
      r.off += (size_t)nbytes;
      t[i] = m->storage + used;
      sizes[i] = size;
      used += (size_t)size;
   }

   static const int cin[4]    = {129, 16, 32, 32};
   static const int cout[4]   = { 16, 32, 32, 64};
   static const int stride[4] = {  2,  2,  1,  1};   /* tensor.h:158-161 */
   static const int proj[4]   = {  1,  1,  0,  1};   /* tensor.h:164-167 */
   int idx = 0, t_in = SO_FRAMES, ok = 1;
   m->basis = t[idx]; ok &= sizes[idx] == SO_FILTERS * SO_FILTER_LEN; idx++;
   for (int l = 0; l < 4; ++l) {
      so_layer *L = &m->layer[l];
      L->cin = cin[l]; L->cout = cout[l]; L->stride = stride[l]; L->has_proj = proj[l];
      L->t_in = t_in; L->t_out = 1 + (t_in - 1) / stride[l]; t_in = L->t_out;
      int D = cout[l];
#define TAKE(field, expect) do { L->field = t[idx]; ok &= (sizes[idx] == (expect)); idx++; } while (0)
      TAKE(dw_w, cin[l] * 5);  TAKE(dw_b, cin[l]);
      TAKE(pw_w, D * cin[l]);  TAKE(pw_b, D);
      if (proj[l]) { TAKE(proj_w, D * cin[l]); TAKE(proj_b, D); }
      TAKE(qkv_w, 3 * D * D);  TAKE(qkv_b, 3 * D);
      TAKE(out_w, D * D);      TAKE(out_b, D);
      TAKE(n1_w, D);           TAKE(n1_b, D);
      TAKE(l1_w, D * D);       TAKE(l1_b, D);
      TAKE(l2_w, D * D);       TAKE(l2_b, D);
      TAKE(n2_w, D);           TAKE(n2_b, D);
      TAKE(conv_w, D * D);     TAKE(conv_b, D);
      TAKE(bn_w, D);           TAKE(bn_b, D);
      TAKE(bn_mean, D);        TAKE(bn_var, D);
#undef TAKE
   }
   m->lstm_w = t[idx]; ok &= sizes[idx] == 2 * 256 * 128; idx++;
   m->lstm_b = t[idx]; ok &= sizes[idx] == 2 * 256; idx++;
   m->dec_w  = t[idx]; ok &= sizes[idx] == 2 * 64; idx++;
   m->dec_b  = t[idx]; ok &= sizes[idx] == 2; idx++;
   if (!ok || idx != 99) { so_model_free(m); return NULL; }
   return m;
}

so_model *so_model_from_file(const char *path)
{
   FILE *f = fopen(path, "rb");
   if (!f) return NULL;
   fseek(f, 0, SEEK_END);
   long n = ftell(f);
   fseek(f, 0, SEEK_SET);
   unsigned char *buf = (unsigned char *)malloc((size_t)n);
   so_model *m = NULL;
   if (buf && fread(buf, 1, (size_t)n, f) == (size_t)n) m = so_model_from_bytes(buf, (size_t)n);
   free(buf);
   fclose(f);
   return m;
}

void so_model_free(so_model *m)
{
   if (!m) return;
   free(m->storage);
   free(m);
}

/* ------------------------------------------------------------------------------------------ */
/* segmenter -- vadc.c:165-221 (feed_probability), :262-299 (combine_or_emit), :223-260 (emit), */
/* :756-768 (ms -> chunks), :1005-1027 (end-of-stream flush)                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct so_seg { int start, end, valid; } so_seg;

static int so_emit(so_seg s, const so_seg_params *p, float *sec, int *chk, int n, int max)
{
   if (n < max) {
      float pad_s = p->speech_pad_ms / 1000.0f;
      float end_p = (s.end * p->seconds_per_chunk) + pad_s;
      float start_p = (s.start * p->seconds_per_chunk) - pad_s;
      if (start_p < 0.0f) start_p = 0.0f;
      if (sec) { sec[2 * n] = start_p; sec[2 * n + 1] = end_p; }
      if (chk) { chk[2 * n] = s.start; chk[2 * n + 1] = s.end; }
   }
   return n + 1;
}

static so_seg so_combine(so_seg buffered, so_seg cur, const so_seg_params *p, float *sec, int *chk, int *n, int max)
{
   float pad_s = p->speech_pad_ms / 1000.0f;
   float cur_start_p = (cur.start * p->seconds_per_chunk) - pad_s;
   if (cur_start_p < 0.0f) cur_start_p = 0.0f;
   if (buffered.valid) {
      float buf_end_p = (buffered.end * p->seconds_per_chunk) + pad_s;
      if (buf_end_p >= cur_start_p) {
         buffered.end = cur.end;
         return buffered;
      }
      *n = so_emit(buffered, p, sec, chk, *n, max);
   }
   return cur;
}

int so_segments(const float *probs, int n_chunks, int total_samples, const so_seg_params *p,
                float *out_seconds, int *out_chunks, int max_segments)
{
   (void)total_samples;
   const int window = SO_CHUNK_SAMPLES;
   float chunk_ms = window / (float)16000 * 1000.0f;
   int min_speech = (int)(p->min_speech_ms / chunk_ms + 0.5f);
   if (min_speech < 1) min_speech = 1;
   int min_silence = (int)(p->min_silence_ms / chunk_ms + 0.5f);
   if (min_silence < 1) min_silence = 1;

   int temp_end = 0, cur_start = 0, triggered = 0, n = 0;
   so_seg buffered = {0, 0, 0};
   int g = 0;
   for (; g < n_chunks; ++g) {
      float pr = probs[g];
      so_seg res = {0, 0, 0};
      if (pr >= p->threshold && temp_end > 0) temp_end = 0;
      if (!triggered) {
         if (pr >= p->threshold) { triggered = 1; cur_start = g; }
      } else if (pr < p->neg_threshold) {
         if (temp_end == 0) temp_end = g;
         if (g - temp_end >= min_silence) {
            if (temp_end - cur_start >= min_speech) { res.start = cur_start; res.end = temp_end; res.valid = 1; }
            cur_start = 0; temp_end = 0; triggered = 0;
         }
      }
      if (res.valid) buffered = so_combine(buffered, res, p, out_seconds, out_chunks, &n, max_segments);
   }
   if (triggered) {
      int audio_len = (g - 1) * window;
      if (audio_len - (cur_start * window) > (min_speech * window)) {
         so_seg fin = { cur_start, audio_len / window, 1 };
         buffered = so_combine(buffered, fin, p, out_seconds, out_chunks, &n, max_segments);
      }
   }
   if (buffered.valid) n = so_emit(buffered, p, out_seconds, out_chunks, n, max_segments);
   return n;
}
