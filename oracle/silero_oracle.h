/*
 * silero_oracle.h -- CPU restatement of the Silero VAD v3.1/16k forward pass as implemented by
 * IntendedConsequence/vadc's C backend (reference @ 2024_10_08).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link/load it.  The shipped path (vadc_amd/csrc, libvadc_amd.so)
 * never includes, links or calls anything under oracle/.
 *
 * Every function states the reference file:line whose arithmetic (including the ORDER of the
 * floating point reductions) it restates.  Build with -ffp-contract=off (oracle/Makefile): the
 * canonical reference build (MSVC /O2 /arch:AVX2, /fp:precise) never fuses mul+add.
 *
 * Pinning (see oracle/README.md, DESIGN.md "Oracle"):
 *   - the 18 in-tree known-answer fixtures of the reference (testdata/ *.testtensor, copied as data
 *     into tests/golden/reference_fixtures/) -- tests/test_oracle_fixtures.py
 *   - goldens generated from the reference's PyTorch restatement silero_vad.py::Silero_V3
 *     (tests/golden/gen_golden_from_python_reference.py) -- tests/test_oracle_golden.py
 *   - bit-level comparison against the reference C sources compiled in place (oracle/_ref,
 *     oracle/build_ref.sh) -- tests/test_oracle_vs_ref.py (build container only)
 */
#ifndef SILERO_ORACLE_H
#define SILERO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
   SO_CHUNK_SAMPLES = 1536,
   SO_PAD           = 128,
   SO_PADDED        = 1792,
   SO_FILTER_LEN    = 256,
   SO_HOP           = 64,
   SO_BINS          = 129,
   SO_FILTERS       = 258,
   SO_FRAMES        = 25,
   SO_HIDDEN        = 64,
   SO_LSTM_LAYERS   = 2,
   SO_LSTM_STEPS    = 7,
   SO_N_LAYERS      = 4
};

/* one encoder layer ("TransformerLayer_Weights", tensor.h:18-57) */
typedef struct so_layer {
   int cin, cout, t_in, t_out, stride, has_proj;
   const float *dw_w, *dw_b;           /* [cin,1,5], [cin]        */
   const float *pw_w, *pw_b;           /* [cout,cin,1], [cout]    */
   const float *proj_w, *proj_b;       /* [cout,cin,1], [cout] or NULL */
   const float *qkv_w, *qkv_b;         /* [3D,D], [3D]            */
   const float *out_w, *out_b;         /* [D,D], [D]              */
   const float *n1_w, *n1_b;
   const float *l1_w, *l1_b;
   const float *l2_w, *l2_b;
   const float *n2_w, *n2_b;
   const float *conv_w, *conv_b;       /* [D,D,1], [D]            */
   const float *bn_w, *bn_b, *bn_mean, *bn_var;
} so_layer;

typedef struct so_model {
   float *storage;                     /* owns every tensor's data */
   int    tensor_count;
   const float *basis;                 /* [258,1,256] */
   so_layer layer[SO_N_LAYERS];
   const float *lstm_w;                /* [2,256,128] = [layer][4H][x(64)|h(64)], gate order i,f,g,o */
   const float *lstm_b;                /* [2,256] */
   const float *dec_w, *dec_b;         /* [2,64,1], [2] */
} so_model;

/* per-chunk intermediates ("taps") for stage-level parity checks; any pointer may be NULL */
typedef struct so_taps {
   float *padded;      /* [1792]    reflect-padded input                           */
   float *stft_conv;   /* [258,25]  basis convolution (re rows 0..128, im 129..257) */
   float *magnitude;   /* [129,25]                                                  */
   float *normalized;  /* [129,25]  after adaptive normalization                    */
   float *l1;          /* [16,13]                                                   */
   float *l2;          /* [32,7]                                                    */
   float *l3;          /* [32,7]                                                    */
   float *l4;          /* [64,7]                                                    */
   float *lstm_out;    /* [7,64]    top-layer h per step                            */
} so_taps;

/* ---- model ---- */
/* Parse a 99-tensor v3.1 weights blob (.testtensor container; positional order tensor.h:114-191).
 * Returns NULL on malformed input. */
so_model *so_model_from_bytes(const void *blob, size_t len);
so_model *so_model_from_file(const char *path);
void      so_model_free(so_model *m);

/* ---- individual ops (shapes are passed explicitly; all row-major fp32) ---- */
void  so_reflect_pad(const float *in, int n, int pad_l, int pad_r, float *out);
void  so_stft_conv(const float *padded, int padded_len, const float *basis, float *out /*[258,frames]*/);
void  so_magnitude(const float *conv, int frames, float *mag /*[129,frames]*/);
void  so_adaptive_norm(float *x, int channels, int frames);
void  so_dw_conv_k5(const float *in, int channels, int t, const float *w, const float *b, float *out);
void  so_conv_k1(const float *in, int cin, int t, const float *w, const float *b, int cout, int stride, float *out);
void  so_conv_block(const float *in, int cin, int t, int cout,
                    const float *dw_w, const float *dw_b, const float *pw_w, const float *pw_b,
                    const float *proj_w, const float *proj_b, float *out);
float so_dot(const float *a, const float *b, int n);
void  so_linear(const float *in, int rows, int k, const float *w, const float *b, int n_out, float *out);
void  so_softmax_rows(float *x, int rows, int cols);
void  so_layer_norm(const float *in, int rows, int features, const float *w, const float *b, float *out);
void  so_batch_norm(const float *in, int channels, int t, const float *mean, const float *var,
                    const float *w, const float *b, float *out);
void  so_attention(const float *in /*[T,D]*/, int t, int d, const float *qkv_w, const float *qkv_b,
                   const float *out_w, const float *out_b, float *out /*[T,D]*/);
void  so_transformer_block(const float *in /*[D,T]*/, int d, int t, const so_layer *L, float *out /*[D,T]*/);
void  so_transformer_layer(const float *in /*[cin,t_in]*/, const so_layer *L, int t_in, float *out /*[cout,t_out]*/);
void  so_lstm_seq(const float *x /*[steps,64]*/, int steps, const float *w, const float *b, int layers,
                  float *h /*[layers,64] in/out*/, float *c /*[layers,64] in/out*/, float *out /*[steps,64]*/);
void  so_decoder(const float *in /*[64,T]*/, int channels, int t, const float *w, const float *b, int n_out, float *out);

/* ---- whole path ---- */
/* One chunk of one stream: samples[1536] (f32, already /32768), h,c [2,64] in/out, out[2]
 * (out[1] is the speech probability).  silero_v3.c:72-215 with batch_size==1. */
void  so_forward_chunk(const so_model *m, const float *samples, float *h, float *c, float out[2], const so_taps *taps);

/* n_chunks consecutive chunks of ONE stream (state carried), f32 input. probs: [n_chunks,2]. */
void  so_forward_stream_f32(const so_model *m, const float *samples, int n_chunks, float *h, float *c, float *probs);
/* same from s16le PCM: sample/32768.0f as in vadc.c:883,898 */
void  so_forward_stream_s16(const so_model *m, const int16_t *pcm, int n_chunks, float *h, float *c, float *probs);

/* ---- segmenter (vadc.c:165-299, 756-768, 1005-1027): probabilities -> speech segments ---- */
typedef struct so_seg_params {
   float threshold, neg_threshold;
   float min_silence_ms, min_speech_ms, speech_pad_ms;
   float seconds_per_chunk;           /* 1536/16000 */
} so_seg_params;
/* Writes up to max_segments (start_s,end_s) float pairs as the reference would print them (before
 * %.2f formatting) and the integer chunk indices; returns the number of segments. */
int   so_segments(const float *probs, int n_chunks, int total_samples, const so_seg_params *p,
                  float *out_seconds /*[max,2]*/, int *out_chunks /*[max,2]*/, int max_segments);

#ifdef __cplusplus
}
#endif
#endif
