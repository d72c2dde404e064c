/*
 * ref_harness.c -- TEST INFRASTRUCTURE, build container only (never travels as a product dependency).
 *
 * Compiles the REFERENCE's own hot-path sources, unmodified and in place from $REFERENCE
 * (default /root/reference), into oracle/_ref/libvadc_ref.so so that the restatement in
 * silero_oracle.c can be checked against the real thing bit for bit.  The include list mirrors the
 * reference's own unity build for its tests (test.c:1-22).
 *
 * What is NOT compiled: the second half of memory.h (`#ifdef MEMORY_IMPLEMENTATION`), which is the
 * reference's Win32-only arena backing (VirtualAlloc / <windows.h>).  No stand-in for <windows.h> is
 * written.  The arena entry points that memory.h *declares* (memory.h:18-64) are defined below by this
 * harness -- a bump allocator over calloc'd memory -- exactly as any caller of the hot path owns its
 * arena (vadc.c:1133-1141).  They return zeroed memory and do no arithmetic, so they cannot influence
 * a single floating point result.  Weights are loaded with the reference's own load_testtensor() +
 * silero_weights_init() as test.c:1756-1770 does (silero.h is not included: it needs the generated
 * silero_v31_16k_weights.c).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <tracy/TracyC.h>

#if !defined(VADC_SLOW)
#define VADC_SLOW 0
#endif

#include "utils.h"
#include "tensor.h"

#include "conv.c"
#include "misc.c"
#include "stft.c"
#include "lstm.c"
#include "transformer.c"
#include "silero_v3.c"

#define MATHS_IMPLEMENTATION
#include "maths.h"

/* ---- arena entry points declared by memory.h:18-64, owned by the harness ---- */
void initializeMemoryArena(MemoryArena *arena, u8 *base, size_t size)
{
   arena->base = base; arena->size = size; arena->previous_used = 0; arena->used = 0; arena->temporaryMemoryCount = 0;
}

void *pushSize(MemoryArena *arena, size_t size, size_t alignment)
{
   size_t top = (size_t)(arena->base + arena->used);
   size_t mask = alignment - 1;
   size_t adj = (top & mask) ? alignment - (top & mask) : 0;
   size += adj;
   if (size > arena->size - arena->used) { fprintf(stderr, "ref_harness: arena exhausted\n"); abort(); }
   void *address = arena->base + arena->used + adj;
   arena->previous_used = arena->used + adj;
   arena->used += size;
   return address;
}

void *pushSizeZeroed(MemoryArena *arena, size_t size, size_t alignment)
{
   void *p = pushSize(arena, size, alignment);
   memset(p, 0, size);
   return p;
}

TemporaryMemory beginTemporaryMemory(MemoryArena *arena)
{
   TemporaryMemory t = { arena, arena->previous_used, arena->used };
   ++arena->temporaryMemoryCount;
   return t;
}

void endTemporaryMemory(TemporaryMemory t)
{
   t.arena->previous_used = t.previous_used;
   t.arena->used = t.used;
   --t.arena->temporaryMemoryCount;
}

const char *copyStringToArena(MemoryArena *arena, const char *s, size_t n)
{
   if (n == 0) n = strlen(s);
   char *c = pushSizeZeroed(arena, n + 1, 8);
   memmove(c, s, n);
   return c;
}

static MemoryArena g_debug_arena;
MemoryArena *DEBUG_getDebugArena()
{
   if (!g_debug_arena.base) {
      size_t size = (size_t)256 << 20;
      initializeMemoryArena(&g_debug_arena, calloc(1, size), size);
   }
   return &g_debug_arena;
}

/* ---- exported entry points (ctypes) ---- */
typedef struct Ref_Handle
{
   MemoryArena    arena;
   Silero_Context ctx;
} Ref_Handle;

void *ref_create(const char *weights_path)
{
   Ref_Handle *h = calloc(1, sizeof(*h));
   size_t size = (size_t)512 << 20;
   initializeMemoryArena(&h->arena, calloc(1, size), size);
   LoadTesttensorResult res = load_testtensor(&h->arena, weights_path);             /* test.c:1757 */
   if (res.tensor_count != 1 + 24 + 24 + 22 + 24 + 2 + 2) { free(h->arena.base); free(h); return 0; }
   h->ctx.weights = silero_weights_init(res);                                       /* test.c:1770 */
   h->ctx.state_lstm_h = tensor_zeros_3d(&h->arena, 2, 1, 64);                      /* silero.h:36-37 */
   h->ctx.state_lstm_c = tensor_zeros_3d(&h->arena, 2, 1, 64);
   return h;
}

void ref_destroy(void *hv)
{
   Ref_Handle *h = hv;
   if (!h) return;
   free(h->arena.base);
   free(h);
}

void ref_reset(void *hv)
{
   Ref_Handle *h = hv;
   memset(h->ctx.state_lstm_h->data, 0, h->ctx.state_lstm_h->nbytes);
   memset(h->ctx.state_lstm_c->data, 0, h->ctx.state_lstm_c->nbytes);
}

void ref_get_state(void *hv, float *h_out, float *c_out)
{
   Ref_Handle *h = hv;
   memcpy(h_out, h->ctx.state_lstm_h->data, 128 * sizeof(float));
   memcpy(c_out, h->ctx.state_lstm_c->data, 128 * sizeof(float));
}

void ref_set_state(void *hv, const float *h_in, const float *c_in)
{
   Ref_Handle *h = hv;
   memcpy(h->ctx.state_lstm_h->data, h_in, 128 * sizeof(float));
   memcpy(h->ctx.state_lstm_c->data, c_in, 128 * sizeof(float));
}

/* the hot path: silero_v3.c:72 with `batch` consecutive chunks of one stream; out [batch,2] */
void ref_run(void *hv, int batch, const float *samples, float *out)
{
   Ref_Handle *h = hv;
   TemporaryMemory mark = beginTemporaryMemory(&h->arena);
   TestTensor *o = silero_run_one_batch_with_context(&h->arena, &h->ctx, batch, 1536, (float *)samples);
   memcpy(out, o->data, sizeof(float) * 2 * batch);
   endTemporaryMemory(mark);
}

/* stage: reflect pad + STFT + magnitude (stft.c:226), samples [batch,1536] -> [batch,129,25] */
void ref_stft(void *hv, int batch, const float *samples, float *out_mag)
{
   Ref_Handle *h = hv;
   TemporaryMemory mark = beginTemporaryMemory(&h->arena);
   TestTensor *in = tensor_zeros_2d(&h->arena, batch, 1536);
   memcpy(in->data, samples, in->nbytes);
   TestTensor *out = tensor_zeros_3d(&h->arena, batch, 129, 25);
   my_stft(&h->arena, in, h->ctx.weights.forward_basis_buffer, out, 64, 128);
   memcpy(out_mag, out->data, out->nbytes);
   endTemporaryMemory(mark);
}

/* stage: adaptive normalization in place (misc.c:1), x [batch,129,25] */
void ref_adaptive_norm(void *hv, int batch, float *x)
{
   Ref_Handle *h = hv;
   TemporaryMemory mark = beginTemporaryMemory(&h->arena);
   TestTensor *t = tensor_zeros_3d(&h->arena, batch, 129, 25);
   memcpy(t->data, x, t->nbytes);
   adaptive_audio_normalization_inplace(&h->arena, t);
   memcpy(x, t->data, t->nbytes);
   endTemporaryMemory(mark);
}

/* stage: encoder (silero_v3.c:4), x [batch,129,25] -> [batch,64,7] */
void ref_encoder(void *hv, int batch, const float *x, float *out)
{
   Ref_Handle *h = hv;
   TemporaryMemory mark = beginTemporaryMemory(&h->arena);
   TestTensor *t = tensor_zeros_3d(&h->arena, batch, 129, 25);
   memcpy(t->data, x, t->nbytes);
   TestTensor *o = tensor_zeros_3d(&h->arena, batch, 64, 7);
   encoder(&h->arena, t, h->ctx.weights.encoder_weights, o);
   memcpy(out, o->data, o->nbytes);
   endTemporaryMemory(mark);
}
