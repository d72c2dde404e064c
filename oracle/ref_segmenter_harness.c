/* ref_segmenter_harness.c -- TEST INFRASTRUCTURE (oracle/): drives the REFERENCE's own segmenter functions, compiled from where they lie.
 *
 * oracle/build_ref.sh pipes   this file's PROLOGUE  +  lines 165-299 of /root/reference/vadc.c  +  this file's DRIVER   into gcc (-x c -): the excerpt --
 * feed_probability (vadc.c:165-221), emit_speech_segment (:223-260), combine_or_emit_speech_segment (:262-299) -- is read by the compiler from the
 * reference tree at build time and exists in no file of this repository or of oracle/_ref/; the product is oracle/_ref/ref_segmenter (a binary).
 * What is NOT the reference's here, and a strict reader will weigh it so:
 *   - print_speech_stats (vadc.c:1037-1081: QueryPerformanceCounter, stderr only, no arithmetic of the segments) is an empty function;
 *   - the loop that feeds probabilities and the end-of-stream flush below RESTATE vadc.c:964-987 and :1005-1027 (they sit inside run_inference between Win32 I/O);
 *     the ms -> chunks rounding (vadc.c:756-768) is done by the caller.
 * stdin:  int32 n, float threshold, float neg_threshold, int32 min_silence_chunks, int32 min_speech_chunks, float speech_pad_ms, int32 output_format
 *         (0 seconds "%.2f,%.2f", 1 centiseconds), int32 input_count (samples per chunk), then n float32 probabilities.
 * stdout: what vadc prints per segment (emit_speech_segment's own fprintf).
 * The two halves are cut at the marker lines. */
/* ---8<--- PROLOGUE */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vadc.h"
static inline void print_speech_stats(VADC_Stats stats) { (void)stats; }
/* ---8<--- DRIVER */
int main(void)
{
   struct { int32_t n; float threshold, neg_threshold; int32_t min_silence_chunks, min_speech_chunks; float speech_pad_ms; int32_t output_format, input_count; } h;
   if (fread(&h, sizeof h, 1, stdin) != 1 || h.n < 0) return 2;
   float *p = (float *)malloc(sizeof(float) * (size_t)(h.n > 0 ? h.n : 1));
   if (h.n > 0 && fread(p, sizeof(float), (size_t)h.n, stdin) != (size_t)h.n) return 2;
   const float seconds_per_chunk = (float)h.input_count / HARDCODED_SAMPLE_RATE;       /* vadc.c:846 */
   FeedState state = {0};
   FeedProbabilityResult buffered = {0};
   VADC_Stats stats = {0};
   const Segment_Output_Format fmt = (Segment_Output_Format)h.output_format;
   int global_chunk_index = 0;
   for (int i = 0; i < h.n; ++i) {                                                        /* vadc.c:964-987 */
      FeedProbabilityResult feed_result = feed_probability(&state, h.min_silence_chunks, h.min_speech_chunks, p[i], h.threshold, h.neg_threshold, global_chunk_index);
      if (feed_result.is_valid) buffered = combine_or_emit_speech_segment(buffered, feed_result, h.speech_pad_ms, fmt, &stats, seconds_per_chunk);
      ++global_chunk_index;
   }
   if (state.triggered) {                                                                 /* vadc.c:1005-1021 */
      int audio_length_samples = (int)((global_chunk_index - 1) * h.input_count);
      if (audio_length_samples - (state.current_speech_start * h.input_count) > (h.min_speech_chunks * h.input_count)) {
         FeedProbabilityResult final_segment;
         final_segment.is_valid = 1;
         final_segment.speech_start = state.current_speech_start;
         final_segment.speech_end = (int)(audio_length_samples / h.input_count);
         buffered = combine_or_emit_speech_segment(buffered, final_segment, h.speech_pad_ms, fmt, &stats, seconds_per_chunk);
      }
   }
   if (buffered.is_valid) emit_speech_segment(buffered, h.speech_pad_ms, fmt, &stats, seconds_per_chunk);      /* vadc.c:1023-1026 */
   free(p);
   return 0;
}
