/*
 * vadc_hip.c -- POSIX command-line host for the MI355X backend: the `vadc < audio.s16le` contract of the
 * reference (README.md:52-60, vadc.c:670-1035) on Linux, in C, on top of the C-ABI of include/vadc_amd.h.
 *
 *   stdin : 16 kHz mono s16le PCM (or a file named on the command line, decoded by an ffmpeg child: vadc.c:531-608)
 *                                        stdout: one "start,end" line per speech segment (seconds, %.2f),
 *                                                 or centiseconds (--output_centi_seconds), or one "%f" line per
 *                                                 1536-sample chunk (--raw_probabilities)
 *   stderr: diagnostics and --stats
 *
 * What is restated from the reference (own code, reference lines for parity checks):
 *   option table and defaults            vadc.c:1084-1124   (values <= 0 are ignored :1215-1218)
 *   96-chunk read window, s16 -> f32      vadc.c:799-805, 873-909   (conversion happens on the device here)
 *   tail: a partial last chunk yields no probability   vadc.c:964
 *   ms -> chunk rounding                 vadc.c:756-768
 *   hysteresis segmenter                 vadc.c:165-221 (feed_probability), :262-299 (combine_or_emit),
 *                                        :223-260 (emit, float32 time arithmetic), :1005-1027 (final flush)
 * The forward pass itself is vadc_amd_run_s16 (GPU); there is no CPU path in this program.
 */
#include <errno.h>
#include <fcntl.h>
#include <inttypes.h>
#include <spawn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "vadc_amd.h"

#define WINDOW_CHUNKS 96
#define CHUNK VADC_AMD_CHUNK_SAMPLES


typedef struct { int start, end, valid; } Segment;
typedef struct { int temp_end, current_start, triggered; } FeedState;

typedef struct {
   float min_silence_ms, min_speech_ms, threshold, neg_threshold_relative, speech_pad_ms;
   int batch, raw_probabilities, centiseconds, stats;
   const char *model;
   int sequence_count;          /* --sequence_count (vadc.c:1117, default 1536) */
   const char *input_file;      /* a bare argument: the file ffmpeg decodes in place of stdin (vadc.c:1225-1229, :810-815) */
   int audio_source;            /* --audio_source N: ffmpeg's -map 0:a:N (vadc.c:537) */
   float start_seconds;         /* --start_seconds F: ffmpeg's -ss (vadc.c:537) */
   const char *probabilities_in; /* --probabilities_in FILE (this program only): float32 speech probabilities, one per chunk, in place of the forward pass -- the
                                    segmenter alone, on any machine (tests/test_segmenter_vs_reference.py feeds it what the reference's own segmenter code was fed) */
} Options;

static double g_total_speech = 0.0;
/* --stats: the reference reports after EVERY emitted segment (print_speech_stats inside emit_speech_segment, vadc.c:259, :1040-1076): a progress line on
 * stderr ending in '\r'.  Same fields in the same order; the wall clock is CLOCK_MONOTONIC instead of QueryPerformanceCounter. */
static int64_t g_total_samples = 0;
static int g_sample_rate = 16000;
static struct timespec g_t0;
static void print_speech_stats(void)
{
   struct timespec t1;
   clock_gettime(CLOCK_MONOTONIC, &t1);
   const double wall = (t1.tv_sec - g_t0.tv_sec) + 1e-9 * (t1.tv_nsec - g_t0.tv_nsec);
   const double dur = (double)g_total_samples / g_sample_rate;
   const int hours = (int)(dur / 3600.0), minutes = (int)((dur - hours * 3600.0) / 60.0);
   const int seconds = (int)(dur - hours * 3600.0 - minutes * 60.0);
   const int ms = (int)((dur - hours * 3600.0 - minutes * 60.0 - seconds) * 1000.0);
   fprintf(stderr, "time=%02d:%02d:%02d.%04d", hours, minutes, seconds, ms);
   fprintf(stderr, " %7.2f speech (%5.1f%%), %5.1f / %5.1f (%5.1fx)\r", g_total_speech, dur > 0 ? g_total_speech / dur * 100.0 : 0.0, dur, wall, wall > 0 ? dur / wall : 0.0);
}

static void emit_segment(Segment s, const Options *o, float spc)
{
   const float pad_s = o->speech_pad_ms / 1000.0f;
   float end_p = (s.end * spc) + pad_s;
   float start_p = (s.start * spc) - pad_s;
   if (start_p < 0.0f) start_p = 0.0f;
   g_total_speech += (double)end_p - (double)start_p;
   if (o->centiseconds) {
      int64_t a = (int64_t)((double)start_p * 100.0 + 0.5), b = (int64_t)((double)end_p * 100.0 + 0.5);
      fprintf(stdout, "%" PRId64 ",%" PRId64 "\n", a, b);
   } else {
      fprintf(stdout, "%.2f,%.2f\n", start_p, end_p);
   }
   fflush(stdout);
   if (o->stats) print_speech_stats();                              /* vadc.c:259 */
}

static Segment combine_or_emit(Segment buffered, Segment cur, const Options *o, float spc)
{
   const float pad_s = o->speech_pad_ms / 1000.0f;
   float cur_start_p = (cur.start * spc) - pad_s;
   if (cur_start_p < 0.0f) cur_start_p = 0.0f;
   if (buffered.valid) {
      float buf_end_p = (buffered.end * spc) + pad_s;
      if (buf_end_p >= cur_start_p) { buffered.end = cur.end; return buffered; }
      emit_segment(buffered, o, spc);
   }
   return cur;
}

static Segment feed_probability(FeedState *st, int min_silence, int min_speech, float p, float thr, float neg_thr, int idx)
{
   Segment r = {0, 0, 0};
   if (p >= thr && st->temp_end > 0) st->temp_end = 0;
   if (!st->triggered) {
      if (p >= thr) { st->triggered = 1; st->current_start = idx; }
   } else if (p < neg_thr) {
      if (st->temp_end == 0) st->temp_end = idx;
      if (idx - st->temp_end >= min_silence) {
         if (st->temp_end - st->current_start >= min_speech) { r.start = st->current_start; r.end = st->temp_end; r.valid = 1; }
         st->current_start = 0; st->temp_end = 0; st->triggered = 0;
      }
   }
   return r;
}

static size_t read_full(int fd, void *buf, size_t want)
{
   size_t got = 0;
   while (got < want) {
      ssize_t n = read(fd, (char *)buf + got, want - got);
      if (n <= 0) break;
      got += (size_t)n;
   }
   return got;
}

static int parse_options(int argc, char **argv, Options *o)
{
   for (int i = 1; i < argc; ++i) {
      const char *a = argv[i];
      if (!strcmp(a, "--raw_probabilities")) { o->raw_probabilities = 1; continue; }
      if (!strcmp(a, "--output_centi_seconds")) { o->centiseconds = 1; continue; }
      if (!strcmp(a, "--stats")) { o->stats = 1; continue; }
      if (strncmp(a, "--", 2) != 0) { o->input_file = a; continue; }  /* vadc.c:1225-1229: a bare argument is the input file ffmpeg decodes (the last one named) */
      if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a); return -1; }
      const char *v = argv[++i];
      if (!strcmp(a, "--model")) { o->model = v; continue; }
      if (!strcmp(a, "--probabilities_in")) { o->probabilities_in = v; continue; }
      float f = (float)atof(v);
      if (f <= 0.0f) continue;                                   /* vadc.c:1215-1218 */
      if (!strcmp(a, "--min_silence")) o->min_silence_ms = f;
      else if (!strcmp(a, "--min_speech")) o->min_speech_ms = f;
      else if (!strcmp(a, "--threshold")) o->threshold = f;
      else if (!strcmp(a, "--neg_threshold_relative")) o->neg_threshold_relative = f;
      else if (!strcmp(a, "--speech_pad")) o->speech_pad_ms = f;
      else if (!strcmp(a, "--batch")) o->batch = (int)f;
      else if (!strcmp(a, "--sequence_count")) o->sequence_count = (int)f;     /* clamped to the backend's range after backend_init (vadc.c:743-752) */
      else if (!strcmp(a, "--audio_source")) o->audio_source = (int)f;          /* ffmpeg's stream selection and seek (vadc.c:532-538): no effect on stdin input */
      else if (!strcmp(a, "--start_seconds")) o->start_seconds = f;
      else { fprintf(stderr, "unknown option %s\n", a); return -1; }
   }
   return 0;
}

/* The reference's init_buffered_stream_ffmpeg (vadc.c:531-608) on POSIX: ffmpeg decodes the named file to mono s16le at the model's rate on a pipe this program
 * reads in place of stdin; ffmpeg gets no stdin of ours and keeps our stderr.  The same arguments as vadc.c:537, handed over as an argument vector (no shell, no
 * quoting of the file name).  Called BEFORE the engine exists: the child is started by a process that has not touched the GPU yet.
 * Returns the read end of the pipe, or -1. */
static pid_t g_ffmpeg_pid = 0;
static int spawn_ffmpeg(const Options *o, int sample_rate)
{
   int fds[2];
   if (pipe(fds) != 0) { fprintf(stderr, "Error creating ffmpeg pipe\n"); return -1; }      /* vadc.c:550 */
   char ss[32], map[32];
   snprintf(ss, sizeof ss, "%f", o->start_seconds);
   snprintf(map, sizeof map, "0:a:%d", o->audio_source);
   char *const argv[] = {"ffmpeg", "-hide_banner", "-loglevel", "error", "-nostats", "-ss", ss, "-i", (char *)o->input_file, "-map", map, "-vn", "-sn", "-dn",
                         "-ac", "1", "-ar", sample_rate == 8000 ? "8k" : "16k", "-f", "s16le", "-", NULL};
   posix_spawn_file_actions_t fa;
   posix_spawn_file_actions_init(&fa);
   posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
   posix_spawn_file_actions_adddup2(&fa, fds[1], 1);
   posix_spawn_file_actions_addclose(&fa, fds[0]);
   posix_spawn_file_actions_addclose(&fa, fds[1]);
   extern char **environ;
   const int rc = posix_spawnp(&g_ffmpeg_pid, "ffmpeg", &fa, NULL, argv, environ);
   posix_spawn_file_actions_destroy(&fa);
   close(fds[1]);
   if (rc != 0) { fprintf(stderr, "Error launching ffmpeg: %s\n", strerror(rc)); close(fds[0]); g_ffmpeg_pid = 0; return -1; }      /* vadc.c:570 */
   return fds[0];
}

/* the rate the weights container runs at, from its header alone (int32 version, int32 tensor count: 37 = the v4 graph's 8 kHz branch), before any engine exists */
static int container_sample_rate(const void *blob, long len)
{
   int32_t count = 0;
   if (len >= 8) memcpy(&count, (const char *)blob + 4, 4);
   return count == 37 ? 8000 : 16000;
}

/* Build-time weights embedding: the counterpart of the reference's cembed.c, which turns the weights file into a C array
 * so that vadc.exe carries its model (vadc.c:1110 default_model = embedded silero_v31_16k_weights).  Here the file is
 * pulled in with the assembler's .incbin (`make vadc_hip_embedded WEIGHTS=path`); --model still overrides it. */
#ifdef VADC_EMBED_WEIGHTS
__asm__(".section .rodata\n"
        ".balign 16\n"
        ".global vadc_embedded_weights_begin\n"
        "vadc_embedded_weights_begin:\n"
        ".incbin \"" VADC_EMBED_WEIGHTS "\"\n"
        ".global vadc_embedded_weights_end\n"
        "vadc_embedded_weights_end:\n"
        ".previous\n");
extern const unsigned char vadc_embedded_weights_begin[], vadc_embedded_weights_end[];
#endif

/* --probabilities_in: the segmenter over given probabilities (no engine, no GPU): the same rounding, feeding, merging, flush and printing as main() below */
static int segment_probabilities_file(const Options *o)
{
   FILE *f = fopen(o->probabilities_in, "rb");
   if (!f) { fprintf(stderr, "cannot open %s\n", o->probabilities_in); return -1; }
   const int chunk = o->sequence_count, sample_rate = 16000;
   if (chunk < 1) { fclose(f); return 2; }
   const float chunk_ms = chunk / (float)sample_rate * 1000.0f;    /* vadc.c:756 */
   int min_speech = (int)(o->min_speech_ms / chunk_ms + 0.5f);   if (min_speech < 1) min_speech = 1;
   int min_silence = (int)(o->min_silence_ms / chunk_ms + 0.5f); if (min_silence < 1) min_silence = 1;
   const float spc = (float)chunk / sample_rate;                   /* vadc.c:846 */
   const float neg_thr = o->threshold - o->neg_threshold_relative; /* vadc.c:1243 */
   FeedState st = {0, 0, 0};
   Segment buffered = {0, 0, 0};
   int global_idx = 0;
   float p;
   g_sample_rate = sample_rate;
   while (fread(&p, sizeof p, 1, f) == 1) {
      Segment r = feed_probability(&st, min_silence, min_speech, p, o->threshold, neg_thr, global_idx);
      if (r.valid) buffered = combine_or_emit(buffered, r, o, spc);
      ++global_idx;
   }
   fclose(f);
   if (st.triggered) {                                             /* vadc.c:1005-1027 */
      int audio_len = (global_idx - 1) * chunk;
      if (audio_len - (st.current_start * chunk) > (min_speech * chunk)) {
         Segment fin = {st.current_start, audio_len / chunk, 1};
         buffered = combine_or_emit(buffered, fin, o, spc);
      }
   }
   if (buffered.valid) emit_segment(buffered, o, spc);
   fflush(stdout);
   return 0;
}

extern char **environ;
static int under_a_profiler(void)
{
   const char *pre = getenv("LD_PRELOAD");
   if (pre && strstr(pre, "rocprof")) return 1;
   for (char **e = environ; e && *e; ++e) if (strncmp(*e, "ROCPROF", 7) == 0 || strncmp(*e, "ROCP_", 5) == 0) return 1;
   return 0;
}

int main(int argc, char **argv)
{
#ifdef VADC_EMBED_WEIGHTS
   const char *default_model = NULL;
#else
   const char *default_model = "silero_v31_16k.testtensor";
#endif
   Options o = {200.0f, 250.0f, 0.5f, 0.15f, 30.0f, WINDOW_CHUNKS, 0, 0, 0, default_model, CHUNK, NULL, 0, 0.0f, NULL};  /* vadc.c:1110-1124 */
   if (parse_options(argc, argv, &o)) return 2;
   if (o.batch > WINDOW_CHUNKS) o.batch = WINDOW_CHUNKS;
   if (o.probabilities_in) return segment_probabilities_file(&o);

   void *blob = NULL;
   long wlen = 0;
   if (o.model) {
      FILE *wf = fopen(o.model, "rb");
      if (!wf) { fprintf(stderr, "cannot open weights %s\n", o.model); return -1; }
      fseek(wf, 0, SEEK_END); wlen = ftell(wf); fseek(wf, 0, SEEK_SET);
      blob = malloc((size_t)wlen);
      if (!blob || fread(blob, 1, (size_t)wlen, wf) != (size_t)wlen) { fprintf(stderr, "cannot read weights\n"); return -1; }
      fclose(wf);
   }
#ifdef VADC_EMBED_WEIGHTS
   else {
      wlen = (long)(vadc_embedded_weights_end - vadc_embedded_weights_begin);
      blob = malloc((size_t)wlen);
      if (!blob) return -1;
      memcpy(blob, vadc_embedded_weights_begin, (size_t)wlen);
   }
#endif

   int in_fd = 0;                                                  /* vadc.c:810-819: the named file through ffmpeg, else stdin */
   if (o.input_file) {
      in_fd = spawn_ffmpeg(&o, container_sample_rate(blob, wlen));
      if (in_fd < 0) return -1;
   }

   vadc_amd_engine *eng = 0;
   if (vadc_amd_create(blob, (size_t)wlen, -1, 1, WINDOW_CHUNKS, VADC_AMD_PRECISION_FP32, &eng) != VADC_AMD_OK) {
      fprintf(stderr, "backend_init failed: %s\n", vadc_amd_last_error());
      return -1;                                                   /* vadc.c:692-695 */
   }
   free(blob);
   fprintf(stderr, "Running with batch size %d\n", o.batch);       /* vadc.c:716 */
   /* vadc.c:743-752: the desired sequence count is clamped to what backend_init reported.  The C backend's range is 1536..1536 (silero.h:41-42); a
    * Silero v4 container reports 512..1536 like the reference's onnxruntime path (onnx_helpers.c:164-170), of which this backend runs 512 / 768 /
    * 1024 / 1280 / 1536: other requests are rounded DOWN to the next of those. */
   vadc_amd_caps caps;
   if (vadc_amd_get_caps(eng, &caps) != VADC_AMD_OK) { fprintf(stderr, "get_caps failed: %s\n", vadc_amd_last_error()); return -1; }
   int seq = o.sequence_count;
   if (seq < caps.input_size_min) seq = caps.input_size_min;
   if (seq > caps.input_size_max) seq = caps.input_size_max;
   if (caps.input_size_step > 0)                                   /* the served windows are input_size_step apart: 512 / 768 / 1024 / 1280 / 1536 (8 kHz: 256 / 512 / 768) */
      seq = caps.input_size_min + (seq - caps.input_size_min) / caps.input_size_step * caps.input_size_step;
   if (seq != o.sequence_count) {
      fprintf(stderr, "--sequence_count %d: the backend runs %d-sample chunks", o.sequence_count, seq);
      if (caps.input_size_step > 0) {
         fprintf(stderr, " (it serves");
         for (int w = caps.input_size_min; w <= caps.input_size_max; w += caps.input_size_step) fprintf(stderr, " %d", w);
         fprintf(stderr, "; the reference's onnxruntime path takes any count in %d .. %d)", caps.input_size_min, caps.input_size_max);
      }
      fprintf(stderr, "\n");
   }
   if (seq != caps.window_samples && vadc_amd_set_option(eng, "window", seq) != VADC_AMD_OK) { fprintf(stderr, "cannot set the window: %s\n", vadc_amd_last_error()); return -1; }
   const int chunk = seq;
   fprintf(stderr, "Running with sequence count %d\n", chunk);     /* vadc.c:753 */

   const int sample_rate = caps.sample_rate;                       /* vadc.h:97 hard-codes 16000; the container of the v4 graph's 8 kHz branch says 8000 */
   const float chunk_ms = chunk / (float)sample_rate * 1000.0f;    /* vadc.c:756 */
   int min_speech = (int)(o.min_speech_ms / chunk_ms + 0.5f);   if (min_speech < 1) min_speech = 1;
   int min_silence = (int)(o.min_silence_ms / chunk_ms + 0.5f); if (min_silence < 1) min_silence = 1;
   const float spc = (float)chunk / sample_rate;                   /* vadc.c:846 */
   const float neg_thr = o.threshold - o.neg_threshold_relative;   /* vadc.c:1243 */

   static int16_t pcm[WINDOW_CHUNKS * CHUNK];
   static float probs[WINDOW_CHUNKS * 2];
   FeedState st = {0, 0, 0};
   Segment buffered = {0, 0, 0};
   int global_idx = 0;
   int64_t total_samples = 0;
   struct timespec t0, t1;
   clock_gettime(CLOCK_MONOTONIC, &t0);
   g_t0 = t0; g_sample_rate = sample_rate;

   const size_t window_bytes = (size_t)WINDOW_CHUNKS * chunk * sizeof(int16_t);      /* vadc.c:799-805: chunks_count = 96 chunks of input_count samples */
   for (;;) {
      size_t bytes = read_full(in_fd, pcm, window_bytes);
      size_t values = bytes / sizeof(int16_t);
      if (values == 0) break;
      total_samples += (int64_t)values;
      g_total_samples = total_samples;
      int n_chunks = (int)(values / chunk);                        /* vadc.c:964: a partial tail chunk is dropped */
      for (int c0 = 0; c0 < n_chunks; c0 += o.batch) {
         int n = n_chunks - c0 < o.batch ? n_chunks - c0 : o.batch;
         if (vadc_amd_run_s16(eng, pcm + (size_t)c0 * chunk, 1, n, probs + 2 * c0) != VADC_AMD_OK) {
            fprintf(stderr, "backend_run failed: %s\n", vadc_amd_last_error());
            return 1;
         }
      }
      for (int i = 0; i < n_chunks; ++i) {
         float p = probs[2 * i + 1];
         if (o.raw_probabilities) {
            printf("%f\n", p);                                     /* vadc.c:995 */
         } else {
            Segment r = feed_probability(&st, min_silence, min_speech, p, o.threshold, neg_thr, global_idx);
            if (r.valid) buffered = combine_or_emit(buffered, r, &o, spc);
         }
         ++global_idx;
      }
      if (bytes < window_bytes) break;
   }
   if (!o.raw_probabilities) {                                     /* vadc.c:1005-1027 */
      if (st.triggered) {
         int audio_len = (global_idx - 1) * chunk;
         if (audio_len - (st.current_start * chunk) > (min_speech * chunk)) {
            Segment fin = {st.current_start, audio_len / chunk, 1};
            buffered = combine_or_emit(buffered, fin, &o, spc);
         }
      }
      if (buffered.valid) emit_segment(buffered, &o, spc);
   }
   fflush(stdout);
   if (g_ffmpeg_pid > 0) {                                          /* the reference drops ffmpeg's handles at once (vadc.c:581-582); here its exit is collected, and a failure that left no audio is said */
      int status = 0;
      close(in_fd);
      if (waitpid(g_ffmpeg_pid, &status, 0) == g_ffmpeg_pid && total_samples == 0 && !(WIFEXITED(status) && WEXITSTATUS(status) == 0))
         fprintf(stderr, "ffmpeg gave no audio for '%s' (exit status %d)\n", o.input_file, WIFEXITED(status) ? WEXITSTATUS(status) : -1);
   }
   if (o.stats) {
      clock_gettime(CLOCK_MONOTONIC, &t1);
      double wall = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
      double dur = (double)total_samples / sample_rate;
      fprintf(stderr, "time=%.2fs speech=%.2fs (%.1f%%), duration=%.2fs (%.1fx)\n", wall, g_total_speech,
              dur > 0 ? 100.0 * g_total_speech / dur : 0.0, dur, wall > 0 ? dur / wall : 0.0);
   }
   vadc_amd_destroy(eng);
   if (getenv("VADC_AMD_TRACE_TEARDOWN")) { fprintf(stderr, "vadc_hip: engine destroyed, leaving main\n"); fflush(stderr); }
   /* The engine is gone and every stream flushed: leave without the HIP runtime's exit handlers.  Beside another process's GPU context they hung one short-lived
    * process in about two hundred on ROCm 7.2, after main() had returned (tools/cli_teardown_probe.py: 400 runs beside a parent that holds an engine). */
   fflush(NULL);
   if (under_a_profiler()) return 0;                                /* rocprofv3 writes its files from exit handlers */
   _exit(0);
}
