"""Host-side handle on the HIP engine, shaped after the reference's backend boundary.

Reference interface mirrored here (names, argument meaning, error behaviour):

    backend_init(arena, model_path, &config) -> handle | NULL     silero.h:48, vadc.c:690-695
    backend_create_tensors(config, backend, buffers)               silero.h:76  (no-op for the C backend)
    backend_run(arena, &context, config)                           silero.h:53-74

`Engine.run(samples)` is backend_run for S independent streams x C consecutive chunks; `Engine.backend_run`
is the literal reference shape (one stream, `batch` consecutive chunks, output [batch, 2]).  All arithmetic
happens in libvadc_amd.so on the GPU; numpy here only owns host buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib

CHUNK = 1536
STAGES = {"magnitude": 0, "normalized": 1, "layer1": 2, "layer2": 3, "layer3": 4, "layer4": 5}
STAGE_SHAPES = {0: (129, 25), 1: (129, 25), 2: (16, 13), 3: (32, 7), 4: (32, 7), 5: (64, 7)}          # Silero v3.1
STAGE_SHAPES_V4 = {0: (129, 24), 1: (129, 24), 2: (16, 12), 3: (32, 6), 4: (32, 3), 5: (64, 3)}       # Silero v4
MODEL_V31, MODEL_V4, MODEL_V5 = 0, 1, 2
KERNELS = ["k_frontend", "k_layer1", "k_layer2", "k_layer3", "k_layer4", "k_lstm", "k_lstm_l1", "k_enc234"]


class VadcAmdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vadc_amd error {code}: {msg}")
        self.code = code


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, weights_blob: bytes, max_streams: int = 1, max_chunks_per_call: int = 96,
                 device: int = -1, precision: int = 0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.vadc_amd_create(weights_blob, len(weights_blob), device, max_streams, max_chunks_per_call,
                                     precision, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise VadcAmdError(rc, self._L.vadc_amd_last_error().decode())
        self.max_streams = max_streams
        self.max_chunks_per_call = max_chunks_per_call
        self.model = self.caps()["model_kind"]          # decided by the weights container (99 tensors v3.1 / 36 v4)
        self.stage_shapes = STAGE_SHAPES_V4 if self.model == MODEL_V4 else STAGE_SHAPES
        self.window = CHUNK
        self.sample_rate = self.caps()["sample_rate"]
        if self.sample_rate == 8000:                    # the v4 graph's 8 kHz branch: 768-sample chunks by default (the same 96 ms)
            self._set_v4_shapes(768)
        self.window = self.caps()["window_samples"]     # 512 for the v5 shapes (plus 64 samples of context the engine keeps per stream)

    def set_window(self, samples: int):
        """samples per chunk: 1536 (default); Silero v4 also 1280 / 1024 / 768 / 512 (option "window": --sequence_count of the reference's onnxruntime path)"""
        self.set_option("window", samples)
        self._set_v4_shapes(samples)

    def _set_v4_shapes(self, samples: int):
        self.window = samples
        if self.model == MODEL_V4:
            t = samples // 64
            t1 = (t + 1) // 2; t2 = (t1 + 1) // 2                 # a k = 1 conv of stride 2 keeps 1 + (T - 1) // 2 steps
            t3 = t2 if self.sample_rate == 8000 else (t2 + 1) // 2
            self.stage_shapes = {0: (129, t), 1: (129, t), 2: (16, t1), 3: (32, t2), 4: (32, t3), 5: (64, t3)}

    @classmethod
    def from_file(cls, path: str, **kw) -> "Engine":
        with open(path, "rb") as f:
            return cls(f.read(), **kw)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.vadc_amd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise VadcAmdError(rc, self._L.vadc_amd_last_error().decode())

    # ---- capabilities (what backend_init writes into Silero_Config) ----
    def caps(self) -> dict:
        c = _lib.Caps()
        self._check(self._L.vadc_amd_get_caps(self._h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in c._fields_}

    # ---- hot path ----
    def run(self, samples: np.ndarray) -> np.ndarray:
        """samples: int16 or float32 [S, C*window] (or [S, C, window]; window = 1536 unless set_window); returns probs float32 [S, C, 2]."""
        a = np.ascontiguousarray(samples)
        if a.ndim == 3:
            a = a.reshape(a.shape[0], -1)
        if a.ndim != 2 or a.shape[1] % self.window != 0 or a.shape[1] == 0:
            raise ValueError("samples must be [streams, chunks*window]")
        S, Cn = a.shape[0], a.shape[1] // self.window
        out = np.empty((S, Cn, 2), np.float32)
        if a.dtype == np.int16:
            self._check(self._L.vadc_amd_run_s16(self._h, _ptr(a), S, Cn, _ptr(out)))
        elif a.dtype == np.float32:
            self._check(self._L.vadc_amd_run_f32(self._h, _ptr(a), S, Cn, _ptr(out)))
        else:
            raise TypeError("samples must be int16 or float32")
        return out

    def backend_run(self, input_samples: np.ndarray, batch_size: int) -> np.ndarray:
        """Literal backend_run (silero.h:53-74): `batch_size` consecutive 1536-sample windows of ONE stream
        (f32 in [-1,1)), output [batch_size, 2] with the speech probability at index 1."""
        x = np.ascontiguousarray(input_samples, dtype=np.float32).reshape(1, batch_size * self.window)
        return self.run(x)[0]

    def run_async(self, samples: np.ndarray, out: np.ndarray):
        """asynchronous host-buffer call (vadc_amd_run_*_async): `samples` (int16 or float32, C-contiguous [S, C*window]) and `out` (float32 [S, C, 2])
        must stay alive and untouched until wait_async(); up to three calls are in flight"""
        if not (samples.flags.c_contiguous and out.flags.c_contiguous and out.dtype == np.float32):
            raise ValueError("run_async needs C-contiguous buffers and a float32 output")
        S, Cn = samples.shape[0], samples.reshape(samples.shape[0], -1).shape[1] // self.window
        if out.shape != (S, Cn, 2):
            raise ValueError("out must be [streams, chunks, 2]")
        fn = {np.dtype(np.int16): self._L.vadc_amd_run_s16_async, np.dtype(np.float32): self._L.vadc_amd_run_f32_async}[samples.dtype]
        self._check(fn(self._h, _ptr(samples), S, Cn, _ptr(out)))

    def wait_async(self):
        self._check(self._L.vadc_amd_wait_async(self._h))

    def unpin(self, buf: np.ndarray):
        """before a host buffer that was passed to run_async is freed: the engine forgets (and un-page-locks) it"""
        self._check(self._L.vadc_amd_unpin(self._h, _ptr(buf)))

    def run_device(self, d_in_ptr: int, dtype, n_streams: int, n_chunks: int, d_probs_ptr: int, hip_stream: int = 0):
        fn = self._L.vadc_amd_run_device_s16 if np.dtype(dtype) == np.int16 else self._L.vadc_amd_run_device_f32
        self._check(fn(self._h, C.c_void_p(d_in_ptr), n_streams, n_chunks, C.c_void_p(d_probs_ptr),
                       C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        self._check(self._L.vadc_amd_synchronize(self._h))

    def join(self, hip_stream: int = 0):
        """make `hip_stream` wait for every call issued so far (see option "defer_join")"""
        self._check(self._L.vadc_amd_join(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def speech_probabilities(self, d_probs: int, n_streams: int, n_chunks: int, d_speech: int, hip_stream: int = 0):
        """d_speech[stream][chunk] = d_probs[stream][chunk][1] on the device, enqueued on `hip_stream` (what a multi-GPU host gathers: 4 B per chunk)"""
        self._check(self._L.vadc_amd_speech_probabilities(self._h, C.c_void_p(d_probs), n_streams, n_chunks, C.c_void_p(d_speech),
                                                         C.c_void_p(hip_stream) if hip_stream else None))

    # ---- state ----
    def reset_streams(self, ids: Optional[np.ndarray] = None):
        if ids is None:
            self._check(self._L.vadc_amd_reset_streams(self._h, None, 0))
        else:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            self._check(self._L.vadc_amd_reset_streams(self._h, _ptr(ids), ids.size))

    def get_state(self, stream: int):
        """h, c as [2, 64] (two layers of 64; Silero v5: the same 128 floats are ONE layer of 128)"""
        h = np.empty((2, 64), np.float32)
        c = np.empty((2, 64), np.float32)
        self._check(self._L.vadc_amd_get_state(self._h, stream, _ptr(h), _ptr(c)))
        return h, c

    def set_state(self, stream: int, h: np.ndarray, c: np.ndarray):
        h = np.ascontiguousarray(h, dtype=np.float32).reshape(2, 64)
        c = np.ascontiguousarray(c, dtype=np.float32).reshape(2, 64)
        self._check(self._L.vadc_amd_set_state(self._h, stream, _ptr(h), _ptr(c)))

    def get_context(self, stream: int) -> np.ndarray:
        """Silero v5: the stream's 64-sample context (tail of its previous window)"""
        ctx = np.empty(64, np.float32)
        self._check(self._L.vadc_amd_get_context(self._h, stream, _ptr(ctx)))
        return ctx

    def set_context(self, stream: int, ctx: np.ndarray):
        ctx = np.ascontiguousarray(ctx, dtype=np.float32).reshape(64)
        self._check(self._L.vadc_amd_set_context(self._h, stream, _ptr(ctx)))

    # ---- stage taps ----
    def stage_from_samples(self, samples_f32: np.ndarray, stage: str) -> np.ndarray:
        x = np.ascontiguousarray(samples_f32, dtype=np.float32).reshape(-1, self.window)
        s = STAGES[stage]
        out = np.empty((x.shape[0],) + self.stage_shapes[s], np.float32)
        self._check(self._L.vadc_amd_debug_stage_from_samples(self._h, _ptr(x), x.shape[0], s, _ptr(out)))
        return out

    def stage_from_stage(self, x: np.ndarray, from_stage: str, to_stage: str) -> np.ndarray:
        f, t = STAGES[from_stage], STAGES[to_stage]
        x = np.ascontiguousarray(x, dtype=np.float32).reshape((-1,) + self.stage_shapes[f])
        out = np.empty((x.shape[0],) + self.stage_shapes[t], np.float32)
        self._check(self._L.vadc_amd_debug_stage_from_stage(self._h, _ptr(x), x.shape[0], f, t, _ptr(out)))
        return out

    def layer1_block(self, y: np.ndarray, what: str) -> np.ndarray:
        """parts of the first encoder layer in isolation: y [n, 16, 25] -> [n, 16, 25]; what = "attention" | "transformer_block" | "layer_norm" |
        "tail" (strided conv's arithmetic + folded BatchNorm + ReLU, every step); what = "conv_block": y [n, 129, 25] -> [n, 16, 25] (the product's input pipeline)"""
        code = {"attention": 1, "transformer_block": 2, "layer_norm": 3, "conv_block": 4, "tail": 5}[what]
        y = np.ascontiguousarray(y, dtype=np.float32).reshape((-1, 129, 25) if code == 4 else (-1, 16, 25))
        out = np.empty((y.shape[0], 16, 25), np.float32)
        self._check(self._L.vadc_amd_debug_layer1_block(self._h, code, _ptr(y), y.shape[0], _ptr(out)))
        return out

    def decoder(self, x: np.ndarray) -> np.ndarray:
        """the decoder alone: x [n, 64, steps] -> [n, 2] (the recurrence kernel's decoder with x in place of the second LSTM layer's output; state untouched)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((x.shape[0], 2), np.float32)
        self._check(self._L.vadc_amd_debug_decoder(self._h, _ptr(x), x.shape[0], _ptr(out)))
        return out

    def lstm_decoder(self, enc: np.ndarray) -> np.ndarray:
        """enc: [S, C, 64, steps] (steps = 7 for v3.1, 3 for v4) -> probs [S, C, 2] (uses and updates the per-stream state)."""
        enc = np.ascontiguousarray(enc, dtype=np.float32)
        S, Cn = enc.shape[0], enc.shape[1]
        out = np.empty((S, Cn, 2), np.float32)
        self._check(self._L.vadc_amd_debug_lstm_decoder(self._h, _ptr(enc), S, Cn, _ptr(out)))
        return out

    def set_option(self, key: str, value: int):
        self._check(self._L.vadc_amd_set_option(self._h, key.encode(), value))

    def get_option(self, key: str) -> int:
        v = C.c_int32()
        self._check(self._L.vadc_amd_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # ---- measurement ----
    def set_profiling(self, on: bool):
        self._check(self._L.vadc_amd_set_profiling(self._h, int(on)))

    def reset_kernel_times(self):
        self._check(self._L.vadc_amd_reset_kernel_times(self._h))

    def kernel_times(self) -> dict:
        out = {}
        for k, name in enumerate(KERNELS):
            n = C.c_int()
            ms = C.c_double()
            self._check(self._L.vadc_amd_get_kernel_time(self._h, k, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out
