"""ctypes bindings for the CPU oracle (oracle/libsilero_oracle.so) and, when present, the reference
build (oracle/_ref/libvadc_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never from the vadc_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsilero_oracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libvadc_ref.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(HERE, f)) for f in
                                             ("silero_oracle.c", "silero_v4_oracle.c", "silero_oracle.h", "silero_v4_oracle.h")):
        subprocess.check_call(["make", "-C", HERE, "libsilero_oracle.so"], stdout=subprocess.DEVNULL)


class _Layer(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("cin", "cout", "t_in", "t_out", "stride", "has_proj")] + \
               [(n, C.POINTER(C.c_float)) for n in (
                   "dw_w", "dw_b", "pw_w", "pw_b", "proj_w", "proj_b", "qkv_w", "qkv_b", "out_w", "out_b",
                   "n1_w", "n1_b", "l1_w", "l1_b", "l2_w", "l2_b", "n2_w", "n2_b", "conv_w", "conv_b",
                   "bn_w", "bn_b", "bn_mean", "bn_var")]


class _Taps(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("padded", "stft_conv", "magnitude", "normalized", "l1", "l2", "l3", "l4", "lstm_out")]


class _TapsV4(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("magnitude", "normalized", "l1", "l2", "l3", "l4", "lstm_out")]


TAP_SHAPES_V4 = {"magnitude": (129, 24), "normalized": (129, 24), "l1": (16, 12), "l2": (32, 6), "l3": (32, 3),
                 "l4": (64, 3), "lstm_out": (3, 64)}


class _SegParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("threshold", "neg_threshold", "min_silence_ms", "min_speech_ms",
                                         "speech_pad_ms", "seconds_per_chunk")]


TAP_SHAPES = {
    "padded": (1792,), "stft_conv": (258, 25), "magnitude": (129, 25), "normalized": (129, 25),
    "l1": (16, 13), "l2": (32, 7), "l3": (32, 7), "l4": (64, 7), "lstm_out": (7, 64),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.so_model_from_bytes.restype = C.c_void_p
        L.so_model_from_bytes.argtypes = [C.c_char_p, C.c_size_t]
        L.so_model_free.argtypes = [C.c_void_p]
        L.so_forward_chunk.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.POINTER(_Taps)]
        L.so_forward_stream_f32.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        L.so_forward_stream_s16.argtypes = [C.c_void_p, _i16p, C.c_int, _f32p, _f32p, _f32p]
        L.so_dot.restype = C.c_float
        L.so_dot.argtypes = [_f32p, _f32p, C.c_int]
        L.so_segments.restype = C.c_int
        L.so_segments.argtypes = [_f32p, C.c_int, C.c_int, C.POINTER(_SegParams), _f32p, _i32p, C.c_int]
        L.so4_model_from_bytes.restype = C.c_void_p
        L.so4_model_from_bytes.argtypes = [C.c_char_p, C.c_size_t]
        L.so4_model_free.argtypes = [C.c_void_p]
        L.so4_forward_chunk.restype = C.c_float
        L.so4_forward_chunk.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.POINTER(_TapsV4)]
        L.so4_forward_stream_f32.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, _f32p, _f32p]
        L.so4_forward_stream_s16.argtypes = [C.c_void_p, _i16p, C.c_int, _f32p, _f32p, _f32p]
        L.so4_forward_stream_f32_w.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p]
        L.so5_model_from_bytes.restype = C.c_void_p
        L.so5_model_from_bytes.argtypes = [C.c_char_p, C.c_size_t]
        L.so5_model_free.argtypes = [C.c_void_p]
        L.so5_forward_stream_f32.argtypes = [C.c_void_p, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _over_streams(fn, n_streams):
    """[fn(0), fn(1), ...] -- on the host's cores when there are several streams: they are independent and the C library keeps no state between calls (a model is
    read-only), the foreign call releases the GIL.  At most 16 threads: a one-GPU box's CPU share."""
    workers = min(n_streams, 16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    if workers <= 1:
        return [fn(s) for s in range(n_streams)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, range(n_streams)))


class Oracle:
    """Whole-path oracle for one weights blob."""

    def __init__(self, weights_blob: bytes):
        self._L = lib()
        self._blob = weights_blob
        self._m = self._L.so_model_from_bytes(weights_blob, len(weights_blob))
        if not self._m:
            raise ValueError("oracle: malformed weights blob")

    def __del__(self):
        if getattr(self, "_m", None):
            self._L.so_model_free(self._m)
            self._m = None

    @staticmethod
    def new_state():
        return np.zeros((2, 64), np.float32), np.zeros((2, 64), np.float32)

    def forward_chunk(self, samples, h, c, taps=False):
        samples = _c(samples)
        assert samples.shape == (1536,) and h.shape == (2, 64) and c.shape == (2, 64)
        out = np.zeros(2, np.float32)
        tp = None
        tapd = {}
        if taps:
            tp = _Taps()
            for k, shp in TAP_SHAPES.items():
                tapd[k] = np.zeros(shp, np.float32)
                setattr(tp, k, _fp(tapd[k]))
        self._L.so_forward_chunk(self._m, samples, h, c, out, C.byref(tp) if tp is not None else None)
        return (out, tapd) if taps else out

    def forward_stream(self, pcm_or_f32, h=None, c=None):
        """One stream, consecutive chunks, state carried. Returns probs [n,2] (and updates h,c in place)."""
        x = np.ascontiguousarray(pcm_or_f32).reshape(-1)
        n = x.size // 1536
        if h is None:
            h, c = self.new_state()
        probs = np.zeros((n, 2), np.float32)
        if x.dtype == np.int16:
            self._L.so_forward_stream_s16(self._m, x, n, h, c, probs)
        else:
            self._L.so_forward_stream_f32(self._m, _c(x), n, h, c, probs)
        return probs

    def forward_streams(self, pcm, h=None, c=None):
        """pcm int16/f32 [S, n*1536] -> probs [S, n] (speech prob only); h,c [S,2,64] updated in place."""
        pcm = np.ascontiguousarray(pcm)
        S = pcm.shape[0]
        n = pcm.shape[1] // 1536
        if h is None:
            h = np.zeros((S, 2, 64), np.float32)
            c = np.zeros((S, 2, 64), np.float32)
        out = np.zeros((S, n), np.float32)

        def one(s):
            out[s] = self.forward_stream(pcm[s], h[s], c[s])[:, 1]
        _over_streams(one, S)
        return out


class OracleV4:
    """Whole-path oracle for Silero v4 / 16 kHz (oracle/silero_v4_oracle.c; weights: 36-tensor container written by
    vadc_amd/onnx_weights.py).  Returns ONE speech probability per chunk."""

    def __init__(self, weights_blob: bytes):
        self._L = lib()
        self._blob = weights_blob
        self._m = self._L.so4_model_from_bytes(weights_blob, len(weights_blob))
        if not self._m:
            raise ValueError("oracle: malformed v4 weights blob")

    def __del__(self):
        if getattr(self, "_m", None):
            self._L.so4_model_free(self._m)
            self._m = None

    new_state = staticmethod(Oracle.new_state)

    def forward_chunk(self, samples, h, c, taps=False):
        samples = _c(samples)
        assert samples.shape == (1536,) and h.shape == (2, 64) and c.shape == (2, 64)
        tp, tapd = None, {}
        if taps:
            tp = _TapsV4()
            for k, shp in TAP_SHAPES_V4.items():
                tapd[k] = np.zeros(shp, np.float32)
                setattr(tp, k, _fp(tapd[k]))
        p = self._L.so4_forward_chunk(self._m, samples, h, c, C.byref(tp) if tp is not None else None)
        return (p, tapd) if taps else p

    def forward_stream(self, pcm_or_f32, h=None, c=None, window=1536):
        x = np.ascontiguousarray(pcm_or_f32).reshape(-1)
        n = x.size // window
        if h is None:
            h, c = self.new_state()
        probs = np.zeros(n, np.float32)
        if window != 1536:                       # 512 ... 1536-sample windows (onnx_helpers.c:164-170)
            xf = x.astype(np.float32) / np.float32(32768) if x.dtype == np.int16 else _c(x)      # vadc.c:883,898
            self._L.so4_forward_stream_f32_w(self._m, _c(xf), n, window, h, c, probs)
        elif x.dtype == np.int16:
            self._L.so4_forward_stream_s16(self._m, x, n, h, c, probs)
        else:
            self._L.so4_forward_stream_f32(self._m, _c(x), n, h, c, probs)
        return probs

    def forward_streams(self, pcm, window=1536):
        pcm = np.ascontiguousarray(pcm)
        return np.stack(_over_streams(lambda s: self.forward_stream(pcm[s], window=window), pcm.shape[0]))


class OracleV5:
    """Whole-path oracle for Silero v5 shapes (oracle/silero_v5_oracle.c; 13-tensor container, see its header).  One probability per
    512-sample chunk; the 64-sample context, h and c [128] are carried per stream."""

    def __init__(self, weights_blob: bytes):
        self._L = lib()
        self._m = self._L.so5_model_from_bytes(weights_blob, len(weights_blob))
        if not self._m:
            raise ValueError("oracle: malformed v5 weights blob")

    def __del__(self):
        if getattr(self, "_m", None):
            self._L.so5_model_free(self._m)
            self._m = None

    @staticmethod
    def new_state():
        return np.zeros(64, np.float32), np.zeros(128, np.float32), np.zeros(128, np.float32)

    def forward_stream(self, pcm_or_f32, state=None):
        x = np.ascontiguousarray(pcm_or_f32).reshape(-1)
        if x.dtype == np.int16:
            x = x.astype(np.float32) / np.float32(32768)            # vadc.c:883,898
        n = x.size // 512
        ctx, h, c = state if state is not None else self.new_state()
        probs = np.zeros(n, np.float32)
        self._L.so5_forward_stream_f32(self._m, _c(x), n, ctx, h, c, probs)
        return probs

    def forward_streams(self, pcm):
        pcm = np.ascontiguousarray(pcm)
        return np.stack(_over_streams(lambda s: self.forward_stream(pcm[s]), pcm.shape[0]))


def segments(probs, threshold=0.5, neg_threshold_relative=0.15, min_silence_ms=200.0, min_speech_ms=250.0,
             speech_pad_ms=30.0, max_segments=4096):
    """Oracle of vadc's hysteresis segmenter.  Returns (seconds[n,2] float32, chunk_indices[n,2] int32)."""
    L = lib()
    probs = _c(probs).reshape(-1)
    p = _SegParams(threshold, threshold - neg_threshold_relative, min_silence_ms, min_speech_ms, speech_pad_ms,
                   np.float32(1536) / np.float32(16000))
    sec = np.zeros((max_segments, 2), np.float32)
    chk = np.zeros((max_segments, 2), np.int32)
    n = L.so_segments(probs, probs.size, probs.size * 1536, C.byref(p), sec, chk, max_segments)
    return sec[:n].copy(), chk[:n].copy()


# ---- individual ops (thin numpy-facing wrappers used by the fixture tests) ----
def _sig(name, argtypes):
    f = getattr(lib(), name)
    f.argtypes = argtypes
    f.restype = None
    return f


def dw_conv_k5(x, w, b):
    x = _c(x); ch, t = x.shape
    out = np.zeros_like(x)
    _sig("so_dw_conv_k5", [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p])(x, ch, t, _c(w).reshape(ch, 5), _c(b), out)
    return out


def conv_k1(x, w, b, stride=1):
    x = _c(x); cin, t = x.shape
    w = _c(w).reshape(-1, cin); cout = w.shape[0]
    t_out = 1 + (t - 1) // stride
    out = np.zeros((cout, t_out), np.float32)
    _sig("so_conv_k1", [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, _f32p])(x, cin, t, w, _c(b), cout, stride, out)
    return out


def conv_block(x, dw_w, dw_b, pw_w, pw_b, proj_w=None, proj_b=None):
    x = _c(x); cin, t = x.shape
    pw_w = _c(pw_w).reshape(-1, cin); cout = pw_w.shape[0]
    out = np.zeros((cout, t), np.float32)
    f = _sig("so_conv_block", [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_void_p, _f32p])
    pw_ = _c(proj_w).reshape(-1, cin) if proj_w is not None else None
    pb_ = _c(proj_b) if proj_b is not None else None
    f(x, cin, t, cout, _c(dw_w).reshape(cin, 5), _c(dw_b), pw_w, _c(pw_b),
      pw_.ctypes.data if pw_ is not None else None, pb_.ctypes.data if pb_ is not None else None, out)
    return out


def softmax_rows(x):
    x = _c(x).copy(); r, c_ = x.shape
    _sig("so_softmax_rows", [_f32p, C.c_int, C.c_int])(x, r, c_)
    return x


def layer_norm(x, w, b):
    x = _c(x); r, f_ = x.shape
    out = np.zeros_like(x)
    _sig("so_layer_norm", [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p])(x, r, f_, _c(w), _c(b), out)
    return out


def batch_norm(x, mean, var, w, b):
    x = _c(x); ch, t = x.shape
    out = np.zeros_like(x)
    _sig("so_batch_norm", [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p])(x, ch, t, _c(mean), _c(var), _c(w), _c(b), out)
    return out


def attention(x, qkv_w, qkv_b, out_w, out_b):
    x = _c(x); t, d = x.shape
    out = np.zeros_like(x)
    _sig("so_attention", [_f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p])(x, t, d, _c(qkv_w), _c(qkv_b), _c(out_w), _c(out_b), out)
    return out


_LAYER_KEYS = ["dw_w", "dw_b", "pw_w", "pw_b", "proj_w", "proj_b", "qkv_w", "qkv_b", "out_w", "out_b",
               "n1_w", "n1_b", "l1_w", "l1_b", "l2_w", "l2_b", "n2_w", "n2_b", "conv_w", "conv_b",
               "bn_w", "bn_b", "bn_mean", "bn_var"]


def make_layer(tensors, has_proj, stride, t_in):
    """tensors: list of arrays in the reference's positional order (24 with proj, 22 without)."""
    keys = [k for k in _LAYER_KEYS if has_proj or k not in ("proj_w", "proj_b")]
    assert len(tensors) == len(keys)
    L = _Layer()
    keep = []
    for k, a in zip(keys, tensors):
        a = _c(a); keep.append(a)
        setattr(L, k, _fp(a))
    pw = keep[keys.index("pw_w")]
    L.cout, L.cin = pw.shape[0], pw.shape[1]
    L.stride, L.has_proj, L.t_in, L.t_out = stride, int(has_proj), t_in, 1 + (t_in - 1) // stride
    L._keep = keep
    return L


def transformer_block(x, layer):
    x = _c(x); d, t = x.shape
    out = np.zeros_like(x)
    _sig("so_transformer_block", [_f32p, C.c_int, C.c_int, C.POINTER(_Layer), _f32p])(x, d, t, C.byref(layer), out)
    return out


def transformer_layer(x, layer):
    x = _c(x); cin, t = x.shape
    out = np.zeros((layer.cout, 1 + (t - 1) // layer.stride), np.float32)
    _sig("so_transformer_layer", [_f32p, C.POINTER(_Layer), C.c_int, _f32p])(x, C.byref(layer), t, out)
    return out


def adaptive_norm(x):
    x = _c(x).copy(); ch, t = x.shape
    _sig("so_adaptive_norm", [_f32p, C.c_int, C.c_int])(x, ch, t)
    return x


def lstm_seq(x, w, b, h, c):
    x = _c(x); steps = x.shape[0]
    h = _c(h).copy(); c = _c(c).copy()
    out = np.zeros((steps, 64), np.float32)
    _sig("so_lstm_seq", [_f32p, C.c_int, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p])(x, steps, _c(w), _c(b), h.shape[0], h, c, out)
    return out, h, c


def decoder(x, w, b):
    x = _c(x); ch, t = x.shape
    w = _c(w).reshape(-1, ch)
    out = np.zeros(w.shape[0], np.float32)
    _sig("so_decoder", [_f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, _f32p])(x, ch, t, w, _c(b), w.shape[0], out)
    return out


def stft_magnitude(samples, basis):
    """samples [1536] f32 -> (conv [258,25], magnitude [129,25])"""
    samples = _c(samples)
    padded = np.zeros(1792, np.float32)
    _sig("so_reflect_pad", [_f32p, C.c_int, C.c_int, C.c_int, _f32p])(samples, 1536, 128, 128, padded)
    conv = np.zeros((258, 25), np.float32)
    _sig("so_stft_conv", [_f32p, C.c_int, _f32p, _f32p])(padded, 1792, _c(basis).reshape(258, 256), conv)
    mag = np.zeros((129, 25), np.float32)
    _sig("so_magnitude", [_f32p, C.c_int, _f32p])(conv, 25, mag)
    return conv, mag


# ---- reference build (build container only) ----
class Reference:
    """The reference's own C hot path (oracle/_ref/libvadc_ref.so).  Raises FileNotFoundError if absent."""

    def __init__(self, weights_path: str):
        if not os.path.exists(REF_LIB_PATH):
            raise FileNotFoundError(REF_LIB_PATH)
        L = C.CDLL(REF_LIB_PATH)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_char_p]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_reset.argtypes = [C.c_void_p]
        L.ref_get_state.argtypes = [C.c_void_p, _f32p, _f32p]
        L.ref_set_state.argtypes = [C.c_void_p, _f32p, _f32p]
        L.ref_run.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.ref_stft.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        L.ref_adaptive_norm.argtypes = [C.c_void_p, C.c_int, _f32p]
        L.ref_encoder.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        self._L = L
        self._h = L.ref_create(weights_path.encode())
        if not self._h:
            raise ValueError("reference: cannot load weights")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.ref_destroy(self._h)
            self._h = None

    def reset(self):
        self._L.ref_reset(self._h)

    def state(self):
        h = np.zeros((2, 64), np.float32); c = np.zeros((2, 64), np.float32)
        self._L.ref_get_state(self._h, h, c)
        return h, c

    def run(self, samples_f32, batch=None):
        x = _c(samples_f32).reshape(-1, 1536)
        n = x.shape[0]
        batch = batch or n
        out = np.zeros((n, 2), np.float32)
        for i in range(0, n, batch):
            b = min(batch, n - i)
            o = np.zeros((b, 2), np.float32)
            self._L.ref_run(self._h, b, np.ascontiguousarray(x[i:i + b]), o)
            out[i:i + b] = o
        return out

    def stft(self, samples_f32):
        x = _c(samples_f32).reshape(-1, 1536)
        out = np.zeros((x.shape[0], 129, 25), np.float32)
        self._L.ref_stft(self._h, x.shape[0], x, out)
        return out

    def adaptive_norm(self, x):
        x = _c(x).reshape(-1, 129, 25).copy()
        self._L.ref_adaptive_norm(self._h, x.shape[0], x)
        return x

    def encoder(self, x):
        x = _c(x).reshape(-1, 129, 25)
        out = np.zeros((x.shape[0], 64, 7), np.float32)
        self._L.ref_encoder(self._h, x.shape[0], x, out)
        return out
