"""Synthetic speech-like 16 kHz s16le streams (SURVEY.md Appendix E recipe).

A voiced harmonic stack through three formants, 4 Hz syllabic amplitude modulation, on/off gating and a
low noise floor.  Drives Silero's speech probability across the whole [0,1] range (plain noise only
reaches ~1e-4), so it is the input for parity tests and for bench.py.  Pure numpy, deterministic per seed.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
CHUNK = 1536


def speech_like(n_samples: int, seed: int = 0) -> np.ndarray:
    """Return int16[n_samples]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    f0_base = 95.0 + 50.0 * rng.random()
    f0 = f0_base + 30.0 * np.sin(2 * np.pi * 0.7 * t + rng.random() * 6.28) + 10.0 * np.sin(2 * np.pi * 3.1 * t)
    phase = 2 * np.pi * np.cumsum(f0) / SAMPLE_RATE
    formants = [(700.0 * (0.85 + 0.3 * rng.random()), 130.0),
                (1220.0 * (0.85 + 0.3 * rng.random()), 70.0),
                (2600.0 * (0.9 + 0.2 * rng.random()), 160.0)]
    mean_f0 = float(f0.mean())
    sig = np.zeros(n_samples, dtype=np.float64)
    for h in range(1, 40):
        f = h * mean_f0
        if f >= SAMPLE_RATE / 2:
            break
        g = 0.02 + sum(np.exp(-0.5 * ((f - fc) / bw) ** 2) for fc, bw in formants)
        sig += g * np.sin(h * phase) / np.sqrt(h)
    am = (0.5 * (1.0 + np.sin(2 * np.pi * 4.0 * t - 1.5))) ** 2
    # on/off gate: alternating speech / silence spans of 1.0 .. 3.5 s, random start state
    gate = np.zeros(n_samples, dtype=np.float64)
    pos, on = 0, bool(rng.integers(0, 2))
    while pos < n_samples:
        span = int(SAMPLE_RATE * (1.0 + 2.5 * rng.random()))
        if on:
            gate[pos:pos + span] = 1.0
        pos += span
        on = not on
    # 10 ms raised-cosine edges so the gate does not click
    k = int(0.01 * SAMPLE_RATE)
    win = np.hanning(2 * k + 1)
    gate = np.convolve(gate, win / win.sum(), mode="same")
    sig = sig * am * gate
    peak = np.abs(sig).max()
    if peak > 0:
        sig = 0.5 * sig / peak
    sig = sig + 0.002 * rng.standard_normal(n_samples)
    return np.clip(np.rint(sig * 32767.0), -32768, 32767).astype(np.int16)


def control_stream(kind: str, n_samples: int, seed: int = 0) -> np.ndarray:
    """Edge-case inputs: 'zeros', 'noise' (-20 dBFS white), 'square' (full-scale clipping)."""
    if kind == "zeros":
        return np.zeros(n_samples, dtype=np.int16)
    if kind == "noise":
        rng = np.random.default_rng(seed)
        return np.clip(np.rint(rng.standard_normal(n_samples) * 3276.7), -32768, 32767).astype(np.int16)
    if kind == "square":
        x = np.where((np.arange(n_samples) // 40) % 2 == 0, 32767, -32768)
        return x.astype(np.int16)
    raise ValueError(kind)


def make_streams(n_streams: int, n_chunks: int, seed0: int = 0) -> np.ndarray:
    """int16[n_streams, n_chunks*1536]; stream k uses seed seed0+k."""
    out = np.empty((n_streams, n_chunks * CHUNK), dtype=np.int16)
    for k in range(n_streams):
        out[k] = speech_like(n_chunks * CHUNK, seed0 + k)
    return out
