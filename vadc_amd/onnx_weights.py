"""Minimal ONNX reader for weights tooling: walks a ModelProto with a hand-written protobuf
varint parser (the `onnx` package is not a dependency) and returns every initializer / Constant tensor of the main
graph and of all nested subgraphs (the Silero graphs select the 16 kHz / 8 kHz branch with `If` nodes).

Only what the weights need is decoded: TensorProto {dims=1, data_type=2, float_data=4, int64_data=7, name=8,
raw_data=9}; GraphProto {node=1, initializer=5}; NodeProto {input=1, output=2, name=3, op_type=4, attribute=5};
AttributeProto {name=1, t=5, g=6, graphs=11}; ModelProto {graph=7}.
"""
from __future__ import annotations
import struct
from typing import Dict, List, Tuple
import numpy as np


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    v = 0; s = 0
    while True:
        c = b[i]; i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7


def _fields(b: bytes):
    """yield (field_number, wire_type, value) — value is int for varint/fixed, bytes for length-delimited"""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = b[i:i + 8]; i += 8
        elif w == 2:
            ln, i = _varint(b, i); v = b[i:i + ln]; i += ln
        elif w == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError(f"unsupported wire type {w}")
        yield f, w, v


_DT = {1: np.float32, 7: np.int64, 6: np.int32, 11: np.float64, 9: np.bool_}


def _tensor(b: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []; dt = 1; name = ""; raw = None; floats: List[float] = []; ints: List[int] = []
    for f, w, v in _fields(b):
        if f == 1:
            if w == 2:                       # packed
                j = 0
                while j < len(v):
                    d, j = _varint(v, j); dims.append(d)
            else:
                dims.append(v)
        elif f == 2: dt = v
        elif f == 4:
            if w == 2: floats.extend(struct.unpack(f"<{len(v) // 4}f", v))
            else: floats.append(struct.unpack("<f", v)[0])
        elif f == 7:
            if w == 2:
                j = 0
                while j < len(v):
                    d, j = _varint(v, j); ints.append(d if d < (1 << 63) else d - (1 << 64))
            else:
                ints.append(v)
        elif f == 8: name = v.decode()
        elif f == 9: raw = v
    np_dt = _DT.get(dt)
    if np_dt is None:
        return name, None
    if raw is not None: a = np.frombuffer(raw, dtype=np_dt).copy()
    elif floats: a = np.asarray(floats, dtype=np_dt)
    else: a = np.asarray(ints, dtype=np_dt)
    return name, a.reshape(dims) if dims else a.reshape(())


def _graph(b: bytes, prefix: str, out: Dict[str, np.ndarray], nodes: List[dict]):
    for f, w, v in _fields(b):
        if f == 5:                            # initializer
            name, a = _tensor(v)
            if a is not None: out[name] = a
        elif f == 1:                          # node
            node = {"inputs": [], "outputs": [], "name": "", "op": "", "scope": prefix}
            attrs = []
            for nf, nw, nv in _fields(v):
                if nf == 1: node["inputs"].append(nv.decode())
                elif nf == 2: node["outputs"].append(nv.decode())
                elif nf == 3: node["name"] = nv.decode()
                elif nf == 4: node["op"] = nv.decode()
                elif nf == 5: attrs.append(nv)
            nodes.append(node)
            for ab in attrs:
                aname = ""; t = None; graphs = []
                for af, aw, av in _fields(ab):
                    if af == 1: aname = av.decode()
                    elif af == 5: t = av
                    elif af == 6: graphs.append(av)
                    elif af == 11: graphs.append(av)
                if t is not None and node["op"] == "Constant" and node["outputs"]:
                    _, a = _tensor(t)
                    if a is not None: out[node["outputs"][0]] = a
                for g in graphs:
                    _graph(g, f"{prefix}{node['name'] or node['op']}/{aname}/", out, nodes)


def load_onnx_tensors(path: str) -> Tuple[Dict[str, np.ndarray], List[dict]]:
    """-> ({tensor name: array} over the main graph and all subgraphs, [node dicts in file order])"""
    b = open(path, "rb").read()
    out: Dict[str, np.ndarray] = {}
    nodes: List[dict] = []
    for f, w, v in _fields(b):
        if f == 7:
            _graph(v, "", out, nodes)
    return out, nodes


# ---- Silero v4 (16 kHz branch of silero_vad_v4.onnx) -> positional .testtensor container --------------------------
# Order consumed by libvadc_amd.so for model kind "v4" (counterpart of tensor.h:114-191 for v3.1):
#   0      forward_basis_buffer [258,1,256]
#   1..6   first_layer ConvBlock 258->16: dw_w [258,1,5], dw_b, pw_w [16,258,1], pw_b, proj_w, proj_b
#   7,8    conv 16->16 stride 2 (BatchNorm already folded by the exporter: initializers 1110/1111)
#   9..14  ConvBlock 16->32;   15,16  conv 32->32 stride 2 (1113/1114)
#   17..20 ConvBlock 32->32 (no proj);   21,22  conv 32->32 stride 2 (1116/1117)
#   23..28 ConvBlock 32->64;   29,30  conv 64->64 stride 1 (1119/1120)
#   31,32  LSTM weights [2,256,128] = [layer][i,f,g,o rows][x(64) | h(64)], biases [2,256] = Wb + Rb
#          (the reference C layout, lstm.c:31-110; ONNX stores W/R/B separately in gate order i,o,f,c)
#   33,34  decoder conv 64->1: weight [1,64,1], bias [1]
#   35     adaptive-normalization filter [1,1,7] (the constants of misc.c:5-13)
#   36     (8 kHz container only) sample_rate [1] = 8000: marks the 8 kHz branch of the graph (`model_8k.*`), whose third strided conv has stride 1
#          (silero_vad.py:178-181) -- 37 tensors select it in libvadc_amd.so
V4_LSTM_16K = (("343", "345", "347"), ("415", "417", "419"))
V4_FOLDED_16K = (("1110", "1111"), ("1113", "1114"), ("1116", "1117"), ("1119", "1120"))
V4_LSTM_8K = (("833", "835", "837"), ("905", "907", "909"))
V4_FOLDED_8K = (("1122", "1123"), ("1125", "1126"), ("1128", "1129"), ("1131", "1132"))


def _lstm_onnx_to_c(W, R, B):
    """ONNX W [1,256,64], R [1,256,64], B [1,512] in gate order i,o,f,c -> ([256,128] rows i,f,g,o, [256])"""
    order = [0, 2, 3, 1]                       # torch/C block k takes ONNX block order[k]
    W, R, B = W[0], R[0], B[0]
    w = np.concatenate([np.concatenate([W[64 * o:64 * o + 64], R[64 * o:64 * o + 64]], axis=1) for o in order], axis=0)
    b = np.concatenate([B[64 * o:64 * o + 64] + B[256 + 64 * o:256 + 64 * o + 64] for o in order])
    return w.astype(np.float32), b.astype(np.float32)


def silero_v4_16k_tensors(onnx_path: str, sr: int = 16000):
    """-> [(name, array)] in the positional order above; sr = 8000 takes the graph's 8 kHz branch and appends the sample-rate marker"""
    t, _ = load_onnx_tensors(onnx_path)
    if sr not in (16000, 8000):
        raise ValueError("sr must be 16000 or 8000")
    model = "model" if sr == 16000 else "model_8k"
    out = [("forward_basis_buffer", t[f"{model}.feature_extractor.forward_basis_buffer"])]
    def block(prefix, proj=True):
        names = ["dw_conv.0.weight", "dw_conv.0.bias", "pw_conv.0.weight", "pw_conv.0.bias"] + (["proj.weight", "proj.bias"] if proj else [])
        return [(f"{prefix}.{n}", t[f"{model}.{prefix}.{n}"]) for n in names]
    out += block("first_layer.0")
    convs = V4_FOLDED_16K if sr == 16000 else V4_FOLDED_8K
    out += [("encoder.0.folded.weight", t[convs[0][0]]), ("encoder.0.folded.bias", t[convs[0][1]])]
    out += block("encoder.3.0")
    out += [("encoder.4.folded.weight", t[convs[1][0]]), ("encoder.4.folded.bias", t[convs[1][1]])]
    out += block("encoder.7.0", proj=False)
    out += [("encoder.8.folded.weight", t[convs[2][0]]), ("encoder.8.folded.bias", t[convs[2][1]])]
    out += block("encoder.11.0")
    out += [("encoder.12.folded.weight", t[convs[3][0]]), ("encoder.12.folded.bias", t[convs[3][1]])]
    ws, bs = zip(*[_lstm_onnx_to_c(t[a], t[b], t[c]) for a, b, c in (V4_LSTM_16K if sr == 16000 else V4_LSTM_8K)])
    out += [("lstm_weights", np.stack(ws)), ("lstm_biases", np.stack(bs))]
    out += [("decoder_weights", t[f"{model}.decoder.decoder.1.weight"]), ("decoder_biases", t[f"{model}.decoder.decoder.1.bias"])]
    out += [("adaptive_normalization_filter", t[f"{model}.adaptive_normalization.filter_"])]
    if sr == 8000:
        out += [("sample_rate", np.asarray([8000.0], np.float32))]
    return [(n, np.ascontiguousarray(a, dtype=np.float32)) for n, a in out]


# ---- Silero v3 / v3.1 (silero_vad_v3.onnx) -> the 99-tensor container of the reference's C backend (tensor.h:114-191) -------------------------
# The exporter folded every BatchNorm into the strided conv in front of it and stored the Linear weights transposed (MatMul operands); the C backend
# wants them as PyTorch does (silero_v3.c:72-215 reads conv + BatchNorm separately, transformer.c:237-295).  The folded conv goes out as the conv and
# the BatchNorm as the identity (weight 1, bias 0, mean 0, var 1 - eps: batch_norm, misc.c:98-141, then divides by sqrt(1.0)).
V3_LAYERS = (  # conv block, transformer block, MatMul initializers (QKV, out, linear1, linear2), folded conv (weight, bias), has projection
    ("first_layer.0", "encoder.0", ("896", "897", "898", "899"), ("879", "880"), True),
    ("encoder.4.0", "encoder.5", ("900", "901", "902", "903"), ("882", "883"), True),
    ("encoder.9.0", "encoder.10", ("904", "905", "906", "907"), ("885", "886"), False),
    ("encoder.14.0", "encoder.15", ("908", "909", "910", "911"), ("888", "889"), True),
)
V3_LSTM = (("929", "930", "931"), ("949", "950", "951"))
V3_BN_EPS = 1e-5


def silero_v3_tensors(onnx_path: str):
    """-> [(name, array)] x 99 in the positional order of tensor.h:114-191 (names as in testdata/silero_v31_16k.testtensor)"""
    t, _ = load_onnx_tensors(onnx_path)
    out = [("forward_basis_buffer", t["feature_extractor.forward_basis_buffer"])]
    for i, (cb, tb, mm, fold, proj) in enumerate(V3_LAYERS):
        L = f"transformer_l{i + 1}"
        out += [(f"{L}.dw_conv_weights", t[f"{cb}.dw_conv.0.weight"]), (f"{L}.dw_conv_biases", t[f"{cb}.dw_conv.0.bias"]),
                (f"{L}.pw_conv_weights", t[f"{cb}.pw_conv.0.weight"]), (f"{L}.pw_conv_biases", t[f"{cb}.pw_conv.0.bias"])]
        if proj:
            out += [(f"{L}.proj_weights", t[f"{cb}.proj.weight"]), (f"{L}.proj_biases", t[f"{cb}.proj.bias"])]
        out += [(f"{L}.attention_weights", t[mm[0]].T), (f"{L}.attention_biases", t[f"{tb}.attention.QKV.bias"]),
                (f"{L}.attention_proj_weights", t[mm[1]].T), (f"{L}.attention_proj_biases", t[f"{tb}.attention.out_proj.bias"]),
                (f"{L}.norm1_weights", t[f"{tb}.norm1.weight"]), (f"{L}.norm1_biases", t[f"{tb}.norm1.bias"]),
                (f"{L}.linear1_weights", t[mm[2]].T), (f"{L}.linear1_biases", t[f"{tb}.linear1.bias"]),
                (f"{L}.linear2_weights", t[mm[3]].T), (f"{L}.linear2_biases", t[f"{tb}.linear2.bias"]),
                (f"{L}.norm2_weights", t[f"{tb}.norm2.weight"]), (f"{L}.norm2_biases", t[f"{tb}.norm2.bias"])]
        d = t[fold[1]].shape[0]
        out += [(f"{L}.conv_weights", t[fold[0]]), (f"{L}.conv_biases", t[fold[1]]),
                (f"{L}.batch_norm_weights", np.ones(d, np.float32)), (f"{L}.batch_norm_biases", np.zeros(d, np.float32)),
                (f"{L}.batch_norm_running_mean", np.zeros(d, np.float32)), (f"{L}.batch_norm_running_var", np.full(d, 1.0 - V3_BN_EPS, np.float32))]
    ws, bs = zip(*[_lstm_onnx_to_c(t[a], t[b], t[c]) for a, b, c in V3_LSTM])
    out += [("weights", np.stack(ws)), ("biases", np.stack(bs))]
    out += [("decoder_weights", t["decoder.1.weight"]), ("decoder_biases", t["decoder.1.bias"])]
    return [(n, np.ascontiguousarray(a, dtype=np.float32)) for n, a in out]


if __name__ == "__main__":
    import sys
    from . import testtensor
    if len(sys.argv) not in (3, 4):
        sys.exit("usage: python -m vadc_amd.onnx_weights silero_vad_v4.onnx out.testtensor [8000]   |   silero_vad_v3.onnx out.testtensor v3")
    if len(sys.argv) == 4 and sys.argv[3] == "v3":
        ts = silero_v3_tensors(sys.argv[1])
        testtensor.dump(sys.argv[2], ts)
        print(f"wrote {sys.argv[2]}: {len(ts)} tensors, {sum(a.size for _, a in ts)} floats")
        sys.exit(0)
    ts = silero_v4_16k_tensors(sys.argv[1], int(sys.argv[3]) if len(sys.argv) == 4 else 16000)
    testtensor.dump(sys.argv[2], ts)
    print(f"wrote {sys.argv[2]}: {len(ts)} tensors, {sum(a.size for _, a in ts)} floats")
