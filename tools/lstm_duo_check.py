"""k_lstm_duo (engine variants 8 / 9) against k_lstm_wavefront_h3 (6) and k_lstm_layer (7): bit-identical probabilities and state, ragged tiles,
calls of one chunk, carried state.  python tools/lstm_duo_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vadc_amd import synth
from vadc_amd.engine import Engine
blob = open("tests/golden/reference_fixtures/silero_v31_16k.testtensor", "rb").read()
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
ok = True
for S, Cn, cuts in ((19, 6, (2, 4)), (1, 5, (1, 4)), (33, 9, (1, 1, 7)), (300, 8, (8,))):
    pcm = synth.make_streams(min(S, 24), Cn, seed0=5 + S)
    pcm = np.ascontiguousarray(np.tile(pcm, ((S + pcm.shape[0] - 1) // pcm.shape[0], 1))[:S])
    e = Engine(blob, max_streams=S, max_chunks_per_call=Cn, device=0)
    out, st = {}, {}
    for v in (6, 7, 8, 9):
        e.set_option("lstm", v); e.reset_streams()
        parts, c = [], 0
        for n in cuts:
            parts.append(e.run(pcm[:, c * 1536:(c + n) * 1536])); c += n
        out[v] = np.concatenate(parts, axis=1)
        assert e.get_option("lstm_kernel") == v, (v, e.get_option("lstm_kernel"))
        st[v] = [e.get_state(s_) for s_ in sorted({0, S // 2, S - 1})]
    for v in (7, 8, 9):
        same = np.array_equal(bits(out[6]), bits(out[v])) and all(np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1])) for a, b in zip(st[6], st[v]))
        print(f"S={S} C={Cn} cuts={cuts}: variant {v} vs 6: {'bit-identical' if same else 'DIFFERENT'}  max|dp| {np.abs(out[6] - out[v]).max():.3e}")
        ok &= same
    e.close()
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
