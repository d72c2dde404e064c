"""The bench line the driver parses: the committed profiles/rNN/bench_default.json (a verbatim `python bench.py` output) must carry every
field of the contract, with consistent arithmetic.  CPU-only: checks the artifact, does not run the bench."""
import glob
import json
import os

import pytest

from conftest import ROOT


def _latest_default():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default.json")))
    assert files, "no profiles/rNN/bench_default.json committed"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_fields():
    d = _latest_default()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "audio-seconds/sec"
    # dtype says what runs: the exact-tree STFT in fp32 on the vector ALU, the GEMMs of layers 2-4 and the LSTM as three fp16 MFMAs on split operands
    assert "f32" in d["dtype"] and "split-f16x3" in d["dtype"]
    # value = streams x chunks x 0.096 s / step time
    cfg = d["config"]
    want = cfg["streams_per_gpu"] * cfg["chunks_per_step"] * 0.096 / (d["ms_per_step"] * 1e-3)
    assert abs(d["value"] - want) / want < 1e-3


def test_roofline_and_cpu_baseline_objects():
    d = _latest_default()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["frac"] <= 1.0
    # achieved = executed FLOP (binding pipe) per launch / average launch duration of the dominant kernel; the algorithmic figure rides along
    flop = r.get("executed_flop_per_chunk", r["algorithmic_flop_per_chunk"])
    want = flop * r["chunks_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12
    assert abs(r["achieved"] - want) / want < 1e-2
    # round 3: the same rate against the FMA peak, and measured HBM bytes over algorithmic bytes of the dominant kernel
    assert abs(r["frac_of_fma_peak"] - r["achieved"] / 157.3) < 1e-3
    assert r["traffic"] is None or abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 1e-2
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_largest_single_gpu_config_rides_along():
    """BASELINE config 3 (4096 streams x 16 chunks, SPLIT16, graph replay) is timed in the same run and printed beside the headline; `value` stays on
    config 2 so that rounds stay comparable; host_fed comes from the asynchronous host-buffer entry points with the synchronous rate beside it"""
    d = _latest_default()
    c = d["configs"]["4096x16"]
    for k in ("value", "ms_per_step", "steps", "precision", "hipgraph", "roofline_kernel", "roofline_frac"):
        assert k in c, k
    assert c["precision"] == "split16" and c["hipgraph"] is True and 0 < c["roofline_frac"] <= 1.0
    want = 4096 * 16 * 0.096 / (c["ms_per_step"] * 1e-3)
    assert abs(c["value"] - want) / want < 1e-3
    assert d["config"]["streams_per_gpu"] == 256 and d["config"]["chunks_per_step"] == 96
    h = d["host_fed"]
    assert h["value"] > 0 and h["synchronous"] > 0 and h["pcie_gb_per_s"] > 0


def test_round4_fields_of_the_line():
    """round 4: the workload string names the arithmetic that runs, every kernel's fraction of its pipe rides in the line, the source of `traffic` is said, and two more
    configurations are timed beside the headline -- the north star's literal shape (10,240 streams x 1 chunk per call, with the latency a chunk sees and the host-fed
    rate) and the literal-fp32-MFMA engine on the headline workload"""
    d = _latest_default()
    assert "split-fp16" in d["config"]["workload"] and "STFT" in d["config"]["workload"]
    sf = d["stage_fracs"]
    for k in ("k_frontend", "k_layer1", "k_enc234", "k_lstm"):
        assert k in sf and 0 < sf[k][0] <= 1.0 and sf[k][1] in ("valu_nofma", "fp16", "fp32"), k
    assert sf["k_frontend"][1] == "valu_nofma" and abs(sf["k_frontend"][0] - d["roofline"]["frac"]) < 1e-3
    assert d["roofline"]["traffic"] is None or "profiles/" in d["roofline"]["traffic_source"]
    ns = d["configs"]["10240x1"]
    want = 10240 * 1 * 0.096 / (ns["ms_per_step"] * 1e-3)
    assert abs(ns["value"] - want) / want < 1e-3 and ns["value"] > 10240                      # >= 10 k concurrent real-time streams, with room
    assert 0 < ns["latency_ms"] < ns["latency_budget_ms"] == 96.0 and ns["host_fed"]["value"] > 10240
    lit = d["configs"]["256x96_fp32_mfma"]
    assert lit["options"] == {"encoder": 3, "lstm": 3, "layer1": 1} and 0 < lit["value"] < d["value"]
    assert len(json.dumps(d, separators=(",", ":"))) < 6000                                      # the driver reads the tail of stdout: the line stays short


def test_round5_fields_of_the_line():
    """round 5: BASELINE config 4 (Silero v4, 4096 streams x 16 chunks) is timed by the same run with the front-end kernel that ran named (the engine has two GEMM
    forms); host_fed is a sustained rate; and every BASELINE configuration has its kernel trace, PMC traffic and PMC compute summaries committed beside the line"""
    d = _latest_default()
    c = d["configs"]["v4_4096x16"]
    want = 4096 * 16 * 0.096 / (c["ms_per_step"] * 1e-3)
    assert abs(c["value"] - want) / want < 1e-3 and c["hipgraph"] is True
    assert c["frontend_kernel"] == "k_frontend_gemm2" and c["roofline_kernel"] == "k_frontend" and 0 < c["roofline_frac"] <= 1.0
    assert d["host_fed"]["pcie_gb_per_s"] > 0 and d["host_fed"]["value"] < d["value"]
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default.json")))[-1]
    for tag in ("bench_256x96", "bench_split16_4096x16", "bench_v4_4096x16"):
        for suffix in ("_kernel_stats.csv", "_pmc_traffic.json", "_pmc_compute.json"):
            assert os.path.exists(os.path.join(os.path.dirname(latest), tag + suffix)), tag + suffix
    v4 = json.load(open(os.path.join(os.path.dirname(latest), "bench_v4_4096x16_pmc_traffic.json")))
    assert v4["frontend_kernel"] == "k_frontend_gemm2" and "k_frontend_gemm2" in open(os.path.join(os.path.dirname(latest), "bench_v4_4096x16_kernel_stats.csv")).read()


def test_round6_fields_of_the_line():
    """round 6: the Silero v5 shapes ride in the default line (256 x 288: the shape round 2 measured at 1.23 M; 4096 x 48), with their kernel trace and PMC summary
    committed; the one-GPU line carries the per-GPU fields of the N-rank schema with N = 1"""
    d = _latest_default()
    assert list(d["configs"])[-2:] == ["v5_256x288", "v5_4096x48"]
    for key, S, Cn in (("v5_256x288", 256, 288), ("v5_4096x48", 4096, 48)):
        c = d["configs"][key]
        want = S * Cn * 0.032 / (c["ms_per_step"] * 1e-3)                          # a v5 window is 512 samples = 32 ms
        assert abs(c["value"] - want) / want < 1e-3, key
    assert d["configs"]["v5_256x288"]["value"] >= 3.0e6                            # the review's bar for the split-fp16 encoder
    assert d["n_gpus"] == 1 and d["total_streams"] == d["config"]["streams_per_gpu"] and abs(d["value_per_gpu"] - d["value"]) < 1.0
    assert d["rccl"]["world_size"] == 1 and "per GPU" in d["metric"]
    latest = os.path.dirname(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default.json")))[-1])
    for name in ("v5_256x288_kernel_stats.csv", "v5_256x288_pmc.json"):
        assert os.path.exists(os.path.join(latest, name)), name
    txt = open(os.path.join(latest, "v5_256x288_kernel_stats.csv")).read()
    for k in ("k_v5_encoder_h3", "k_v5_wih", "k_v5_lstm_h3"):
        assert k in txt, k


def test_n_rank_line_as_committed():
    """what `python bench.py --gpus 8` prints, pinned on committed lines: the dry run at world 8 (CPU, gloo, stand-in engine) and the one-GPU rehearsal at world 6 (the real
    rank code, six ranks sharing GPU 0, gloo through the host) -- job total under a truthful label, the per-GPU figure, what the process group reports, 4 B per chunk on the
    wire, and BASELINE config 5's shape timed under the same ranks"""
    from test_bench_spawn import _assert_n_rank_schema
    latest = os.path.dirname(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default.json")))[-1])
    d = json.load(open(os.path.join(latest, "bench_dry_run_world8.json")))
    assert d["dry_run"] is True
    _assert_n_rank_schema(d, 8, 256, 96, 4096, 16, "gloo")
    assert d["configs"]["8x4096x16"]["total_streams"] == 32768                     # BASELINE config 5: 32,768 streams sharded 4096 per GPU
    r = json.load(open(os.path.join(latest, "bench_rehearsal_world6.json")))
    S, Cn = r["config"]["streams_per_gpu"], r["config"]["chunks_per_step"]
    c5 = [k for k in r["configs"] if k.startswith("6x")][0]
    _, s5, c5n = c5.split("x")
    _assert_n_rank_schema(r, 6, S, Cn, int(s5), int(c5n), "gloo")
    assert "rehearsal" in r["configs"][c5]["rccl"]["note"]


def test_kernel_stats_list_every_kernel_of_the_step():
    """the tracked rocprofv3 summary carries every kernel of the step, the three k_lstm_layer launches (whose names rocprofv3 leaves mangled) included: layer 0, layer 1 and,
    since round 6, the REDO form behind the pair (a few microseconds: every workgroup leaves at once unless a tile's layer 1 gave up)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_256x96_kernel_stats.csv")))
    txt = open(files[-1]).read()
    for name in ("k_frontend_sym", "k_layer1_regs", "k_enc_fused", "k_lstm_layer"):
        assert name in txt, name
    assert txt.count("k_lstm_layer") == 3


@pytest.mark.parametrize("name", ["bench_256x96_kernel_stats.csv", "bench_256x96_pmc_traffic.json"])
def test_rocprof_summaries_are_committed(name):
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", name))
    assert files, name
    if name.endswith(".csv"):
        txt = open(sorted(files)[-1]).read()
        assert "k_frontend" in txt and "AverageNs" in txt
