"""bench.py's stdout contract (CPU): the ONE JSON line must stay well inside the driver's stdout window (round 3's 21-KB line
was cut and the headline went unparsed), carry every contract key plus `roofline` and `cpu_baseline`, and pick the PMC traffic
summary that matches the configuration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _cls(ms, frac, hbm=False):
    return {"bound": "hbm" if hbm else "mfma", "launches_per_step": 67.0, "ms_per_step": ms, "achieved": 1234.56789, "unit": "TFLOP/s",
            "peak": 2500.0, "frac": frac, "algorithmic_tflop_per_step": 123.456789}


def _roofline():
    names = ["conv3x3_row_persistent_256x256", "conv1x1_persistent_256x256", "conv3x3_row_512x128", "conv_other_tiles", "dense_wgrad",
             "pointwise_wgrad", "depthwise_wgrad", "depthwise_fwd_fanout", "depthwise_dgrad_sum", "depthwise_dgrad_epilogue",
             "depthwise_fwd", "bn_param_sums", "losses"]
    return {"bound": "mfma", "kernel": "k" * 200, "achieved": 1262.123456, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.50484938,
            "traffic": 2911308624.124031, "traffic_note": "n" * 300, "launches_per_step": 129.0, "ms_per_step_in_kernel": 160.39123,
            "algorithmic_tflop_per_step": 202.4912345, "classes": {n: _cls(100.123456 / (i + 1), 0.5441234) for i, n in enumerate(names)}}


def _full_record():
    sub = {"metric": "m" * 60, "value": 38.351234, "unit": "images/sec", "steps": 8, "warmup": 2, "ms_per_step": 208.612345, "dtype": "bf16",
           "roofline": _roofline(), "dense_wgrad": {"frac": 0.41234, "achieved": 1030.0}, "losses": {"hint": 1.0}, "what": "w" * 120,
           "config": {"plan": "P86", "mode": "A", "arch": "gscnn", "hint_loss": "mse", "per_gpu_batch": 8}}
    return {"metric": "images/sec KD train step, DeepLabV3+(WRN38) student 1024x2048", "value": 41.7123456, "unit": "images/sec", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 191.7891234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic", "_hw": "1024x2048",
            "profiler": {"events_in_timed_region": True, "ms_per_step_without_events": 190.123456, "ab_steps": 8},
            "config": {"workload": "w" * 400, "plan": "P92", "mode": "A", "arch": "deeplab", "hint_loss": "mse", "per_gpu_batch": 8,
                       "global_batch": 8, "parallelism": "dp1", "ranks_seen": 1, "backend": None,
                       "rank_devices": [{"rank": 0, "device": 0, "name": "AMD Instinct MI355X", "pci_bus_id": 5}], "replicas_identical_after_run": None, "teacher_overlap": False,
                       "teacher_backend": "hip", "per_step_host_syncs": False, "share_frozen_prefix": False,
                       "per_gpu_batch_sweep": {str(n): {"images_per_sec": 40.123456, "ms_per_step": 24.9, "steps": 8} for n in (1, 2, 4)}},
            "roofline": _roofline(), "dense_wgrad": None,
            "losses": {"hint": 123.456789012, "supervised": 3.0123456, "kd": 1e-6, "teacher": 3.0123456},
            "cpu_baseline": {"unit": "images/sec", "cores": 16, "host_cpu_count": 256, "kind": "port", "value": 0.01761234,
                             "step_1024x2048_s": 56.7812, "step_512x1024_s": 13.651234, "sample": "s" * 300},
            "sub_records": {n: sub for n in ("P79", "modeB", "gscnn_P86", "weighted_hint", "share_prefix", "ref_logging", "f32_parity")}}


def test_compact_line_is_short_and_complete():
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 15000            # the full record is the size that broke round 3's driver parse
    line = json.dumps(bench.compact_record(full, "gpurun_out/bench_full.json"))
    assert len(line) < 4096, len(line)
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["config"]["workload"] and "model" not in rec["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert abs(rec["value"] - full["value"]) < 1e-2 and abs(rec["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3
    assert set(rec["sub_records"]) == {"P79", "modeB", "gscnn_P86", "weighted_hint", "share_prefix", "ref_logging", "f32_parity"}
    assert rec["sub_records"]["modeB"]["wgrad_frac"] == 0.412
    assert rec["batch_sweep"] == {"1": 40.12, "2": 40.12, "4": 40.12}
    assert len(rec["roofline"]["classes"]) == 13 and rec["roofline"]["classes"]["losses"][1] == 0.544


def test_traffic_summary_follows_the_configuration():
    import bench
    for mode, arch, plan, suffix in (("A", "deeplab", "P92", "traffic_pmc.json"), ("B", "deeplab", "P92", "traffic_modeB_pmc.json")):
        t, src = bench.conv_traffic(plan, 8, 1024, 2048, "bf16", mode, arch)
        assert t and t > 1e9 and src.endswith(suffix), (mode, t, src)
    assert bench.conv_traffic("P79", 8, 1024, 2048, "bf16", "A", "deeplab") == (None, None)
    assert bench.conv_traffic("P92", 8, 512, 1024, "bf16", "A", "deeplab") == (None, None)
