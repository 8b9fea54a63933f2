"""CPU tests of bench.py's output plumbing (mpv_amd/bench_common.py): the ONE stdout line stays short whatever the secondary records hold
(a 20 KB line was cut to its 8 000-character tail by the driver and could not be parsed), secondary records go to stderr and the
records file, and a PMC traffic file is quoted only for the binary it was measured on."""
import json

import pytest

from mpv_amd import bench_common as BC


def recorded_headline():
    """A headline record of the shape bench.py builds (numbers from profiles/r03/bench_n1_final.json), with the verbose members that used to be
    embedded in the stdout line."""
    kernels = {"kernel_%02d<with, template, arguments>" % i: {"launches": 1000 + i, "fetch_bytes_per_launch": 1.2345678e8, "write_bytes_per_launch": 9.87e7,
                                                               "hbm_bytes": 1e12, "avg_us": 123.4, "share_pct": 3.3, "GBps": 2500.0} for i in range(40)}
    secondary = {"value": 0.153, "unit": "windows/s", "ms_per_step": 6530.0, "dtype": "f32 " + "x" * 300,
                 "roofline": {"bound": "hbm", "achieved": 3000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.375, "traffic": 1.96e13,
                              "traffic_detail": {"kernels": kernels}, "note": "n" * 900,
                              "compute": {"achieved_tflops": 88.2, "peak_tflops": 833.3, "frac": 0.106}},
                 "cpu_baseline": {"value": 2.7e-4, "unit": "windows/s", "cores": 16, "kind": "port", "sample": "s" * 350}}
    out = {"metric": "rolling windows/sec (N=32, T=50) + MPV max-abs err vs CPU ref", "value": 312.4, "unit": "windows/s", "n_gpus": 1, "steps": 20,
           "warmup": 1, "ms_per_step": 579.5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (128x128 layers as 3x f16 MFMA with fp32 accumulate)", "data": "synthetic",
           "config": {"workload": "NsDiff cond_var on 100-node BA-graph biomass trajectory (BASELINE configs[1]): " + "w" * 200, "windows_per_step_per_gpu": 181,
                      "truncated_for_profiling": False, "noise": "on-device Philox4x32-10", "weights": "seeded default init", "summary": "fused", "parallelism": "1 rank"},
           "roofline": {"bound": "mfma", "achieved": 334.6, "peak": 833.33, "unit": "TFLOP/s", "frac": 0.4015, "traffic": 30290368, "traffic_stale": False,
                        "traffic_detail": {"hbm_bytes_per_launch": 30290368, "source": "profiles/r04/pmc_traffic_c2.json", "kernels": kernels},
                        "kernel": "nsdiff_chain_f16_kernel<1,true>", "kernel_avg_ms": 578.27, "launches": 20, "flop_per_launch": 1.935e14,
                        "peak_source": "p" * 200, "mfma_executed_tflops": 984.0, "vs_fp32_matrix_peak": 2.13, "note": "n" * 400},
           "cpu_baseline": {"value": 0.0478, "unit": "windows/s", "cores": 16, "kind": "port", "best_window_value": 0.05, "per_window_s": [20.9] * 4,
                            "sample": "first 4 of the 181 windows, full N=32 x T=50, oracle/nsdiff_oracle.py (torch CPU fp32 port of the reference eager path), 83.8 s"},
           "mpv_max_abs_err": 6.1e-5, "mpv_max_abs_err_ulps": 1.0, "mpv_max_rel_err": 9.8e-8, "samples_max_rel_err": 1.1e-5}
    others = {k: secondary for k in ("mode_f32", "mode_f16x2", "mode_f16", "other_summary_path", "full_nsdiff", "c5", "c4", "c3", "nsdiff_spatial", "tmdm")}
    out["other"] = {k: BC.digest(v) for k, v in others.items()}
    out["other"]["trained_checkpoints"] = {"wofx": BC.digest(secondary), "sim": BC.digest(secondary)}
    out["records_file"] = "gpurun_out/bench_records.jsonl"
    return out, others


def test_stdout_line_is_short_and_keeps_the_measurement():
    out, others = recorded_headline()
    assert len(json.dumps(out)) > 8000                                     # the verbose form is what broke the driver's parser
    line = json.dumps(BC.compact(out))
    assert len(line) < BC.LINE_LIMIT <= 4000
    rec = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "mpv_max_abs_err", "other"):
        assert key in rec, key
    assert rec["config"]["workload"].startswith("NsDiff cond_var on 100-node BA-graph biomass")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms"):
        assert key in rec["roofline"], key
    assert "traffic_detail" not in rec["roofline"] and "note" not in rec["roofline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in rec["cpu_baseline"], key
    assert rec["other"]["c3"] == {"value": 0.153, "unit": "windows/s", "bound": "hbm", "frac": 0.375}
    assert set(rec["other"]) == set(others) | {"trained_checkpoints"}


def test_compact_refuses_an_overlong_line():
    out, others = recorded_headline()
    out["other_configs"] = others                                          # round 3's mistake: secondary records inside the line
    with pytest.raises(ValueError, match="emit_record"):
        BC.compact(out)


def test_secondary_records_go_to_stderr_and_the_records_file(tmp_path, monkeypatch, capsys):
    path = tmp_path / "records.jsonl"
    monkeypatch.setenv("MPV_BENCH_RECORDS", str(path))
    out, others = recorded_headline()
    BC.emit_record("run", {"argv": []}, fresh=True)
    for k, v in others.items():
        BC.emit_record(k, v)
    BC.print_line(out)
    cap = capsys.readouterr()
    stdout_lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(stdout_lines) == 1 and json.loads(stdout_lines[0])["value"] == 312.4            # stdout: the headline line alone
    err = [json.loads(ln[len("[bench record] "):]) for ln in cap.err.splitlines() if ln.startswith("[bench record] ")]
    assert [r["record"] for r in err] == ["run"] + list(others)
    recs = [json.loads(ln) for ln in path.read_text().splitlines()]
    assert [r["record"] for r in recs] == ["run"] + list(others)
    assert "kernels" in recs[-1]["roofline"]["traffic_detail"]             # the full record survives there
    BC.emit_record("run", {"argv": []}, fresh=True)                        # a new run starts a new file
    assert len(path.read_text().splitlines()) == 1


def test_digest_of_a_failed_record():
    assert "error" in BC.digest({"error": "RuntimeError: boom"})
    assert BC.digest({"value": 2.0, "unit": "windows/s", "frac": 0.3}) == {"value": 2.0, "unit": "windows/s", "frac": 0.3}


def test_pmc_traffic_is_bound_to_the_binary(tmp_path):
    """profiles/*/pmc_traffic_<key>.json carries the sha256 of the library it was measured on; another binary gets `stale` and the roofline
    falls back to the compute view with `traffic: None`."""
    d = tmp_path / "profiles" / "r09"
    d.mkdir(parents=True)
    lib = tmp_path / "libmpv_hip.so"
    lib.write_bytes(b"binary one")
    sha = BC.lib_sha256(str(lib))
    (d / "pmc_traffic_c3.json").write_text(json.dumps({"_lib_sha256": sha, "_commit": "abc", "bytes_per_window": 8.0e12, "windows_in_step": 1,
                                                       "dominant_kernel": "k", "kernels": {}}))
    tr = BC.measured_traffic("c3", root=str(tmp_path), lib_hash=sha)
    assert tr["stale"] is False and tr["source"].endswith("pmc_traffic_c3.json")
    roof = BC.multi_kernel_roofline(tr, 0.25, 120.0, "f16x3", "kernels")
    assert roof["bound"] == "hbm" and roof["traffic"] == 8.0e12 and abs(roof["achieved"] - 2000.0) < 1e-9 and abs(roof["frac"] - 0.25) < 1e-12
    lib.write_bytes(b"binary two")
    tr = BC.measured_traffic("c3", root=str(tmp_path), lib_hash=BC.lib_sha256(str(lib)), src_hash="unrelated")
    assert tr["stale"] is True
    roof = BC.multi_kernel_roofline(tr, 0.25, 120.0, "f16x3", "kernels")
    assert roof["traffic"] is None and roof["traffic_stale"] is True and roof["bound"] == "mfma"
    # a rebuild of UNCHANGED sources (another box, another hash of the .so) keeps the measurement; changed sources do not
    (d / "pmc_traffic_c3.json").write_text(json.dumps({"_lib_sha256": sha, "_csrc_sha256": "s1", "bytes_per_window": 8.0e12}))
    other = BC.lib_sha256(str(lib))
    assert BC.measured_traffic("c3", root=str(tmp_path), lib_hash=other, src_hash="s1")["stale"] is False
    assert BC.measured_traffic("c3", root=str(tmp_path), lib_hash=other, src_hash="s2")["stale"] is True
    (d / "pmc_traffic_c3.json").write_text(json.dumps({"bytes_per_window": 8.0e12}))       # a file from before the binding: never quoted
    assert BC.measured_traffic("c3", root=str(tmp_path), lib_hash=sha)["stale"] is True
    assert BC.measured_traffic("nope", root=str(tmp_path)) is None


def test_compute_fraction_is_stated_against_the_executing_pipe():
    """Split-f16 contractions execute 3 f16 MFMA products per algorithmic product: their ceiling is 2500 / 3, not the fp32 matrix peak
    (VERDICT r3 item 3a); the fp32-pipe ratio survives only as vs_fp32_matrix_peak."""
    roof = BC.multi_kernel_roofline(None, 1.0, 88.2, "f16x3", "k")
    assert abs(roof["peak"] - 2500.0 / 3) < 1e-9 and abs(roof["frac"] - 88.2 / (2500.0 / 3)) < 1e-12
    assert abs(roof["compute"]["vs_fp32_matrix_peak"] - 88.2 / 157.3) < 1e-12
    roof = BC.multi_kernel_roofline(None, 1.0, 88.2, "f32", "k")
    assert roof["peak"] == 157.3
    assert BC.split_peak("f16x2")[0] == 1250.0 and BC.split_peak("f16")[0] == 2500.0
