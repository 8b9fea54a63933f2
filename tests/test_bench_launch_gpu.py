"""GPU tests (-m gpu) of the commands the driver runs: `python bench.py` (BENCH) and `python bench.py --gpus N` (SCALE; bench.py starts its N
workers itself when no launcher did).  Every bench runs as a fresh child process of pytest -- nothing is exec'd from a GPU-initialised process --
and the N > 1 cases use `--backend gloo` with both ranks on the one visible card (the RCCL branches are covered at world 1 by
tests/test_rccl_world1_gpu.py; the N > 1 RCCL curve itself is the driver's to measure on an 8-GPU node)."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def run_bench(tmp_path, *argv, env=None, timeout=900):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MPV_BENCH_RECORDS=str(tmp_path / "records.jsonl"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NCCL_DEBUG", "MPV_BENCH_WORKER"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.monotonic()
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=e, cwd=str(ROOT))
    return proc, time.monotonic() - t0


def only_line(proc):
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d:\n%s" % (len(lines), proc.stdout[-2000:])
    assert len(lines[0]) < 4000, "the stdout line is %d characters" % len(lines[0])
    return json.loads(lines[0])


def test_default_shaped_run_prints_one_short_line(tmp_path):
    """The BENCH command with every secondary leg that embeds records (other GEMM modes, sample-writing path, trained checkpoints, full NsDiff, the CPU
    baseline + parity on shared noise), bounded for the test: 2 steps, one CPU window, without the three other BASELINE configs (minutes each; their
    records take the same emit_record path, tests/test_bench_records.py).  VERDICT r3 item 1: the line must parse and stay far below the driver's
    8 000-character tail."""
    proc, _ = run_bench(tmp_path, "--steps", "2", "--warmup", "1", "--cpu-windows", "1", "--no-other-configs")
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = only_line(proc)
    assert rec["metric"].startswith("rolling windows/sec") and rec["unit"] == "windows/s" and rec["n_gpus"] == 1 and rec["steps"] == 2
    assert rec["value"] > 50 and abs(rec["ms_per_step"] * rec["value"] / 1e3 - rec["config"]["windows_per_step_per_gpu"]) < 1e-3 * rec["config"]["windows_per_step_per_gpu"]
    roof, cpu = rec["roofline"], rec["cpu_baseline"]
    assert roof["bound"] == "mfma" and 0.1 < roof["frac"] < 1.0 and roof["kernel_avg_ms"] <= rec["ms_per_step"] and "traffic" in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-6
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
    assert rec["mpv_max_rel_err"] < 1e-4
    assert set(rec["other"]) >= {"mode_f32", "mode_f16x2", "mode_f16", "other_summary_path", "trained_checkpoints", "full_nsdiff"}
    recs = [json.loads(ln) for ln in (tmp_path / "records.jsonl").read_text().splitlines()]
    names = [r["record"] for r in recs]
    assert names[0] == "run" and names[-1] == "headline" and {"mode_f32", "full_nsdiff", "trained_checkpoints"} <= set(names)
    assert sum(ln.startswith("[bench record] ") for ln in proc.stderr.splitlines()) == len(recs)


def test_self_launched_two_ranks_headline(tmp_path):
    """`python bench.py --gpus 2` without a launcher (the SCALE command): two workers, one JSON line relayed from rank 0, both ranks' curves gathered."""
    proc, _ = run_bench(tmp_path, "--gpus", "2", "--backend", "gloo", "--max-windows", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = only_line(proc)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["collective"]["backend"] == "gloo"
    assert rec["config"]["windows_per_step_per_gpu"] == 6 and rec["config"]["truncated_for_profiling"] is True
    # whole-job windows / MAX-over-ranks time (both numbers are printed with six significant digits)
    assert abs(rec["value"] - 2 * 6 * rec["steps"] / (rec["ms_per_step"] * rec["steps"] / 1e3)) < 1e-4 * rec["value"]
    assert "rank 1/2" in proc.stderr and "rank 0/2" in proc.stderr


def test_self_launched_two_ranks_c4_strong_scaling(tmp_path):
    """`--config c4 --gpus 2`: ONE trajectory's windows split in contiguous blocks over the ranks, the curve all-gathered."""
    proc, _ = run_bench(tmp_path, "--config", "c4", "--gpus", "2", "--backend", "gloo", "--max-windows", "5", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline")
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = only_line(proc)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["windows_per_step"] == 5 and rec["config"]["windows_of_rank0"] == 3 and rec["config"]["truncated_for_profiling"] is True


def test_a_dead_worker_fails_the_launch_instead_of_hanging(tmp_path):
    """Rank 1 exits before the rendezvous (MPV_BENCH_FAIL_RANK): the launcher must notice, stop rank 0 (blocked in init_process_group) and return 1."""
    proc, took = run_bench(tmp_path, "--gpus", "2", "--backend", "gloo", "--max-windows", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                           env={"MPV_BENCH_FAIL_RANK": "1", "MPV_BENCH_LAUNCH_TIMEOUT": "240", "MPV_BENCH_DIST_TIMEOUT": "120"}, timeout=400)
    assert proc.returncode == 1, (proc.returncode, proc.stderr[-2000:])
    assert took < 200, "the launcher took %.0f s to give up" % took
    assert "rank 1 exited with code 3" in proc.stderr
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
