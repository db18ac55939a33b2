"""bench.py's N > 1 watchdog (round 5), host logic only: the supervisor follows its workers' progress markers, ends an
attempt whose set-up overruns the budget by killing the workers' process groups, starts FRESH workers once with `--gather rccl`, prints
the successful attempt's JSON line (with its own record of the attempts) -- or an {"error": ...} line and a non-zero code when the second
attempt fails too.  The workers here are tiny stand-in scripts (no GPU, no torch); the real thing is exercised on the GPU box by
tests/test_gpu_parity.py::test_bench_watchdog_falls_back_to_rccl_when_the_peer_self_test_stalls."""
import io
import json
import os
import sys
import time

import bench

WORKER = r'''
import json, os, sys, time
g, rank = sys.argv[1], int(os.environ["RANK"])
def mark(p): print(f"[bench-progress] rank={rank} phase={p} t=0.0", file=sys.stderr, flush=True)
mark("process_group"); mark("handles"); mark("peer_gather")
mode = os.environ.get("FAKE_MODE", "stall_on_peer")
if mode == "always_stall" or (mode == "stall_on_peer" and g != "rccl"):
    if rank == 1:
        time.sleep(1e6)                       # a hung IPC map: never returns, never exits
    time.sleep(1e6)
if mode == "crash" and g != "rccl" and rank == 1:
    sys.exit(9)
mark("warmup"); mark("timed"); mark("report")
print("worker chatter on stderr", file=sys.stderr, flush=True)
if rank == 0:
    print(json.dumps({"metric": "m", "value": 123.0, "gather": g, "n_gpus": 2}), flush=True)
mark("done")
'''


def _run(tmp_path, capsys, mode, gather="auto"):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    log = io.StringIO()
    t0 = time.time()
    rc = bench.supervise([(0, 0), (1, 1)], 2, lambda g, attempt: [sys.executable, str(script), g],
                         lambda r, lr, attempt: dict(os.environ, RANK=str(r), FAKE_MODE=mode), setup_timeout=2.0, run_timeout=5.0,
                         gather=gather, is_printer=True, log=log, grace=0.5)
    out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    return rc, out, log.getvalue(), time.time() - t0


def test_supervisor_falls_back_to_rccl_after_a_stalled_peer_set_up(tmp_path, capsys):
    rc, out, log, dt = _run(tmp_path, capsys, "stall_on_peer")
    assert rc == 0 and len(out) == 1 and dt < 30
    d = json.loads(out[0])
    assert d["value"] == 123.0 and d["gather"] == "rccl"
    att = d["watchdog"]["attempts"]
    assert [a["gather"] for a in att] == ["auto", "rccl"] and att[0]["error"] and att[1]["error"] is None
    assert att[0]["last_phase"] == {"0": "peer_gather", "1": "peer_gather"}
    assert "starting fresh workers with --gather rccl" in log and "worker chatter" in log


def test_supervisor_retries_after_a_crashed_rank_and_passes_a_clean_run_through(tmp_path, capsys):
    rc, out, log, _ = _run(tmp_path, capsys, "crash")
    d = json.loads(out[0])
    assert rc == 0 and "rank 1: 9" in d["watchdog"]["attempts"][0]["error"] and d["gather"] == "rccl"
    rc, out, log, _ = _run(tmp_path, capsys, "ok")
    d = json.loads(out[0])
    assert rc == 0 and d["gather"] == "auto" and len(d["watchdog"]["attempts"]) == 1


def test_supervisor_reports_an_error_line_when_the_fallback_fails_too(tmp_path, capsys):
    rc, out, log, dt = _run(tmp_path, capsys, "always_stall")
    assert rc != 0 and len(out) == 1 and dt < 30
    d = json.loads(out[0])
    assert d["value"] is None and "attempt 1 (--gather auto)" in d["error"] and "attempt 2 (--gather rccl)" in d["error"]
    # nothing of a failed attempt lingers: the stand-in workers (own sessions) were killed with their process groups
    rc, out, log, _ = _run(tmp_path, capsys, "always_stall", gather="rccl")
    assert rc != 0 and len(json.loads(out[0])["watchdog"]["attempts"]) == 1      # already rccl: one attempt only


def test_supervisor_takes_its_workers_with_it_when_it_is_terminated(tmp_path):
    """torchrun ends a failed job by SIGTERM to the processes it started -- the supervisors.  Their workers live in sessions of their own
    (so that a whole attempt can be killed as process groups) and must not outlive the supervisor on the GPU."""
    import signal
    import subprocess
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    pidfile = tmp_path / "pids"
    driver = tmp_path / "driver.py"
    driver.write_text(f"""
import os, sys, subprocess
sys.path.insert(0, {str(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))!r})
import bench
real = subprocess.Popen
def spy(*a, **k):
    p = real(*a, **k)
    open({str(pidfile)!r}, "a").write(str(p.pid) + "\\n")
    return p
subprocess.Popen = spy
bench.supervise([(0, 0), (1, 1)], 2, lambda g, attempt: [sys.executable, {str(script)!r}, g],
                lambda r, lr, attempt: dict(os.environ, RANK=str(r), FAKE_MODE="always_stall"), 60.0, 60.0, "auto", True)
""")
    sup = subprocess.Popen([sys.executable, str(driver)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.time() + 20
    while time.time() < deadline and (not pidfile.exists() or len(pidfile.read_text().split()) < 2):
        time.sleep(0.1)
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    time.sleep(0.5)
    sup.send_signal(signal.SIGTERM)
    sup.wait(timeout=20)
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = open(f"/proc/{pid}/stat").read().split()[2] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid
