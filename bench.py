#!/usr/bin/env python3
"""bench.py -- throughput of the hot path (log-mel front-end + UiT-XS encoder + head) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            # any N: for N > 1 without a torchrun environment the script
                                                                   # starts its own N ranks as CHILD processes (never an exec)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                     # the same, launched from outside: one rank per GPU over RCCL

N > 1 runs under a watchdog (round 5): the process that was launched never touches the GPU -- it SUPERVISES a worker child per rank,
follows the workers' progress markers (RCCL init, peer-gather self-test, warm-up, timed steps), and when the set-up or the run
overruns its wall-clock budget it kills the workers' process groups and starts FRESH workers once more with `--gather rccl`; if that
fails too, rank 0 prints a JSON line with "error" and the exit code is non-zero (a hung IPC map or RCCL bootstrap must not become a
silent driver timeout).

A "step" = one pass of the whole path over one batch of synthetic 16 kHz x 1 s clips already resident in HBM:
uitk_frontend -> (N>1: 256-byte MAX all-reduce of the batch amax) -> uitk_encode -> (N>1: RCCL all-gather of the
[B/N,537] outputs).  Both collectives are asynchronous and overlap the NEXT step's kernels: step n+1's front-end is
enqueued (on a second model replica) before step n's encoder, so its amax all-reduce runs under that encoder, and step
n's gather runs under step n+1's kernels.  Exactly K front-ends, K encoders and K gathers complete inside the timed region.  Workload = BASELINE.json configs[2]: UiT-XS, 8192 clips per GPU (weak scaling: 65536 at N=8 =
configs[3]).  Prints ONE JSON line (rank 0).  The CPU baseline (oracle/torch_port.py, never the thing measured on
the GPU) is timed on rank 0 at N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from uit_mobile_amd.synth import make_state_dict  # noqa: E402
from uit_mobile_amd import models  # noqa: E402
from uit_mobile_amd.sharded import (PeerGather, PeerGatherUnavailable, amax_allreduce_async, amax_allreduce_finish,  # noqa: E402
                                    gather_outputs_async)

GEMM_FLOP = {"uit_xs": 68.66e6, "uit_xxs": 35.18e6, "uit_xxxs": 24.03e6}   # BASELINE.md section 3, per clip
FRONTEND_ALG_FLOP = 1.34e6       # algorithmic fp32 work per clip, mel counted sparse: (11 520 rFFT + 771 power + ~1 000
                                 # banded mel) x 101 frames.  The dense-mel count (4.56 MFLOP) is NOT what the kernel executes.
# What mel_frontend_kernel executes per frame: window 512 + two radix-16 stages 32 x 168 + twiddles 1 440 +
# real-FFT split & power 3 328 on the VALU = 10 656; the banded filterbank 52 steps x 64 mels x 2 = 6 656 on the exact-fp32
# matrix pipe (v_mfma_f32_4x4x1, same FLOP rate as the VALU); log10 / init_bn / convert ~640.  x 101 frames (+5 % dead slots
# of the bundled tail passes).
FRONTEND_EXEC_FLOP = (10656 + 6656 + 640) * 101 * 1.05
PARITY_NOTE = {
    "auto": "the parity mode and the factories' default (round 5): 'f16c' -- single fp16 weights with calibrated rounding -- held to the "
            "library's absolute budget on its held-out clips (6e-4 on logits), else rebuilt on split weights (f16l6 / f16x2); `effective_precision` "
            "says which one ran; all 36 trained-like sweep checkpoints inside the 1e-3 gate on the wide held-out set "
            "(tests/test_gpu_parity.py::test_parity_mode_all_sweep_seeds)",
    "f16c": "the parity mode (round 4): single fp16 weights with calibrated rounding; all 36 trained-like sweep checkpoints inside the "
            "1e-3 gate on held-out inputs (worst 4.4e-4 on logits: tests/test_gpu_parity.py::test_parity_mode_all_sweep_seeds)",
    "f16l6": "round 3's parity mode: inside the 1e-3 gate on all 36 trained-like sweep checkpoints (worst 6.0e-4) and every fixture",
    "f16": "single fp16 weights rounded to nearest: inside 1e-3 at init scale and up to |logit| ~ 12; 0.6 .. 1.2e-3 at |logit| 13-15",
    "f16x2": "inside the 1e-3 gate with 2-3x the margin of f16 (every weight as two fp16 terms: the reference point of f16l6)",
    "f32": "the last rung of the parity ladder (round 5): the library's plain fp32 forward on the vector ALUs behind the product front-end -- where "
           "'auto' ends when no fp16 mode is inside the budget; 2e-6 from an fp64 evaluation of the same mel (tools/fe_split.py)",
    "bf16": "outside the 1e-3 gate at trained magnitudes (measured: tests/test_gpu_parity.py)",
    "fp8": "outside the 1e-3 gate (measured: tests/test_gpu_parity.py); weights AND activations e4m3, weights rounded to nearest",
    "fp8c": "BASELINE config #5 on the fp8 MFMA: as fp8, the weights GPTQ-rounded to the MX e4m3 grid (round 5); outside the 1e-3 gate, measured",
    "fp8w": "BASELINE config #5 read literally (fp8 WEIGHTS): qkv / fc1 / fc2 weights GPTQ-rounded to the MX e4m3 grid, expanded to fp16 at pack "
            "time -> the f16 kernel on fp16 activations; outside the 1e-3 gate, measured",
}
PEAK_MFMA_TFLOPS = 2500.0        # dense bf16/f16, MI355X_MICROARCH.md chip table
PEAK_MFMA_FP8_TFLOPS = 5000.0    # dense MX-scaled fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), same table
PEAK_HBM_GBS = 8000.0
PEAK_VALU_TFLOPS = 157.3


def synth_batch(B, T, first_clip, device, i16):
    """Device-resident synthetic PCM, a function of the GLOBAL clip index only (=> independent of the rank count).
    Uniform noise in [-0.5,0.5); every 8th clip a loud tone, every 8th+2 silent over its last quarter, 8th+3 quiet."""
    out = torch.empty((B, T), device=device, dtype=torch.int16)
    t = torch.arange(T, device=device, dtype=torch.float32)
    CH = 512
    for lo in range(0, B, CH):
        n = min(CH, B - lo)
        g = torch.Generator(device=device)
        g.manual_seed(977 + (first_clip + lo) // CH)
        x = torch.rand((n, T), device=device, generator=g) - 0.5
        idx = torch.arange(first_clip + lo, first_clip + lo + n, device=device)
        tone = 0.9 * torch.sin(2 * np.pi * (440.0 + 37.0 * (idx % 64).float())[:, None] * t[None] / 16000.0)
        x = torch.where((idx % 8 == 1)[:, None], tone, x)
        x[:, T - T // 4:] = torch.where((idx % 8 == 2)[:, None], torch.zeros((), device=device), x[:, T - T // 4:])
        x = torch.where((idx % 8 == 3)[:, None], x * 0.001, x)
        out[lo:lo + n] = torch.clamp(torch.round(x * 32768.0), -32768, 32767).to(torch.int16)
    return out if i16 else out.float() / 32768.0


def kernel_sources_sha16():
    """identity of the kernel sources a PMC figure belongs to (csrc/*.hip + the layout header)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("encoder.hip", "frontend.hip", "epilogue.hip", "uitk_dev.h", "pack.cpp"):
        h.update((ROOT / "uit_mobile_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_C_LEG = """
import sys, time
sys.path[:0] = [{root!r}]
import numpy as np
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
from oracle.oracle import Oracle
o = Oracle(make_state_dict({model!r}, 13))
x = synth_pcm_i16({B}, {T}, seed=3)
o.forward(x[:8])
n, t0 = 0, time.perf_counter()
while True:
    o.forward(x); n += {B}
    el = time.perf_counter() - t0
    if el >= {budget}: break
print(n / el, n, el)
"""


_PROC_LEG = """
import sys, time
sys.path[:0] = [{root!r}]
import numpy as np, torch
torch.set_num_threads(1)
from uit_mobile_amd.synth import make_state_dict, synth_pcm_i16
from oracle.torch_port import TorchPort
tp = TorchPort(make_state_dict({model!r}, 13))
x = torch.from_numpy(synth_pcm_i16({B}, {T}, seed=3 + {seed}).astype(np.float32) / 32768.0)
tp.forward(x[:8])
print("ready", flush=True)
sys.stdin.readline()                      # all workers start together
n, t0 = 0, time.perf_counter()
while True:
    tp.forward(x); n += {B}
    el = time.perf_counter() - t0
    if el >= {budget}: break
print(n, el, flush=True)
"""


def cpu_process_parallel_leg(model_name, T, nproc, budget_s, batch=64):
    """The fair many-core statement of the CPU path: `nproc` independent worker PROCESSES, one torch thread each (torch's
    intra-op pool does not scale at these GEMM shapes: 64 threads measured 1.07x one thread), every worker looping the same
    op-for-op port over its own batch; all released together, rate = clips of all workers / the slowest worker's time."""
    import subprocess
    procs = []
    for i in range(nproc):
        code = _PROC_LEG.format(root=str(ROOT), model=model_name, B=batch, T=T, seed=i,
                                budget=budget_s)
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                      env=dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")))
    try:
        for pr in procs:
            if pr.stdout.readline().strip() != "ready":
                raise RuntimeError("worker failed to start")
        for pr in procs:
            pr.stdin.write("go\n"); pr.stdin.flush()
        res = [tuple(float(v) for v in pr.stdout.readline().split()) for pr in procs]
        n, el = sum(r[0] for r in res), max(r[1] for r in res)
        return {"impl": "torch_port", "batch": batch, "processes": nproc, "threads": 1, "clips_per_s": round(n / el, 1), "clips": int(n),
                "seconds": round(el, 2)}
    finally:
        for pr in procs:
            try:
                pr.kill()
            except Exception:
                pass


def cpu_baseline(model_name, sd, T, budget_s):
    """The reference CPU path timed on this host: (1) oracle/torch_port.py -- the same ATen CPU operators the
    reference executes, op for op -- and (2) the plain-C restatement oracle/uit_oracle.c (OpenMP), each at B = 64 and 1024, one
    thread and many.  torch's intra-op pool thrashes on many-core hosts at these tiny GEMM shapes, so a short probe picks the
    best thread count out of {8, 16, 32, 64, all}; `cores` is the thread count of the headline (fastest) leg.  Every leg is
    bounded (`budget_s` split over the legs); the C legs run in a child process so that OMP_NUM_THREADS applies."""
    import subprocess
    from oracle.torch_port import TorchPort
    ncpu = os.cpu_count() or 1
    tp = TorchPort(sd)
    wave = synth_batch(1024, T, 0, "cpu", False)
    best, best_rate = 1, 0.0
    for nt in sorted({min(n, ncpu) for n in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(nt)
        tp.forward(wave[:64])
        t0 = time.perf_counter()
        tp.forward(wave[:128])
        rate = 128 / (time.perf_counter() - t0)
        if rate > best_rate:
            best, best_rate = nt, rate
        if time.perf_counter() - t0 > 5.0:
            break
    leg_budget = max(budget_s / 8.0, 1.0)
    legs = []
    for impl in ("torch_port", "c_restatement"):
        for Bc in (64, 1024):
            for nt in (1, best if impl == "torch_port" else ncpu):
                if impl == "torch_port":
                    torch.set_num_threads(nt)
                    tp.forward(wave[:min(Bc, 64)])
                    n, t0 = 0, time.perf_counter()
                    while True:
                        tp.forward(wave[:Bc])
                        n += Bc
                        el = time.perf_counter() - t0
                        if el >= leg_budget:
                            break
                    rate = n / el
                else:
                    if nt == 1 and Bc > 64:
                        continue            # one batch of 1024 on one thread of the scalar C loops alone takes ~25 s
                    env = dict(os.environ, OMP_NUM_THREADS=str(nt))
                    code = _C_LEG.format(root=str(ROOT), model=model_name, B=Bc, T=T,
                                         budget=leg_budget)
                    try:
                        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
                        rate, n, el = (float(v) for v in out.stdout.split()[-3:])
                    except Exception as e:      # the baseline is reported, never required
                        legs.append({"impl": impl, "batch": Bc, "threads": nt, "error": str(e)[:80]})
                        continue
                legs.append({"impl": impl, "batch": Bc, "threads": nt, "clips_per_s": round(rate, 1), "clips": int(n),
                             "seconds": round(el, 2)})
    torch.set_num_threads(best)
    # headline: N single-thread worker processes (N = physical-core-sized, bounded so that the default bench run stays short)
    nproc = max(1, min(ncpu // 2 if ncpu >= 16 else ncpu, 64))
    head = None
    try:
        head = cpu_process_parallel_leg(model_name, T, nproc, max(budget_s / 4.0, 2.0))
        legs.append(head)
    except Exception as e:              # the baseline is reported, never required
        legs.append({"impl": "torch_port", "processes": nproc, "error": str(e)[:80]})
    if head is None:
        head = max((l for l in legs if l.get("impl") == "torch_port" and "clips_per_s" in l), key=lambda l: l["clips_per_s"])
        head = dict(head, processes=1)
    cores = head.get("processes", 1) * head["threads"]
    return {"value": head["clips_per_s"], "unit": "clips/s", "cores": cores, "kind": "port", "host_cpus": ncpu,
            "cpu_model": cpu_model_string(),
            "sample": f"{head['clips']} synthetic 1 s clips (batches of {head['batch']}) through oracle/torch_port.py "
                      f"(torch {torch.__version__} CPU fp32, same ATen ops as the reference), {head['seconds']} s, "
                      f"{head.get('processes', 1)} worker processes x {head['threads']} torch thread(s); the single-process legs "
                      f"(1 thread / best-of-probe threads, B = 64 / 1024, torch port and C restatement) are in `legs`",
            "legs": legs}


# ---------------------------------------------------------------------------------------------------------------- N > 1: watchdog
PHASES_SETUP = ("start", "process_group", "handles", "peer_gather", "collectives", "spinup")      # then "warmup", "timed", "report", "done"
_phase_state = {"name": "start", "t0": time.perf_counter(), "deadline": None, "rank": 0}


def phase(name, deadline_s=None):
    """Worker side: progress marker on stderr (the supervisor follows these) + the wall-clock budget of the phases up to the next marker."""
    _phase_state["name"] = name
    if deadline_s is not None:
        _phase_state["deadline"] = time.perf_counter() + deadline_s
    print(f"[bench-progress] rank={_phase_state['rank']} phase={name} t={time.perf_counter() - _phase_state['t0']:.1f}", file=sys.stderr, flush=True)


def start_worker_watchdog(rank):
    """Worker side: a daemon thread that ends the process (exit code 75) when the current phase overruns its budget -- a hung
    hipIpcOpenMemHandle / RCCL bootstrap / peer copy cannot be cancelled from inside, only left behind."""
    import threading
    _phase_state["rank"] = rank

    def run():
        while True:
            time.sleep(0.5)
            d = _phase_state["deadline"]
            if d is not None and time.perf_counter() > d and _phase_state["name"] != "done":
                print(f"[bench-progress] rank={rank} timeout phase={_phase_state['name']} t={time.perf_counter() - _phase_state['t0']:.1f}",
                      file=sys.stderr, flush=True)
                os._exit(75)
    threading.Thread(target=run, daemon=True).start()


def supervise(ranks, world, worker_argv, env_for, setup_timeout, run_timeout, gather, is_printer, log=sys.stderr, grace=20.0):
    """Supervisor side (no GPU call in this process, ever).  `ranks`: the (rank, local_rank) pairs THIS process is responsible for -- all
    of them when `python bench.py --gpus N` was typed, its own one under an external torchrun.  Starts one worker per rank in its own
    session (= process group), follows the [bench-progress] markers on their stderr, relays everything else.  A worker that exits
    non-zero, or set-up / run overrunning the budget (+ `grace` seconds: the workers' own watchdogs fire first), ends the attempt: every worker's process group is killed, and ONE more attempt is
    made with fresh workers and `--gather rccl` (if the first was not already that).  Returns the exit code; the rank-0 worker's JSON
    line is printed only for the attempt that succeeded, an {"error": ...} line otherwise (is_printer: this supervisor owns rank 0)."""
    import signal
    import subprocess
    import threading
    attempts = []
    gathers = [gather] + (["rccl"] if gather != "rccl" else [])
    live = []                                               # worker processes of the running attempt (each its own session / process group)

    def on_term(signum, frame):                             # torchrun (or a user) ends the supervisor: its workers must not outlive it on the GPU
        for pr in live:
            try:
                os.killpg(pr.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
        os._exit(128 + signum)
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            signal.signal(sg, on_term)
    for attempt, g in enumerate(gathers):
        procs, last, out_lines, timed_out = [], {}, [], []
        t_start = time.perf_counter()

        def pump_err(pr, r):
            for ln in pr.stderr:
                if ln.startswith("[bench-progress]"):
                    kv = dict(f.split("=", 1) for f in ln.split()[1:] if "=" in f)
                    last[r] = (kv.get("phase", "?"), time.perf_counter() - t_start, "timeout" in ln.split()[2:3])
                else:
                    log.write(ln); log.flush()

        def pump_out(pr, keep):
            for ln in pr.stdout:                             # (every worker's stdout is drained; the first rank's is what is kept)
                if keep:
                    out_lines.append(ln)

        threads = []
        for r, lr in ranks:
            pr = subprocess.Popen(worker_argv(g, attempt), env=env_for(r, lr, attempt), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            procs.append((r, pr))
            live.append(pr)
            last[r] = ("spawned", 0.0, False)
            for fn, a in ((pump_err, (pr, r)), (pump_out, (pr, r == ranks[0][0]))):
                th = threading.Thread(target=fn, args=a, daemon=True)
                th.start()
                threads.append(th)
        why = None
        while True:
            time.sleep(0.2)
            codes = [pr.poll() for _, pr in procs]
            if any(c not in (None, 0) for c in codes):
                bad = [(r, c) for (r, _), c in zip(procs, codes) if c not in (None, 0)]
                why = "worker exit codes " + ", ".join(f"rank {r}: {c}" for r, c in bad)
                break
            if all(c == 0 for c in codes):
                break
            now = time.perf_counter() - t_start
            in_setup = any(last[r][0] in PHASES_SETUP + ("spawned",) for r, _ in procs)
            # (a retry's set-up budget is doubled: under an external torchrun the supervisors of the ranks do not talk to each other, and
            # the ones whose first worker died early must wait for the ones whose first worker ran into its deadline)
            st = setup_timeout * (attempt + 1)
            if in_setup and now > st + grace:
                why = f"set-up still running after {now:.0f} s (budget {st:.0f} s)"
                break
            if now > st + run_timeout + 2.0 * grace:
                why = f"run still going after {now:.0f} s (budget {st:.0f} + {run_timeout:.0f} s)"
                break
        if why is not None:                                  # end the attempt: whole process groups, nothing of it may linger on the GPU
            for _, pr in procs:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
            for _, pr in procs:
                try:
                    pr.wait(timeout=30)
                except Exception:
                    pass
        for th in threads:
            th.join(timeout=5)
        del live[:]
        rec = {"gather": g, "seconds": round(time.perf_counter() - t_start, 1), "error": why,
               "last_phase": {str(r): last[r][0] + (" (its own watchdog fired)" if last[r][2] else "") for r, _ in procs},
               "last_phase_at_s": {str(r): round(last[r][1], 1) for r, _ in procs}}
        attempts.append(rec)
        if why is None:
            if is_printer:
                js = [ln for ln in out_lines if ln.lstrip().startswith("{")]
                for ln in out_lines:
                    if ln not in js[-1:]:
                        log.write(ln)
                if js:
                    try:                                      # the one JSON line, with the supervisor's record of the attempts
                        line = json.loads(js[-1])
                        line["watchdog"] = {"attempts": attempts, "setup_timeout_s": setup_timeout, "run_timeout_s": run_timeout}
                        print(json.dumps(line), flush=True)
                    except ValueError:
                        sys.stdout.write(js[-1]); sys.stdout.flush()
                else:
                    print(json.dumps({"metric": "1s-clips/sec/node (mel+UiT-XS fwd)", "value": None, "n_gpus": world,
                                      "error": "workers exited 0 without a JSON line", "watchdog": {"attempts": attempts}}), flush=True)
                    return 1
            return 0
        print(f"bench watchdog: attempt {attempt + 1} (--gather {g}) failed: {why}; last phases {rec['last_phase']}"
              + ("; starting fresh workers with --gather rccl" if attempt + 1 < len(gathers) else ""), file=log, flush=True)
    if is_printer:
        print(json.dumps({"metric": "1s-clips/sec/node (mel+UiT-XS fwd)", "value": None, "unit": "clips/s", "n_gpus": world,
                          "error": "; ".join(f"attempt {i + 1} (--gather {a['gather']}): {a['error']}" for i, a in enumerate(attempts)),
                          "watchdog": {"attempts": attempts, "setup_timeout_s": setup_timeout, "run_timeout_s": run_timeout}}), flush=True)
    return 1


def supervise_from_args(args):
    """N > 1, this process is not a worker: supervise the workers of the ranks this process stands for."""
    import socket
    external = "WORLD_SIZE" in os.environ                    # launched by torchrun: one supervisor per rank, each with ONE worker
    if external:
        world, rank, lr = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
        ranks, port0 = [(rank, lr)], int(os.environ.get("MASTER_PORT", "29500"))
    else:
        world, ranks = args.gpus, [(r, r) for r in range(args.gpus)]
        # one rendezvous port per attempt (port0 + attempt below): a pair of adjacent ports that are both free right now
        for _ in range(64):
            with socket.socket() as sk, socket.socket() as sk2:
                sk.bind(("127.0.0.1", 0))
                port0 = sk.getsockname()[1]
                try:
                    sk2.bind(("127.0.0.1", port0 + 1))
                    break
                except OSError:
                    continue
    argv_base = [a for a in sys.argv[1:]]

    def worker_argv(g, attempt):
        out, skip = [], False
        for a in argv_base:                                  # replace --gather / --setup-timeout with this attempt's
            if skip:
                skip = False
                continue
            if a in ("--gather", "--setup-timeout"):
                skip = True
                continue
            if a.startswith("--gather=") or a.startswith("--setup-timeout="):
                continue
            out.append(a)
        return [sys.executable, str(Path(__file__).resolve()), "--worker", "--gather", g, "--setup-timeout", str(args.setup_timeout * (attempt + 1))] + out

    def env_for(r, lr, attempt):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(lr), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if external and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true":
            # the agent's store outlives the workers: a fresh key namespace per attempt (torch's env:// rendezvous prefixes its keys
            # with the restart count)
            env["TORCHELASTIC_RESTART_COUNT"] = str(int(os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")) + attempt)
            env["MASTER_PORT"] = str(port0)
        else:
            env["MASTER_PORT"] = str(port0 + attempt)       # rank 0's worker hosts the store: a new port per attempt (same on every rank)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        return env

    return supervise(ranks, world, worker_argv, env_for, args.setup_timeout, args.run_timeout, args.gather, is_printer=ranks[0][0] == 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)        # 0.2 s of GPU time: past the clock ramp of the first ~50 steps
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--model", default="uit_xs", choices=list(GEMM_FLOP))
    ap.add_argument("--batch-per-gpu", type=int, default=8192)
    # every precision of models.uit._PRECISIONS that runs the bench workload (pooling 'mean'); 'f16a' / 'f16ax' are built for pooling 'token'
    # only (tools/variant_bench.py times them)
    ap.add_argument("--precision", default="auto", choices=list(PARITY_NOTE),
                    help="auto (default) = THE parity mode and the factories' default: the fp16 kernel on weights with calibrated rounding "
                         "(uitk_calibrate_weights on the library's own signals), held to the library's absolute budget on its held-out clips, "
                         "else rebuilt on split weights -- the line's `effective_precision` says what ran; f16c = the same without the "
                         "escalation; f16 = the same kernel on weights rounded to nearest; f16l6 = round 3's parity mode")
    ap.add_argument("--pcm16", action="store_true", help="feed int16 PCM instead of float32 waves")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-dtypes", action="store_true", help="skip the f16 / bf16 / fp8 / f16x2 sub-records (N=1 only)")
    ap.add_argument("--input-buffers", type=int, default=3,
                    help="distinct synthetic input batches rotated step by step (>= 3: 3 x 524 MB of float32 waves cannot be served "
                         "by the 256 MB Infinity Cache, so the front-end really reads its PCM from HBM)")
    ap.add_argument("--spinup-steps", type=int, default=0,
                    help="extra untimed steps before the --warmup steps (default 0: the set-up itself -- weight calibration, the "
                         "front-end's ~140 tuning launches -- already leaves the device at its working clocks)")
    ap.add_argument("--gather", default="auto", choices=["auto", "rccl", "peer"],
                    help="N>1: 'peer' = copy-engine fan-out into IPC-mapped peer buffers (no CU, all xGMI links at once), 'rccl' = "
                         "all_gather_into_tensor; 'auto' (default) = peer when its IPC self-check passes on every rank, else rccl")
    ap.add_argument("--gloo-device-tensors", action="store_true", help=argparse.SUPPRESS)   # lab: the pre-hardening gloo path (tools/wd_repeat.sh)
    ap.add_argument("--gather-dtype", default="f32", choices=["f32", "f16"],
                    help="N>1: payload of the gather: fp32 probabilities (2 148 B per clip; default -- what the single-GPU path and the "
                         "reference deliver) or, opt-in, float16 (1 074 B per clip, ~2.4e-4 rounding on probabilities near 1)")
    ap.add_argument("--gather-topk", type=int, default=0,
                    help="N>1: gather only the device-side top-k (uitk_topk before the collective: 8 k bytes per clip) instead of [B,537]")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL; gloo only for exercising the N>1 code path on a 1-GPU box)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="test hook: every rank uses cuda:0 (with --backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--host-producers", type=int, default=8, help="--from-host: producer threads filling pinned batches")
    ap.add_argument("--worker", action="store_true", help="internal: this process IS a rank (started by the supervisor of an N>1 run)")
    ap.add_argument("--setup-timeout", type=float, default=420.0,
                    help="N>1 watchdog: wall-clock budget (s) for a worker's set-up -- import, RCCL init, handles, peer-gather self-test")
    ap.add_argument("--run-timeout", type=float, default=600.0, help="N>1 watchdog: wall-clock budget (s) for warm-up + timed steps + report")
    ap.add_argument("--inject-stall", default="", help="test hook: 'PHASE' or 'PHASE:RANK' -- that rank sleeps forever when it reaches PHASE "
                                                        "with --gather auto / peer (the watchdog test)")
    ap.add_argument("--from-host", action="store_true",
                    help="additionally report the PCIe-inclusive rate (pinned int16 host buffers -> HostClipStreamer)")
    args = ap.parse_args()

    if args.gpus > 1 and not args.worker:
        # N > 1: this process only supervises (it never touches the GPU); the ranks are its worker children -- see supervise()
        raise SystemExit(supervise_from_args(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to benchmark)")
    if args.all_ranks_on_device0:
        local_rank = 0
    if args.gloo_device_tensors:
        import uit_mobile_amd.sharded as _sh
        _sh._HOST_STAGED = False
    if world > 1:
        start_worker_watchdog(rank)
        phase("process_group", args.setup_timeout)          # one budget for the whole set-up, a second one from "warmup" on

    def maybe_stall(name):
        if args.inject_stall and args.gather in ("auto", "peer"):
            ph, _, rk = args.inject_stall.partition(":")
            if ph == name and (rk == "" or int(rk) == rank):
                time.sleep(1e6)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    if world > 1:
        phase("handles")
    T, Bl = 16000, args.batch_per_gpu
    sd = make_state_dict(args.model, 13)
    # two replicas of the model (weights 3 MB, workspace 100 MB each): batch n+1's front-end is enqueued on the other
    # replica BEFORE batch n's encoder, so that at N>1 the 256-byte amax all-reduce of batch n+1 (RCCL's stream) runs
    # under batch n's encoder instead of idling the GPU between the two kernels.  Same kernels, same order per batch.
    reps = []
    for _ in range(2):
        r = getattr(models, args.model)(outputdim=537, target_length=102, precision=args.precision)
        r.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        reps.append(r.to(dev).eval())
    m = reps[0]
    nbuf_in = max(1, args.input_buffers)
    xs = [synth_batch(Bl, T, (rank + i * world) * Bl, dev, args.pcm16) for i in range(nbuf_in)]
    x = xs[0]
    # HIP events on the launch stream (torch's current stream is the one handed to the C ABI), four per step:
    # around the front-end kernel and around the encoder kernel.
    n_ev = args.warmup + args.steps
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_ev)]
    cur = [0]
    import collections
    pending = collections.deque()
    # what crosses xGMI per clip: fp32 / float16 probabilities, or the packed device-side top-k (values bit-cast next to indices)
    if args.gather_topk > 0:
        from uit_mobile_amd.postprocess import topk as dev_topk
        pay_cols, pay_dtype = 2 * args.gather_topk, torch.int32

        def payload(probs):
            v, i = dev_topk(probs, args.gather_topk)
            return torch.cat([v.contiguous().view(torch.int32), i.to(torch.int32)], 1)
    else:
        pay_cols, pay_dtype = 537, {"f32": torch.float32, "f16": torch.float16}[args.gather_dtype]
    # 4 rotating buffers: at most 2 gathers are in flight and one result is being read.  --gather auto: the peer fan-out when its IPC
    # self-check passes on EVERY rank (PeerGather raises on all ranks together otherwise), else RCCL with the halved payload.
    peer, gather_kind = None, "none (1 rank)"
    if world > 1:
        gather_kind = "rccl"
        phase("peer_gather")
        maybe_stall("peer_gather")
        if args.gather in ("auto", "peer"):
            try:
                peer = PeerGather(Bl, pay_cols, dev, dtype=pay_dtype, nbuf=4)
                peer.self_test()                           # a known pattern through every buffer, verified on every rank
                gather_kind = "peer"
            except PeerGatherUnavailable as e:
                peer = None
                if args.gather == "peer":
                    raise
                if rank == 0:
                    print(f"bench: peer gather unavailable ({e}); using RCCL", file=sys.stderr)
    if args.gather_topk <= 0:
        payload = (lambda probs: probs.to(torch.float16)) if pay_dtype == torch.float16 else (lambda probs: probs)

    def fe(n, rec=True):
        """front-end of step n (+ at N>1 the start of its amax all-reduce)"""
        r = reps[n & 1]
        if rec:
            evs[n][0].record()
        st = r.frontend(xs[n % nbuf_in])
        if rec:
            evs[n][1].record()
        red = amax_allreduce_async(r.amax_tensor(dev)) if world > 1 else None
        return st, red

    def enc(n, st, red, rec=True):
        """encoder of step n; its gather is asynchronous: at most 2 in flight, all complete inside the timed region"""
        r = reps[n & 1]
        if red is not None:
            amax_allreduce_finish(red)
        if rec:
            evs[n][2].record()
        probs = r.encode(st)
        if rec:
            evs[n][3].record()
        send = payload(probs)
        pending.append(peer.gather_async(send) if peer is not None else gather_outputs_async(send))
        return pending.popleft().wait() if len(pending) > 2 else None

    def run(first, count, rec=True):
        """`count` whole steps starting at event slot `first`: exactly `count` front-ends and `count` encoders"""
        nxt = fe(first, rec)
        for n in range(first, first + count):
            this, nxt = nxt, (fe(n + 1, rec) if n + 1 < first + count else None)
            enc(n, *this, rec)

    def drain():
        out = None
        while pending:
            out = pending.popleft().wait()
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, not steps: library handles, workspace of the benchmark's batch size and (N>1) the RCCL communicator, so that
    # neither the W warm-up steps nor the K timed steps contain a hipMalloc or a communicator bootstrap
    with torch.no_grad():
        for r in reps:
            # model.prepare (round 5): the handle (auto / f16c: weight calibration + acceptance check + a possible escalation), the
            # workspace, the front-end's pass split for this shape (measured once per process) and one warm-up forward
            r.prepare(dev, sample=x)                          # (tuned and warmed up on the benchmark's own batch)
        if world > 1:
            phase("collectives")
            amax_allreduce_finish(amax_allreduce_async(torch.zeros(64, device=dev)))
            gather_outputs_async(torch.zeros((64, 537), device=dev)).wait()
            phase("spinup")
        # device spin-up (disclosed in the JSON line as config.setup): the GPU idles while the synthetic batch and the
        # models are built, and its clocks take ~100 ms of load to ramp; a short --warmup would otherwise time the ramp
        run(0, args.spinup_steps, rec=False)
        drain()
    barrier()
    if world > 1:
        phase("warmup", args.run_timeout)
    with torch.no_grad():
        if args.warmup > 0:
            run(0, args.warmup)
        drain()
        barrier()
        if world > 1:
            phase("timed")
        t0 = time.perf_counter()
        run(args.warmup, args.steps)
        out = drain()                                      # every gather of the timed steps has completed ...
        barrier()                                          # ... on every rank, inside the timed region
        el = time.perf_counter() - t0
    if world > 1:
        phase("report")
    # per-kernel durations of the K timed steps, read after the closing synchronize
    fe_ms = [e[0].elapsed_time(e[1]) for e in evs[args.warmup:]]
    enc_ms = [e[2].elapsed_time(e[3]) for e in evs[args.warmup:]]
    period_ms = [a[3].elapsed_time(b[3]) for a, b in zip(evs[args.warmup:-1], evs[args.warmup + 1:])]      # encoder end to encoder end
    assert out.shape == (world * Bl, pay_cols) and (pay_dtype == torch.int32 or bool(torch.isfinite(out.float()).all()))
    # parity of what was just timed (rank 0; the checker, outside the timed region): 32 rows of the LAST timed step's output against
    # the CPU oracle on the same clips.  The subset holds the batch's loudest clip (found with the product's own mel), so the oracle
    # computes the batch's top_db floor; the loud tones repeat with period 64 in the global clip index, i.e. on every rank.
    parity = None
    if rank == 0 and args.gather_topk <= 0:
        try:
            from oracle.oracle import Oracle
            last_x = xs[(args.warmup + args.steps - 1) % nbuf_in]
            tones = torch.arange(1, min(Bl, 1024), 8, device=dev)
            with torch.no_grad():
                loud = int(tones[reps[0].melspec_db(last_x[tones]).amax(dim=(1, 2)).argmax()])
            rows = sorted({loud, 0, 2, 3, Bl // 3, Bl // 2 + 1, Bl - 2, Bl - 1} | {int(i) for i in np.linspace(5, Bl - 6, 24)})    # 32 rows: both
            # rounds of encoder workgroups, every wave slot, the loudest clip (sets the top_db floor) and the first / last rows
            sub = last_x[rows].cpu().numpy()
            ref = Oracle(sd).forward(sub)
            got = out[rows].float().cpu().numpy()
            dev_rows = np.abs(got - ref["probs"]).max(1)
            parity = {"max_abs_dprob": float(dev_rows.max()), "rows": rows,
                      "what": f"{len(rows)} rows of the last timed step's gathered output vs oracle/uit_oracle.c on the same clips"}
            if parity["max_abs_dprob"] > 1e-3:             # say WHICH rows (a diagnosis, not a pass): row -> deviation
                parity["rows_over_1e-3"] = {int(r): float(d) for r, d in zip(rows, dev_rows) if d > 1e-3}
                # ... and WHAT they hold: the same rows computed again, locally, from every input buffer -- a gathered row that matches
                # another buffer's recomputation is another step's output (transport / ordering), one that matches none is a compute error
                try:
                    with torch.no_grad():
                        again = [reps[0](xb).float().cpu().numpy()[rows] for xb in xs]
                    last = (args.warmup + args.steps - 1) % nbuf_in
                    parity["recomputed_last_vs_oracle"] = float(np.abs(again[last] - ref["probs"]).max())
                    parity["gathered_vs_recomputed_buffer"] = {str(k): float(np.abs(got - a).max()) for k, a in enumerate(again)}
                    parity["last_buffer"] = last
                except Exception as e2:
                    parity["recompute_error"] = str(e2)[:200]
        except Exception as e:                              # the check is reported, never silently skipped
            parity = {"max_abs_dprob": None, "error": str(e)[:200]}
    t = torch.tensor([el], device=dev, dtype=torch.float64)
    devices = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device": dev.index, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
                "effective_precision": reps[0].effective_precision(dev)}
        devices = [None] * world
        dist.all_gather_object(devices, mine)
    el = float(t.item())
    eff = reps[0].effective_precision(dev)                 # what 'auto' resolved to on this rank ('f16c' unless the budget check escalated)

    if rank == 0:
        ms = el / args.steps * 1e3
        value = world * Bl * args.steps / el
        enc = float(np.mean(enc_ms))
        fe = float(np.mean(fe_ms))
        in_bytes = 2 if args.pcm16 else 4
        achieved = Bl * GEMM_FLOP[args.model] / (enc * 1e-3) / 1e12
        # precision fp8 (config #5): qkv / fc1 / fc2 (94 % of the GEMM FLOPs) run on the MX fp8 MFMA -> priced against its peak
        peak = PEAK_MFMA_FP8_TFLOPS if eff in ("fp8", "fp8c") else PEAK_MFMA_TFLOPS
        # HBM bytes per encoder launch: measured with rocprofv3 --pmc (tools/prof.sh, separate passes, corrected as MI355X_MICROARCH.md
        # prescribes) and committed with the hash of the kernel sources it was measured on; a figure measured on OTHER sources
        # (or on another precision) is not quoted.
        traffic, traffic_source = None, None
        pmc = ROOT / "profiles" / "pmc_latest.json"
        if pmc.exists():
            try:
                rec_ = json.loads(pmc.read_text())
                src_now = kernel_sources_sha16()
                wl_now = {"model": args.model, "batch_per_gpu": Bl}
                if rec_.get("workload", {"model": "uit_xs", "batch_per_gpu": 8192}) != wl_now:
                    traffic_source = (f"not quoted: profiles/pmc_latest.json was measured on "
                                      f"{rec_.get('workload', {'model': 'uit_xs', 'batch_per_gpu': 8192})}, this run is {wl_now}")
                elif rec_.get("kernel_sources_sha16") == src_now and rec_.get("precision", "f16") in (args.precision, eff):
                    traffic = rec_.get("encoder_hbm_bytes_per_launch")
                    traffic_source = f"profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, kernel sources {src_now})"
                else:
                    traffic_source = (f"not quoted: profiles/pmc_latest.json was measured on kernel sources "
                                      f"{rec_.get('kernel_sources_sha16')} / precision {rec_.get('precision', 'f16')}, this run is "
                                      f"{src_now} / {args.precision} (re-run tools/prof.sh)")
            except Exception:
                traffic = None
        line = {
            "metric": "1s-clips/sec/node (mel+UiT-XS fwd)", "value": round(value, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic type: fp16 MFMA operands (fp32 accumulate / residual stream / LayerNorm / softmax); `precision` names the mode
            "dtype": {"f16c": "f16", "f16l6": "f16l6", "f16x2": "f16x2", "fp8w": "f16 (fp8-grid weights)"}.get(eff, eff), "precision": args.precision,
            "effective_precision": eff,
            "max_abs_dprob": None if parity is None else parity.get("max_abs_dprob"), "parity_check": parity,
            # what the library's own acceptance check measured when it calibrated this handle's weights (logits, held-out clips of the
            # library, against the unrounded network; include/uitk.h: uitk_get_calib_check) -- None for the uncalibrated precisions
            "calibration_check": (lambda r: None if r is None else {k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in r.items()})(
                reps[0].calibration_report() if args.precision in ("auto", "f16c") else None),
            "step_ms_median": round(float(np.median(period_ms)), 4) if period_ms else None,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1, "rank_devices": devices,
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and args.backend == "nccl" else None,
            "gather": gather_kind if world > 1 else None,
            "data": "synthetic",
            "config": {"workload": f"{args.model} (UiT-XS family, seeded random-init weights), {Bl} synthetic "
                                   f"16 kHz x 1 s clips per GPU ({'int16 PCM' if args.pcm16 else 'float32 wave'} "
                                   f"resident in HBM), global batch {world * Bl}",
                       "global_batch": world * Bl, "clip_samples": T, "input_buffers": nbuf_in,
                       "weights": PARITY_NOTE[args.precision] + (
                           " -- NOT in this run: the library's acceptance check refused the calibrated image, the handle runs nearest-rounded fp16 weights"
                           if args.precision == "f16c" and (reps[0].calibration_report() or {}).get("accepted") is False else "")
                                  + (f" -- this run ESCALATED to '{eff}': {reps[0].calibration_report().get('reason')}"
                                     if args.precision == "auto" and eff not in ("f16c", "f16") else ""),
                       "setup": "handles (weight calibration) + workspace + front-end pass-split tuning + RCCL communicator"
                                + (f", then {args.spinup_steps} extra untimed steps" if args.spinup_steps else ""),
                       "parallelism": f"dp{world} (batch shards, in-place RCCL MAX all-reduce of the 64 amax slots + "
                                      f"{'copy-engine peer fan-out' if peer is not None else 'RCCL all-gather'} of "
                                      f"{'the device-side top-%d' % args.gather_topk if args.gather_topk else '[B,537] ' + ('f16' if pay_dtype == torch.float16 else 'f32')}; "
                                      f"{pay_cols * (2 if pay_dtype == torch.float16 else 4)} B per clip cross xGMI)"},
            "roofline": {"bound": "mfma", "kernel": "uit_encoder_kernel", "achieved": round(achieved, 2),
                         "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 5),
                         "traffic": traffic, "traffic_source": traffic_source, "launch_ms": round(enc, 4),
                         "algorithmic_flops_per_launch": Bl * GEMM_FLOP[args.model],
                         # the WHOLE step (front-end + encoder + gaps) against the same peak: GEMM FLOPs / ms_per_step / peak
                         "step_frac": round(Bl * GEMM_FLOP[args.model] / (ms * 1e-3) / 1e12 / peak, 5)},
            "frontend": {"kernel": "mel_frontend_kernel", "launch_ms": round(fe, 4), "pass_split_pm": reps[0].get_option("fe_asym_pm").get(dev.index),
                         # algorithmic bytes: PCM in + 24 patches x 256 16-bit operands out per clip
                         "hbm_GBs": round(Bl * (T * in_bytes + 24 * 256 * 2) / (fe * 1e-3) / 1e9, 1),
                         "hbm_frac": round(Bl * (T * in_bytes + 24 * 256 * 2) / (fe * 1e-3) / 1e9 / PEAK_HBM_GBS, 5),
                         # fp32 work the kernel EXECUTES (VALU + exact-fp32 matrix pipe, both 157.3 TFLOP/s peak) ...
                         "fp32_exec_TFLOPs": round(Bl * FRONTEND_EXEC_FLOP / (fe * 1e-3) / 1e12, 2),
                         "valu_frac": round(Bl * FRONTEND_EXEC_FLOP / (fe * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 5),
                         # ... and the algorithmic (sparse-mel) count of SURVEY.md 8d against the same peak
                         "alg_frac": round(Bl * FRONTEND_ALG_FLOP / (fe * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 5)},
        }
        if world == 1 and not args.no_other_dtypes:
            # the metric's own label says bf16 (BASELINE configs #2-#4) and config #5 says fp8: both are on record next to the
            # parity dtype.  Same workload, same pipeline, 100 timed steps after 40 warm-up steps; NOT parity modes.
            line["other_dtypes"] = {}
            for prec in ("f16", "bf16", "fp8c", "fp8w", "f16x2", "f16l6", "f32"):
                if prec == eff or (prec == "f16" and eff == "f16c"):      # (f16c runs f16's kernels)
                    continue
                rr = []
                for _ in range(2):
                    r = getattr(models, args.model)(outputdim=537, target_length=102, precision=prec)
                    r.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
                    r = r.to(dev).eval()
                    r.reserve(Bl, T, dev)
                    rr.append(r)
                with torch.no_grad():
                    def steps(n):
                        st = rr[0].frontend(x)
                        for i in range(n):
                            nxt = rr[(i + 1) & 1].frontend(x) if i + 1 < n else None
                            o = rr[i & 1].encode(st)
                            st = nxt
                        return o
                    nw, nt_ = (2, 5) if prec == "f32" else (40, 100)     # (the fp32 rung: ~45 ms per step)
                    steps(nw)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    steps(nt_)
                    torch.cuda.synchronize()
                    el2 = time.perf_counter() - t0
                line["other_dtypes"][prec] = {"value": round(Bl * nt_ / el2, 1), "unit": "clips/s", "ms_per_step": round(el2 * 1e3 / nt_, 4),
                                              "steps": nt_, "parity": PARITY_NOTE[prec]}
                del rr
        if world == 1 and args.from_host:
            # PCIe-inclusive rate (never `value`): producer threads fill pinned int16 batches from pageable host memory (what the
            # reference's DataLoader workers + collate do), HostClipStreamer overlaps the H2D copy of batch i+1 with the kernels of
            # batch i.  Ceiling: 32 000 B per clip over PCIe Gen5 x16 (~63 GB/s) = ~2.0 M clips/s.
            from uit_mobile_amd.streaming import HostClipStreamer, PinnedBatchFeeder
            pool = [synth_batch(Bl, T, i * Bl, "cpu", True).numpy() for i in range(3)]       # pageable source clips
            nb_host = 48

            def fill(i, out):
                np.copyto(out, pool[i % len(pool)])
            st = HostClipStreamer(m, Bl, T)
            res = {}
            for producers in sorted({1, args.host_producers, 2 * args.host_producers}):
                with torch.no_grad():
                    for _ in st.run(PinnedBatchFeeder(fill, 3, Bl, T, producers=producers)):
                        pass
                    # buffers allocated and pinned BEFORE the clock starts (a loader does that once); the producers start with it
                    feeder = PinnedBatchFeeder(fill, nb_host, Bl, T, producers=producers, autostart=False)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in st.run(feeder):
                        pass
                    torch.cuda.synchronize()
                rate = nb_host * Bl / (time.perf_counter() - t0)
                del feeder
                res[f"producers_{producers}"] = {"clips_per_s": round(rate, 1), "pcie_GBs": round(rate * T * 2 / 1e9, 2)}
            best = max(res.values(), key=lambda r: r["clips_per_s"])
            line["from_host_int16"] = {"value": best["clips_per_s"], "unit": "clips/s", "pcie_GBs": best["pcie_GBs"],
                                       "pcie_ceiling_GBs": 63.0, "by_producers": res,
                                       "note": "PCIe-inclusive: producer threads fill pinned int16 batches (PinnedBatchFeeder), async H2D "
                                               "overlapped with compute (HostClipStreamer); never the headline value"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, sd, T, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        phase("done")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
