# repeats of the watchdog scenario of tests/test_gpu_parity.py::test_bench_watchdog_falls_back_to_rccl_when_the_peer_self_test_stalls (8 ranks on
# one GPU over gloo; a stall injected into rank 3's peer-gather set-up -> killed -> fresh workers with --gather rccl), printing the parity
# check of each run:   bash tools/wd_repeat.sh [with stall = 4] [without stall = 0] [extra bench flags, e.g. --gloo-device-tensors]
show='
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); pc=d.get("parity_check",{})
        print("run", d.get("max_abs_dprob"), {k: pc[k] for k in pc if k not in ("rows","what","max_abs_dprob")}, d.get("gather"), [a["seconds"] for a in d["watchdog"]["attempts"]], flush=True)'
COMMON="--gpus 8 --steps 4 --warmup 2 --batch-per-gpu 256 --backend gloo --model uit_xxxs --precision f16 --all-ranks-on-device0 --setup-timeout 100 --run-timeout 120 ${3:-}"
for i in $(seq 1 ${1:-4}); do python bench.py $COMMON --gather auto --inject-stall peer_gather:3 2>/dev/null | python -c "$show"; done
for i in $(seq 1 ${2:-0}); do python bench.py $COMMON --gather rccl 2>/dev/null | python -c "$show"; done
