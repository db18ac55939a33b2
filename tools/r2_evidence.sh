#!/bin/bash
# Round-2 evidence run on the GPU box: every number DESIGN.md quotes for workloads other than the bench line, as logs.
O=gpurun_out/r2_evidence; mkdir -p $O
B="python3 bench.py --no-cpu-baseline"
$B --model uit_xxxs --batch-per-gpu 4096 > $O/config2_uit_xxxs_4096.json 2> $O/config2.err      # BASELINE config #2
$B --model uit_xxs > $O/uit_xxs_8192.json 2>> $O/config2.err
$B --pcm16 > $O/uit_xs_8192_int16.json 2>> $O/config2.err
$B --precision bf16 > $O/uit_xs_8192_bf16.json 2>> $O/config2.err
$B --from-host > $O/uit_xs_8192_from_host.json 2>> $O/config2.err
timeout 600 python3 tools/stress.py > $O/stress.log 2>&1
timeout 600 python3 tools/long_clip_bench.py > $O/long_clip_bench.log 2>&1
timeout 300 python3 tools/occ_test.py > $O/occ_test.log 2>&1
timeout 300 python3 tools/phase_timing.py > $O/phase_timing.log 2>&1
bash tools/prof.sh r2a > $O/prof.log 2>&1
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/prof.log; cat $O/stress.log $O/long_clip_bench.log | grep -v amdgpu; for f in $O/*.json; do echo $f; python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'), d['frontend']['launch_ms'], d['roofline']['launch_ms'])"; done
