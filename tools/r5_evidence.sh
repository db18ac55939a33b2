set -u
O=gpurun_out/r5e; mkdir -p $O
python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest_gpu.log
PREC=auto bash tools/prof.sh r5e > $O/prof.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --model uit_xxxs --batch-per-gpu 4096 --no-cpu-baseline --no-other-dtypes > $O/evidence_config2_uit_xxxs_4096.json 2>/dev/null
python bench.py --model uit_xxs --no-cpu-baseline --no-other-dtypes > $O/evidence_uit_xxs.json 2>/dev/null
python bench.py --pcm16 --no-cpu-baseline --no-other-dtypes > $O/evidence_pcm16.json 2>/dev/null
python bench.py --from-host --no-cpu-baseline --no-other-dtypes > $O/evidence_from_host.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes > $O/evidence_driver_settings_20_5.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --all-ranks-on-device0 --batch-per-gpu 4096 --steps 50 --warmup 10 > $O/evidence_selflaunch_2ranks_one_gpu.json 2> $O/evidence_selflaunch.err
python tools/long_clip_bench.py > $O/long_clip_bench.log 2>&1
python tools/occ_test.py > $O/occ_test.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; head -c 700 $O/bench_default.json; echo; tail -5 $O/prof.log
