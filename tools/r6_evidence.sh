set -u
# the other workloads of the round's evidence (the default bench, the suite and the profiles: tools/r6_final.sh): BASELINE config #2,
# UiT-2XS, int16 PCM, PCIe-inclusive, bench.py --gpus 2 as typed (both ranks on the one GPU), long clips, enqueue cost, the variants
O=gpurun_out/${TAG:-r6e}; mkdir -p $O
python bench.py --model uit_xxxs --batch-per-gpu 4096 --no-cpu-baseline --no-other-dtypes > $O/evidence_config2_uit_xxxs_4096.json 2>/dev/null
python bench.py --model uit_xxs --no-cpu-baseline --no-other-dtypes > $O/evidence_uit_xxs.json 2>/dev/null
python bench.py --pcm16 --no-cpu-baseline --no-other-dtypes > $O/evidence_pcm16.json 2>/dev/null
python bench.py --from-host --no-cpu-baseline --no-other-dtypes > $O/evidence_from_host.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --all-ranks-on-device0 --batch-per-gpu 4096 --steps 50 --warmup 10 > $O/evidence_selflaunch_2ranks_one_gpu.json 2> $O/evidence_selflaunch.err
python tools/long_clip_bench.py > $O/long_clip_bench.log 2>&1
python tools/occ_test.py > $O/occ_test.log 2>&1
python tools/variant_bench.py pooling 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/variant_bench_pooling.log
python tools/variant_bench.py attention 8000 2>&1 | grep -v "amdgpu.ids\|Warning" > $O/variant_bench_attention_short.log
for f in $O/evidence_*.json; do echo "$(basename $f): $(head -c 260 $f)"; done; tail -n 3 $O/long_clip_bench.log; tail -n 3 $O/occ_test.log
