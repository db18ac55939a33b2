set -u
O=gpurun_out/r5d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -s -p no:cacheprovider -k "pooling_token or full_attention_factory_pooling or calibration_acceptance or auto_escalates or random_search or wide_held_out" 2>&1 | grep -v "^$" > $O/pytest_selected.log
python tools/variant_bench.py pooling > $O/variant_bench_pooling.log 2>&1
grep -n "pooling token\|full attention + GELU\|random search\|vs the imported\|passed\|failed\|^E  " $O/pytest_selected.log | cut -c1-400; grep encoder $O/variant_bench_pooling.log
