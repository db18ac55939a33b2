set -u
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -s -p no:cacheprovider -k "calibration_statistics or calibration_acceptance or watchdog or auto_escalates or pooling_token or full_attention_factory_pooling or full_attention_factories" 2>&1 | grep -v "^$" > $O/pytest_selected.log
for r in 1 2; do for lib in uit_mobile_amd/libuitk.so tools/ab/libuitk_lab_base.so tools/ab/libuitk_fe_dpp.so; do UITK_LIB_PATH=$PWD/$lib python tools/fe_variant_ab.py; done; done > $O/fe_dpp_transpose_ab.log 2>&1
AB_PRECS=f16 bash tools/ab_libs.sh tools/ab/libuitk_lab_base.so tools/ab/libuitk_enc_dynprio.so > $O/enc_dynprio_ab.log 2>&1
python tools/variant_bench.py pooling > $O/variant_bench_pooling.log 2>&1
python tools/variant_bench.py attention > $O/variant_bench_attention.log 2>&1
tail -12 $O/pytest_selected.log; cat $O/fe_dpp_transpose_ab.log | grep -v Warning; cat $O/enc_dynprio_ab.log; tail -14 $O/variant_bench_pooling.log; tail -5 $O/variant_bench_attention.log
