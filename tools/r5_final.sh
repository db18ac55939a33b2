set -u
# the round's closing run on the final sources: smoke, [full GPU suite unless SKIP_SUITE=1], rocprofv3 stats + PMC of the default bench
# (-> profiles/pmc_latest.json, keyed by the kernel sources' hash), then the default bench itself and the driver's 20 / 5 settings
O=gpurun_out/r5h; mkdir -p $O
python __graft_entry__.py --smoke > $O/smoke.log 2>&1
if [ "${SKIP_SUITE:-0}" != "1" ]; then python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest_gpu.log; fi
PREC=auto bash tools/prof.sh r5h > $O/prof.log 2>&1
cp gpurun_out/prof_r5h/pmc_latest.json profiles/pmc_latest.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes > $O/evidence_driver_settings_20_5.json 2>/dev/null
tail -3 $O/smoke.log; head -c 600 $O/bench_default.json; echo; tail -4 $O/prof.log | cut -c1-160
