set -u
# the round's evidence run on the current sources: smoke, [full GPU suite unless SKIP_SUITE=1], rocprofv3 stats + PMC of the default bench
# (-> profiles/pmc_latest.json, keyed by the kernel sources' hash), then the default bench itself and the driver's 20 / 5 settings.
#   TAG=r6a bash tools/r6_final.sh      -> gpurun_out/<TAG>/...   (copy into profiles/round6/)
TAG=${TAG:-r6a}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py --smoke > $O/smoke.log 2>&1
if [ "${SKIP_SUITE:-0}" != "1" ]; then python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v "^$" > $O/pytest_gpu.log; fi
PREC=auto bash tools/prof.sh $TAG > $O/prof.log 2>&1
cp gpurun_out/prof_$TAG/pmc_latest.json $O/pmc_latest.json
cp gpurun_out/prof_$TAG/${TAG}_pmc.json gpurun_out/prof_$TAG/steady.log $O/ 2>/dev/null
cp gpurun_out/prof_$TAG/${TAG}_steady.csv gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_dispatches.csv $O/ 2>/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes > $O/evidence_driver_settings_20_5.json 2>/dev/null
tail -3 $O/smoke.log; tail -3 $O/pytest_gpu.log 2>/dev/null; head -c 600 $O/bench_default.json; echo; tail -4 $O/prof.log | cut -c1-160
