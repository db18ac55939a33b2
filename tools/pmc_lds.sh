#!/bin/bash
# LDS bank-conflict counters of the front-end with / without the pack-time bank spreading of the mel windows
export TMPDIR=/tmp
for e in "UITK_PACK_NO_BANK_SPREAD=1" "UITK_BANK_SPREAD_DEFAULT=1"; do
  O=gpurun_out/pmc_lds_$(echo $e | cut -d= -f1)
  export $e
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O -- python3 bench.py --steps 10 --warmup 3 --spinup-steps 5 --no-cpu-baseline --no-other-dtypes > $O.log 2>&1
  unset UITK_PACK_NO_BANK_SPREAD
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "uitk" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000: agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print("$e", k, {c: round(x) for c, x in m.items()}, "conflict/active = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
PY
done
