#!/bin/bash
# Same-box A/B of HBM traffic between library builds: tools/pmc_ab.sh out_dir libA.so libB.so ...   (rocprofv3 --pmc FETCH_SIZE, then
# WRITE_SIZE, of tools/kb.py -- UiT-XS x 8192, precision f16; separate passes, no trace domains).  Prints per library the mean of the
# encoder's and the front-end's counters over the benchmark-grid launches: FETCH_SIZE x 2 (the guide's gfx950 correction) + WRITE_SIZE.
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp KB_PREC=f16
for lib in "$@"; do
  name=$(basename $lib .so)
  export UITK_LIB_PATH=$PWD/$lib
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/${name}_$C -- python3 tools/kb.py > $OUT/${name}_$C.log 2>&1
  done
  python3 - "$OUT" "$name" <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{name}_{C}/*/*_counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "uitk" in r["Kernel_Name"] and r["Counter_Name"] == C]
        by = collections.defaultdict(list)
        for r in rows:
            by[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
        for kind in ("uit_encoder_kernel", "mel_frontend_kernel"):
            cands = {k: v for k, v in by.items() if kind in k[0]}
            if cands:
                k = max(cands, key=lambda k: len(cands[k]))
                tot[kind][C] = (sum(cands[k]) / len(cands[k]), len(cands[k]))
for kind, v in tot.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        mb = (2 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024 / 1e6
        print(f"{name:28s} {kind:22s} FETCH_SIZE {v['FETCH_SIZE'][0]:10.1f} KiB (x2)  WRITE_SIZE {v['WRITE_SIZE'][0]:9.1f} KiB  -> {mb:7.1f} MB per launch ({v['FETCH_SIZE'][1]} launches)")
PY
done
