#!/bin/bash
# GPU box: shader clock and MFMA-busy of the encoder for ablation builds (timing experiments): tools/clock_probe.sh lib abl...
export TMPDIR=/tmp
lib=$1; shift
for abl in "$@"; do
  out=gpurun_out/clk_$abl
  rm -rf $out
  UITK_LIB_PATH=$PWD/$lib UITK_ENC_ABLATE=$abl rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 tools/kb1.py > $out.log 2>&1
  python3 - "$out" "$abl" <<'PY'
import csv, glob, sys, collections
d, abl = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + '/*/*_counter_collection.csv'):
    rows += [r for r in csv.DictReader(open(f)) if 'uit_encoder_kernel' in r['Kernel_Name']]
by = collections.defaultdict(dict)
for r in rows:
    by[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    if 'Start_Timestamp' in r: by[r['Dispatch_Id']]['ns'] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
v = [x for x in by.values() if 'GRBM_GUI_ACTIVE' in x]
v = v[len(v) // 2:]
import statistics as st
g = st.median(x['GRBM_GUI_ACTIVE'] for x in v) / 8
m = st.median(x['SQ_VALU_MFMA_BUSY_CYCLES'] for x in v)
ns = st.median(x.get('ns', 0) for x in v)
print(f"ablate {abl:>4}: cycles/XCD {g:9.0f}  duration {ns/1e3:7.1f} us  clock {g/ns if ns else 0:5.2f} GHz  mfma_busy {m/(g*1024):.3f}")
PY
done
