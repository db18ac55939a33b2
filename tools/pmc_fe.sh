#!/bin/bash
# Wide PMC sweep of the front-end kernel (benchmark grid only): tools/pmc_fe.sh <tag>  ->  gpurun_out/pmc_fe_<tag>.log
# (TA_* / TCP_* groups abort rocprofv3 on this pool -- signal 6, then a hang until the call's limit -- and are not collected.)
# One rocprofv3 --pmc pass per counter group (no trace domains beside it); per-launch means of the B=8192 launches.
export TMPDIR=/tmp
TAG=${1:-x}
O=gpurun_out/pmc_fe_$TAG
mkdir -p $O
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $O/g$i -- python3 bench.py --steps 10 --warmup 3 --spinup-steps 5 --no-cpu-baseline --no-other-dtypes > $O/g$i.log 2>&1
done <<LIST
SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SALU SQ_INST_CYCLES_SALU
SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_INSTS SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_CYCLES SQ_BUSY_CU_CYCLES
GRBM_GUI_ACTIVE
LIST
python3 - <<PY > $O.log
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/g*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "uitk" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, x in v.items():
        print(f"   {c:40s} {sum(x) / len(x):16.0f}   (n={len(x)})")
PY
cat $O.log; grep -il "error\|invalid" $O/g*.log
