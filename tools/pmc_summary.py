"""Summarise a tools/prof.sh output dir: python tools/pmc_summary.py gpurun_out/prof_<tag> [out.json]"""
import collections, csv, glob, json, sys
base = sys.argv[1]
out = {}
ks = glob.glob(base + '/trace/*/*_kernel_stats.csv')
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    out['kernel_stats'] = [{k: r[k] for k in ('Name', 'Calls', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage')} for r in rows if 'uitk' in r['Name']]
pm = {}
for d in sorted(glob.glob(base + '/pmc_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        if 'uitk' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        pm.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
for k, v in pm.items():
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        cyc = v['GRBM_GUI_ACTIVE'] / 8.0
        v['derived_mfma_busy_frac'] = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)
        v['derived_valu_per_mfma'] = v['SQ_INSTS_VALU'] / max(v['SQ_INSTS_MFMA'], 1)
    if 'SQ_WAVE_CYCLES' in v and 'SQ_WAIT_ANY' in v:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
            v['derived_frac_' + c] = v[c] / v['SQ_WAVE_CYCLES']
    # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B? (rocprofv3: kilobytes)
    if 'FETCH_SIZE' in v:
        v['derived_fetch_MB'] = v['FETCH_SIZE'] * 1024 / 1e6
    if 'WRITE_SIZE' in v:
        v['derived_write_MB'] = v['WRITE_SIZE'] * 1024 / 1e6
out['pmc_avg_per_launch'] = pm
s = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(s)
for k, v in pm.items():
    print(k)
    for c in sorted(v):
        print(f"   {c:32s} {v[c]:.6g}")
for r in out.get('kernel_stats', []):
    print(r['Name'][:60], r['AverageNs'], r['Calls'])
