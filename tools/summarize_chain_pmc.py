#!/usr/bin/env python3
"""Summary of tools/profile_chain.sh's counters for the chain kernel: python tools/summarize_chain_pmc.py gpurun_out/prof_chain_<tag> [bases]"""
import csv, glob, json, sys
src = sys.argv[1]
bases = float(sys.argv[2]) if len(sys.argv) > 2 else 400e6
vals = {}
for f in glob.glob(src + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stream8_kernel' in r['Kernel_Name']:
            vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in vals.items()}
steps = bases / 64
cyc = m['GRBM_GUI_ACTIVE'] / 8            # summed over the 8 XCDs
out = {"valu_per_step": round(m['SQ_INSTS_VALU'] / steps, 1), "salu_per_step": round(m['SQ_INSTS_SALU'] / steps, 1),
       "lds_per_step": round(m['SQ_INSTS_LDS'] / steps, 2), "kernel_cycles": round(cyc),
       "valu_inst_per_clk_per_simd": round(m['SQ_INSTS_VALU'] / cyc / 1024, 3),
       "valu_busy": round(m['SQ_ACTIVE_INST_VALU'] * 4 / cyc / 1024, 3), "salu_per_clk_per_cu": round(m['SQ_INSTS_SALU'] / cyc / 256, 3),
       "lds_busy": round(m.get('SQ_ACTIVE_INST_LDS', 0) / cyc / 256, 3), "wait_any_frac": round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 3)}
for f in glob.glob(src + '/trace/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'stream8_kernel' in r['Name']:
            out["kernel"] = r['Name'].split('(')[0].replace('void ', ''); out["calls"] = int(r['Calls']); out["average_ns"] = float(r['AverageNs'])
print(json.dumps(out, indent=1))
