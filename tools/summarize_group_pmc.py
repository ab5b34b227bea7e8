import csv, glob, json, sys
src = sys.argv[1]; bases = 400e6
vals = {}
for f in glob.glob(src + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stream8_kernel' in r['Kernel_Name']:
            vals.setdefault((r['Kernel_Name'].split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
kernels = sorted({k for k, _ in vals})
for kn in kernels:
    m = {c: sum(v) / len(v) for (k, c), v in vals.items() if k == kn}
    steps = bases / 64; cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(kn)
    print(json.dumps({"valu_per_step": round(m['SQ_INSTS_VALU'] / steps, 1), "salu_per_step": round(m['SQ_INSTS_SALU'] / steps, 1),
        "lds_per_step": round(m['SQ_INSTS_LDS'] / steps, 2), "kernel_cycles": round(cyc), "valu_busy": round(m['SQ_ACTIVE_INST_VALU'] * 4 / cyc / 1024, 3),
        "salu_per_clk_per_cu": round(m['SQ_INSTS_SALU'] / cyc / 256, 3), "lds_busy": round(m.get('SQ_ACTIVE_INST_LDS', 0) / cyc / 256, 3),
        "lds_conflict_frac": round(m['SQ_LDS_BANK_CONFLICT'] / max(1, m['SQ_LDS_IDX_ACTIVE']), 3), "wait_any_frac": round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 3)}))
