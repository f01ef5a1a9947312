"""Fold the passes of tools/prof_roofline_pmc.sh into one JSON (stdout): per-launch means of every counter over the dispatches of
the dominant kernel, HBM traffic (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md), the matrix pipe's busy fraction and the
effective clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration; reads high on launches shorter than ~0.3 ms, ibid.).
usage: python tools/make_roofline_pmc.py <dir> <fp32|bf16> [tiles per launch = 16]"""
import csv
import glob
import json
import os
import sys

root, dt = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
want = ('conv_ws32_kernel', 'conv_f32_kernel') if dt == 'fp32' else ('conv_ws16_kernel', 'conv_ws_kernel', 'conv_fwd_kernel')
out = {'dtype': dt, 'tiles_per_launch': B, 'counters': {}}
dur_ns = None
for f in sorted(glob.glob(os.path.join(root, 'stats', '*kernel_stats.csv'))):
    for r in csv.DictReader(open(f)):
        if any(w in r['Name'] for w in want) and int(r['Calls']) >= 30:
            out['kernel'] = r['Name']
            dur_ns = float(r['AverageNs'])
            out['avg_duration_us'] = dur_ns / 1e3
            out['calls'] = int(r['Calls'])
for f in sorted(glob.glob(os.path.join(root, 'pmc*', '*counter_collection.csv'))):
    acc = {}
    for r in csv.DictReader(open(f)):
        if any(w in r['Kernel_Name'] for w in want):
            acc.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
            acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for name, per in acc.items():
        out['counters'][name] = sum(per.values()) / len(per)
c = out['counters']
if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
    out['fetch_bytes_corrected_x2'] = c['FETCH_SIZE'] * 1024 * 2
    out['write_bytes'] = c['WRITE_SIZE'] * 1024
    out['hbm_bytes_per_launch'] = out['fetch_bytes_corrected_x2'] + out['write_bytes']
    esz = 4 if dt == 'fp32' else 2
    out['algorithmic_bytes_per_launch'] = 2 * B * 256 * 256 * 64 * esz
if 'GRBM_GUI_ACTIVE' in c and dur_ns:
    out['clock_mhz'] = c['GRBM_GUI_ACTIVE'] / 8.0 / dur_ns * 1e3
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        # busy cycles are summed over the chip's SIMDs (1024 = 256 CUs x 4); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
        out['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)
        mfmas = (3 if dt == 'fp32' else 1) * 2.0 * B * 256 * 256 * 64 * 64 * 9 / 32768.0
        out['mfma_busy_cycles_expected_32_per_mfma'] = mfmas * 32
out['note'] = ('means over the dispatches of the kernel in `bench.py --mode roofline --dtype %s --batch %d --steps 40` (incl. warm-up); separate '
               'rocprofv3 --pmc passes; FETCH_SIZE x 2 (gfx950 counts half of a 16-B/lane read stream)' % (dt, B))
print(json.dumps(out, indent=1))
