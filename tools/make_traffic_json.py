"""Turn the two rocprofv3 PMC passes over `bench.py --mode roofline` (FETCH_SIZE, WRITE_SIZE - separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/<round>/dominant_conv_traffic.json.
usage: python tools/make_traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv
import json
import sys


def mean_counter(path, name):
    vals = {}
    kn = None
    for r in csv.DictReader(open(path)):
        if ('conv_ws_kernel' in r['Kernel_Name'] or 'conv_fwd_kernel' in r['Kernel_Name']) and r['Counter_Name'] == name:
            vals.setdefault(r['Dispatch_Id'], 0.0)
            kn = r['Kernel_Name']
            vals[r['Dispatch_Id']] += float(r['Counter_Value'])
    v = list(vals.values())
    return sum(v) / len(v), kn


fetch_kb, kname = mean_counter(sys.argv[1], 'FETCH_SIZE')
write_kb, _ = mean_counter(sys.argv[2], 'WRITE_SIZE')
fetch = fetch_kb * 1024 * 2          # gfx950: FETCH_SIZE counts half of a 16 B/lane coalesced read stream
write = write_kb * 1024
out = {
    'kernel': kname + ' 3x3 64->64 @256x256 x16 tiles',
    'FETCH_SIZE_KB_mean': fetch_kb, 'WRITE_SIZE_KB_mean': write_kb,
    'fetch_bytes_corrected_x2': fetch, 'write_bytes': write,
    'hbm_bytes_per_launch': fetch + write,
    'algorithmic_bytes_per_launch': 2 * 16 * 256 * 256 * 64 * 2,
    'note': 'gfx950: FETCH_SIZE reports half of a 16-B/lane coalesced read stream (MI355X_MICROARCH.md, HBM section); '
            'separate --pmc passes; LDS-DMA weight loads hit L2',
}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out))
