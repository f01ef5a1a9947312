# HBM bytes of one whole training step from the PMC counters (two separate passes, FETCH_SIZE x 2 on gfx950 - MI355X_MICROARCH.md):
#   bash tools/prof_step_traffic.sh [fp32|bf16] [train|infer]     (through gpurun)
# train: one step = the launches after the second-to-last adam_kernel up to the last one (16 tiles); infer: one batch of 64 tiles = the
# launches after the second-to-last tile_maps_kernel (the first launch of a batch's post-processing chain; rounds 1-5: probmaps_kernel) up to the last one (dispatch ids follow the enqueue order: batch i's post-processing is
# queued between the forwards of batches i and i + 1; under the counters the launches run one at a time).
DT=${1:-fp32}
MODE=${2:-train}
export CDNET_TRAFFIC_MARK=$([ "$MODE" = infer ] && echo tile_maps_kernel || echo adam_kernel)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/step_pmc_$c -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode $MODE --dtype $DT --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/step_pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, os, collections, json
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/'
tot = {}
per = collections.defaultdict(lambda: [0.0, 0.0])
for ci, c in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):
    rows = list(csv.DictReader(open(root + 'step_pmc_%s/t_counter_collection.csv' % c)))
    by = collections.defaultdict(float)
    name = {}
    for r in rows:
        if r['Counter_Name'] == c:
            by[int(r['Dispatch_Id'])] += float(r['Counter_Value'])
            name[int(r['Dispatch_Id'])] = r['Kernel_Name']
    ids = sorted(by)
    adam = [i for i in ids if os.environ['CDNET_TRAFFIC_MARK'] in name[i]]
    lo, hi = adam[-2], adam[-1]                      # one step: after the second-to-last Adam launch up to the last one
    kb = 0.0
    for i in ids:
        if lo < i <= hi:
            kb += by[i]
            n = name[i]
            short = (n.split('(anonymous namespace)::')[1] if '(anonymous namespace)::' in n else n).split('(')[0][:40]
            per[short][ci] += by[i]
    tot[c] = kb
fetch, write = tot['FETCH_SIZE'] * 1024 * 2, tot['WRITE_SIZE'] * 1024
print(json.dumps({'hbm_read_GB_per_step': fetch / 1e9, 'hbm_write_GB_per_step': write / 1e9, 'hbm_GB_per_step': (fetch + write) / 1e9, 'mode': os.environ['CDNET_TRAFFIC_MARK']}))
for k, v in sorted(per.items(), key=lambda kv: -(kv[1][0] * 2 + kv[1][1]))[:18]:
    print('%-42s read %7.2f GB  write %6.2f GB' % (k, v[0] * 2048 / 1e9, v[1] * 1024 / 1e9))
PY
