# round 6: full GPU suite + default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_full
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > $O/tests.txt
tail -5 $O/tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
echo "bench rc $?"
python3 - <<'PY'
import json, os
l = json.loads(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r6_full/bench_line.json').read().strip().splitlines()[-1])
print('value', l['value'], 'ms', l['ms_per_step'])
print('box', {k: l['box'][k] for k in ('copy_GBs', 'mfma_TFLOPs', 'mfma_clock_mhz')})
for k in ('inference', 'bf16', 'image', 'image_postproc', 'cdm', 'unet_cfg1', 'hrnet_cfg5', 'dp1_forced_allreduce', 'train_e2e'):
    v = l.get(k, {})
    print(k, {kk: v[kk] for kk in v if kk in ('value', 'ms_per_step', 'ms_per_image', 'ms_per_batch', 'vs_value', 'error')}, (v.get('inference') or {}).get('value'))
print('roofline', {k: l['roofline'][k] for k in ('frac', 'ms_per_launch', 'traffic', 'mfma_busy_frac', 'clock_mhz', 'traffic_source')})
print('hrnet fp32', (l.get('hrnet_cfg5', {}).get('fp32') or {}).get('ms_per_step'))
print('cpu hrnet', l.get('cpu_baseline_hrnet', {}).get('value'))
PY
