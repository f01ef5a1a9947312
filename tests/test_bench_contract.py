"""bench.py's line as the driver reads it (the contract of the task statement): the keys of the committed round-3 line, the binding roof of
the dominant layer per precision, the path-level roofline helper, and the PMC summaries the `roofline` object quotes.  CPU only: nothing
here launches a kernel (the line under profiles/r03/ was produced on the GPU by tools/refresh_profiles.sh)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, 'profiles', 'r03', 'bench_default_line.json')) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_line_has_the_contract_keys():
    d = _line()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['dtype'] == 'fp32' and d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - d['config']['tiles_per_gpu_per_step'] / d['ms_per_step'] * 1e3) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    # fp32 mode: three bf16 MFMAs per product make the dominant layer matrix-bound (92.8 us against 67.1 us for its 537 MB)
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.3 < r['frac'] < 0.7
    assert abs(r['achieved'] - 3 * r['algorithmic_flops'] / 1e12 / (r['ms_per_launch'] / 1e3)) < 1e-6 * r['achieved']
    assert 0.95 < r['traffic'] / r['algorithmic_bytes'] < 1.1            # PMC bytes per launch vs the algorithmic 537 MB
    assert 0.3 < r['mfma_busy_frac'] < 1.0 and 1000 < r['clock_mhz'] < 2600
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['unit'] == d['unit'] and 'sample' in c


def test_pmc_summaries_agree_with_their_counters():
    for dt, esz in (('fp32', 4), ('bf16', 2)):
        with open(os.path.join(ROOT, 'profiles', 'r03', 'dominant_conv_%s_pmc.json' % dt)) as f:
            p = json.load(f)
        c = p['counters']
        assert abs(p['hbm_bytes_per_launch'] - (c['FETCH_SIZE'] * 2048 + c['WRITE_SIZE'] * 1024)) < 1.0      # FETCH_SIZE x 2 on gfx950
        assert p['algorithmic_bytes_per_launch'] == 2 * 16 * 256 * 256 * 64 * esz
        # the busy-cycle counter is exactly 32 cycles per 32x32x16 MFMA
        assert c['SQ_VALU_MFMA_BUSY_CYCLES'] == p['mfma_busy_cycles_expected_32_per_mfma']
        assert abs(p['mfma_busy_frac'] - c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)) < 1e-9


def test_path_roofline_helper():
    import bench
    r = bench.path_roofline('train', 'fp32', 16, 16.0)
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['roofline_ms'] / 16.0) < 1e-12 and r['roofline_ms'] == max(r['hbm_ms'], r['mfma_ms'])
    b = bench.path_roofline('infer', 'bf16', 64, 8.0)
    assert b['bound'] == 'hbm' and b['peak_TFLOPs'] == 2500.0 and abs(r['peak_TFLOPs'] - 2500.0 / 3) < 1e-9
