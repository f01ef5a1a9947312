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
    # (the round-3 line priced the fp32 mode's 3x MFMA work as useful: frac = 3 x flops / time / peak; since round 4 `frac` is the
    #  algorithmic one - test_roofline_frac_is_algorithmic below - and that view lives under mfma_work_frac)
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.3 < r['frac'] < 0.7
    assert 0.95 < r['traffic'] / r['algorithmic_bytes'] < 1.1            # PMC bytes per launch vs the algorithmic 537 MB
    assert 0.3 < r['mfma_busy_frac'] < 1.0 and 1000 < r['clock_mhz'] < 2600
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['unit'] == d['unit'] and 'sample' in c


def test_committed_round4_line_carries_the_algorithmic_fraction_and_live_traffic():
    """profiles/r04/bench_default_line.json (tools/refresh_profiles.sh r04 on the GPU box): `roofline.frac` is the algorithmic fraction
    (HBM-bound by SURVEY 8d's figures in both precisions), the fp32 mode's 3x MFMA work sits under `mfma_work_frac`, and `traffic` was
    measured inside that run by the PMC child passes (it must agree with the separately collected summary next to it)."""
    with open(os.path.join(ROOT, 'profiles', 'r04', 'bench_default_line.json')) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    assert d['dtype'] == 'fp32' and d['n_gpus'] == 1 and d['vs_baseline'] is None and d['config']['process_group'] is None
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['frac'] - r['algorithmic_bytes'] / 8000e9 / (r['ms_per_launch'] / 1e3)) < 1e-9 and 0.25 < r['frac'] < 0.6
    assert abs(r['mfma_work_frac'] - 3 * r['algorithmic_flops'] / 2500e12 / (r['ms_per_launch'] / 1e3)) < 1e-9
    assert r['traffic_source'].startswith('measured in this run') and 0.98 < r['traffic'] / r['algorithmic_bytes'] < 1.05
    with open(os.path.join(ROOT, 'profiles', 'r04', 'dominant_conv_fp32_pmc.json')) as f:
        p = json.load(f)
    assert abs(r['traffic'] - p['hbm_bytes_per_launch']) < 0.01 * p['hbm_bytes_per_launch']
    b = d['roofline_bf16']
    assert b['bound'] == 'hbm' and 'conv_ws16_kernel' in b['kernel'] and abs(b['frac'] - b['hbm_frac']) < 1e-12
    for k, v in d['roofline_path'].items():
        assert v['bound'] == 'hbm' and v['peak_TFLOPs'] == 2500.0 and abs(v['frac'] - v['hbm_frac']) < 1e-12, k


def test_pmc_summaries_agree_with_their_counters():
    for rnd, dt, esz in (('r03', 'fp32', 4), ('r03', 'bf16', 2), ('r04', 'fp32', 4), ('r04', 'bf16', 2)):
        with open(os.path.join(ROOT, 'profiles', rnd, 'dominant_conv_%s_pmc.json' % dt)) as f:
            p = json.load(f)
        c = p['counters']
        assert abs(p['hbm_bytes_per_launch'] - (c['FETCH_SIZE'] * 2048 + c['WRITE_SIZE'] * 1024)) < 1.0      # FETCH_SIZE x 2 on gfx950
        assert p['algorithmic_bytes_per_launch'] == 2 * 16 * 256 * 256 * 64 * esz
        # the busy-cycle counter is exactly 32 cycles per 32x32x16 MFMA
        assert c['SQ_VALU_MFMA_BUSY_CYCLES'] == p['mfma_busy_cycles_expected_32_per_mfma']
        assert abs(p['mfma_busy_frac'] - c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)) < 1e-9


def test_roofline_frac_is_algorithmic():
    """`roofline.frac` = max(algorithmic flops / dense bf16 MFMA peak, algorithmic bytes / 8 TB/s) / measured time (2*MACs of the
    layer, one read of its input + one write of its output) in BOTH precisions; the fp32 mode's three bf16 MFMAs per product are reported
    under `mfma_work_frac`, never as useful work (round-3 verdict, item 5)."""
    import bench
    flops, px = 2.0 * 16 * 256 * 256 * 64 * 64 * 9, 16 * 256 * 256
    for dt, esz, ms in (('fp32', 4, 0.1897), ('bf16', 2, 0.0814)):
        r = bench.dominant_roofline(ms, 16, dt, traffic=1.0, kernel='k')
        nbytes = px * 128 * esz
        t_m, t_h = flops / 2500e12 * 1e3, nbytes / 8000e9 * 1e3
        assert r['algorithmic_flops'] == flops and r['algorithmic_bytes'] == nbytes
        assert abs(r['frac'] - max(t_m, t_h) / ms) < 1e-12
        assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0          # 33.6 / 67.1 us of HBM against 30.9 us of MFMA
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and abs(r['frac'] - r['hbm_frac']) < 1e-12
        assert abs(r['mfma_work_frac'] - (3 if dt == 'fp32' else 1) * r['mfma_frac']) < 1e-12
        assert abs(r['mfma_frac'] - flops / (ms / 1e3) / 2500e12) < 1e-12
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'mfma_busy_frac', 'clock_mhz', 'ms_per_launch'):
            assert k in r, k
    assert abs(bench.dominant_roofline(0.1897, 16, 'fp32')['frac'] - 0.354) < 2e-3          # the round-3 driver measurement: 0.35, not 0.489


def test_path_roofline_helper():
    import bench
    r = bench.path_roofline('train', 'fp32', 16, 16.0)
    assert r['bound'] == 'hbm' and abs(r['frac'] - r['roofline_ms'] / 16.0) < 1e-12 and r['roofline_ms'] == max(r['hbm_ms'], r['mfma_ms'])
    assert r['peak_TFLOPs'] == 2500.0 and abs(r['mfma_work_frac'] - 3 * r['mfma_frac']) < 1e-12
    assert abs(r['frac'] - (1540.0 * 16 + 4 * 81.9) / 8000.0 / 16.0) < 1e-12                # 24.97 GB algorithmic per step
    b = bench.path_roofline('infer', 'bf16', 64, 8.0)
    assert b['bound'] == 'hbm' and b['peak_TFLOPs'] == 2500.0 and b['mfma_work_frac'] == b['mfma_frac']


def test_check_tiles_raises_the_references_constant_ddm_assertion():
    """pipeline.check_tiles: the device flag of infer_tiles -> the reference's assertion (test_dam.py:535); host logic, CPU tensors"""
    import pytest
    import torch
    from cdnet_amd import pipeline
    pipeline.check_tiles({'ddm_constant': torch.tensor([False, False, False])})
    with pytest.raises(AssertionError, match='constant direction-difference map'):
        pipeline.check_tiles({'ddm_constant': torch.tensor([False, True, False])})


def test_check_tiles_reads_the_ddm_kernels_minmax():
    """without a flag tensor check_tiles compares the (min, max) codes cdnet_ddm_codes left per tile"""
    import pytest
    import torch
    from cdnet_amd import pipeline
    pipeline.check_tiles({'minmax': torch.tensor([[0, 2], [0, 1]], dtype=torch.int32)})
    with pytest.raises(AssertionError, match=r'tile\(s\) \[1\]'):
        pipeline.check_tiles({'minmax': torch.tensor([[0, 2], [1, 1], [0, 1]], dtype=torch.int32)})


def test_committed_round5_line_carries_every_leg_and_its_cpu_baseline():
    """profiles/r05/bench_default_line.json: the legs the round-4 verdict asked for - the target generation alone and inside the step, BASELINE
    config 1 at its own size, the 1000x1000 post-processing, the 1-rank step with the all-reduce forced, a CPU baseline beside every rate, the
    image leg's true protocol, and the 16-bit dominant layer quoted at the 64-tile size with the PMC summary of THAT size beside it"""
    with open(os.path.join(ROOT, 'profiles', 'r05', 'bench_default_line.json')) as f:
        d = json.loads([ln for ln in f.read().splitlines() if ln.startswith('{')][-1])
    assert d['dtype'] == 'fp32' and d['n_gpus'] == 1 and d['vs_baseline'] is None and d['scaling'] == 'weak'
    for k in ('inference', 'bf16', 'image', 'image_postproc', 'cdm', 'train_e2e', 'unet_cfg1', 'dp1_forced_allreduce', 'roofline', 'roofline_bf16',
              'roofline_bf16_16tiles', 'roofline_path', 'cpu_baseline', 'cpu_baseline_infer', 'cpu_baseline_cdm', 'cpu_baseline_image_postproc',
              'cpu_baseline_unet_cfg1'):
        assert k in d, k
    assert d['image']['steps'] == 8 and d['image']['warmup'] == 2
    assert d['train_e2e']['vs_value'] >= 0.95 and d['train_e2e']['serial']['vs_value'] >= 0.95
    assert abs(d['train_e2e']['vs_value'] - d['train_e2e']['value'] / d['value']) < 1e-9
    assert d['cdm']['roofline']['bound'] == 'hbm' and d['cdm']['roofline']['algorithmic_bytes'] == 16 * 256 * 256 * 5
    assert d['unet_cfg1']['tiles_per_gpu_per_step'] == 4
    fa = d['dp1_forced_allreduce']
    assert fa['process_group'] == 'nccl, 1 rank(s)' and fa['buckets'] == fa['buckets_released_during_backward'] == 4 and 0.9 < fa['vs_value'] < 1.1
    assert fa['side_stream_probe']['group'] is True and d['config']['side_stream_probe']['group'] is False
    for k in ('cpu_baseline', 'cpu_baseline_infer', 'cpu_baseline_cdm', 'cpu_baseline_image_postproc', 'cpu_baseline_unet_cfg1'):
        assert d[k]['kind'] == 'port' and d[k]['cores'] >= 1 and 'sample' in d[k] and d[k]['value'] > 0, k
    b = d['roofline_bf16']
    assert '64 tiles' in b['kernel'] and b['algorithmic_bytes'] == 2 * 64 * 256 * 256 * 64 * 2 and 0.3 < b['frac'] < 0.6
    with open(os.path.join(ROOT, 'profiles', 'r05', 'dominant_conv_bf16_64tiles_pmc.json')) as f:
        p = json.load(f)
    assert p['tiles_per_launch'] == 64 and p['algorithmic_bytes_per_launch'] == b['algorithmic_bytes']
    assert 1.0 < p['hbm_bytes_per_launch'] / p['algorithmic_bytes_per_launch'] < 1.2          # quad requests: 1.15 x (1.32 x with pairs)


def test_committed_round6_line_is_comparable_across_boxes_and_carries_config_5():
    """profiles/r06/bench_default_line.json: what the round-5 verdict asked of the line - the box record measured in the run and the rates per unit
    of it, the dominant kernel's traffic AND matrix-pipe occupancy / clock measured live in both arithmetics, BASELINE config 5 (HRNet18_rev1 at
    its per-rank size) with its roofline and CPU baseline, config 1's roofline, the kernel instantiation that really runs at each batch size"""
    with open(os.path.join(ROOT, 'profiles', 'r06', 'bench_default_line.json')) as f:
        d = json.loads([ln for ln in f.read().splitlines() if ln.startswith('{')][-1])
    assert d['dtype'] == 'fp32' and d['n_gpus'] == 1 and d['vs_baseline'] is None and d['scaling'] == 'weak'
    box = d['box']
    assert 3000 < box['copy_GBs'] < 8000 and 800 < box['mfma_TFLOPs'] < 2500 and 1000 < box['mfma_clock_mhz'] < 2600
    assert abs(d['value_per_box_mfma'] - d['value'] / box['mfma_TFLOPs']) < 1e-9
    assert abs(d['inference_per_box_copy'] - d['inference']['value'] / box['copy_GBs']) < 1e-9
    assert abs(d['bf16']['inference_per_box_copy'] - d['bf16']['inference']['value'] / box['copy_GBs']) < 1e-9
    r = d['roofline']
    assert r['traffic_source'].startswith('measured in this run') and 'third live pass' in r['traffic_source']
    assert 0.98 < r['traffic'] / r['algorithmic_bytes'] < 1.05 and 0.4 < r['mfma_busy_frac'] < 0.9 and 1500 < r['clock_mhz'] < 2600
    b = d['roofline_bf16']
    assert b['traffic_source'].startswith('measured in this run') and '--batch 64' in b['traffic_source'] and 1.0 < b['traffic'] / b['algorithmic_bytes'] < 1.25
    assert b['kernel'].startswith('conv_ws16_kernel<64,0,false,false,0,4,4,true,false,true,true>')          # the QUAD-request instantiation at 64 tiles
    assert d['roofline_bf16_16tiles']['kernel'].startswith('conv_ws16_kernel<64,0,false,false,0,4,4,true,false,true,false>')
    h = d['hrnet_cfg5']
    assert h['tiles_per_gpu_per_step'] == 4 and h['dtype'] == 'bf16' and 20 < h['ms_per_step'] < 60 and 'fp32' in h
    rp = h['roofline_path']
    assert rp['bound'] == 'hbm' and abs(rp['algorithmic_MB_per_step'] - (3 * 3340.0 * 4 + 4 * 38.5)) < 1e-6 and abs(rp['frac'] - rp['roofline_ms'] / h['ms_per_step']) < 1e-9
    c = d['cpu_baseline_hrnet']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and '512x512' in c['sample']
    u = d['unet_cfg1']['roofline_path']
    assert abs(u['algorithmic_GFLOP_per_step'] - 4 * 289.1) < 1e-6 and 0.05 < u['frac'] < 0.3
    assert d['image_postproc']['ms_per_image'] < 0.35                     # (round 5: 0.50 ms)
    assert d['config']['side_stream_probe'].get('k', 0) == 0


def test_box_and_step_roofline_helpers():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    r = bench.step_roofline(1000.0, 8000.0, 2.0, 'bf16')
    assert r['bound'] == 'hbm' and abs(r['roofline_ms'] - 1.0) < 1e-12 and abs(r['frac'] - 0.5) < 1e-12 and r['mfma_work_per_product'] == 1
    r = bench.step_roofline(5000.0, 800.0, 4.0, 'fp32')
    assert r['bound'] == 'mfma' and abs(r['frac'] - 0.5) < 1e-12 and r['mfma_work_per_product'] == 3
