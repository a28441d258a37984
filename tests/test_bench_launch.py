"""bench.py's launch contract: `--gpus N` starts its own N ranks, never reports fewer ranks as N GPUs.
CPU part: the refusals that need no device.  GPU part (one-GPU test box): the un-launched 2-rank rehearsal."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, 'bench.py')


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _line(stdout):
    rows = [l for l in stdout.splitlines() if l.startswith('{')]
    assert len(rows) == 1, stdout[-2000:]
    return json.loads(rows[0])


def test_world_size_mismatch_is_refused():
    """A torchrun environment that disagrees with --gpus must not produce a number (WORLD_SIZE=1 with --gpus 8 used
    to run one rank silently)."""
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '0'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0
    assert 'refusing' in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_no_device_no_number():
    import torch
    if torch.cuda.is_available():
        pytest.skip('needs a box without a HIP device')
    for args in (['--gpus', '2'], ['--gpus', '1'], ['--train']):
        r = _run(args + ['--steps', '1', '--warmup', '0'])
        assert r.returncode != 0, args
        assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


@pytest.mark.gpu
def test_self_launch_two_ranks_on_this_box():
    """`python bench.py --gpus 2` outside torchrun: two ranks are started, both answer an all-reduce, the line says
    n_gpus 2 with both ranks listed.  The test box has one GPU, so the rehearsal shares it (--share-devices, gloo)."""
    import torch
    ndev = torch.cuda.device_count()
    args = ['--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '4', '--no-cpu-baseline', '--no-other-configs']
    r = _run(args + ([] if ndev >= 2 else ['--share-devices']), {'CTDET_TUNE': '0'})
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 8
    assert len(line['per_rank_ms_per_step']) == 2 and all(v > 0 for v in line['per_rank_ms_per_step'])
    d = line['dist']
    assert d['self_launched'] and [x['rank'] for x in d['ranks']] == [0, 1]
    if ndev >= 2:
        assert d['backend'] == 'nccl' and d['rccl_ranks'] == 2 and not d['devices_shared']
        assert len({x['device'] for x in d['ranks']}) == 2
        ids = [x['pci_bus_id'] for x in d['ranks']]
        assert None in ids or len(set(ids)) == 2, ids
    else:
        assert d['backend'] == 'gloo' and d['rccl_ranks'] == 0 and d['devices_shared']


@pytest.mark.gpu
def test_more_ranks_than_gpus_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 7
    r = _run(['--gpus', str(n), '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert 'refusing' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]


@pytest.mark.gpu
def test_train_mode_two_ranks_reports_allreduce_and_overlap():
    """bench.py --train --gpus 2 (self-launched): a data-parallel training step with the all-reduce timed alone,
    the step without it and the overlap fraction on the line (BASELINE configs[3], small shape here)."""
    import torch
    ndev = torch.cuda.device_count()
    args = ['--train', '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '300', '--batch', '2', '--classes', '20',
            '--phase', '1']
    r = _run(args + ([] if ndev >= 2 else ['--share-devices']), {'CTDET_TUNE': '0'})
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 2 and line['metric'].startswith('images/sec training step')
    ar = line['allreduce']
    assert ar['ms_alone'] > 0 and ar['buckets'] >= 1 and 0.0 <= ar['overlap_frac'] <= 1.0
    assert ar['bucket_MiB'] > 0 and ar['buckets'] * ar['bucket_MiB'] * 2 ** 20 >= line['grad_bytes']
    assert line['grad_bytes'] > 100e6 and len(line['per_rank_ms_per_step']) == 2
    d = line['dist']
    if ndev >= 2:
        assert d['backend'] == 'nccl' and d['rccl_ranks'] == 2 and len({x['device'] for x in d['ranks']}) == 2
    else:
        assert d['backend'] == 'gloo' and d['devices_shared']


@pytest.mark.gpu
def test_eight_rank_rehearsals_on_this_box():
    """The 8-rank paths the driver's SCALE run takes, rehearsed on whatever this box has (one GPU: --share-devices, gloo;
    eight: RCCL): `bench.py --gpus 8` weak scaling (bs 4 per rank), strong scaling (ONE bs-32 batch in eight shards of 4: the
    reference's DataParallel scatter, train.py:296-297) and `bench.py --train --gpus 8` (eight gradient shards, the bucket schedule
    at world 8, the global loss normaliser of multibox_loss_combined.py:119-122).  Asserts the contract of the lines, not speeds."""
    import torch
    ndev = torch.cuda.device_count()
    share = [] if ndev >= 8 else ['--share-devices']
    common = ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-other-configs']
    r = _run(['--gpus', '8', '--batch', '4'] + common + share, {'CTDET_TUNE': '0'}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 8 and line['scaling'] == 'weak' and line['config']['global_batch'] == 32
    assert len(line['per_rank_ms_per_step']) == 8 and [x['rank'] for x in line['dist']['ranks']] == list(range(8))
    assert abs(line['value'] - 32 / (line['ms_per_step'] * 1e-3)) < 0.02 * line['value']
    assert line['dist']['devices_shared'] == (ndev < 8) and line['dist']['rccl_ranks'] == (8 if ndev >= 8 else 0)
    r = _run(['--gpus', '8', '--batch', '32', '--scaling', 'strong'] + common + share, {'CTDET_TUNE': '0'}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 8 and line['scaling'] == 'strong' and line['config']['global_batch'] == 32
    assert abs(line['value'] - 32 / (line['ms_per_step'] * 1e-3)) < 0.02 * line['value']
    r = _run(['--train', '--gpus', '8', '--steps', '1', '--warmup', '1', '--size', '300', '--batch', '2', '--classes', '20', '--phase', '1']
             + share, {'CTDET_TUNE': '0'}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r.stdout)
    assert line['n_gpus'] == 8 and line['config']['global_batch'] == 16 and len(line['per_rank_ms_per_step']) == 8
    ar, model = line['allreduce'], line['allreduce_model_8gpu']
    assert ar['buckets'] >= 4 and ar['ms_alone'] > 0 and 0.0 <= ar['overlap_frac'] <= 1.0
    # the prediction a real 8-GPU run is read against: 2 (N - 1) / N of the gradient buffer through every GPU
    assert model['ranks'] == 8 and abs(model['bytes_through_each_gpu'] - 1.75 * line['grad_bytes']) < 8
    assert 0 < model['ms_per_step_seven_rings'] < model['ms_per_step_one_ring'] < 10.0
