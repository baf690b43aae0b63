"""CPU: `python bench.py --gpus 2` launches its own ranks (a child `torch.distributed.run`, never an exec), shards BASELINE
configs[3]'s global batch over them (dist.shard_range) and relays ONE JSON line - checked with the --dry-run flag (gloo, no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    return r


def test_bench_gpus2_launches_itself_and_shards_configs3():
    r = _run(['--gpus', '2', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                              # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['dry_run'] is True
    assert out['scaling'] == 'strong' and out['config']['global_batch'] == 1024      # N > 1 default = BASELINE configs[3]
    assert out['shard_sum'] == 1024 and out['shard_rank0'] == [0, 512]
    assert out['steps'] == 2 and out['warmup'] == 1 and out['unit'] == 'utterances/s'


def test_bench_weak_scaling_when_a_per_gpu_batch_is_given_and_single_rank_needs_no_launcher():
    r = _run(['--gpus', '2', '--dry-run', '--steps', '1', '--batch', '96'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['scaling'] == 'weak' and out['config']['global_batch'] == 192 and out['shard_rank0'] == [0, 96]
    r = _run(['--dry-run', '--steps', '1'])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['ranks_seen'] == 1 and out['config']['global_batch'] == 256    # N = 1 default = configs[2]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _run(['--gpus', '4', '--dry-run'], {'RANK': '0', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_bench_gpus8_dry_run_covers_configs3_and_configs4_sharding():
    """BASELINE configs[3] (1024 utterances over 8 ranks) and configs[4] (10 000 files over 8 ranks) as the launcher + sharding + collectives
    of an 8-rank job on CPU: 128 utterances per rank, file shards of 1250 that cover the list exactly."""
    r = _run(['--gpus', '8', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 8 and out['ranks_seen'] == 8 and out['scaling'] == 'strong'
    assert out['config']['global_batch'] == 1024 and out['shard_sum'] == 1024 and out['shard_rank0'] == [0, 128]
    c4 = out['configs4']
    assert c4['files_sharded'] == 10000 and c4['shard_rank0'] == [0, 1250] and c4['batches_of_128_all_ranks'] == 8 * 10
