# -*- coding: utf-8 -*-
'''
`python bench.py --gpus N` as written -- without torch.distributed.run in
front -- starts the N ranks itself, as a child process, and relays rank 0's
one JSON line and the exit code (bench.py: launch_ranks).

CPU: the launcher starts the ranks and hands their failure back (there is no
GPU here: every rank stops at "flow_amd needs an AMD GPU", no JSON line).
GPU (-m gpu): two gloo ranks sharing the one GPU produce the line.
'''
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--gpus', '2', '--backend', 'gloo', '--nx', '192', '--steps', '2',
        '--warmup', '1', '--no-cpu-baseline', '--no-hbm-resident',
        '--no-fast-leg', '--spmv-reps', '5']


def _run(extra_env=None, timeout=600):
    env = dict(os.environ)
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                'MASTER_PORT'):
        env.pop(key, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + ARGS,
                          env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_launcher_relays_the_ranks_failure_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present: see the gpu test')
    out = _run()
    assert out.returncode != 0
    assert out.stdout.strip() == b''
    err = out.stderr.decode('utf-8', 'replace')
    # both ranks were started and stopped where the product path must stop
    assert 'needs an AMD GPU' in err or 'libflow_hip.so' in err, err[-2000:]


def test_under_a_launcher_the_world_size_must_match():
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'],
        env=dict(os.environ, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0'),
        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0
    assert b'WORLD_SIZE' in out.stderr


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(hip):
    out = _run(timeout=900)
    assert out.returncode == 0, out.stderr.decode('utf-8', 'replace')[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['value'] > 0.0
    assert rec['scaling'] == 'strong'


@pytest.mark.gpu
def test_weak_scaling_and_bring_up_report(hip):
    """`--weak`: two ranks, a channel twice as long (the same cells); the line
    says so, carries the collective micro-benchmark, and every rank named its
    bring-up stages on stderr."""
    global ARGS
    saved, ARGS = ARGS, ARGS + ['--weak']
    try:
        out = _run(timeout=900)
    finally:
        ARGS = saved
    err = out.stderr.decode('utf-8', 'replace')
    assert out.returncode == 0, err[-3000:]
    rec = json.loads([l for l in out.stdout.decode().splitlines()
                      if l.strip()][0])
    assert rec['scaling'] == 'weak' and rec['n_gpus'] == 2
    assert '384 x' in rec['config']['workload']          # 2 x 192 columns
    us = rec['config']['collective_us']
    assert us['used'] == 'torch' and us['torch'] > 0.0    # (gloo: no library path)
    for r in (0, 1):
        for name in ('process group (gloo)', 'first collective',
                     'first time step (prepare)', 'plateau window'):
            assert '[bench rank %d/2] stage: %s' % (r, name) in err, (r, name)


@pytest.mark.gpu
def test_a_stage_that_hangs_ends_the_rank_with_its_name(hip):
    """The watchdog of bench.py's bring-up: a rank whose process group never
    forms (its peer does not exist) gives up with exit code 3 and a line that
    names rank and stage -- it does not sit in the lease."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
         '--backend', 'gloo', '--nx', '64', '--stage-timeout', '8'],
        env=dict(os.environ, RANK='0', WORLD_SIZE='2', LOCAL_RANK='0',
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)),
        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = out.stderr.decode('utf-8', 'replace')
    assert out.returncode == 3, (out.returncode, err[-2000:])
    assert "FAILED: stage 'process group (gloo)'" in err and 'rank 0/2' in err
