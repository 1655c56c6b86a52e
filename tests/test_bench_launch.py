"""`python bench.py --gpus N` must start its own ranks when no launcher did: the driver calls it exactly like `--gpus 1`.
The launch / rendezvous / single-JSON-line plumbing is exercised here with 2 ranks on gloo (no GPU, no model work: --launch-selftest)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest'], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # stdout carries exactly one line: rank 0's JSON
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2 and rec['all_gather_ok'] and rec['gpus_requested'] == 2


def test_bench_under_an_external_launcher_does_not_respawn():
    """the driver's N > 1 form: torch.distributed.run starts the ranks, bench.py must join them (WORLD_SIZE is set) instead of spawning again"""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch-selftest']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['world_size_seen'] == 2
