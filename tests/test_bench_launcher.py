"""bench.py --gpus N outside torchrun starts its own workers as a child process.
CPU part: the launcher builds the torch.distributed.run command, relays rank 0's JSON line and
propagates a failing child — checked with a stand-in child.  GPU part (-m gpu): the real thing,
two ranks sharing the one GPU of the test box (CC_BENCH_ONE_DEVICE rehearsal, gloo collective)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def load_bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeChild:
    def __init__(self, lines, rc):
        self.stdout = iter(lines)
        self._rc = rc

    def wait(self):
        return self._rc


def run_launcher(monkeypatch, capsys, lines, rc, argv):
    bench = load_bench()
    seen = {}

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen['cmd'], seen['env'] = cmd, env
        return FakeChild(lines, rc)

    monkeypatch.setattr(subprocess, 'Popen', fake_popen)
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    bench.main()
    return seen, capsys.readouterr()


def test_launcher_starts_torchrun_child_and_relays_the_line(monkeypatch, capsys):
    line = json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': 4})
    seen, io = run_launcher(monkeypatch, capsys, ['some warning\n', line + '\n'], 0,
                            ['--gpus', '4', '--steps', '3', '--warmup', '1'])
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert int(cmd[cmd.index('--master-port') + 1]) > 0
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert io.out.strip() == line                      # exactly the one JSON line on stdout
    assert 'some warning' in io.err


def test_launcher_propagates_a_failing_child(monkeypatch, capsys):
    with pytest.raises(SystemExit) as err:
        run_launcher(monkeypatch, capsys, ['boom\n'], 3, ['--gpus', '2'])
    assert 'status 3' in str(err.value)


def test_launcher_requires_a_result_line(monkeypatch, capsys):
    with pytest.raises(SystemExit) as err:
        run_launcher(monkeypatch, capsys, [], 0, ['--gpus', '2'])
    assert 'without a result line' in str(err.value)


def test_launcher_parent_does_not_import_torch():
    """The parent must not touch torch / HIP before (or after) it starts the workers."""
    code = ('import sys, subprocess, types\n'
            'class C:\n'
            '    stdout = iter([\'{"metric": "x", "n_gpus": 2}\\n\'])\n'
            '    def wait(self): return 0\n'
            'subprocess.Popen = lambda *a, **k: C()\n'
            f'sys.argv = ["bench.py", "--gpus", "2"]\n'
            f'sys.path.insert(0, {ROOT!r})\n'
            'import runpy\n'
            f'runpy.run_path({os.path.join(ROOT, "bench.py")!r}, run_name="__main__")\n'
            'assert "torch" not in sys.modules, "launcher imported torch"\n'
            'assert "cellcomm_amd" not in sys.modules\n')
    env = {k: v for k, v in os.environ.items() if k != 'WORLD_SIZE'}
    proc = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=120)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert json.loads(proc.stdout.strip())['n_gpus'] == 2


def test_workload_is_named_after_the_baseline_config_the_arguments_match():
    """config.workload must not claim configs[2] for other shapes."""
    import types
    bench = load_bench()
    mk = lambda **kw: types.SimpleNamespace(**dict(dict(cells=100000, genes=30000, latent=64, k=15, arch='cont', dtype='f32'), **kw))
    assert bench.workload_name(mk(), 1) == 'BASELINE configs[2]'
    assert bench.workload_name(mk(dtype='f16'), 4) == 'BASELINE configs[2]'
    assert bench.workload_name(mk(cells=10000, genes=20000, latent=32), 1).startswith('BASELINE configs[1]')
    assert bench.workload_name(mk(cells=1000000), 8) == 'BASELINE configs[3]'
    assert bench.workload_name(mk(cells=500000, genes=33000, k=50, dtype='f16'), 8) == 'BASELINE configs[4]'
    assert bench.workload_name(mk(cells=62500, genes=33000, k=50, dtype='f16'), 1).startswith('custom')
    assert bench.workload_name(mk(cells=125000), 1).startswith('custom')
    assert bench.workload_name(mk(cells=4242), 1) == 'custom'
    assert bench.workload_name(mk(arch='cls'), 1) == 'custom'


@pytest.mark.gpu
def test_bench_gpus_2_without_two_devices_fails_fast(hip):
    """`bench.py --gpus 2` on a host with ONE GPU (no rehearsal flag): every worker leaves with a one-line reason before
    any collective, the launcher reports the failure — no hang, no re-exec.  (On a host with >= 2 GPUs the run is
    simply valid and this test has nothing to check.)"""
    import time
    if hip.cc_device_count() >= 2:
        pytest.skip('needs a one-GPU host')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CC_BENCH_ONE_DEVICE')}
    t0 = time.time()
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                           '--cells', '2000', '--genes', '2000', '--latent', '8', '--no-cpu-baseline', '--no-alt'],
                          capture_output=True, text=True, env=env, timeout=120)
    assert proc.returncode != 0
    assert time.time() - t0 < 120
    assert 'needs 2 visible GPUs' in proc.stderr, proc.stderr[-3000:]
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]


@pytest.mark.gpu
def test_bench_gpus_2_self_launch_on_one_device(hip):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(CC_BENCH_ONE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                           '--cells', '6000', '--genes', '2000', '--latent', '32', '--no-cpu-baseline', '--no-alt'],
                          capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['steps'] == 2 and res['value'] > 0
    assert res['config']['cells'] == 6000 and res['config']['workload'].startswith('custom')
    by_rank = res['stages_by_rank']                              # slowest / fastest rank per stage, for reading a bad curve
    assert {'gather', 'gemm_l1', 'host_encode_ms', 'host_allgather_ms', 'host_knn_ms'} <= set(by_rank)
    assert all(v['max'] >= v['min'] >= 0 and 0 <= v['argmax_rank'] < 2 for v in by_rank.values())
    assert res['allgather_ms'] == by_rank['host_allgather_ms']
    # parity evidence of the run itself: each rank re-encoded the first 256 rows of the OTHER rank's
    # block and found them bit-identical in the gathered encodings; 32 queries per rank bit-exact against the oracle
    pc = res['parity_check']
    assert pc['ok'] and pc['gathered_rows_bit_identical'] and pc['ranks_checked'] == 2 and not pc['errors']
    assert pc['knn_queries_bit_exact'] == pc['knn_queries_checked'] == 64
    assert 'gloo' in pc['exchange']


@pytest.mark.gpu
def test_bench_native_comm_one_rank(hip):
    """`--native-comm`: the exchange through cc_knn_l2_sharded (RCCL bound by the library).  A one-GPU box can host one
    rank only (RCCL refuses two ranks on one device); the two-rank rehearsal flag is refused with a reason."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CC_BENCH_ONE_DEVICE')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--cells', '6000', '--genes', '2000',
           '--latent', '32', '--no-cpu-baseline', '--no-alt', '--native-comm']
    proc = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    res = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{')][-1])
    pc = res['parity_check']
    assert pc['ok'] and pc['knn_queries_bit_exact'] == 32 and 'cc_knn_l2_sharded' in pc['exchange']
    assert 'cc_knn_l2_sharded' in res['config']['parallelism'] and res['value'] > 0
    env.update(CC_BENCH_ONE_DEVICE='1')
    proc = subprocess.run(cmd + ['--gpus', '2'], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode != 0 and 'needs one GPU per rank' in proc.stderr


@pytest.mark.gpu
def test_bench_gpus_2_alt_legs_on_request(hip):
    """The side legs (f16x2, f16, sparse) under two ranks, as `--alt-multi` runs them: every leg reports a value and the
    line keeps its per-rank stage table."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(CC_BENCH_ONE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                           '--cells', '5000', '--genes', '2000', '--latent', '16', '--no-cpu-baseline', '--alt-multi'],
                          capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    res = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['n_gpus'] == 2 and 'stages_by_rank' in res
    for leg in ('alt', 'alt_f16', 'alt_sparse'):
        assert res[leg]['value'] > 0 and res[leg]['roofline']['achieved'] > 0, leg
    assert 'dense_host' not in res                      # a one-GPU side number
