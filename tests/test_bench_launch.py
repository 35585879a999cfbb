"""`python bench.py --gpus N` launches its own ranks (VERDICT r5 "next" item 2): the multi-GPU scaling run must not depend on
the caller wrapping it in torch.distributed.run.  What it replaces: nn.DataParallel inside one process (reference
nisqa/NISQA_model.py:56-57) -> one process per GPU, clips sharded, one all_gather of the result rows."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_bench_gpus2_launches_its_own_ranks_cpu_box():
    """No GPU here: the two ranks must each be STARTED (torch.distributed.run env) and each refuse with the no-CPU-path message --
    not the old 'must be launched with torch.distributed.run' exit of the parent."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU box: the -m gpu test below runs the real thing')
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-side', '--no-cpu-baseline'])
    assert r.returncode != 0
    assert 'must be launched with' not in r.stderr
    # every rank that got as far as its first line refuses with the no-CPU-path message; the launcher ends the others as soon as one
    # rank has failed, so one message is all that is certain -- that, and the launcher's own report of a failed local rank
    assert r.stderr.count('bench.py needs an MI355X (no CPU path)') >= 1, r.stderr[-2000:]
    assert 'local_rank' in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ''                                       # no JSON line from a failed launch


@pytest.mark.gpu
def test_bench_gpus2_plain_python_one_json_line():
    """The form the driver may use for the scaling curve: plain `python bench.py --gpus 2 ...`.  Both ranks share cuda:0 (test knob,
    gloo instead of RCCL: RCCL refuses two ranks on one device); rc 0, exactly ONE JSON line, world_size_seen == 2."""
    r = _run(['--gpus', '2', '--steps', '5', '--warmup', '2', '--no-side', '--no-cpu-baseline'], {'NISQA_BENCH_SHARED_GPU': '1'})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['world_size_seen'] == 2 and d['steps'] == 5 and d['warmup'] == 2
    assert d['config']['collective_backend'] == 'gloo'                  # 'nccl' (= RCCL) when every rank has its own GPU
    assert d['scaling'] == 'weak' and d['value'] > 0 and d['unit'] == 'clips/s'


@pytest.mark.gpu
def test_bench_predict_csv_gpus2_plain_python_one_json_line():
    r = _run(['--workload', 'predict_csv', '--gpus', '2', '--clips', '1024', '--bs', '64', '--distinct', '16', '--workers', '4',
              '--warmup', '1'], {'NISQA_BENCH_SHARED_GPU': '1'})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['world_size_seen'] == 2 and d['scaling'] == 'strong'


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields():
    """One JSON line with the driver's contract keys, the roofline of the dominant kernel measured inside the timed region (only the
    two events around the CNN kernel are recorded there: an event record costs stream time, profiles/r06_stage_event_cost.txt) and
    the other stage times from the untimed second pass."""
    r = _run(['--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-side'])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'stage_ms', 'stage_ms_note'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['vs_baseline'] is None and d['dtype'].startswith('bf16x6')
    roof = d['roofline']
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    assert 0.02 < roof['frac'] < 0.2 and abs(roof['avg_launch_ms'] - d['stage_ms']['cnn_front']) < 1e-3
    assert 0.5 * d['ms_per_step'] < d['stage_ms']['cnn_front'] < d['ms_per_step']          # the dominant kernel, and it fits the step
    assert set(d['stage_ms']) == {'mel', 'cnn_front', 'cnn_back', 'selfatt', 'pool'}
    assert d['overlap_2_streams']['steps'] >= 200 and d['steady']['steps'] >= 400
