"""bench.py's self-launching path (`python bench.py --gpus N` without a rank environment), exercised on CPU with a stub worker:
argument forwarding, rank environment, the last-line contract, exit-code propagation, and the refusal to label a run with a
GPU count it did not use.  Replaces what train_gen.py:295 (`nn.DataParallel(model, gpu_ids)`) does for the reference: one
command, N devices."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

STUB = textwrap.dedent('''
    import json, os, sys
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    mode = os.environ.get("STUB_MODE", "ok")
    if mode == "rank1_dies" and rank == 1:
        sys.exit(7)
    if rank == 0:
        print("some library banner on stdout")
        seen = world if mode != "short_comm" else world - 1
        n = world if mode != "mislabel" else 1
        if mode != "no_json":
            print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": n, "argv": sys.argv[1:],
                              "config": {"rccl": {"world_size": world, "ranks_seen_by_allreduce": seen}}}))
''')


@pytest.fixture()
def stub(tmp_path):
    f = tmp_path / "stub_worker.py"
    f.write_text(STUB)
    return [str(f)]


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GSTVD_BENCH_ONE_GPU")}
    e.update(kw)
    return e


def test_launcher_forwards_arguments_and_relays_the_last_line(stub):
    argv = ["--gpus", "2", "--steps", "7", "--warmup", "2", "--rows-per-gpu", "10"]
    rc, line, head = bench.launch_ranks(2, argv, worker_cmd=stub, visible_gpus=2, env=_env(), timeout=300)
    assert rc == 0
    assert line["n_gpus"] == 2 and line["argv"] == argv                 # every flag reaches the ranks unchanged
    assert line["config"]["rccl"]["ranks_seen_by_allreduce"] == 2
    assert "some library banner" in head and "stub" not in head        # chatter is relayed IN FRONT of the JSON line


def test_launcher_refuses_fewer_visible_gpus_than_asked_for(stub, capfd):
    rc, line, _ = bench.launch_ranks(8, ["--gpus", "8"], worker_cmd=stub, visible_gpus=1, env=_env(), timeout=60)
    assert rc != 0 and line is None
    assert "8" in capfd.readouterr().err


@pytest.mark.parametrize("mode", ["mislabel", "short_comm", "no_json"])
def test_launcher_rejects_a_line_that_is_not_an_n_rank_measurement(stub, mode):
    rc, line, _ = bench.launch_ranks(2, ["--gpus", "2"], worker_cmd=stub, visible_gpus=2, env=_env(STUB_MODE=mode), timeout=300)
    assert rc != 0 and line is None


def test_launcher_propagates_a_rank_failure(stub):
    rc, line, _ = bench.launch_ranks(2, ["--gpus", "2"], worker_cmd=stub, visible_gpus=2, env=_env(STUB_MODE="rank1_dies"), timeout=300)
    assert rc != 0 and line is None


def test_plain_gpus_n_on_a_box_without_n_gpus_exits_non_zero_with_a_message():
    """`python bench.py --gpus 2` where fewer than 2 devices are visible (this container: none): non-zero, a clear message, no JSON
    line -- never a silent 1-GPU number labelled n_gpus 1 (bench.py:207's `or world == 1`)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300)
    assert r.returncode != 0
    assert b"--gpus 2" in r.stderr and b"visible" in r.stderr
    assert bench.parse_last_json(r.stdout.decode())[0] is None


def test_worker_refuses_a_world_size_that_differs_from_gpus():
    """Under an outer launcher with the wrong rank count the worker exits before any GPU call."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"],
                       env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and b"WORLD_SIZE = 1" in r.stderr and not r.stdout.strip()


def test_parse_last_json_and_rank_check():
    obj, head = bench.parse_last_json('noise\n{"a": 1}\n{"n_gpus": 4, "config": {"rccl": {"world_size": 4, "ranks_seen_by_allreduce": 4}}}\n\n')
    assert obj["n_gpus"] == 4 and head.endswith('{"a": 1}')
    assert bench.check_rank_count(obj, 4) is None
    assert bench.check_rank_count(obj, 8) is not None
    assert bench.check_rank_count({"n_gpus": 1, "config": {"rccl": None}}, 1) is None
    assert bench.parse_last_json("{\"a\": 1}\ntrailing text")[0] is None


def test_leg_children_get_a_clean_rank_environment_and_failures_stay_in_their_slot(monkeypatch):
    """run_leg(): the child is `bench.py <args> <extra> --leg NAME` with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* of its own and
    WITHOUT the elastic agent's variables (TORCHELASTIC_USE_AGENT_STORE made the leg's rank 0 look for the parent launcher's store
    on the leg's port: the rendezvous hung -- found in the one-GPU validation run); a failing or hanging child becomes
    {"error": ...} for rank 0 and None elsewhere, never an exception."""
    seen = {}

    class R:
        def __init__(self, rc, out):
            self.returncode, self.stdout = rc, out

    def fake_run(cmd, env=None, stdout=None, timeout=None):
        seen.update(cmd=cmd, env=env, stdout=stdout, timeout=timeout)
        return R(0, b'noise\n{"value": 5.0, "n_gpus": 2, "config": {"leg": "rows10"}}\n')

    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "abc")
    monkeypatch.setattr(subprocess, "run", fake_run)
    res = bench.run_leg("rows10", ["--rows-per-gpu", "10"], ["--gpus", "2", "--steps", "3"], rank=0, world=2, local=0, port=29612)
    assert res["value"] == 5.0
    assert seen["cmd"][-4:] == ["--rows-per-gpu", "10", "--leg", "rows10"] and "--gpus" in seen["cmd"]
    e = seen["env"]
    assert not [k for k in e if k.startswith("TORCHELASTIC_")]
    assert (e["RANK"], e["WORLD_SIZE"], e["LOCAL_RANK"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("0", "2", "0", "127.0.0.1", "29612")
    assert seen["stdout"] == subprocess.PIPE
    # a non-zero rank keeps its child's stdout off the shared terminal and reports nothing
    assert bench.run_leg("rows10", [], [], rank=1, world=2, local=1, port=29612) is None and seen["stdout"] == subprocess.DEVNULL
    # failures: exit code, missing line, timeout
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R(3, b""))
    assert "exit code 3" in bench.run_leg("x", [], [], 0, 2, 0, 1)["error"]
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R(0, b"no json here\n"))
    assert "no JSON" in bench.run_leg("x", [], [], 0, 2, 0, 1)["error"]

    def hang(*a, **k):
        raise subprocess.TimeoutExpired(cmd="bench.py", timeout=k.get("timeout"))
    monkeypatch.setattr(subprocess, "run", hang)
    assert "no result within" in bench.run_leg("x", [], [], 0, 2, 0, 1)["error"]
