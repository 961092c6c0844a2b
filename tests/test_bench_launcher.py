"""bench.py --gpus N (N > 1) started WITHOUT a launcher must start its own ranks: a child
`python -m torch.distributed.run` process (never exec), before the HIP library is loaded, whose
rank-0 JSON line and exit code it forwards.  No GPU: the spawn is mocked, and one real spawn runs
a stand-in rank program through the same launcher."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_spawn_command_line_and_forwarding(capsys):
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        out = 'noise on stdout\n{"metric": "V-cycle throughput", "value": 1.0, "n_gpus": 4}\n'
        return types.SimpleNamespace(returncode=0, stdout=out)

    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], run=fake_run)
    cmd = seen["cmd"]
    assert rc == 0
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    lines = [ln for ln in out.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 4      # ONE JSON line on stdout
    assert "noise on stdout" in out.err and "starting 4 ranks" in out.err


def test_spawn_forwards_failure(capsys):
    def fake_run(cmd, **kw):
        return types.SimpleNamespace(returncode=2, stdout="")
    assert bench.spawn_ranks(2, ["--gpus", "2"], run=fake_run) == 2

    def silent(cmd, **kw):
        return types.SimpleNamespace(returncode=0, stdout="")
    assert bench.spawn_ranks(2, ["--gpus", "2"], run=silent) != 0      # exit 0 without a line is an error


def test_plain_invocation_spawns_before_loading_the_library(monkeypatch):
    """main() with --gpus 2 and no RANK in the environment goes to spawn_ranks and imports neither
    the package nor torch first."""
    called = {}

    def fake_spawn(n, argv, run=None):
        called["n"], called["argv"] = n, list(argv)
        called["pkg_loaded"] = "codes_of_ipd_ssn_amg_method_amd" in sys.modules
        return 7
    monkeypatch.setattr(bench, "spawn_ranks", fake_spawn)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    saved = sys.modules.pop("codes_of_ipd_ssn_amg_method_amd", None)
    try:
        try:
            bench.main()
            raise AssertionError("main() returned")
        except SystemExit as e:
            assert e.code == 7
    finally:
        if saved is not None:
            sys.modules["codes_of_ipd_ssn_amg_method_amd"] = saved
    assert called["n"] == 2 and called["argv"] == ["--gpus", "2", "--steps", "3"]
    assert called["pkg_loaded"] is False


def test_real_child_launch_with_a_stand_in_rank_program(tmp_path, monkeypatch, capsys):
    """One real torch.distributed.run child (2 ranks, CPU): the launcher's plumbing end to end."""
    prog = tmp_path / "rank_prog.py"
    prog.write_text(
        "import os, json\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "print('rank %d here' % r)\n"
        "if r == 0:\n"
        "    print(json.dumps({'metric': 'stand-in', 'n_gpus': w}))\n")
    monkeypatch.setattr(bench, "rank_command",
                        lambda n, argv, port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                               "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                                               "--master-port", str(port), str(prog)])
    rc = bench.spawn_ranks(2, [])
    out = capsys.readouterr()
    assert rc == 0
    assert json.loads(out.out.strip())["n_gpus"] == 2


def test_pmc_summary_is_quoted_only_for_its_own_workload():
    ns = types.SimpleNamespace(n1=1024, mask="bernoulli", rho=1.0, cycle="v")
    assert bench.pmc_summary_for(ns) is not None                    # the committed default-workload summary
    ns = types.SimpleNamespace(n1=1024, mask="tree", rho=1.0, cycle="w")
    pm = bench.pmc_summary_for(ns)
    assert pm is None or pm["_workload"]["mask"] == "tree"
