"""CPU tier: `python bench.py --gpus N` with no launcher starts its own N ranks.

Under test is bench.spawn_ranks / rank_env / pick_backend only -- host logic, no GPU: the children here are a tiny stand-in
script, not bench.py (which needs a GPU).  What must hold: every child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT as torch.distributed.run would set them; rank 0's stdout (the ONE JSON line) is relayed verbatim and only rank
0's; the parent exits with the worst child return code; a failing rank takes the others down instead of leaving them
waiting in a collective; and the parent never initialises a GPU (it must not even import the product package)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "fake_rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def _run_parent(tmp_path, body, n, extra_env=None):
    """spawn_ranks in a python child of its own (so that its stdout relay can be captured)."""
    script = _script(tmp_path, body)
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(%d, argv=['--gpus', '%d', '--tag', 'x'], script=%r, timeout=60))" % (ROOT, n, n, script))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("MASTER_PORT", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_rank_env_matches_the_launcher_contract():
    e = bench.rank_env({"PATH": "/bin", "WORLD_SIZE": "1"}, 3, 8, 29511)
    assert e["RANK"] == "3" and e["LOCAL_RANK"] == "3" and e["WORLD_SIZE"] == "8" and e["LOCAL_WORLD_SIZE"] == "8"
    assert e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511"
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin"


def test_spawn_relays_rank0_json_and_sets_env(tmp_path):
    body = """
        import json, os, sys
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        rec["argv"] = sys.argv[1:]
        # every rank prints; only rank 0's line may reach the parent's stdout
        print(json.dumps(rec), flush=True)
        open(os.path.join(%r, "seen_%%s" %% rec["RANK"]), "w").write(json.dumps(rec))
    """ % str(tmp_path)
    r = _run_parent(tmp_path, body, 4)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["RANK"] == "0" and rec["WORLD_SIZE"] == "4" and rec["MASTER_ADDR"] == "127.0.0.1"
    assert rec["argv"] == ["--gpus", "4", "--tag", "x"]
    ports = set()
    for k in range(4):
        seen = json.loads((tmp_path / ("seen_%d" % k)).read_text())
        assert seen["RANK"] == str(k) and seen["LOCAL_RANK"] == str(k) and seen["WORLD_SIZE"] == "4"
        ports.add(seen["MASTER_PORT"])
    assert len(ports) == 1 and int(ports.pop()) > 0


def test_spawn_returns_worst_rc_and_stops_the_other_ranks(tmp_path):
    body = """
        import os, sys, time
        r = int(os.environ["RANK"])
        if r == 2:
            sys.exit(7)
        time.sleep(600)          # a rank waiting in a collective for the one that died
    """
    import time
    t0 = time.time()
    r = _run_parent(tmp_path, body, 3)
    assert r.returncode != 0
    assert r.returncode in (7, 143) or r.returncode >= 128, r.returncode      # worst of {7, terminated ranks}
    assert time.time() - t0 < 60, "the surviving ranks were not stopped"
    assert r.stdout.strip() == ""


def test_bench_main_self_launches_when_no_launcher_env(tmp_path, monkeypatch):
    """main(): `--gpus N` (N > 1) without WORLD_SIZE in the environment goes to spawn_ranks BEFORE any GPU call; with
    WORLD_SIZE set and different from --gpus it refuses."""
    called = {}

    def fake_spawn(n, *a, **k):
        called["n"] = n
        return 0

    monkeypatch.setattr(bench, "spawn_ranks", fake_spawn)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called == {"n": 8}
    monkeypatch.setenv("WORLD_SIZE", "2")
    try:
        bench.main()
        raise AssertionError("must refuse")
    except SystemExit as e:
        assert "WORLD_SIZE=2" in str(e.code)


def test_pick_backend(monkeypatch):
    monkeypatch.delenv("LBBNN_BENCH_BACKEND", raising=False)
    assert bench.pick_backend(8, 8) == "nccl" and bench.pick_backend(1, 1) == "nccl"
    assert bench.pick_backend(2, 1) == "gloo"            # fewer cards than ranks: rehearsal, ranks share a card
    monkeypatch.setenv("LBBNN_BENCH_BACKEND", "gloo")
    assert bench.pick_backend(8, 8) == "gloo"


def test_parent_does_not_touch_the_gpu_or_the_product():
    """The self-launching parent must not initialise HIP: spawn_ranks may only use the standard library."""
    import ast
    import inspect
    src = inspect.getsource(bench.spawn_ranks) + inspect.getsource(bench.rank_env) + inspect.getsource(bench._free_port)
    names = {n.id for n in ast.walk(ast.parse(textwrap.dedent(src))) if isinstance(n, ast.Name)}
    assert "torch" not in names and "bnn_amd" not in names and "dist" not in names
