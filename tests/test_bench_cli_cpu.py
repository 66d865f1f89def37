"""bench.py's launch contract, the part that needs no GPU: `--gpus N` never silently measures fewer devices."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_over)
    return subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_present_is_an_error_not_a_smaller_run():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2)])
    assert r.returncode != 0 and not r.stdout.strip()
    assert "--gpus %d" % (have + 2) in r.stderr and "%d GPU" % have in r.stderr


def test_world_size_must_agree_with_gpus():
    r = _run(["--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and not r.stdout.strip() and "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0 and not r.stdout.strip()


def test_profile_pointer_names_committed_files():
    """profiles/CURRENT names the profile set bench.py quotes its traffic constants from; the files must exist"""
    sys.path.insert(0, ROOT)
    import bench
    tag = open(os.path.join(ROOT, "profiles", "CURRENT")).read().split()[0]
    for suffix in ("_pmc_traffic_conv3x3.json", "_pmc_step_traffic.json"):
        rel = bench._profile(suffix)
        assert rel == os.path.join("profiles", tag + suffix) and os.path.exists(os.path.join(ROOT, rel))
    assert bench._profile("_pmc_step_traffic.json", "no_such_net") is None
