"""bench.py's own launch logic on the CPU: `--gpus N` without a launcher starts N rank processes (gloo rendezvous on
127.0.0.1), the parent relays rank 0's JSON line and fails when a rank fails.  BENCH_DRY_RUN=1 replaces the GPU work of
a rank by a stand-in, so what runs here is the spawn / rendezvous / max-over-ranks / gather-by-frame path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update({"BENCH_DRY_RUN": "1"}, **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("config,scaling,frames", [("c2", "weak", 512), ("c4", "strong", 4096)])
def test_gpus_2_spawns_two_ranks(config, scaling, frames):
    p = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--config", config)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout  # ONE JSON line and nothing else (gloo's chatter goes to stderr)
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == scaling and out["config"]["dry_run"] is True
    # max over ranks of the stand-in times (rank r reports (r+1) ms per step): rank 1's
    assert abs(out["ms_per_step"] - 2.0) < 1e-6
    assert abs(out["value"] - frames * 3 / 6e-3) < 1e-3 * out["value"]


def test_single_rank_runs_in_process():
    p = run_bench("--gpus", "1", "--steps", "2", "--warmup", "0")
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["metric"].startswith("frames/sec at 1920x1200 BGR8")


def test_under_a_launcher_nothing_is_spawned():
    """With RANK / WORLD_SIZE set (torch.distributed.run's contract) bench.py is one rank: world 1 here."""
    p = run_bench("--gpus", "1", "--steps", "2", "--warmup", "0", env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_failed_rank_fails_the_parent():
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "c2", env_extra={"BENCH_DRY_RUN": "0", "HIP_VISIBLE_DEVICES": "",
                                                                                               "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0


def fake_sysfs(root, gpus):
    """A sysfs tree with one CPU node and `gpus` = [(pci address, numa node)] as KFD GPU nodes, two NUMA nodes of CPUs."""
    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)

    put(os.path.join(root, "class/kfd/kfd/topology/nodes/0/properties"), "cpu_cores_count 8\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (addr, node) in enumerate(gpus):
        dom, bus, devfn = addr.split(":")
        dev, fn = devfn.split(".")
        loc = (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn)
        put(os.path.join(root, f"class/kfd/kfd/topology/nodes/{i + 1}/properties"),
            f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain {int(dom, 16)}\n")
        put(os.path.join(root, "bus/pci/devices", addr, "numa_node"), f"{node}\n")
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus) // 2)
    put(os.path.join(root, "devices/system/node/node0/cpulist"), ",".join(map(str, cpus[:half])) + "\n")
    put(os.path.join(root, "devices/system/node/node1/cpulist"), f"{cpus[half] if len(cpus) > half else cpus[0]}-{cpus[-1]}\n")
    return cpus[:half], cpus[half:] or cpus[:1]


def test_pin_to_gpu_numa(tmp_path, monkeypatch):
    from image_processing_amd.shard import gpu_pci_addresses, pin_to_gpu_numa

    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    root = str(tmp_path)
    n0, n1 = fake_sysfs(root, [("0000:05:00.0", 0), ("0000:85:00.0", 1), ("0001:c5:00.0", -1)])
    assert gpu_pci_addresses(root) == ["0000:05:00.0", "0000:85:00.0", "0001:c5:00.0"]
    calls = []
    info = pin_to_gpu_numa(1, sysfs=root, setaffinity=lambda pid, cpus: calls.append((pid, sorted(cpus))))
    assert info == {"numa_node": 1, "cpus": len(n1), "pci": "0000:85:00.0", "pinned": True} and calls == [(0, n1)]
    # numa_node -1, a rank beyond the GPUs, no sysfs at all: nothing is touched, nothing raises
    for rank, sysfs in ((2, root), (7, root), (0, os.path.join(root, "nowhere"))):
        calls.clear()
        assert pin_to_gpu_numa(rank, sysfs=sysfs, setaffinity=lambda *a: calls.append(a))["pinned"] is False and not calls
    # a visible-devices list maps the rank to its entry
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")
    assert pin_to_gpu_numa(0, sysfs=root, setaffinity=lambda *a: None)["pci"] == "0000:85:00.0"


def test_streamed_leg_and_numa_placement_of_two_ranks(tmp_path):
    """`--config c4 --streamed --gpus 2` (dry run): each rank pins itself to its GPU's NUMA node before anything else, and
    the line carries the streamed leg's keys beside the device-resident value."""
    n0, n1 = fake_sysfs(str(tmp_path), [("0000:05:00.0", 0), ("0000:85:00.0", 1)])
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "c4", "--streamed", env_extra={"BENCH_SYSFS_ROOT": str(tmp_path)})
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["numa"]["pinned"] is True and out["numa"]["numa_node"] == 0 and out["numa"]["affinity_now"] == n0  # rank 0's line
    assert {"value", "unit", "pcie_gbps", "slots", "frames"} <= set(out["streamed"]) and out["streamed"]["frames"] == 4096
    q = run_bench("--gpus", "1", "--steps", "1", "--warmup", "0", "--config", "c4", "--streamed", "--no-numa",
                  env_extra={"BENCH_SYSFS_ROOT": str(tmp_path)})
    assert q.returncode == 0 and json.loads(q.stdout.strip().splitlines()[-1])["numa"]["pinned"] is False


def test_offline_counters_are_quoted_only_for_the_running_build(monkeypatch):
    """bench.py quotes HBM traffic / VALU counts from profiles/r0x_pmc_summary.json only when that file was collected on the
    library (or the sources) this process runs: a matching hash gives the entry, any other build gives None and the reason."""
    sys.path.insert(0, ROOT)
    import bench

    mine = bench.build_hashes()
    assert mine["src_sha16"] and len(mine["src_sha16"]) == 16
    summary = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
    pmc, src = bench.offline_counters("c2")
    if summary["c2"]["lib_sha16"] == mine["lib_sha16"] or summary["c2"]["src_sha16"] == mine["src_sha16"]:
        assert pmc and pmc["hbm_bytes_per_launch"] > 2.3e9 and src.endswith("r03_pmc_summary.json")
    else:
        assert pmc is None and "another kernel build" in src
    monkeypatch.setattr(bench, "build_hashes", lambda: {"lib_sha16": "0" * 16, "src_sha16": "1" * 16})
    pmc, why = bench.offline_counters("c2")
    assert pmc is None and "another kernel build" in why and "not quoted" in why
    pmc, why = bench.offline_counters("no-such-config")
    assert pmc is None and "does not cover" in why
