"""Host-side logic of bench.py that an 8-GPU run depends on and no 1-GPU box exercises: which cores / memory node belong to a rank's GPU
(read from sysfs without touching the GPU runtime).  A fake sysfs tree stands for the node."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import bench  # noqa: E402


def _fake_sysfs(root, gpus):
    """gpus: list of (domain, bus, dev, fn, numa_node, cpulist); KFD node 0 is the CPU (simd_count 0)"""
    nodes = os.path.join(root, "class/kfd/kfd/topology/nodes")
    os.makedirs(os.path.join(nodes, "0"))
    open(os.path.join(nodes, "0", "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for k, (dom, bus, dev, fn, node, cpus) in enumerate(gpus):
        d = os.path.join(nodes, str(k + 1))
        os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % ((bus << 8) | (dev << 3) | fn, dom))
        p = os.path.join(root, "bus/pci/devices", "%04x:%02x:%02x.%d" % (dom, bus, dev, fn))
        os.makedirs(p)
        open(os.path.join(p, "numa_node"), "w").write("%d\n" % node)
        open(os.path.join(p, "local_cpulist"), "w").write(cpus + "\n")


def test_cpulist():
    assert bench._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert bench._parse_cpulist("") == set()


def test_gpu_locality_from_sysfs(tmp_path, monkeypatch):
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    _fake_sysfs(str(tmp_path), [(0, 0x05, 0, 0, 0, "0-31,128-159"), (0, 0x15, 0, 0, 0, "0-31,128-159"), (0, 0x85, 0, 0, 1, "64-95,192-223")])
    assert bench.gpu_host_locality(0, str(tmp_path)) == ("0000:05:00.0", 0, set(range(0, 32)) | set(range(128, 160)))
    assert bench.gpu_host_locality(2, str(tmp_path))[1] == 1
    assert bench.gpu_host_locality(3, str(tmp_path)) is None  # no such GPU: the caller runs unbound
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert bench.gpu_host_locality(0, str(tmp_path))[0] == "0000:85:00.0"
    assert bench.gpu_host_locality(1, str(tmp_path))[0] == "0000:05:00.0"


def test_no_sysfs_is_not_an_error(tmp_path):
    assert bench.gpu_host_locality(0, str(tmp_path / "nothing")) is None
    info = bench.bind_rank_to_gpu_node(0, 1)  # (one rank: never rebinds, whatever the box has)
    assert info["bound"] is False and info["cpus"] == len(os.sched_getaffinity(0))
