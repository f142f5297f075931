import os, sys
sys.path.insert(0, '/root/repo')
base = "/sys/class/kfd/kfd/topology/nodes"
try:
    for n in sorted(os.listdir(base), key=int):
        props = dict(l.split(None, 1) for l in open(os.path.join(base, n, "properties")).read().splitlines() if " " in l)
        print(n, {k: props.get(k) for k in ("simd_count", "location_id", "domain", "drm_render_minor", "gpu_id")})
except Exception as e:
    print("ERR", repr(e))
import bench
print(bench.gpu_host_locality(0))
os.system("ls /sys/bus/pci/devices | head -40; ls /sys/class/drm | head")
