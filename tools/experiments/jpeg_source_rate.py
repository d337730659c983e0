"""Rate of flgpu_transform with JPEG files as sources over caller-thread counts (tools/latency/latency_probe.c; the files are the
ones bench.py's latency_jpeg_sources leg uses).   python tools/experiments/jpeg_source_rate.py [requests] [reps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
files = bench.synthetic_jpeg_files() if os.environ.get("PROBE_PIXELS") != "1" else []
exe = os.path.join(ROOT, "tools", "latency", "latency_probe")
for threads in (8, 16, 24, 32, 64):
    for r in range(reps):
        out = subprocess.run([exe, str(threads), str(n), "1920", "1080", "w=300&h=200", "3", "0", "0", "0"] + list(files), capture_output=True, text=True, timeout=300)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"threads {threads:3d}: {d['images_per_s']:9.1f} images/s  p50 {d['p50_ms']:.2f} ms  p99 {d['p99_ms']:.2f} ms  host CPU {d.get('host_cpu_ms_per_request', 0):.2f} ms/request  failed {d['failed']}", flush=True)
        except Exception:
            print("threads", threads, "->", out.stdout[-200:], out.stderr[-200:], flush=True)
