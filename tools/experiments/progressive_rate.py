"""JPEG-source request rate for baseline and progressive files of the same pictures (tools/latency/latency_probe through
flgpu_transform): python tools/experiments/progressive_rate.py [threads ...]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from PIL import Image
import synth
exe = os.path.join(ROOT, "tools", "latency", "latency_probe")
d = tempfile.mkdtemp(prefix="flgpu_prog_")
files = {"baseline": [], "progressive": []}
for i in range(4):
    img = Image.fromarray(synth.photo(1080, 1920, 3, index=3000 + i))
    for kind in files:
        p = os.path.join(d, f"{kind}{i}.jpg")
        img.save(p, "JPEG", quality=85, subsampling=2, progressive=(kind == "progressive"))
        files[kind].append(p)
for threads in [int(x) for x in sys.argv[1:]] or [1, 16, 64]:
    for kind, paths in files.items():
        r = subprocess.run([exe, str(threads), str(64 * max(threads, 4)), "1920", "1080", "w=300&h=200", "3", "0", "0", "0"] + paths, capture_output=True, text=True, timeout=300)
        if r.returncode:
            print(kind, threads, "FAILED", r.stderr[-200:]); continue
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"{kind:12s} threads {threads:3d}: {j['images_per_s']:9.1f} images/s  p50 {j['p50_ms']:.2f} ms  p99 {j['p99_ms']:.2f} ms  host CPU {j['host_cpu_ms_per_request']:.2f} ms/request  on device {j.get('entropy_decoded_on_device')}", flush=True)
