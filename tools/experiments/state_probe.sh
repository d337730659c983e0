#!/bin/bash
# runs the resample bench continuously and dumps rocm-smi clocks/temperatures at several points in time, with the kernel time
# of consecutive short windows: shows the fast (first seconds) and the settled state of a box
cd "$(dirname "$0")/../.."
python - <<'PY' &
import importlib, os, sys, time, torch
sys.path.insert(0, os.getcwd())
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
with fl.State(device=0, profile=True) as st:
    p = fl.make_params(300, 200)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
    for _ in range(3): run(stream)
    torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 14:
        st.reset_stats()
        for _ in range(200): run(stream)
        torch.cuda.synchronize()
        s = st.stats()
        print(f"t={time.time()-t0:5.1f}s kernel_ms {s['resample_ms']/max(s['resample_launches'],1):.4f}", flush=True)
PY
BP=$!
sleep 5
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -E "clock level|Temperature|Power \(W\)" | sed 's/GPU\[0\]\s*: //' | tr -s ' ' | tr '\n' ';'; echo
  sleep 2
done
wait $BP
