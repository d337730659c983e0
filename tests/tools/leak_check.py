"""One-off: contexts (with their queue lanes, pinned pools, scratch, CMYK tables) must give everything back on destroy."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import synth
from bench import load_package
fl = load_package()
img = synth.photo(360, 640, 3)
free0 = None
for i in range(60):
    st = fl.State(device=0)
    st.process_pixels(img, fl.make_params(300, 200, blur_sigma=10.0, front_end=fl.FE_JPEG))
    st.process_batch([img] * 4, [fl.make_params(100, 80)] * 4)
    st.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if i == 4: free0 = free
    if i % 10 == 9: print("iteration", i + 1, "device free MiB", free >> 20, flush=True)
print("leaked MiB over 55 create/destroy cycles:", (free0 - free) >> 20)
import resource
print("host max RSS MiB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)
