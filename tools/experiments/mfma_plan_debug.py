"""Prints the matrix-pipe plan (FLGPU_DEBUG_MFMA=1) of one request on a 1920x1080 Rgb8 source.
   FLGPU_DEBUG_MFMA=1 python tools/experiments/mfma_plan_debug.py 400 300"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
w, h = int(sys.argv[1]), int(sys.argv[2])
with fl.State() as st:
    img = np.zeros((1080, 1920, 3), np.uint8)
    out = st.process_pixels(img, fl.make_params(w, h))
    print(out.shape, st.stats()["mfma_launches"])
