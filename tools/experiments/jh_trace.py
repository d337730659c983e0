"""Rounds and walk times of the device entropy decoder (a library built with -DFL_JH_TRACE prints them per workgroup):
   ABL_FILE=fl_jpeghuff_dev.hip bash tools/build_ablate.sh jhtrace:0:-DFL_JH_TRACE && FLGPU_LIB=tools/libfanlin_gpu_ablate_jhtrace.so python tools/experiments/jh_trace.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
fl = importlib.import_module("fanlin-rs_amd")
files = bench.synthetic_jpeg_files()
data = open(files[0], "rb").read()
with fl.State(device=0) as st:
    st.debug_set("device_huffman_always", 1)
    out = st.process_jpeg_pixels(data, fl.make_params(w=300, h=200))
    print("file bytes", len(data), "stats", {k: v for k, v in st.stats().items() if "jpeg" in k})
