R=$PWD
python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
F="/tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg"
for i in 1 2 3; do
  echo "default: $($R/tools/latency/latency_probe 64 4096 1920 1080 'w=300&h=200' 3 0 0 0 $F 2>&1 | tail -1 | cut -c1-125) $($R/tools/latency/latency_probe 64 4096 1920 1080 'w=300&h=200' 3 0 0 0 $F 2>&1 | tail -1 | grep -o '"entropy_decoded_on_device": [0-9]*')"
  echo "always : $(FLGPU_DEVICE_HUFFMAN_ALWAYS=1 $R/tools/latency/latency_probe 64 4096 1920 1080 'w=300&h=200' 3 0 0 0 $F 2>&1 | tail -1 | cut -c1-125)"
done
