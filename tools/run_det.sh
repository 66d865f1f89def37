mkdir -p gpurun_out/det
python tools/det_probe.py 4 > gpurun_out/det/p1.txt 2>/dev/null
python tools/det_probe.py 4 > gpurun_out/det/p2.txt 2>/dev/null
POISON=1 python tools/det_probe.py 4 > gpurun_out/det/p3.txt 2>/dev/null
diff gpurun_out/det/p1.txt gpurun_out/det/p2.txt | head -20; echo ==== poison; diff gpurun_out/det/p1.txt gpurun_out/det/p3.txt | head -30
