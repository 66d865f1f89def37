mkdir -p gpurun_out/b2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b2/s4.json 2> gpurun_out/b2/s4.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --net newUNetTrans > gpurun_out/b2/unet.json 2> gpurun_out/b2/unet.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net base_transformer_pos_s4_resnet50 > gpurun_out/b2/r50_256.json 2> gpurun_out/b2/r50_256.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --net base_transformer_pos_s4_resnet50 --img 1024 --batch 8 > gpurun_out/b2/r50_1024.json 2> gpurun_out/b2/r50_1024.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net xbd_unet_transformer --img 1024 --batch 1 > gpurun_out/b2/xbd_b1.json 2> gpurun_out/b2/xbd_b1.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net xbd_unet_transformer --img 1024 --batch 4 > gpurun_out/b2/xbd_b4.json 2> gpurun_out/b2/xbd_b4.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net base_transformer_pos_s4_dd8_o5 --img 512 --batch 8 > gpurun_out/b2/o5_512.json 2> gpurun_out/b2/o5_512.err
for f in gpurun_out/b2/*.json; do echo $f; cut -c1-700 $f; done
tail -3 gpurun_out/b2/*.err | cut -c1-300
