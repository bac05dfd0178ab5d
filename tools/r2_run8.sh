python -m pytest tests/test_hip_train.py tests/test_hip_ddp.py tests/test_hip_engine.py tests/test_hip_inference.py -q -m gpu 2>&1 | tail -25 > gpurun_out/r2_t8_gpu_tests.log
tools/prof_noside.sh r2_t8_train_noside --mode train > gpurun_out/r2_t8_train_noside.txt 2>&1
python3 tools/ktrace.py gpurun_out/r2_t8_train_noside 7 > gpurun_out/r2_t8_train_noside_shapes.txt 2>&1
rm -rf gpurun_out/r2_t8_train_noside
