set -x
python -m pytest tests/test_hip_train.py -x -q -s 2>&1 | tail -25 > gpurun_out/r2_t1_train_tests.log
tools/prof_noside.sh r2_base_train_noside --mode train > gpurun_out/r2_base_train_noside.txt 2>&1
python3 tools/ktrace.py gpurun_out/r2_base_train_noside 7 > gpurun_out/r2_base_train_noside_shapes.txt 2>&1
tools/prof_noside.sh r2_base_fwd_noside --mode fwd > gpurun_out/r2_base_fwd_noside.txt 2>&1
python3 tools/ktrace.py gpurun_out/r2_base_fwd_noside 7 > gpurun_out/r2_base_fwd_noside_shapes.txt 2>&1
rm -rf gpurun_out/r2_base_train_noside/*/*.db gpurun_out/r2_base_fwd_noside/*/*.db
ls gpurun_out/r2_base_train_noside/*
