python -m pytest tests/test_hip_ops.py tests/test_hip_bwd_ops.py -q -k "proj_maxpool or maxpool" 2>&1 | tail -3
python3 tools/opbench.py projpool 8 8 112 112 96 192 20 2>&1 | grep projpool
python3 tools/opbench.py projpool 8 8 56 56 192 384 20 2>&1 | grep projpool
python3 tools/opbench.py projpool 8 8 28 28 384 768 20 2>&1 | grep projpool
