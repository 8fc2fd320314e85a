OMOK_PROBE_LOG=1 python tools/play_plies.py 15 4096 800 16 1 2>&1 | grep -v amdgpu.ids | tail -3
OMOK_PROBE_LOG=1 python tools/play_plies.py 9 16384 200 8 1 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests/test_gpu_fc0_format.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | grep "precision\[n=.*difference path\|passed\|failed\|Error\|error\|assert" | tail -20
