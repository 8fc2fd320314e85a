python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_headline_path.py tests/test_gpu_slots.py -x -q -m gpu 2>&1 | tail -4
AB_BOARDS=15,9 python tools/ab_lib.py tools/ab/libomok_nobatch.so omok-ai_amd/libomok_mi355x.so tools/ab/libomok_nobatch.so omok-ai_amd/libomok_mi355x.so > gpurun_out/exp12_ab.txt 2>&1; cat gpurun_out/exp12_ab.txt
for g in 64 1024; do OMOK_MI355X_LIB=$PWD/tools/ab/libomok_nobatch.so python tools/play_plies.py 15 $g 800 16 3 2>/dev/null | tail -1; python tools/play_plies.py 15 $g 800 16 3 2>/dev/null | tail -1; done
