python -m pytest tests -x -q -m gpu > gpurun_out/r4_t2.log 2>&1; tail -3 gpurun_out/r4_t2.log
tools/kstats_plies.sh gpurun_out/exp9 new 15 4096 800 16 2 | grep -E "==|k_round|k_softmax_scatter|k_scan|k_group|k_advance|k_sample|k_mirror"
AB_BOARDS=15,9 python tools/ab_lib.py tools/ab/libomok_prev.so omok-ai_amd/libomok_mi355x.so tools/ab/libomok_prev.so omok-ai_amd/libomok_mi355x.so > gpurun_out/exp9_ab.txt 2>&1
cat gpurun_out/exp9_ab.txt
