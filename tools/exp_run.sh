set -x
export AB_BOARDS=15
python tools/ab_lib.py omok-ai_amd/libomok_mi355x.so tools/ab/libomok_e8.so tools/ab/libomok_e1.so tools/ab/libomok_e3.so tools/ab/libomok_e6.so tools/ab/libomok_e4.so tools/ab/libomok_e7.so omok-ai_amd/libomok_mi355x.so tools/ab/libomok_e8.so > gpurun_out/exp2_ab.txt 2>&1
cat gpurun_out/exp2_ab.txt
tools/pmc_kernel.sh gpurun_out/exp2 default_fetch FETCH_SIZE 15 4096 800 16 1
tools/pmc_kernel.sh gpurun_out/exp2 default_write WRITE_SIZE 15 4096 800 16 1
OMOK_MI355X_LIB=$PWD/tools/ab/libomok_e6.so tools/pmc_kernel.sh gpurun_out/exp2 e6_fetch FETCH_SIZE 15 4096 800 16 1
OMOK_MI355X_LIB=$PWD/tools/ab/libomok_e8.so tools/pmc_kernel.sh gpurun_out/exp2 e8_fetch FETCH_SIZE 15 4096 800 16 1
grep -h "k_sib_children2\|^{" gpurun_out/exp2/*.txt
python -m pytest tests/test_gpu_headline_path.py -x -q -m gpu 2>&1 | tail -5
