tools/kstats_plies.sh gpurun_out/exp11 mixed 15 4096 800 16 2 5 | head -18
tools/kstats_plies.sh gpurun_out/exp11 fp6 15 4096 800 16 2 3 | head -18
for g in 64 128 160; do for dm in 0 1024 2048; do echo "games $g delta_min $dm"; OMOK_SIB_DELTA_MIN=$dm python tools/play_plies.py 15 $g 800 16 3 5 2>/dev/null | tail -1; done; done
